"""C1 (800x450, 64 spp) lane activity per sample-block size: wave iterations of the path kernel, their lane utilisation,
and the path kernel's duration without statistics.  usage: python tools/c1_lanes.py [blocks...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

blocks = [int(x) for x in sys.argv[1:]] or [128, 32, 16, 8, 4]
w, h, spp = 800, 450, 64
for blk in blocks:
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_active_camera(W.default_camera(w, h))
    rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
    rt.set_sample_block(blk)
    scene = W.scene_s1()
    best = 1e9
    for _ in range(6):
        rt.render(scene, w, h, False, spp=spp)
        best = min(best, rt.last_paths_kernel_ms())
    seq = rt.last_kernel_ms()[0]
    rt.enable_stats()
    rt.render(scene, w, h, False, spp=spp)
    st = rt.get_stats()
    pr = rt.get_lane_profile()
    top = list(pr.values())[0]
    print("block %3d: paths kernel %.3f ms (sequence %.3f ms); wave iterations %d, lane utilisation %.3f, path samples %d, "
          "lane iterations per path sample %.2f, us per wave iteration at 7168 resident waves %.2f" %
          (blk, best, seq, top[0], top[1] / max(1, 64 * top[0]), st["samples"], top[1] / max(1, st["samples"]),
           best * 1e3 / max(1.0, top[0] / 7168.0)), flush=True)
    rt.close()
