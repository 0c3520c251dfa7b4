"""Path-kernel duration on the C1 frame (800x450) as the number of tasks grows at a fixed sample-block size: the slope is
the steady-state cost per sample at that block size, the intercept is what a short launch pays for starting and draining.
usage: python tools/c1_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

w, h = 800, 450
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.default_camera(w, h))
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1()
for blk in (4, 16, 64):
    rt.set_sample_block(blk)
    pts = []
    for spp in (16, 32, 64, 128, 256, 512, 1024):
        best = 1e9
        for _ in range(5):
            rt.render(scene, w, h, False, spp=spp)
            best = min(best, rt.last_paths_kernel_ms())
        pts.append((spp, best))
    (s0, t0), (s1, t1) = pts[-2], pts[-1]
    slope = (t1 - t0) / (s1 - s0)
    print("block %3d: " % blk + "  ".join("%d spp %.3f ms" % p for p in pts))
    print("           slope %.3f us per spp (= %.1f Gsamples/s steady state); intercepts: " % (slope * 1e3, w * h / slope / 1e6)
          + "  ".join("%d spp %.3f ms" % (s, t - slope * s) for s, t in pts), flush=True)
rt.close()
