"""Lane activity of the integrators' loops (ir_get_lane_profile) on C2 (scan) or C3 (LBVH): python tools/lane_profile.py [spp] [c2|c3] [sample block, 0 = auto]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
spp = int(sys.argv[1]) if len(sys.argv) > 1 else (1024 if cfg == "c2" else 256)
w, h = 1920, 1080
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.default_camera(w, h) if cfg == "c2" else W.camera_s2(w, h))
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1() if cfg == "c2" else W.scene_s2()
if len(sys.argv) > 3:
    rt.set_sample_block(int(sys.argv[3]))
    print("sample block", rt.get_sample_block(w, h, spp))
rt.enable_stats()
rt.render(scene, w, h, False, spp=spp)
st = rt.get_stats()
pr = rt.get_lane_profile()
print("samples %d traces/sample %.3f" % (st["samples"], st["traces"] / st["samples"]))
if cfg != "c2":
    ast = rt.get_accel_stats()
    print("box tests/trace %.1f leaf sphere tests/trace %.2f" % (ast["bvh_node_visits"] / st["traces"], ast["bvh_leaf_tests"] / st["traces"]))
    pr = dict(zip(("rounds", "shade", "inner", "leaf"), pr.values()))
for k, (waves, lanes) in pr.items():
    print("%-9s wave executions %12d  active lanes %14d  utilisation %.3f  share of loop iterations %.3f" %
          (k, waves, lanes, lanes / max(1, 64 * waves), waves / max(1, list(pr.values())[0][0])))
rt.close()
