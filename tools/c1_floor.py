"""What a path-kernel launch costs before it does any work: the C1 frame with a camera that sees only sky (zero tasks),
then 1, 2, 4, 8 samples per pixel in one block each.  usage: python tools/c1_floor.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

w, h = 800, 450
rt = pta.RayTracer(0, 0, device=0)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1()


def best(spp, reps=6):
    b = 1e9
    for _ in range(reps):
        rt.render(scene, w, h, False, spp=spp)
        b = min(b, rt.last_paths_kernel_ms())
    return b, rt.last_kernel_ms()[0]


cam = W.default_camera(w, h)
up = pta.PinholeCamera(np.array([0.0, 50.0, 2.0], np.float32), np.array([1.5, 0.0, 0.0], np.float32), 35.0, 55.0, [w, h])
rt.set_active_camera(up)
rt.set_sample_block(8)
p, s = best(8)
hits = int((rt.get_primary_hits() >= 0).sum())
print(f"camera looking up ({hits} path pixels): path kernel {p*1e3:.1f} us, launch sequence {s*1e3:.1f} us")
rt.set_active_camera(cam)
for spp in (1, 2, 4, 8, 16):
    rt.set_sample_block(max(spp, 1))
    p, s = best(spp)
    print(f"C1 frame, {spp:2d} spp in one block per pixel: path kernel {p*1e3:.1f} us, launch sequence {s*1e3:.1f} us")
rt.close()
