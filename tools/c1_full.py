"""C1 (S1 default+emitter scene, 800x450, 64 spp = 23.04 M samples) in full on both sides: the GPU path and
the CPU restatement of the reference (4 worker threads + 64x64 tiles + main-thread merge as renderer.rs:45,105-159, and
all host cores), same seed, same samples; checks parity of that very frame on the way.
usage (GPU box): python tools/c1_full.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

w, h, spp, B, R = 800, 450, 64, 8, 2
scene = W.scene_s1()
cam = W.default_camera(w, h)
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(cam)
rt.set_integrator(bounces=B, max_compulsory_bounces=R, seed=W.RNG_SEED)
rt.render(scene, w, h, False, spp=spp)
best_wall, best_kernel = 1e9, 1e9
for _ in range(10):
    t0 = time.perf_counter()
    rt.render(scene, w, h, False, spp=spp)
    best_wall = min(best_wall, time.perf_counter() - t0)
    best_kernel = min(best_kernel, rt.last_kernel_ms()[0] * 1e-3)
s, c = rt.get_radiance()
n = w * h * spp
print(f"C1 {w}x{h}x{spp} = {n / 1e6:.2f} M samples, depth {B}")
print(f"GPU: kernel {best_kernel * 1e3:.3f} ms ({n / best_kernel / 1e6:.0f} Msamples/s), render() wall {best_wall * 1e3:.3f} ms "
      f"({n / best_wall / 1e6:.0f} Msamples/s)")
ocam = O.camera_from(cam, w, h)
cores = min(16, os.cpu_count() or 1)
for threads, label in ((4, "4 worker threads (the reference's pool size)"), (cores, f"{cores} threads")):
    os_, oc, secs, _ = O.render(ocam, scene, spp, bounces=B, rr_start=R, seed=W.RNG_SEED, math_mode=0, threads=threads)
    print(f"CPU restatement, libm, {label}: {secs * 1e3:.1f} ms ({n / secs / 1e6:.1f} Msamples/s)  -> GPU kernel is {secs / best_kernel:.0f}x")
d = s[..., :3].astype(np.float64) - os_[..., :3].astype(np.float64)
print(f"rel-RMSE GPU vs CPU (libm) on this frame: {np.sqrt((d ** 2).sum(-1).mean()) / np.sqrt((os_[..., :3].astype(np.float64) ** 2).sum(-1).mean()):.2e}")
om, _, _, _ = O.render(ocam, scene, spp, bounces=B, rr_start=R, seed=W.RNG_SEED, math_mode=1, threads=cores)
print("bit-identical to the CPU restatement with the device's polynomials:", bool(np.array_equal(s.view(np.uint32), om.view(np.uint32))))
rt.close()
