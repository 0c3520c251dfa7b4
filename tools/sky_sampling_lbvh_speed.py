"""Throughput of the LBVH integrator with and without sky importance sampling on the 10k-sphere scene under a sky image.
usage (GPU box): python tools/sky_sampling_lbvh_speed.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402

w, h, spp = 1920, 1080, 64
scene = W.scene_s2(skybox=W.synthetic_sky(sun_scale=4000.0))
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.camera_s2(w, h))
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
for name, mode in (("bsdf", _capi.IR_SKY_SAMPLING_BSDF), ("mis", _capi.IR_SKY_SAMPLING_MIS)):
    rt.set_sky_sampling(mode)
    best = 1e9
    for _ in range(3):
        rt.render(scene, w, h, False, spp=spp)
        best = min(best, rt.last_kernel_ms()[0])
    print(f"{name}: {best:.2f} ms, {w * h * spp / best / 1e6:.2f} Gsamples/s", flush=True)
rt.close()
