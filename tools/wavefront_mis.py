#!/usr/bin/env python3
"""Regeneration against the wavefront schedule with sky importance sampling on (IR_SKY_SAMPLING_MIS): C5's scene (S1 under
the sun-dominated sky) and the 10 k-sphere scene under the same sky, 256 spp at 1920x1080; device time of the launch
sequences (HIP events), accumulators compared bit for bit.  usage: tools/wavefront_mis.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402

w, h, spp = 1920, 1080, 256
sky = W.synthetic_sky(1024, 512, sun_scale=4000.0)
cases = [("C5 scene (S1, env-lit)", W.scene_s1(emitter=False, skybox=sky), W.default_camera(w, h)),
         ("10 k spheres under the same sky (LBVH)", W.scene_s2(skybox=sky), W.camera_s2(w, h))]
for name, scene, cam in cases:
    res = {}
    for label, sched, paths, block in (("regeneration", _capi.IR_SCHEDULE_REGENERATION, 0, 128),
                                       ("wavefront, 32 Mi paths per batch", _capi.IR_SCHEDULE_WAVEFRONT, 33554432, 128),
                                       ("wavefront, one batch", _capi.IR_SCHEDULE_WAVEFRONT, 536870912, 256)):
        rt = pta.RayTracer(0, 0, device=0)
        rt.set_active_camera(cam)
        rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
        rt.set_sky_sampling(_capi.IR_SKY_SAMPLING_MIS)
        rt.set_sample_block(block)
        rt.set_schedule(sched, paths)
        best = 1e30
        for _ in range(2):
            rt.render(scene, w, h, False, spp=spp)
            best = min(best, rt.get_last_render_time() * 1e3)
        res[label] = (best, rt.get_radiance()[0].tobytes() if block == 128 else None)
        rt.close()
        print(f"{name:42s} {label:34s} {best:8.2f} ms = {w * h * spp / best / 1e6:6.1f} Gsamples/s", flush=True)
    print(f"{name:42s} regeneration and wavefront (same 128-sample blocks) bit-identical: "
          f"{res['regeneration'][1] == res['wavefront, 32 Mi paths per batch'][1]}", flush=True)
