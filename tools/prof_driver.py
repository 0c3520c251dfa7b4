#!/usr/bin/env python3
"""Minimal driver for rocprofv3 passes: renders one configuration a few times through the C ABI (no torch).
usage: prof_driver.py [c1|c2|c3] [reps] [spp] [sample_block] [regen|wavefront] [max_paths]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if cfg == "c1":
    w, h, spp, scene, cam = 800, 450, 64, W.scene_s1(), W.default_camera(800, 450)
elif cfg == "c3":
    w, h, spp, scene, cam = 1920, 1080, 4, W.scene_s2(), W.camera_s2(1920, 1080)
else:
    w, h, spp, scene, cam = 1920, 1080, 1024, W.scene_s1(), W.default_camera(1920, 1080)
if len(sys.argv) > 3:
    spp = int(sys.argv[3])
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(cam)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
if len(sys.argv) > 4:
    rt.set_sample_block(int(sys.argv[4]))
if len(sys.argv) > 5:
    from path_tracer_amd import _capi
    rt.set_schedule({"regen": _capi.IR_SCHEDULE_REGENERATION, "wavefront": _capi.IR_SCHEDULE_WAVEFRONT}[sys.argv[5]],
                    int(sys.argv[6]) if len(sys.argv) > 6 else 0)
for _ in range(reps):
    rt.render(scene, w, h, False, spp=spp)
    ms, n = rt.last_kernel_ms()
    print(f"{cfg} {w}x{h} {spp}spp: {ms:.3f} ms in {n} launches -> {w*h*spp/ms/1e3:.1f} Msamples/s", flush=True)
rt.close()
