#!/usr/bin/env python3
"""Minimal driver for rocprofv3 passes: renders one configuration a few times through the C ABI (no torch).
usage: [IR_PROF_SKY=WxH] [IR_PROF_FILTER=bilinear] prof_driver.py [c1|c2|c3|c4|c5] [reps] [spp] [sample_block] [regen|wavefront] [max_paths]"""
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
elif cfg in ("c4", "c5"):
    # IR_PROF_SKY=WxH: size of the synthetic sky (8192x4096 = 537 MB on the device, HBM-resident)
    sw, sh = (int(v) for v in os.environ.get("IR_PROF_SKY", "1024x512").split("x"))
    if cfg == "c4":
        w, h, spp, scene, cam = 3840, 2160, 4096, W.scene_s1(skybox=W.synthetic_sky(sw, sh)), W.default_camera(3840, 2160)
    else:
        w, h, spp, scene, cam = 1920, 1080, 4096, W.scene_s1(emitter=False, skybox=W.synthetic_sky(sw, sh, sun_scale=4000.0)), W.default_camera(1920, 1080)
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
if os.environ.get("IR_PROF_FILTER") == "bilinear":
    from path_tracer_amd import _capi as _c
    rt.set_sky_filter(_c.IR_SKY_BILINEAR)
for _ in range(reps):
    rt.render(scene, w, h, False, spp=spp)
    ms, n = rt.last_kernel_ms()
    print(f"{cfg} {w}x{h} {spp}spp: {ms:.3f} ms in {n} launches -> {w*h*spp/ms/1e3:.1f} Msamples/s", flush=True)
rt.close()
