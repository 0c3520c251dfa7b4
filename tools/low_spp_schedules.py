"""Kernel time of low-spp launches (the interactive pattern, app.rs:84-99) on the 10k-sphere scene: regeneration vs
wavefront schedule.  usage (GPU box): python tools/low_spp_schedules.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402

w, h = 1920, 1080
scene = W.scene_s2()
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.camera_s2(w, h))
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
for spp in (1, 2, 4, 8, 16, 64):
    row = []
    for name, sch in (("regeneration", _capi.IR_SCHEDULE_REGENERATION), ("wavefront", _capi.IR_SCHEDULE_WAVEFRONT)):
        rt.set_schedule(sch)
        best = 1e9
        for _ in range(5):
            rt.render(scene, w, h, False, spp=spp)
            best = min(best, rt.last_kernel_ms()[0])
        row.append(f"{name} {best:.3f} ms ({w * h * spp / best / 1e6:.2f} Gsamples/s)")
    print(f"{spp:>3} spp: " + ", ".join(row), flush=True)
rt.close()
