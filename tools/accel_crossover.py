#!/usr/bin/env python3
"""LDS scan vs LBVH as a function of the sphere count (S2-style scenes, 1280x720, 64 spp, depth 8): where should
IR_ACCEL_AUTO switch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta
from path_tracer_amd import _capi, workloads as W
w, h, spp = 1280, 720, 64
rt = pta.RayTracer(0, 0, device=0)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
rt.set_active_camera(W.camera_s2(w, h))
for n in (4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096):
    scene = W.scene_s2(n=n)
    res = {}
    for name, acc in (("scan", _capi.IR_ACCEL_LINEAR), ("lbvh", _capi.IR_ACCEL_LBVH)):
        rt.set_accel(acc)
        best = 1e9
        for _ in range(2):
            rt.render(scene, w, h, False, spp=spp if n <= 1024 else 8)
            best = min(best, rt.last_kernel_ms()[0])
        res[name] = w * h * (spp if n <= 1024 else 8) / best / 1e6
    print(f"n={n:5d}  scan {res['scan']:8.2f} Gsamples/s   lbvh {res['lbvh']:8.2f} Gsamples/s", flush=True)
rt.close()
