#!/usr/bin/env python3
"""C2 render time vs samples per accumulation block (ir_set_sample_block)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta
from path_tracer_amd import workloads as W
rt = pta.RayTracer(0, 0, device=0)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1()
for (w, h, spp) in [(1920, 1080, 1024), (3840, 2160, 1024), (800, 450, 64)]:
    rt.set_active_camera(W.default_camera(w, h))
    for block in (16, 32, 64, 128, 256, 512, 1024):
        rt.set_sample_block(block)
        best = 1e9
        for _ in range(3):
            rt.render(scene, w, h, False, spp=spp)
            best = min(best, rt.last_kernel_ms()[0])
        print(f"{w}x{h} {spp:5d} spp block {block:5d}: {best:8.3f} ms  {w*h*spp/best/1e6:7.1f} Gsamples/s", flush=True)
rt.close()
