#!/usr/bin/env python3
"""Is C2 limited by wave-round quantisation?  Same scene, different (resolution, spp) splits of similar work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta
from path_tracer_amd import workloads as W
rt = pta.RayTracer(0, 0, device=0)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1()
for (w, h, spp) in [(1920, 1080, 1024), (1920, 1080, 4096), (3840, 2160, 256), (3840, 2160, 1024), (960, 540, 4096), (1920, 1080, 256), (2560, 1440, 576), (1600, 900, 1475)]:
    rt.set_active_camera(W.default_camera(w, h))
    best = 1e9
    for _ in range(3):
        rt.render(scene, w, h, False, spp=spp)
        best = min(best, rt.last_kernel_ms()[0])
    waves = ((w + 7) // 8) * ((h + 7) // 8)
    print(f"{w}x{h} {spp:5d} spp: {best:8.3f} ms  {w*h*spp/best/1e6:7.1f} Gsamples/s  waves {waves}", flush=True)
rt.close()
