#!/usr/bin/env python3
"""The frontend's per-redraw pattern (frontend/src/app.rs:84-99): prepare_pixels (1 more sample per pixel) followed by
get_output (resolve + copy to the host), through the Python mirror of the reference API.  Prints per-frame latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta
from path_tracer_amd import workloads as W
rt = pta.RayTracer(0, 0, device=0)
for (w, h) in [(800, 450), (1920, 1080), (3840, 2160)]:
    scene = W.scene_s1()
    rt.set_active_camera(W.default_camera(w, h))
    rt.render_updated(scene, w, h)
    rt.get_output()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        rt.prepare_pixels(scene, w, h)
    t1 = time.perf_counter()
    for _ in range(n):
        rt.prepare_pixels(scene, w, h)
        rt.get_output()
    t2 = time.perf_counter()
    k = rt.last_kernel_ms()[0]
    print(f"{w}x{h}: prepare_pixels {1e3*(t1-t0)/n:.3f} ms/frame (kernel {k:.3f} ms), prepare_pixels + get_output {1e3*(t2-t1)/n:.3f} ms/frame "
          f"-> {n/(t2-t1):.0f} fps, {rt.samples_accumulated()} spp accumulated", flush=True)
rt.close()
