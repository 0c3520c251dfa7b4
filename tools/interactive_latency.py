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
if '--frames-only' in sys.argv:
    sys.exit(0)

# The editor's pattern (frontend/src/viewport.rs:29-52 mutates the spheres, app.rs:84-99 redraws): every frame moves the
# spheres, restarts the accumulation (render_updated) and presents.  The LBVH is rebuilt (or refitted) per frame.
import numpy as np
from path_tracer_amd import _capi
rt = pta.RayTracer(0, 0, device=0)
w, h = 1920, 1080
for n_spheres in (10000, 100000, 1000000):
    base = W.scene_s2(n=n_spheres)
    rt.set_active_camera(W.camera_s2(w, h))
    for name, mode in (("host build", _capi.IR_BUILD_HOST), ("device build", _capi.IR_BUILD_DEVICE), ("device refit", _capi.IR_BUILD_REFIT)):
        if mode == _capi.IR_BUILD_HOST and n_spheres > 100000:
            continue
        rt.set_accel_build(mode)
        frames = 30 if n_spheres <= 100000 else 10
        scenes = []
        for k in range(frames + 1):
            xyzr = base.xyzr.copy()
            xyzr[1:, 0] += np.float32(0.002 * k)
            scenes.append(W.FlatScene(xyzr, base.material_id, base.materials9, default_sky_color=(0.6, 0.7, 0.9), skybox=None))
        rt.render_updated(scenes[0], w, h)
        rt.get_output()
        t0 = time.perf_counter()
        build = 0.0
        for k in range(1, frames + 1):
            rt.render_updated(scenes[k], w, h)
            rt.get_output()
            build += rt.get_accel_build_info()["seconds"]
        t1 = time.perf_counter()
        print(f"edit {n_spheres} spheres per frame, 1080p, 1 spp, {name}: {1e3 * (t1 - t0) / frames:.2f} ms/frame "
              f"({frames / (t1 - t0):.0f} fps), of which LBVH {1e3 * build / frames:.3f} ms, kernel {rt.last_kernel_ms()[0]:.3f} ms", flush=True)
rt.close()
