"""Kernel time of low-spp launches on the S1 scene (brute force), accumulating (the frontend's prepare_pixels) — run with
IR_DEBUG_DIRECT=0 (always list + tasks) and without it (a lane per pixel whenever the launch is one block per pixel; the
default block is 128 samples) to see where the two meet.  SPPS=1,2,.. picks the sample counts.  usage (GPU box): python tools/direct_crossover.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

rt = pta.RayTracer(0, 0, device=0)
print("IR_DEBUG_DIRECT =", os.environ.get("IR_DEBUG_DIRECT", "(default: on)"))
for (w, h) in [(800, 450), (1920, 1080), (3840, 2160)]:
    scene = W.scene_s1()
    rt.set_active_camera(W.default_camera(w, h))
    rt.render_updated(scene, w, h)
    row = []
    for spp in [int(v) for v in os.environ.get("SPPS", "1,2,3,4,8,16").split(",")]:
        ts = []
        for _ in range(30):
            rt.render(scene, w, h, True, spp=spp)
            ts.append(rt.last_kernel_ms()[0])
        ts.sort()
        row.append(f"{spp} spp {ts[len(ts) // 2]:.3f}")
    print(f"{w}x{h} kernel ms (median of 30): " + ", ".join(row), flush=True)
rt.close()
