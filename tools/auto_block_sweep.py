"""Launch time against the sample-block size over frame sizes and sample counts (S1 scene, depth 8, accumulating): what
IR_SAMPLE_BLOCK_AUTO should pick.  usage (GPU box): python tools/auto_block_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

rt = pta.RayTracer(0, 0, device=0)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1()
blocks = (0, 4, 8, 16, 32, 64, 128)
print("kernel ms, median of 9; block 0 = auto (the size it picked in brackets)")
for (w, h) in [(400, 225), (800, 450), (1920, 1080), (3840, 2160)]:
    rt.set_active_camera(W.default_camera(w, h))
    rt.render_updated(scene, w, h)
    for spp in (4, 8, 16, 32, 64, 128, 256):
        row = []
        for b in blocks:
            if b > spp and b != 128:
                continue
            rt.set_sample_block(b)
            ts = []
            for _ in range(9):
                rt.render(scene, w, h, True, spp=spp)
                ts.append(rt.last_kernel_ms()[0])
            ts.sort()
            tag = f"auto[{rt.get_sample_block(w, h, spp)}]" if b == 0 else str(b)
            row.append((tag, ts[4]))
        best = min(t for _, t in row)
        print(f"{w}x{h} {spp:4d} spp: " + "  ".join(f"{k} {t:.3f}{'*' if t == best else ''}" for k, t in row), flush=True)
rt.set_sample_block(128)
rt.close()
