"""C1 (800x450, 64 spp, depth 8) under sustained load: N renders enqueued back to back on the context's stream (no host
synchronisation in between, so the GPU neither idles nor drops its clock between the short launches), for the default
sample block (one block of 64 per pixel) and IR_SAMPLE_BLOCK_AUTO.  usage: python tools/c1_rate.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w, h, spp = 800, 450, 64
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.default_camera(w, h))
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
scene = W.scene_s1()
for label, block in (("default block (128)", 128), ("block 32", 32), ("block 16", 16), ("block 4", 4), ("auto block", _capi.IR_SAMPLE_BLOCK_AUTO)):
    rt.set_sample_block(block)
    rt.render(scene, w, h, False, spp=spp)
    one = []
    for _ in range(5):
        rt.render(scene, w, h, False, spp=spp)
        one.append(rt.last_kernel_ms()[0])
    t0 = time.perf_counter()
    for _ in range(reps):
        _capi.check(rt._ctx, _capi.lib.ir_render_async(rt._ctx, w, h, spp, 0))
    rt.synchronize()
    dt = time.perf_counter() - t0
    print(f"C1 {label}: samples per block {rt.get_sample_block(w, h, spp)}; one render at a time {min(one):.3f} ms "
          f"({w * h * spp / min(one) / 1e6:.1f} Gsamples/s); {reps} back to back {dt / reps * 1e3:.3f} ms each "
          f"({w * h * spp * reps / dt / 1e9:.1f} Gsamples/s)")
rt.close()
