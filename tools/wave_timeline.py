#!/usr/bin/env python3
"""When do the path kernel's waves start and end?  Every wave stamps the 100 MHz realtime counter at its first and last
instruction (IR_DEBUG_WAVE_TIMES, pt_kernels.hip stamp_wave); this prints, for a launch, how long the kernel ran and when
which share of the waves had finished: the length of the ramp and of the drain of a persistent launch, measured.
usage: tools/wave_timeline.py [c1|c2|c3] [spp] [sample block, 0 = auto]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = os.path.join(tempfile.gettempdir(), "ir_wave_times_%d.txt" % os.getpid())
os.environ["IR_DEBUG_WAVE_TIMES"] = path
import numpy as np  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import workloads as W  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
if cfg == "c1":
    w, h, spp, scene, cam = 800, 450, 64, W.scene_s1(), W.default_camera(800, 450)
elif cfg == "c3":
    w, h, spp, scene, cam = 1920, 1080, 256, W.scene_s2(), W.camera_s2(1920, 1080)
else:
    w, h, spp, scene, cam = 1920, 1080, 1024, W.scene_s1(), W.default_camera(1920, 1080)
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(cam)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
rt.set_sample_block(blk)
for rep in range(3):
    rt.render(scene, w, h, False, spp=spp)
t = np.loadtxt(path, dtype=np.uint64).reshape(-1, 2).astype(np.int64)
os.remove(path)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # us
total = end.max()
ms, n = rt.last_kernel_ms()
print(f"{cfg} {w}x{h} {spp} spp, sample block {rt.get_sample_block(w, h, spp)}: {len(t)} waves, first start to last end {total:.1f} us "
      f"(HIP events around the kernel: {ms * 1e3:.1f} us)")
print("waves started: 50 %% by %.1f us, 90 %% by %.1f us, all by %.1f us" % tuple(np.percentile(start, [50, 90, 100])))
q = [1, 10, 25, 50, 75, 90, 99, 100]
print("waves finished: " + ", ".join("%d %% by %.1f us" % (p, v) for p, v in zip(q, np.percentile(end, q))))
busy = (end - start).sum()
print(f"wave-time used / (waves x kernel time) = {busy / (len(t) * total):.3f}; the last wave ends {total - np.percentile(end, 50):.1f} us after the median wave")
rt.close()
