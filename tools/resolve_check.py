"""resolve_kernel against the oracle's resolve (libm powf) over a wide range of accumulator values: histogram of the
per-channel byte differences.  usage (GPU box): python tools/resolve_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta
from path_tracer_amd import workloads as W
from tests import oracle_lib as oracle
w, h = 1024, 512
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.default_camera(w, h))
rt.render(W.scene_s1(), w, h, False, spp=1)
rng = np.random.default_rng(5)
n = w * h
counts = rng.integers(1, 5000, n).astype(np.uint32).reshape(h, w)
mean = np.exp(rng.uniform(np.log(1e-12), np.log(1e6), (n, 3))).astype(np.float32)
mean[:1000] = 0.0
mean[1000:2000] = rng.uniform(0, 1, (1000, 3))
sums = np.zeros((h, w, 4), dtype=np.float32)
sums[..., :3] = mean.reshape(h, w, 3) * counts[..., None]
sums[..., 3] = counts
sums[0, 0] = (np.inf, np.nan, -1.0, 7.0)
# the corners of powf: a mean of exactly -1 (Reinhard -> -inf -> +inf), other negatives (NaN -> 0), -0, denormals, huge
counts[0, 1:8] = 1
sums[0, 1] = (-1.0, -0.5, -2.0, 1.0)
sums[0, 2] = (-0.0, 1e-40, 3e38, 1.0)
sums[0, 3] = (-np.inf, 1e-45, 1.1754944e-38, 1.0)
sums[0, 4] = (-1.0000001, -0.99999994, 0.0, 1.0)
rt.set_radiance(sums, counts)
out = rt.get_output().copy().reshape(h, w)
ref = oracle.resolve_argb(sums, counts).reshape(h, w)
for shift, name in ((16, "r"), (8, "g"), (0, "b"), (24, "a")):
    d = ((out >> shift) & 0xFF).astype(np.int32) - ((ref >> shift) & 0xFF).astype(np.int32)
    print(name, {int(k): int(v) for k, v in zip(*np.unique(d, return_counts=True))})
rt.close()
