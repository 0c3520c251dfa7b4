#!/usr/bin/env python3
"""Where in the frame are the expensive pixels?  Renders every 64x64 tile on its own (tile partition with one rank per tile)
with the statistics build and prints, per tile row, traces per sample (and LBVH box tests per sample): the per-row cost
profile behind the order in which the path kernel deals out the list (pt_kernels.hip decode_task).
usage: tools/cost_map.py [c2|c3] [spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w, h = 1920, 1080
scene, cam = (W.scene_s2(), W.camera_s2(w, h)) if cfg == "c3" else (W.scene_s1(), W.default_camera(w, h))
tx, ty = (w + 63) // 64, (h + 63) // 64
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(cam)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
rows = np.zeros((ty, 4))
for t in range(tx * ty):
    rt.set_partition(_capi.IR_PARTITION_TILES, t, tx * ty)
    rt.enable_stats()
    rt.render(scene, w, h, False, spp=spp)
    st = rt.get_stats()
    pst = rt.get_primary_stats()
    ast = rt.get_accel_stats() if cfg == "c3" else {"bvh_node_visits": 0}
    rows[t // tx] += (st["samples"], st["traces"], pst["path_pixels"], ast["bvh_node_visits"])
rt.close()
print(f"{cfg} {w}x{h}, {spp} spp per tile render; tile rows from the top of the frame (the list of path pixels runs the same way)")
for r in range(ty):
    s, tr, pp, nd = rows[r]
    path_samples = pp * spp
    print(f"tile row {r:2d}: path pixels {int(pp):7d}  bounce rays per path sample {(tr - s) / max(path_samples, 1):5.2f}"
          + (f"  box tests per path sample {nd / max(path_samples, 1):7.1f}" if cfg == "c3" else ""))
