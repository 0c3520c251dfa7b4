#!/usr/bin/env python3
"""LBVH split rule: Morton codes (IR_BUILD_HOST, the tree the device builder reproduces) against the binned surface-area
heuristic (IR_BUILD_HOST_SAH) on the uniformly scattered spheres of BASELINE config 3 and on clustered spheres.
Build time (host, upload included) and path-kernel rate at 64 spp, 1920x1080; radiance compared bit for bit.
usage: tools/sah_vs_morton.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402
from path_tracer_amd.scene import FlatScene  # noqa: E402


def clustered(n, clusters, spread, seed):
    rng = np.random.default_rng(seed)
    base = W.scene_s2(n=n)
    xyzr, mid = base.flat_spheres()
    xyzr = np.array(xyzr, dtype=np.float32)
    c = np.stack([rng.uniform(-40, 40, clusters), rng.uniform(0.5, 6.0, clusters), rng.uniform(-80, 0, clusters)], 1)
    k = rng.integers(0, clusters, n)
    xyzr[:, :3] = (c[k] + rng.normal(scale=spread, size=(n, 3))).astype(np.float32)
    xyzr[:, 3] = rng.uniform(0.03, 0.15, n).astype(np.float32)
    xyzr[0] = (0.0, -100.5, 0.0, 100.0)
    return FlatScene(xyzr, np.asarray(mid), base.flat_materials(), default_sky_color=(0.6, 0.7, 0.9))


def main():
    w, h, spp = 1920, 1080, 64
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_active_camera(W.camera_s2(w, h))
    rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
    rt.set_sample_block(0)
    rt.set_accel(_capi.IR_ACCEL_LBVH)
    scenes = [("S2: 10 000 uniformly scattered", W.scene_s2()), ("100 000 uniformly scattered", W.scene_s2(n=100000)),
              ("10 000 in 40 clusters (sigma 1.0)", clustered(10000, 40, 1.0, 1)),
              ("100 000 in 200 clusters (sigma 0.6)", clustered(100000, 200, 0.6, 2))]
    for name, scene in scenes:
        res = {}
        for label, build in (("morton", _capi.IR_BUILD_HOST), ("sah", _capi.IR_BUILD_HOST_SAH)):
            rt.set_accel_build(build)
            rt.render(scene, w, h, False, spp=1)  # builds
            b = rt.get_accel_build_info()
            info = rt.get_accel_info()
            best = 1e30
            for _ in range(3):
                rt.render(scene, w, h, False, spp=spp)
                best = min(best, rt.last_kernel_ms()[0])
            res[label] = (b["seconds"] * 1e3, info["nodes"], info["depth"], w * h * spp / best / 1e3, rt.get_radiance()[0].tobytes())
        m, s = res["morton"], res["sah"]
        print(f"{name:38s} morton: build {m[0]:7.2f} ms, {m[1]} nodes depth {m[2]}, {m[3]:8.0f} Msamples/s | "
              f"sah: build {s[0]:7.2f} ms, {s[1]} nodes depth {s[2]}, {s[3]:8.0f} Msamples/s ({s[3] / m[3]:.2f}x) | "
              f"radiance bit-identical: {m[4] == s[4]}", flush=True)
    rt.close()


if __name__ == "__main__":
    main()
