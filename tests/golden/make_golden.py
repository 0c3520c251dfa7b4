"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle.cpp).  The reference itself cannot run here (Rust,
no toolchain), so these vectors pin the ORACLE's behaviour (regression) and let the GPU tests check the HIP path
against committed data without the oracle.  The only reference-owned known answers are the two camera tests, which
are hard-coded in tests/test_oracle_camera.py from pinhole_camera.rs:138-239.

Run: python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from path_tracer_amd import workloads as W  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = W.RNG_SEED


def main():
    g = {}
    # RNG known answers: draw(seed, pixel, sample, k)
    keys = [(SEED, 0, 0, 0), (SEED, 0, 0, 1), (SEED, 1, 0, 0), (SEED, 12345, 678, 9), (0, 0, 0, 0),
            (0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 23), (SEED, 2073599, 1023, 23)]
    g["rng_keys"] = np.array(keys, dtype=np.uint64)
    g["rng_vals"] = np.array([O.rng_draw(*k) for k in keys], dtype=np.float32)
    g["mix32_in"] = np.array([0, 1, 0xDEADBEEF, 0xFFFFFFFF], dtype=np.uint32)
    g["mix32_out"] = np.array([O.lib.orc_rng_mix32(int(x)) for x in g["mix32_in"]], dtype=np.uint32)

    # S1 primary hits: full 800x450 CRC + a 160x90 map
    scene = W.scene_s1()
    cam = W.default_camera(800, 450)
    idx, t = O.primary_hits(O.camera_from(cam, 800, 450), scene)
    g["s1_hits_800x450_crc"] = np.array([zlib.crc32(idx.tobytes()), zlib.crc32(t.tobytes())], dtype=np.uint32)
    cam = W.default_camera(160, 90)
    idx, t = O.primary_hits(O.camera_from(cam, 160, 90), scene)
    g["s1_hits_160x90"] = idx.astype(np.int8)
    # S2 (10k spheres): CRC of 480x270 + 64x36 crop
    s2 = W.scene_s2()
    cam2 = W.camera_s2(480, 270)
    idx2, t2 = O.primary_hits(O.camera_from(cam2, 480, 270), s2)
    g["s2_hits_480x270_crc"] = np.array([zlib.crc32(idx2.tobytes()), zlib.crc32(t2.tobytes())], dtype=np.uint32)
    g["s2_hits_crop"] = idx2[100:136, 200:264].astype(np.int16)

    # S1 radiance sums, 64x36, 8 spp, bounces 8 / rr 2, both math modes
    cam = W.default_camera(64, 36)
    oc = O.camera_from(cam, 64, 36)
    for mode in (0, 1):
        s, c, _, st = O.render(oc, scene, 8, bounces=8, rr_start=2, seed=SEED, math_mode=mode, want_stats=True)
        g[f"s1_sum_64x36_8spp_math{mode}"] = s
        g[f"s1_stats_math{mode}"] = np.array([st[k] for k in ("samples", "traces", "tests", "disc_pos", "hits")], dtype=np.uint64)
    # constant-sky variant (scene.rs:57 fallback), reference default depth 5
    sc0 = W.scene_s1(skybox=None)
    s, c, _, _ = O.render(oc, sc0, 8, bounces=5, rr_start=2, seed=SEED, math_mode=1)
    g["s1_nosky_sum_64x36_8spp_b5_math1"] = s

    # resolve table
    sums = np.array([[0, 0, 0, 0], [1, 1, 1, 1], [4, 2, 1, 4], [0.5, 100.0, 1e-3, 2], [np.nan, -1.0, np.inf, 1],
                     [3e38, 3e38, 3e38, 7]], dtype=np.float32)
    counts = np.array([0, 1, 4, 2, 1, 7], dtype=np.uint32)
    g["resolve_sums"], g["resolve_counts"] = sums, counts
    g["resolve_argb"] = O.resolve_argb(sums, counts)
    np.savez_compressed(os.path.join(OUT, "golden_r1.npz"), **g)
    print("wrote", os.path.join(OUT, "golden_r1.npz"), {k: getattr(v, "shape", None) for k, v in g.items()})


if __name__ == "__main__":
    main()
