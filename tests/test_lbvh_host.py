"""LBVH == linear scan, checked on the CPU: tests/cpp/lbvh_check runs the product's own trace_lbvh / trace_linear
(pt_device.h compiled for the host, same -ffp-contract=off arithmetic as on the device) over primary, bounce-like,
axis-parallel and inside-sphere rays and compares (t, index) bitwise.  The scan is the restatement of
integrator.rs:106-127; the GPU tests repeat the comparison on the device against the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "lbvh_check")
SRC = [os.path.join(ROOT, "tests", "cpp", "lbvh_check.hip"), os.path.join(ROOT, "path_tracer_amd", "csrc", "lbvh.cpp")]
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def checker():
    deps = SRC + [os.path.join(ROOT, "path_tracer_amd", "csrc", f) for f in ("pt_device.h", "lbvh.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available to build the host checker")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "-o", EXE] + SRC)
    return EXE


def _rays(xyzr, n_each, rng, cam_rays):
    out = [cam_rays.reshape(-1, 6)]
    n = xyzr.shape[0]
    # bounce-like: origin just off a sphere surface, non-unit direction in the upper hemisphere (sampler.rs:43-53)
    i = rng.integers(0, n, n_each)
    nr = rng.normal(size=(n_each, 3)).astype(np.float32)
    nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    pos = xyzr[i, :3] + nr * xyzr[i, 3:4]
    o = (pos + nr * np.float32(1.1920929e-07)).astype(np.float32)
    d = rng.normal(size=(n_each, 3)).astype(np.float32)
    d *= np.sign((d * nr).sum(1, keepdims=True))
    d *= rng.uniform(0.05, 1.0, (n_each, 1)).astype(np.float32)
    out.append(np.concatenate([o, d], 1))
    # axis-parallel and grazing rays along rows of spheres; rays starting inside spheres; far origins
    o2 = xyzr[rng.integers(0, n, n_each), :3] + rng.normal(scale=0.3, size=(n_each, 3)).astype(np.float32)
    d2 = np.zeros((n_each, 3), dtype=np.float32)
    d2[np.arange(n_each), rng.integers(0, 3, n_each)] = rng.choice([-1.0, 1.0, 0.25, -3.0], n_each)
    out.append(np.concatenate([o2.astype(np.float32), d2], 1))
    o3 = rng.uniform(-60, 60, (n_each, 3)).astype(np.float32)
    tgt = xyzr[rng.integers(0, n, n_each)]
    d3 = (tgt[:, :3] + rng.normal(size=(n_each, 3)) * tgt[:, 3:4] * 1.02 - o3).astype(np.float32)  # near-tangent
    out.append(np.concatenate([o3, d3], 1))
    return np.ascontiguousarray(np.concatenate(out, 0), dtype=np.float32)


@pytest.mark.parametrize("split", ["morton", "sah"])
@pytest.mark.parametrize("n", [1, 3, 5, 64, 1000, 10000])
def test_lbvh_matches_scan(checker, oracle, tmp_path, n, split):
    from path_tracer_amd import workloads as W
    scene = W.scene_s2(n=n)
    xyzr, _ = scene.flat_spheres()
    xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
    cam = W.camera_s2(160, 90)
    rays = oracle.camera_from(cam, 160, 90).primary_rays()
    rng = np.random.default_rng(n)
    r = _rays(xyzr, 20000 if n >= 1000 else 4000, rng, rays)
    sp, rp = tmp_path / "s.bin", tmp_path / "r.bin"
    xyzr.tofile(sp)
    r.tofile(rp)
    res = subprocess.run([checker, str(sp), str(rp), split], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok")


@pytest.mark.parametrize("split", ["morton", "sah"])
def test_lbvh_duplicates_and_ties(checker, tmp_path, split):
    """Coincident spheres: the scan keeps the lowest index (strict '<', integrator.rs:121); so must the LBVH."""
    rng = np.random.default_rng(7)
    base = rng.uniform(-5, 5, (200, 3)).astype(np.float32)
    xyzr = np.concatenate([np.repeat(base, 5, 0), np.full((1000, 1), 0.4, np.float32)], 1)
    xyzr = xyzr[rng.permutation(1000)]
    o = rng.uniform(-8, 8, (30000, 3)).astype(np.float32)
    d = (xyzr[rng.integers(0, 1000, 30000), :3] - o + rng.normal(scale=0.2, size=(30000, 3))).astype(np.float32)
    sp, rp = tmp_path / "s.bin", tmp_path / "r.bin"
    np.ascontiguousarray(xyzr, dtype=np.float32).tofile(sp)
    np.ascontiguousarray(np.concatenate([o, d], 1), dtype=np.float32).tofile(rp)
    res = subprocess.run([checker, str(sp), str(rp), split], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr


def test_sah_tree_on_clustered_and_degenerate_scenes(checker, tmp_path):
    """The surface-area build (IR_BUILD_HOST_SAH) on what its binning can trip over: tight clusters far apart, all centres
    on one line, all centres equal, huge and tiny radii side by side: same (t, index) as the scan."""
    rng = np.random.default_rng(11)
    scenes = []
    centres = rng.uniform(-200, 200, (12, 3))
    scenes.append(np.concatenate([(centres[rng.integers(0, 12, 3000)] + rng.normal(scale=0.5, size=(3000, 3))),
                                  rng.uniform(0.05, 0.3, (3000, 1))], 1))
    line = np.zeros((500, 4)); line[:, 0] = np.linspace(-50, 50, 500); line[:, 3] = 0.3
    scenes.append(line)
    same = np.zeros((300, 4)); same[:, 2] = -5.0; same[:, 3] = rng.uniform(0.1, 2.0, 300)
    scenes.append(same)
    mixed = np.concatenate([rng.uniform(-10, 10, (800, 3)), rng.choice([1e-3, 0.2, 30.0], (800, 1))], 1)
    scenes.append(mixed)
    for k, xyzr in enumerate(scenes):
        xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
        n = xyzr.shape[0]
        o = (xyzr[rng.integers(0, n, 8000), :3] + rng.normal(scale=3.0, size=(8000, 3))).astype(np.float32)
        d = (xyzr[rng.integers(0, n, 8000), :3] - o + rng.normal(scale=0.1, size=(8000, 3))).astype(np.float32)
        sp, rp = tmp_path / f"s{k}.bin", tmp_path / f"r{k}.bin"
        xyzr.tofile(sp)
        np.ascontiguousarray(np.concatenate([o, d], 1), dtype=np.float32).tofile(rp)
        res = subprocess.run([checker, str(sp), str(rp), "sah"], capture_output=True, text=True)
        assert res.returncode == 0, f"scene {k}: " + res.stdout + res.stderr
