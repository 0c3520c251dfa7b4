"""AddressSanitizer + UBSan over the host code a user's scene reaches besides the EXR reader (tests/test_exr.py has that
one): the LBVH builder (csrc/lbvh.cpp), the sky importance-sampling tables (csrc/sky_distribution.cpp), the camera and
sky-lookup helpers (csrc/host_math.cpp), and the traversal code of pt_device.h compiled for the host (the stack, the
climb records, the 48-byte record decode).  Host code only: GPU sanitizers are not available on the pool.  The reference
has no counterpart (Rust's bounds checks play this part there)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "path_tracer_amd", "csrc")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _clean(r):
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_builders_and_host_math_under_sanitizers(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = tmp_path / "host_asan"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-static-libasan", "-static-libubsan", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_asan_main.cpp"), os.path.join(CSRC, "lbvh.cpp"),
                           os.path.join(CSRC, "sky_distribution.cpp"), os.path.join(CSRC, "host_math.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV, timeout=300)
    _clean(r)
    assert r.stdout.strip().endswith("ALL OK"), r.stdout[-2000:]


def test_traversal_on_the_host_under_sanitizers(tmp_path):
    """tests/cpp/lbvh_check (scan == LBVH == climb, bit for bit) with the host half built -fsanitize=address,undefined."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from path_tracer_amd import workloads as W
    exe = tmp_path / "lbvh_check_asan"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-o", str(exe),
                           os.path.join(ROOT, "tests", "cpp", "lbvh_check.hip"), os.path.join(CSRC, "lbvh.cpp")])
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 1000):
        xyzr, _ = W.scene_s2(n=n).flat_spheres()
        xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
        m = 1500
        i = rng.integers(0, n, m)
        nr = rng.normal(size=(m, 3)).astype(np.float32)
        nr /= np.linalg.norm(nr, axis=1, keepdims=True)
        o = xyzr[i, :3] + nr * xyzr[i, 3:4]
        d = rng.normal(size=(m, 3)).astype(np.float32)
        d[::50] = 0.0
        d[::50, 0] = 1.0                       # axis-parallel rays: 0 * inf in the slab test
        (tmp_path / "s.bin").write_bytes(xyzr.tobytes())
        (tmp_path / "r.bin").write_bytes(np.concatenate([o, d], 1).astype(np.float32).tobytes())
        r = subprocess.run([str(exe), str(tmp_path / "s.bin"), str(tmp_path / "r.bin")], capture_output=True, text=True, env=ENV, timeout=300)
        _clean(r)
        assert r.stdout.startswith("ok "), r.stdout
