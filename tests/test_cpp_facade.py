"""Compiles and runs the C++ façade test (include/insploray.hpp over the C ABI)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "path_tracer_amd", "lib")
EXE = os.path.join(ROOT, "tests", "cpp", "facade_test")


def _build():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", EXE,
                           "-L", LIBDIR, "-linsploray_hip", f"-Wl,-rpath,{LIBDIR}"])


def test_cpp_facade_host_side(tmp_path):
    _build()
    out = subprocess.run([EXE], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "facade ok" in out.stdout


@pytest.mark.gpu
def test_cpp_facade_renders(tmp_path):
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE, "gpu"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "gpu render ok" in out.stdout
