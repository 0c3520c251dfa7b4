"""The C-ABI library loads on a box without a GPU and exports every symbol include/insploray_hip.h declares (and the
ctypes prototype table covers exactly that set).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "insploray_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ir_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("ir_create", "ir_destroy", "ir_set_spheres", "ir_set_materials", "ir_set_sky_color", "ir_set_skybox",
                 "ir_set_pinhole", "ir_set_primary_rays", "ir_set_integrator", "ir_render", "ir_get_output_argb",
                 "ir_get_radiance", "ir_get_primary_hits", "ir_last_render_seconds", "ir_last_error", "ir_exr_load"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from path_tracer_amd import _capi
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in insploray_hip.h but not exported"
    assert sorted(_capi.PROTOTYPES) == declared_functions()
    assert _capi.lib.ir_abi_version() == 3


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import path_tracer_amd as pta
    with pytest.raises(pta.InsplorayError) as e:
        pta.RayTracer(8, 8)
    assert "no HIP device" in str(e.value)


def test_null_context_is_an_error_not_a_crash():
    from path_tracer_amd import _capi
    assert _capi.lib.ir_render(None, 4, 4, 1, 0) == _capi.IR_EINVAL
    assert _capi.lib.ir_set_spheres(None, None, None, 0) == _capi.IR_EINVAL
    assert _capi.lib.ir_get_output_argb(None, None, 0) == _capi.IR_EINVAL
    _capi.lib.ir_destroy(None)
    assert _capi.lib.ir_last_error(None) == b"null context"


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under path_tracer_amd/ or include/ may name it."""
    bad = []
    for base in ("path_tracer_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(d, f), errors="replace").read()
                    if re.search(r"oracle_lib|liboracle|oracle/|import oracle|orc_", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """PyTorch-ROCm bundles its own HIP runtime; two copies in one process leave the second without a GPU.  _capi.py loads
    torch's copy first when torch is installed, so the library and a later `import torch` share one (no GPU needed to see it)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import path_tracer_amd._capi as c; import torch\n"
            "paths = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(paths), paths)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "IR_SYSTEM_HIP_RUNTIME"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout
