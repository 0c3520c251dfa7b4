"""The kernels' short sqrt / reciprocal / divide sequences must return the bits of the IEEE f32 operations the
reference (and the oracle) evaluate: proven on the GPU under test through the C ABI (ir_selftest_exact_math),
exhaustively for one-operand forms and on 2^32 random cases for the others."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

NAMES = ["sqrt", "1/sqrt", "1/x", "x/pi", "x/2pi", "(x-1)/(x+1)", "a/b", "scatter weight", "roulette divide",
         "one-divide scan"]


def test_exact_math_selftest():
    import path_tracer_amd as pta
    from path_tracer_amd import _capi
    rt = pta.RayTracer(0, 0, device=0)
    try:
        out = (C.c_uint64 * 10)()
        _capi.check(rt._ctx, _capi.lib.ir_selftest_exact_math(rt._ctx, out, 10, 32))
        assert dict(zip(NAMES, [int(v) for v in out])) == dict.fromkeys(NAMES, 0)
        with pytest.raises(pta.InsplorayError):
            _capi.check(rt._ctx, _capi.lib.ir_selftest_exact_math(rt._ctx, out, 9, 32))
    finally:
        rt.close()
