"""ir_render_noalloc: the entry point that allocates nothing (block sums in a static device array, outputs in the caller's
buffers) against ir_render, bit for bit, on two calls in a row, the second with a smaller input.  Collected last (zz):
the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest
import torch  # noqa: F401  (at collection time: PyTorch-ROCm's bundled HIP runtime has to be the process's first, _capi.py)

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF


def _noalloc(rt, w, h, spp, acc, sums, counts):
    from path_tracer_amd import _capi
    _capi.check(rt._ctx, _capi.lib.ir_render_noalloc(rt._ctx, w, h, spp, 1 if acc else 0, sums.data_ptr(), counts.data_ptr()))
    _capi.check(rt._ctx, _capi.lib.ir_synchronize(rt._ctx))


def test_no_alloc_entry_point_matches_ir_render():
    import torch
    import path_tracer_amd as pta
    from path_tracer_amd import _capi, workloads as W
    rt = pta.RayTracer(0, 0, device=0)
    try:
        rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
        scene = W.scene_s1()
        # ---- call 1: 640x360, 512 spp = 4 sample blocks, so the block sums go through the static scratch
        w, h, spp = 640, 360, 512
        rt.set_active_camera(W.default_camera(w, h))
        rt.render(scene, w, h, False, spp=spp)  # also brings the scene's device tables up to date
        want_s, want_c = rt.get_radiance()
        want_s, want_c = want_s.copy(), want_c.copy()
        sums = torch.full((h, w, 4), 7.0, dtype=torch.float32, device="cuda")  # stale contents: accumulate == 0 must clear
        counts = torch.full((h, w), 9, dtype=torch.int32, device="cuda")
        _noalloc(rt, w, h, spp, False, sums, counts)
        assert sums.cpu().numpy().tobytes() == want_s.tobytes()
        assert counts.cpu().numpy().astype(np.uint32).tobytes() == want_c.tobytes()
        # the context's own film was not touched by the call
        s_own, c_own = rt.get_radiance()
        assert s_own.tobytes() == want_s.tobytes() and c_own.tobytes() == want_c.tobytes()
        # accumulate != 0 continues the sample sequence exactly as a second ir_render would
        rt.render(scene, w, h, False, spp=256)
        rt.render(scene, w, h, True, spp=256)
        two_s, _ = rt.get_radiance()
        two_s = two_s.copy()
        _noalloc(rt, w, h, 256, False, sums, counts)
        _noalloc(rt, w, h, 256, True, sums, counts)
        assert sums.cpu().numpy().tobytes() == two_s.tobytes()
        assert (counts.cpu().numpy() == 512).all()

        # ---- call 2, smaller input (fewer pixels, fewer samples, ragged last block): the scratch still holds call 1's sums
        w2, h2, spp2 = 200, 120, 300
        rt.set_active_camera(W.default_camera(w2, h2))
        rt.render(scene, w2, h2, False, spp=spp2)
        want2_s, want2_c = rt.get_radiance()
        want2_s, want2_c = want2_s.copy(), want2_c.copy()
        sums2 = torch.zeros((h2, w2, 4), dtype=torch.float32, device="cuda")
        counts2 = torch.zeros((h2, w2), dtype=torch.int32, device="cuda")
        _noalloc(rt, w2, h2, spp2, False, sums2, counts2)
        assert sums2.cpu().numpy().tobytes() == want2_s.tobytes()
        assert counts2.cpu().numpy().astype(np.uint32).tobytes() == want2_c.tobytes()

        # ---- limits and state errors are codes with messages, not faults
        bs = torch.zeros((2048, 4096, 4), dtype=torch.float32, device="cuda")
        bc = torch.zeros((2048, 4096), dtype=torch.int32, device="cuda")
        big = _capi.lib.ir_render_noalloc(rt._ctx, 4096, 2048, 256, 0, bs.data_ptr(), bc.data_ptr())
        assert big == _capi.IR_ELIMIT and b"static scratch" in _capi.lib.ir_last_error(rt._ctx)
        assert _capi.lib.ir_render_noalloc(rt._ctx, w, h, 1, 0, None, counts.data_ptr()) == _capi.IR_EINVAL
        scene.spheres[0].radius = 0.4           # moving a sphere re-uploads the geometry at once (linear scan: nothing else to do)
        scene.materials[0].albedo = [0.5, 0.6, 0.7]  # recolouring leaves the shading table stale until an ir_render
        rt._upload_scene(scene)
        assert _capi.lib.ir_render_noalloc(rt._ctx, w2, h2, 8, 0, sums2.data_ptr(), counts2.data_ptr()) == _capi.IR_ESTATE
        rt.render(scene, w2, h2, False, spp=0)
        rt.render(scene, w2, h2, False, spp=8)
        w3, _ = rt.get_radiance()
        _noalloc(rt, w2, h2, 8, False, sums2, counts2)
        assert sums2.cpu().numpy().tobytes() == w3.tobytes()
    finally:
        rt.close()


def test_no_alloc_with_the_lbvh():
    """The tree integrator through the same entry point (10 000 spheres, two blocks)."""
    import torch
    import path_tracer_amd as pta
    from path_tracer_amd import workloads as W
    rt = pta.RayTracer(0, 0, device=0)
    try:
        rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
        scene = W.scene_s2()
        w, h, spp = 320, 180, 192
        rt.set_active_camera(W.camera_s2(w, h))
        rt.render(scene, w, h, False, spp=spp)
        want_s, want_c = rt.get_radiance()
        want_s = want_s.copy()
        sums = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        counts = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        _noalloc(rt, w, h, spp, False, sums, counts)
        assert sums.cpu().numpy().tobytes() == want_s.tobytes()
        assert (counts.cpu().numpy() == spp).all()
    finally:
        rt.close()
