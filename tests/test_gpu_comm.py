"""The exchange step inside the C ABI (ir_comm_*, ir_combine*, ir_pack_owned_tiles / ir_unpack_tiles) on the one GPU a test
box has.  RCCL refuses two ranks on one device, so the transport runs with a world of ONE rank (every RCCL call the
N-rank path makes except send/recv to a peer), and the N-rank logic around it — which tiles a rank owns, the payload layout, the
root's unpack — is exercised by letting one GPU play every rank in turn through the transport-independent halves.
N > 1 over xGMI itself: unmeasured until a SCALE record exists."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF


@pytest.fixture()
def tracer():
    import path_tracer_amd as pta
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    yield rt
    rt.close()


def _unique_id():
    from path_tracer_amd import _capi
    buf = C.create_string_buffer(_capi.IR_COMM_ID_BYTES)
    _capi.check(None, _capi.lib.ir_comm_unique_id(buf))
    return buf


@pytest.mark.parametrize("mode", ["samples", "tiles", "none"])
def test_combine_with_a_world_of_one(tracer, mode):
    from path_tracer_amd import _capi, workloads as W
    lib, ctx = _capi.lib, tracer._ctx
    w, h = 500, 300  # ragged edge tiles
    scene = W.scene_s1()
    tracer.set_active_camera(W.default_camera(w, h))
    _capi.check(ctx, lib.ir_comm_init(ctx, _unique_id(), 0, 1))
    assert lib.ir_comm_init(ctx, _unique_id(), 0, 1) == _capi.IR_ESTATE  # already has one
    if mode != "none":
        tracer.set_partition(_capi.IR_PARTITION_TILES if mode == "tiles" else _capi.IR_PARTITION_SAMPLES, 0, 1)
    tracer.render(scene, w, h, False, spp=64)
    own_s, own_c = tracer.get_radiance()
    own_s, own_c = own_s.copy(), own_c.copy()
    _capi.check(ctx, lib.ir_combine(ctx, 0))
    assert lib.ir_last_combine_ms(ctx) > 0.0
    comb_s, comb_c = tracer.get_radiance()  # the root's readbacks show the combined film
    assert comb_s.tobytes() == own_s.tobytes() and comb_c.tobytes() == own_c.tobytes()
    argb = tracer.get_output().copy()
    # the rank's own film is intact: rendering on accumulates onto it, and readbacks return to it
    tracer.render(scene, w, h, True, spp=64)
    s2, c2 = tracer.get_radiance()
    assert (c2 == 128).all() and not np.array_equal(s2, own_s)
    # overlap: snapshot, exchange on its own stream, next render right behind it; the combined film is the snapshot's
    _capi.check(ctx, lib.ir_combine_async(ctx, 0))
    tracer.render(scene, w, h, False, spp=16)
    _capi.check(ctx, lib.ir_combine_wait(ctx))
    s3, c3 = tracer.get_radiance()
    assert (c3 == 16).all()  # a render after the combine: readbacks show the context's film again
    assert lib.ir_combine(ctx, 5) == _capi.IR_EINVAL
    _capi.check(ctx, lib.ir_comm_destroy(ctx))
    assert lib.ir_combine(ctx, 0) == _capi.IR_ESTATE
    assert ((argb >> 24) == 0xBA).all()


def test_partition_and_communicator_must_agree(tracer):
    from path_tracer_amd import _capi, workloads as W
    lib, ctx = _capi.lib, tracer._ctx
    tracer.set_active_camera(W.default_camera(64, 64))
    tracer.render(W.scene_s1(), 64, 64, False, spp=1)
    _capi.check(ctx, lib.ir_comm_init(ctx, _unique_id(), 0, 1))
    tracer.set_partition(_capi.IR_PARTITION_SAMPLES, 1, 2)
    assert lib.ir_combine(ctx, 0) == _capi.IR_ESTATE
    assert b"partition is rank 1 of 2" in lib.ir_last_error(ctx)


@pytest.mark.parametrize("world", [2, 8, 5])
def test_tile_gather_halves_with_one_gpu_playing_every_rank(tracer, world):
    """render rank r's tiles -> ir_pack_owned_tiles -> (the bytes RCCL would carry) -> ir_unpack_tiles on the root: the
    root's film equals the one-GPU render bit for bit."""
    import torch
    import path_tracer_amd as pta
    from path_tracer_amd import _capi, workloads as W
    lib = _capi.lib
    w, h, spp = 1000, 520, 32  # 16 x 9 tiles, ragged right and top
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    tracer.render(scene, w, h, False, spp=spp)
    whole_s, whole_c = tracer.get_radiance()
    whole_s, whole_c = whole_s.copy(), whole_c.copy()
    root = pta.RayTracer(0, 0, device=0)
    try:
        root.set_active_camera(cam)
        root.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
        root.render(scene, w, h, False, spp=0)  # sized, zeroed
        total = 0
        for rank in range(world):
            tracer.set_partition(_capi.IR_PARTITION_TILES, rank, world)
            tracer.render(scene, w, h, False, spp=spp)
            nbytes = lib.ir_tile_payload_bytes(w, h, rank, world)
            assert nbytes == len(range(rank, 16 * 9, world)) * 4096 * 20
            total += nbytes
            payload = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            assert lib.ir_pack_owned_tiles(tracer._ctx, payload.data_ptr(), nbytes - 20) == _capi.IR_EINVAL
            _capi.check(tracer._ctx, lib.ir_pack_owned_tiles(tracer._ctx, payload.data_ptr(), nbytes))
            wire = payload.cpu().clone().cuda()  # what a transport would deliver
            _capi.check(root._ctx, lib.ir_unpack_tiles(root._ctx, wire.data_ptr(), nbytes, rank, world))
        assert total == 16 * 9 * 4096 * 20
        got_s, got_c = root.get_radiance()
        assert got_s.tobytes() == whole_s.tobytes() and got_c.tobytes() == whole_c.tobytes()
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
        root.close()


def test_auto_sample_block_and_memory_bounded_blocks(tracer, oracle):
    """IR_SAMPLE_BLOCK_AUTO picks the block from the launch grid (pure function; the oracle mirrors it), and a block size of 1
    (the reference's strictly sequential sum order) is cut into launches whose block sums fit a fixed budget
    (this used to ask for 136 GB at 1080p x 4096 spp)."""
    from path_tracer_amd import _capi, workloads as W
    lib, ctx = _capi.lib, tracer._ctx
    scene = W.scene_s1()
    # C1's shape: 800x450 = 13x8 tiles = 1664 workgroups-worth of pixels (a small frame: blocks down to 4); 64 spp -> blocks of 4
    # (16 x 1664 >= 26000); 1080p = 8160: 64 spp -> 16; from 32-sample blocks on, the size within an eighth whose last block
    # comes closest to a quarter block (128 spp: 32 -> 4 x 30 + 8; 256: 64 -> 4 x 60 + 16; 1024: 128 -> 8 x 124 + 32); a launch
    # of few samples ends in one block per pixel
    w, h, spp = 800, 450, 64
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    tracer.set_sample_block(_capi.IR_SAMPLE_BLOCK_AUTO)
    sb = C.c_uint32()
    _capi.check(ctx, lib.ir_get_sample_block(ctx, w, h, spp, C.byref(sb)))
    assert sb.value == 4
    for (w_, h_, spp_, want_sb) in ((1920, 1080, 1024, 124), (1920, 1080, 256, 60), (1920, 1080, 128, 30), (1920, 1080, 64, 16), (1920, 1080, 1000, 138), (1920, 1080, 65536, 131), (1920, 1080, 200, 62),
                                    (1920, 1080, 8, 8), (3840, 2160, 64, 128), (800, 450, 256, 16), (800, 450, 128, 8), (64, 64, 1, 4)):
        _capi.check(ctx, lib.ir_get_sample_block(ctx, w_, h_, spp_, C.byref(sb)))
        assert sb.value == want_sb, (w_, h_, spp_, sb.value)
    try:
        tracer.render(scene, w, h, False, spp=spp)
        s, c = tracer.get_radiance()
        ocam = oracle.camera_from(cam, w, h)
        oracle.set_sample_block(4)
        want, wcnt, _, _ = oracle.render(ocam, scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1, threads=16)
        assert np.array_equal(s.view(np.uint32), want.view(np.uint32)) and np.array_equal(c, wcnt)
        # block of 1 at 1080p, 4096 spp
        tracer.set_sample_block(1)
        W_, H_ = 1920, 1080
        cam2 = W.default_camera(W_, H_)
        tracer.set_active_camera(cam2)
        tracer.render(scene, W_, H_, False, spp=4096)
        assert tracer.last_kernel_ms()[1] > 1  # several launches
        s, c = tracer.get_radiance()
        assert (c == 4096).all()
        ocam2 = oracle.camera_from(cam2, W_, H_)
        for (x, y) in [(960, 540), (1100, 560), (5, 5)]:
            ex = oracle.pixel_samples(ocam2, scene, x, y, 4096, bounces=8, rr_start=2, seed=SEED, math_mode=1)
            rows = np.zeros((4097, 4), dtype=np.float32)
            rows[1:, :3] = ex
            rows[1:, 3] = 1.0
            assert s[y, x].tobytes() == np.add.accumulate(rows, axis=0, dtype=np.float32)[-1].tobytes()
    finally:
        oracle.set_sample_block(128)
        tracer.set_sample_block(128)
