"""IR_SCHEDULE_WAVEFRONT (kernel per stage, HBM path queues, ballot/prefix compaction between bounces) must produce
the same accumulator, bit for bit, as IR_SCHEDULE_REGENERATION and as the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF


@pytest.fixture(scope="module")
def pta():
    import path_tracer_amd
    return path_tracer_amd


@pytest.fixture(scope="module")
def tracer(pta):
    rt = pta.RayTracer(0, 0, device=0)
    yield rt
    rt.close()


def _both(pta, tracer, scene, cam, w, h, spp, bounces, rr, max_paths=0, accumulate_more=0):
    from path_tracer_amd import _capi
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=bounces, max_compulsory_bounces=rr, seed=SEED)
    out = []
    try:
        for sched in (_capi.IR_SCHEDULE_REGENERATION, _capi.IR_SCHEDULE_WAVEFRONT):
            tracer.set_schedule(sched, max_paths)
            tracer.render(scene, w, h, False, spp=spp)
            if accumulate_more:
                tracer.render(scene, w, h, True, spp=accumulate_more)
            out.append(tracer.get_radiance())
    finally:
        tracer.set_schedule(_capi.IR_SCHEDULE_AUTO)
    (s0, c0), (s1, c1) = out
    assert np.array_equal(c0, c1)
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32)), f"max abs diff {np.abs(s0 - s1).max()}"
    return s1, c1


@pytest.mark.parametrize("skybox", ["synthetic", None])
@pytest.mark.parametrize("bounces,rr", [(8, 2), (5, 2), (1, 0), (0, 2), (3, 0)])
def test_wavefront_equals_regeneration_scan(pta, tracer, oracle, skybox, bounces, rr):
    from path_tracer_amd import workloads as W
    w, h, spp = 160, 90, 20
    scene = W.scene_s1(skybox=skybox)
    cam = W.default_camera(w, h)
    s, c = _both(pta, tracer, scene, cam, w, h, spp, bounces, rr)
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=bounces, rr_start=rr, seed=SEED, math_mode=1)
    assert np.array_equal(c, oc) and np.array_equal(s.view(np.uint32), os_.view(np.uint32))


def test_wavefront_batches_blocks_and_accumulation(pta, tracer):
    """Queues far smaller than pixels x samples: many batches per block, several blocks, then a second call on top."""
    from path_tracer_amd import workloads as W
    w, h = 200, 120                       # 4 x 2 tiles = 32768 queue slots per sample
    tracer.set_sample_block(16)
    try:
        _both(pta, tracer, W.scene_s1(), W.default_camera(w, h), w, h, 40, 8, 2, max_paths=3 * 32768, accumulate_more=7)
    finally:
        tracer.set_sample_block(128)


def test_wavefront_lbvh_scene(pta, tracer, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 128, 72, 6
    scene = W.scene_s2()
    cam = W.camera_s2(w, h)
    s, c = _both(pta, tracer, scene, cam, w, h, spp, 8, 2)
    assert tracer.get_accel_info()["accel"] == _capi.IR_ACCEL_LBVH
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32))


def test_wavefront_tile_partition(pta, tracer):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 200, 130, 6
    scene = W.scene_s1()
    tracer.set_active_camera(W.default_camera(w, h))
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    try:
        tracer.set_schedule(_capi.IR_SCHEDULE_WAVEFRONT)
        tracer.render(scene, w, h, False, spp=spp)
        whole, wc = tracer.get_radiance()
        acc = np.zeros_like(whole); cnt = np.zeros_like(wc)
        for r in range(3):
            tracer.set_partition(_capi.IR_PARTITION_TILES, r, 3)
            tracer.render(scene, w, h, False, spp=spp)
            s, c = tracer.get_radiance()
            assert ((c == 0) | (c == spp)).all()
            acc += s; cnt += c
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
        tracer.set_schedule(_capi.IR_SCHEDULE_AUTO)
    assert np.array_equal(acc.view(np.uint32), whole.view(np.uint32)) and np.array_equal(cnt, wc)


@pytest.mark.parametrize("scene_kind", ["s1", "s2"])
def test_wavefront_with_auto_sample_block(pta, tracer, oracle, scene_kind):
    """IR_SAMPLE_BLOCK_AUTO (0) must be resolved before the wavefront branch uses it (it once reached the queues raw: a
    zero-sample batch and IR_EDEVICE).  Same bits as regeneration and the oracle, with the block ir_get_sample_block names."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = (160, 90, 40) if scene_kind == "s1" else (128, 72, 12)
    scene = W.scene_s1() if scene_kind == "s1" else W.scene_s2(n=500)
    cam = W.default_camera(w, h) if scene_kind == "s1" else W.camera_s2(w, h)
    tracer.set_sample_block(_capi.IR_SAMPLE_BLOCK_AUTO)
    try:
        sb = tracer.get_sample_block(w, h, spp)
        assert 0 < sb < spp          # several blocks: the order of the block sums matters
        s, c = _both(pta, tracer, scene, cam, w, h, spp, 8, 2, accumulate_more=5)
        oracle.set_sample_block(sb)
        oc = oracle.camera_from(cam, w, h)
        os_, ocnt, _, _ = oracle.render(oc, scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
    finally:
        tracer.set_sample_block(128)
        oracle.set_sample_block(128)
    assert (c == spp + 5).all()
    # (the oracle comparison covers the first call: render it again alone)
    tracer.set_sample_block(_capi.IR_SAMPLE_BLOCK_AUTO)
    try:
        tracer.set_schedule(_capi.IR_SCHEDULE_WAVEFRONT)
        tracer.render(scene, w, h, False, spp=spp)
        s1, c1 = tracer.get_radiance()
    finally:
        tracer.set_schedule(_capi.IR_SCHEDULE_AUTO)
        tracer.set_sample_block(128)
    assert np.array_equal(c1, ocnt) and np.array_equal(s1.view(np.uint32), os_.view(np.uint32))
