"""Parity at BASELINE.json's full sizes.  The oracle cannot render 2.1 G samples in a test, so the full-size frames are
checked (a) against the oracle on a seeded subset of pixels, whose samples the oracle evaluates one by one and which
are then summed here in the accumulation contract's order (blocks of 128 from zero, merged in order), and (b) through
properties that do not depend on the size: partitions add up to the whole, a render split into two calls equals one
call, both closest-hit structures and both schedules give the same bits, every pixel has exactly spp samples."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF
W_, H_ = 1920, 1080
BLOCK = 128


@pytest.fixture(scope="module")
def pta():
    import path_tracer_amd
    return path_tracer_amd


@pytest.fixture(scope="module")
def tracer(pta):
    rt = pta.RayTracer(0, 0, device=0)
    yield rt
    rt.close()


@pytest.fixture(scope="module")
def c2(pta, tracer):
    """The C2 frame (S1 + emitter + synthetic sky, 1920x1080, 1024 spp, depth 8), rendered once."""
    from path_tracer_amd import workloads as W
    scene = W.scene_s1()
    cam = W.default_camera(W_, H_)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    tracer.render(scene, W_, H_, False, spp=1024)
    s, c = tracer.get_radiance()
    return scene, cam, s.copy(), c.copy()


def _contract_sum(samples, block=None):
    """(spp, 3) per-sample radiance -> float32 (r, g, b, weight) summed as ir_render defines it."""
    block = block or BLOCK
    total = np.zeros(4, dtype=np.float32)
    for s0 in range(0, samples.shape[0], block):
        blk = np.zeros(4, dtype=np.float32)
        for v in samples[s0:s0 + block]:
            blk[0] += v[0]; blk[1] += v[1]; blk[2] += v[2]; blk[3] += np.float32(1.0)
        total += blk
    return total


def test_c2_every_pixel_has_its_samples(c2):
    _, _, s, c = c2
    assert (c == 1024).all()
    assert (s[..., 3] == np.float32(1024.0)).all()
    # brdf*cos/pdf is 0/0 when the sampled cosine rounds to zero (about once per 4e9 bounces): the reference, the oracle
    # and the kernels then carry a NaN in that pixel; at 2.1 G samples a pixel or two may
    bad = ~np.isfinite(s).all(-1)
    assert bad.sum() <= 8
    assert (s[~bad][:, :3] >= 0).all()
    # the frame is an image, not a constant: sky, ground, two spheres
    assert len(np.unique(np.round(s[::40, ::40, 0] / 1024, 2))) > 50


def test_c2_sampled_pixels_equal_the_oracle(c2, oracle):
    scene, cam, s, _ = c2
    ocam = oracle.camera_from(cam, W_, H_)
    rng = np.random.default_rng(2024)
    pixels = [(int(rng.integers(0, W_)), int(rng.integers(0, H_))) for _ in range(24)]
    pixels += [(0, 0), (W_ - 1, H_ - 1), (W_ // 2, H_ // 2), (W_ // 2 + 130, H_ // 2 - 40), (960, 300), (1100, 560)]
    worst = 0.0
    for (x, y) in pixels:
        exact = oracle.pixel_samples(ocam, scene, x, y, 1024, bounces=8, rr_start=2, seed=SEED, math_mode=1)
        want = _contract_sum(exact)
        assert s[y, x].tobytes() == want.tobytes(), f"pixel ({x},{y}): {s[y, x]} vs {want}"
        libm = _contract_sum(oracle.pixel_samples(ocam, scene, x, y, 1024, bounces=8, rr_start=2, seed=SEED, math_mode=0))
        worst = max(worst, float(np.abs(s[y, x, :3] - libm[:3]).max() / max(float(libm[:3].max()), 1e-6)))
    assert worst < 1e-3, worst  # against libm transcendentals, per pixel at converged spp (north_star's bar)


def test_c2_tile_partition_adds_up_bit_for_bit(c2, tracer):
    from path_tracer_amd import _capi
    scene, cam, s, c = c2
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    total = np.zeros_like(s)
    counts = np.zeros_like(c)
    try:
        for rank in range(4):
            tracer.set_partition(_capi.IR_PARTITION_TILES, rank, 4)
            tracer.render(scene, W_, H_, False, spp=1024)
            ps, pc = tracer.get_radiance()
            assert not (total[..., 3] * ps[..., 3]).any()  # disjoint supports
            total += ps
            counts += pc
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    assert total.tobytes() == s.tobytes() and counts.tobytes() == c.tobytes()


def test_c2_sample_partition_and_split_calls(c2, tracer):
    from path_tracer_amd import _capi
    scene, cam, s, c = c2
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    # two calls of whole sample blocks == one call, bit for bit (same blocks, same merge order)
    tracer.render(scene, W_, H_, False, spp=640)
    tracer.render(scene, W_, H_, True, spp=384)
    s2, c2_ = tracer.get_radiance()
    assert s2.tobytes() == s.tobytes() and c2_.tobytes() == c.tobytes()
    # sample-range partition over 2 ranks: rank r renders samples [512 r, 512 r + 512): the same 8 blocks, merged into two
    # partial frames instead of one, so the sum of the partial frames differs from the whole only by f32 summation order
    total = np.zeros(s.shape, dtype=np.float64)
    try:
        for rank in range(2):
            tracer.set_partition(_capi.IR_PARTITION_SAMPLES, rank, 2)
            tracer.render(scene, W_, H_, False, spp=512)
            ps, pc = tracer.get_radiance()
            assert (pc == 512).all()
            total += ps
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    assert np.allclose(total[..., :3], s[..., :3], rtol=2e-6, atol=1e-4, equal_nan=True)
    assert (total[..., 3] == 1024).all()


def test_c2_structures_and_schedules_agree(c2, tracer):
    from path_tracer_amd import _capi
    scene, cam, s, c = c2
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    tracer.set_accel(_capi.IR_ACCEL_LBVH)
    try:
        tracer.render(scene, W_, H_, False, spp=1024)
        sl, cl = tracer.get_radiance()
    finally:
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    assert sl.tobytes() == s.tobytes() and cl.tobytes() == c.tobytes()
    tracer.set_schedule(_capi.IR_SCHEDULE_WAVEFRONT)
    try:
        tracer.render(scene, W_, H_, False, spp=1024)
        sw, cw = tracer.get_radiance()
    finally:
        tracer.set_schedule(_capi.IR_SCHEDULE_AUTO)
    assert sw.tobytes() == s.tobytes() and cw.tobytes() == c.tobytes()


def test_c3_full_size_hits_and_sampled_pixels(pta, tracer, oracle):
    """C3: 10 000 spheres, 1920x1080, 256 spp through the LBVH.  Primary hits of a pixel subset against the oracle's scan
    of the same rays; radiance sums of a few pixels against the oracle's per-sample values."""
    from path_tracer_amd import _capi, workloads as W
    scene = W.scene_s2()
    cam = W.camera_s2(W_, H_)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    idx, t = tracer.get_primary_hits(scene, W_, H_, with_t=True)
    assert tracer.get_accel_info()["accel"] == _capi.IR_ACCEL_LBVH
    ocam = oracle.camera_from(cam, W_, H_)
    rays = ocam.primary_rays()
    xyzr, _ = scene.flat_spheres()
    rng = np.random.default_rng(77)
    for _ in range(400):
        x, y = int(rng.integers(0, W_)), int(rng.integers(0, H_))
        oi, ot, _, _ = oracle.trace(xyzr, rays[y, x, :3], rays[y, x, 3:])
        assert idx[y, x] == oi
        if oi >= 0:
            assert np.float32(t[y, x]).tobytes() == np.float32(ot).tobytes()
    assert len(np.unique(idx)) > 1000
    tracer.render(scene, W_, H_, False, spp=256)
    s, c = tracer.get_radiance()
    assert (c == 256).all() and (s[..., 3] == np.float32(256.0)).all()
    for (x, y) in [(960, 540), (200, 900), (1700, 650), (1000, 420), (50, 1000), (1333, 777)]:
        want = _contract_sum(oracle.pixel_samples(ocam, scene, x, y, 256, bounces=8, rr_start=2, seed=SEED, math_mode=1))
        assert s[y, x].tobytes() == want.tobytes(), f"pixel ({x},{y}): {s[y, x]} vs {want}"


@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_auto_block_with_a_short_last_block_at_full_size(pta, tracer, oracle, workload):
    """IR_SAMPLE_BLOCK_AUTO on the bench launches: blocks that do not divide the sample count (C2: 8 x 124 + 32, C3: 4 x 60 +
    16), the short last blocks dealt out after all the full ones from a single queue (pt_kernels.hip decode_task).  Every
    pixel got each of its samples once (the weight channel counts them), sampled pixels equal the oracle's per-sample
    values summed in that block structure."""
    from path_tracer_amd import _capi, workloads as W
    if workload == "c2":
        scene, cam, spp, want_block = W.scene_s1(), W.default_camera(W_, H_), 1024, 124
        pixels = [(960, 800), (1100, 560), (300, 1000), (1500, 700), (40, 1079), (1919, 600), (960, 300), (700, 650)]
    else:
        scene, cam, spp, want_block = W.scene_s2(), W.camera_s2(W_, H_), 256, 60
        pixels = [(960, 540), (200, 900), (1700, 650), (1333, 777)]
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    tracer.set_sample_block(_capi.IR_SAMPLE_BLOCK_AUTO)
    try:
        block = tracer.get_sample_block(W_, H_, spp)
        assert block == want_block and spp % block != 0
        tracer.render(scene, W_, H_, False, spp=spp)
        s, c = tracer.get_radiance()
    finally:
        tracer.set_sample_block(BLOCK)
    assert (c == spp).all() and (s[..., 3] == np.float32(spp)).all()
    ocam = oracle.camera_from(cam, W_, H_)
    for (x, y) in pixels:
        want = _contract_sum(oracle.pixel_samples(ocam, scene, x, y, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1), block)
        assert s[y, x].tobytes() == want.tobytes(), f"pixel ({x},{y}): {s[y, x]} vs {want}"


def test_redraw_loop_at_full_hd_equals_reference_style_sums(pta, tracer, oracle):
    """The reference's own call pattern at 1080p (frontend/src/app.rs:84-99): render_updated once, then prepare_pixels per
    redraw, one sample per pixel each (a single-block launch: the lane-per-pixel kernel).  After 8 frames a seeded subset of
    pixels holds the oracle's samples 0..7 added one by one, the order of 8 successive reference calls; every pixel has 8."""
    from path_tracer_amd import workloads as W
    scene = W.scene_s1()
    cam = W.default_camera(W_, H_)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    tracer.render_updated(scene, W_, H_)
    for _ in range(7):
        tracer.prepare_pixels(scene, W_, H_)
    s, c = tracer.get_radiance()
    assert (c == 8).all() and (s[..., 3] == np.float32(8.0)).all()
    ocam = oracle.camera_from(cam, W_, H_)
    rng = np.random.default_rng(77)
    pixels = [(int(rng.integers(0, W_)), int(rng.integers(0, H_))) for _ in range(40)]
    pixels += [(0, 0), (W_ - 1, H_ - 1), (W_ // 2, H_ // 2), (W_ // 2 + 130, H_ // 2 - 40), (960, 300), (1100, 560)]
    for (x, y) in pixels:
        exact = oracle.pixel_samples(ocam, scene, x, y, 8, bounces=8, rr_start=2, seed=SEED, math_mode=1)
        want = np.zeros(4, dtype=np.float32)
        for v in exact:                                   # one merge per call: frame += (0 + sample)
            want += np.array([v[0], v[1], v[2], 1.0], dtype=np.float32)
        assert s[y, x].tobytes() == want.tobytes(), f"pixel ({x},{y}): {s[y, x]} vs {want}"
    # the same 8 samples in one call are one block summed from zero: the same additions in the same order
    tracer.render(scene, W_, H_, False, spp=8)
    s8, c8 = tracer.get_radiance()
    assert s8.tobytes() == s.tobytes() and c8.tobytes() == c.tobytes()
