"""BASELINE.json's configs C1, C4 and C5 at their full sizes (C2 and C3: tests/test_gpu_full_size.py).

C1 (800x450, 64 spp) is small enough for the oracle to render in full: the whole frame is compared bit for bit.
C4 (3840x2160, 4096 spp, tile partition over 8 ranks) and C5 (env-lit S1, 1920x1080, 65 536 spp, sample-range partition
over 2/4/8 ranks) are compared with the oracle on seeded pixels, whose samples the oracle evaluates one by one and which
are summed here in the accumulation contract's order, and through what does not depend on the size: the 8 tile ranks add
up to the whole frame bit for bit, the sample-range ranks add up to it within f32 summation order.  One GPU plays every
rank in turn (the split is a pure function of (rank, world), reference analogue renderer.rs:105-159)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF
BLOCK = 128


@pytest.fixture(scope="module")
def tracer():
    import path_tracer_amd as pta
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    yield rt
    rt.close()


def contract_sum(samples, block=BLOCK):
    """(spp, 3) per-sample radiance -> f32 (r, g, b, weight) as ir_render defines it: blocks of `block` samples summed
    from zero in sample order, block sums merged in block order (np.add.accumulate adds sequentially)."""
    total = np.zeros(4, dtype=np.float32)
    for s0 in range(0, samples.shape[0], block):
        blk = samples[s0:s0 + block]
        rows = np.zeros((blk.shape[0] + 1, 4), dtype=np.float32)
        rows[1:, :3] = blk
        rows[1:, 3] = 1.0
        total = total + np.add.accumulate(rows, axis=0, dtype=np.float32)[-1]
    return total


def test_c1_full_frame_equals_the_oracle(tracer, oracle):
    """C1: the reference's own CPU-sized case, every pixel of the frame."""
    from path_tracer_amd import workloads as W
    w, h, spp = 800, 450, 64
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    tracer.render(scene, w, h, False, spp=spp)
    s, c = tracer.get_radiance()
    hits = tracer.get_primary_hits()
    ocam = oracle.camera_from(cam, w, h)
    oidx, _ = oracle.primary_hits(ocam, scene)
    assert hits.tobytes() == oidx.tobytes()
    want, wcnt, _, _ = oracle.render(ocam, scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1, threads=16)
    assert np.array_equal(s.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(c, wcnt)
    libm, _, _, _ = oracle.render(ocam, scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=0, threads=16)
    d = s[..., :3].astype(np.float64) - libm[..., :3].astype(np.float64)
    rel_rmse = np.sqrt((d ** 2).sum(-1).mean()) / np.sqrt((libm[..., :3].astype(np.float64) ** 2).sum(-1).mean())
    assert rel_rmse < 1e-3, rel_rmse  # north_star's bar against libm transcendentals
    argb = tracer.get_output()
    ref = oracle.resolve_argb(want, wcnt)
    diff = np.abs(((argb[:, None] >> np.array([24, 16, 8, 0])) & 255).astype(np.int32) -
                  ((ref[:, None] >> np.array([24, 16, 8, 0])) & 255).astype(np.int32))
    assert diff.max() <= 1  # device powf vs libm


def test_c4_4k_4096spp_tile_partition_over_8_ranks(tracer, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp, world = 3840, 2160, 4096, 8
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    tracer.render(scene, w, h, False, spp=spp)
    s, c = tracer.get_radiance()
    s, c = s.copy(), c.copy()
    assert (c == spp).all() and (s[..., 3] == np.float32(spp)).all()
    ocam = oracle.camera_from(cam, w, h)
    rng = np.random.default_rng(404)
    pixels = [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(6)]
    pixels += [(0, 0), (w - 1, h - 1), (w // 2, h // 2), (2200, 1120)]
    for (x, y) in pixels:
        exact = oracle.pixel_samples(ocam, scene, x, y, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
        want = contract_sum(exact)
        assert s[y, x].tobytes() == want.tobytes(), f"pixel ({x},{y}): {s[y, x]} vs {want}"
    total = np.zeros_like(s)
    counts = np.zeros_like(c)
    try:
        for rank in range(world):
            tracer.set_partition(_capi.IR_PARTITION_TILES, rank, world)
            tracer.render(scene, w, h, False, spp=spp)
            ps, pc = tracer.get_radiance()
            owned = pc != 0
            # 64x64 tiles dealt round-robin: 60 x 34 tiles at 4K
            ty, tx = np.nonzero(owned[::64, ::64])
            assert ((ty * 60 + tx) % world == rank).all() and len(ty) == len(range(rank, 60 * 34, world))
            assert not (counts * pc).any()  # disjoint supports
            total += ps
            counts += pc
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    assert total.tobytes() == s.tobytes() and counts.tobytes() == c.tobytes()


def test_c5_65536spp_env_lit_sample_range_partition(tracer, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 1920, 1080, 65536
    scene = W.scene_s1(emitter=False, skybox=W.synthetic_sky(sun_scale=4000.0))
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    tracer.render(scene, w, h, False, spp=spp)
    assert tracer.last_kernel_ms()[1] >= 16  # the call is cut into launches of at most 4096 spp
    s, c = tracer.get_radiance()
    s, c = s.copy(), c.copy()
    assert (c == spp).all() and (s[..., 3] == np.float32(spp)).all()
    ocam = oracle.camera_from(cam, w, h)
    worst = 0.0
    for (x, y) in [(960, 540), (1100, 560), (700, 200), (300, 900), (1919, 0), (1000, 330)]:
        exact = oracle.pixel_samples(ocam, scene, x, y, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
        want = contract_sum(exact)
        assert s[y, x].tobytes() == want.tobytes(), f"pixel ({x},{y}): {s[y, x]} vs {want}"
        libm = contract_sum(oracle.pixel_samples(ocam, scene, x, y, spp, bounces=8, rr_start=2, seed=SEED, math_mode=0))
        worst = max(worst, float(np.abs(s[y, x, :3] - libm[:3]).max() / max(float(libm[:3].max()), 1e-6)))
    assert worst < 1e-3, worst
    finite = np.isfinite(s).all(-1)
    assert (~finite).sum() <= 64  # 0/0 weights, about one per 4e9 bounces: 136 G samples here
    try:
        for world in (2, 4, 8):
            total = np.zeros(s.shape, dtype=np.float64)
            for rank in range(world):
                tracer.set_partition(_capi.IR_PARTITION_SAMPLES, rank, world)
                tracer.render(scene, w, h, False, spp=spp // world)
                ps, pc = tracer.get_radiance()
                assert (pc == spp // world).all()
                if rank == 1:
                    # rank 1 renders sample indices [spp/world, 2 spp/world): check one pixel against the oracle at that base
                    ex = oracle.pixel_samples(ocam, scene, 1100, 560, spp // world, bounces=8, rr_start=2, seed=SEED,
                                              math_mode=1, sample_base=spp // world)
                    assert ps[560, 1100].tobytes() == contract_sum(ex).tobytes()
                total += ps
            assert (total[..., 3] == spp).all()
            ok = finite & np.isfinite(total).all(-1)
            assert np.allclose(total[ok][:, :3], s[ok][:, :3], rtol=1e-4, atol=1e-3), world
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)


def test_c5_sky_resident_in_hbm_nearest_and_bilinear(tracer, oracle):
    """north_star's "EXR environment lookup from HBM": the C5 frame under an 8192 x 4096 sky (403 MB
    as Vec<Vec3>, 537 MB as float4 on the device, twice the Infinity Cache), with the reference's nearest lookup
    (exr.rs:13-24) and with the bilinear variant.  Sampled pixels against the oracle's per-sample values summed in the
    contract's order, bit for bit; the sky pixels (one lookup per pixel and launch) included."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 1920, 1080, 4096
    sky = W.synthetic_sky(8192, 4096, sun_scale=4000.0)
    scene = W.scene_s1(emitter=False, skybox=sky)
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    ocam = oracle.camera_from(cam, w, h)
    pixels = [(960, 540), (1100, 560), (700, 200), (300, 900), (1919, 0), (1000, 330), (5, 1075), (1500, 620)]
    results = {}
    try:
        for name, filt in (("nearest", _capi.IR_SKY_NEAREST), ("bilinear", _capi.IR_SKY_BILINEAR)):
            tracer.set_sky_filter(filt)
            oracle.set_sky_filter(filt)
            tracer.render(scene, w, h, False, spp=spp)
            s, c = tracer.get_radiance()
            results[name] = s.copy()
            assert (c == spp).all()
            for (x, y) in pixels:
                exact = oracle.pixel_samples(ocam, scene, x, y, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
                want = contract_sum(exact)
                assert s[y, x].tobytes() == want.tobytes(), f"{name} pixel ({x},{y}): {s[y, x]} vs {want}"
    finally:
        tracer.set_sky_filter(_capi.IR_SKY_NEAREST)
        oracle.set_sky_filter(0)
    # the two filters read different texels: the frames differ, but only by what a texel's neighbours differ by
    a, b = results["nearest"][..., :3], results["bilinear"][..., :3]
    ok = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    assert not np.array_equal(a, b)
    assert abs(float(a[ok].mean()) / float(b[ok].mean()) - 1.0) < 0.02
