"""GPU parity tests proper: the HIP path (through the C ABI, via the Python mirror of the reference API) against the
CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star):
  * primary-ray closest-hit sphere indices: bit-identical (memcmp);
  * radiance, same counter-based stream, oracle with the device's polynomial sin/cos/acos/atan2 (math_mode=1):
    bit-identical sums (stronger than required; proves the integrator logic and operation order);
  * radiance vs the libm oracle (math_mode=0, what the Rust reference calls): relative RMSE < 1e-3;
  * resolved ARGB: at most 1 LSB per channel (device powf vs libm powf).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x0123456789ABCDEF


def rel_rmse(a, b):
    d = (a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    return float(np.sqrt((d ** 2).sum(-1).mean()) / np.sqrt((b[..., :3].astype(np.float64) ** 2).sum(-1).mean()))


@pytest.fixture(scope="module")
def pta():
    import path_tracer_amd
    return path_tracer_amd


@pytest.fixture(scope="module")
def tracer(pta):
    rt = pta.RayTracer(0, 0, device=0)
    yield rt
    rt.close()


def _setup(pta, tracer, scene, cam, bounces, rr):
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=bounces, max_compulsory_bounces=rr, seed=SEED)


@pytest.mark.parametrize("w,h", [(800, 450), (97, 61), (64, 64), (1, 1)])
def test_primary_hits_bit_identical_s1(pta, tracer, oracle, w, h):
    from path_tracer_amd import workloads as W
    scene = W.scene_s1(skybox=None)
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
    oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
    assert idx.tobytes() == oidx.tobytes()
    assert t.tobytes() == ot.tobytes()
    if w >= 64:
        assert set(np.unique(idx)) >= {-1, 1}


@pytest.mark.parametrize("accel", ["linear", "lbvh"])
def test_primary_hits_bit_identical_10k(pta, tracer, oracle, accel):
    """C3 scene: the LDS-staged scan and the LBVH both return the reference scan's (index, t) bit for bit."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 480, 270
    scene = W.scene_s2()
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    tracer.set_accel(_capi.IR_ACCEL_LINEAR if accel == "linear" else _capi.IR_ACCEL_LBVH)
    try:
        idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
        info = tracer.get_accel_info()
    finally:
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    assert info["accel"] == (_capi.IR_ACCEL_LINEAR if accel == "linear" else _capi.IR_ACCEL_LBVH)
    oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
    assert idx.tobytes() == oidx.tobytes()
    assert t.tobytes() == ot.tobytes()
    assert len(np.unique(idx)) > 100


@pytest.mark.parametrize("accel", ["linear", "lbvh"])
def test_radiance_bit_exact_10k(pta, tracer, oracle, accel):
    """Bounce rays (non-unit directions, origins on sphere surfaces) through both structures == oracle scan."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 96, 54, 4
    scene = W.scene_s2()
    cam = W.camera_s2(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.set_accel(_capi.IR_ACCEL_LINEAR if accel == "linear" else _capi.IR_ACCEL_LBVH)
    try:
        tracer.enable_stats()
        tracer.render(scene, w, h, False, spp=spp)
        s, c = tracer.get_radiance()
        st = tracer.get_stats()
        ast = tracer.get_accel_stats()
        # the same launch through the kernels without statistics (for the scan: the 1024-thread workgroups that 160 KB of
        # staged spheres select, pt_kernels.hip launch_render)
        tracer.render(scene, w, h, False, spp=spp)
        s_plain, c_plain = tracer.get_radiance()
    finally:
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    assert s_plain.tobytes() == s.tobytes() and c_plain.tobytes() == c.tobytes()
    os_, oc, _, ost = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED,
                                    math_mode=1, want_stats=True)
    assert c.tobytes() == oc.tobytes()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"
    # algorithmic counters: traces, hits and scan-equivalent tests are structure independent
    assert {k: st[k] for k in ("samples", "traces", "tests", "hits")} == {k: ost[k] for k in ("samples", "traces", "tests", "hits")}
    if accel == "linear":
        assert st["disc_pos"] == ost["disc_pos"] and ast == {"bvh_node_visits": 0, "bvh_leaf_tests": 0}
    else:
        assert 0 < ast["bvh_leaf_tests"] < st["tests"] // 50 and ast["bvh_node_visits"] > 0


def test_more_spheres_than_lds_holds(pta, tracer, oracle):
    """20 000 spheres cannot be staged in LDS: AUTO must pick the LBVH, LINEAR must be refused, hits == oracle scan."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 160, 90
    scene = W.scene_s2(n=20000)
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
    assert tracer.get_accel_info()["accel"] == _capi.IR_ACCEL_LBVH
    oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
    assert idx.tobytes() == oidx.tobytes() and t.tobytes() == ot.tobytes()
    with pytest.raises(pta.InsplorayError):
        tracer.set_accel(_capi.IR_ACCEL_LINEAR)
    tracer.set_accel(_capi.IR_ACCEL_AUTO)


@pytest.mark.parametrize("skybox", ["synthetic", None])
@pytest.mark.parametrize("bounces,rr", [(8, 2), (5, 2), (1, 0), (3, 5)])
def test_radiance_bit_exact_same_stream(pta, tracer, oracle, skybox, bounces, rr):
    from path_tracer_amd import workloads as W
    w, h, spp = 160, 90, 32
    scene = W.scene_s1(skybox=skybox)
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, bounces, rr)
    tracer.render(scene, w, h, False, spp=spp)
    s, c = tracer.get_radiance()
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=bounces, rr_start=rr, seed=SEED,
                                  math_mode=1)
    assert c.tobytes() == oc.tobytes()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"


def test_bilinear_sky_bit_exact_same_stream(pta, tracer, oracle):
    """IR_SKY_BILINEAR (north-star variant; the reference is nearest-only) against the oracle's bilinear switch."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 160, 90, 16
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.set_sky_filter(_capi.IR_SKY_BILINEAR)
    oracle.set_sky_filter(1)
    try:
        tracer.render(scene, w, h, False, spp=spp)
        s, c = tracer.get_radiance()
        os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
    finally:
        tracer.set_sky_filter(_capi.IR_SKY_NEAREST)
        oracle.set_sky_filter(0)
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"
    tracer.render(scene, w, h, False, spp=spp)
    s_near, _ = tracer.get_radiance()
    assert not np.array_equal(s, s_near)  # the switch does something
    assert rel_rmse(s, s_near) < 0.05     # ... but the image is the same sky


@pytest.mark.parametrize("schedule", ["regeneration", "wavefront"])
def test_bilinear_sky_footprint_table_whole_frame(pta, tracer, oracle, schedule):
    """A sky of 128 MB or more (as float4) is looked up through the footprint table (four texels of a bilinear cell in one
    64-byte record, pt_device.h sky_bilinear_quad): the whole frame, wrap-around columns and pole rows included, must
    equal the oracle's bilinear lookup bit for bit; a camera that looks straight up and one that looks along the seam."""
    from path_tracer_amd import _capi, workloads as W
    from path_tracer_amd.cameras import PinholeCamera
    w, h, spp = 96, 54, 24
    scene = W.scene_s1(skybox=W.synthetic_sky(4096, 2048))   # 134 MB on the device
    tracer.set_sky_filter(_capi.IR_SKY_BILINEAR)
    tracer.set_schedule(_capi.IR_SCHEDULE_WAVEFRONT if schedule == "wavefront" else _capi.IR_SCHEDULE_REGENERATION)
    oracle.set_sky_filter(1)
    try:
        for cam in (W.default_camera(w, h), PinholeCamera((0.0, 0.3, 2.0), (1.5707964, 0.0, 0.0), 35.0, 55.0, [w, h]),
                    PinholeCamera((0.0, 0.2, 2.0), (0.0, 1.5707964, 0.0), 20.0, 55.0, [w, h])):
            _setup(pta, tracer, scene, cam, 8, 2)
            tracer.render(scene, w, h, False, spp=spp)
            s, c = tracer.get_radiance()
            os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
            assert np.array_equal(c, oc)
            assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"
    finally:
        tracer.set_sky_filter(_capi.IR_SKY_NEAREST)
        tracer.set_schedule(_capi.IR_SCHEDULE_AUTO)
        oracle.set_sky_filter(0)


def test_radiance_vs_libm_oracle_rel_rmse(pta, tracer, oracle):
    from path_tracer_amd import workloads as W
    w, h, spp = 320, 180, 64
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.render(scene, w, h, False, spp=spp)
    s, c = tracer.get_radiance()
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=0)
    assert c.tobytes() == oc.tobytes()
    e = rel_rmse(s, os_)
    assert e < 1e-3, e  # north_star tolerance


def test_accumulate_equals_successive_calls(pta, tracer, oracle):
    """spp samples in one call == spp reference-style 1-sample prepare_pixels calls (bitwise)."""
    from path_tracer_amd import workloads as W
    w, h = 96, 54
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 5, 2)
    tracer.render_updated(scene, w, h)
    for _ in range(6):
        tracer.prepare_pixels(scene, w, h)
    s1, c1 = tracer.get_radiance()
    assert tracer.samples_accumulated() == 7
    tracer.render(scene, w, h, False, spp=7)
    s7, c7 = tracer.get_radiance()
    assert np.array_equal(s1.view(np.uint32), s7.view(np.uint32))
    assert (c1 == 7).all() and (c7 == 7).all()
    os_, _, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, 7, bounces=5, rr_start=2, seed=SEED, math_mode=1)
    assert np.array_equal(s7.view(np.uint32), os_.view(np.uint32))
    # resolution change resets (renderer.rs:84-88)
    tracer.prepare_pixels(scene, w + 1, h)
    assert tracer.samples_accumulated() == 1
    assert tracer.get_current_size() == [w + 1, h]


@pytest.mark.parametrize("block,spp", [(64, 200), (16, 40), (1, 5), (4096, 70), (2, 45), (3, 50)])  # (the last two: 23 + 12 and 17 + 9 blocks, merge_kernel's batches of eight and their remainders)
def test_sample_blocks_match_oracle(pta, tracer, oracle, block, spp):
    """A multi-sample call = consecutive blocks, each summed from zero (the reference's TileAccumulator) and merged in
    order (merge_tile); blocks run as independent workgroups on the GPU, the oracle restates the same sum."""
    from path_tracer_amd import workloads as W
    w, h = 100, 70
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.set_sample_block(block)
    oracle.set_sample_block(block)
    try:
        tracer.render(scene, w, h, False, spp=spp)
        tracer.render(scene, w, h, True, spp=spp // 2 + 1)      # second call accumulates onto a non-zero frame
        s, c = tracer.get_radiance()
        oc = oracle.camera_from(cam, w, h)
        os_, ocnt, _, _ = oracle.render(oc, scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
        oracle.render(oc, scene, spp // 2 + 1, bounces=8, rr_start=2, seed=SEED, math_mode=1, sample_base=spp, sums=os_, counts=ocnt)
    finally:
        tracer.set_sample_block(128)
        oracle.set_sample_block(128)
    assert np.array_equal(c, ocnt) and (c == spp + spp // 2 + 1).all()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"


def test_resolve_argb_within_one_lsb(pta, tracer, oracle):
    from path_tracer_amd import workloads as W
    w, h = 160, 90
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.render(scene, w, h, False, spp=16)
    out = tracer.get_output().copy()
    s, c = tracer.get_radiance()
    ref = oracle.resolve_argb(s, c)
    for shift in (0, 8, 16, 24):
        a = ((out >> shift) & 0xFF).astype(np.int32)
        b = ((ref >> shift) & 0xFF).astype(np.int32)
        assert np.abs(a - b).max() <= 1
    assert ((out >> 24) == 0xBA).all()  # alpha goes through Reinhard + gamma (full_buffer.rs:53-62)


def test_resolve_corners_and_wide_range(pta, tracer, oracle):
    """The tone map's a^(1/2.2) runs on the hardware log2/exp2: against the oracle (libm powf) over 18 decades of radiance and
    at powf's corners — a mean of exactly -1 (Reinhard gives -inf, powf +inf, the byte 255), other negatives and NaN (0),
    -0, denormals, infinities — never more than 1 LSB, and exact at the corners."""
    from path_tracer_amd import workloads as W
    w, h = 256, 128
    tracer.set_active_camera(W.default_camera(w, h))
    tracer.render(W.scene_s1(skybox=None), w, h, False, spp=0)
    rng = np.random.default_rng(5)
    counts = rng.integers(1, 5000, (h, w)).astype(np.uint32)
    mean = np.exp(rng.uniform(np.log(1e-12), np.log(1e6), (h, w, 3))).astype(np.float32)
    sums = np.zeros((h, w, 4), dtype=np.float32)
    sums[..., :3] = mean * counts[..., None]
    sums[..., 3] = counts
    counts[0, :8] = 1
    sums[0, 0] = (np.inf, np.nan, -1.0, 1.0)
    sums[0, 1] = (-1.0, -0.5, -2.0, 1.0)
    sums[0, 2] = (-0.0, 1e-40, 3e38, 1.0)
    sums[0, 3] = (-np.inf, 1e-45, 1.1754944e-38, 1.0)
    sums[0, 4] = (-1.0000001, -0.99999994, 0.0, 1.0)
    sums[0, 5:8] = 0.0
    sums[0, 5:8, 3] = 1.0
    tracer.set_radiance(sums, counts)
    out = tracer.get_output().copy().reshape(h, w)
    ref = oracle.resolve_argb(sums, counts).reshape(h, w)
    for shift in (0, 8, 16, 24):
        d = ((out >> shift) & 0xFF).astype(np.int32) - ((ref >> shift) & 0xFF).astype(np.int32)
        assert np.abs(d).max() <= 1 and (d != 0).mean() < 1e-3
    assert (out[0, :8] == ref[0, :8]).all(), ([hex(int(v)) for v in out[0, :8]], [hex(int(v)) for v in ref[0, :8]])
    assert (out[0, 1] >> 16) & 0xFF == 255 and (out[0, 1] >> 8) & 0xFF == 0      # mean -1 -> 255; mean -0.5 -> NaN -> 0


def test_stats_match_oracle(pta, tracer, oracle):
    from path_tracer_amd import workloads as W
    w, h, spp = 128, 72, 8
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.enable_stats()
    tracer.render(scene, w, h, False, spp=spp)
    st = tracer.get_stats()
    _, _, _, ost = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED,
                                 math_mode=1, want_stats=True)
    assert st == ost


def test_tile_and_sample_partitions_sum_to_whole(pta, tracer, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 200, 130, 8
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    tracer.render(scene, w, h, False, spp=spp)
    whole, wc = tracer.get_radiance()
    # tiles: disjoint supports, sum == whole exactly
    acc = np.zeros_like(whole); cnt = np.zeros_like(wc)
    for r in range(3):
        tracer.set_partition(_capi.IR_PARTITION_TILES, r, 3)
        tracer.render(scene, w, h, False, spp=spp)
        s, c = tracer.get_radiance()
        assert ((c == 0) | (c == spp)).all()
        acc += s; cnt += c
    assert np.array_equal(acc.view(np.uint32), whole.view(np.uint32)) and np.array_equal(cnt, wc)
    # sample ranges: 2 ranks x 4 spp == samples 0..7, equal up to f32 summation order
    acc = np.zeros_like(whole); cnt = np.zeros_like(wc)
    for r in range(2):
        tracer.set_partition(_capi.IR_PARTITION_SAMPLES, r, 2)
        tracer.render(scene, w, h, False, spp=spp // 2)
        s, c = tracer.get_radiance()
        acc += s; cnt += c
    tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    assert np.array_equal(cnt, wc)
    assert np.allclose(acc, whole, rtol=1e-5, atol=1e-5)


def test_generic_camera_rays_match_pinhole(pta, tracer):
    from path_tracer_amd import workloads as W
    w, h = 40, 24
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    _setup(pta, tracer, scene, cam, 5, 2)
    tracer.render(scene, w, h, False, spp=4)
    s0, _ = tracer.get_radiance()

    class Wrapped(pta.Camera):  # any other `dyn Camera` implementor goes through host-evaluated rays
        def __init__(self, inner):
            self.inner = inner
        def get_ray(self, x, y):
            return self.inner.get_ray(x, y)
        def set_image_resolutions(self, r):
            self.inner.set_image_resolutions(r)

    tracer.set_active_camera(Wrapped(cam))
    tracer.render(scene, w, h, False, spp=4)
    s1, _ = tracer.get_radiance()
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32))


def test_errors_are_reported_not_fatal(pta, tracer):
    from path_tracer_amd import _capi
    with pytest.raises(pta.InsplorayError):
        tracer.set_partition(_capi.IR_PARTITION_TILES, 5, 2)
    with pytest.raises(pta.InsplorayError):
        pta.RayTracer(4, 4, device=4096)


@pytest.mark.parametrize("w,h,spp,bounces", [(97, 61, 1, 8), (160, 90, 3, 8), (64, 64, 4, 5), (33, 17, 2, 0), (1, 1, 1, 8)])
def test_lbvh_low_spp_matches_oracle(pta, tracer, oracle, w, h, spp, bounces):
    """Launches of a few samples per pixel with the LBVH (the frontend's redraw asks for one, app.rs:84-99): same sums as
    the oracle's scan, frame edges that cut 16x16 blocks, zero depth, accumulation over successive calls."""
    from path_tracer_amd import _capi, workloads as W
    scene = W.scene_s2(n=3000, skybox=W.synthetic_sky(128, 64))
    cam = W.camera_s2(w, h)
    _setup(pta, tracer, scene, cam, bounces, 2)
    tracer.set_accel(_capi.IR_ACCEL_LBVH)
    try:
        tracer.render(scene, w, h, False, spp=spp)
        tracer.render(scene, w, h, True, spp=spp)
        s, c = tracer.get_radiance()
        assert tracer.get_accel_info()["accel"] == _capi.IR_ACCEL_LBVH
    finally:
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    ocam = oracle.camera_from(cam, w, h)
    os_, oc, _, _ = oracle.render(ocam, scene, spp, bounces=bounces, rr_start=2, seed=SEED, math_mode=1)
    os_, oc, _, _ = oracle.render(ocam, scene, spp, bounces=bounces, rr_start=2, seed=SEED, math_mode=1, sample_base=spp, sums=os_, counts=oc)
    assert c.tobytes() == oc.tobytes()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"


def test_lbvh_low_spp_tile_partition(pta, tracer, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 200, 120, 2
    scene = W.scene_s2(n=2000)
    cam = W.camera_s2(w, h)
    _setup(pta, tracer, scene, cam, 8, 2)
    tracer.set_accel(_capi.IR_ACCEL_LBVH)
    try:
        total = np.zeros((h, w, 4), dtype=np.float32)
        for rank in range(3):
            tracer.set_partition(_capi.IR_PARTITION_TILES, rank, 3)
            tracer.render(scene, w, h, False, spp=spp)
            s, c = tracer.get_radiance()
            total += s
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
    assert np.array_equal(total.view(np.uint32), os_.view(np.uint32))
