"""IR_SKY_SAMPLING_MIS (sky importance sampling, an extension: the reference samples the BSDF only,
integrator.rs:64-70,84-89) on the GPU against the oracle's restatement: same counter-based stream, device polynomials
(math_mode=1) => bit-identical sums; against the libm oracle rel-RMSE < 1e-3; unbiased against the default estimator."""
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


@pytest.fixture()
def mis(tracer, oracle):
    from path_tracer_amd import _capi
    tracer.set_sky_sampling(_capi.IR_SKY_SAMPLING_MIS)
    oracle.set_sky_sampling(1)
    yield tracer
    tracer.set_sky_sampling(_capi.IR_SKY_SAMPLING_BSDF)
    oracle.set_sky_sampling(0)


def _sun_sky():
    from path_tracer_amd import workloads as W
    return W.synthetic_sky(512, 256, sun_scale=4000.0)


@pytest.mark.parametrize("sky", ["synthetic", "sun"])
@pytest.mark.parametrize("bounces,rr", [(8, 2), (5, 0), (2, 2), (1, 0)])
def test_bit_exact_same_stream(pta, mis, oracle, sky, bounces, rr):
    from path_tracer_amd import workloads as W
    w, h, spp = 160, 90, 16
    scene = W.scene_s1(skybox="synthetic" if sky == "synthetic" else _sun_sky(), emitter=(sky == "synthetic"))
    cam = W.default_camera(w, h)
    mis.set_active_camera(cam)
    mis.set_integrator(bounces=bounces, max_compulsory_bounces=rr, seed=SEED)
    mis.render(scene, w, h, False, spp=spp)
    s, c = mis.get_radiance()
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=bounces, rr_start=rr, seed=SEED, math_mode=1)
    assert c.tobytes() == oc.tobytes()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"


@pytest.mark.parametrize("sky_filter", ["nearest", "bilinear"])
def test_lbvh_kernel_bit_exact(pta, mis, oracle, sky_filter):
    """The LBVH integrator runs the shadow ray as one more traversal of its lane state machine: same sums as the oracle's
    scan, on the 3-sphere scene (forced LBVH) and on 10 000 spheres under a sky image."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 96, 54, 6
    mis.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    mis.set_accel(_capi.IR_ACCEL_LBVH)
    if sky_filter == "bilinear":
        mis.set_sky_filter(_capi.IR_SKY_BILINEAR)
        oracle.set_sky_filter(1)
    try:
        for scene, cam in ((W.scene_s1(), W.default_camera(w, h)),
                           (W.scene_s2(skybox=W.synthetic_sky(256, 128, sun_scale=500.0)), W.camera_s2(w, h))):
            mis.set_active_camera(cam)
            mis.enable_stats()
            mis.render(scene, w, h, False, spp=spp)
            s, c = mis.get_radiance()
            st = mis.get_stats()
            assert mis.get_accel_info()["accel"] == _capi.IR_ACCEL_LBVH
            os_, oc, _, ost = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED,
                                            math_mode=1, want_stats=True)
            assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"
            assert {k: st[k] for k in ("samples", "traces", "tests", "hits")} == {k: ost[k] for k in ("samples", "traces", "tests", "hits")}
    finally:
        mis.set_accel(_capi.IR_ACCEL_AUTO)
        mis.set_sky_filter(_capi.IR_SKY_NEAREST)
        oracle.set_sky_filter(0)


def test_bilinear_filter_and_stats(pta, mis, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 128, 72, 8
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    mis.set_active_camera(cam)
    mis.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    mis.set_sky_filter(_capi.IR_SKY_BILINEAR)
    oracle.set_sky_filter(1)
    try:
        mis.enable_stats()
        mis.render(scene, w, h, False, spp=spp)
        s, c = mis.get_radiance()
        st = mis.get_stats()
        os_, oc, _, ost = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1,
                                        want_stats=True)
    finally:
        mis.set_sky_filter(_capi.IR_SKY_NEAREST)
        oracle.set_sky_filter(0)
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.abs(s - os_).max()}"
    # shadow rays are traces too
    assert {k: st[k] for k in ("samples", "traces", "tests", "disc_pos", "hits")} == ost


def test_vs_libm_oracle_and_blocks(pta, mis, oracle):
    """Against libm transcendentals (what a Rust build of this estimator would call): rel-RMSE < 1e-3; several sample
    blocks and accumulation onto an existing frame keep working in this mode."""
    from path_tracer_amd import workloads as W
    w, h = 96, 54
    scene = W.scene_s1(skybox=_sun_sky(), emitter=False)
    cam = W.default_camera(w, h)
    mis.set_active_camera(cam)
    mis.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    mis.render(scene, w, h, False, spp=300)
    mis.render(scene, w, h, True, spp=40)
    s, c = mis.get_radiance()
    ocam = oracle.camera_from(cam, w, h)
    o1, c1, _, _ = oracle.render(ocam, scene, 300, bounces=8, rr_start=2, seed=SEED, math_mode=1)
    o1, c1, _, _ = oracle.render(ocam, scene, 40, bounces=8, rr_start=2, seed=SEED, math_mode=1, sample_base=300, sums=o1, counts=c1)
    assert np.array_equal(s.view(np.uint32), o1.view(np.uint32))
    l1, lc, _, _ = oracle.render(ocam, scene, 300, bounces=8, rr_start=2, seed=SEED, math_mode=0)
    l1, lc, _, _ = oracle.render(ocam, scene, 40, bounces=8, rr_start=2, seed=SEED, math_mode=0, sample_base=300, sums=l1, counts=lc)
    assert rel_rmse(s, l1) < 1e-3


def test_same_mean_as_reference_estimator_lower_noise(pta, tracer, oracle):
    """On the GPU alone: MIS and the reference's BSDF-only estimator converge to the same image; under the sun-dominated
    sky MIS gets there with far less noise."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 160, 90
    scene = W.scene_s1(skybox=_sun_sky(), emitter=False)
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    img, noise = {}, {}
    for mode in (_capi.IR_SKY_SAMPLING_BSDF, _capi.IR_SKY_SAMPLING_MIS):
        tracer.set_sky_sampling(mode)
        try:
            tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
            tracer.render(scene, w, h, False, spp=8192)
            s, c = tracer.get_radiance()
            img[mode] = s[..., :3] / c[..., None]
            halves = []
            for seed in (1, 2):
                tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=seed)
                tracer.render(scene, w, h, False, spp=64)
                s, c = tracer.get_radiance()
                halves.append(s[..., :3] / c[..., None])
            noise[mode] = float(np.sqrt(((halves[0] - halves[1]) ** 2).mean()))
        finally:
            tracer.set_sky_sampling(_capi.IR_SKY_SAMPLING_BSDF)
    a, b = img[_capi.IR_SKY_SAMPLING_BSDF], img[_capi.IR_SKY_SAMPLING_MIS]
    assert np.allclose(a.mean(axis=(0, 1)), b.mean(axis=(0, 1)), rtol=0.01), (a.mean(axis=(0, 1)), b.mean(axis=(0, 1)))
    assert noise[_capi.IR_SKY_SAMPLING_MIS] < noise[_capi.IR_SKY_SAMPLING_BSDF] / 8, noise


def test_constant_sky_unchanged_and_unsupported_combinations(pta, mis, oracle):
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 64, 36, 4
    cam = W.default_camera(w, h)
    mis.set_active_camera(cam)
    mis.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    # no sky image: nothing to importance-sample, the mode is inert
    scene = W.scene_s1(skybox=None)
    mis.render(scene, w, h, False, spp=spp)
    a, _ = mis.get_radiance()
    mis.set_sky_sampling(_capi.IR_SKY_SAMPLING_BSDF)
    mis.render(scene, w, h, False, spp=spp)
    b, _ = mis.get_radiance()
    assert a.tobytes() == b.tobytes()
    mis.set_sky_sampling(_capi.IR_SKY_SAMPLING_MIS)
    sky_scene = W.scene_s1()
    with pytest.raises(pta.InsplorayError):
        mis.set_sky_sampling(7)
    # replacing the sky image rebuilds the tables
    mis.render(sky_scene, w, h, False, spp=spp)
    first, _ = mis.get_radiance()
    other = W.scene_s1(skybox=_sun_sky())
    mis.render(other, w, h, False, spp=spp)
    s, c = mis.get_radiance()
    os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), other, spp, bounces=8, rr_start=2, seed=SEED, math_mode=1)
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)) and not np.array_equal(s, first)


@pytest.mark.parametrize("accel", ["scan", "lbvh"])
@pytest.mark.parametrize("bilinear", [False, True])
def test_wavefront_schedule_same_bits(pta, mis, oracle, accel, bilinear):
    """The sky strategy in the wavefront schedule (shadow ray inside the bounce stages): same accumulator bits as the
    regeneration kernels and the oracle, scan and tree, several batches per block and several blocks."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 96, 54, 24
    scene = W.scene_s1(skybox=_sun_sky(), emitter=False)
    cam = W.default_camera(w, h)
    mis.set_active_camera(cam)
    mis.set_integrator(bounces=6, max_compulsory_bounces=2, seed=SEED)
    mis.set_accel(_capi.IR_ACCEL_LBVH if accel == "lbvh" else _capi.IR_ACCEL_LINEAR)
    mis.set_sky_filter(_capi.IR_SKY_BILINEAR if bilinear else _capi.IR_SKY_NEAREST)
    oracle.set_sky_filter(1 if bilinear else 0)
    mis.set_sample_block(16)
    oracle.set_sample_block(16)
    try:
        out = {}
        for name, sched, paths in (("wavefront", _capi.IR_SCHEDULE_WAVEFRONT, 65536), ("regeneration", _capi.IR_SCHEDULE_REGENERATION, 0)):  # wavefront first: it must build the sky tables itself
            mis.set_schedule(sched, paths)
            mis.render(scene, w, h, False, spp=spp)
            out[name] = mis.get_radiance()
        assert out["wavefront"][0].tobytes() == out["regeneration"][0].tobytes()
        assert out["wavefront"][1].tobytes() == out["regeneration"][1].tobytes()
        os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=6, rr_start=2, seed=SEED, math_mode=1)
        assert np.array_equal(out["wavefront"][0].view(np.uint32), os_.view(np.uint32)) and np.array_equal(out["wavefront"][1], oc)
    finally:
        mis.set_schedule(_capi.IR_SCHEDULE_AUTO)
        mis.set_accel(_capi.IR_ACCEL_AUTO)
        mis.set_sky_filter(_capi.IR_SKY_NEAREST)
        oracle.set_sky_filter(0)
        mis.set_sample_block(128)
        oracle.set_sample_block(128)
