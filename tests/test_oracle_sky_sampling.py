"""Sky importance sampling (the build's IR_SKY_SAMPLING_MIS, an extension; the reference samples the BSDF only,
integrator.rs:64-70,84-89): properties of the oracle's restatement that do not need a GPU.

 * the tables are a probability distribution (monotone CDFs ending at 1, density integrating to 1 over the sphere);
 * the estimator has the reference estimator's expectation (means agree within the Monte Carlo error) and much lower
   variance under a sun-dominated sky (workload C5's lighting);
 * with no sky image the mode changes nothing.
The reference holds no fixture for this (it has no such mode): parity unpinned beyond these properties."""
import numpy as np
import pytest

from path_tracer_amd import workloads as W


@pytest.fixture()
def mis(oracle):
    oracle.set_sky_sampling(1)
    yield oracle
    oracle.set_sky_sampling(0)


def test_tables_are_a_distribution(oracle):
    sky = W.synthetic_sky(128, 64, sun_scale=500.0)
    m, c, p = oracle.sky_distribution(sky)
    assert m[0] == 0.0 and m[-1] == 1.0 and np.all(np.diff(m) >= 0)
    assert np.all(c[:, 0] == 0.0) and np.all(c[:, -1] == 1.0) and np.all(np.diff(c, axis=1) >= 0)
    # pdf = P(texel) w h / (2 pi^2): summing P gives 1
    total = p.astype(np.float64).sum() * (2 * np.pi ** 2) / (128 * 64)
    assert abs(total - 1.0) < 1e-5
    # the sun's texel is the most likely one
    lum = (sky.pixels_buffer.reshape(64, 128, 3) * np.array([0.2126, 0.7152, 0.0722])).sum(-1)
    assert np.unravel_index(np.argmax(p), p.shape)[0] == np.unravel_index(np.argmax(lum), lum.shape)[0]


def test_black_and_non_finite_images_fall_back_to_uniform(oracle):
    from path_tracer_amd.workloads import ExrImage
    w, h = 16, 8
    for fill in (0.0, -1.0, np.inf, np.nan):
        img = ExrImage(np.full((w * h, 3), fill, dtype=np.float32), w, h)
        m, c, p = oracle.sky_distribution(img)
        assert m[-1] == 1.0 and np.all(np.diff(m) > 0)
        assert np.allclose(c[3], np.arange(w + 1) / w)
        rows = np.sin(np.pi * (np.arange(h) + 0.5) / h)
        assert np.allclose(np.diff(m), rows / rows.sum(), rtol=1e-5)


def test_same_expectation_lower_variance(oracle):
    w, h = 32, 18
    sky = W.synthetic_sky(256, 128, sun_scale=4000.0)
    scene = W.scene_s1(skybox=sky, emitter=False)
    cam = oracle.camera_from(W.default_camera(w, h), w, h)
    mean, noise = {}, {}
    for mode in (0, 1):
        oracle.set_sky_sampling(mode)
        try:
            s, c, _, _ = oracle.render(cam, scene, 2048, bounces=8, rr_start=2, seed=11)
            a, ca, _, _ = oracle.render(cam, scene, 128, bounces=8, rr_start=2, seed=1)
            b, cb, _, _ = oracle.render(cam, scene, 128, bounces=8, rr_start=2, seed=2)
        finally:
            oracle.set_sky_sampling(0)
        mean[mode] = (s[..., :3] / c[..., None]).mean(axis=(0, 1))
        noise[mode] = float(np.sqrt((((a - b)[..., :3] / 128) ** 2).mean()))
    assert np.allclose(mean[0], mean[1], rtol=0.02), (mean, noise)
    assert noise[1] < noise[0] / 8, noise


def test_constant_sky_is_unchanged(mis):
    w, h = 24, 14
    scene = W.scene_s1(skybox=None)
    cam = mis.camera_from(W.default_camera(w, h), w, h)
    a, _, _, _ = mis.render(cam, scene, 8, bounces=8, rr_start=2)
    mis.set_sky_sampling(0)
    b, _, _, _ = mis.render(cam, scene, 8, bounces=8, rr_start=2)
    assert a.tobytes() == b.tobytes()


def test_primary_sky_pixels_and_depth_one_are_unchanged(mis):
    """Camera rays are not sampled from anything (weight 1), and with bounces = 1 no bounce ray is ever traced."""
    w, h = 24, 14
    scene = W.scene_s1()
    cam = mis.camera_from(W.default_camera(w, h), w, h)
    a, _, _, _ = mis.render(cam, scene, 4, bounces=1, rr_start=2)
    mis.set_sky_sampling(0)
    b, _, _, _ = mis.render(cam, scene, 4, bounces=1, rr_start=2)
    assert a.tobytes() == b.tobytes()
