"""Seeded random configurations, HIP path against the oracle, bit for bit: scenes (0-300 spheres incl. zero and huge
radii, overlapping and coincident spheres, out-of-range material ids, emissive / black / over-unity albedos), camera
poses, frame shapes that cut tiles and blocks, depth / roulette / spp / sample-block settings, sky kinds and filters,
both closest-hit structures, both builders, both schedules, both sky estimators."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pta():
    import path_tracer_amd
    return path_tracer_amd


@pytest.fixture(scope="module")
def tracer(pta):
    rt = pta.RayTracer(0, 0, device=0)
    yield rt
    rt.close()


def _random_case(rng, large=False):
    from path_tracer_amd import workloads as W
    n = int(rng.choice([1000, 2047, 2048, 3000, 5000])) if large else int(rng.choice([0, 1, 2, 3, 5, 17, 63, 64, 65, 130, 300]))
    xyzr = np.zeros((n, 4), dtype=np.float32)
    xyzr[:, 0] = rng.uniform(-6, 6, n)
    xyzr[:, 1] = rng.uniform(-2, 4, n)
    xyzr[:, 2] = rng.uniform(-14, 1, n)
    xyzr[:, 3] = rng.uniform(0.05, 1.2, n)
    if n >= 3:
        xyzr[0] = (0.0, -100.5, 0.0, 100.0)       # ground
        xyzr[1, 3] = 0.0                            # a point
        xyzr[2] = xyzr[min(3, n - 1)]               # coincident pair
    if n >= 17:
        xyzr[5, 3] = -0.7                           # negative radius (r*r is what the test uses)
        xyzr[6, :3] = xyzr[7, :3]                   # concentric
    m = int(rng.integers(1, 6))
    mats = np.zeros((m, 9), dtype=np.float32)
    mats[:, :3] = rng.uniform(0.0, 1.0, (m, 3))
    mats[:, 3] = 0.5
    if m > 1:
        mats[1, 5:8] = rng.uniform(0.2, 1.0, 3)
        mats[1, 8] = rng.uniform(0.5, 6.0)
    if m > 2:
        mats[2, :3] = 0.0                           # black: contribution becomes +0
    if m > 3:
        mats[3, :3] = (1.5, 0.2, 1.0)               # over-unity albedo
    mid = rng.integers(-2, m + 2, n).astype(np.int32)   # some ids out of range -> default material
    sky_kind = rng.choice(["none", "synthetic", "tiny"])
    sky = None
    if sky_kind == "synthetic":
        sky = W.synthetic_sky(64, 32, sun_scale=float(rng.choice([40.0, 2000.0])))
    elif sky_kind == "tiny":
        sky = W.ExrImage(rng.uniform(0, 2, (3 * 2, 3)).astype(np.float32), 3, 2)
    scene = W.FlatScene(xyzr, mid, mats, default_sky_color=tuple(rng.uniform(0, 1, 3)), skybox=sky)
    w, h = int(rng.choice([1, 7, 31, 64, 65, 97])), int(rng.choice([1, 5, 33, 64, 70]))
    if large:
        w, h = min(w, 65), min(h, 33)
        xyzr[:, 0] *= 3.0
        xyzr[:, 2] *= 3.0
    from path_tracer_amd import PinholeCamera
    cam = PinholeCamera(tuple(rng.uniform(-1, 1, 3) + np.array([0, 0.5, 3.0])), tuple(rng.uniform(-0.4, 0.4, 3)),
                        float(rng.uniform(20, 60)), float(rng.uniform(30, 60)), [w, h])
    cfg = dict(bounces=int(rng.choice([0, 1, 2, 5, 8, 12])), rr=int(rng.choice([0, 1, 2, 20])),
               spp=int(rng.choice([1, 2, 3, 7])), block=int(rng.choice([1, 2, 128])),
               seed=int(rng.integers(0, 2 ** 63)), accel=str(rng.choice(["auto", "linear", "lbvh"])),
               build=str(rng.choice(["host", "device", "refit"])), schedule=str(rng.choice(["regeneration", "wavefront"])),
               bilinear=bool(rng.integers(0, 2)) and sky is not None, mis=bool(rng.integers(0, 2)))
    return scene, cam, w, h, cfg


@pytest.mark.parametrize("case", list(range(200)) + [f"large{i}" for i in range(40)])
def test_random_configuration_bit_exact(pta, tracer, oracle, case):
    from path_tracer_amd import _capi
    large = isinstance(case, str)
    # IR_FUZZ_SHIFT=<n> moves every case to another seed (one-off wider runs: IR_FUZZ_SHIFT=100000 pytest tests/test_gpu_fuzz.py)
    rng = np.random.default_rng(9000 + int(os.environ.get("IR_FUZZ_SHIFT", "0")) + (int(case[5:]) + 5000 if large else case))
    scene, cam, w, h, cfg = _random_case(rng, large)
    if large and cfg["accel"] == "linear":
        cfg["accel"] = "auto"
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=cfg["bounces"], max_compulsory_bounces=cfg["rr"], seed=cfg["seed"])
    tracer.set_sample_block(cfg["block"])
    oracle.set_sample_block(cfg["block"])
    tracer.set_accel({"auto": _capi.IR_ACCEL_AUTO, "linear": _capi.IR_ACCEL_LINEAR, "lbvh": _capi.IR_ACCEL_LBVH}[cfg["accel"]])
    tracer.set_accel_build({"host": _capi.IR_BUILD_HOST, "device": _capi.IR_BUILD_DEVICE, "refit": _capi.IR_BUILD_REFIT}[cfg["build"]])
    tracer.set_schedule(_capi.IR_SCHEDULE_WAVEFRONT if cfg["schedule"] == "wavefront" else _capi.IR_SCHEDULE_REGENERATION)
    tracer.set_sky_filter(_capi.IR_SKY_BILINEAR if cfg["bilinear"] else _capi.IR_SKY_NEAREST)
    oracle.set_sky_filter(1 if cfg["bilinear"] else 0)
    tracer.set_sky_sampling(_capi.IR_SKY_SAMPLING_MIS if cfg["mis"] else _capi.IR_SKY_SAMPLING_BSDF)
    oracle.set_sky_sampling(1 if cfg["mis"] else 0)
    try:
        idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
        tracer.render(scene, w, h, False, spp=cfg["spp"])
        tracer.render(scene, w, h, True, spp=cfg["spp"])
        s, c = tracer.get_radiance()
        ocam = oracle.camera_from(cam, w, h)
        oidx, ot = oracle.primary_hits(ocam, scene)
        kw = dict(bounces=cfg["bounces"], rr_start=cfg["rr"], seed=cfg["seed"], math_mode=1)
        os_, oc, _, _ = oracle.render(ocam, scene, cfg["spp"], **kw)
        os_, oc, _, _ = oracle.render(ocam, scene, cfg["spp"], sample_base=cfg["spp"], sums=os_, counts=oc, **kw)
    finally:
        tracer.set_sample_block(128)
        oracle.set_sample_block(128)
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
        tracer.set_accel_build(_capi.IR_BUILD_AUTO)
        tracer.set_schedule(_capi.IR_SCHEDULE_AUTO)
        tracer.set_sky_filter(_capi.IR_SKY_NEAREST)
        oracle.set_sky_filter(0)
        tracer.set_sky_sampling(_capi.IR_SKY_SAMPLING_BSDF)
        oracle.set_sky_sampling(0)
    assert idx.tobytes() == oidx.tobytes(), cfg
    assert t.tobytes() == ot.tobytes(), cfg
    assert c.tobytes() == oc.tobytes(), cfg
    same = (s.view(np.uint32) == os_.view(np.uint32)) | (np.isnan(s) & np.isnan(os_))
    assert same.all(), (cfg, float(np.nanmax(np.abs(s - os_))))
