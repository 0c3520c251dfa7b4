"""Device LBVH build (pt_lbvh_build.hip) against the host build (lbvh.cpp): the two builders must give byte-identical
wide nodes, leaf spheres, leaf indices, climb records and traversal starts, and therefore the same hits; both equal the reference's linear scan
(integrator.rs:106-127), checked through the oracle where the scan is affordable.

Scenes cover what the radix tree has to get right: distinct codes, long runs of equal Morton codes (which continue on
the bits of the sorted position), all spheres at one point, non-finite centres and radii, and the smallest trees, and a count above
rocPRIM's switch from merge sort to onesweep."""
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


def _scene(xyzr):
    from path_tracer_amd import workloads as W
    xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
    n = xyzr.shape[0]
    mats = np.zeros((1, 9), dtype=np.float32)
    mats[0, :4] = (0.8, 0.7, 0.6, 0.5)
    return W.FlatScene(xyzr, np.zeros(n, dtype=np.int32), mats, default_sky_color=(0.6, 0.7, 0.9), skybox=None)


def _random(n, seed):
    rng = np.random.default_rng(seed)
    xyzr = np.empty((n, 4), dtype=np.float32)
    xyzr[:, 0] = rng.uniform(-40, 40, n)
    xyzr[:, 1] = rng.uniform(-2, 6, n)
    xyzr[:, 2] = rng.uniform(-80, 0, n)
    xyzr[:, 3] = rng.uniform(0.1, 0.6, n)
    return xyzr


def _duplicates(n, seed):
    """Few distinct centres, many spheres each: runs of equal codes far longer than a leaf."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-10, 10, (7, 3)).astype(np.float32)
    centres[:, 2] -= 30
    xyzr = np.empty((n, 4), dtype=np.float32)
    xyzr[:, :3] = centres[rng.integers(0, 7, n)]
    xyzr[:, 3] = rng.uniform(0.2, 1.5, n)
    return xyzr


def _one_point(n, seed):
    rng = np.random.default_rng(seed)
    xyzr = np.zeros((n, 4), dtype=np.float32)
    xyzr[:, 2] = -20
    xyzr[:, 3] = rng.uniform(0.5, 3.0, n)
    return xyzr


def _non_finite(n, seed):
    xyzr = _random(n, seed)
    xyzr[5, 0] = np.inf
    xyzr[17, 1] = np.nan
    xyzr[40, 3] = np.nan
    xyzr[41, 3] = np.inf
    xyzr[77, 3] = -0.4       # negative radius: |r| bounds it, r*r tests it
    xyzr[90, :3] = (-0.0, 0.0, -0.0)
    return xyzr


def _grid(n, seed):
    """Lattice: many exact coordinate ties and zero signs."""
    k = int(round(n ** (1 / 3))) + 1
    g = np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3)[:n].astype(np.float32)
    xyzr = np.empty((n, 4), dtype=np.float32)
    xyzr[:, 0] = g[:, 0] - k // 2
    xyzr[:, 1] = g[:, 1] - k // 2
    xyzr[:, 2] = -g[:, 2] * 2 - 5
    xyzr[:, 3] = 0.3
    return xyzr


SCENES = {
    "random": _random, "duplicates": _duplicates, "one_point": _one_point, "non_finite": _non_finite, "grid": _grid,
}


def _build(tracer, scene, cam, w, h, build):
    from path_tracer_amd import _capi
    tracer.set_accel(_capi.IR_ACCEL_LBVH)
    tracer.set_accel_build(build)
    try:
        idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
        info = tracer.get_accel_info()
        binfo = tracer.get_accel_build_info()
        arrays = tracer.download_accel() + tracer.download_accel_climb()
    finally:
        tracer.set_accel_build(_capi.IR_BUILD_AUTO)
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    return idx, t, info, binfo, arrays


@pytest.mark.parametrize("kind,n", [("random", 3), ("random", 5), ("random", 1000), ("random", 10000), ("duplicates", 5000),
                                    ("one_point", 257), ("non_finite", 2000), ("grid", 4096)])
def test_surface_area_build_gives_the_same_hits_and_radiance(pta, tracer, oracle, kind, n):
    """IR_BUILD_HOST_SAH: another tree over the same spheres (lbvh.h LbvhSplit::Sah); primary (index, t) and the radiance of a
    few samples per pixel are bit for bit those of the Morton tree — the result never depends on the tree."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 96, 54
    xyzr = SCENES[kind](n, 4321 + n)
    scene = _scene(xyzr)
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=6, max_compulsory_bounces=2, seed=W.RNG_SEED)
    out = {}
    for build in (_capi.IR_BUILD_HOST, _capi.IR_BUILD_HOST_SAH):
        idx, t, info, binfo, arrays = _build(tracer, scene, cam, w, h, build)
        assert binfo["builder"] == build
        assert np.array_equal(np.sort(arrays[2]), np.arange(n, dtype=np.int32))
        tracer.set_accel(_capi.IR_ACCEL_LBVH)
        tracer.set_accel_build(build)
        try:
            tracer.render(scene, w, h, False, spp=6)
            rad = tracer.get_radiance()[0].copy()
        finally:
            tracer.set_accel_build(_capi.IR_BUILD_AUTO)
            tracer.set_accel(_capi.IR_ACCEL_AUTO)
        out[build] = (idx, t, rad, info)
    a, b = out[_capi.IR_BUILD_HOST], out[_capi.IR_BUILD_HOST_SAH]
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
    assert a[2].tobytes() == b[2].tobytes()
    if n >= 1000 and kind == "random":
        assert a[3] != b[3]  # really another tree


@pytest.mark.parametrize("kind,n", [
    ("random", 3), ("random", 4), ("random", 5), ("random", 64), ("random", 1000), ("random", 10000), ("random", 200000), ("random", 2100000),
    ("duplicates", 300), ("duplicates", 5000), ("one_point", 3), ("one_point", 257), ("non_finite", 2000), ("grid", 4096),
])
def test_device_build_equals_host_build(pta, tracer, oracle, kind, n):
    from path_tracer_amd import _capi, workloads as W
    w, h = 96, 54
    xyzr = SCENES[kind](n, 1234 + n)
    scene = _scene(xyzr)
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    hidx, ht, hinfo, hb, harr = _build(tracer, scene, cam, w, h, _capi.IR_BUILD_HOST)
    didx, dt, dinfo, db, darr = _build(tracer, scene, cam, w, h, _capi.IR_BUILD_DEVICE)
    assert hb["builder"] == _capi.IR_BUILD_HOST and db["builder"] == _capi.IR_BUILD_DEVICE
    assert hinfo == dinfo, (hinfo, dinfo)
    assert hinfo["nodes"] >= 1 and hinfo["depth"] >= 1
    for name, a, b in zip(("wide", "leaf_spheres", "leaf_index", "climb_records", "sphere_start"), harr, darr):
        assert a.shape == b.shape, name
        assert a.tobytes() == b.tobytes(), f"{name}: {np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))} words differ"
    assert np.array_equal(np.sort(harr[2]), np.arange(n, dtype=np.int32))
    # every sphere starts at a leaf that holds it, and its climb record (if the tree has more than the one leaf) exists
    start = harr[4]
    leaf = -(start[:, 0] + 1)
    first, count = leaf >> 3, leaf & 7
    pos = np.empty(n, dtype=np.int64)
    pos[harr[2]] = np.arange(n)
    assert ((pos >= first) & (pos < first + count)).all()
    has_record = start[:, 1] != 0x7fffffff
    assert (start[has_record, 1] == 2 * hinfo["nodes"] + first[has_record]).all()
    assert hidx.tobytes() == didx.tobytes() and ht.tobytes() == dt.tobytes()
    if n <= 10000:
        oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
        assert didx.tobytes() == oidx.tobytes() and dt.tobytes() == ot.tobytes()


def test_tree_too_deep_falls_back_to_host(pta, tracer, oracle):
    """A Morton chain deeper than the traversal stack holds: the device build reports it and the host's median-split tree is used, as with IR_BUILD_HOST."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 64, 36
    n = 400
    xyzr = np.zeros((n, 4), dtype=np.float32)
    # codes 1 << b for every b < 30 (one quantised coordinate 2^j on one axis) split off one per level; the run of equal
    # codes at the origin adds log2(run) levels on the position bits
    k = 0
    for j in range(10):
        for axis in range(3):
            xyzr[k, axis] = 2.0 ** j / 1024.0
            k += 1
    xyzr[k, :3] = 1.0
    xyzr[:, 3] = 0.3
    scene = _scene(xyzr)
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    hidx, ht, hinfo, hb, harr = _build(tracer, scene, cam, w, h, _capi.IR_BUILD_HOST)
    didx, dt, dinfo, db, darr = _build(tracer, scene, cam, w, h, _capi.IR_BUILD_DEVICE)
    assert hinfo == dinfo and hinfo["depth"] + 1 <= 32
    for a, b in zip(harr, darr):
        assert a.tobytes() == b.tobytes()
    oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
    assert didx.tobytes() == oidx.tobytes() and dt.tobytes() == ot.tobytes()


def test_rebuild_after_edit_and_auto_choice(pta, tracer, oracle):
    """The editor's loop: move spheres, render again.  AUTO picks the device builder for large scenes, the host builder
    for small ones, and each edit gives the scan's hits."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 96, 54
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    xyzr = _random(6000, 7)
    for step in range(3):
        xyzr[:, 0] += 0.37 * step
        scene = _scene(xyzr.copy())
        idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
        assert tracer.get_accel_build_info()["builder"] == _capi.IR_BUILD_DEVICE
        oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
        assert idx.tobytes() == oidx.tobytes() and t.tobytes() == ot.tobytes()
    small = _scene(_random(500, 9))
    tracer.get_primary_hits(small, w, h)
    assert tracer.get_accel_build_info()["builder"] == _capi.IR_BUILD_HOST


def test_refit_keeps_tree_and_follows_spheres(pta, tracer, oracle):
    """IR_BUILD_REFIT: full device build first, then only the boxes follow the edits; hits stay the scan's even after
    moves the tree was not built for; a changed sphere count rebuilds."""
    from path_tracer_amd import _capi, workloads as W
    w, h = 96, 54
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    rng = np.random.default_rng(5)
    xyzr = _random(6000, 21)
    tracer.set_accel(_capi.IR_ACCEL_LBVH)
    tracer.set_accel_build(_capi.IR_BUILD_REFIT)
    try:
        tracer.get_primary_hits(_scene(xyzr), w, h)
        assert tracer.get_accel_build_info()["builder"] == _capi.IR_BUILD_DEVICE
        wide0, leaf0, index0 = tracer.download_accel()
        for step in range(3):
            xyzr = xyzr.copy()
            xyzr[:, :3] += rng.normal(0, 0.05 * (step + 1), (6000, 3)).astype(np.float32)
            xyzr[rng.integers(0, 6000, 20), :3] += rng.normal(0, 15.0, (20, 3)).astype(np.float32)  # a few stray far
            xyzr[rng.integers(0, 6000, 5), 3] *= 4.0
            scene = _scene(xyzr)
            idx, t = tracer.get_primary_hits(scene, w, h, with_t=True)
            assert tracer.get_accel_build_info()["builder"] == _capi.IR_BUILD_REFIT
            wide, leaf, index = tracer.download_accel()
            assert index.tobytes() == index0.tobytes()                      # same leaf order ...
            assert np.array_equal(wide.view(np.int32)[:, [3, 7]], wide0.view(np.int32)[:, [3, 7]])  # ... same tree ...
            assert np.array_equal(leaf[:, :3], xyzr[index, :3])             # ... new spheres
            oidx, ot = oracle.primary_hits(oracle.camera_from(cam, w, h), scene)
            assert idx.tobytes() == oidx.tobytes() and t.tobytes() == ot.tobytes()
        # radiance through the refitted tree (bounce rays)
        tracer.set_integrator(bounces=4, max_compulsory_bounces=2, seed=7)
        tracer.render(scene, w, h, False, spp=2)
        s, c = tracer.get_radiance()
        os_, oc = oracle.render(oracle.camera_from(cam, w, h), scene, 2, bounces=4, rr_start=2, seed=7, math_mode=1)[:2]
        assert np.array_equal(s.view(np.uint32), os_.view(np.uint32))
        # another sphere count: nothing to refit
        tracer.get_primary_hits(_scene(_random(6001, 22)), w, h)
        assert tracer.get_accel_build_info()["builder"] == _capi.IR_BUILD_DEVICE
    finally:
        tracer.set_accel_build(_capi.IR_BUILD_AUTO)
        tracer.set_accel(_capi.IR_ACCEL_AUTO)


def test_radiance_with_device_built_tree(pta, tracer, oracle):
    """Bounce rays through the device-built tree: bit-exact sums against the oracle's scan (C3 scene)."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 96, 54, 4
    scene = W.scene_s2()
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=0x0123456789ABCDEF)
    tracer.set_accel(_capi.IR_ACCEL_LBVH)
    tracer.set_accel_build(_capi.IR_BUILD_DEVICE)
    try:
        tracer.render(scene, w, h, False, spp=spp)
        s, c = tracer.get_radiance()
        assert tracer.get_accel_build_info()["builder"] == _capi.IR_BUILD_DEVICE
    finally:
        tracer.set_accel_build(_capi.IR_BUILD_AUTO)
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    os_, oc, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=8, rr_start=2, seed=0x0123456789ABCDEF, math_mode=1)[:3]
    assert c.tobytes() == oc.tobytes()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32))
