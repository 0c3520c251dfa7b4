"""Edge cases of the render path on the GPU against the oracle (bitwise): empty and degenerate inputs, the reference's
quirks (default material, near-root-only intersection, ties), odd frame shapes, rotated cameras."""
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


def _compare(pta, tracer, oracle, scene, cam, w, h, spp, bounces=5, rr=2, accel=None):
    from path_tracer_amd import _capi
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=bounces, max_compulsory_bounces=rr, seed=SEED)
    tracer.set_accel(accel if accel is not None else _capi.IR_ACCEL_AUTO)
    try:
        tracer.render(scene, w, h, False, spp=spp)
        s, c = tracer.get_radiance()
        idx, t = tracer.get_primary_hits(with_t=True)
    finally:
        tracer.set_accel(_capi.IR_ACCEL_AUTO)
    oc = oracle.camera_from(cam, w, h)
    os_, ocnt, _, _ = oracle.render(oc, scene, spp, bounces=bounces, rr_start=rr, seed=SEED, math_mode=1)
    oidx, ot = oracle.primary_hits(oc, scene)
    assert np.array_equal(c, ocnt)
    assert idx.tobytes() == oidx.tobytes() and t.tobytes() == ot.tobytes()
    assert np.array_equal(s.view(np.uint32), os_.view(np.uint32)), f"max abs diff {np.nanmax(np.abs(s - os_))}"
    return s, idx


def test_empty_scene_is_all_sky(pta, tracer, oracle):
    from path_tracer_amd import workloads as W
    scene = pta.Scene(default_sky_color=(0.25, 0.5, 0.75))
    s, idx = _compare(pta, tracer, oracle, scene, W.default_camera(70, 40), 70, 40, 3)
    assert (idx == -1).all() and np.allclose(s[..., :3], np.array([0.75, 1.5, 2.25], dtype=np.float32))
    scene.skybox = W.synthetic_sky(64, 32)
    _compare(pta, tracer, oracle, scene, W.default_camera(70, 40), 70, 40, 3)


@pytest.mark.parametrize("w,h", [(1, 1), (1, 37), (130, 1), (65, 17), (257, 129)])
def test_odd_frame_shapes(pta, tracer, oracle, w, h):
    from path_tracer_amd import workloads as W
    _compare(pta, tracer, oracle, W.scene_s1(), W.default_camera(w, h), w, h, 5, bounces=8)


def test_zero_sized_and_zero_sample_renders_are_no_ops(pta, tracer):
    from path_tracer_amd import workloads as W
    scene = W.scene_s1(skybox=None)
    tracer.set_active_camera(W.default_camera(0, 0))
    tracer.render(scene, 0, 0, False)                      # RayTracer::new(0, 0) then render(.., 0, 0, ..)
    assert tracer.get_current_size() == [0, 0] and tracer.get_output().size == 0
    tracer.set_active_camera(W.default_camera(32, 16))
    tracer.render(scene, 32, 16, False, spp=0)
    s, c = tracer.get_radiance()
    assert not s.any() and not c.any() and (tracer.get_output() == 0).all()   # unrendered pixel resolves to 0x00000000


def test_default_material_for_bad_ids(pta, tracer, oracle):
    """material_id < 0 or out of range => DEFAULT_MATERIAL, white Lambertian (integrator.rs:8-14,53-60)."""
    from path_tracer_amd import workloads as W
    scene = W.scene_s1(skybox=None)
    scene.spheres[0].material_id = -1
    scene.spheres[2].material_id = 99
    s, _ = _compare(pta, tracer, oracle, scene, W.default_camera(96, 54), 96, 54, 6, bounces=8)
    scene.materials = []                                    # no materials at all
    _compare(pta, tracer, oracle, scene, W.default_camera(96, 54), 96, 54, 6, bounces=8)


def test_camera_inside_a_sphere_sees_through_it(pta, tracer, oracle):
    """Only the near root is tested (integrator.rs:121): from inside, a sphere is invisible."""
    from path_tracer_amd import workloads as W
    scene = W.scene_s1(skybox=None)
    scene.spheres[0].radius = 3.0                           # camera at (0,0,2) is now inside sphere 0
    _, idx = _compare(pta, tracer, oracle, scene, W.default_camera(64, 36), 64, 36, 4)
    assert not (idx == 0).any()


@pytest.mark.parametrize("accel", ["linear", "lbvh"])
def test_coincident_spheres_keep_lowest_index(pta, tracer, oracle, accel):
    """Strict '<' in the scan (integrator.rs:121): among identical spheres the first one wins, with either structure."""
    from path_tracer_amd import _capi, workloads as W
    base = W.scene_s2(n=40)
    xyzr, mid = base.flat_spheres()
    xyzr = np.concatenate([xyzr, xyzr[5:25], xyzr[5:25]], 0)          # two more copies of 20 spheres
    mid = np.concatenate([mid, (mid[5:25] + 1) % 64, (mid[5:25] + 2) % 64], 0)   # with other materials: order matters
    scene = pta.FlatScene(xyzr, mid, base.flat_materials(), default_sky_color=(0.6, 0.7, 0.9))
    s, idx = _compare(pta, tracer, oracle, scene, W.camera_s2(160, 90), 160, 90, 4, bounces=8,
                      accel=_capi.IR_ACCEL_LINEAR if accel == "linear" else _capi.IR_ACCEL_LBVH)
    assert idx.max() < 40


def test_rotated_camera_pose(pta, tracer, oracle):
    """The Blender pose of the reference's own camera test (pinhole_camera.rs:162-239) through the whole path."""
    from path_tracer_amd import workloads as W
    cam = pta.PinholeCamera((-2.4027, -2.5716, 3.5259), (0.1975, -0.7941, -1.9074), 35.0, 55.0, [120, 80])
    scene = W.scene_s1()
    for sp in scene.spheres:
        sp.position = [sp.position[0], sp.position[2], sp.position[1]]   # any scene the pose looks at
    _compare(pta, tracer, oracle, scene, cam, 120, 80, 4, bounces=8)


@pytest.mark.parametrize("bounces,rr", [(0, 2), (1, 0), (2, 0), (16, 0), (16, 16)])
def test_depth_and_roulette_extremes(pta, tracer, oracle, bounces, rr):
    from path_tracer_amd import workloads as W
    s, _ = _compare(pta, tracer, oracle, W.scene_s1(), W.default_camera(80, 45), 80, 45, 6, bounces=bounces, rr=rr)
    if bounces == 0:
        assert not s[..., :3].any() and (s[..., 3] == 6).all()     # integrator.rs:47,92: (0,0,0,1) per sample


def test_lbvh_radiance_on_mid_sized_scene(pta, tracer, oracle):
    """A scene just above the AUTO crossover: the LBVH kernel's scheduling must not change any sample."""
    from path_tracer_amd import _capi, workloads as W
    scene = W.scene_s2(n=200)
    _compare(pta, tracer, oracle, scene, W.camera_s2(128, 72), 128, 72, 20, bounces=8)
    assert tracer.get_accel_info()["accel"] == _capi.IR_ACCEL_LBVH


def test_limits_are_errors_not_crashes(pta, tracer):
    from path_tracer_amd import _capi, workloads as W
    scene = W.scene_s1(skybox=None)
    tracer.set_active_camera(W.default_camera(70000, 70000))
    with pytest.raises(pta.InsplorayError) as e:
        tracer.render(scene, 70000, 70000, False)            # 4.9 G pixels: refused before anything is allocated
    assert e.value.code == _capi.IR_ELIMIT
    with pytest.raises(pta.InsplorayError):
        tracer.set_sample_block(4097)  # (0 is IR_SAMPLE_BLOCK_AUTO)
    with pytest.raises(pta.InsplorayError):
        tracer.set_schedule(7)
    with pytest.raises(pta.InsplorayError):
        tracer.set_sky_filter(5)
    with pytest.raises(pta.InsplorayError):
        tracer.set_accel_build(9)
    # a sky beyond what the kernels' 32-bit texel offsets reach (2^28 texels, 2^24 per side) is refused before it is read
    import ctypes as C
    tiny = (C.c_float * 3)(1.0, 1.0, 1.0)
    for (sw, sh) in ((1 << 24, 1), (1, 1 << 24), (1 << 14, 1 << 14)):
        assert _capi.lib.ir_set_skybox(tracer._ctx, tiny, sw, sh) == _capi.IR_ELIMIT
    # the context is still usable afterwards
    tracer.set_active_camera(W.default_camera(32, 18))
    tracer.render(scene, 32, 18, False, spp=2)
    assert (tracer.get_radiance()[1] == 2).all()


def test_refused_non_accumulating_render_leaves_a_zeroed_film(pta):
    """A non-accumulating render that is refused after the reset (no camera set) must not leave old sums under
    samples_done == 0: the fills that the task schedule normally folds into its primary pass are issued on the error path."""
    import ctypes as C
    from path_tracer_amd import _capi, workloads as W
    w, h = 64, 40
    rt = pta.RayTracer(0, 0, device=0)
    try:
        ctx = rt._ctx
        # scene through the mirror, no camera at all: a frame is sized and rendered by hand through the C ABI
        rt._upload_scene(W.scene_s1(skybox=None))
        assert _capi.lib.ir_render(ctx, w, h, 0, 0) == _capi.IR_OK          # sizes the film
        ones = np.ones((h, w, 4), dtype=np.float32)
        rt.set_radiance(ones, np.full((h, w), 3, dtype=np.uint32))
        rt.set_samples_accumulated(3)
        rc = _capi.lib.ir_render(ctx, w, h, 4, 0)
        assert rc == _capi.IR_ESTATE, _capi.lib.ir_last_error(ctx)
        s, c = rt.get_radiance()
        assert rt.samples_accumulated() == 0 and not s.any() and not c.any()
        # an accumulating render that is refused changes nothing
        rt.set_radiance(ones, np.full((h, w), 3, dtype=np.uint32))
        rt.set_samples_accumulated(3)
        assert _capi.lib.ir_render(ctx, w, h, 4, 1) == _capi.IR_ESTATE
        s, c = rt.get_radiance()
        assert rt.samples_accumulated() == 3 and (s == 1.0).all() and (c == 3).all()
    finally:
        rt.close()


def test_mapped_output_equals_copied_output(pta, tracer):
    """ir_map_output_argb (the reference's borrowed `&[u32]`, renderer.rs:164-169) against ir_get_output_argb's copy."""
    import ctypes as C
    from path_tracer_amd import _capi, workloads as W
    for (w, h) in ((97, 61), (320, 180), (97, 61)):
        scene = W.scene_s1()
        tracer.set_active_camera(W.default_camera(w, h))
        tracer.render(scene, w, h, False, spp=3)
        mapped = tracer.get_output()
        copied = np.zeros(w * h, dtype=np.uint32)
        _capi.check(tracer._ctx, _capi.lib.ir_get_output_argb(tracer._ctx, copied.ctypes.data_as(_capi.c_u32_p), w * h))
        assert mapped.shape == (w * h,) and mapped.tobytes() == copied.tobytes()
        assert (mapped >> 24 == 0xBA).all()  # alpha of a rendered pixel: Reinhard(1) ** (1/2.2) * 255 truncated
    with pytest.raises(pta.InsplorayError):
        _capi.check(tracer._ctx, _capi.lib.ir_map_output_argb(tracer._ctx, None, None))


def test_partial_scene_edits_keep_only_what_they_may(pta, tracer, oracle):
    """ir_set_spheres / ir_set_materials keep the device tables of what did not change (moving spheres keeps the shading
    table, recolouring keeps the LBVH): every step of an edit sequence still renders the oracle's frame."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 96, 54, 3
    base = W.scene_s2(n=400, skybox=None)
    cam = W.camera_s2(w, h)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=6, max_compulsory_bounces=2, seed=SEED)
    xyzr, mid, mats = base.xyzr.copy(), base.material_id.copy(), base.materials9.copy()
    steps = []
    steps.append((xyzr.copy(), mid.copy(), mats.copy()))                       # the scene
    xyzr[1:, 0] += np.float32(0.3)
    steps.append((xyzr.copy(), mid.copy(), mats.copy()))                       # geometry only
    mid = ((mid + 7) % mats.shape[0]).astype(np.int32)
    steps.append((xyzr.copy(), mid.copy(), mats.copy()))                       # material ids only
    mats[:, :3] = mats[::-1, :3]
    steps.append((xyzr.copy(), mid.copy(), mats.copy()))                       # materials only
    steps.append((xyzr.copy(), mid.copy(), mats.copy()))                       # nothing
    steps.append((xyzr[:300].copy(), mid[:300].copy(), mats.copy()))           # fewer spheres
    frames = []
    for accel in (_capi.IR_ACCEL_LBVH, _capi.IR_ACCEL_LINEAR):
        tracer.set_accel(accel)
        try:
            for (x, m, t) in steps:
                scene = W.FlatScene(x, m, t, default_sky_color=(0.6, 0.7, 0.9), skybox=None)
                tracer.render(scene, w, h, False, spp=spp)
                s, c = tracer.get_radiance()
                os_, oc, _, _ = oracle.render(oracle.camera_from(cam, w, h), scene, spp, bounces=6, rr_start=2, seed=SEED, math_mode=1)
                assert np.array_equal(s.view(np.uint32), os_.view(np.uint32))
                frames.append(s.copy())
        finally:
            tracer.set_accel(_capi.IR_ACCEL_AUTO)
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[1], frames[2]) and not np.array_equal(frames[2], frames[3])
    assert np.array_equal(frames[3], frames[4])


def test_sky_edited_in_place_is_uploaded_again(pta, tracer, oracle):
    """The reference reads the Scene afresh on every render (renderer.rs:122): a sky whose texels are edited in place
    between two renders must show in the second one (the upload used to be keyed on id() alone)."""
    from path_tracer_amd import workloads as W
    w, h, spp = 64, 36, 4
    scene = W.scene_s1()
    cam = W.default_camera(w, h)
    tracer.set_active_camera(cam)
    tracer.set_integrator(bounces=5, max_compulsory_bounces=2, seed=7)
    tracer.render(scene, w, h, False, spp=spp)
    before, _ = tracer.get_radiance()
    before = before.copy()
    scene.skybox.pixels_buffer *= np.float32(0.25)  # same object, same buffer, new contents
    tracer.render(scene, w, h, False, spp=spp)
    after, _ = tracer.get_radiance()
    ocam = oracle.camera_from(cam, w, h)
    want, _, _, _ = oracle.render(ocam, scene, spp, bounces=5, rr_start=2, seed=7, math_mode=1)
    assert np.array_equal(after.view(np.uint32), want.view(np.uint32))
    assert not np.array_equal(after, before)


def test_launch_timers_and_lane_profile(pta, tracer):
    """ir_last_kernel_ms / ir_last_paths_kernel_ms / ir_get_lane_profile: what bench.py's roofline object is built from.
    The path kernel's time lies inside the launch sequence's; the lane profile counts every path sample once at the loop
    top per bounce, never more than 64 lanes per wave execution, and is only there after a statistics render."""
    from path_tracer_amd import workloads as W
    w, h, spp = 320, 180, 32
    scene = W.scene_s1()
    tracer.set_active_camera(W.default_camera(w, h))
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=SEED)
    tracer.render(scene, w, h, False, spp=spp)
    seq_ms, launches = tracer.last_kernel_ms()
    paths_ms = tracer.last_paths_kernel_ms()
    assert launches == 1 and 0.0 < paths_ms <= seq_ms < 1000.0
    # renders enqueued without waiting: every launch's path kernel can still be read afterwards, oldest first
    before = len(tracer.paths_kernel_history())
    for _ in range(5):
        tracer.render(scene, w, h, True, spp=spp, blocking=False)
    tracer.synchronize()
    hist = tracer.paths_kernel_history()
    assert len(hist) == min(64, before + 5) and all(0.0 < t < 1000.0 for t in hist[-5:])
    assert hist[-1] == tracer.last_paths_kernel_ms()
    assert tracer.paths_kernel_history(3) == hist[-3:]
    for _ in range(70):                                # more launches than the ring holds
        tracer.render(scene, 64, 36, True, spp=1, blocking=False)
    tracer.synchronize()
    hist = tracer.paths_kernel_history()
    assert len(hist) == 64 and all(0.0 < t < 1000.0 for t in hist)
    with pytest.raises(pta.InsplorayError):
        tracer.get_lane_profile()  # no statistics render yet
    tracer.enable_stats()
    tracer.render(scene, w, h, False, spp=spp)
    st = tracer.get_stats()
    prof = tracer.get_lane_profile()
    waves, lanes = prof["loop"]
    assert waves > 0 and lanes <= 64 * waves
    # a lane passes the loop top once per hit it shades (the primary hit of every sample of a path pixel included)
    assert lanes == st["hits"]
    for k in ("roulette", "hit", "sky"):
        assert prof[k][0] <= waves and prof[k][1] <= lanes


_SINGLE_BLOCK_FRAMES = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import path_tracer_amd as pta
from path_tracer_amd import _capi, workloads as W
rt = pta.RayTracer(0, 0, device=0)
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
for sky, mode in ((None, _capi.IR_SKY_SAMPLING_BSDF), ("synthetic", _capi.IR_SKY_SAMPLING_BSDF), ("synthetic", _capi.IR_SKY_SAMPLING_MIS)):
    scene = W.scene_s1(skybox=sky)
    rt.set_sky_sampling(mode)
    for (w, h) in ((203, 117), (64, 64)):
        rt.set_active_camera(W.default_camera(w, h))
        rt.render(scene, w, h, False, spp=1)       # a frame that starts with the launch
        rt.render(scene, w, h, True, spp=1)        # prepare_pixels: one more sample
        rt.enable_stats()                          # (applies to the next render)
        rt.render(scene, w, h, True, spp=37)       # a longer single block, the statistics instantiation
        s, c = rt.get_radiance()
        st = rt.get_stats()
        print(hashlib.sha256(s.tobytes() + c.tobytes()).hexdigest(), sorted(st.items()))
rt.close()
"""


def test_single_block_launch_same_bits_with_and_without_task_list(pta):
    """A launch of one sample block per pixel runs a lane per pixel (paths_kernel DIRECT); with IR_DEBUG_DIRECT=0 the same
    launch goes through the primary pass, the list and the task dispenser.  Same sums, same counts, same statistics."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for direct in ("1", "0"):
        env = dict(os.environ, IR_DEBUG_DIRECT=direct)
        r = subprocess.run([sys.executable, "-c", _SINGLE_BLOCK_FRAMES, root], env=env, capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == 6, outs
