"""Host-side mirror of the reference's public surface (no GPU needed): names, argument meaning, flattening."""
import numpy as np
import pytest

import path_tracer_amd as pta
from path_tracer_amd import workloads as W


def test_public_surface_matches_reference_names():
    # core-engine/src/lib.rs:8-16, renderer.rs:29-173, scene.rs:5-93, pinhole_camera.rs:25-130, exr.rs:12-49
    for name in ("new", ):
        pass
    for m in ("set_active_camera", "get_current_size", "prepare_pixels", "render_updated", "render", "get_output",
              "get_last_render_time"):
        assert callable(getattr(pta.RayTracer, m))
    for m in ("get_ray", "set_position", "set_rotation", "set_image_resolutions", "compute_transformation_matrix",
              "on_update"):
        assert callable(getattr(pta.Camera, m)) and callable(getattr(pta.PinholeCamera, m))
    for m in ("set_focal_length", "set_sensor_size", "get_camera_to_world_matrix"):
        assert callable(getattr(pta.PinholeCamera, m))
    assert callable(pta.Scene.get_example_scene) and callable(pta.ExrImage.load_exr_image) and callable(pta.ExrImage.sample)
    m = pta.Matrial()
    assert (m.albedo, m.roughness, m.metalic, m.emission_color, m.emissive_power) == ([1, 1, 1], 0.5, 0.0, [0, 0, 0], 0.0)


def test_camera_setters_recompute_cached_state():
    cam = pta.PinholeCamera((0, 0, 2), (0, 0, 0), 35.0, 55.0, [1920, 1080])
    assert cam.aspect_ratio == pytest.approx(1920 / 1080)
    assert np.allclose(cam.forward, [0, 0, -1]) and np.allclose(cam.up, [0, 1, 0]) and np.allclose(cam.right, [1, 0, 0])
    f0 = cam.fov
    cam.set_focal_length(70.0)
    assert cam.fov < f0 and cam.fov == pytest.approx(2 * np.arctan(55 / 140), abs=1e-6)
    cam.set_sensor_size(35.0)
    assert cam.fov == pytest.approx(2 * np.arctan(35 / 140), abs=1e-6)
    cam.set_position((1, 2, 3))
    assert np.allclose(cam.get_camera_to_world_matrix()[:3, 3], [1, 2, 3])
    cam.set_rotation((0, np.pi / 2, 0))
    assert np.allclose(cam.forward, [-1, 0, 0], atol=1e-6)
    cam.set_image_resolutions([100, 50])
    assert cam.aspect_ratio == 2.0 and cam.image_size == [100, 50]
    r = cam.get_ray(50, 25)
    assert np.allclose(r.origin, [1, 2, 3])


def test_scene_flattening():
    sc = W.scene_s1(skybox=None)
    xyzr, mid = sc.flat_spheres()
    assert xyzr.shape == (3, 4) and xyzr.dtype == np.float32 and list(mid) == [0, 1, 2]
    assert np.allclose(xyzr[1], [0, -100.5, 0, 100])
    m = sc.flat_materials()
    assert m.shape == (3, 9) and np.allclose(m[2], [1, 1, 1, 0.5, 0, 1.0, 0.8, 0.6, 1.0])
    s2 = W.scene_s2()
    assert s2.flat_spheres()[0].shape == (10000, 4) and s2.flat_materials().shape == (64, 9)
    assert (s2.flat_spheres()[1] >= 0).all() and (s2.flat_spheres()[1] < 64).all()
    # deterministic generators
    assert W.scene_s2().flat_spheres()[0].tobytes() == s2.flat_spheres()[0].tobytes()
    a, b = W.synthetic_sky(64, 32), W.synthetic_sky(64, 32)
    assert a.pixels_buffer.tobytes() == b.pixels_buffer.tobytes() and a.pixels_buffer.max() > 10.0


def test_exr_image_validation():
    with pytest.raises(ValueError):
        pta.ExrImage(np.zeros((5, 3)), 2, 2)
    with pytest.raises(pta.InsplorayError):
        pta.ExrImage().sample((0, 1, 0))
