"""The reference's own two unit tests (core-engine/src/cameras/pinhole_camera.rs:138-160 and :162-239), run against
the oracle's camera restatement AND against the product's host camera code (ir_pinhole_compute): these are the only
known answers the reference holds, and they pin Mat4::from_rotation_{x,y,z}, from_translation, the product order
T * Rz * Ry * Rx and transform_point3."""
import math

import numpy as np
import pytest

F32_EPS = 1.1920929e-07

BLENDER_POS = (-2.4027, -2.5716, 3.5259)
BLENDER_ROT = (0.1975, -0.7941, -1.9074)
BLENDER_MATRIX = np.array([[-0.2315, 0.9717, 0.0458, -2.4027],
                           [-0.6616, -0.1918, 0.7249, -2.5716],
                           [0.7132, 0.1375, 0.6873, 3.5259],
                           [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
LOCAL = [(-1.2510, 0.5574, 0.6953), (-0.3731, -1.2838, 0.5934), (0.0941, 1.1836, -0.7080),
         (1.1367, -0.5054, -0.4624), (0.2396, 0.5597, 1.4485)]
GLOBAL = [(-1.5396, -1.3468, 3.1881), (-3.5366, -1.6484, 3.4910), (-1.3068, -3.3742, 3.2692),
          (-3.1782, -3.5619, 3.9493), (-1.8480, -1.7874, 4.7693)]


def _oracle_cam(oracle, pos, rot, focal, sensor):
    return oracle.OracleCamera(pos, rot, focal, sensor, 1920, 1080)


def _product_cam(pos, rot, focal, sensor):
    import path_tracer_amd as pta
    return pta.PinholeCamera(pos, rot, focal, sensor, [1920, 1080])


def test_fov_calculation(oracle):
    # pinhole_camera.rs:138-160
    expected = np.float32(2.0) * np.arctan(np.float32(55.0) / (np.float32(2.0) * np.float32(35.0)), dtype=np.float32)
    assert abs(_oracle_cam(oracle, (0, 0, 0), (0, 0, 0), 35.0, 55.0).fov - float(expected)) < F32_EPS
    assert abs(_product_cam((0, 0, 0), (0, 0, 0), 35.0, 55.0).fov - float(expected)) < F32_EPS
    assert abs(float(expected) - 2.0 * math.atan(55.0 / 70.0)) < 1e-6


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_transformation_matrix_validation_for_camera(oracle, which):
    # pinhole_camera.rs:162-239 (Blender known answer), tolerance 1e-4 as in the reference
    if which == "oracle":
        cam = _oracle_cam(oracle, BLENDER_POS, BLENDER_ROT, 55.0, 35.0)
        m = cam.matrix
        tp = cam.transform_point3
    else:
        cam = _product_cam(BLENDER_POS, BLENDER_ROT, 55.0, 35.0)
        m = cam.get_camera_to_world_matrix()
        def tp(p):
            return (m.astype(np.float64) @ np.array([*p, 1.0]))[:3]
    assert np.abs(m - BLENDER_MATRIX).max() < 1e-4
    for loc, glo in zip(LOCAL, GLOBAL):
        exp = (BLENDER_MATRIX.astype(np.float64) @ np.array([*loc, 1.0]))[:3]
        got = np.asarray(tp(loc), dtype=np.float64)
        assert np.abs(got - exp).max() < 1e-4
        assert np.abs(got - np.array(glo)).max() < 2e-4  # Blender's printed 4-digit globals


def test_product_camera_state_is_bitwise_the_oracle(oracle):
    """host_math.cpp and oracle.cpp are separate restatements; they must agree to the bit."""
    import path_tracer_amd as pta
    rng = np.random.default_rng(7)
    for _ in range(50):
        pos = rng.normal(size=3) * 3
        rot = rng.uniform(-3.2, 3.2, size=3)
        focal, sensor = rng.uniform(10, 100), rng.uniform(10, 100)
        w, h = int(rng.integers(1, 4000)), int(rng.integers(1, 3000))
        p = pta.PinholeCamera(pos, rot, focal, sensor, [w, h])
        o = oracle.OracleCamera(p.position, p.rotation, p.focal_length, p.sensor_size, w, h)
        assert p.local_to_world.tobytes() == o.m.tobytes()
        assert np.float32(p.fov).tobytes() == np.float32(o.fov).tobytes()
        assert np.float32(p.aspect_ratio).tobytes() == np.float32(o.aspect).tobytes()
        assert np.concatenate([p.forward, p.up, p.right]).tobytes() == o.dirs.tobytes()
    # primary rays: product host get_ray == oracle get_ray
    p = pta.PinholeCamera((0.3, 1.0, 2.0), (-0.2, 0.4, 0.1), 35.0, 55.0, [33, 17])
    rays = oracle.camera_from(p, 33, 17).primary_rays()
    for (x, y) in [(0, 0), (32, 16), (5, 9), (16, 8)]:
        r = p.get_ray(x, y)
        assert np.concatenate([r.origin, r.direction]).tobytes() == rays[y, x].tobytes()
        assert abs(np.linalg.norm(r.direction) - 1.0) < 1e-6


def test_camera_directions_default_pose(oracle):
    cam = _oracle_cam(oracle, (0, 0, 2), (0, 0, 0), 35.0, 55.0)
    assert np.allclose(cam.dirs, [0, 0, -1, 0, 1, 0, 1, 0, 0])
