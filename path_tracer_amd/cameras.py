"""insploray::cameras — mirror of reference core-engine/src/cameras/{mod.rs,pinhole_camera.rs}.

`Camera` is the plugin trait (cameras/mod.rs:7-15); `PinholeCamera` (pinhole_camera.rs:7-130) is the fast path
whose cached state goes to the device as a parameter block.  All f32 camera arithmetic is done by the C library
(ir_pinhole_compute / ir_pinhole_get_ray) in glam's operation order, not in numpy.
"""
import ctypes as C

import numpy as np

from . import _capi
from .ray import Ray


class Camera:
    """trait Camera (cameras/mod.rs:7-15).  Implementors other than PinholeCamera are evaluated on the host:
    the renderer calls get_ray for every pixel (or get_rays, if provided) and uploads the rays."""

    def get_ray(self, x, y):  # -> Ray
        raise NotImplementedError

    def set_position(self, position):
        raise NotImplementedError

    def set_rotation(self, rotation):
        raise NotImplementedError

    def set_image_resolutions(self, image_resolution):
        raise NotImplementedError

    def compute_transformation_matrix(self):
        raise NotImplementedError

    def on_update(self):
        raise NotImplementedError


def _f3(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))
    return a


class PinholeCamera(Camera):
    """struct PinholeCamera (pinhole_camera.rs:7-23); rotation is XYZ Euler in radians."""

    def __init__(self, position, rotation, focal_length, sensor_size, image_size):
        self.position = _f3(position)
        self.rotation = _f3(rotation)
        self.focal_length = float(focal_length)
        self.sensor_size = float(sensor_size)
        self.image_size = [0, 0]
        self.aspect_ratio = 0.0
        self.local_to_world = np.zeros(16, dtype=np.float32)  # column-major, glam Mat4
        self.forward = np.zeros(3, dtype=np.float32)
        self.up = np.zeros(3, dtype=np.float32)
        self.right = np.zeros(3, dtype=np.float32)
        self.fov = 0.0
        self.set_image_resolutions(image_size)
        self.on_update()

    # -- pinhole_camera.rs:48-61
    def set_focal_length(self, focal_length):
        self.focal_length = float(focal_length)
        self._recompute()

    def set_sensor_size(self, sensor_size):
        self.sensor_size = float(sensor_size)
        self._recompute()

    def get_camera_to_world_matrix(self):
        """4x4 numpy matrix M with M[row, col] (the column-major buffer is self.local_to_world)."""
        return self.local_to_world.reshape(4, 4).T.copy()

    # -- trait impl, pinhole_camera.rs:76-130
    def get_ray(self, x, y):
        out = np.zeros(6, dtype=np.float32)
        rc = _capi.lib.ir_pinhole_get_ray(_capi.fptr(self.local_to_world), _capi.fptr(self.position),
                                          C.c_float(self.fov), C.c_float(self.aspect_ratio),
                                          self.image_size[0], self.image_size[1], int(x), int(y), _capi.fptr(out))
        _capi.check(None, rc)
        return Ray(out[:3].copy(), out[3:].copy())

    def set_position(self, position):
        self.position = _f3(position)
        self.on_update()

    def set_rotation(self, rotation):
        self.rotation = _f3(rotation)
        self.on_update()

    def set_image_resolutions(self, image_resolution):
        self.image_size = [int(image_resolution[0]), int(image_resolution[1])]
        # pinhole_camera.rs:112 (f32 division; 0/0 = NaN for the reference's initial [0, 0] size)
        w, h = np.float32(self.image_size[0]), np.float32(self.image_size[1])
        with np.errstate(divide="ignore", invalid="ignore"):
            self.aspect_ratio = float(np.float32(w) / np.float32(h))

    def compute_transformation_matrix(self):
        self._recompute()

    def on_update(self):
        self._recompute()

    def _recompute(self):
        fov = C.c_float()
        aspect = C.c_float()
        dirs = np.zeros(9, dtype=np.float32)
        w, h = self.image_size
        rc = _capi.lib.ir_pinhole_compute(_capi.fptr(self.position), _capi.fptr(self.rotation),
                                          C.c_float(self.focal_length), C.c_float(self.sensor_size),
                                          w, h, _capi.fptr(self.local_to_world), C.byref(fov), C.byref(aspect),
                                          _capi.fptr(dirs))
        _capi.check(None, rc)
        self.fov = fov.value
        self.forward, self.up, self.right = dirs[0:3].copy(), dirs[3:6].copy(), dirs[6:9].copy()


# `SharedCamera = Arc<RwLock<dyn Camera + Send + Sync>>` (cameras/mod.rs:17): in Python a plain reference.
SharedCamera = Camera
