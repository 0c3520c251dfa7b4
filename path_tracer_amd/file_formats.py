"""insploray::file_formats — mirror of reference core-engine/src/file_formats/exr.rs."""
import ctypes as C

import numpy as np

from . import _capi


class ExrImage:
    """exr.rs:6-10: pixels_buffer (height*width, 3) f32 row-major, row 0 = top of the file."""

    def __init__(self, pixels_buffer=None, width=0, height=0):
        if pixels_buffer is None:
            pixels_buffer = np.zeros((0, 3), dtype=np.float32)
        self.pixels_buffer = np.ascontiguousarray(pixels_buffer, dtype=np.float32).reshape(-1, 3)
        self.width = int(width)
        self.height = int(height)
        if self.pixels_buffer.shape[0] != self.width * self.height:
            raise ValueError("pixels_buffer does not hold width*height texels")

    def sample(self, direction):
        """exr.rs:13-24: equirectangular nearest-texel lookup."""
        d = np.ascontiguousarray(np.asarray(direction, dtype=np.float32).reshape(3))
        out = np.zeros(3, dtype=np.float32)
        rc = _capi.lib.ir_exr_sample(_capi.fptr(self.pixels_buffer), self.width, self.height, _capi.fptr(d),
                                     _capi.fptr(out))
        if rc != _capi.IR_OK:
            raise _capi.InsplorayError(rc, "ExrImage.sample on an empty image")
        return out

    @staticmethod
    def load_exr_image(path):
        """exr.rs:26-49.  Raises InsplorayError where the reference returns Err."""
        p = _capi.c_float_p()
        w, h = C.c_uint32(), C.c_uint32()
        rc = _capi.lib.ir_exr_load(str(path).encode(), C.byref(p), C.byref(w), C.byref(h))
        if rc != _capi.IR_OK:
            raise _capi.InsplorayError(rc, _capi.lib.ir_exr_last_error().decode("utf-8", "replace"))
        try:
            n = w.value * h.value
            buf = np.ctypeslib.as_array(p, shape=(n, 3)).copy()
        finally:
            _capi.lib.ir_free(p)
        return ExrImage(buf, w.value, h.value)

    def save(self, path, compression=3, pixel_type=2):
        """Extension ("save image" is a reference TODO): write an RGB scanline EXR."""
        rc = _capi.lib.ir_exr_save(str(path).encode(), _capi.fptr(self.pixels_buffer), self.width, self.height,
                                   int(compression), int(pixel_type))
        if rc != _capi.IR_OK:
            raise _capi.InsplorayError(rc, _capi.lib.ir_exr_last_error().decode("utf-8", "replace"))


def save_png(path, argb, width, height):
    """Extension ("save image" is a reference TODO): the resolved 0xAARRGGBB frame of RayTracer.get_output (row 0 =
    bottom row) as an 8-bit RGB PNG, written top-down."""
    a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(-1)
    if a.size != width * height:
        raise ValueError(f"frame has {a.size} pixels, expected {width}x{height}")
    rc = _capi.lib.ir_png_save(str(path).encode(), a.ctypes.data_as(_capi.c_u32_p), int(width), int(height))
    if rc != _capi.IR_OK:
        raise _capi.InsplorayError(rc, _capi.lib.ir_exr_last_error().decode("utf-8", "replace"))
