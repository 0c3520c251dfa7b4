"""ctypes wrapper around oracle/liboracle.so (the CPU restatement of the reference; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_LIB = os.path.join(_DIR, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


if not os.path.exists(_LIB):
    build()

lib = C.CDLL(_LIB)
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_u = C.POINTER(C.c_uint32)
_u64 = C.POINTER(C.c_uint64)


def _fp(a):
    return a.ctypes.data_as(_f) if a is not None else None


lib.orc_pinhole_compute.argtypes = [_f, _f, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _f, _f, _f, _f]
lib.orc_transform_point3.argtypes = [_f, _f, _f]
lib.orc_tan_half_fov.argtypes = [C.c_float]
lib.orc_tan_half_fov.restype = C.c_float
lib.orc_primary_rays.argtypes = [_f, _f, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _f]
lib.orc_primary_hits.argtypes = [_f, _f, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _f, C.c_uint32, _i, _f]
lib.orc_trace.argtypes = [_f, C.c_uint32, _f, _f, _f, _f, _f]
lib.orc_trace.restype = C.c_int32
lib.orc_sample_hemisphere.argtypes = [_f, C.c_float, C.c_float, C.c_int, _f]
lib.orc_sky_lookup.argtypes = [_f, C.c_uint32, C.c_uint32, _f, C.c_int, _f]
lib.orc_set_sky_filter.argtypes = [C.c_int]
lib.orc_set_sample_block.argtypes = [C.c_uint32]
lib.orc_set_sky_sampling.argtypes = [C.c_int]
lib.orc_sky_distribution.argtypes = [_f, C.c_uint32, C.c_uint32, _f, _f, _f]
lib.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
lib.orc_rng_draw.restype = C.c_float
lib.orc_rng_mix32.argtypes = [C.c_uint32]
lib.orc_rng_mix32.restype = C.c_uint32
lib.orc_poly_sincos.argtypes = [C.c_float, _f, _f]
lib.orc_poly_acos.argtypes = [C.c_float]
lib.orc_poly_acos.restype = C.c_float
lib.orc_poly_atan2.argtypes = [C.c_float, C.c_float]
lib.orc_poly_atan2.restype = C.c_float
lib.orc_resolve_argb.argtypes = [_f, _u, C.c_size_t, _u]
_SCENE_ARGS = [_f, _f, C.c_float, C.c_float, C.c_uint32, C.c_uint32,  # camera m16 pos fov aspect w h
               _f, _i, C.c_uint32, _f, C.c_uint32,                      # spheres, matid, n, mat9, m
               _f, _f, C.c_uint32, C.c_uint32,                          # sky color, skybox, sw, sh
               C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]    # bounces rr seed rng_mode math_mode
lib.orc_render.argtypes = _SCENE_ARGS + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _f, _u, _u64]
lib.orc_render.restype = C.c_double
lib.orc_pixel_samples.argtypes = _SCENE_ARGS + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _f]


class OracleCamera:
    """PinholeCamera state computed by the oracle (pinhole_camera.rs:59-69,110-129)."""

    def __init__(self, position, rotation, focal_length, sensor_size, width, height):
        self.position = np.ascontiguousarray(position, dtype=np.float32)
        self.rotation = np.ascontiguousarray(rotation, dtype=np.float32)
        self.m = np.zeros(16, dtype=np.float32)
        self.dirs = np.zeros(9, dtype=np.float32)
        fov, aspect = C.c_float(), C.c_float()
        lib.orc_pinhole_compute(_fp(self.position), _fp(self.rotation), focal_length, sensor_size, width, height,
                                _fp(self.m), C.byref(fov), C.byref(aspect), _fp(self.dirs))
        self.fov, self.aspect, self.w, self.h = fov.value, aspect.value, width, height

    @property
    def matrix(self):
        return self.m.reshape(4, 4).T.copy()

    def transform_point3(self, p):
        out = np.zeros(3, dtype=np.float32)
        lib.orc_transform_point3(_fp(self.m), _fp(np.ascontiguousarray(p, dtype=np.float32)), _fp(out))
        return out

    def primary_rays(self):
        out = np.zeros((self.h, self.w, 6), dtype=np.float32)
        lib.orc_primary_rays(_fp(self.m), _fp(self.position), self.fov, self.aspect, self.w, self.h, _fp(out))
        return out


def camera_from(cam, width, height):
    """OracleCamera with the same parameters as a path_tracer_amd.PinholeCamera"""
    return OracleCamera(cam.position, cam.rotation, cam.focal_length, cam.sensor_size, width, height)


def _scene_arrays(scene):
    xyzr, mid = scene.flat_spheres()
    xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
    mid = np.ascontiguousarray(mid, dtype=np.int32)
    mats = np.ascontiguousarray(scene.flat_materials(), dtype=np.float32)
    sky = np.ascontiguousarray(scene.default_sky_color, dtype=np.float32)
    box = scene.skybox
    boxbuf = np.ascontiguousarray(box.pixels_buffer, dtype=np.float32) if box is not None else None
    return xyzr, mid, mats, sky, boxbuf, (box.width if box else 0), (box.height if box else 0)


def _scene_args(cam, scene, bounces, rr_start, seed, rng_mode, math_mode, keep):
    xyzr, mid, mats, sky, boxbuf, sw, sh = _scene_arrays(scene)
    keep.extend([xyzr, mid, mats, sky, boxbuf])
    return [_fp(cam.m), _fp(cam.position), cam.fov, cam.aspect, cam.w, cam.h,
            _fp(xyzr), mid.ctypes.data_as(_i), xyzr.shape[0], _fp(mats), mats.shape[0],
            _fp(sky), _fp(boxbuf), sw, sh, bounces, rr_start, C.c_uint64(seed), rng_mode, math_mode]


def primary_hits(cam, scene):
    xyzr, _ = scene.flat_spheres()
    xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
    idx = np.zeros((cam.h, cam.w), dtype=np.int32)
    t = np.zeros((cam.h, cam.w), dtype=np.float32)
    lib.orc_primary_hits(_fp(cam.m), _fp(cam.position), cam.fov, cam.aspect, cam.w, cam.h, _fp(xyzr), xyzr.shape[0],
                         idx.ctypes.data_as(_i), _fp(t))
    return idx, t


def render(cam, scene, spp, bounces=5, rr_start=2, seed=0x0123456789ABCDEF, rng_mode=0, math_mode=0, sample_base=0,
           threads=8, tile_rank=0, tile_world=1, sums=None, counts=None, want_stats=False):
    """Returns (sums (h,w,4) f32, counts (h,w) u32, seconds, stats dict or None); accumulates into sums/counts."""
    if sums is None:
        sums = np.zeros((cam.h, cam.w, 4), dtype=np.float32)
    if counts is None:
        counts = np.zeros((cam.h, cam.w), dtype=np.uint32)
    stats = (C.c_uint64 * 5)() if want_stats else None
    keep = []
    args = _scene_args(cam, scene, bounces, rr_start, seed, rng_mode, math_mode, keep)
    secs = lib.orc_render(*args, spp, sample_base, threads, tile_rank, tile_world, _fp(sums),
                          counts.ctypes.data_as(_u), stats)
    st = dict(zip(("samples", "traces", "tests", "disc_pos", "hits"), [int(v) for v in stats])) if want_stats else None
    return sums, counts, secs, st


def pixel_samples(cam, scene, x, y, spp, bounces=5, rr_start=2, seed=0x0123456789ABCDEF, rng_mode=0, math_mode=0,
                  sample_base=0):
    out = np.zeros((spp, 3), dtype=np.float32)
    keep = []
    args = _scene_args(cam, scene, bounces, rr_start, seed, rng_mode, math_mode, keep)
    lib.orc_pixel_samples(*args, x, y, spp, sample_base, _fp(out))
    return out


def resolve_argb(sums, counts):
    s = np.ascontiguousarray(sums, dtype=np.float32).reshape(-1, 4)
    c = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1)
    out = np.zeros(c.shape[0], dtype=np.uint32)
    lib.orc_resolve_argb(_fp(s), c.ctypes.data_as(_u), c.shape[0], out.ctypes.data_as(_u))
    return out


def trace(xyzr, o, d):
    xyzr = np.ascontiguousarray(xyzr, dtype=np.float32).reshape(-1, 4)
    t = C.c_float()
    pos = np.zeros(3, dtype=np.float32)
    n = np.zeros(3, dtype=np.float32)
    idx = lib.orc_trace(_fp(xyzr), xyzr.shape[0], _fp(np.ascontiguousarray(o, dtype=np.float32)),
                        _fp(np.ascontiguousarray(d, dtype=np.float32)), C.byref(t), _fp(pos), _fp(n))
    return idx, t.value, pos, n


def sample_hemisphere(n, r1, r2, math_mode=0):
    out = np.zeros(3, dtype=np.float32)
    lib.orc_sample_hemisphere(_fp(np.ascontiguousarray(n, dtype=np.float32)), r1, r2, math_mode, _fp(out))
    return out


def sky_lookup(box, d, math_mode=0):
    out = np.zeros(3, dtype=np.float32)
    buf = np.ascontiguousarray(box.pixels_buffer, dtype=np.float32)
    lib.orc_sky_lookup(_fp(buf), box.width, box.height, _fp(np.ascontiguousarray(d, dtype=np.float32)), math_mode, _fp(out))
    return out


def set_sky_filter(f):
    """0 = nearest (the reference), 1 = bilinear (the build's extension); process-global."""
    lib.orc_set_sky_filter(int(f))


def set_sky_sampling(mode):
    """0 = BSDF sampling only (the reference), 1 = sky importance sampling with MIS (the build's extension); process-global."""
    lib.orc_set_sky_sampling(int(mode))


def sky_distribution(box):
    """(marginal (h+1,), cond (h, w+1), pdf (h, w)) of the sky importance-sampling tables."""
    w, h = box.width, box.height
    buf = np.ascontiguousarray(box.pixels_buffer, dtype=np.float32)
    m = np.zeros(h + 1, dtype=np.float32)
    c = np.zeros((h, w + 1), dtype=np.float32)
    p = np.zeros((h, w), dtype=np.float32)
    lib.orc_sky_distribution(_fp(buf), w, h, _fp(m), _fp(c), _fp(p))
    return m, c, p


def set_sample_block(n):
    """Samples per accumulation block of a multi-sample render (process-global; the build's default is 128)."""
    lib.orc_set_sample_block(int(n))


def rng_draw(seed, pixel, sample, k):
    return lib.orc_rng_draw(C.c_uint64(seed), pixel, sample, k)


def poly_sincos(x):
    s, c = C.c_float(), C.c_float()
    lib.orc_poly_sincos(x, C.byref(s), C.byref(c))
    return s.value, c.value
