"""Synthetic scenes and skies of BASELINE.json's configs, generated deterministically from integer
hashes so that every box builds bit-identical inputs without any asset file (the reference's assets/env/*.exr are
absent from the checkout, .MISSING_LARGE_BLOBS).

S1 "default+emitter": the reference example scene (scene.rs:62-90) plus one emissive sphere as the frontend editor
could add it (viewport.rs:44-77); camera = Viewport::default() (viewport.rs:169-178).
S2 "10k": 10 000 random Lambertian/emissive spheres on the ground sphere.
"""
import numpy as np

from .cameras import PinholeCamera
from .file_formats import ExrImage
from .scene import FlatScene, Matrial, Scene, Sphere

SKY_SEED = 0x1A5B10
S2_SEED = 0x5EED0001
RNG_SEED = 0x0123456789ABCDEF


def _mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x21F0AAAD)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x735A2D97)
    x ^= x >> np.uint32(15)
    return x


def _unit(seed, index, stream):
    """uniform [0,1) f64 from (seed, index, stream), 24 bits"""
    with np.errstate(over="ignore"):
        k = _mix32(np.uint32(seed) ^ (np.asarray(index, dtype=np.uint32) * np.uint32(0x9E3779B9)))
        k = _mix32(k + np.uint32(stream) * np.uint32(0x85EBCA6B))
    return (k >> np.uint32(8)).astype(np.float64) / float(1 << 24)


def synthetic_sky(width=1024, height=512, seed=SKY_SEED, sun_scale=40.0, sun_dir=(0.35, 0.55, -0.75)):
    """Equirectangular HDR sky: smooth vertical gradient + one bright sun lobe + low-amplitude hash noise.
    Row 0 is the zenith (+Y), as ExrImage::sample assumes (exr.rs:15-21).  Every texel is a function of its own (x, y),
    so the image is evaluated in bands of rows: an 8192 x 4096 sky (403 MB as Vec<Vec3>; 537 MB padded to
    float4 on the device, far beyond the 256 MB Infinity Cache) does not need several GB of f64 temporaries."""
    out = np.empty((height, width, 3), dtype=np.float32)
    xs = (np.arange(width, dtype=np.float64) + 0.5) / width
    phi = xs[None, :] * 2.0 * np.pi - np.pi          # u = (phi + pi) / (2 pi)
    horizon = np.array([0.85, 0.80, 0.75])
    zenith = np.array([0.25, 0.45, 0.95])
    ground = np.array([0.18, 0.16, 0.14])
    s = np.asarray(sun_dir, dtype=np.float64)
    s = s / np.linalg.norm(s)
    band = max(1, (1 << 19) // width)
    for y0 in range(0, height, band):
        y1 = min(height, y0 + band)
        ys = (np.arange(y0, y1, dtype=np.float64) + 0.5) / height
        theta = ys[:, None] * np.pi                  # v = theta / pi
        dy = np.cos(theta) * np.ones_like(phi)
        st = np.sin(theta)
        dx = st * np.cos(phi)
        dz = st * np.sin(phi)
        t = 0.5 * (dy + 1.0)
        up = horizon[None, None, :] * (1.0 - t[..., None]) * 2.0 + zenith[None, None, :] * (2.0 * t[..., None] - 1.0)
        down = ground[None, None, :] * (1.0 - 2.0 * t[..., None]) + horizon[None, None, :] * (2.0 * t[..., None])
        img = np.where((t >= 0.5)[..., None], up, down)
        c = np.clip(dx * s[0] + dy * s[1] + dz * s[2], 0.0, 1.0)
        lobe = sun_scale * c ** 256 + 0.6 * c ** 8
        img = img + lobe[..., None] * np.array([1.0, 0.92, 0.8])[None, None, :]
        idx = (np.arange(y0 * width, y1 * width, dtype=np.uint64).astype(np.uint32)).reshape(y1 - y0, width)
        noise = np.stack([_unit(seed, idx, k) for k in range(3)], axis=-1)
        img = img * (0.97 + 0.06 * noise)
        out[y0:y1] = img.astype(np.float32)
    return ExrImage(out.reshape(-1, 3), width, height)


def default_camera(width, height):
    """Viewport::default (viewport.rs:169-178): position (0,0,2), rotation 0, focal 35, sensor 55."""
    return PinholeCamera((0.0, 0.0, 2.0), (0.0, 0.0, 0.0), 35.0, 55.0, [width, height])


def scene_s1(skybox="synthetic", emitter=True):
    """S1: reference example scene + emissive sphere.  skybox: "synthetic" | None | ExrImage."""
    scene = Scene(default_sky_color=(0.6, 0.7, 0.9))
    scene.materials.append(Matrial(albedo=(1.0, 0.0, 1.0)))
    scene.spheres.append(Sphere((0.0, 0.0, 0.0), 0.5, 0))
    scene.materials.append(Matrial(albedo=(0.2, 0.3, 1.0)))
    scene.spheres.append(Sphere((0.0, -100.5, 0.0), 100.0, 1))
    if emitter:
        scene.materials.append(Matrial(albedo=(1.0, 1.0, 1.0), emission_color=(1.0, 0.8, 0.6), emissive_power=1.0))
        scene.spheres.append(Sphere((1.0, 0.25, -0.5), 0.25, 2))
    if isinstance(skybox, str) and skybox == "synthetic":
        scene.skybox = synthetic_sky()
    else:
        scene.skybox = skybox
    return scene


def scene_s2(n=10000, n_materials=64, seed=S2_SEED, skybox=None):
    """S2: ground sphere + (n-1) spheres, r in U[0.1,0.6], x in U[-40,40], z in U[-80,0], y = r - 0.5; every 20th
    material emissive (colour U[0.5,1]^3, power U[1,8]), the others albedo U[0.1,0.9]^3."""
    i = np.arange(n, dtype=np.uint32)
    r = 0.1 + 0.5 * _unit(seed, i, 0)
    x = -40.0 + 80.0 * _unit(seed, i, 1)
    z = -80.0 + 80.0 * _unit(seed, i, 2)
    y = r - 0.5
    xyzr = np.stack([x, y, z, r], axis=1).astype(np.float32)
    xyzr[0] = (0.0, -100.5, 0.0, 100.0)
    mid = (_mix32(i ^ np.uint32(seed)) % np.uint32(n_materials)).astype(np.int32)
    mid[0] = 0
    j = np.arange(n_materials, dtype=np.uint32)
    mats = np.zeros((n_materials, 9), dtype=np.float32)
    for k in range(3):
        mats[:, k] = 0.1 + 0.8 * _unit(seed ^ 0xABCD, j, k)
    mats[:, 3] = 0.5
    emissive = (j % 20) == 19
    for k in range(3):
        mats[:, 5 + k] = np.where(emissive, 0.5 + 0.5 * _unit(seed ^ 0xABCD, j, 3 + k), 0.0)
    mats[:, 8] = np.where(emissive, 1.0 + 7.0 * _unit(seed ^ 0xABCD, j, 6), 0.0)
    return FlatScene(xyzr, mid, mats, default_sky_color=(0.6, 0.7, 0.9), skybox=skybox)


def camera_s2(width, height):
    return PinholeCamera((0.0, 3.0, 6.0), (-0.35, 0.0, 0.0), 35.0, 55.0, [width, height])
