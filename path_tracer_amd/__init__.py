"""path_tracer_amd — MI355X (gfx950) backend for InsploRay's core-engine render path.

Python mirror of the reference crate's public surface (core-engine/src/lib.rs:8-16):
    insploray::renderer::RayTracer, insploray::scene::{Scene, Sphere, Matrial},
    insploray::cameras::{Camera, PinholeCamera, SharedCamera}, insploray::file_formats::ExrImage
over the C ABI of include/insploray_hip.h.  Importing this package loads libinsploray_hip.so and fails loudly if it
has not been built; there is no CPU fallback.
"""
from . import _capi
from ._capi import InsplorayError
from .cameras import Camera, PinholeCamera, SharedCamera
from .file_formats import ExrImage
from .ray import Ray
from .renderer import RayTracer
from .scene import FlatScene, Material, Matrial, Scene, Sphere

__all__ = ["RayTracer", "Scene", "FlatScene", "Sphere", "Matrial", "Material", "Camera", "PinholeCamera",
           "SharedCamera", "ExrImage", "Ray", "InsplorayError"]
