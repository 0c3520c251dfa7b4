"""insploray::scene — mirror of reference core-engine/src/scene.rs (AoS, field-mutable, caller-owned)."""
import sys

import numpy as np

from .file_formats import ExrImage


class Matrial:  # sic: the reference spells it `Matrial` (scene.rs:5)
    """scene.rs:5-25"""

    def __init__(self, albedo=(1.0, 1.0, 1.0), roughness=0.5, metalic=0.0, emission_color=(0.0, 0.0, 0.0),
                 emissive_power=0.0):
        self.albedo = list(albedo)
        self.roughness = roughness
        self.metalic = metalic
        self.emission_color = list(emission_color)
        self.emissive_power = emissive_power


Material = Matrial


class Sphere:
    """scene.rs:27-31"""

    def __init__(self, position=(0.0, 0.0, 0.0), radius=1.0, material_id=-1):
        self.position = list(position)
        self.radius = radius
        self.material_id = material_id


class Scene:
    """scene.rs:34-41"""

    def __init__(self, spheres=None, materials=None, default_sky_color=(0.0, 0.0, 0.0), skybox=None):
        self.spheres = spheres if spheres is not None else []
        self.materials = materials if materials is not None else []
        self.default_sky_color = list(default_sky_color)
        self.skybox = skybox  # Option<ExrImage>

    @staticmethod
    def get_example_scene():
        """scene.rs:44-93: two Lambertian spheres, sky (0.6, 0.7, 0.9), skybox from ./assets/env if it loads."""
        skybox = None
        try:
            skybox = ExrImage.load_exr_image("./assets/env/default_skybox_1.exr")
        except Exception as e:  # the reference prints and continues with skybox: None (scene.rs:46-60)
            print(f"Failed loading EXR: {e}", file=sys.stderr)
        scene = Scene(default_sky_color=(0.6, 0.7, 0.9), skybox=skybox)
        scene.materials.append(Matrial(albedo=(1.0, 0.0, 1.0)))
        scene.spheres.append(Sphere(position=(0.0, 0.0, 0.0), radius=0.5, material_id=0))
        scene.materials.append(Matrial(albedo=(0.2, 0.3, 1.0)))
        scene.spheres.append(Sphere(position=(0.0, -100.5, 0.0), radius=100.0, material_id=1))
        return scene

    # ---- flattening for the C ABI (ir_set_spheres / ir_set_materials)
    def flat_spheres(self):
        n = len(self.spheres)
        xyzr = np.zeros((n, 4), dtype=np.float32)
        mid = np.zeros(n, dtype=np.int32)
        for i, s in enumerate(self.spheres):
            xyzr[i, 0:3] = s.position
            xyzr[i, 3] = s.radius
            mid[i] = s.material_id
        return xyzr, mid

    def flat_materials(self):
        m = np.zeros((len(self.materials), 9), dtype=np.float32)
        for i, t in enumerate(self.materials):
            m[i, 0:3] = t.albedo
            m[i, 3] = t.roughness
            m[i, 4] = t.metalic
            m[i, 5:8] = t.emission_color
            m[i, 8] = t.emissive_power
        return m


class FlatScene(Scene):
    """Scene given directly as arrays (large generated scenes): xyzr (n,4) f32, material_id (n,) i32,
    materials9 (m,9) f32.  Same fields otherwise."""

    def __init__(self, xyzr, material_id, materials9, default_sky_color=(0.0, 0.0, 0.0), skybox=None):
        super().__init__(default_sky_color=default_sky_color, skybox=skybox)
        self.xyzr = np.ascontiguousarray(xyzr, dtype=np.float32).reshape(-1, 4)
        self.material_id = np.ascontiguousarray(material_id, dtype=np.int32).reshape(-1)
        self.materials9 = np.ascontiguousarray(materials9, dtype=np.float32).reshape(-1, 9)

    def flat_spheres(self):
        return self.xyzr, self.material_id

    def flat_materials(self):
        return self.materials9
