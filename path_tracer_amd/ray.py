"""insploray::ray::Ray (reference core-engine/src/ray.rs:3-6)."""
import numpy as np


class Ray:
    __slots__ = ("origin", "direction")

    def __init__(self, origin, direction):
        self.origin = np.asarray(origin, dtype=np.float32)
        self.direction = np.asarray(direction, dtype=np.float32)
