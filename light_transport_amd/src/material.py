"""Plain records mirroring src/material.py (Color :9-13, Material :28-37)."""
import numpy as np


class Color:
    def __init__(self, ambient, diffuse, specular):
        self.ambient = np.asarray(ambient, dtype=np.float64)
        self.diffuse = np.asarray(diffuse, dtype=np.float64)
        self.specular = np.asarray(specular, dtype=np.float64)

    def __repr__(self):
        return "Color(diffuse=%s)" % (self.diffuse,)


class Material:
    """Surface record; ``ior`` is what the photon walk reads at boundaries."""

    def __init__(self, color, shininess, reflection, ior, emission=0.0, transmission=0.0, is_diffuse=True,
                 is_mirror=False):
        self.color = color
        self.shininess = float(shininess)
        self.reflection = float(reflection)
        self.ior = float(ior)
        self.emission = float(emission)
        self.transmission = float(transmission)
        self.is_diffuse = bool(is_diffuse)
        self.is_mirror = bool(is_mirror)

    def __repr__(self):
        return "Material(ior=%g, emission=%g, transmission=%g)" % (self.ior, self.emission, self.transmission)
