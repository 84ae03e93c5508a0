"""Numeric constants and the presets the hot path uses.

Mirrors src/constants.py of the reference: the numeric constants (:7-12), the
vertex-kind enum ``Medium`` (:17-24) and the ``GLASS`` / ``GLASS_MAT`` preset
(:82-85, the mesh material of examples/LTS.ipynb cell 15).  The colour presets
for the surface renderer are out of scope.
"""
import enum

import numpy as np

from .material import Color, Material

inv_pi = 1.0 / np.pi
inv_2_pi = 0.5 * inv_pi
inv_4_pi = 0.25 * inv_pi
pi_over_2 = np.pi / 2.0
pi_over_4 = 0.5 * pi_over_2
EPSILON = 0.000001

ZEROS = np.zeros(3, dtype=np.float64)
ONES = np.ones(3, dtype=np.float64)


class Medium(enum.Enum):
    """Kinds of path vertex (NOT optical media; those are ``OpticalMedium``)."""
    NONE = 0
    DIFFUSE = 1
    GLOSSY = 2
    REFLECTIVE = 3
    TRANSMISSIVE = 4
    LIGHT = 5
    CAMERA = 6


def _rgb(*v):
    return np.array(v, dtype=np.float64)


WHITE = Color(ambient=_rgb(1, 1, 1), diffuse=_rgb(1, 1, 1), specular=_rgb(1, 1, 1))
GLASS = Color(ambient=_rgb(0.0, 0.0, 0.0), diffuse=_rgb(0.588235, 0.670588, 0.729412), specular=_rgb(0.9, 0.9, 0.9))
GLASS_MAT = Material(color=GLASS, shininess=96, reflection=0.2, ior=1.5, transmission=1.0,
                     is_diffuse=False, is_mirror=False)
