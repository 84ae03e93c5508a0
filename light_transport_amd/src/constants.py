"""Numeric constants and the presets the hot path uses.

Mirrors src/constants.py of the reference: the numeric constants (:7-12), the
vertex-kind enum ``Medium`` (:17-24) and the ``GLASS`` / ``GLASS_MAT`` preset
(:82-85, the mesh material of examples/LTS.ipynb cell 15) and the colour / material
presets the notebooks hand to the surface renderer (:25-85), kept as one table.
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


# name: (ambient, diffuse, specular) -- the values of constants.py:25-84
_COLOURS = {
    "WHITE": ((1, 1, 1), (1, 1, 1), (1, 1, 1)),
    "WHITE_2": ((0, 0, 0), (0.55, 0.55, 0.55), (0.7, 0.7, 0.7)),
    "RED": ((0.1, 0, 0), (0.7, 0, 0), (1, 1, 1)),
    "LEFT": ((0.1, 0, 0), (10, 2, 2), (1, 1, 1)),
    "PURPLE": ((0.1, 0, 0.1), (0.7, 0, 0.7), (1, 1, 1)),
    "YELLOW": ((0.05, 0.05, 0.0), (0.5, 0.5, 0.4), (0.7, 0.7, 0.04)),
    "SILVER": ((0.23125,) * 3, (0.2775,) * 3, (0.773911,) * 3),
    "GREEN": ((0, 0.1, 0), (0, 0.6, 0), (1, 1, 1)),
    "RIGHT": ((0, 0.1, 0), (2, 10, 2), (1, 1, 1)),
    "GREY": ((0.1, 0.1, 0.1), (0.6, 0.6, 0.6), (1, 1, 1)),
    "SURFACE": ((0.1, 0.1, 0.1), (6, 6, 6), (1, 1, 1)),
    "TURQUOISE": ((0.1, 0.18725, 0.1745), (0.396, 0.74151, 0.69102), (0.297254, 0.30829, 0.306678)),
    "BRONZE": ((0.2125, 0.1275, 0.054), (0.714, 0.4284, 0.18144), (0.393548, 0.271906, 0.166721)),
    "GLASS": ((0.0, 0.0, 0.0), (0.588235, 0.670588, 0.729412), (0.9, 0.9, 0.9)),
}
for _name, (_a, _d, _s) in _COLOURS.items():
    globals()[_name] = Color(ambient=_rgb(*_a), diffuse=_rgb(*_d), specular=_rgb(*_s))
TURQUOISE_MAT = Material(color=globals()["TURQUOISE"], shininess=0.1, reflection=2, ior=1.65)
BRONZE_MAT = Material(color=globals()["PURPLE"], shininess=10, reflection=0.75, ior=1.180, transmission=1.0,
                      is_diffuse=False, is_mirror=True)      # the mirror preset wears PURPLE (:79)
WHITE, GLASS = globals()["WHITE"], globals()["GLASS"]
GLASS_MAT = Material(color=GLASS, shininess=96, reflection=0.2, ior=1.5, transmission=1.0,
                     is_diffuse=False, is_mirror=False)
