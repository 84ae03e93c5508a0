"""Geometric records mirroring src/primitives.py.

``PreComputedTriangle`` keeps the reference's constructor and field names
(:99-112): homogeneous float64[4] vertices (w = 1), ``centroid``, ``edge_1``,
``edge_2``, unit ``normal`` (w = 0) and the plane constant ``num``.  The unused
12-float ``transformation`` / ``intersect`` of the reference (:113-209) are not
carried.  ``AABB`` mirrors :75-80.
"""
import enum

import numpy as np


class ShapeOptions(enum.Enum):
    TRIANGLE = 1
    SPHERE = 2
    PLANE = 3
    AABB = 4
    TRIANGLEPC = 5


def _h(v, w):
    v = np.asarray(v, dtype=np.float64).ravel()
    if v.size == 3:
        v = np.append(v, w)
    if v.size != 4:
        raise ValueError("expected a 3- or 4-vector")
    return np.ascontiguousarray(v, dtype=np.float64)


class AABB:
    def __init__(self, min_point, max_point):
        self.type = ShapeOptions.AABB.value
        self.min_point = np.asarray(min_point, dtype=np.float64)
        self.max_point = np.asarray(max_point, dtype=np.float64)
        self.centroid = (self.min_point + self.max_point) / 2


class PreComputedTriangle:
    def __init__(self, vertex_1, vertex_2, vertex_3, material, is_light=False):
        self.type = ShapeOptions.TRIANGLEPC.value
        self.vertex_1 = _h(vertex_1, 1.0)
        self.vertex_2 = _h(vertex_2, 1.0)
        self.vertex_3 = _h(vertex_3, 1.0)
        self.material = material
        self.is_light = bool(is_light)
        self.centroid = (self.vertex_1 + self.vertex_2 + self.vertex_3) / 3
        self.edge_1 = self.vertex_2 - self.vertex_1
        self.edge_2 = self.vertex_3 - self.vertex_1
        raw = np.cross(self.edge_1[:3], self.edge_2[:3])
        self.normal = np.append(raw / np.sqrt(np.dot(raw, raw)), 0.0)
        self.num = float(np.dot(self.vertex_1[:3], raw))

    def vertices3(self):
        """[3, 3] float64 array of the Cartesian vertices (what the C ABI takes)."""
        return np.stack([self.vertex_1[:3], self.vertex_2[:3], self.vertex_3[:3]])
