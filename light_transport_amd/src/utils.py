"""Sampling helpers and the nearest-hit wrapper, evaluated on the GPU.

Mirrors src/utils.py: ``hit_object`` (:53-68), ``create_orthonormal_system``
(:72-80), ``concentric_sample_disk`` (:115-128),
``cosine_weighted_hemisphere_sampling`` (:132-161).  Directions come back as
homogeneous float64[4] (w = 0) like the reference's.
"""
import numpy as np

from .._lib import default_context
from .bvh_new import intersect_bvh


def _xyz(v):
    return np.asarray(v, dtype=np.float64).ravel()[:3]


def hit_object(primitives, bvh, ray):
    nearest_object, min_distance = intersect_bvh(ray, primitives, bvh)
    if nearest_object is None:
        return None, None, None, None
    intersected_point = ray.origin + min_distance * ray.direction
    return nearest_object, min_distance, intersected_point, nearest_object.normal


def create_orthonormal_system(normal, ctx=None):
    out = (ctx or default_context()).eval("ONB", _xyz(normal)[None])[0]
    return out[:3].copy(), out[3:].copy()


def concentric_sample_disk(u, ctx=None):
    return (ctx or default_context()).eval("DISK", np.asarray(u, dtype=np.float64).ravel()[None, :2])[0].copy()


def cosine_weighted_hemisphere_sampling_batch(normals, incoming, rand, ctx=None):
    inp = np.concatenate([np.asarray(normals, dtype=np.float64)[..., :3], np.asarray(incoming, dtype=np.float64)[..., :3],
                          np.asarray(rand, dtype=np.float64)[..., :2]], axis=-1)
    out = (ctx or default_context()).eval("COSINE_HEMI", inp)
    return out[:, :3], out[:, 3]


def cosine_weighted_hemisphere_sampling(normal_at_intersection, incoming_direction, rand, ctx=None):
    d, pdf = cosine_weighted_hemisphere_sampling_batch(_xyz(normal_at_intersection)[None], _xyz(incoming_direction)[None],
                                                       np.asarray(rand, dtype=np.float64).ravel()[None, :2], ctx)
    return np.append(d[0], 0.0), float(pdf[0])
