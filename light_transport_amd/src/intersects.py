"""Ray/triangle and ray/box tests, evaluated on the GPU.

Mirrors src/intersects.py: ``triangle_intersect`` (:46-104, Moller-Trumbore with
the reference's four epsilons), ``intersect_bounds`` (:179-196, slab test with
the far side widened by 1 + 2*gamma(3)), ``gamma`` / ``get_machine_epsilon``
(:229-235).  The scalar functions keep the reference signatures; the ``*_batch``
forms are what callers with many rays should use (one kernel launch).
"""
import numpy as np

from .._lib import default_context


def get_machine_epsilon():
    return np.finfo(np.float32).eps * 0.5


def gamma(n):
    eps = get_machine_epsilon()
    return (n * eps) / (1 - n * eps)


def _xyz(v):
    return np.asarray(v, dtype=np.float64).ravel()[:3]


def triangle_intersect_batch(ray_origins, ray_directions, triangles, ctx=None):
    """t per (ray, triangle) pair, NaN where the reference returns None.
    ``triangles``: [n, 3, 3] array or a sequence of PreComputedTriangle."""
    ctx = ctx or default_context()
    if not isinstance(triangles, np.ndarray):
        triangles = np.stack([t.vertices3() for t in triangles])
    return ctx.triangle_intersect(np.asarray(ray_origins, dtype=np.float64)[..., :3],
                                  np.asarray(ray_directions, dtype=np.float64)[..., :3], triangles)


def triangle_intersect(ray_origin, ray_direction, triangle):
    t = triangle_intersect_batch(_xyz(ray_origin)[None], _xyz(ray_direction)[None], triangle.vertices3()[None])[0]
    return None if np.isnan(t) else float(t)


def intersect_bounds_batch(boxes_min, boxes_max, ray_origins, ray_directions, tmax=None, ctx=None):
    ctx = ctx or default_context()
    boxes = np.concatenate([np.asarray(boxes_min, dtype=np.float64)[..., :3],
                            np.asarray(boxes_max, dtype=np.float64)[..., :3]], axis=-1)
    return ctx.intersect_bounds(np.asarray(ray_origins, dtype=np.float64)[..., :3],
                                np.asarray(ray_directions, dtype=np.float64)[..., :3], boxes, tmax).astype(bool)


def intersect_bounds(bounds, ray, inv_dir=None, hit0=None, hit1=None):
    """``inv_dir`` is accepted for signature compatibility; the kernel derives it
    from ``ray.direction`` exactly as the reference's caller does (bvh_new.py:418)."""
    return bool(intersect_bounds_batch(_xyz(bounds.min_point)[None], _xyz(bounds.max_point)[None],
                                       _xyz(ray.origin)[None], _xyz(ray.direction)[None], tmax=ray.tmax)[0])
