"""Wavefront OBJ ingest without pywavefront / pyvista.

Role of src/io.py:11-40 of the reference (``load_obj`` via pywavefront into legacy
``Triangle`` objects, with a stale ``Material(...)`` call).  This loader reads the
forms found in the reference's own assets (examples/obj/*.obj): ``v x y z``,
``f a b c``, ``f a b c d ...`` (fan-triangulated), ``f a/t``, ``f a//n``,
``f a/t/n`` and negative (relative) indices; everything else (vt, vn, g, o, s,
usemtl, mtllib) is skipped.  It returns ``PreComputedTriangle`` objects, the
primitive the BVH builder and the photon walk consume.
"""
import numpy as np

from .primitives import PreComputedTriangle


def read_obj(path):
    """-> (vertices [V, 3] float64, faces [F, 3] int64), polygons fan-triangulated."""
    verts, faces = [], []
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if f.size and (f.min() < 0 or f.max() >= len(v)):
        raise ValueError("%s: face index out of range" % path)
    return v, f


def triangles_from_mesh(vertices, faces, material, scale=1.0, translate=(0.0, 0.0, 0.0), drop_degenerate=True):
    v = np.asarray(vertices, dtype=np.float64) * float(scale) + np.asarray(translate, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if drop_degenerate and len(f):
        n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        f = f[np.array([np.dot(r, r) > 0.0 for r in n])]
    return PreComputedTriangle.batch(v[f[:, 0]], v[f[:, 1]], v[f[:, 2]], material) if len(f) else []


def load_obj(file_path, material=None, scale=1.0, translate=(0.0, 0.0, 0.0)):
    """Reference call shape (src/io.py:11-40): ``objects, dimension = load_obj(file_path)``.  ``objects`` are
    ``PreComputedTriangle`` primitives (the reference built legacy ``Triangle`` objects with a stale ``Material(...)``
    call; the BVH builder and the kernels consume PreComputedTriangle), ``dimension`` = abs(max(xmax, ymax, zmax)) of
    the file's vertices as the reference computes it (:24-27, before any scale / translate).  ``material`` defaults
    to the reference's choice, a red material with shininess 100 and reflection 0.5 (:33); scale / translate are
    extensions."""
    from .constants import RED
    from .material import Material
    if material is None:
        material = Material(color=RED, shininess=100, reflection=0.5, ior=1.0)
    v, f = read_obj(file_path)
    dimension = abs(float(v.max(axis=0).max())) if len(v) else 0.0
    return triangles_from_mesh(v, f, material, scale, translate), dimension
