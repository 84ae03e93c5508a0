"""Cornell-box geometry without PyVista.

``get_cornell_box(dim, surface_mat, left_wall_mat, right_wall_mat)`` keeps the
reference's signature (src/cornell_box.py:9) and its eight quads (right, left,
back, bottom and four ceiling strips around the 2x2 light opening, :12-19 and
:89-96), each cut along a fixed diagonal (the reference lets
``pv.Rectangle(...).triangulate()`` choose it).  ``get_cone`` restates
``pv.Cone(radius=2, height=5)`` of examples/LTS.ipynb cell 11 as apex + hexagon.
"""
import numpy as np

from .primitives import PreComputedTriangle


def _quad(p0, p1, p2, p3, material, is_light=False):
    return [PreComputedTriangle(p0, p1, p2, material, is_light), PreComputedTriangle(p0, p2, p3, material, is_light)]


def get_cornell_box(dim, surface_mat, left_wall_mat, right_wall_mat):
    a, b, c, d = (-dim, -dim, -dim), (-dim, -dim, dim), (dim, -dim, dim), (dim, -dim, -dim)
    e, f, g, h = (-dim, dim, -dim), (-dim, dim, dim), (dim, dim, dim), (dim, dim, -dim)
    i, j, k, l = (-1, dim, -dim), (-1, dim, -1), (-1, dim, 1), (-1, dim, dim)
    m, n, o, p = (1, dim, dim), (1, dim, 1), (1, dim, -1), (1, dim, -dim)
    tris = []
    tris += _quad(d, c, g, h, right_wall_mat)
    tris += _quad(f, b, a, e, left_wall_mat)
    tris += _quad(e, a, d, h, surface_mat)   # back
    tris += _quad(a, b, c, d, surface_mat)   # bottom
    tris += _quad(h, g, m, p, surface_mat)   # ceiling strips
    tris += _quad(n, m, l, k, surface_mat)
    tris += _quad(p, o, j, i, surface_mat)
    tris += _quad(i, l, f, e, surface_mat)
    return tris


def get_front_wall(dim, surface_mat):
    """Closes the open (camera) side z = +dim; the volumetric configs need a closed cavity."""
    return _quad((-dim, -dim, dim), (-dim, dim, dim), (dim, dim, dim), (dim, -dim, dim), surface_mat)


def get_light_quad(dim, source_mat):
    """The 2x2 ceiling light of LTS.ipynb cell 16 (two triangles, normal -y)."""
    return [PreComputedTriangle((-1, dim, -1), (1, dim, 1), (-1, dim, 1), source_mat, True),
            PreComputedTriangle((-1, dim, -1), (1, dim, -1), (1, dim, 1), source_mat, True)]


def get_cone(material, radius=2.0, height=5.0, resolution=6, center=(0.0, 0.0, 0.0)):
    """Apex on +x, regular ``resolution``-gon base on -x: resolution side faces +
    (resolution - 2) base faces (10 triangles at the default 6), outward normals."""
    cx, cy, cz = center
    apex = (cx + height / 2, cy, cz)
    ang = 2 * np.pi * np.arange(resolution) / resolution
    ring = [(cx - height / 2, cy + radius * np.cos(t), cz + radius * np.sin(t)) for t in ang]
    tris = [PreComputedTriangle(apex, ring[q], ring[(q + 1) % resolution], material) for q in range(resolution)]
    tris += [PreComputedTriangle(ring[0], ring[q + 1], ring[q], material) for q in range(1, resolution - 1)]
    return tris


def get_floor(x_dim, y_dim, z_dim, surface_mat):
    return _quad((-x_dim, -y_dim, -z_dim), (-x_dim, -y_dim, z_dim), (x_dim, -y_dim, z_dim), (x_dim, -y_dim, -z_dim),
                 surface_mat)
