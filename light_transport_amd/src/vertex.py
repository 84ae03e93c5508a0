"""Path vertex record (role of src/vertex.py:24-37) and its construction from the
device-side capture of light sub-paths.

The reference's ``Vertex`` has the fields below and nothing that fills them for a
light sub-path ever ran (``generate_light_subpaths`` / ``random_walk``,
bdpt.py:18-147,258-268, use stale signatures).  Here the photon walk itself stores
the first K vertices of every path (``lt_set_vertex_capture``); ``Vertex`` objects
are built from those records on request.
"""
import numpy as np

from .constants import Medium

VOLUME = 7  # interaction site inside a medium: no value in the reference's Medium enum


class Vertex:
    def __init__(self, point):
        self.point = np.asarray(point, dtype=np.float64)
        self.g_norm = np.zeros(3, dtype=np.float64)
        self.color = np.zeros(3, dtype=np.float64)
        self.pdf_pos = 0.0
        self.pdf_dir = 0.0
        self.importance = 0.0
        self.pdf_fwd = 0.0
        self.pdf_rev = 0.0
        self.hit_light = False
        self.medium = Medium.NONE.value
        self.throughput = np.ones(3, dtype=np.float64)
        self.geometry_term = np.zeros(3, dtype=np.float64)
        self.is_delta = False

    @classmethod
    def from_record(cls, rec, prev=None, nxt=None):
        """One captured vertex.  ``prev`` / ``nxt``: the neighbouring records of the same path, from which the
        reference's pdf_fwd / pdf_rev (bdpt.py:25-27,137) follow: pdf_fwd = density of the direction that LED here =
        prev.pdf_dir, pdf_rev = density of coming back from the next vertex = nxt.pdf_dir (Henyey-Greenstein and the
        Fresnel split are symmetric in the two directions); solid-angle measure -- ``convert_density`` turns them
        into the area measure bdpt.py:270-276 uses."""
        v = cls(np.array(rec["point"], dtype=np.float64))
        v.medium = int(rec["kind"])                       # Medium.LIGHT / REFLECTIVE / TRANSMISSIVE, or VOLUME
        v.throughput = np.full(3, float(rec["throughput"]))
        v.hit_light = v.medium == Medium.LIGHT.value
        v.is_delta = v.medium in (Medium.REFLECTIVE.value, Medium.TRANSMISSIVE.value)
        v.direction = np.array(rec["direction"], dtype=np.float64)
        v.g_norm = np.array(rec["g_norm"], dtype=np.float64)
        v.pdf_pos, v.pdf_dir = float(rec["pdf_pos"]), float(rec["pdf_dir"])
        v.pdf_fwd = float(prev["pdf_dir"]) if prev is not None else v.pdf_pos      # light vertex: bdpt.py:266 passes the emission pdf
        v.pdf_rev = float(nxt["pdf_dir"]) if nxt is not None else 0.0
        v.step = int(rec["step"])
        v.optical_medium = int(rec["medium"])
        return v


def convert_density(pdf, current_v, next_v):
    """Solid-angle density at ``current_v`` -> density per unit area at ``next_v`` (role of bdpt.py:270-276, with its
    truth test of an array replaced by "the geometric normal is non-zero")."""
    path = next_v.point[:3] - current_v.point[:3]
    d2 = float(np.dot(path, path))
    if d2 == 0.0:
        return 0.0
    if np.any(next_v.g_norm[:3] != 0.0):
        pdf = pdf * abs(float(np.dot(next_v.g_norm[:3], path / np.sqrt(d2))))
    return pdf / d2
