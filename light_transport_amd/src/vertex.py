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
    def from_record(cls, rec):
        v = cls(np.array(rec["point"], dtype=np.float64))
        v.medium = int(rec["kind"])                       # Medium.LIGHT / REFLECTIVE / TRANSMISSIVE, or VOLUME
        v.throughput = np.full(3, float(rec["throughput"]))
        v.hit_light = v.medium == Medium.LIGHT.value
        v.is_delta = v.medium in (Medium.REFLECTIVE.value, Medium.TRANSMISSIVE.value)
        v.direction = np.array(rec["direction"], dtype=np.float64)
        v.step = int(rec["step"])
        v.optical_medium = int(rec["medium"])
        return v
