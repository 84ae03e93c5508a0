"""Vector helpers mirroring src/vectors.py (normalize :6-7, get_direction :31-38)."""
import numpy as np


def normalize(vector):
    v = np.asarray(vector, dtype=np.float64)
    return v / np.sqrt(np.dot(v, v))


def unit_vector(vector):
    return normalize(vector)


def angle_between(v1, v2):
    return float(np.arccos(np.clip(np.dot(normalize(v1), normalize(v2)), -1.0, 1.0)))


def get_direction(p1, p2):
    return normalize(np.asarray(p1, dtype=np.float64) - np.asarray(p2, dtype=np.float64))
