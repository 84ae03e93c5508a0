"""Ray record mirroring src/rays.py:14-18 (tmax = inf)."""
import numpy as np


class Ray:
    def __init__(self, origin, direction):
        self.origin = np.asarray(origin, dtype=np.float64)
        self.direction = np.asarray(direction, dtype=np.float64)
        self.tmax = np.inf
