#!/usr/bin/env python3
"""BASELINE config 2 through the object API: 1e7 photons into a homogeneous semi-infinite slab, 256^3 grid.

    python examples/photon_slab.py [n_photons]

Mirrors how the reference's notebooks drive render_scene (examples/LTS_fix1.ipynb cell 26): build plain objects, make
one call, get a NumPy array back."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from light_transport_amd.src.photon_tracing import (LayeredSlab, OpticalMedium, PencilBeam, PhotonTracer, VoxelGrid,
                                                    fluence)

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
tissue = OpticalMedium(mu_a=0.1, mu_s=10.0, g=0.9, ior=1.0)                 # per mm
slab = LayeredSlab([tissue], [np.inf])
grid = VoxelGrid((256, 256, 256), origin=(-12.8, -12.8, 0.0), voxel=0.1)     # mm
tracer = PhotonTracer().configure(slab, grid, PencilBeam((0, 0, 0), (0, 0, 1)))
t0 = time.time()
tracer.run(n, seed=0)
absorbed = tracer.absorbed()                                                 # [nz, ny, nx] float64
dt = time.time() - t0
c = tracer.counters()
print("%d photons, %d photon-steps: %.1f ms on the device = %.1f G photon-steps/s (%.2f s wall for this first call, "
      "which also allocates the 40 GB deposit log; later calls reuse it)"
      % (n, c["steps"], tracer.kernel_ms(), c["steps"] / tracer.kernel_ms() / 1e6, dt))
print("absorbed %.4f, diffuse reflectance %.4f, lost outside the grid %.2e (fractions of launched weight)"
      % (c["w_absorbed"] / n, c["w_escaped_top"] / n, c["w_lost_outside_grid"] / n))
phi = fluence(absorbed, slab, grid, n)      # (heterogeneous scenes: configure(..., quantity="fluence") and tracer.fluence())
print("fluence on the beam axis at depth 0.05 / 1.05 / 5.05 mm: %.3f / %.3f / %.4f per mm^2"
      % (phi[0, 128, 128], phi[10, 128, 128], phi[50, 128, 128]))
