"""Participating-medium sampling, evaluated on the GPU.

``henyey_greenstein(cosTheta, g)`` mirrors src/medium_samples.py:14-16 (pbrt
sign convention: for g > 0 the value peaks at cosTheta = -1).
``sample_henyey_greenstein`` is the inverse-CDF sampler the photon walk uses, in
the deflection-angle convention: its density is henyey_greenstein(-cos, g)
(the reference has no sampler).
"""
import numpy as np

from .._lib import default_context


def henyey_greenstein(cosTheta, g, ctx=None):
    c = np.asarray(cosTheta, dtype=np.float64)
    inp = np.stack([c.ravel(), np.broadcast_to(np.float64(g), c.size)], axis=1)
    out = (ctx or default_context()).eval("HG_PDF", inp)[:, 0].reshape(c.shape)
    return float(out) if out.ndim == 0 else out


def sample_henyey_greenstein(xi, g, ctx=None):
    x = np.asarray(xi, dtype=np.float64)
    inp = np.stack([x.ravel(), np.broadcast_to(np.float64(g), x.size)], axis=1)
    out = (ctx or default_context()).eval("HG_SAMPLE", inp)[:, 0].reshape(x.shape)
    return float(out) if out.ndim == 0 else out
