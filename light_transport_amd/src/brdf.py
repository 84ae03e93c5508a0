"""``get_reflected_direction`` (role of src/brdf.py:8-9), evaluated on the GPU.
The Phong terms of the reference (:13-47) belong to its surface renderer and
are out of scope."""
import numpy as np

from .._lib import default_context


def get_reflected_direction_batch(vectors, axes, ctx=None):
    inp = np.concatenate([np.asarray(vectors, dtype=np.float64)[..., :3], np.asarray(axes, dtype=np.float64)[..., :3]], axis=-1)
    return (ctx or default_context()).eval("REFLECT", inp)


def get_reflected_direction(vector, axis, ctx=None):
    v = np.asarray(vector, dtype=np.float64).ravel()
    r = get_reflected_direction_batch(v[None, :3], np.asarray(axis, dtype=np.float64).ravel()[None, :3], ctx)[0]
    return np.append(r, 0.0) if v.size == 4 else r.copy()
