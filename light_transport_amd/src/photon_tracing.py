"""Volumetric photon transport -- the slot the reference left empty.

``src/photon_tracing.py`` of the reference is a 0-byte file and the medium
interaction of its walk is a TODO (bdpt.py:40).  This module fills the slot in
the reference's style -- configure plain objects, make ONE call, get a NumPy
array back, like ``render_scene(scene, primitives, bvh)``
(path_tracing_fix1.py:139-169):

    slab  = LayeredSlab([OpticalMedium(0.1, 10.0, 0.9, 1.0)], [np.inf])
    grid  = VoxelGrid((256, 256, 256), origin=(-12.8, -12.8, 0.0), voxel=(0.1,) * 3)
    dose  = trace_photons(slab, None, None, 10_000_000, seed=0, grid=grid,
                          source=PencilBeam((0, 0, 0), (0, 0, 1)))

The walk itself (hop / drop / spin) runs in the HIP kernels
behind include/lt.h; nothing here computes on the host.
"""
import numpy as np

from .. import _lib
from .bvh_new import linear_bvh_arrays, triangles_array


class OpticalMedium:
    """mu_a, mu_s in 1/length; g = Henyey-Greenstein anisotropy; ior as Material.ior."""

    def __init__(self, mu_a, mu_s, g, ior=1.0):
        self.mu_a, self.mu_s, self.g, self.ior = float(mu_a), float(mu_s), float(g), float(ior)

    def as_tuple(self):
        return (self.mu_a, self.mu_s, self.g, self.ior)

    def __repr__(self):
        return "OpticalMedium(mu_a=%g, mu_s=%g, g=%g, ior=%g)" % self.as_tuple()


class LayeredSlab:
    """Plane-parallel layers stacked along +z from ``z0``; the last thickness may be inf."""

    def __init__(self, media, thicknesses, z0=0.0, n_above=1.0, n_below=1.0):
        if len(media) != len(thicknesses) or not media:
            raise ValueError("LayeredSlab: one thickness per medium, at least one layer")
        self.media = list(media)
        self.z_bounds = np.concatenate([[float(z0)], float(z0) + np.cumsum(np.asarray(thicknesses, dtype=np.float64))])
        self.n_above, self.n_below = float(n_above), float(n_below)


class MeshVolume:
    """Media bounded by triangles.  ``media``: list of OpticalMedium; every
    primitive carries ``med_front`` / ``med_back`` (medium index on the +normal /
    -normal side, -1 = exterior) -- see ``set_triangle_media``."""

    def __init__(self, media, start_medium=0):
        self.media = list(media)
        self.start_medium = int(start_medium)


def set_triangle_media(primitives, front, back):
    for p in primitives:
        p.med_front, p.med_back = int(front), int(back)
    return primitives


class VoxelGrid:
    """Tally of absorbed photon weight: shape (nx, ny, nz), C-order [nz][ny][nx]."""

    def __init__(self, shape, origin, voxel, dtype="f64"):
        self.shape = tuple(int(s) for s in shape)
        self.origin = tuple(float(o) for o in origin)
        self.voxel = tuple(float(v) for v in (voxel if np.ndim(voxel) else (voxel,) * 3))
        self.dtype = dtype

    @property
    def voxel_volume(self):
        return self.voxel[0] * self.voxel[1] * self.voxel[2]


class PencilBeam:
    def __init__(self, position, direction):
        self.position, self.direction = tuple(map(float, position)), tuple(map(float, direction))


class AreaLight:
    """Parallelogram emitting cosine-weighted about ``normal`` (role of the two
    ``is_light`` triangles + ``sample_light``, light_samples.py:90-116)."""

    def __init__(self, corner, edge_1, edge_2, normal):
        self.corner, self.edge_1, self.edge_2 = (tuple(map(float, v)) for v in (corner, edge_1, edge_2))
        self.normal = tuple(map(float, normal))


def uniform_table(n_photons, steps, seed=0):
    """Table RNG in the reference's manner (scene.py:68-69): uniforms drawn up
    front by NumPy's legacy MT19937 and addressed by (photon, step)."""
    return np.random.RandomState(seed).rand(int(n_photons), int(steps), 4)


DEFAULT_MAX_STEPS = 1000000      # lt.h: the walk's hard cap


class PhotonTracer:
    """Owns one device context; ``configure`` once, ``run`` as often as needed
    (runs accumulate until ``reset``)."""

    def __init__(self, device_id=0, ctx=None):
        self.ctx = ctx or _lib.Context(device_id)
        self.grid = None
        self._mu_a_of_voxel = None

    def configure(self, geometry, grid, source, primitives=None, linear_bvh=None, max_steps=None):
        ctx = self.ctx
        ctx.set_media([m.as_tuple() for m in geometry.media])
        start_medium = 0
        if isinstance(geometry, LayeredSlab):
            ctx.set_layers(geometry.z_bounds, np.arange(len(geometry.media), dtype=np.int32), geometry.n_above,
                           geometry.n_below)
        elif isinstance(geometry, MeshVolume):
            if primitives is None or linear_bvh is None:
                raise ValueError("MeshVolume needs primitives (BVH order) and linear_bvh")
            mf = np.array([getattr(p, "med_front", -1) for p in primitives], dtype=np.int32)
            mb = np.array([getattr(p, "med_back", -1) for p in primitives], dtype=np.int32)
            ctx.set_mesh(triangles_array(primitives), mf, mb, linear_bvh_arrays(linear_bvh))
            start_medium = geometry.start_medium
        else:
            raise TypeError("geometry must be a LayeredSlab or a MeshVolume")
        ctx.set_grid(grid.shape, grid.origin, grid.voxel, grid.dtype)
        if isinstance(source, PencilBeam):
            ctx.set_source(_lib.SRC_PENCIL, source.position, source.direction, None, start_medium)
        elif isinstance(source, AreaLight):
            ctx.set_source(_lib.SRC_COSINE_QUAD, source.corner, source.normal,
                           list(source.edge_1) + list(source.edge_2), start_medium)
        else:
            raise TypeError("source must be a PencilBeam or an AreaLight")
        # every per-call setting is (re)set here, so that a call's result never depends on what an earlier user of the
        # same context left behind (the function-style API shares one context per device)
        ctx.set_max_steps(DEFAULT_MAX_STEPS if max_steps is None else max_steps)
        ctx.set_tally_mode("auto", 0)
        ctx.set_overlap(0)
        ctx.set_launch_config(0, 0)
        ctx.set_vertex_capture(0)
        self.grid = grid
        return self

    def run(self, n_photons, seed=0, photon_offset=0, rng_table=None, f32_walk=False, wait=True):
        self.ctx.launch(n_photons, seed=seed, photon_offset=photon_offset, rng_table=rng_table, f32_walk=f32_walk)
        if wait:
            self.ctx.sync()
        return self

    def reset(self):
        self.ctx.zero_tally()

    def absorbed(self):
        """float64 [nz, ny, nx]: absorbed photon weight per voxel."""
        return self.ctx.read_grid()

    def absorbed_raw(self):
        return self.ctx.read_grid_raw()

    def counters(self):
        return self.ctx.read_counters()

    def kernel_ms(self):
        return self.ctx.last_kernel_ms()


def generate_light_subpaths(tracer, n_photons, max_depth, seed=0, photon_offset=0, as_objects=False):
    """Role of generate_light_subpaths (bdpt.py:258-268): trace ``n_photons`` from the configured source and return
    the first ``max_depth`` vertices of every path.  Returns (records [n, max_depth], counts [n]); with
    ``as_objects`` a list of lists of ``Vertex``.  The grid tally is accumulated as in a normal run."""
    tracer.ctx.set_vertex_capture(max_depth)
    try:
        tracer.run(n_photons, seed=seed, photon_offset=photon_offset)
        rec, cnt = tracer.ctx.read_vertices(n_photons)
    finally:
        tracer.ctx.set_vertex_capture(0)
    if not as_objects:
        return rec, cnt
    from .vertex import Vertex
    out = []
    for i in range(int(n_photons)):
        m = int(cnt[i])
        out.append([Vertex.from_record(rec[i, k], rec[i, k - 1] if k > 0 else None, rec[i, k + 1] if k + 1 < m else None)
                    for k in range(m)])
    return out


def fluence(absorbed, mu_a, voxel_volume, n_photons):
    """Host post-step (Appendix C.4): fluence = absorbed / (mu_a * dV * N)."""
    return np.asarray(absorbed, dtype=np.float64) / (float(mu_a) * float(voxel_volume) * float(n_photons))


def trace_photons(geometry, primitives, linear_bvh, n_photons, seed=0, grid=None, source=None, rng_table=None,
                  f32_walk=False, max_steps=None, device_id=0, return_counters=False):
    """One call, one array back: absorbed weight per voxel, float64 [nz, ny, nx]."""
    if grid is None or source is None:
        raise ValueError("trace_photons needs grid= and source=")
    tr = PhotonTracer(ctx=_lib.default_context(device_id))
    tr.configure(geometry, grid, source, primitives, linear_bvh, max_steps)
    tr.run(n_photons, seed=seed, rng_table=rng_table, f32_walk=f32_walk)
    out = tr.absorbed()
    return (out, tr.counters()) if return_counters else out
