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
        self.quantity = "absorbed"
        self.photons = 0          # traced since the last reset (the fluence normalisation)

    def configure(self, geometry, grid, source, primitives=None, linear_bvh=None, max_steps=None, quantity="absorbed"):
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
        ctx.set_tally_quantity(quantity)     # "absorbed": weight absorbed per voxel; "fluence": w / mu_t per interaction
        self.grid, self.quantity, self.photons = grid, quantity, 0
        return self

    def run(self, n_photons, seed=0, photon_offset=0, rng_table=None, f32_walk=False, wait=True):
        self.ctx.launch(n_photons, seed=seed, photon_offset=photon_offset, rng_table=rng_table, f32_walk=f32_walk)
        self.photons += int(n_photons)
        if wait:
            self.ctx.sync()
        return self

    def reset(self):
        self.ctx.zero_tally()
        self.photons = 0

    def absorbed(self):
        """float64 [nz, ny, nx]: absorbed photon weight per voxel (configure(..., quantity="absorbed"))."""
        if self.quantity != "absorbed":
            raise ValueError("this tracer tallies %r: absorbed() needs configure(..., quantity='absorbed')" % self.quantity)
        return self.ctx.read_grid()

    def fluence(self):
        """float64 [nz, ny, nx]: fluence per launched photon (1 / area) -- the grid of a tracer configured with
        quantity="fluence", where every interaction adds w / mu_t to its voxel, divided by voxel volume x photons.
        Correct in heterogeneous media (two layers, a mesh of another medium) and in media that do not absorb
        (mu_a = 0), where absorbed / mu_a has no meaning.  Normalised tally, role of path_tracing_fix1.py:162-166."""
        if self.quantity != "fluence":
            raise ValueError("this tracer tallies %r: fluence() needs configure(..., quantity='fluence')" % self.quantity)
        if self.photons <= 0:
            raise ValueError("fluence(): nothing traced since the last reset")
        return self.ctx.read_grid() / (self.grid.voxel_volume * float(self.photons))

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


def fluence(absorbed, geometry, grid, n_photons):
    """Host post-step for an ABSORBED-weight grid of a LayeredSlab whose layer planes lie on voxel boundaries (Appendix
    C.4): fluence = absorbed / (mu_a(layer of the voxel) * dV * N), NaN where mu_a = 0.  Anything else -- a mesh, a plane
    inside a voxel, a medium that does not absorb -- has no per-voxel mu_a: trace with quantity="fluence" instead, where
    the device tallies w / mu_t directly (PhotonTracer.fluence / trace_photons(..., quantity="fluence"))."""
    if not isinstance(geometry, LayeredSlab):
        raise TypeError("fluence(): a per-voxel mu_a exists for a LayeredSlab only; use quantity='fluence' for meshes")
    a = np.asarray(absorbed, dtype=np.float64)
    nz = a.shape[0]
    z_lo = grid.origin[2] + grid.voxel[2] * np.arange(nz)
    z_hi = z_lo + grid.voxel[2]
    mu_a = np.full(nz, np.nan)
    tol = 1e-9 * grid.voxel[2]
    for k, m in enumerate(geometry.media):
        inside = (z_lo >= geometry.z_bounds[k] - tol) & (z_hi <= geometry.z_bounds[k + 1] + tol)
        mu_a[inside] = m.mu_a
    straddle = ~np.isfinite(mu_a) & (z_hi > geometry.z_bounds[0] + tol) & (z_lo < geometry.z_bounds[-1] - tol)
    if straddle.any():
        raise ValueError("fluence(): a layer plane cuts through voxel row(s) %s; use quantity='fluence'" % np.flatnonzero(straddle)[:4])
    with np.errstate(divide="ignore", invalid="ignore"):
        out = a / (np.where(mu_a > 0, mu_a, np.nan)[:, None, None] * grid.voxel_volume * float(n_photons))
    return out


def trace_photons(geometry, primitives, linear_bvh, n_photons, seed=0, grid=None, source=None, rng_table=None,
                  f32_walk=False, max_steps=None, device_id=0, return_counters=False, quantity="absorbed"):
    """One call, one array back, float64 [nz, ny, nx]: absorbed weight per voxel, or -- quantity="fluence" -- the
    fluence per launched photon (every interaction adds w / mu_t on the device; correct in heterogeneous media)."""
    if grid is None or source is None:
        raise ValueError("trace_photons needs grid= and source=")
    tr = PhotonTracer(ctx=_lib.default_context(device_id))
    tr.configure(geometry, grid, source, primitives, linear_bvh, max_steps, quantity)
    tr.run(n_photons, seed=seed, rng_table=rng_table, f32_walk=f32_walk)
    out = tr.fluence() if quantity == "fluence" else tr.absorbed()
    return (out, tr.counters()) if return_counters else out
