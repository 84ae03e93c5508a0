"""ctypes front-end of the CPU oracle (oracle/liblt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  See
oracle/lt_oracle.h for what is pinned by reference golden vectors and what is
"parity unpinned" (the volumetric walk, absent from the reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FN = dict(HG_PDF=0, HG_SAMPLE=1, ONB=2, DISK=3, COSINE_HEMI=4, REFLECT=5, BOUNDARY=6, SPIN=7)
_FN_SHAPE = {0: (2, 1), 1: (2, 1), 2: (3, 6), 3: (2, 2), 4: (8, 4), 5: (6, 3), 6: (8, 5), 7: (5, 3)}
FX_SCALE = 2.0 ** 40
VERTEX_DTYPE = np.dtype([("point", "<f8", 3), ("direction", "<f8", 3), ("g_norm", "<f8", 3), ("throughput", "<f8"),
                         ("pdf_pos", "<f8"), ("pdf_dir", "<f8"), ("kind", "<i4"), ("medium", "<i4"), ("step", "<u4"),
                         ("pad_", "<u4")])


class Medium(C.Structure):
    _fields_ = [("mu_a", C.c_double), ("mu_s", C.c_double), ("g", C.c_double), ("n", C.c_double)]


class BvhNode(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3), ("offset", C.c_int32),
                ("n_prims", C.c_int32), ("axis", C.c_int32), ("pad_", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("photons", C.c_uint64), ("steps", C.c_uint64)] + [
        (k, C.c_double) for k in ("w_absorbed", "w_lost_outside_grid", "w_escaped_top", "w_escaped_bottom",
                                  "w_escaped_mesh", "w_specular", "w_roulette_net", "w_capped")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class _Scene(C.Structure):
    _fields_ = [
        ("n_media", C.c_int), ("media", C.POINTER(Medium)),
        ("n_layers", C.c_int), ("z_bounds", C.POINTER(C.c_double)), ("layer_medium", C.POINTER(C.c_int32)),
        ("n_above", C.c_double), ("n_below", C.c_double),
        ("n_tris", C.c_int), ("verts", C.POINTER(C.c_double)), ("med_front", C.POINTER(C.c_int32)),
        ("med_back", C.POINTER(C.c_int32)), ("n_nodes", C.c_int), ("nodes", C.POINTER(BvhNode)),
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("origin", C.c_double * 3), ("voxel", C.c_double * 3),
        ("src_type", C.c_int), ("src_pos", C.c_double * 3), ("src_dir", C.c_double * 3),
        ("src_extra", C.c_double * 6), ("start_medium", C.c_int), ("max_steps", C.c_uint32), ("quantity", C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liblt_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("lt_oracle.c", "lt_walk.inc", "lt_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.lto_run.restype = C.c_int
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def nodes_to_struct(nodes):
    """nodes: dict of arrays lo[N,3], hi[N,3], offset[N], n_prims[N], axis[N]."""
    n = len(nodes["offset"])
    arr = (BvhNode * max(n, 1))()
    for i in range(n):
        for k in range(3):
            arr[i].lo[k] = float(nodes["lo"][i][k])
            arr[i].hi[k] = float(nodes["hi"][i][k])
        arr[i].offset = int(nodes["offset"][i])
        arr[i].n_prims = int(nodes["n_prims"][i])
        arr[i].axis = int(nodes["axis"][i])
    return arr, n


class OracleScene:
    """Plain description of one transport problem; mirrors the lt_set_* calls."""

    def __init__(self, media, grid_shape, origin, voxel, layers=None, mesh=None, source=None, max_steps=1000000, quantity=0):
        self.media = [tuple(map(float, m)) for m in media]  # (mu_a, mu_s, g, n)
        self.nx, self.ny, self.nz = (int(v) for v in grid_shape)  # (nx, ny, nz)
        self.origin = tuple(map(float, origin))
        self.voxel = tuple(map(float, voxel))
        self.layers = layers  # dict(z_bounds, medium_idx, n_above, n_below)
        self.mesh = mesh      # dict(verts[T,3,3], med_front, med_back, nodes)
        self.source = source or dict(type=0, pos=(0, 0, 0), dir=(0, 0, 1), extra=(0,) * 6, start_medium=0)
        self.max_steps = int(max_steps)
        self.quantity = {"absorbed": 0, "fluence": 1}.get(quantity, quantity)     # lt_set_tally_quantity
        self._keep = []

    def _c(self):
        s = _Scene()
        keep = self._keep = []
        med = (Medium * len(self.media))(*[Medium(*m) for m in self.media])
        keep.append(med)
        s.n_media, s.media = len(self.media), med
        if self.layers is not None:
            zb = np.ascontiguousarray(self.layers["z_bounds"], dtype=np.float64)
            mi = np.ascontiguousarray(self.layers["medium_idx"], dtype=np.int32)
            keep += [zb, mi]
            s.n_layers, s.z_bounds, s.layer_medium = len(mi), _dp(zb), _ip(mi)
            s.n_above, s.n_below = float(self.layers.get("n_above", 1.0)), float(self.layers.get("n_below", 1.0))
        if self.mesh is not None:
            v = np.ascontiguousarray(self.mesh["verts"], dtype=np.float64).reshape(-1, 3, 3)
            mf = np.ascontiguousarray(self.mesh["med_front"], dtype=np.int32)
            mb = np.ascontiguousarray(self.mesh["med_back"], dtype=np.int32)
            nodes, nn = nodes_to_struct(self.mesh["nodes"])
            keep += [v, mf, mb, nodes]
            s.n_tris, s.verts, s.med_front, s.med_back = v.shape[0], _dp(v), _ip(mf), _ip(mb)
            s.n_nodes, s.nodes = nn, nodes
        s.nx, s.ny, s.nz = self.nx, self.ny, self.nz
        s.origin[:] = self.origin
        s.voxel[:] = self.voxel
        src = self.source
        s.src_type = int(src.get("type", 0))
        s.src_pos[:] = [float(x) for x in src["pos"]]
        d = np.asarray(src["dir"], dtype=np.float64)
        d = d / np.linalg.norm(d)
        s.src_dir[:] = list(d)
        ex = list(src.get("extra", (0,) * 6)) + [0] * 6
        s.src_extra[:] = [float(x) for x in ex[:6]]
        s.start_medium = int(src.get("start_medium", 0))
        s.max_steps = self.max_steps
        s.quantity = int(self.quantity)
        return s

    def run(self, n_photons, seed=0, photon_offset=0, rng_table=None, walk_f32=False, threads=1,
            want_fx=False, want_f64=True):
        """Returns (grid[nz,ny,nx] float64 or None, grid_fx uint64 or None, counters dict)."""
        s = self._c()
        nvox = self.nx * self.ny * self.nz
        g64 = np.zeros(nvox, dtype=np.float64) if want_f64 else None
        gfx = np.zeros(nvox, dtype=np.uint64) if want_fx else None
        tab, tsteps = None, 0
        if rng_table is not None:
            tab = np.ascontiguousarray(rng_table, dtype=np.float64)
            assert tab.ndim == 3 and tab.shape[0] == n_photons and tab.shape[2] == 4
            tsteps = tab.shape[1]
        cnt = Counters()
        rc = lib().lto_run(C.byref(s), C.c_uint64(n_photons), C.c_uint64(photon_offset), C.c_uint64(seed),
                           _dp(tab), C.c_uint64(tsteps), C.c_int(1 if walk_f32 else 0), _dp(g64),
                           gfx.ctypes.data_as(C.POINTER(C.c_uint64)) if gfx is not None else None,
                           C.byref(cnt), C.c_int(threads))
        if rc != 0:
            raise RuntimeError("lto_run failed: %d" % rc)
        shape = (self.nz, self.ny, self.nx)
        return (g64.reshape(shape) if g64 is not None else None,
                gfx.reshape(shape) if gfx is not None else None, cnt.as_dict())

    def run_capture(self, n_photons, max_vertices, seed=0, photon_offset=0):
        """f4: (grid, counters, vertices [n, K] records, counts [n]) with the light sub-path vertices stored."""
        s = self._c()
        g64 = np.zeros(self.nx * self.ny * self.nz, dtype=np.float64)
        v = np.zeros((n_photons, max_vertices), dtype=VERTEX_DTYPE)
        cnt = np.zeros(n_photons, dtype=np.uint32)
        c = Counters()
        rc = lib().lto_run_capture(C.byref(s), C.c_uint64(n_photons), C.c_uint64(photon_offset), C.c_uint64(seed),
                                   _dp(g64), C.byref(c), v.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
                                   C.c_uint32(max_vertices))
        if rc != 0:
            raise RuntimeError("lto_run_capture failed: %d" % rc)
        return g64.reshape(self.nz, self.ny, self.nx), c.as_dict(), v, cnt

    def intersect_rays(self, origins, dirs, tmax=None, use_bvh=True):
        s = self._c()
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(np.broadcast_to(tmax, (n,)), dtype=np.float64)
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        rc = lib().lto_intersect_rays(C.byref(s), _dp(o), _dp(d), _dp(tm), C.c_size_t(n), C.c_int(int(use_bvh)),
                                      _ip(prim), _dp(t))
        if rc != 0:
            raise RuntimeError("lto_intersect_rays failed: %d" % rc)
        return prim, t


class SurfaceMaterial(C.Structure):
    _fields_ = [("diffuse", C.c_double * 3), ("emission", C.c_double), ("ior", C.c_double),
                ("transmission", C.c_double), ("is_diffuse", C.c_int32), ("is_mirror", C.c_int32),
                ("is_light", C.c_int32), ("pad_", C.c_int32)]


class PointLight(C.Structure):
    _fields_ = [("source", C.c_double * 3), ("normal", C.c_double * 3), ("radiance", C.c_double * 3),
                ("total_area", C.c_double)]


def render_surface(scene, mats, lights, camera, f_distance, xs, ys, rand_0, rand_1, light_choice, image, old=False):
    """mats: [T, 9] rows (diffuse[3], emission, ior, transmission, is_diffuse, is_mirror, is_light);
    lights: [L, 10] rows (source[3], normal[3], radiance[3], total_area).  rand_0 and image updated in place.
    old=True: the recursive path_tracing_old integrator, light_choice [H, W, S, choices_per_sample]."""
    s = scene._c()
    mats = np.asarray(mats, dtype=np.float64); lights = np.asarray(lights, dtype=np.float64)
    ma = (SurfaceMaterial * len(mats))()
    for i, r in enumerate(mats):
        ma[i].diffuse[:] = list(r[:3]); ma[i].emission, ma[i].ior, ma[i].transmission = r[3], r[4], r[5]
        ma[i].is_diffuse, ma[i].is_mirror, ma[i].is_light = int(r[6]), int(r[7]), int(r[8])
    la = (PointLight * len(lights))()
    for i, r in enumerate(lights):
        la[i].source[:] = list(r[:3]); la[i].normal[:] = list(r[3:6]); la[i].radiance[:] = list(r[6:9]); la[i].total_area = r[9]
    H, W, S, D = rand_0.shape
    assert rand_0.dtype == np.float64 and rand_0.flags["C_CONTIGUOUS"] and image.flags["C_CONTIGUOUS"]
    lc = np.ascontiguousarray(light_choice, dtype=np.int32)
    r1 = np.ascontiguousarray(rand_1, dtype=np.float64)
    xs = np.ascontiguousarray(xs, dtype=np.float64); ys = np.ascontiguousarray(ys, dtype=np.float64)
    cam = (C.c_double * 3)(*[float(x) for x in np.asarray(camera).ravel()[:3]])
    if old:
        assert lc.shape[:3] == (H, W, S)
        rc = lib().lto_render_surface_old(C.byref(s), ma, la, C.c_int(len(lights)), C.c_int(W), C.c_int(H), C.c_int(S),
                                          C.c_int(D), cam, C.c_double(f_distance), _dp(xs), _dp(ys), _dp(rand_0), _dp(r1),
                                          _ip(lc), C.c_int(lc.shape[3]), _dp(image))
    else:
        rc = lib().lto_render_surface(C.byref(s), ma, la, C.c_int(len(lights)), C.c_int(W), C.c_int(H), C.c_int(S),
                                      C.c_int(D), cam, C.c_double(f_distance), _dp(xs), _dp(ys), _dp(rand_0), _dp(r1),
                                      _ip(lc), _dp(image))
    if rc != 0:
        raise RuntimeError("lto_render_surface failed: %d" % rc)
    return image


def eval_fn(name, inp):
    fn = FN[name]
    k_in, k_out = _FN_SHAPE[fn]
    a = np.ascontiguousarray(inp, dtype=np.float64).reshape(-1, k_in)
    out = np.empty((a.shape[0], k_out), dtype=np.float64)
    rc = lib().lto_eval(C.c_int(fn), _dp(a), C.c_size_t(a.shape[0]), _dp(out))
    if rc != 0:
        raise RuntimeError("lto_eval failed: %d" % rc)
    return out


def triangle_intersect(origins, dirs, tris):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 9)
    out = np.empty(o.shape[0], dtype=np.float64)
    lib().lto_triangle_intersect(_dp(o), _dp(d), _dp(t), C.c_size_t(o.shape[0]), _dp(out))
    return out


def intersect_bounds(origins, dirs, boxes, tmax=None):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
    n = o.shape[0]
    tm = None if tmax is None else np.ascontiguousarray(np.broadcast_to(tmax, (n,)), dtype=np.float64)
    out = np.empty(n, dtype=np.int32)
    lib().lto_intersect_bounds(_dp(o), _dp(d), _dp(tm), _dp(b), C.c_size_t(n), _ip(out))
    return out


def triangle_fields(tris):
    t = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 9)
    out = np.empty((t.shape[0], 16), dtype=np.float64)
    lib().lto_triangle_fields(_dp(t), C.c_size_t(t.shape[0]), _dp(out))
    return out


def rng_raw(seed, photon_id, count):
    out = np.empty(count, dtype=np.uint32)
    lib().lto_rng_raw(C.c_uint64(seed), C.c_uint64(photon_id), C.c_uint32(count),
                      out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def conservation_residual(c):
    tot = (c["w_absorbed"] + c["w_lost_outside_grid"] + c["w_escaped_top"] + c["w_escaped_bottom"]
           + c["w_escaped_mesh"] + c["w_specular"] + c["w_roulette_net"] + c["w_capped"])
    return tot - float(c["photons"])
