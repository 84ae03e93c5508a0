"""ctypes binding of liblt_hip.so (the C ABI declared in include/lt.h).

The product has no CPU fallback: if the shared library or a gfx950 device is
missing, every compute entry point raises ``LtError``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LT_HIP_LIBRARY") or os.path.join(_HERE, "liblt_hip.so")   # override: A/B builds of the same ABI

TALLY = {"f32": 0, "f64": 1, "u64fx": 2}
TALLY_NP = {0: np.float32, 1: np.float64, 2: np.uint64}
FX_SCALE = 2.0 ** 40
SRC_PENCIL, SRC_COSINE_QUAD = 0, 1
FLAG_F32_WALK = 1
FN = dict(HG_PDF=0, HG_SAMPLE=1, ONB=2, DISK=3, COSINE_HEMI=4, REFLECT=5, BOUNDARY=6, SPIN=7, WALK_MATH=8, WALK_MATH_RAW=9)
_FN_SHAPE = {0: (2, 1), 1: (2, 1), 2: (3, 6), 3: (2, 2), 4: (8, 4), 5: (6, 3), 6: (8, 5), 7: (5, 3), 8: (1, 5), 9: (1, 3)}

# every symbol include/lt.h declares (tests check the library exports them all)
SYMBOLS = [
    "lt_abi_version", "lt_create", "lt_destroy", "lt_last_error", "lt_set_media", "lt_set_layers", "lt_set_mesh",
    "lt_set_grid", "lt_set_source", "lt_set_max_steps", "lt_set_launch_config", "lt_launch", "lt_sync",
    "lt_last_kernel_ms", "lt_zero_tally", "lt_read_grid", "lt_read_grid_f64", "lt_read_counters",
    "lt_grid_device_ptr", "lt_counters_device_ptr", "lt_stream", "lt_reduce_grid", "lt_intersect_rays",
    "lt_triangle_intersect", "lt_intersect_bounds", "lt_eval", "lt_rng_raw", "lt_device_info",
    "lt_set_surface_materials", "lt_set_lights", "lt_render_surface", "lt_set_vertex_capture", "lt_read_vertices",
    "lt_set_tally_mode", "lt_last_log_stages", "lt_reserve_log", "lt_render_surface_old", "lt_set_overlap", "lt_last_log_hot_tiles",
    "lt_last_log_info", "lt_set_tally_quantity", "lt_set_tuning", "lt_mesh_accel_info", "lt_build_bvh",
]

# lt_vertex as a NumPy record (112 bytes, same layout as the C struct)
VERTEX_DTYPE = np.dtype([("point", "<f8", 3), ("direction", "<f8", 3), ("g_norm", "<f8", 3), ("throughput", "<f8"),
                         ("pdf_pos", "<f8"), ("pdf_dir", "<f8"), ("kind", "<i4"), ("medium", "<i4"), ("step", "<u4"),
                         ("pad_", "<u4")])
VERTEX_LIGHT, VERTEX_REFLECTIVE, VERTEX_TRANSMISSIVE, VERTEX_VOLUME = 5, 3, 4, 7


class LtError(RuntimeError):
    pass


NODE_DTYPE = np.dtype([("lo", "<f8", 3), ("hi", "<f8", 3), ("offset", "<i4"), ("n_prims", "<i4"), ("axis", "<i4"), ("pad_", "<i4")])   # lt_bvh_node


def build_bvh_arrays(verts, split_method=1):
    """lt_build_bvh (host C++, no device): verts [T, 3, 3] -> (order [T] int32: ordered_prims[i] = input triangle order[i],
    nodes: structured array of lt_bvh_node records in pre-order)."""
    v = _f64(verts).reshape(-1, 3, 3)
    T = v.shape[0]
    order = np.empty(T, dtype=np.int32)
    nodes = np.zeros(max(2 * T, 1), dtype=NODE_DTYPE)
    nn = C.c_int(0)
    rc = lib().lt_build_bvh(_dp(v), C.c_int(T), C.c_int(int(split_method)), _ip(order), nodes.ctypes.data_as(C.c_void_p),
                            C.c_int(len(nodes)), C.byref(nn))
    if rc:
        raise LtError("lt_build_bvh failed (%d): needs >= 1 triangle with finite vertices and split_method 0 or 1" % rc)
    return order, nodes[:nn.value]


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


class SurfaceMaterial(C.Structure):
    _fields_ = [("diffuse", C.c_double * 3), ("emission", C.c_double), ("ior", C.c_double),
                ("transmission", C.c_double), ("is_diffuse", C.c_int32), ("is_mirror", C.c_int32),
                ("is_light", C.c_int32), ("pad_", C.c_int32)]


class PointLight(C.Structure):
    _fields_ = [("source", C.c_double * 3), ("normal", C.c_double * 3), ("radiance", C.c_double * 3),
                ("total_area", C.c_double)]


def pack_surface_materials(primitives):
    """lt_surface_material per primitive from the reference-style objects (Material + is_light)."""
    arr = (SurfaceMaterial * max(len(primitives), 1))()
    for i, p in enumerate(primitives):
        m = p.material
        arr[i].diffuse[:] = [float(x) for x in m.color.diffuse[:3]]
        arr[i].emission, arr[i].ior, arr[i].transmission = float(m.emission), float(m.ior), float(m.transmission)
        arr[i].is_diffuse, arr[i].is_mirror, arr[i].is_light = int(m.is_diffuse), int(m.is_mirror), int(p.is_light)
    return arr


def pack_lights(lights):
    """lt_point_light per Light sample (scene.py:12-17); radiance = emission * color.diffuse."""
    arr = (PointLight * max(len(lights), 1))()
    for i, l in enumerate(lights):
        arr[i].source[:] = [float(x) for x in l.source[:3]]
        arr[i].normal[:] = [float(x) for x in l.normal[:3]]
        arr[i].radiance[:] = [float(l.material.emission * x) for x in l.material.color.diffuse[:3]]
        arr[i].total_area = float(l.total_area)
    return arr


_lib = None


def build(verbose=False):
    """Compile liblt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LtError("liblt_hip.so is not built (run light_transport_amd.build() or "
                          "`make -C light_transport_amd/csrc`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.lt_last_error.restype = C.c_char_p
        L.lt_last_error.argtypes = [C.c_void_p]
        L.lt_stream.restype = C.c_void_p
        L.lt_stream.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


class Context:
    """One lt_ctx: one GPU, one HIP stream, one voxel grid."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        rc = lib().lt_create(C.byref(self._h), C.c_int(device_id))
        if rc != 0:
            msg = lib().lt_last_error(None)
            raise LtError("lt_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.device_id = device_id
        self._grid_shape = None
        self._tally = None

    # -- plumbing ---------------------------------------------------------
    def _ck(self, rc, what):
        if rc != 0:
            msg = lib().lt_last_error(self._h)
            raise LtError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().lt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene ------------------------------------------------------------
    def set_media(self, media):
        arr = (Medium * len(media))(*[Medium(*map(float, m)) for m in media])
        self._ck(lib().lt_set_media(self._h, arr, C.c_int(len(media))), "lt_set_media")

    def set_layers(self, z_bounds, medium_idx, n_above=1.0, n_below=1.0):
        zb = _f64(z_bounds)
        mi = np.ascontiguousarray(medium_idx, dtype=np.int32)
        if zb.size != mi.size + 1:
            raise LtError("set_layers: need len(z_bounds) == len(medium_idx) + 1")
        self._ck(lib().lt_set_layers(self._h, _dp(zb), _ip(mi), C.c_int(mi.size), C.c_double(n_above),
                                     C.c_double(n_below)), "lt_set_layers")

    def set_mesh(self, verts, med_front, med_back, nodes):
        """nodes: lt_bvh_node records -- a structured array (NODE_DTYPE, as build_bvh_arrays returns) or a dict of arrays
        lo [N, 3], hi [N, 3], offset, n_prims, axis (as linear_bvh_arrays returns)."""
        v = _f64(verts).reshape(-1, 3, 3)
        mf = np.ascontiguousarray(med_front, dtype=np.int32)
        mb = np.ascontiguousarray(med_back, dtype=np.int32)
        if isinstance(nodes, np.ndarray) and nodes.dtype == NODE_DTYPE:
            arr = np.ascontiguousarray(nodes)
        else:       # packed with array assignments: a per-node Python loop costs ~0.2 s on a 20 000-node tree
            n_ = len(nodes["offset"])
            arr = np.zeros(n_, dtype=NODE_DTYPE)
            arr["lo"] = np.asarray(nodes["lo"], dtype=np.float64).reshape(n_, 3); arr["hi"] = np.asarray(nodes["hi"], dtype=np.float64).reshape(n_, 3)
            arr["offset"] = nodes["offset"]; arr["n_prims"] = nodes["n_prims"]; arr["axis"] = nodes["axis"]
        n = len(arr)
        if mf.size != v.shape[0] or mb.size != v.shape[0]:
            raise LtError("set_mesh: medium arrays must have one entry per triangle")
        self._ck(lib().lt_set_mesh(self._h, _dp(v), _ip(mf), _ip(mb), C.c_int(v.shape[0]), arr.ctypes.data_as(C.c_void_p), C.c_int(n)),
                 "lt_set_mesh")

    def set_grid(self, shape, origin, voxel, dtype="f64"):
        nx, ny, nz = (int(s) for s in shape)
        o = (C.c_double * 3)(*map(float, origin))
        vx = (C.c_double * 3)(*map(float, voxel))
        t = TALLY[dtype] if isinstance(dtype, str) else int(dtype)
        self._ck(lib().lt_set_grid(self._h, C.c_int(nx), C.c_int(ny), C.c_int(nz), o, vx, C.c_int(t)), "lt_set_grid")
        self._grid_shape = (nz, ny, nx)
        self._tally = t

    def set_source(self, type=SRC_PENCIL, pos=(0, 0, 0), dir=(0, 0, 1), extra=None, start_medium=0):
        p = (C.c_double * 3)(*map(float, pos))
        d = (C.c_double * 3)(*map(float, dir))
        ex = (C.c_double * 6)(*map(float, (list(extra) + [0.0] * 6)[:6])) if extra is not None else None
        self._ck(lib().lt_set_source(self._h, C.c_int(type), p, d, ex, C.c_int(start_medium)), "lt_set_source")

    def set_max_steps(self, n):
        self._ck(lib().lt_set_max_steps(self._h, C.c_uint32(int(n))), "lt_set_max_steps")

    def set_tally_quantity(self, quantity):
        """"absorbed" (default): an interaction adds the absorbed weight w mu_a / mu_t to its voxel; "fluence": it adds
        w / mu_t, so that grid / (voxel volume x photons) IS the fluence, in heterogeneous media too (lt.h)."""
        q = {"absorbed": 0, "fluence": 1}.get(quantity, quantity)
        self._ck(lib().lt_set_tally_quantity(self._h, C.c_int(int(q))), "lt_set_tally_quantity")

    def mesh_accel_info(self):
        """dict(kind, march_dims, march_entries, clearance_dims) of the current mesh (lt.h: lt_mesh_accel_info)."""
        kind, ent = C.c_int(0), C.c_uint64(0)
        md, cd = (C.c_int * 3)(), (C.c_int * 3)()
        self._ck(lib().lt_mesh_accel_info(self._h, C.byref(kind), md, C.byref(ent), cd), "lt_mesh_accel_info")
        return dict(kind=kind.value, march_dims=tuple(md), march_entries=ent.value, clearance_dims=tuple(cd))

    def set_tuning(self, key, value=-1):
        """An experiment knob of the library (lt.h: lt_set_tuning); value < 0 (default) restores the built-in default."""
        self._ck(lib().lt_set_tuning(self._h, key.encode(), C.c_int64(int(value))), "lt_set_tuning")

    def tuning(self, **knobs):
        """Context manager: the given knobs for the duration of the block, defaults restored afterwards."""
        ctx = self

        class _T:
            def __enter__(self_):
                for k, v in knobs.items():
                    ctx.set_tuning(k, v)
                return ctx

            def __exit__(self_, *a):
                for k in knobs:
                    ctx.set_tuning(k, -1)
        return _T()

    def set_launch_config(self, blocks_per_cu=0, threads_per_block=0):
        self._ck(lib().lt_set_launch_config(self._h, C.c_int(blocks_per_cu), C.c_int(threads_per_block)),
                 "lt_set_launch_config")

    def set_tally_mode(self, mode="auto", log_bytes=0):
        """'atomic': global atomics per deposit; 'log': deposit log + tile partition + LDS reduce;
        'auto' (default): log for layered slabs and f32 mesh walks, atomic for f64 mesh walks."""
        m = {"atomic": 0, "log": 1, "auto": 2}[mode] if isinstance(mode, str) else int(mode)
        self._ck(lib().lt_set_tally_mode(self._h, C.c_int(m), C.c_uint64(int(log_bytes))), "lt_set_tally_mode")

    def set_overlap(self, lanes=0):
        """Overlap inside one launch: 2 = sub-batches alternate between two streams of the ctx (one batch's log
        reduction runs beside the next batch's walk), 1 = one stream, 0 = auto (try both once, keep the faster)."""
        self._ck(lib().lt_set_overlap(self._h, C.c_int(int(lanes))), "lt_set_overlap")

    def last_log_info(self):
        """dict(records, overflow_records, batches, lanes) of the last log-mode launch, or None."""
        rec, ovf, bat, lanes = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        if lib().lt_last_log_info(self._h, C.byref(rec), C.byref(ovf), C.byref(bat), C.byref(lanes)) != 0:
            return None
        return dict(records=rec.value, overflow_records=ovf.value, batches=bat.value, lanes=lanes.value)

    def last_log_hot_tiles(self):
        """(hot tiles, pilot-count threshold) of the last log-mode launch: (0, 0) unless the hot-tile two-pass form ran."""
        h, t = C.c_uint32(), C.c_uint32()
        self._ck(lib().lt_last_log_hot_tiles(self._h, C.byref(h), C.byref(t)), "lt_last_log_hot_tiles")
        return h.value, t.value

    def reserve_log(self, n_photons):
        self._ck(lib().lt_reserve_log(self._h, C.c_uint64(int(n_photons))), "lt_reserve_log")

    # -- run --------------------------------------------------------------
    def launch(self, n_photons, seed=0, photon_offset=0, rng_table=None, f32_walk=False):
        tab, steps = None, 0
        if rng_table is not None:
            tab = _f64(rng_table)
            if tab.ndim != 3 or tab.shape[0] != n_photons or tab.shape[2] != 4:
                raise LtError("rng_table must have shape [n_photons, steps, 4]")
            steps = tab.shape[1]
        flags = FLAG_F32_WALK if f32_walk else 0
        self._captured_max_vertices = getattr(self, "_max_vertices", 0)
        self._ck(lib().lt_launch(self._h, C.c_uint64(int(n_photons)), C.c_uint64(int(photon_offset)),
                                 C.c_uint64(int(seed) & (2 ** 64 - 1)), _dp(tab), C.c_uint64(steps),
                                 C.c_uint32(flags)), "lt_launch")

    def sync(self):
        self._ck(lib().lt_sync(self._h), "lt_sync")

    def last_kernel_ms(self):
        ms = C.c_double()
        self._ck(lib().lt_last_kernel_ms(self._h, C.byref(ms)), "lt_last_kernel_ms")
        return ms.value

    def last_log_stages(self):
        """dict(walk_ms, scan_ms, partition_ms, reduce_ms, records, batches) of the last log-mode launch, or None."""
        ms = (C.c_double * 4)()
        rec, bat = C.c_uint64(), C.c_uint64()
        if lib().lt_last_log_stages(self._h, ms, C.byref(rec), C.byref(bat)) != 0:
            return None
        return dict(walk_ms=ms[0], scan_ms=ms[1], partition_ms=ms[2], reduce_ms=ms[3], records=rec.value, batches=bat.value)

    def zero_tally(self):
        self._ck(lib().lt_zero_tally(self._h), "lt_zero_tally")

    # -- readback ---------------------------------------------------------
    def read_grid_raw(self):
        out = np.empty(self._grid_shape, dtype=TALLY_NP[self._tally])
        self._ck(lib().lt_read_grid(self._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)), "lt_read_grid")
        return out

    def read_grid_into(self, buf):
        """Raw tally into a caller-owned contiguous buffer of exactly the grid's size in bytes (e.g. pinned host
        memory: the D2H copy then runs at the link's rate)."""
        a = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
        self._ck(lib().lt_read_grid(self._h, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)), "lt_read_grid")
        return buf

    def read_grid(self):
        """Absorbed weight per voxel as float64 [nz, ny, nx]."""
        out = np.empty(self._grid_shape, dtype=np.float64)
        self._ck(lib().lt_read_grid_f64(self._h, _dp(out), C.c_size_t(out.size)), "lt_read_grid_f64")
        return out

    def read_counters(self):
        c = Counters()
        self._ck(lib().lt_read_counters(self._h, C.byref(c)), "lt_read_counters")
        return c.as_dict()

    def grid_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(lib().lt_grid_device_ptr(self._h, C.byref(p), C.byref(n)), "lt_grid_device_ptr")
        return p.value, n.value

    def counters_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(lib().lt_counters_device_ptr(self._h, C.byref(p), C.byref(n)), "lt_counters_device_ptr")
        return p.value, n.value

    def stream(self):
        return lib().lt_stream(self._h)

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mhz, mem = C.c_int(), C.c_int(), C.c_size_t()
        self._ck(lib().lt_device_info(self._h, name, C.c_size_t(256), C.byref(cus), C.byref(mhz), C.byref(mem)),
                 "lt_device_info")
        return dict(name=name.value.decode(), cus=cus.value, clock_mhz=mhz.value, hbm_bytes=mem.value)

    # -- device-side queries ----------------------------------------------
    def intersect_rays(self, origins, dirs, tmax=None, use_bvh=True):
        o, d = _f64(origins).reshape(-1, 3), _f64(dirs).reshape(-1, 3)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(np.broadcast_to(tmax, (n,)), dtype=np.float64)
        prim, t = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
        self._ck(lib().lt_intersect_rays(self._h, _dp(o), _dp(d), _dp(tm), C.c_size_t(n), C.c_int(int(use_bvh)),
                                         _ip(prim), _dp(t)), "lt_intersect_rays")
        return prim, t

    def triangle_intersect(self, origins, dirs, tris):
        o, d, t = _f64(origins).reshape(-1, 3), _f64(dirs).reshape(-1, 3), _f64(tris).reshape(-1, 9)
        out = np.empty(o.shape[0], dtype=np.float64)
        self._ck(lib().lt_triangle_intersect(self._h, _dp(o), _dp(d), _dp(t), C.c_size_t(o.shape[0]), _dp(out)),
                 "lt_triangle_intersect")
        return out

    def intersect_bounds(self, origins, dirs, boxes, tmax=None):
        o, d, b = _f64(origins).reshape(-1, 3), _f64(dirs).reshape(-1, 3), _f64(boxes).reshape(-1, 6)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(np.broadcast_to(tmax, (n,)), dtype=np.float64)
        out = np.empty(n, dtype=np.int32)
        self._ck(lib().lt_intersect_bounds(self._h, _dp(o), _dp(d), _dp(tm), _dp(b), C.c_size_t(n), _ip(out)),
                 "lt_intersect_bounds")
        return out

    def eval(self, name, inp):
        fn = FN[name]
        k_in, k_out = _FN_SHAPE[fn]
        a = _f64(inp).reshape(-1, k_in)
        out = np.empty((a.shape[0], k_out), dtype=np.float64)
        self._ck(lib().lt_eval(self._h, C.c_int(fn), _dp(a), C.c_size_t(a.shape[0]), _dp(out)), "lt_eval")
        return out

    # -- light sub-path vertices (f4) ----------------------------------------
    def set_vertex_capture(self, max_vertices_per_photon):
        self._ck(lib().lt_set_vertex_capture(self._h, C.c_uint32(int(max_vertices_per_photon))), "lt_set_vertex_capture")
        self._max_vertices = int(max_vertices_per_photon)

    def read_vertices(self, n_photons):
        """-> (vertices [n_photons, K] record array of VERTEX_DTYPE, counts [n_photons] uint32) of the last launch."""
        k = getattr(self, "_captured_max_vertices", 0)     # the K in effect at the capturing launch
        v = np.zeros((int(n_photons), k), dtype=VERTEX_DTYPE)
        cnt = np.zeros(int(n_photons), dtype=np.uint32)
        self._ck(lib().lt_read_vertices(self._h, v.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
                                        C.c_uint64(int(n_photons))), "lt_read_vertices")
        return v, cnt

    # -- surface path tracing (f2) ------------------------------------------
    def set_surface_materials(self, mats):
        self._ck(lib().lt_set_surface_materials(self._h, mats, C.c_int(len(mats))), "lt_set_surface_materials")

    def set_lights(self, lights):
        self._ck(lib().lt_set_lights(self._h, lights, C.c_int(len(lights))), "lt_set_lights")

    def render_surface(self, camera, f_distance, xs, ys, rand_0, rand_1, light_choice, image, old=False):
        """rand_0 [H,W,S,D] and image [H,W,3] are updated in place (C-contiguous float64).  old=True: the recursive
        path_tracing_old integrator; light_choice is then [H,W,S,choices_per_sample] and image is overwritten."""
        H, W, S, D = rand_0.shape
        for a in (rand_0, rand_1, image):
            if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
                raise LtError("render_surface: tables and image must be C-contiguous float64")
        lc = np.ascontiguousarray(light_choice, dtype=np.int32)
        lc_ok = (lc.ndim == 4 and lc.shape[:3] == (H, W, S) and lc.shape[3] > 0) if old else lc.shape == rand_0.shape
        if rand_1.shape != rand_0.shape or not lc_ok or image.shape != (H, W, 3):
            raise LtError("render_surface: inconsistent table / image shapes")
        cam = (C.c_double * 3)(*[float(x) for x in np.asarray(camera).ravel()[:3]])
        xs, ys = _f64(xs), _f64(ys)
        if xs.size != W or ys.size != H:
            raise LtError("render_surface: xs / ys must have W / H entries")
        if old:
            self._ck(lib().lt_render_surface_old(self._h, C.c_int(W), C.c_int(H), C.c_int(S), C.c_int(D), cam,
                                                 C.c_double(f_distance), _dp(xs), _dp(ys), _dp(rand_0), _dp(rand_1),
                                                 _ip(lc), C.c_int(lc.shape[3]), _dp(image)), "lt_render_surface_old")
            return image
        self._ck(lib().lt_render_surface(self._h, C.c_int(W), C.c_int(H), C.c_int(S), C.c_int(D), cam,
                                         C.c_double(f_distance), _dp(xs), _dp(ys), _dp(rand_0), _dp(rand_1), _ip(lc),
                                         _dp(image)), "lt_render_surface")
        return image

    def rng_raw(self, seed, photon_id, count):
        out = np.empty(count, dtype=np.uint32)
        self._ck(lib().lt_rng_raw(self._h, C.c_uint64(seed), C.c_uint64(photon_id), C.c_uint32(count),
                                  out.ctypes.data_as(C.POINTER(C.c_uint32))), "lt_rng_raw")
        return out


_default_ctx = {}


def default_context(device_id=0):
    """Process-wide context per device, for the function-style API."""
    ctx = _default_ctx.get(device_id)
    if ctx is None:
        ctx = _default_ctx[device_id] = Context(device_id)
    return ctx
