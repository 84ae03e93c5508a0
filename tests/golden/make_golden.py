#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE's own
function bodies.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

How the reference is executed.  Every reference module does ``import numba`` and
Numba is not installed in this image (and cannot be: no network).  The reference's
jitted functions are ordinary Python/NumPy underneath their decorators, so this
script puts a small identity-decorator module named ``numba`` in a temp directory
on sys.path (njit/jit return the function unchanged, jitclass returns the class,
prange = range, type tokens are inert) and imports the reference from
/root/reference.  What is captured is therefore the output of the reference's
function bodies run by CPython + NumPy float64 -- no JIT.  The shipped
``__pycache__/*.pyc`` files are NOT loaded (sys.pycache_prefix is redirected) and
nothing is written under /root/reference.

Only numbers are stored: inputs and the reference's outputs (.npz).  No reference
source text is copied.  Fixture ids follow SURVEY.md section 8(c).
"""
import os
import sys
import tempfile
import textwrap

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(OUT))

STUB = textwrap.dedent('''
    """Identity stand-in for the numba surface the reference touches (test tooling)."""
    import sys, types

    def _decorator(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs and not isinstance(args[0], _Tok):
            return args[0]
        return lambda f: f
    njit = jit = _decorator
    prange = range

    class _Tok:
        def __getitem__(self, k): return self
        def __call__(self, *a, **k): return self
        def define(self, *a, **k): return None
    float64 = float32 = intp = uintp = int8 = int64 = boolean = _Tok()
    def optional(x): return _Tok()
    def deferred_type(): return _Tok()
    def typeof(x): return _Tok()

    types_mod = types.ModuleType("numba.types")
    types_mod.ListType = lambda x: _Tok()
    types_mod.Array = lambda **k: _Tok()
    sys.modules["numba.types"] = types_mod
    types = types_mod

    class _CT:
        instance_type = _Tok()
    def _jitclass(spec=None):
        def wrap(cls):
            cls.class_type = _CT()
            return cls
        return wrap
    experimental = types_mod.__class__("numba.experimental")
    experimental.jitclass = _jitclass
    sys.modules["numba.experimental"] = experimental

    class _List(list):
        @classmethod
        def empty_list(cls, t): return cls()
    typed = types_mod.__class__("numba.typed")
    typed.List = _List
    sys.modules["numba.typed"] = typed
''')


def import_reference():
    tmp = tempfile.mkdtemp(prefix="lt_golden_")
    os.makedirs(os.path.join(tmp, "numba"))
    with open(os.path.join(tmp, "numba", "__init__.py"), "w") as f:
        f.write(STUB)
    sys.pycache_prefix = os.path.join(tmp, "pycache")  # neither read the shipped .pyc nor write into REF
    sys.dont_write_bytecode = True
    sys.path[:0] = [tmp, REF]
    import importlib
    base = "LightTransportSimulator.light_transport.src."
    return {m: importlib.import_module(base + m) for m in
            ("medium_samples", "intersects", "primitives", "utils", "brdf", "scene", "rays", "material", "bvh_new",
             "constants", "light_samples", "path_tracing_fix1", "path_tracing_old")}


def h4(v, w):
    return np.ascontiguousarray(np.append(np.asarray(v, dtype=np.float64), w))


def main():
    R = import_reference()
    sys.path.insert(0, REPO)
    if sys.argv[1:] == ["g9"]:          # only one fixture; the others are unchanged by it
        g9_render_old(R)
        return
    if sys.argv[1:] == ["g10b"]:
        g10b_obj_meshes_more(R)
        return
    if sys.argv[1:] == ["g10"]:
        g10_obj_meshes(R)
        return
    rs = np.random.RandomState(20240925)
    mat = R["material"].Material(R["material"].Color(np.zeros(3), np.ones(3), np.ones(3)), 1.0, 0.1, 1.5)
    PCT = R["primitives"].PreComputedTriangle

    # ---- G1: henyey_greenstein (medium_samples.py:14-16)
    cos_t = np.linspace(-1.0, 1.0, 201)
    gs = np.array([-0.9, -0.5, -0.1, 0.0, 0.1, 0.5, 0.75, 0.9, 0.99])
    hg = np.stack([R["medium_samples"].henyey_greenstein(cos_t, g) for g in gs])
    np.savez(os.path.join(OUT, "g1_henyey_greenstein.npz"), cos_theta=cos_t, g=gs, value=hg)

    # ---- G2: triangle_intersect (intersects.py:46-104), 10^4 random pairs + edge cases
    n = 10000
    tris = rs.uniform(-2, 2, size=(n, 3, 3))
    org = rs.uniform(-3, 3, size=(n, 3))
    tgt = tris.mean(axis=1) + rs.normal(0, 0.7, size=(n, 3))       # aim near the triangle: ~half hit
    dirs = tgt - org
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    k = 400   # rays parallel to the plane, rays starting on the plane, rays through a vertex / an edge
    e1, e2 = tris[:k, 1] - tris[:k, 0], tris[:k, 2] - tris[:k, 0]
    dirs[:100] = e1[:100] / np.linalg.norm(e1[:100], axis=1, keepdims=True)
    org[100:200] = tris[100:200, 0] + 0.25 * e1[100:200] + 0.25 * e2[100:200]
    d = tris[200:300, 1] - org[200:300]; dirs[200:300] = d / np.linalg.norm(d, axis=1, keepdims=True)
    d = (tris[300:400, 0] + 0.5 * e1[300:400]) - org[300:400]; dirs[300:400] = d / np.linalg.norm(d, axis=1, keepdims=True)
    t_ref = np.full(n, np.nan)
    for i in range(n):
        tri = PCT(h4(tris[i, 0], 1), h4(tris[i, 1], 1), h4(tris[i, 2], 1), mat)
        t = R["intersects"].triangle_intersect(h4(org[i], 1), h4(dirs[i], 0), tri)
        if t is not None:
            t_ref[i] = t
    np.savez(os.path.join(OUT, "g2_triangle_intersect.npz"), tris=tris, origins=org, dirs=dirs, t=t_ref)

    # ---- G3: intersect_bounds (intersects.py:179-196) incl. axis-parallel rays (inv_dir = +-inf)
    lo = rs.uniform(-2, 1, size=(n, 3)); hi = lo + rs.uniform(0.05, 2, size=(n, 3))
    org = rs.uniform(-3, 3, size=(n, 3))
    dirs = rs.normal(size=(n, 3))
    aim = lo + rs.rand(n, 3) * (hi - lo) - org                     # 60 % aimed at a point inside the box
    dirs[4000:] = aim[4000:]
    dirs[:1500, 0] = 0.0; dirs[500:2000, 1] = 0.0; dirs[1000:1500, 2] = 1.0   # 1, 2 zero components
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    org[2000:2300] = 0.5 * (lo[2000:2300] + hi[2000:2300])                     # origin inside the box
    org[2300:2600, 0] = lo[2300:2600, 0]; dirs[2300:2600, 0] = 0.0           # sliding along a face: 0 * inf = NaN lanes
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    tmax = np.where(rs.rand(n) < 0.5, np.inf, rs.uniform(0.1, 6.0, size=n))
    hit = np.zeros(n, dtype=np.int32)
    Ray = R["rays"].Ray
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(n):
            ray = Ray(h4(org[i], 1), h4(dirs[i], 0)); ray.tmax = tmax[i]
            box = R["primitives"].AABB(h4(lo[i], 1), h4(hi[i], 1))
            hit[i] = bool(R["intersects"].intersect_bounds(box, ray, 1 / ray.direction))
    np.savez(os.path.join(OUT, "g3_intersect_bounds.npz"), lo=lo, hi=hi, origins=org, dirs=dirs, tmax=tmax, hit=hit)

    # ---- G4: nearest hit on the hand-restated Cornell box + cone: brute force with the
    # reference's triangle_intersect and its predicate EPSILON < t < min_distance (bvh_new.py:438)
    from light_transport_amd.src import cornell_box as cb, constants as K
    scene = (cb.get_cornell_box(7.5, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(7.5, K.GLASS_MAT)
             + cb.get_light_quad(7.5, K.GLASS_MAT) + cb.get_cone(K.GLASS_MAT))
    verts = np.stack([t.vertices3() for t in scene])
    ref_tris = [PCT(h4(v[0], 1), h4(v[1], 1), h4(v[2], 1), mat) for v in verts]
    nr = 4000
    org = rs.uniform(-7.0, 7.0, size=(nr, 3))
    dirs = rs.normal(size=(nr, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    tmax = np.where(rs.rand(nr) < 0.7, np.inf, rs.uniform(0.5, 10.0, size=nr))
    EPS = R["constants"].EPSILON
    prim = -np.ones(nr, dtype=np.int32); tt = np.full(nr, np.inf)
    for i in range(nr):
        best = tmax[i]
        for j, tri in enumerate(ref_tris):
            t = R["intersects"].triangle_intersect(h4(org[i], 1), h4(dirs[i], 0), tri)
            if t is not None and EPS < t < best:
                best, prim[i] = t, j
        if prim[i] >= 0:
            tt[i] = best
    # the reference's own builder on the same triangles: only its invariants are pinned
    B = R["bvh_new"]
    boxes = [B.BoundedBox(t, i) for i, t in enumerate(ref_tris)]
    root, boxes, ordered, total = B.build_bvh(ref_tris, boxes, 0, len(boxes), [], 0)
    lin, _ = B.flatten_bvh([B.LinearBVHNode() for _ in range(total)], root, 0)
    np.savez(os.path.join(OUT, "g4_scene_nearest_hit.npz"), verts=verts, origins=org, dirs=dirs, tmax=tmax,
             prim=prim, t=tt, ref_total_nodes=np.int64(total),
             ref_leaf_prim_sum=np.int64(sum(nd.n_primitives for nd in lin)))

    # ---- G5: sampling frames (utils.py:72-161, brdf.py:8-9)
    m = 2000
    nrm = rs.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[:6] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=np.float64)
    wi = rs.normal(size=(m, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    u = rs.rand(m, 2); u[:4] = [[0.5, 0.5], [0.0, 0.0], [0.5, 0.25], [0.25, 0.5]]
    onb = np.zeros((m, 6)); dsk = np.zeros((m, 2)); hemi = np.zeros((m, 4)); refl = np.zeros((m, 3))
    U = R["utils"]
    for i in range(m):
        v2, v3 = U.create_orthonormal_system(h4(nrm[i], 0))
        onb[i, :3], onb[i, 3:] = v2, v3
        dsk[i] = U.concentric_sample_disk(u[i])
        dd, pdf = U.cosine_weighted_hemisphere_sampling(h4(nrm[i], 0), h4(wi[i], 0), [u[i, 0], u[i, 1]])
        hemi[i, :3], hemi[i, 3] = dd[:3], pdf
        refl[i] = R["brdf"].get_reflected_direction(h4(wi[i], 0), h4(nrm[i], 0))[:3]
    np.savez(os.path.join(OUT, "g5_sampling.npz"), normals=nrm, incoming=wi, u=u, onb=onb, disk=dsk, cosine_hemi=hemi,
             reflected=refl)

    # ---- G6: PreComputedTriangle / AABB derived fields (primitives.py:75-80, 99-112)
    tv = rs.uniform(-5, 5, size=(500, 3, 3))
    f = np.zeros((500, 13))
    for i in range(500):
        t = PCT(h4(tv[i, 0], 1), h4(tv[i, 1], 1), h4(tv[i, 2], 1), mat)
        f[i] = np.concatenate([t.centroid[:3], t.edge_1[:3], t.edge_2[:3], t.normal[:3], [t.num]])
    bb = R["primitives"].AABB(np.array([-1.0, 2.0, 3.0]), np.array([4.0, 6.0, 5.0]))
    np.savez(os.path.join(OUT, "g6_triangle_fields.npz"), tris=tv, fields=f, aabb_centroid=bb.centroid)

    # ---- G7: Scene table RNG (scene.py:68-69) after np.random.seed(0)
    np.random.seed(0)
    sc = R["scene"].Scene(np.zeros(4), [], width=6, height=5, max_depth=4, f_distance=5, number_of_samples=3)
    np.savez(os.path.join(OUT, "g7_scene_tables.npz"), shape=np.array(sc.rand_0.shape), rand_0=sc.rand_0,
             rand_1=sc.rand_1, image_shape=np.array(sc.image.shape))
    g8_render(R)
    g9_render_old(R)
    g10_obj_meshes(R)
    g10b_obj_meshes_more(R)
    print("golden vectors written to", OUT)


def _g10_chunk(args):
    """Brute-force nearest hit of a chunk of rays over all triangles with the reference's triangle_intersect and its
    predicate EPSILON < t < min_distance (bvh_new.py:438); runs in a forked worker (the reference is already imported)."""
    tri_fn, eps, ref_tris, org, dirs = args
    prim = -np.ones(len(org), dtype=np.int32); tt = np.full(len(org), np.inf); second = np.full(len(org), np.inf)
    for i in range(len(org)):
        o4, d4 = h4(org[i], 1), h4(dirs[i], 0)
        best = np.inf
        for j, tri in enumerate(ref_tris):
            t = tri_fn(o4, d4, tri)
            if t is not None and eps < t:
                if t < best:
                    second[i] = best; best, prim[i] = t, j
                elif t < second[i]:
                    second[i] = t
        if prim[i] >= 0:
            tt[i] = best
    return prim, tt, second


def g10_obj_meshes(R):
    """G10: the reference's on-disk input.  Vertices and faces of examples/obj/{teapot,cow,pumpkin}.obj
    (6320 / 5804 / 10000 triangles: the meshes beyond the LDS budget), parsed HERE by a minimal reader of its own, and
    the brute-force nearest hit of 600 rays per mesh computed with the reference's triangle_intersect on
    PreComputedTriangles built by the reference's own constructor.  `second` = the second-nearest t (ties / shared
    edges can be told apart in the tests).  Numbers only."""
    import multiprocessing as mp
    obj_dir = os.path.join(REF, "LightTransportSimulator", "light_transport", "examples", "obj")
    mat = R["material"].Material(R["material"].Color(np.zeros(3), np.ones(3), np.ones(3)), 1.0, 0.1, 1.5)
    PCT = R["primitives"].PreComputedTriangle
    tri_fn, eps = R["intersects"].triangle_intersect, R["constants"].EPSILON
    out = {}
    rs = np.random.RandomState(1010)
    for name in ("teapot", "cow", "pumpkin"):
        verts, faces = [], []
        for line in open(os.path.join(obj_dir, name + ".obj"), errors="replace"):
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in p[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                faces += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
        v, f = np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32)
        nz = np.array([np.dot(n, n) > 0 for n in np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])])
        f = f[nz]                                             # zero-area faces have no normal: PreComputedTriangle would hold NaNs
        ref_tris = [PCT(h4(v[a], 1), h4(v[b], 1), h4(v[c], 1), mat) for a, b, c in f]
        lo, hi = v.min(axis=0), v.max(axis=0)
        ext = hi - lo
        nr = 600
        org = lo - 0.3 * ext + rs.rand(nr, 3) * 1.6 * ext     # inside and around the bounding box
        tgt = lo + rs.rand(nr, 3) * ext
        dirs = tgt - org; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        chunks = np.array_split(np.arange(nr), 16)
        with mp.get_context("fork").Pool(8) as pool:
            res = pool.map(_g10_chunk, [(tri_fn, eps, ref_tris, org[c], dirs[c]) for c in chunks])
        out[name + "_verts"] = v; out[name + "_faces"] = f
        out[name + "_origins"] = org; out[name + "_dirs"] = dirs
        out[name + "_prim"] = np.concatenate([r[0] for r in res]); out[name + "_t"] = np.concatenate([r[1] for r in res])
        out[name + "_second"] = np.concatenate([r[2] for r in res])
        print(name, len(v), "vertices", len(f), "triangles", int((out[name + "_prim"] >= 0).sum()), "of", nr, "rays hit", flush=True)
    np.savez_compressed(os.path.join(OUT, "g10_obj_meshes.npz"), **out)


def g10b_obj_meshes_more(R):
    """G10b: the REST of the reference's OBJ assets -- examples/obj/{wine-glass,glass,diamond,square}.obj (wine-glass: 12 673
    quad faces `f a//n b//n c//n d//n`, the largest asset; glass: plain quads; diamond: 9 triangles; square: one quad) --
    exactly as g10_obj_meshes does it (own minimal reader, fan triangulation, the reference's PreComputedTriangle and
    triangle_intersect, brute force over every triangle), written to a file of its own with its own random stream so that
    g10_obj_meshes.npz stays byte for byte what it was."""
    import multiprocessing as mp
    obj_dir = os.path.join(REF, "LightTransportSimulator", "light_transport", "examples", "obj")
    mat = R["material"].Material(R["material"].Color(np.zeros(3), np.ones(3), np.ones(3)), 1.0, 0.1, 1.5)
    PCT = R["primitives"].PreComputedTriangle
    tri_fn, eps = R["intersects"].triangle_intersect, R["constants"].EPSILON
    out = {}
    rs = np.random.RandomState(1011)
    for name, nr in (("wine-glass", 600), ("glass", 600), ("diamond", 400), ("square", 200)):
        key = name.replace("-", "_")
        verts, faces = [], []
        for line in open(os.path.join(obj_dir, name + ".obj"), errors="replace"):
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in p[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                faces += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
        v, f = np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32)
        nz = np.array([np.dot(n, n) > 0 for n in np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])])
        f = f[nz]
        ref_tris = [PCT(h4(v[a], 1), h4(v[b], 1), h4(v[c], 1), mat) for a, b, c in f]
        lo, hi = v.min(axis=0), v.max(axis=0)
        ext = np.maximum(hi - lo, 1e-3 * float((hi - lo).max()))      # (square.obj is flat: give the ray cloud some depth)
        org = lo - 0.3 * ext + rs.rand(nr, 3) * 1.6 * ext
        tgt = lo + rs.rand(nr, 3) * (hi - lo)
        dirs = tgt - org; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        chunks = np.array_split(np.arange(nr), 16)
        with mp.get_context("fork").Pool(8) as pool:
            res = pool.map(_g10_chunk, [(tri_fn, eps, ref_tris, org[c], dirs[c]) for c in chunks])
        out[key + "_verts"] = v; out[key + "_faces"] = f
        out[key + "_origins"] = org; out[key + "_dirs"] = dirs
        out[key + "_prim"] = np.concatenate([r[0] for r in res]); out[key + "_t"] = np.concatenate([r[1] for r in res])
        out[key + "_second"] = np.concatenate([r[2] for r in res])
        print(name, len(v), "vertices", len(f), "triangles", int((out[key + "_prim"] >= 0).sum()), "of", nr, "rays hit", flush=True)
    np.savez_compressed(os.path.join(OUT, "g10b_obj_meshes_more.npz"), **out)


def g8_render(R):
    """G8: the reference's working product, path_tracing_fix1.render_scene (:139-169), on the notebook's scene
    (examples/LTS.ipynb cells 11-18: Cornell box half-width 7.5, pv.Cone(radius=2, height=5), 2x2 ceiling light,
    camera (0, 0, depth + 0.5), f_distance = depth), restated by hand without PyVista, 24 x 16 px, 4 spp, D = 8.

    RNG control.  The Scene tables come from np.random.seed(0) as in LTS_fix1.ipynb cell 26.  The one generator
    outside the tables, np.random.choice in cast_one_shadow_ray (light_samples.py:38), is intercepted and served
    from a pre-drawn table light_choice[i, j, s, bounce] -- the same table is an input of the build's renderer.
    Two reference renders are stored per scene: 'asis' (the reference's own intersect_bvh, which mis-skips nodes
    on a small fraction of rays) and 'brute' (module attribute intersect_bvh replaced by a scan over
    all primitives with the reference's triangle_intersect and its predicate EPSILON < t < min_distance).
    """
    import contextlib
    import io
    from light_transport_amd.src import cornell_box as cb
    K, M, PR = R["constants"], R["material"], R["primitives"]
    PCT, Material = PR.PreComputedTriangle, M.Material
    depth = 7.5
    surface = Material(color=K.WHITE_2, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
    left = Material(color=K.RED, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
    right = Material(color=K.GREEN, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
    source_mat = Material(color=K.WHITE, shininess=1, reflection=0.9, ior=1.5, emission=200)
    mirror = Material(color=K.PURPLE, shininess=10, reflection=0.75, ior=1.180, transmission=1.0, is_diffuse=False,
                      is_mirror=True)
    class _M:  # geometry only: my builders are fed placeholder materials, the reference objects get the real ones
        pass
    box = cb.get_cornell_box(depth, "surface", "left", "right")
    cone = cb.get_cone("cone")
    lights_geo = cb.get_light_quad(depth, "source")

    def to_ref(tris, mats, is_light=False):
        out = []
        for t in tris:
            out.append(PCT(h4(t.vertex_1[:3], 1), h4(t.vertex_2[:3], 1), h4(t.vertex_3[:3], 1), mats[t.material], is_light))
        return out

    EPS = K.EPSILON

    def brute(ray, primitives, linear_bvh):
        best, tri = ray.tmax, None
        for p in primitives:
            t = R["intersects"].triangle_intersect(ray.origin, ray.direction, p)
            if t is not None and EPS < t < best:
                best, tri = t, p
        return tri, best

    store = {}
    for name, cone_mat in (("glass", K.GLASS_MAT), ("mirror", mirror)):
        mats = dict(surface=surface, left=left, right=right, cone=cone_mat, source=source_mat)
        objects = to_ref(box, mats) + to_ref(cone, mats) + to_ref(lights_geo, mats, True)
        light_1, light_2 = objects[-2], objects[-1]
        np.random.seed(1)
        lights = R["light_samples"].generate_area_light_samples(light_1, light_2, source_mat, 40, 4)
        B = R["bvh_new"]
        boxes = [B.BoundedBox(o, i) for i, o in enumerate(objects)]
        root, boxes, ordered, total = B.build_bvh(objects, boxes, 0, len(boxes), [], 0)
        lin, _ = B.flatten_bvh([B.LinearBVHNode() for _ in range(total)], root, 0)
        H, W, S, D = 16, 24, 4, 8   # W != H: with W == H the reference's shared x/y jitter (:156-157) puts every
        # anti-diagonal pixel exactly on a box edge (two walls at the same t), an order-dependent tie
        lc = np.random.RandomState(77).randint(0, len(lights), size=(H, W, S, D)).astype(np.int32)

        def fake_choice(n, size=None, _lc=lc):
            f = sys._getframe(2)            # cast_one_shadow_ray <- trace_path
            idx, b = f.f_locals["rand_idx"], f.f_locals["bounce"]
            return np.array([_lc[int(idx[0]), int(idx[1]), int(idx[2]), int(b)]])

        camera = np.array([0, 0, depth + 0.5, 1], dtype=np.float64)
        real_choice = np.random.choice
        PT, U, LS = R["path_tracing_fix1"], R["utils"], R["light_samples"]
        real_bvh = (U.intersect_bvh, LS.intersect_bvh)
        try:
            np.random.choice = fake_choice
            for variant in ("asis", "brute"):
                if variant == "brute":
                    U.intersect_bvh = brute; LS.intersect_bvh = brute
                np.random.seed(0)
                sc = R["scene"].Scene(camera=camera, lights=lights, width=W, height=H, max_depth=D, f_distance=depth,
                                      number_of_samples=S)
                r0, r1 = sc.rand_0.copy(), sc.rand_1.copy()
                with contextlib.redirect_stdout(io.StringIO()):
                    img = PT.render_scene(sc, ordered, lin)
                store["%s_%s_image" % (name, variant)] = img.copy()
                store["%s_%s_rand_0_after" % (name, variant)] = sc.rand_0.copy()
        finally:
            np.random.choice = real_choice
            U.intersect_bvh, LS.intersect_bvh = real_bvh
        store[name + "_verts"] = np.stack([np.stack([o.vertex_1[:3], o.vertex_2[:3], o.vertex_3[:3]]) for o in objects])
        store[name + "_mats"] = np.array([[*o.material.color.diffuse, o.material.emission, o.material.ior,
                                           o.material.transmission, o.material.is_diffuse, o.material.is_mirror,
                                           o.is_light] for o in objects], dtype=np.float64)
        store[name + "_lights"] = np.array([[*l.source[:3], *l.normal[:3],
                                             *(l.material.emission * l.material.color.diffuse), l.total_area]
                                            for l in lights], dtype=np.float64)
        store[name + "_rand_0"], store[name + "_rand_1"], store[name + "_light_choice"] = r0, r1, lc
    store["camera"] = np.array([0, 0, depth + 0.5]); store["f_distance"] = np.float64(depth)
    np.savez_compressed(os.path.join(OUT, "g8_render_fix1.npz"), **store)
    for name in ("glass", "mirror"):
        a, b = store[name + "_asis_image"], store[name + "_brute_image"]
        print("G8 %s: image sum asis %.12f brute %.12f, pixels differing %d" % (
            name, a.sum(), b.sum(), int((np.abs(a - b).max(axis=2) > 1e-12).sum())))


def g9_render_old(R):
    """G9: path_tracing_old.render_scene (:140-171), the recursive integrator examples/LTS.ipynb calls, on G8's scene
    (glass cone and mirror cone), 24 x 16 px, 3 spp, D = 8 (a path can visit up to 2^8 - 1 = 255 surface points).

    RNG control as in G8, except that one path now casts many shadow rays per bounce index: np.random.choice is served
    from light_choice[i, j, s, k], k = running count of shadow rays of that (pixel, sample) in the order the reference
    casts them (depth-first).  Nearest hits: 'brute' only (see G8 for why)."""
    import contextlib
    import io
    from light_transport_amd.src import cornell_box as cb
    K, M, PR = R["constants"], R["material"], R["primitives"]
    PCT, Material = PR.PreComputedTriangle, M.Material
    depth = 7.5
    surface = Material(color=K.WHITE_2, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
    left = Material(color=K.RED, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
    right = Material(color=K.GREEN, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
    source_mat = Material(color=K.WHITE, shininess=1, reflection=0.9, ior=1.5, emission=200)
    mirror = Material(color=K.PURPLE, shininess=10, reflection=0.75, ior=1.180, transmission=1.0, is_diffuse=False,
                      is_mirror=True)
    box, cone, lights_geo = cb.get_cornell_box(depth, "surface", "left", "right"), cb.get_cone("cone"), \
        cb.get_light_quad(depth, "source")

    def to_ref(tris, mats, is_light=False):
        return [PCT(h4(t.vertex_1[:3], 1), h4(t.vertex_2[:3], 1), h4(t.vertex_3[:3], 1), mats[t.material], is_light)
                for t in tris]

    EPS = K.EPSILON

    def brute(ray, primitives, linear_bvh):
        best, tri = ray.tmax, None
        for p in primitives:
            t = R["intersects"].triangle_intersect(ray.origin, ray.direction, p)
            if t is not None and EPS < t < best:
                best, tri = t, p
        return tri, best

    store = {}
    for name, cone_mat in (("glass", K.GLASS_MAT), ("mirror", mirror)):
        mats = dict(surface=surface, left=left, right=right, cone=cone_mat, source=source_mat)
        objects = to_ref(box, mats) + to_ref(cone, mats) + to_ref(lights_geo, mats, True)
        np.random.seed(1)
        lights = R["light_samples"].generate_area_light_samples(objects[-2], objects[-1], source_mat, 40, 4)
        B = R["bvh_new"]
        boxes = [B.BoundedBox(o, i) for i, o in enumerate(objects)]
        root, boxes, ordered, total = B.build_bvh(objects, boxes, 0, len(boxes), [], 0)
        lin, _ = B.flatten_bvh([B.LinearBVHNode() for _ in range(total)], root, 0)
        H, W, S, D = 16, 24, 3, 8
        Q = 16          # >= the most shadow rays any path of this fixture casts (asserted below)
        lc = np.random.RandomState(78).randint(0, len(lights), size=(H, W, S, Q)).astype(np.int32)
        served = np.zeros((H, W, S), dtype=np.int64)

        def fake_choice(n, size=None, _lc=lc, _served=served):
            idx = sys._getframe(2).f_locals["rand_idx"]          # cast_one_shadow_ray <- trace_path
            key = (int(idx[0]), int(idx[1]), int(idx[2]))
            k = _served[key]
            assert k < _lc.shape[3]
            _served[key] += 1
            return np.array([_lc[key + (int(k),)]])

        camera = np.array([0, 0, depth + 0.5, 1], dtype=np.float64)
        real_choice = np.random.choice
        PT, U, LS = R["path_tracing_old"], R["utils"], R["light_samples"]
        real_bvh = (U.intersect_bvh, LS.intersect_bvh)
        try:
            np.random.choice = fake_choice
            U.intersect_bvh = brute; LS.intersect_bvh = brute
            np.random.seed(0)
            sc = R["scene"].Scene(camera=camera, lights=lights, width=W, height=H, max_depth=D, f_distance=depth,
                                  number_of_samples=S)
            r0, r1 = sc.rand_0.copy(), sc.rand_1.copy()
            with contextlib.redirect_stdout(io.StringIO()):
                img = PT.render_scene(sc, ordered, lin)
            store[name + "_image"] = img.copy()
            store[name + "_rand_0_after"] = sc.rand_0.copy()
            store[name + "_shadow_rays"] = served.copy()
        finally:
            np.random.choice = real_choice
            U.intersect_bvh, LS.intersect_bvh = real_bvh
        store[name + "_verts"] = np.stack([np.stack([o.vertex_1[:3], o.vertex_2[:3], o.vertex_3[:3]]) for o in objects])
        store[name + "_mats"] = np.array([[*o.material.color.diffuse, o.material.emission, o.material.ior,
                                           o.material.transmission, o.material.is_diffuse, o.material.is_mirror,
                                           o.is_light] for o in objects], dtype=np.float64)
        store[name + "_lights"] = np.array([[*l.source[:3], *l.normal[:3],
                                             *(l.material.emission * l.material.color.diffuse), l.total_area]
                                            for l in lights], dtype=np.float64)
        store[name + "_rand_0"], store[name + "_rand_1"], store[name + "_light_choice"] = r0, r1, lc
        print("G9 %s: image sum %.12f, shadow rays per path: mean %.2f max %d, non-finite pixels %d" % (
            name, np.nansum(img), served.mean(), served.max(), int((~np.isfinite(img)).sum())))
    store["camera"] = np.array([0, 0, depth + 0.5]); store["f_distance"] = np.float64(depth)
    np.savez_compressed(os.path.join(OUT, "g9_render_old.npz"), **store)


if __name__ == "__main__":
    main()
