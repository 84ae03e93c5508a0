"""Transport problems used by the tests, described ONCE and applied both to the
CPU oracle (oracle.OracleScene) and to the HIP path (Context.set_*), so both sides
see identical inputs.  C1..C4 follow BASELINE.json's configs / SURVEY.md 8(d)."""
import numpy as np

from light_transport_amd.src import bvh_new as B
from light_transport_amd.src import constants as K
from light_transport_amd.src import cornell_box as cb
from oracle import oracle as O


class Problem:
    def __init__(self, media, grid_shape, origin, voxel, layers=None, mesh=None, source=None, max_steps=1000000):
        self.media, self.grid_shape, self.origin, self.voxel = media, grid_shape, origin, voxel
        self.layers, self.mesh, self.max_steps = layers, mesh, max_steps
        self.source = source or dict(type=0, pos=(0.0, 0.0, 0.0), dir=(0.0, 0.0, 1.0), extra=(0.0,) * 6, start_medium=0)

    def oracle(self):
        return O.OracleScene(self.media, self.grid_shape, self.origin, self.voxel, layers=self.layers, mesh=self.mesh,
                             source=self.source, max_steps=self.max_steps)

    def apply(self, ctx, dtype="f64"):
        ctx.set_media(self.media)
        if self.layers is not None:
            ctx.set_layers(self.layers["z_bounds"], self.layers["medium_idx"], self.layers.get("n_above", 1.0),
                           self.layers.get("n_below", 1.0))
        else:
            ctx.set_mesh(self.mesh["verts"], self.mesh["med_front"], self.mesh["med_back"], self.mesh["nodes"])
        ctx.set_grid(self.grid_shape, self.origin, self.voxel, dtype)
        s = self.source
        ctx.set_source(s.get("type", 0), s["pos"], s["dir"], s.get("extra"), s.get("start_medium", 0))
        ctx.set_max_steps(self.max_steps)
        ctx.set_tally_quantity("absorbed")
        return ctx


def slab(n=64, voxel=0.4, media=((0.1, 10.0, 0.9, 1.0),), thickness=np.inf, n_above=1.0, n_below=1.0, **kw):
    """C1 (n=64, voxel 0.4) / C2 (n=256, voxel 0.1): homogeneous semi-infinite slab, pencil beam."""
    half = n * voxel / 2
    return Problem(list(media), (n, n, n), (-half, -half, 0.0), (voxel,) * 3,
                   layers=dict(z_bounds=[0.0, thickness], medium_idx=[0], n_above=n_above, n_below=n_below), **kw)


def two_layer(n=64, voxel=0.2, **kw):
    """C3: epidermis 0-0.1 mm over dermis, ambient n = 1 (values of SURVEY.md 8(d))."""
    half = n * voxel / 2
    media = [(0.43, 10.7, 0.79, 1.5), (0.27, 18.7, 0.82, 1.4)]
    return Problem(media, (n, n, n), (-half, -half, 0.0), (voxel,) * 3,
                   layers=dict(z_bounds=[0.0, 0.1, np.inf], medium_idx=[0, 1], n_above=1.0, n_below=1.0), **kw)


def cornell_scene(split_method=1):
    """The 30-triangle Cornell cavity + cone of config 4, in BVH order, with media labels."""
    dim = 7.5
    walls = (cb.get_cornell_box(dim, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(dim, K.GLASS_MAT)
             + cb.get_light_quad(dim, K.GLASS_MAT))
    cone = cb.get_cone(K.GLASS_MAT)
    for t in walls:   # normals point into the cavity: front = cavity medium, back = exterior
        t.med_front, t.med_back = 0, -1
    for t in cone:    # normals point out of the cone: front = cavity medium, back = cone medium
        t.med_front, t.med_back = 0, 1
    ordered, linear = B.build_linear_bvh(walls + cone, split_method)
    return ordered, linear


def cornell(n=64, **kw):
    """C4: C1's medium fills the cube, a cone of a second medium sits in it, cosine source on the ceiling quad."""
    dim = 7.5
    ordered, linear = cornell_scene()
    mesh = dict(verts=B.triangles_array(ordered), med_front=np.array([t.med_front for t in ordered], np.int32),
                med_back=np.array([t.med_back for t in ordered], np.int32), nodes=B.linear_bvh_arrays(linear))
    media = [(0.1, 10.0, 0.9, 1.0), (1.0, 5.0, 0.8, 1.5)]
    src = dict(type=1, pos=(-1.0, dim, -1.0), dir=(0.0, -1.0, 0.0), extra=(2.0, 0.0, 0.0, 0.0, 0.0, 2.0), start_medium=0)
    voxel = 2 * dim / n
    return Problem(media, (n, n, n), (-dim, -dim, -dim), (voxel,) * 3, mesh=mesh, source=src, **kw)


def assert_grid_close(g, go, rtol=1e-9, atol=1e-12, max_bad=0):
    d = np.abs(g - go)
    bad = int((d > atol + rtol * np.abs(go)).sum())
    assert bad <= max_bad, "%d voxels outside |d| <= %g + %g*E (max abs %g)" % (bad, atol, rtol, d.max())


def icosphere(subdiv=3, radius=1.0, center=(0.0, 0.0, 0.0)):
    """(vertices, faces) of a subdivided icosahedron, outward-facing: 20 * 4^subdiv triangles."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v) * radius + np.asarray(center, dtype=np.float64), np.array(f, dtype=np.int64)


def sphere_in_box(subdiv=4, n=48, split_method=0, **kw):
    """f1 scene: a finely tessellated sphere (20 * 4^subdiv triangles; 5120 at subdiv 4 -- beyond the LDS
    budget, so the walk traverses it in global memory) of an absorbing medium inside a closed box."""
    from light_transport_amd.src.io import triangles_from_mesh
    dim = 4.0
    walls = cb.get_cornell_box(dim, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(dim, K.GLASS_MAT) \
        + cb.get_light_quad(dim, K.GLASS_MAT)
    for t in walls:
        t.med_front, t.med_back = 0, -1
    vs, fs = icosphere(subdiv, 1.5, (0.3, -0.2, 0.1))
    ball = triangles_from_mesh(vs, fs, K.GLASS_MAT)
    for t in ball:
        t.med_front, t.med_back = 0, 1
    ordered, linear = B.build_linear_bvh(walls + ball, split_method)
    mesh = dict(verts=B.triangles_array(ordered), med_front=np.array([t.med_front for t in ordered], np.int32),
                med_back=np.array([t.med_back for t in ordered], np.int32), nodes=B.linear_bvh_arrays(linear))
    media = [(0.05, 5.0, 0.8, 1.0), (0.8, 8.0, 0.9, 1.37)]
    src = dict(type=1, pos=(-1.0, dim, -1.0), dir=(0.0, -1.0, 0.0), extra=(2.0, 0.0, 0.0, 0.0, 0.0, 2.0), start_medium=0)
    voxel = 2 * dim / n
    return Problem(media, (n, n, n), (-dim, -dim, -dim), (voxel,) * 3, mesh=mesh, source=src, **kw), ordered, linear


def open_sheets(n=24, **kw):
    """An OPEN mesh beyond the LDS budget: a bumpy 40 x 40 sheet (3200 triangles) over an exactly flat 30 x 30 one (1800), no box
    around them -- photons roam all around the mesh, outside its bounding box too, and the flat sheet has zero extent on z.
    Medium 1 between the sheets, exterior below the flat one; a handful of zero-area triangles are mixed in."""
    from light_transport_amd.src.io import triangles_from_mesh

    def sheet(m, half, zf):
        xs = np.linspace(-half, half, m + 1)
        X, Y = np.meshgrid(xs, xs, indexing="ij")
        V = np.stack([X, Y, zf(X, Y)], axis=-1).reshape(-1, 3)
        idx = lambda i, j: i * (m + 1) + j      # noqa: E731
        F = []
        for i in range(m):
            for j in range(m):      # counter-clockwise seen from +z: normals point up
                F += [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)), (idx(i, j), idx(i + 1, j + 1), idx(i, j + 1))]
        return V, np.array(F)
    v1, f1 = sheet(40, 4.0, lambda X, Y: 0.3 * np.sin(1.3 * X) * np.cos(0.9 * Y))
    v2, f2 = sheet(30, 3.0, lambda X, Y: np.full_like(X, -2.0))
    bumpy = triangles_from_mesh(v1, f1, K.GLASS_MAT, drop_degenerate=False)
    flat = triangles_from_mesh(v2, f2, K.GLASS_MAT, drop_degenerate=False)
    deg = triangles_from_mesh(np.array([[0.5, 0.5, 1.0], [1.5, 1.5, 1.0], [1.0, 1.0, 1.0], [2.0, 0.0, -1.0]]),
                              np.array([(0, 1, 2), (3, 3, 3), (0, 0, 1)]), K.GLASS_MAT, drop_degenerate=False)      # collinear / coincident vertices
    for t in bumpy + deg:
        t.med_front, t.med_back = 0, 1
    for t in flat:
        t.med_front, t.med_back = 1, -1
    ordered, linear = B.build_linear_bvh(bumpy + flat + deg, 0)
    mesh = dict(verts=B.triangles_array(ordered), med_front=np.array([t.med_front for t in ordered], np.int32),
                med_back=np.array([t.med_back for t in ordered], np.int32), nodes=B.linear_bvh_arrays(linear))
    media = [(0.05, 5.0, 0.8, 1.0), (0.8, 8.0, 0.9, 1.37)]
    src = dict(type=0, pos=(0.1, -0.2, 3.0), dir=(0.05, 0.02, -1.0), extra=(0.0,) * 6, start_medium=0)
    return Problem(media, (n, n, n), (-6.0, -6.0, -6.0), (12.0 / n,) * 3, mesh=mesh, source=src, **kw), ordered, linear


def chain_mesh(T, seed=5):
    """T small triangles strung along x under a flattened tree that is a CHAIN: every interior node splits one triangle off
    (pre-order: interior i at 2i, its leaf at 2i + 1, the rest behind it) -- depth T - 1.  Returns (verts [T, 3, 3], nodes)."""
    rs = np.random.RandomState(seed)
    verts = np.zeros((T, 3, 3))
    for k in range(T):
        verts[k] = np.array([0.3 * k, 0.0, 0.0]) + rs.uniform(-0.12, 0.12, size=(3, 3))
    lo, hi = verts.min(axis=1), verts.max(axis=1)
    N = 2 * T - 1
    nodes = dict(lo=np.zeros((N, 3)), hi=np.zeros((N, 3)), offset=np.zeros(N, np.int32), n_prims=np.zeros(N, np.int32),
                 axis=np.zeros(N, np.int32))
    for i in range(T - 1):
        nodes["lo"][2 * i], nodes["hi"][2 * i] = lo[i:].min(axis=0), hi[i:].max(axis=0)
        nodes["offset"][2 * i] = 2 * i + 2                                  # second child
        nodes["lo"][2 * i + 1], nodes["hi"][2 * i + 1] = lo[i], hi[i]
        nodes["offset"][2 * i + 1], nodes["n_prims"][2 * i + 1] = i, 1
    nodes["lo"][N - 1], nodes["hi"][N - 1] = lo[T - 1], hi[T - 1]
    nodes["offset"][N - 1], nodes["n_prims"][N - 1] = T - 1, 1
    return verts, nodes


def chain_rays(verts, n, seed=6):
    """Rays aimed at (or just past) random points of random triangles of a chain_mesh, from origins around the chain: most
    hit, many cross the boxes of several links, a quarter run nearly along the chain's axis (every level is entered)."""
    rs = np.random.RandomState(seed)
    T = len(verts)
    o = rs.uniform(-1, 0.3 * T + 1, size=(n, 3)) * [1, 0, 0] + rs.uniform(-1, 1, size=(n, 3)) * [0, 1, 1]
    o[: n // 4, 1:] *= 0.05
    k = rs.randint(0, T, n)
    bary = rs.dirichlet([1, 1, 1], n)
    tgt = np.einsum("nk,nkc->nc", bary, verts[k]) + rs.normal(0, 0.03, size=(n, 3))
    d = tgt - o
    return o, d / np.linalg.norm(d, axis=1, keepdims=True), k


def g8_inputs(g8, name):
    """Build the mesh (my BVH builder over the fixture's triangles) and tables of a G8 render."""
    from light_transport_amd.src.io import triangles_from_mesh
    verts = g8[name + "_verts"]
    tris = triangles_from_mesh(verts.reshape(-1, 3), np.arange(len(verts) * 3).reshape(-1, 3), K.GLASS_MAT,
                               drop_degenerate=False)
    for k, t in enumerate(tris):
        t.fixture_index = k
    ordered, linear = B.build_linear_bvh(tris)
    order = np.array([t.fixture_index for t in ordered])
    mesh = dict(verts=B.triangles_array(ordered), med_front=-np.ones(len(ordered), np.int32),
                med_back=-np.ones(len(ordered), np.int32), nodes=B.linear_bvh_arrays(linear))
    H, W, S, D = g8[name + "_rand_0"].shape
    return dict(mesh=mesh, mats=g8[name + "_mats"][order], lights=g8[name + "_lights"], camera=g8["camera"],
                f_distance=float(g8["f_distance"]), xs=np.linspace(-1, 1, W),
                ys=np.linspace(1 / (W / H), -1 / (W / H), H),   # Scene.top / bottom, scene.py:60-63
                rand_0=g8[name + "_rand_0"], rand_1=g8[name + "_rand_1"], light_choice=g8[name + "_light_choice"],
                shape=(H, W, S, D))


def check_g8_image(img, rand_0_after, g8, name):
    """Exact-arithmetic parity target: the reference render with a correct nearest hit ('brute').  Against the
    reference as shipped ('asis') only the few pixels reached through its BVH bug B3 may differ."""
    ref = g8[name + "_brute_image"]
    np.testing.assert_allclose(img, ref, rtol=1e-9, atol=1e-12)
    marks = np.isinf(g8[name + "_brute_rand_0_after"])
    assert np.array_equal(np.isinf(rand_0_after), marks) and marks.sum() > 100
    asis = g8[name + "_asis_image"]
    bad = int((np.abs(img - asis).max(axis=2) > 1e-9).sum())
    assert bad <= 8, "%d pixels differ from the reference-as-shipped render" % bad
    assert img.sum() > 10 and (img.max(axis=2) > 0.2).sum() > 50


def check_g9_image(img, rand_0_after, g9, name):
    """path_tracing_old.render_scene as the reference ran it (correct nearest hit), incl. the markers in rand_0."""
    np.testing.assert_allclose(img, g9[name + "_image"], rtol=1e-9, atol=1e-12)
    marks = np.isinf(g9[name + "_rand_0_after"])
    assert np.array_equal(np.isinf(rand_0_after), marks) and marks.sum() > 100
    assert g9[name + "_shadow_rays"].max() > 4 and img.sum() > 10   # the recursion really branched


def obj_in_box(verts, faces, n=40, split_method=0, half=4.0, **kw):
    """f3 scene: a mesh that came through the OBJ loader (G10: teapot / cow / pumpkin of the reference's assets),
    scaled to fit and centred in a closed box, SAH-built; an absorbing medium on the back side of its triangles.
    Returns (problem, ordered triangles, linear BVH, index of every ordered mesh triangle in `faces` or -1 for walls)."""
    from light_transport_amd.src.io import triangles_from_mesh
    v = np.asarray(verts, dtype=np.float64)
    lo, hi = v.min(axis=0), v.max(axis=0)
    scale = 0.6 * 2 * half / float((hi - lo).max())
    shift = -0.5 * (lo + hi) * scale
    walls = cb.get_cornell_box(half, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(half, K.GLASS_MAT) \
        + cb.get_light_quad(half, K.GLASS_MAT)
    for t in walls:
        t.med_front, t.med_back, t.face_index = 0, -1, -1
    body = triangles_from_mesh(v, faces, K.GLASS_MAT, scale, shift, drop_degenerate=False)
    for k, t in enumerate(body):
        t.med_front, t.med_back, t.face_index = 0, 1, k
    ordered, linear = B.build_linear_bvh(walls + body, split_method)
    mesh = dict(verts=B.triangles_array(ordered), med_front=np.array([t.med_front for t in ordered], np.int32),
                med_back=np.array([t.med_back for t in ordered], np.int32), nodes=B.linear_bvh_arrays(linear))
    media = [(0.05, 5.0, 0.8, 1.0), (0.8, 8.0, 0.9, 1.37)]
    src = dict(type=1, pos=(-1.0, half, -1.0), dir=(0.0, -1.0, 0.0), extra=(2.0, 0.0, 0.0, 0.0, 0.0, 2.0), start_medium=0)
    voxel = 2 * half / n
    prob = Problem(media, (n, n, n), (-half, -half, -half), (voxel,) * 3, mesh=mesh, source=src, **kw)
    return prob, ordered, linear, np.array([t.face_index for t in ordered]), scale, shift


def share_an_edge(tri_a, tri_b, tol=1e-9):
    """Two triangles ([3, 3] vertex arrays) with at least two common vertices."""
    common = sum(1 for p in tri_a if np.any(np.all(np.abs(tri_b - p) <= tol * (1.0 + np.abs(p).max()), axis=1)))
    return common >= 2


def check_hits_against_fixture(prim, t, fix_prim, fix_t, fix_second, tris, rtol=1e-12):
    """Nearest hits (triangle index into `tris`, distance) against a brute-force fixture.  Index and decision work is
    exact: a differing triangle is accepted ONLY as a tie -- the same distance to rtol on two triangles that share an
    edge (the ray passes through the common edge; which of the two claims it is decided in the last bit)."""
    assert np.array_equal(prim >= 0, fix_prim >= 0), "hit / miss decisions differ"
    hit = fix_prim >= 0
    np.testing.assert_allclose(t[hit], fix_t[hit], rtol=rtol)
    diff = np.flatnonzero(hit & (prim != fix_prim))
    for i in diff:
        assert abs(fix_second[i] - fix_t[i]) <= 1e-9 * fix_t[i], "ray %d: another triangle although the hit is not a tie" % i
        assert share_an_edge(tris[prim[i]], tris[fix_prim[i]]), "ray %d: tie between triangles that share no edge" % i
    return len(diff)
