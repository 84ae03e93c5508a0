"""CPU suite: host-side mirror of the reference API, and the C-ABI library's
surface (loads, exports every symbol of include/lt.h, fails loudly without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

import light_transport_amd as lt
from light_transport_amd import _lib
from light_transport_amd.src import bvh_new as B
from light_transport_amd.src import constants as K
from light_transport_amd.src import cornell_box as cb
from light_transport_amd.src import photon_tracing as PT
from light_transport_amd.src import primitives as P
from light_transport_amd.src import scene as SC
from light_transport_amd.src.stl4py import partition
from tests import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "lt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    assert os.path.exists(lt.LIB_PATH), "liblt_hip.so not built: run __graft_entry__.build()"
    L = ctypes.CDLL(lt.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert sorted(_lib.SYMBOLS) == names
    assert L.lt_abi_version() == 3


def test_no_cpu_fallback_create_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path is exercised on CPU-only hosts")
    with pytest.raises(lt.LtError) as e:
        lt.Context(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
    with pytest.raises(lt.LtError):
        PT.trace_photons(PT.LayeredSlab([PT.OpticalMedium(0.1, 10, 0.9)], [np.inf]), None, None, 10,
                         grid=PT.VoxelGrid((4, 4, 4), (0, 0, 0), 1.0), source=PT.PencilBeam((0, 0, 0), (0, 0, 1)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "light_transport_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"oracle[/.]", r"liblt_oracle", r"\blto_"):
                    assert not re.search(pat, txt), "product file reaches into oracle/: %s (%s)" % (os.path.join(dp, f), pat)


def test_precomputed_triangle_matches_reference_fields(golden_dir):
    g6 = np.load(os.path.join(golden_dir, "g6_triangle_fields.npz"))
    for i in range(0, 500, 7):
        v = g6["tris"][i]
        t = P.PreComputedTriangle(np.append(v[0], 1.0), np.append(v[1], 1.0), np.append(v[2], 1.0), K.GLASS_MAT)
        got = np.concatenate([t.centroid[:3], t.edge_1[:3], t.edge_2[:3], t.normal[:3], [t.num]])
        np.testing.assert_allclose(got, g6["fields"][i], rtol=1e-13, atol=1e-12)
        assert t.vertex_1.shape == (4,) and t.vertex_1[3] == 1.0 and t.normal[3] == 0.0
        assert t.vertex_1.flags["C_CONTIGUOUS"]
    box = P.AABB(np.array([-1.0, 2.0, 3.0]), np.array([4.0, 6.0, 5.0]))
    np.testing.assert_array_equal(box.centroid, g6["aabb_centroid"])


def test_scene_table_rng_matches_reference(golden_dir):
    g7 = np.load(os.path.join(golden_dir, "g7_scene_tables.npz"))
    np.random.seed(0)
    sc = SC.Scene(np.zeros(4), [], width=6, height=5, max_depth=4, f_distance=5, number_of_samples=3)
    assert tuple(g7["shape"]) == sc.rand_0.shape == (5, 6, 3, 4)
    np.testing.assert_array_equal(sc.rand_0, g7["rand_0"])   # legacy MT19937 stream, same draw order
    np.testing.assert_array_equal(sc.rand_1, g7["rand_1"])
    assert sc.image.shape == tuple(g7["image_shape"])
    tab = PT.uniform_table(7, 5, seed=0)
    assert tab.shape == (7, 5, 4) and tab.min() >= 0 and tab.max() < 1
    np.testing.assert_array_equal(tab.ravel()[:4], np.random.RandomState(0).rand(4))


def test_partition():
    rs = np.random.RandomState(1)
    for _ in range(50):
        n = rs.randint(0, 30)
        a = list(rs.randint(0, 10, size=n))
        lo, hi = sorted(rs.randint(0, n + 1, size=2)) if n else (0, 0)
        b = list(a)
        k = partition(b, lambda x: x < 5, first=lo, last=hi)
        assert sorted(b[lo:hi]) == sorted(a[lo:hi]) and b[:lo] == a[:lo] and b[hi:] == a[hi:]
        assert all(x < 5 for x in b[lo:k]) and all(x >= 5 for x in b[k:hi])


def check_tree(linear, n_prims):
    seen = np.zeros(n_prims, int)

    def walk(i, depth):
        nd = linear[i]
        if nd.n_primitives > 0:
            seen[nd.primitives_offset:nd.primitives_offset + nd.n_primitives] += 1
            return i + 1, depth
        nxt, d0 = walk(i + 1, depth + 1)
        assert nd.second_child_offset == nxt          # B1 fixed: FIRST index of the right subtree
        assert nd.axis in (0, 1, 2)
        end, d1 = walk(nd.second_child_offset, depth + 1)
        return end, max(d0, d1)
    end, depth = walk(0, 0)
    assert end == len(linear) and np.all(seen == 1)
    return depth


@pytest.mark.parametrize("split_method", [0, 1])
def test_build_and_flatten_bvh(split_method):
    dim = 7.5
    prims = (cb.get_cornell_box(dim, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(dim, K.GLASS_MAT)
             + cb.get_light_quad(dim, K.GLASS_MAT) + cb.get_cone(K.GLASS_MAT))
    assert len(prims) == 30 and len(cb.get_cone(K.GLASS_MAT)) == 10   # LTS.ipynb cell 11 -> 10
    boxes = [B.BoundedBox(p, i) for i, p in enumerate(prims)]
    root, boxes, ordered, total = B.build_bvh(prims, boxes, 0, len(boxes), [], 0, split_method)
    linear, used = B.flatten_bvh([B.LinearBVHNode() for _ in range(total)], root, 0)
    assert used == total == len(linear)
    assert sum(n.n_primitives for n in linear) == len(prims)          # LTS.ipynb cell 23
    assert sorted(map(id, ordered)) == sorted(map(id, prims))
    depth = check_tree(linear, len(prims))
    assert depth <= len(prims)        # (no limit in the library: the device traversal is stackless)
    # every node bounds its primitives; children are spatially separated on the split axis (B2 fixed)
    for nd in linear:
        if nd.n_primitives > 0:
            for t in ordered[nd.primitives_offset:nd.primitives_offset + nd.n_primitives]:
                for v in (t.vertex_1, t.vertex_2, t.vertex_3):
                    assert np.all(v[:3] >= nd.bounds.min_point[:3] - 1e-12) and np.all(v[:3] <= nd.bounds.max_point[:3] + 1e-12)
    arr = B.linear_bvh_arrays(linear)
    assert arr["lo"].shape == (total, 3) and (arr["n_prims"] > 0).sum() >= 10


def _rand_tris(rs, n, spread=5.0, size=0.2):
    return [P.PreComputedTriangle(c + rs.normal(0, size, 3), c + rs.normal(0, size, 3), c + rs.normal(0, size, 3), K.GLASS_MAT)
            for c in rs.uniform(-spread, spread, size=(n, 3))]


def test_build_linear_bvh_without_the_library(monkeypatch):
    """A host where liblt_hip.so is not built / cannot load (it is linked against the HIP runtime) can still PREPARE a scene:
    build_linear_bvh falls back to the Python builder, which gives the same tree node for node.  (Computing on it still
    needs the library: there is no CPU fallback for that.)"""
    from light_transport_amd import _lib
    prims = cb.get_cornell_box(7.5, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_cone(K.GLASS_MAT)
    ordered, linear = B.build_linear_bvh(prims, 1)
    want = B.linear_bvh_arrays(linear)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblt_hip.so")
    ordered2, linear2 = B.build_linear_bvh(prims, 1)
    got = B.linear_bvh_arrays(linear2)
    assert [id(p) for p in ordered2] == [id(p) for p in ordered] and len(linear2) == len(linear)
    for k in ("lo", "hi", "offset", "n_prims"):
        assert np.array_equal(got[k], want[k]), k
    with pytest.raises(_lib.LtError):
        _lib.lib()


@pytest.mark.parametrize("split_method", [0, 1])
def test_host_builder_equals_the_python_builder_node_for_node(split_method, golden_dir):
    """lt_build_bvh (C++, what build_linear_bvh uses) against build_bvh + flatten_bvh (the Python mirror of the reference's
    call shapes, which G4 pins): same node count, bit-equal bounds, same offsets / counts / axes, same ordered_prims --
    on the notebook's scene, random triangles, triplicated triangles (coincident centroids: leaves of three), a flat
    sheet (zero extent on one axis), far outliers, and the reference's 6320-triangle teapot."""
    from light_transport_amd.src.io import triangles_from_mesh
    rs = np.random.RandomState(3)
    mk = lambda a, b, c: P.PreComputedTriangle(a, b, c, K.GLASS_MAT)          # noqa: E731
    base = _rand_tris(rs, 40)
    g = np.load(os.path.join(golden_dir, "g10_obj_meshes.npz"))
    cases = {
        "notebook scene": cb.get_cornell_box(7.5, K.GLASS_MAT, K.GLASS_MAT, K.GLASS_MAT) + cb.get_front_wall(7.5, K.GLASS_MAT)
        + cb.get_light_quad(7.5, K.GLASS_MAT) + cb.get_cone(K.GLASS_MAT),
        "random": _rand_tris(rs, 700),
        "triplicates": base + [mk(p.vertex_1[:3], p.vertex_2[:3], p.vertex_3[:3]) for _ in range(2) for p in base],
        "flat sheet": [mk([x, y, 0.0], [x + 0.3, y, 0.0], [x, y + 0.3, 0.0]) for x in np.arange(0, 5, 0.5) for y in np.arange(0, 5, 0.5)],
        "outliers": _rand_tris(rs, 100) + [mk(c + [1e4, 0, 0], c + [1e4, 1, 0], c + [1e4, 0, 1]) for c in rs.uniform(-1, 1, size=(5, 3))],
        "teapot": triangles_from_mesh(g["teapot_verts"], g["teapot_faces"], K.GLASS_MAT, drop_degenerate=False),
    }
    for name, prims in cases.items():
        assert len({id(p) for p in prims}) == len(prims)
        boxes = [B.BoundedBox(p, i) for i, p in enumerate(prims)]
        root, boxes, py_ordered, total = B.build_bvh(prims, boxes, 0, len(boxes), [], 0, split_method)
        py_linear, used = B.flatten_bvh([B.LinearBVHNode() for _ in range(total)], root, 0)
        want = B.linear_bvh_arrays(py_linear)
        ordered, linear = B.build_linear_bvh(prims, split_method)
        got = B.linear_bvh_arrays(linear)
        assert len(linear) == used == total, name
        assert [id(p) for p in ordered] == [id(p) for p in py_ordered], name
        for k in ("lo", "hi", "offset", "n_prims"):
            assert np.array_equal(got[k], want[k]), (name, k)
        interior = want["n_prims"] == 0
        assert np.array_equal(got["axis"][interior], want["axis"][interior]), name
        # the sequence view hands out the reference's node objects
        nd = linear[0]
        assert isinstance(nd, B.LinearBVHNode) and nd.n_primitives == 0 and nd.second_child_offset == want["offset"][0]
        assert np.array_equal(nd.bounds.min_point, want["lo"][0]) and len(list(linear)) == total
        if name == "triplicates":
            assert want["n_prims"].max() == 3


def test_batch_constructor_equals_the_constructor_bit_for_bit(golden_dir):
    """PreComputedTriangle.batch (what load_obj / triangles_from_mesh use: whole-array arithmetic) gives objects whose every
    field -- homogeneous vertices, centroid, edges, unit normal, plane constant -- equals PreComputedTriangle(...)'s (pinned
    by G6) bit for bit, degenerate triangles (NaN normals) included."""
    g = np.load(os.path.join(golden_dir, "g10_obj_meshes.npz"))
    v, f = g["cow_verts"], g["cow_faces"][:1500]
    v1, v2, v3 = v[f[:, 0]].copy(), v[f[:, 1]].copy(), v[f[:, 2]].copy()
    v3[7] = v1[7]; v2[11] = v1[11]; v3[11] = v1[11]            # a sliver with two equal vertices, a point
    got = P.PreComputedTriangle.batch(v1, v2, v3, K.GLASS_MAT)
    assert len(got) == 1500
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(1500):
            want = P.PreComputedTriangle(v1[i], v2[i], v3[i], K.GLASS_MAT)
            for k in ("vertex_1", "vertex_2", "vertex_3", "centroid", "edge_1", "edge_2", "normal"):
                assert np.array_equal(getattr(got[i], k), getattr(want, k), equal_nan=True), (i, k)
            assert got[i].num == want.num and got[i].type == want.type and got[i].material is want.material and got[i].is_light is False
            assert np.array_equal(got[i].vertices3(), want.vertices3())
    assert np.isnan(got[11].normal[:3]).all() and got[7].vertex_1.shape == (4,) and got[7].vertex_1[3] == 1.0


def test_bvh_on_many_random_triangles():
    rs = np.random.RandomState(3)
    prims = [P.PreComputedTriangle(c + rs.normal(0, 0.2, 3), c + rs.normal(0, 0.2, 3), c + rs.normal(0, 0.2, 3), K.GLASS_MAT)
             for c in rs.uniform(-5, 5, size=(300, 3))]
    for sm in (0, 1):
        ordered, linear = B.build_linear_bvh(prims, sm)
        check_tree(linear, 300)


def test_photon_tracing_objects():
    slab = PT.LayeredSlab([PT.OpticalMedium(0.43, 10.7, 0.79, 1.5), PT.OpticalMedium(0.27, 18.7, 0.82, 1.4)], [0.1, np.inf])
    np.testing.assert_array_equal(slab.z_bounds, [0.0, 0.1, np.inf])
    g = PT.VoxelGrid((8, 9, 10), (0, 0, 0), 0.5)
    assert g.voxel == (0.5, 0.5, 0.5) and abs(g.voxel_volume - 0.125) < 1e-15
    # host post-step for an absorbed-weight grid: per-row mu_a of a layered slab whose planes lie on voxel boundaries
    g2 = PT.VoxelGrid((2, 2, 4), (0, 0, 0), (0.5, 0.5, 0.05))
    f = PT.fluence(np.ones((4, 2, 2)), slab, g2, 100)
    np.testing.assert_allclose(f[:2], 1.0 / (0.43 * g2.voxel_volume * 100)); np.testing.assert_allclose(f[2:], 1.0 / (0.27 * g2.voxel_volume * 100))
    with pytest.raises(ValueError):          # a plane inside a voxel row: no per-voxel mu_a -> trace with quantity="fluence"
        PT.fluence(np.ones((4, 2, 2)), slab, PT.VoxelGrid((2, 2, 4), (0, 0, 0), (0.5, 0.5, 0.04)), 100)
    with pytest.raises(TypeError):
        PT.fluence(np.ones((4, 2, 2)), PT.MeshVolume([PT.OpticalMedium(0.1, 1.0, 0.0)]), g2, 100)
    with pytest.raises(ValueError):
        PT.LayeredSlab([], [])
    prob = S.cornell(16)
    assert prob.mesh["verts"].shape == (30, 3, 3) and set(prob.mesh["med_back"]) == {-1, 1}


def test_obj_loader(tmp_path):
    from light_transport_amd.src.io import load_obj, read_obj
    p = tmp_path / "m.obj"
    p.write_text("# comment\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvt 0 0\n"
                 "f 1 2 3 4\nf 1//1 2//1 5//1\nf 1/1/1 3/1/1 5/1/1\nf -5 -4 -1\ns off\nusemtl x\n")
    v, f = read_obj(str(p))
    assert v.shape == (5, 3) and f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4], [0, 2, 4], [0, 1, 4]]
    tris, dim = load_obj(str(p), K.GLASS_MAT, scale=2.0, translate=(1, 0, 0))
    assert len(tris) == 5 and np.allclose(tris[0].vertex_2[:3], [3, 0, 0]) and tris[0].vertex_2[3] == 1.0
    objects, dimension = load_obj(str(p))          # the reference's call shape: (objects, dimension), default material
    assert len(objects) == 5 and dimension == float(v.max()) and objects[0].material.reflection == 0.5
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        read_obj(str(tmp_path / "bad.obj"))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference assets only exist in the build container")
def test_obj_loader_on_reference_assets():
    """Face/vertex counts of the reference's own OBJ files (Assets)."""
    from light_transport_amd.src.io import read_obj
    d = "/root/reference/LightTransportSimulator/light_transport/examples/obj/"
    v, f = read_obj(d + "cube.obj")
    assert len(v) == 8 and len(f) == 12            # 6 quads -> 12 triangles
    v, f = read_obj(d + "teapot.obj")
    assert len(v) == 3644 and len(f) == 6320
    v, f = read_obj(d + "cow.obj")
    assert len(v) == 4583 and len(f) == 5804
    v, f = read_obj(d + "pumpkin.obj")
    assert len(v) == 5002 and len(f) == 10000


def test_job_pipeline_scheduling(monkeypatch):
    """JobPipeline hands jobs to its contexts in turn, completes the oldest job before re-using its context and
    returns results in submission order (logic only: a recording stand-in replaces the device context)."""
    from light_transport_amd import pipeline as PL
    log = []

    class FakeCtx:
        n = 0

        def __init__(self, device_id):
            self.id = FakeCtx.n; FakeCtx.n += 1; self.job = None
        def set_launch_config(self, b, t): log.append(("cfg", self.id, b, t))
        def zero_tally(self): log.append(("zero", self.id))
        def launch(self, n, seed=0, photon_offset=0, f32_walk=False): self.job = (n, seed, photon_offset); log.append(("launch", self.id, seed))
        def sync(self): log.append(("sync", self.id))
        def read_grid(self): return ("grid", self.job)
        def read_grid_raw(self): return ("raw", self.job)
        def read_counters(self): return {"photons": self.job[0]}
        def close(self): log.append(("close", self.id))

    monkeypatch.setattr(PL._lib, "Context", FakeCtx)
    configured = []
    pipe = PL.JobPipeline(lambda c: configured.append(c.id), depth=2)
    assert configured == [0, 1] and [e for e in log if e[0] == "cfg"] == [("cfg", 0, 2, 256), ("cfg", 1, 2, 256)]
    out = list(pipe.run(dict(n_photons=100 + k, seed=k, tag="job%d" % k) for k in range(5)))
    assert [t for t, _, _ in out] == ["job%d" % k for k in range(5)]
    assert [g[1][1] for _, g, _ in out] == [0, 1, 2, 3, 4] and [c["photons"] for _, _, c in out] == [100, 101, 102, 103, 104]
    launches = [e for e in log if e[0] in ("launch", "sync")]
    # contexts alternate; job k-2 is synced right before job k is launched on the same context
    assert launches[:4] == [("launch", 0, 0), ("launch", 1, 1), ("sync", 0), ("launch", 0, 2)]
    assert [e[1] for e in log if e[0] == "launch"] == [0, 1, 0, 1, 0]
    assert pipe.submit(10, seed=9, tag="x") is None and pipe.submit(11, seed=10, tag="y") is None   # both contexts free
    assert [t for t, _, _ in pipe.drain()] == ["x", "y"]
    pipe.close()
    with pytest.raises(ValueError):
        PL.JobPipeline(lambda c: None, depth=0)
    one = PL.JobPipeline(lambda c: None, depth=1, raw=True)     # depth 1: strictly one job at a time, no launch cap
    assert not [e for e in log if e[0] == "cfg" and e[1] == one.ctxs[0].id]
    assert one.submit(5, tag="a") is None and one.submit(6, tag="b")[0] == "a" and one.drain()[0][1][0] == "raw"
