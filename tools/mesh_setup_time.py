"""Wall clock of the host path of a real mesh: OBJ text -> load_obj -> build_linear_bvh -> lt_set_mesh -> first launch (which
builds the march grid: candidate lists on the host, clearances on the device).   python tools/mesh_setup_time.py [mesh]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from light_transport_amd.src import bvh_new as B, constants as K
from light_transport_amd.src.io import load_obj
name = sys.argv[1] if len(sys.argv) > 1 else "pumpkin"
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                         "g10b_obj_meshes_more.npz" if name in ("wine_glass", "glass") else "g10_obj_meshes.npz"))
v, f = g[name + "_verts"], g[name + "_faces"]
path = os.path.join(tempfile.mkdtemp(), name + ".obj")
with open(path, "w") as fh:
    for p in v: fh.write("v %r %r %r\n" % tuple(float(x) for x in p))
    for a, b, c in f: fh.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
ctx = lt.Context(0)
ctx.set_media([(0.05, 5.0, 0.8, 1.0), (0.8, 8.0, 0.9, 1.37)])
ctx.set_grid((32, 32, 32), tuple(v.min(axis=0) - 0.1), tuple((v.max(axis=0) - v.min(axis=0) + 0.2) / 32), "f64")
ctx.set_source(0, tuple(v.mean(axis=0)), (0.0, 0.0, 1.0), None, 0)
for rep in range(3):
    t0 = time.perf_counter()
    objects, dim = load_obj(path, K.GLASS_MAT); t1 = time.perf_counter()
    ordered, linear = B.build_linear_bvh(objects, 0); t2 = time.perf_counter()
    ctx.set_mesh(B.triangles_array(ordered), np.zeros(len(ordered), np.int32), np.ones(len(ordered), np.int32), linear.records); t3 = time.perf_counter()
    ctx.launch(1024, seed=rep); ctx.sync(); t4 = time.perf_counter()
    print("%s %d triangles: load_obj %.3f  build_linear_bvh %.3f  set_mesh %.3f  first launch (tables + march grid) %.3f  TOTAL %.3f s  %s" % (
        name, len(ordered), t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, ctx.mesh_accel_info()), flush=True)
ctx.close()
