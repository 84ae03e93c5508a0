"""Context number for f2: the reference's LTS_fix1 notebook render (300 x 300 px, 50 spp, depth 8; 149.45 s incl. Numba
JIT on the authors' machine) through render_scene -> k_render_surface."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from light_transport_amd.src import constants as K, cornell_box as cb, bvh_new as B
from light_transport_amd.src.material import Material, Color
from light_transport_amd.src.light_samples import generate_area_light_samples
from light_transport_amd.src.path_tracing_fix1 import render_scene
from light_transport_amd.src.path_tracing_old import render_scene as render_scene_old
from light_transport_amd.src.scene import Scene
depth = 7.5
def col(d): return Color(np.zeros(3), np.array(d, dtype=np.float64), np.ones(3))
surf = Material(color=col([0.55, 0.55, 0.55]), shininess=30, reflection=0.1, ior=1.521, transmission=1)
left = Material(color=col([0.7, 0, 0]), shininess=30, reflection=0.1, ior=1.521, transmission=1)
right = Material(color=col([0, 0.6, 0]), shininess=30, reflection=0.1, ior=1.521, transmission=1)
src = Material(color=K.WHITE, shininess=1, reflection=0.9, ior=1.5, emission=200)
lq = cb.get_light_quad(depth, src)
objects = cb.get_cornell_box(depth, surf, left, right) + cb.get_cone(K.GLASS_MAT) + lq
np.random.seed(1)
lights = generate_area_light_samples(lq[0], lq[1], src, 1000, 4)
ordered, linear = B.build_linear_bvh(objects)
ctx = lt.Context(0)
for (w, h, s, d, fn, what) in ((150, 150, 100, 4, render_scene_old, "path_tracing_old (LTS.ipynb cell 36)"), (150, 150, 100, 4, render_scene, "path_tracing_fix1"),
                               (300, 300, 50, 8, render_scene, "path_tracing_fix1 (LTS_fix1.ipynb cell 26)")):
    np.random.seed(0)
    t0 = time.time()
    sc = Scene(camera=np.array([0, 0, depth + 0.5, 1.0]), lights=lights, width=w, height=h, max_depth=d, f_distance=depth,
               number_of_samples=s)
    t1 = time.time()
    img = fn(sc, ordered, linear, ctx=ctx)
    t2 = time.time()
    print("%-44s %dx%d, %d spp, depth %d: tables %.2f s, render_scene wall %.3f s (kernel %.1f ms, %.2e paths/s), image mean %.4f" % (
        what, w, h, s, d, t1 - t0, t2 - t1, ctx.last_kernel_ms(), w * h * s / (ctx.last_kernel_ms() * 1e-3), img.mean()), flush=True)
