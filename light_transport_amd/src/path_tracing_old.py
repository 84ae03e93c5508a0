"""The recursive surface integrator examples/LTS.ipynb calls (role of src/path_tracing_old.py:17-171).

``render_scene(scene, primitives, bvh)`` keeps the reference's call and effects: it consumes ``scene.rand_0`` /
``scene.rand_1`` [H, W, S, D], OVERWRITES ``scene.image`` with ``clip(mean colour)`` (:167), writes the +inf markers
into ``scene.rand_0`` and returns ``scene.image``.  What differs from ``path_tracing_fix1``: a diffuse hit calls
``trace_path(ray, bounce + 1)`` on the shared ray and then continues from the ray the callee left behind (:68-80), so a
path of depth D casts up to 2^D - 1 shadow rays; emission counts at bounce 0 only (:45); the direct term is not
throughput-weighted (:80); roulette starts after bounce 3 (:127).  The recursion runs in the HIP kernel
``k_render_surface_old`` on an explicit per-lane stack (``max_depth`` <= 24).

``light_choice`` [H, W, S, Q]: light sample of the k-th shadow ray of a path, in depth-first order, entry k mod Q
(the reference: ``np.random.choice``, light_samples.py:38); drawn from NumPy's global generator when not given, with
Q = min(2^D - 1, 64).
"""
import numpy as np

from .._lib import default_context, pack_lights, pack_surface_materials
from .bvh_new import linear_bvh_arrays, triangles_array


def render_scene(scene, primitives, bvh, light_choice=None, ctx=None):
    ctx = ctx or default_context()
    none = -np.ones(len(primitives), dtype=np.int32)
    ctx.set_mesh(triangles_array(primitives), none, none, linear_bvh_arrays(bvh))
    ctx._mesh_key = None
    ctx.set_surface_materials(pack_surface_materials(primitives))
    ctx.set_lights(pack_lights(scene.lights))
    H, W, S, D = scene.rand_0.shape
    if light_choice is None:
        light_choice = np.random.randint(0, len(scene.lights), size=(H, W, S, min(2 ** D - 1, 64)))
    xs = np.linspace(scene.left, scene.right, scene.width)    # :142
    ys = np.linspace(scene.top, scene.bottom, scene.height)   # :141
    scene.rand_0 = np.ascontiguousarray(scene.rand_0, dtype=np.float64)
    scene.rand_1 = np.ascontiguousarray(scene.rand_1, dtype=np.float64)
    scene.image = np.ascontiguousarray(scene.image, dtype=np.float64)
    ctx.render_surface(scene.camera, scene.f_distance, xs, ys, scene.rand_0, scene.rand_1, light_choice, scene.image,
                       old=True)
    return scene.image
