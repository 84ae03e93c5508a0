"""Surface path tracer launcher (role of src/path_tracing_fix1.py:139-169).

``render_scene(scene, primitives, bvh)`` keeps the reference's call: it consumes
the ``Scene`` tables (``rand_0`` / ``rand_1`` [H, W, S, D]), accumulates
``0.25 * clip(mean colour)`` into ``scene.image``, writes the +inf markers for unused
bounces into ``scene.rand_0`` and returns ``scene.image``.  ``trace_path`` (:18-136)
itself -- nearest hit, emission, one shadow ray, cosine bounce / mirror / glass
branch, Russian roulette -- runs in the HIP kernel ``k_render_surface``, one lane
per pixel.

The one generator the reference leaves outside its tables is the light pick of
``cast_one_shadow_ray`` (``np.random.choice``, light_samples.py:38); here it is a table
too: ``light_choice`` [H, W, S, D] (drawn from NumPy's global generator when not
given), which makes a render a pure function of its inputs.
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
        light_choice = np.random.randint(0, len(scene.lights), size=(H, W, S, D))
    xs = np.linspace(scene.left, scene.right, scene.width)    # :142
    ys = np.linspace(scene.top, scene.bottom, scene.height)   # :141
    scene.rand_0 = np.ascontiguousarray(scene.rand_0, dtype=np.float64)
    scene.rand_1 = np.ascontiguousarray(scene.rand_1, dtype=np.float64)
    scene.image = np.ascontiguousarray(scene.image, dtype=np.float64)
    ctx.render_surface(scene.camera, scene.f_distance, xs, ys, scene.rand_0, scene.rand_1, light_choice, scene.image)
    return scene.image
