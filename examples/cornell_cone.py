#!/usr/bin/env python3
"""The reference's Cornell-box + cone scene (examples/LTS.ipynb cells 11-22) twice on the GPU:
  1. the reference's own product -- render_scene (surface path tracer) -> image [H, W, 3];
  2. the volumetric photon walk the reference left empty -- a scattering medium fills the cavity, the cone is a second
     medium, photons start on the ceiling light -> absorbed energy [nz, ny, nx].

    python examples/cornell_cone.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from light_transport_amd.src import constants as K
from light_transport_amd.src.bvh_new import BoundedBox, LinearBVHNode, build_bvh, flatten_bvh
from light_transport_amd.src.cornell_box import get_cone, get_cornell_box, get_front_wall, get_light_quad
from light_transport_amd.src.light_samples import generate_area_light_samples
from light_transport_amd.src.material import Material
from light_transport_amd.src.path_tracing_fix1 import render_scene
from light_transport_amd.src.photon_tracing import (AreaLight, MeshVolume, OpticalMedium, VoxelGrid, set_triangle_media,
                                                    trace_photons)
from light_transport_amd.src.scene import Scene

depth = 7.5


surface = Material(color=K.WHITE_2, shininess=30, reflection=0.1, ior=1.5210, transmission=1)     # notebook cell 13
left = Material(color=K.RED, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
right = Material(color=K.GREEN, shininess=30, reflection=0.1, ior=1.5210, transmission=1)
source = Material(color=K.WHITE, shininess=1, reflection=0.9, ior=1.5, emission=200)


def bvh_of(objects):                      # notebook cells 19-22
    boxes = [BoundedBox(o, i) for i, o in enumerate(objects)]
    root, boxes, ordered, total = build_bvh(objects, boxes, 0, len(boxes), [], 0)
    linear, _ = flatten_bvh([LinearBVHNode() for _ in range(total)], root, 0)
    return ordered, linear


# ---- 1. surface render
light_tris = get_light_quad(depth, source)
objects = get_cornell_box(depth, surface, left, right) + get_cone(K.GLASS_MAT) + light_tris
lights = generate_area_light_samples(light_tris[0], light_tris[1], source, 1000, 4)
primitives, linear_bvh = bvh_of(objects)
np.random.seed(0)
scene = Scene(camera=np.array([0, 0, depth + 0.5, 1.0]), lights=lights, width=150, height=150, max_depth=4,
              f_distance=depth, number_of_samples=12)
image = render_scene(scene, primitives, linear_bvh)
print("render_scene: image %s, mean %.4f, max %.4f" % (image.shape, image.mean(), image.max()))

# ---- 2. photon transport in the same geometry (closed cavity)
walls = set_triangle_media(get_cornell_box(depth, surface, left, right) + get_front_wall(depth, surface)
                           + get_light_quad(depth, source), front=0, back=-1)      # normals point into the cavity
cone = set_triangle_media(get_cone(K.GLASS_MAT), front=0, back=1)                  # normals point out of the cone
primitives, linear_bvh = bvh_of(walls + cone)
volume = MeshVolume([OpticalMedium(0.1, 10.0, 0.9, 1.0), OpticalMedium(1.0, 5.0, 0.8, K.GLASS_MAT.ior)], start_medium=0)
grid = VoxelGrid((128, 128, 128), origin=(-depth,) * 3, voxel=2 * depth / 128)
lamp = AreaLight(corner=(-1, depth, -1), edge_1=(2, 0, 0), edge_2=(0, 0, 2), normal=(0, -1, 0))
dose, counters = trace_photons(volume, primitives, linear_bvh, 2_000_000, seed=1, grid=grid, source=lamp,
                               return_counters=True)
print("trace_photons: absorbed grid %s, absorbed fraction %.3f, left through the walls %.3f, %.0f steps per photon"
      % (dose.shape, counters["w_absorbed"] / 2e6, counters["w_escaped_mesh"] / 2e6, counters["steps"] / 2e6))
