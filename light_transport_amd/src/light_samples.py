"""Area-light point samples (role of src/light_samples.py:17-32).

``generate_area_light_samples(tri_1, tri_2, source_mat, number_of_samples, total_area)``
draws, like the reference, ``a = uniform(0, 1, n)`` then ``b = uniform(1, 0, n)`` from
NumPy's global generator and maps them onto the first triangle; both lights of a
pair share that point and differ in the normal only (the reference's quirk B7:
the second triangle's point is computed and discarded, :26-29) -- kept as is so
that a seeded run produces the reference's light list.  The shadow-ray estimator
``cast_one_shadow_ray`` (:36-61) runs inside the render kernel.
"""
import numpy as np

from .scene import Light


def generate_area_light_samples(tri_1, tri_2, source_mat, number_of_samples, total_area):
    n = int(number_of_samples)
    a = np.random.uniform(0, 1, n)
    b = np.random.uniform(1, 0, n)
    ra = np.sqrt(a)
    pts = (tri_1.vertex_1[None, :] * (1 - ra)[:, None] + tri_1.vertex_2[None, :] * (ra * (1 - b))[:, None]
           + tri_1.vertex_3[None, :] * (b * ra)[:, None])
    lights = []
    for p in pts:
        lights.append(Light(source=p, material=source_mat, normal=tri_1.normal, total_area=total_area))
        lights.append(Light(source=p, material=source_mat, normal=tri_2.normal, total_area=total_area))
    return lights
