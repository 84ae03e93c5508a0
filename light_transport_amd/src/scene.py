"""Launch configuration records mirroring src/scene.py (Light :12-17, Camera
:24-27, Scene :53-73).  ``Scene`` keeps the reference's pre-drawn uniform tables
(``rand_0`` / ``rand_1`` of shape [H, W, S, D], :68-69): the "table RNG" that the
photon walk's parity mode re-uses (see photon_tracing.uniform_table)."""
import numpy as np


class Light:
    def __init__(self, source, material, normal, total_area):
        self.source = np.asarray(source, dtype=np.float64)
        self.material = material
        self.normal = np.asarray(normal, dtype=np.float64)
        self.total_area = float(total_area)


class Camera:
    def __init__(self, position, focal_length):
        self.position = np.asarray(position, dtype=np.float64)
        self.focal_length = int(focal_length)


class Scene:
    def __init__(self, camera, lights, width=400, height=400, max_depth=3, f_distance=5, number_of_samples=8):
        self.camera = np.asarray(camera, dtype=np.float64)
        self.lights = lights
        self.width, self.height = int(width), int(height)
        self.max_depth = int(max_depth)
        self.aspect_ratio = width / height
        self.left, self.right = -1, 1
        self.top, self.bottom = 1 / self.aspect_ratio, -1 / self.aspect_ratio
        self.f_distance = f_distance
        self.number_of_samples = int(number_of_samples)
        shape = (self.height, self.width, self.number_of_samples, self.max_depth)
        self.image = np.zeros((self.height, self.width, 3), dtype=np.float64)
        # drawn in the reference's order so a seeded global NumPy RNG gives the same tables
        self.rand_0 = np.random.rand(*shape)
        self.rand_1 = np.random.rand(*shape)
        with np.errstate(divide="ignore"):
            self.rand_0_logit = np.log(self.rand_0 / (1 - self.rand_0))
            self.rand_1_logit = np.log(self.rand_1 / (1 - self.rand_1))
        self.bounce_record = np.ones(shape, dtype=np.int8)
        self.record_log_pdf = np.zeros(shape, dtype=np.float64)
