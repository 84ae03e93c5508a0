"""MI355X-native photon-transport hot path behind the LightTransportSimulator API.

Host side: plain Python mirroring the reference's modules (``light_transport_amd.src``);
device side: hand-written HIP kernels for gfx950 reached through the C ABI of
``include/lt.h`` (``liblt_hip.so``).  No CPU fallback exists.
"""
from ._lib import Context, LtError, build, default_context, lib, LIB_PATH  # noqa: F401
from .pipeline import JobPipeline  # noqa: F401

__all__ = ["Context", "LtError", "build", "default_context", "lib", "LIB_PATH", "JobPipeline"]
