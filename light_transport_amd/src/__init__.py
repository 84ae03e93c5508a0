"""Host-side mirror of LightTransportSimulator/light_transport/src (same module
and class names, same constructor signatures), NumPy only.  Scene set-up runs on
the host as in the reference; every per-ray / per-photon computation is routed to
the gfx950 kernels through the C ABI (light_transport_amd._lib)."""
