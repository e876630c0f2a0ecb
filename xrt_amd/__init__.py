"""xrt_amd — MI355X (gfx950) compute backend for the hot path of xrt's raycing
engine: ray-surface intersection + reflect/refract amplitudes (P1) and the
Fresnel-Kirchhoff diffraction integral (P2), as hand-written HIP kernels behind
a C ABI (include/xrt_hip.h), with a host-side mirror of xrt's operator API in
``xrt_amd.backends.raycing``.
"""
__version__ = '0.1.0'
