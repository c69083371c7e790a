"""adypt_amd — MI355X-native wavefront path tracer behind Adypt's scene interface.

The compute path is hand-written HIP for gfx950 in ``adypt_amd/csrc`` (libadypt_hip.so, C-ABI in ``include/``);
this package is the thin host-side mirror of the reference's interface (``api``), the procedural stand-in scenes
(``scenes``) and the multi-GPU tile plumbing (``distributed``).  Importing ``adypt_amd.api`` requires the built
shared library — there is no fallback.
"""
__all__ = ["api", "scenes", "distributed"]
