"""boa_hip -- MI355X-native engine for the Body-and-Organ-Analysis hot path (host side).

Python host code mirroring the reference's operator surface for this path; all per-voxel work runs in
hand-written HIP kernels (libboa_hip.so, C ABI in include/boa_hip.h).  No CPU fallback exists: the compute
entry points raise if the library cannot be loaded.
"""
__version__ = "0.1.0"
