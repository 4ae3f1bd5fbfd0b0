"""unidom_amd -- MI355X-native (gfx950) differentiable-physics hot path of Kuroki1931/UniDOM.

Hand-written HIP kernels (csrc/) behind a C ABI (include/unidom_hip.h); this package is the host-side
mirror of the reference's simulator / env / APG interface for that path.
"""
__version__ = "0.1.0"
