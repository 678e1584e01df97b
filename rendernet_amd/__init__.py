"""rendernet_amd -- MI355X-native (gfx950) forward render path of RenderNet.

Hand-written HIP kernels behind a C ABI (include/rendernet_hip.h, rendernet_amd/csrc/), a ctypes
binding (rendernet_amd/_lib.py), tensor-level operators (rendernet_amd/ops.py), mirrors of the
reference's `tools/` builders (rendernet_amd/tools/) and the Phong-shader graph + Session-like
runner (rendernet_amd/shader.py).  There is no CPU fallback: operators raise if the HIP library
or a HIP device is missing.
"""
__version__ = "0.1.0"
