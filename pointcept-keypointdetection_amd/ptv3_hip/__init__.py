"""ctypes binding of libptv3_hip.so (C ABI: include/ptv3_hip.h) + tensor-level op wrappers.

The library is the product; torch is only used for device memory and the current HIP stream.
There is NO CPU fallback: every op raises if the library is missing or a tensor is not on the GPU.
"""
from .lib import lib, library_path, PTV3_F32, PTV3_BF16  # noqa: F401
from . import ops  # noqa: F401
