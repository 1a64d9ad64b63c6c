"""MI355X-native SIFT hot path behind the `sift_anatomy` C API.

The product is libsift_amd.so (HIP kernels + C ABI, see include/sift_amd.h); this package
is its Python binding plus the deterministic synthetic-image generators used by the
benchmark and the tests.
"""
from . import abi, synth  # noqa: F401
from .api import Context, SiftAmdError, lib, device_count  # noqa: F401
