"""libacm_amd - MI355X-native ACM decode path (drop-in for markokr/libacm's decode path).

The product is the C-ABI shared library libacm_amd/lib/libacm_hip.so (headers in
include/); this package is thin plumbing around it: build recipes, ctypes
bindings, the synthetic-stream writer, and the multi-GPU batch front end.
"""
from . import _build  # noqa: F401

__all__ = ["_build"]
