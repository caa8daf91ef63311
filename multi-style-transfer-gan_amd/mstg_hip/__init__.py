"""mstg_hip -- hand-written gfx950 (MI355X) kernels for the multi-style-transfer-gan hot path, behind a C ABI.

Importing the package does not touch the GPU; the shared library is loaded on first use and its absence is an
error (there is no CPU / eager-PyTorch fallback).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "layers", "build", "optim", "dp"]
