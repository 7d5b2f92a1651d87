"""zipc_amd -- MI355X-native Zipc_deflate hot path (deflate / inflate / CRC-32 / Adler-32).

Layout: csrc/ holds the HIP kernels and the C ABI (include/zipc_hip.h);
zipc_deflate.py mirrors the reference's ``Zipc_deflate`` module on top of it;
batch.py drives the device-resident batch forms; shard.py splits a batch of
independent members over the ranks of a node.
"""
from . import _lib  # noqa: F401
from ._lib import Context, ZipcHipError, default_context  # noqa: F401

__all__ = ["Context", "ZipcHipError", "default_context"]
