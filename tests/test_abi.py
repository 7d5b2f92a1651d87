"""The C-ABI library builds for gfx950, loads without a GPU, and exports every
symbol include/zipc_hip.h declares.  No compute calls here."""
import ctypes as C
import os
import re

import util  # noqa: F401  (sets sys.path through conftest)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="zipc_hip.h", prefix="zipc_hip_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from zipc_amd import _lib

    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(s[0] for s in _lib.SYMBOLS) == names  # the Python binding covers the whole header


def test_host_library_exports_every_declared_symbol():
    """include/zipc_host.h (the host layer: Zipc member glue + ZIP container)"""
    from zipc_amd import zipc_host

    L = zipc_host.lib()
    names = declared_symbols("zipc_host.h", "zipc_host_")
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert sorted(s[0] for s in zipc_host.SYMBOLS) == names
    assert C.sizeof(zipc_host.MemberRec) == 88 and C.sizeof(zipc_host.MemberOpts) == 24


def test_abi_version_and_struct_layout():
    from zipc_amd import _lib
    from zipc_amd.batch import DESC_DTYPE, RESULT_DTYPE

    assert _lib.lib().zipc_hip_abi_version() == 1
    assert C.sizeof(_lib.StreamDesc) == 48 == DESC_DTYPE.itemsize
    assert C.sizeof(_lib.StreamResult) == 16 == RESULT_DTYPE.itemsize


def test_error_strings_are_the_references():
    from zipc_amd import _lib

    L = _lib.lib()
    assert L.zipc_hip_strerror(1) == b"Corrupted data stream"  # zipc_deflate.ml:233
    assert L.zipc_hip_strerror(2) == b"Expected decompression size exceeded"  # :29
    assert L.zipc_hip_strerror(4) == b"Window size too large"  # :729
    assert L.zipc_hip_strerror(5) == b"Preset dictionary unsupported"  # :730
    assert b"Checksum mismatch" in L.zipc_hip_strerror(6)  # :104


def test_deflate_bound_covers_stored_worst_case():
    from zipc_amd import _lib

    L = _lib.lib()
    for n in (0, 1, 65534, 65535, 65536, 1 << 20, (1 << 20) + 7):
        blocks = n // 65534 + 1
        assert L.zipc_hip_deflate_bound(n) >= n + 5 * blocks


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product fails loudly instead of computing."""
    import torch

    if torch.cuda.is_available():
        return
    import pytest

    import zipc_amd

    with pytest.raises(zipc_amd.ZipcHipError):
        zipc_amd.Context(0)
