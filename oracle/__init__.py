"""CPU oracle for the Zipc_deflate hot path -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/zd_oracle.c (a plain-C restatement of the reference's
src/zipc_deflate.ml).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; zipc_amd/ (the product) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libzd_oracle.so")

OK, ERR_CORRUPTED, ERR_SIZE_EXCEEDED, ERR_ZLIB_METHOD, ERR_ZLIB_WINDOW, ERR_ZLIB_DICT, \
    ERR_CHECKSUM, ERR_NOMEM = range(8)
CRC_NOP, CRC_CRC32, CRC_ADLER32 = 0, 1, 2
LEVEL_NONE, LEVEL_FAST, LEVEL_DEFAULT, LEVEL_BEST = 0, 1, 2, 3
LEVELS = {"none": 0, "fast": 1, "default": 2, "best": 3}
BLOCK_STORED, BLOCK_FIXED, BLOCK_DYNAMIC = 0, 1, 2

MESSAGES = {
    ERR_CORRUPTED: "Corrupted data stream",
    ERR_SIZE_EXCEEDED: "Expected decompression size exceeded",
    ERR_ZLIB_WINDOW: "Window size too large",
    ERR_ZLIB_DICT: "Preset dictionary unsupported",
}


class BlockInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("final", C.c_int), ("src_start", C.c_uint32),
                ("src_len", C.c_uint32), ("n_syms", C.c_uint32),
                ("nlen", C.c_int64), ("flen", C.c_int64), ("dlen", C.c_int64)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "zd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                              os.path.getmtime(os.path.join(_HERE, "zd_oracle.h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # ZD_ORACLE_LIB: another build of the same source (tests/test_sanitizers.py: oracle/libzd_oracle_asan.so)
    L = C.CDLL(os.environ["ZD_ORACLE_LIB"]) if os.environ.get("ZD_ORACLE_LIB") else C.CDLL(build())
    u8p = C.c_void_p
    L.zd_crc32.restype = C.c_uint32
    L.zd_crc32.argtypes = [u8p, C.c_size_t]
    L.zd_crc32_update.restype = C.c_uint32
    L.zd_crc32_update.argtypes = [C.c_uint32, u8p, C.c_size_t]
    L.zd_adler32.restype = C.c_uint32
    L.zd_adler32.argtypes = [u8p, C.c_size_t]
    L.zd_adler32_update.restype = C.c_uint32
    L.zd_adler32_update.argtypes = [C.c_uint32, u8p, C.c_size_t]
    L.zd_inflate.restype = C.c_int
    L.zd_inflate.argtypes = [u8p, C.c_size_t, C.c_int, C.c_size_t, C.c_int,
                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                             C.POINTER(C.c_uint32)]
    L.zd_inflate_into.restype = C.c_int
    L.zd_inflate_into.argtypes = [u8p, C.c_size_t, C.c_int, C.c_size_t, C.c_int,
                                  u8p, C.c_size_t, C.POINTER(C.c_size_t),
                                  C.POINTER(C.c_uint32)]
    L.zd_zlib_decompress.restype = C.c_int
    L.zd_zlib_decompress.argtypes = [u8p, C.c_size_t, C.c_int, C.c_size_t,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]
    L.zd_deflate.restype = C.c_int
    L.zd_deflate.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                             C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    L.zd_deflate_trace.restype = C.c_int
    L.zd_deflate_trace.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_uint32), C.POINTER(BlockInfo),
                                   C.c_size_t, C.POINTER(C.c_size_t)]
    L.zd_zlib_compress.restype = C.c_int
    L.zd_zlib_compress.argtypes = [u8p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    L.zd_deflate_bound.restype = C.c_size_t
    L.zd_deflate_bound.argtypes = [C.c_size_t]
    L.zd_huffman_lengths_of_freqs.restype = None
    L.zd_huffman_lengths_of_freqs.argtypes = [C.POINTER(C.c_int64), C.c_int, C.c_int,
                                              C.POINTER(C.c_int)]
    L.zd_free.restype = None
    L.zd_free.argtypes = [C.c_void_p]
    _lib = L
    return L


def _buf(data):
    """bytes-like -> (keepalive, address, length) without copying when possible."""
    if isinstance(data, (bytes, bytearray)):
        b = bytes(data) if isinstance(data, bytearray) else data
        return b, C.cast(C.c_char_p(b), C.c_void_p), len(b)
    mv = memoryview(data).cast("B")
    if mv.readonly:
        b = mv.tobytes()
        return b, C.cast(C.c_char_p(b), C.c_void_p), len(b)
    arr = (C.c_ubyte * len(mv)).from_buffer(mv)
    return arr, C.cast(arr, C.c_void_p), len(mv)


def _take(ptr, n):
    try:
        return C.string_at(ptr, n) if n else b""
    finally:
        lib().zd_free(ptr)


def crc32(data) -> int:
    k, p, n = _buf(data)
    return lib().zd_crc32(p, n)


def crc32_update(state: int, data) -> int:
    k, p, n = _buf(data)
    return lib().zd_crc32_update(state, p, n)


def adler32(data) -> int:
    k, p, n = _buf(data)
    return lib().zd_adler32(p, n)


def adler32_update(state: int, data) -> int:
    k, p, n = _buf(data)
    return lib().zd_adler32_update(state, p, n)


def inflate(data, decompressed_size=None, crc_op=CRC_NOP):
    """-> (status, bytes, checksum)"""
    k, p, n = _buf(data)
    out, out_len, crc = C.c_void_p(), C.c_size_t(), C.c_uint32()
    st = lib().zd_inflate(p, n, int(decompressed_size is not None),
                          decompressed_size or 0, crc_op, C.byref(out), C.byref(out_len),
                          C.byref(crc))
    if st != OK:
        return st, b"", 0
    return st, _take(out, out_len.value), crc.value


def zlib_decompress(data, decompressed_size=None):
    """-> (status, bytes, adler, expect, found)"""
    k, p, n = _buf(data)
    out, out_len = C.c_void_p(), C.c_size_t()
    adler, expect, found = C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = lib().zd_zlib_decompress(p, n, int(decompressed_size is not None),
                                  decompressed_size or 0, C.byref(out), C.byref(out_len),
                                  C.byref(adler), C.byref(expect), C.byref(found))
    if st != OK:
        return st, b"", 0, expect.value, found.value
    return st, _take(out, out_len.value), adler.value, expect.value, found.value


def deflate(data, level=LEVEL_BEST, crc_op=CRC_NOP):
    """-> (status, bytes, checksum).  NB default level is BEST like zd.ml:817."""
    k, p, n = _buf(data)
    out, out_len, crc = C.c_void_p(), C.c_size_t(), C.c_uint32()
    st = lib().zd_deflate(p, n, level, crc_op, C.byref(out), C.byref(out_len), C.byref(crc))
    if st != OK:
        return st, b"", 0
    return st, _take(out, out_len.value), crc.value


def deflate_trace(data, level=LEVEL_BEST, crc_op=CRC_NOP, max_blocks=4096):
    """-> (status, bytes, checksum, [BlockInfo])"""
    k, p, n = _buf(data)
    out, out_len, crc = C.c_void_p(), C.c_size_t(), C.c_uint32()
    blocks = (BlockInfo * max_blocks)()
    nb = C.c_size_t()
    st = lib().zd_deflate_trace(p, n, level, crc_op, C.byref(out), C.byref(out_len),
                                C.byref(crc), blocks, max_blocks, C.byref(nb))
    if st != OK:
        return st, b"", 0, []
    return st, _take(out, out_len.value), crc.value, list(blocks[: min(nb.value, max_blocks)])


def zlib_compress(data, level=LEVEL_BEST):
    """-> (status, bytes, adler)"""
    k, p, n = _buf(data)
    out, out_len, adler = C.c_void_p(), C.c_size_t(), C.c_uint32()
    st = lib().zd_zlib_compress(p, n, level, C.byref(out), C.byref(out_len), C.byref(adler))
    if st != OK:
        return st, b"", 0
    return st, _take(out, out_len.value), adler.value


def deflate_bound(n: int) -> int:
    return lib().zd_deflate_bound(n)


def huffman_retries(reset=False):
    """(code-length code, litlen/dist codes): how often Huffman.lengths_of_freqs' flatten-and-retry
    branch (zd.ml:470-473) has run in this process -- lets a test prove that its input reaches it"""
    a = (C.c_int * 2).in_dll(lib(), "zd_huffman_retries")
    got = (int(a[0]), int(a[1]))
    if reset:
        a[0] = a[1] = 0
    return got


def huffman_lengths(freqs, max_code_len):
    n = len(freqs)
    f = (C.c_int64 * n)(*freqs)
    out = (C.c_int * n)()
    lib().zd_huffman_lengths_of_freqs(f, n - 1, max_code_len, out)
    return list(out)
