"""ctypes binding of libzipc_hip.so (include/zipc_hip.h).

The library is the product: if it is missing this module raises, and if no GPU is
usable every call returns an error status -- there is no CPU fallback anywhere in
this package.
"""
from __future__ import annotations

import ctypes as C
import threading
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZIPC_HIP_LIB") or os.path.join(HERE, "lib", "libzipc_hip.so")

OK = 0
ERR_CORRUPTED = 1
ERR_SIZE_EXCEEDED = 2
ERR_ZLIB_METHOD = 3
ERR_ZLIB_WINDOW = 4
ERR_ZLIB_DICT = 5
ERR_CHECKSUM = 6
ERR_DST_TOO_SMALL = 16
ERR_HIP = 17
ERR_INVALID_ARG = 18
ERR_NO_DEVICE = 19
ERR_NOMEM = 20

CRC_NOP, CRC_CRC32, CRC_ADLER32, CRC_ADLER32_RFC1950 = 0, 1, 2, 3
LEVEL_NONE, LEVEL_FAST, LEVEL_DEFAULT, LEVEL_BEST = 0, 1, 2, 3
STREAM_HAS_LIMIT = 1


class StreamDesc(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("src_len", C.c_uint64), ("dst_off", C.c_uint64),
                ("dst_cap", C.c_uint64), ("limit", C.c_uint64), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class StreamResult(C.Structure):
    _fields_ = [("status", C.c_uint32), ("checksum", C.c_uint32), ("out_len", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


# every symbol include/zipc_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SZ = C.c_size_t
_U32P = C.POINTER(C.c_uint32)
_SZP = C.POINTER(C.c_size_t)
SYMBOLS = [
    ("zipc_hip_abi_version", C.c_int, []),
    ("zipc_hip_device_count", C.c_int, []),
    ("zipc_hip_create", C.c_int, [C.POINTER(_P), C.c_int]),
    ("zipc_hip_destroy", None, [_P]),
    ("zipc_hip_stream", _P, [_P]),
    ("zipc_hip_synchronize", C.c_int, [_P]),
    ("zipc_hip_last_error", C.c_char_p, [_P]),
    ("zipc_hip_chain_check", C.c_int, [_P, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    ("zipc_hip_last_inflate_blocks", C.c_uint, [_P]),
    ("zipc_hip_lds_exchange_ordered", C.c_int, [_P]),
    ("zipc_hip_debug_set_slices", None, [C.c_long]),
    ("zipc_hip_debug_chain_positions", _SZ, [_SZ, _SZ]),
    ("zipc_hip_debug_chain_links", C.c_int, [_P, _P, _P, _SZ, _SZ, _SZ, C.c_int, _P, _SZ, _P]),
    ("zipc_hip_strerror", C.c_char_p, [C.c_int]),
    ("zipc_hip_set_profiling", C.c_int, [_P, C.c_int]),
    ("zipc_hip_set_adler_rfc1950", C.c_int, [_P, C.c_int]),
    ("zipc_hip_reset_kernel_times", C.c_int, [_P]),
    ("zipc_hip_kernel_times", C.c_int, [_P, C.POINTER(KernelTime), _SZ, _SZP]),
    ("zipc_hip_crc32", C.c_int, [_P, _P, _SZ, _U32P]),
    ("zipc_hip_adler32", C.c_int, [_P, _P, _SZ, _U32P]),
    ("zipc_hip_inflate", C.c_int, [_P, _P, _SZ, C.c_int, _SZ, C.c_int, _P, _SZ, _SZP, _U32P]),
    ("zipc_hip_zlib_decompress", C.c_int,
     [_P, _P, _SZ, C.c_int, _SZ, _P, _SZ, _SZP, _U32P, _U32P, _U32P]),
    ("zipc_hip_deflate_bound", _SZ, [_SZ]),
    ("zipc_hip_zlib_bound", _SZ, [_SZ]),
    ("zipc_hip_deflate", C.c_int, [_P, _P, _SZ, C.c_int, C.c_int, _P, _SZ, _SZP, _U32P]),
    ("zipc_hip_zlib_compress", C.c_int, [_P, _P, _SZ, C.c_int, _P, _SZ, _SZP, _U32P]),
    ("zipc_hip_deflate_many", C.c_int, [_P, _SZ, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    ("zipc_hip_inflate_many", C.c_int, [_P, _SZ, _P, _P, _P, C.c_int, _P, _P, _P]),
    ("zipc_hip_inflate_many_check", C.c_int, [_P, _SZ, _P, _P, _P, C.c_int, _P, _P]),
    ("zipc_hip_inflate_batch", C.c_int, [_P, _P, _P, _P, _P, _SZ, _SZ, C.c_int]),
    ("zipc_hip_deflate_batch", C.c_int, [_P, _P, _P, _P, _P, _SZ, _SZ, _SZ, C.c_int, C.c_int]),
    ("zipc_hip_checksum_device", C.c_int, [_P, _P, _SZ, C.c_int, C.c_int, _P]),
    ("zipc_hip_reserve", C.c_int, [_P, _SZ, _SZ, _SZ]),
]

_lib = None


def lib():
    """The loaded library; raises if libzipc_hip.so has not been built."""
    global _lib
    if _lib is None:
        # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  A
        # process that uses both must load torch's copy FIRST: if libzipc_hip.so pulls
        # in /opt/rocm's runtime before torch is imported, torch later fails with
        # "No HIP GPUs are available" (two runtimes, one SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "zipc_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C zipc_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class ZipcHipError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().zipc_hip_strerror(status).decode()
        super().__init__("%s (status %d)%s" % (msg, status, (": " + detail) if detail else ""))


class Context:
    """zipc_hip_ctx: a HIP stream + device scratch on one GPU."""

    def __init__(self, device: int = 0):
        self._h = _P()
        st = lib().zipc_hip_create(C.byref(self._h), device)
        if st != OK:
            self._h = None
            raise ZipcHipError(st, "zipc_hip_create(device=%d)" % device)
        self.device = device

    @property
    def handle(self):
        return self._h

    def check(self, st):
        if st != OK:
            raise ZipcHipError(st, lib().zipc_hip_last_error(self._h).decode())

    def synchronize(self):
        self.check(lib().zipc_hip_synchronize(self._h))

    def set_adler_rfc1950(self, on: bool):
        """the zlib forms and checksum_device of this context compute RFC 1950's Adler-32 (what zlib
        computes) instead of the reference's signed-remainder value (include/zipc_hip.h)"""
        self.check(lib().zipc_hip_set_adler_rfc1950(self._h, int(on)))

    def last_inflate_blocks(self) -> int:
        """blocks the last one-stream inflate was decoded by, a wave each (0: by the stream's one wave)"""
        return int(lib().zipc_hip_last_inflate_blocks(self._h))

    def lds_exchange_ordered(self) -> bool:
        """hash chains are built by ordered LDS exchange (the context's probe passed; include/zipc_hip.h)"""
        return bool(lib().zipc_hip_lds_exchange_ordered(self._h))

    def chain_check(self):
        """(positions whose links both chain kernels made and were compared, how many differed): the create-time probe in
        the kernel's own shape plus the first streams of this context's first deflate batch (include/zipc_hip.h)"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        self.check(lib().zipc_hip_chain_check(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_profiling(self, on: bool):
        self.check(lib().zipc_hip_set_profiling(self._h, int(on)))

    def reset_kernel_times(self):
        self.check(lib().zipc_hip_reset_kernel_times(self._h))

    def kernel_times(self):
        n = C.c_size_t()
        arr = (KernelTime * 64)()
        self.check(lib().zipc_hip_kernel_times(self._h, arr, 64, C.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms))
                for i in range(min(n.value, 64))}

    def close(self):
        if self._h:
            lib().zipc_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = threading.local()  # a context serves one thread at a time: every thread gets its own


def default_context(device: int = 0) -> Context:
    """the calling thread's context on `device` (made on first use)"""
    per_thread = getattr(_default, "ctx", None)
    if per_thread is None:
        per_thread = _default.ctx = {}
    ctx = per_thread.get(device)
    if ctx is None:
        ctx = per_thread[device] = Context(device)
    return ctx
