"""Host-side mirror of the reference's module ``Zipc_deflate`` (src/zipc_deflate.mli).

Same names, argument meaning and error behaviour as the OCaml signature; every
function runs on the GPU through the C ABI of include/zipc_hip.h (host forms).
OCaml's ``result`` is mirrored by :class:`Ok` / :class:`Error` so the parity tests
read like the reference's own (``test/test.ml``):

    cs = zipc_deflate.deflate(s, level="fast").get_ok()
    assert zipc_deflate.inflate(cs).get_ok() == s

Reference quirks kept on purpose: the default ``level`` is
``"best"`` (zipc_deflate.ml:817, although the .mli says `Default); Adler-32 uses
the reference's signed 32-bit remainder.  ``start``/``len`` select the byte range
like ``?start ?len``; for ``deflate`` and ``zlib_*`` the reference mis-handles
``start <> 0`` (Q4/Q5), so here the range is simply sliced first.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from ._lib import (CRC_ADLER32, CRC_CRC32, CRC_NOP, ERR_CHECKSUM, ERR_DST_TOO_SMALL,
                   ERR_ZLIB_METHOD, OK, default_context, lib)

LEVELS = {"none": 0, "fast": 1, "default": 2, "best": 3,
          "None": 0, "Fast": 1, "Default": 2, "Best": 3}


class Ok:
    __slots__ = ("value",)

    def __init__(self, value):
        self.value = value

    def get_ok(self):
        return self.value

    def is_ok(self):
        return True

    def is_error(self):
        return False

    def __repr__(self):
        return "Ok(%r)" % (self.value,)


class Error:
    __slots__ = ("error",)

    def __init__(self, error):
        self.error = error

    def get_ok(self):
        raise ValueError("Result.get_ok on Error %r" % (self.error,))

    def is_ok(self):
        return False

    def is_error(self):
        return True

    def __repr__(self):
        return "Error(%r)" % (self.error,)


def _range(s, start, length):
    """default_length (zipc_deflate.ml:9-10) + OCaml's bounds checks."""
    s = bytes(s) if not isinstance(s, bytes) else s
    n = len(s) - start if length is None else length
    if start < 0 or n < 0 or start + n > len(s):
        raise ValueError("index out of bounds")  # Invalid_argument in the reference
    return s[start:start + n] if (start or n != len(s)) else s


def _message(status, detail=None):
    msg = lib().zipc_hip_strerror(status).decode()
    if status == ERR_ZLIB_METHOD and detail is not None:
        msg = msg % detail
    return msg


def crc_error(expect, found):
    # crc_error zipc_deflate.ml:103-104 (the unbalanced parenthesis is the reference's)
    return "Checksum mismatch, expected %x found %x)" % (expect, found)


class _Checksum:
    @staticmethod
    def equal(a, b):
        return (a & 0xFFFFFFFF) == (b & 0xFFFFFFFF)

    @classmethod
    def check(cls, expect, found):
        return Ok(None) if cls.equal(expect, found) else Error(crc_error(expect, found))

    @staticmethod
    def pp(crc):
        return "%x" % (crc & 0xFFFFFFFF)


class Crc_32(_Checksum):
    """ZIP CRC-32 checksums (zipc_deflate.mli:24-49)."""

    @staticmethod
    def string(s, start=0, len=None, ctx=None):
        data = _range(s, start, len)
        ctx = ctx or default_context()
        out = C.c_uint32()
        ctx.check(lib().zipc_hip_crc32(ctx.handle, data, data.__len__(), C.byref(out)))
        return out.value


class Adler_32(_Checksum):
    """Adler-32 checksums (zipc_deflate.mli:52-75)."""

    @staticmethod
    def string(s, start=0, len=None, ctx=None):
        data = _range(s, start, len)
        ctx = ctx or default_context()
        out = C.c_uint32()
        ctx.check(lib().zipc_hip_adler32(ctx.handle, data, data.__len__(), C.byref(out)))
        return out.value


def _inflate(s, decompressed_size, start, length, crc_op, ctx):
    data = _range(s, start, length)
    ctx = ctx or default_context()
    n = len(data)
    has_limit = decompressed_size is not None
    # without a limit the reference starts at 3x the input and doubles
    # (zipc_deflate.ml:553, Buf.grow :27-38); same policy for the device buffer
    cap = decompressed_size if has_limit else max(3 * n, 1024)
    while True:
        dst = C.create_string_buffer(max(cap, 1))
        out_len, crc = C.c_size_t(), C.c_uint32()
        st = lib().zipc_hip_inflate(ctx.handle, data, n, int(has_limit), decompressed_size or 0,
                                    crc_op, dst, cap, C.byref(out_len), C.byref(crc))
        if st == ERR_DST_TOO_SMALL and not has_limit:
            cap *= 2
            continue
        if st == OK:
            return Ok((dst.raw[:out_len.value], crc.value))
        if st in (_lib.ERR_HIP, _lib.ERR_INVALID_ARG, _lib.ERR_NO_DEVICE, _lib.ERR_NOMEM):
            ctx.check(st)
        return Error(_message(st))


def inflate(s, decompressed_size=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:79-91"""
    r = _inflate(s, decompressed_size, start, len, CRC_NOP, ctx)
    return Ok(r.value[0]) if r.is_ok() else r


def inflate_and_crc_32(s, decompressed_size=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:93-97"""
    return _inflate(s, decompressed_size, start, len, CRC_CRC32, ctx)


def inflate_and_adler_32(s, decompressed_size=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:99-103"""
    return _inflate(s, decompressed_size, start, len, CRC_ADLER32, ctx)


def zlib_decompress(s, decompressed_size=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:104-118: Ok (bytes, adler) | Error ((expect, found) | None, msg)"""
    data = _range(s, start, len)
    ctx = ctx or default_context()
    n = data.__len__()
    has_limit = decompressed_size is not None
    cap = decompressed_size if has_limit else max(3 * n, 1024)
    while True:
        dst = C.create_string_buffer(max(cap, 1))
        out_len = C.c_size_t()
        adler, expect, found = C.c_uint32(), C.c_uint32(), C.c_uint32()
        st = lib().zipc_hip_zlib_decompress(ctx.handle, data, n, int(has_limit),
                                            decompressed_size or 0, dst, cap, C.byref(out_len),
                                            C.byref(adler), C.byref(expect), C.byref(found))
        if st == ERR_DST_TOO_SMALL and not has_limit:
            cap *= 2
            continue
        if st == OK:
            return Ok((dst.raw[:out_len.value], adler.value))
        if st == ERR_CHECKSUM:
            return Error(((expect.value, found.value), crc_error(expect.value, found.value)))
        if st in (_lib.ERR_HIP, _lib.ERR_INVALID_ARG, _lib.ERR_NO_DEVICE, _lib.ERR_NOMEM):
            ctx.check(st)
        return Error((None, _message(st, data[0] & 0x0F if n else 0)))


def _level(level):
    if level is None:
        return LEVELS["best"]  # make_encoder ?(level = `Best) zipc_deflate.ml:817
    if isinstance(level, int):
        return level
    return LEVELS[level]


def _deflate(s, level, start, length, crc_op, ctx):
    data = _range(s, start, length)
    ctx = ctx or default_context()
    n = data.__len__()
    cap = lib().zipc_hip_deflate_bound(n)
    dst = C.create_string_buffer(cap)
    out_len, crc = C.c_size_t(), C.c_uint32()
    st = lib().zipc_hip_deflate(ctx.handle, data, n, _level(level), crc_op, dst, cap,
                                C.byref(out_len), C.byref(crc))
    if st != OK:
        ctx.check(st)
    return Ok((crc.value, dst.raw[:out_len.value]))


def deflate(s, level=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:128-137"""
    return Ok(_deflate(s, level, start, len, CRC_NOP, ctx).value[1])


def crc_32_and_deflate(s, level=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:139-143: Ok (crc, bytes)"""
    return _deflate(s, level, start, len, CRC_CRC32, ctx)


def adler_32_and_deflate(s, level=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:145-149: Ok (adler, bytes)"""
    return _deflate(s, level, start, len, CRC_ADLER32, ctx)


def zlib_compress(s, level=None, start=0, len=None, ctx=None):
    """zipc_deflate.mli:152-162: Ok (adler, bytes)"""
    data = _range(s, start, len)
    ctx = ctx or default_context()
    n = data.__len__()
    cap = lib().zipc_hip_zlib_bound(n)
    dst = C.create_string_buffer(cap)
    out_len, adler = C.c_size_t(), C.c_uint32()
    st = lib().zipc_hip_zlib_compress(ctx.handle, data, n, _level(level), dst, cap,
                                      C.byref(out_len), C.byref(adler))
    if st != OK:
        ctx.check(st)
    return Ok((adler.value, dst.raw[:out_len.value]))
