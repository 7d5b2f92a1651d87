"""ctypes binding of libzipc_host.so (include/zipc_host.h): the C++ host layer that
mirrors the reference's `Zipc` module -- archive model, member glue, ZIP container --
over the MI355X codec.  Thin: every call is one C function; no logic lives here.
"""
import ctypes as C
import os

from . import _lib

OK, ERROR, INVALID, FAILURE = 0, 1, 2, 3
_P, _SZ = C.c_void_p, C.c_size_t
_ERRCAP = 512


class MemberRec(C.Structure):
    _fields_ = [("path", C.POINTER(C.c_char)), ("path_len", _SZ), ("is_dir", C.c_int), ("mode", C.c_int),
                ("mtime", C.c_int64), ("compression", C.c_int), ("gp_flags", C.c_int), ("version_made_by", C.c_int),
                ("version_needed_to_extract", C.c_int), ("start", C.c_uint64), ("compressed_size", C.c_uint64),
                ("decompressed_size", C.c_uint64), ("decompressed_crc_32", C.c_uint32), ("is_encrypted", C.c_int),
                ("can_extract", C.c_int)]


class MemberOpts(C.Structure):
    _fields_ = [("has_mtime", C.c_int), ("mtime", C.c_int64), ("has_mode", C.c_int), ("mode", C.c_int)]


SYMBOLS = [
    ("zipc_host_empty", _P, []),
    ("zipc_host_of_binary_string", C.c_int, [_P, _SZ, C.POINTER(_P), C.c_char_p, _SZ]),
    ("zipc_host_free", None, [_P]),
    ("zipc_host_string_has_magic", C.c_int, [_P, _SZ]),
    ("zipc_host_member_count", _SZ, [_P]),
    ("zipc_host_member_at", C.c_int, [_P, _SZ, C.POINTER(MemberRec)]),
    ("zipc_host_find", C.c_int, [_P, C.c_char_p, _SZ, C.POINTER(_SZ)]),
    ("zipc_host_remove", C.c_int, [_P, C.c_char_p, _SZ]),
    ("zipc_host_member_pp", _SZ, [_P, _SZ, C.c_int, C.c_char_p, _SZ]),
    ("zipc_host_add_dir", C.c_int, [_P, C.c_char_p, _SZ, C.POINTER(MemberOpts), C.c_char_p, _SZ]),
    ("zipc_host_add_file_made", C.c_int,
     [_P, C.c_char_p, _SZ, C.c_int, _P, _SZ, _SZ, C.c_int64, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int,
      C.POINTER(MemberOpts), C.c_char_p, _SZ]),
    ("zipc_host_add_file_stored", C.c_int, [_P, C.c_char_p, _SZ, _P, _SZ, C.POINTER(MemberOpts), C.c_char_p, _SZ]),
    ("zipc_host_add_file_deflate", C.c_int,
     [_P, C.c_char_p, _SZ, _P, _SZ, C.c_int, C.POINTER(MemberOpts), C.c_char_p, _SZ]),
    ("zipc_host_add_files_deflate", C.c_int, [_P, _SZ, _P, _P, _P, _P, C.c_int, C.c_char_p, _SZ]),
    ("zipc_host_encoding_size", _SZ, [_P]),
    ("zipc_host_to_binary_string", C.c_int, [_P, C.c_char_p, _SZ, _P, _SZ, C.POINTER(_SZ), C.c_char_p, _SZ]),
    ("zipc_host_member_to_binary_string", C.c_int,
     [_P, _SZ, C.c_int, _P, _SZ, C.POINTER(_SZ), C.POINTER(C.c_uint32), C.c_char_p, _SZ]),
    ("zipc_host_extract_all", C.c_int, [_P, C.POINTER(_P), C.c_char_p, _SZ]),
    ("zipc_host_set_devices", C.c_int, [C.POINTER(C.c_int), _SZ]),
    ("zipc_host_devices", _SZ, [C.POINTER(C.c_int), _SZ]),
    ("zipc_host_set_thread_device", None, [C.c_int]),
    ("zipc_host_partition", None, [C.POINTER(_SZ), _SZ, _SZ, C.POINTER(_SZ)]),
    ("zipc_host_extraction_count", _SZ, [_P]),
    ("zipc_host_extraction_at", C.c_int,
     [_P, _SZ, C.POINTER(C.POINTER(C.c_char)), C.POINTER(_SZ), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char)),
      C.POINTER(_SZ)]),
    ("zipc_host_extraction_free", None, [_P]),
    ("zipc_host_ptime_to_date_time", None, [C.c_int64, C.POINTER(C.c_int * 6)]),
    ("zipc_host_ptime_of_dos_date_time", C.c_int64, [C.c_int, C.c_int]),
    ("zipc_host_ptime_to_dos_date_time", None, [C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("zipc_host_ptime_pp", _SZ, [C.c_int64, C.c_char_p, _SZ]),
    ("zipc_host_fpath", _SZ, [C.c_int, C.c_char_p, _SZ, C.c_char_p, _SZ]),
    ("zipc_host_fpath_pp_mode", _SZ, [C.c_int, C.c_char_p, _SZ]),
]

_host = None


def lib():
    """libzipc_host.so (built in-tree by zipc_amd/host/Makefile); fails loudly when missing."""
    global _host
    if _host is None:
        _lib.lib()  # libzipc_hip.so first (and torch before it, see _lib.py)
        # ZIPC_HOST_LIB: another build of the same sources (tests/test_sanitizers.py: the host layer under the address sanitizer)
        path = os.environ.get("ZIPC_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libzipc_host.so")
        if not os.path.exists(path):
            raise ImportError("libzipc_host.so is not built: run `make -C zipc_amd/host` (%s)" % path)
        L = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _host = L
    return _host


class ZipcError(Exception):
    """the reference's `Error msg` (code ERROR), Invalid_argument (INVALID) or a library failure"""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
        self.msg = msg


def _opts(mtime, mode):
    o = MemberOpts()
    o.has_mtime, o.mtime = (1, mtime) if mtime is not None else (0, 0)
    o.has_mode, o.mode = (1, mode) if mode is not None else (0, 0)
    return C.byref(o)


def _check(code, err):
    if code != OK:
        raise ZipcError(code, err.value.decode("utf-8", "replace"))


class Archive:
    """Zipc.t"""

    def __init__(self, handle=None):
        self._h = handle if handle is not None else lib().zipc_host_empty()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().zipc_host_free(self._h)
            self._h = None

    @classmethod
    def of_binary_string(cls, s):
        h, err = _P(), C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_of_binary_string(s, len(s), C.byref(h), err, _ERRCAP), err)
        return cls(h)

    def member_count(self):
        return lib().zipc_host_member_count(self._h)

    def member(self, index):
        m = MemberRec()
        if lib().zipc_host_member_at(self._h, index, C.byref(m)) != OK:
            raise IndexError(index)
        d = {f: getattr(m, f) for f, _ in MemberRec._fields_ if f not in ("path", "path_len")}
        d["path"] = C.string_at(m.path, m.path_len)
        return d

    def members(self):
        return [self.member(i) for i in range(self.member_count())]

    def find(self, path):
        i = _SZ()
        return i.value if lib().zipc_host_find(self._h, path, len(path), C.byref(i)) == OK else None

    def remove(self, path):
        lib().zipc_host_remove(self._h, path, len(path))

    def pp(self, index, long_form=False):
        b = C.create_string_buffer(70000)
        n = lib().zipc_host_member_pp(self._h, index, int(long_form), b, len(b))
        return b.raw[:n].decode("utf-8", "replace")

    def add_dir(self, path, mtime=None, mode=None):
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_add_dir(self._h, path, len(path), _opts(mtime, mode), err, _ERRCAP), err)

    def add_file_made(self, path, compression, compressed_bytes, decompressed_size, decompressed_crc_32, start=0,
                      compressed_size=None, gp_flags=0x800, version_made_by=(3 << 8) | 20,
                      version_needed_to_extract=20, mtime=None, mode=None):
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_add_file_made(self._h, path, len(path), compression, compressed_bytes,
                                             len(compressed_bytes), start,
                                             -1 if compressed_size is None else compressed_size, decompressed_size,
                                             decompressed_crc_32, gp_flags, version_made_by, version_needed_to_extract,
                                             _opts(mtime, mode), err, _ERRCAP), err)

    def add_file_stored(self, path, data, mtime=None, mode=None):
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_add_file_stored(self._h, path, len(path), data, len(data), _opts(mtime, mode), err,
                                               _ERRCAP), err)

    def add_file_deflate(self, path, data, level=None, mtime=None, mode=None):
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_add_file_deflate(self._h, path, len(path), data, len(data),
                                                -1 if level is None else level, _opts(mtime, mode), err, _ERRCAP), err)

    def add_files_deflate(self, files, level=None):
        """files: [(path, data)]: one batch on the GPU"""
        n = len(files)
        paths = (C.c_char_p * n)(*[p for p, _ in files])
        plens = (_SZ * n)(*[len(p) for p, _ in files])
        datas = (C.c_char_p * n)(*[d for _, d in files])
        dlens = (_SZ * n)(*[len(d) for _, d in files])
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_add_files_deflate(self._h, n, paths, plens, datas, dlens, -1 if level is None else level,
                                                 err, _ERRCAP), err)

    def encoding_size(self):
        return lib().zipc_host_encoding_size(self._h)

    def to_binary_string(self, first=None):
        cap = self.encoding_size()
        dst, n, err = C.create_string_buffer(max(cap, 1)), _SZ(), C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_to_binary_string(self._h, first, len(first) if first is not None else 0, dst, cap,
                                                C.byref(n), err, _ERRCAP), err)
        return dst.raw[:n.value]

    def member_to_binary_string(self, index, check_crc=True):
        """File.to_binary_string (check_crc) / to_binary_string_no_crc_check -> (bytes, crc)"""
        cap = self.member(index)["decompressed_size"]
        dst, n, crc = C.create_string_buffer(max(cap, 1)), _SZ(), C.c_uint32()
        err = C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_member_to_binary_string(self._h, index, int(check_crc), dst, cap, C.byref(n),
                                                       C.byref(crc), err, _ERRCAP), err)
        return dst.raw[:n.value], crc.value

    def extract_all(self):
        """[(path, bytes | ZipcError)] for every file member, one batch on the GPU"""
        x, err = _P(), C.create_string_buffer(_ERRCAP)
        _check(lib().zipc_host_extract_all(self._h, C.byref(x), err, _ERRCAP), err)
        out = []
        try:
            for i in range(lib().zipc_host_extraction_count(x)):
                p, pl, ok = C.POINTER(C.c_char)(), _SZ(), C.c_int()
                d, dl = C.POINTER(C.c_char)(), _SZ()
                lib().zipc_host_extraction_at(x, i, C.byref(p), C.byref(pl), C.byref(ok), C.byref(d), C.byref(dl))
                data = C.string_at(d, dl.value)
                out.append((C.string_at(p, pl.value), data if ok.value else ZipcError(ERROR, data.decode())))
        finally:
            lib().zipc_host_extraction_free(x)
        return out


def string_has_magic(s):
    return bool(lib().zipc_host_string_has_magic(s, len(s)))


def ptime_to_date_time(t):
    o = (C.c_int * 6)()
    lib().zipc_host_ptime_to_date_time(t, C.byref(o))
    return (o[0], o[1], o[2]), (o[3], o[4], o[5])


def ptime_of_dos_date_time(dos_date, dos_time):
    return lib().zipc_host_ptime_of_dos_date_time(dos_date, dos_time)


def ptime_to_dos_date_time(t):
    d, tm = C.c_int(), C.c_int()
    lib().zipc_host_ptime_to_dos_date_time(t, C.byref(d), C.byref(tm))
    return d.value, tm.value


def ptime_pp(t):
    b = C.create_string_buffer(64)
    n = lib().zipc_host_ptime_pp(t, b, 64)
    return b.raw[:n].decode()


def _fpath(which, p):
    b = C.create_string_buffer(len(p) + 8)
    n = lib().zipc_host_fpath(which, p, len(p), b, len(b))
    return b.raw[:n]


def fpath_ensure_unix(p):
    return _fpath(0, p)


def fpath_ensure_directoryness(p):
    return _fpath(1, p)


def fpath_sanitize(p):
    return _fpath(2, p)


def fpath_pp_mode(m):
    b = C.create_string_buffer(16)
    n = lib().zipc_host_fpath_pp_mode(m, b, 16)
    return b.raw[:n].decode()


def set_devices(devices=()):
    """The devices add_files_deflate / extract_all spread the members of an archive over (empty: every visible one)."""
    arr = (C.c_int * max(len(devices), 1))(*devices)
    if lib().zipc_host_set_devices(arr, len(devices)) != 0:
        raise RuntimeError("zipc_host_set_devices failed")


def devices():
    n = lib().zipc_host_devices(None, 0)
    arr = (C.c_int * max(n, 1))()
    lib().zipc_host_devices(arr, n)
    return list(arr[:n])


def partition(sizes, n_devices):
    """[(lo, hi)] per device: the ranges a batch of members with these sizes is cut into"""
    n = len(sizes)
    arr = (_SZ * max(n, 1))(*sizes)
    bounds = (_SZ * (n_devices + 1))()
    lib().zipc_host_partition(arr, n, n_devices, bounds)
    return [(int(bounds[k]), int(bounds[k + 1])) for k in range(n_devices)]

