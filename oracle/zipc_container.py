"""CPU restatement of the reference's ZIP container and member model (`src/zipc.ml`).

TEST INFRASTRUCTURE ONLY -- imported by tests/ to check the C++ host layer
(zipc_amd/host) byte for byte; the product never loads it.  Pure Python: the
container is a few dozen fixed-offset fields per member.  Codec calls go to the
deflate oracle (oracle/zd_oracle.c) like the reference's go to Zipc_deflate.

Each function cites the rows of /root/reference/src/zipc.ml it follows.  Pinned by
tests/test_zipc_container.py against the reference's own container test
(test/test.ml:57-118: the zip-docs.zip fixture's member kinds, times, modes, sizes
and CRCs) and against Python's zipfile / Info-ZIP unzip as independent readers.
"""
import struct

UINT32_MAX = 0xFFFFFFFF

# compression zipc.ml:23-35
STORED, DEFLATE, BZIP2, LZMA, ZSTD, XZ = 0, 8, 12, 14, 93, 95
_COMP_NAMES = {BZIP2: "bz2", DEFLATE: "defl", LZMA: "lzma", STORED: "none", XZ: "xz", ZSTD: "zst"}


def compression_to_string(c):  # zipc.ml:33-35
    return _COMP_NAMES.get(c, "%04d" % c)


# ---- Fpath zipc.ml:39-62
def fpath_ensure_unix(p):
    return p.replace(b"\\", b"/")


def fpath_ensure_directoryness(p):
    if p == b"":
        return b"./"
    return p if p.endswith(b"/") else p + b"/"


def fpath_sanitize(p):
    segs = [s for seg in p.split(b"/") for s in seg.split(b"\\")]
    return b"/".join(s for s in segs if s not in (b"", b"..", b"."))


def fpath_mode_string(m):  # pp_mode zipc.ml:53-61
    def ent(v):
        return ("r" if v & 4 else "-") + ("w" if v & 2 else "-") + ("x" if v & 1 else "-")
    return ent(m >> 6) + ent(m >> 3) + ent(m)


# ---- Ptime zipc.ml:64-124
JD_POSIX_EPOCH = 2440588
DOS_EPOCH = 315532800


def _div(a, b):  # OCaml's / truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _mod(a, b):  # OCaml's mod has the sign of the dividend
    return a - b * _div(a, b)


def ptime_to_date_time(t):  # zipc.ml:67-87
    jd = _div(t, 86400) + JD_POSIX_EPOCH
    jd_rem = _mod(t, 86400)
    hh, hh_rem = _div(jd_rem, 3600), _mod(jd_rem, 3600)
    mm, ss = _div(hh_rem, 60), _mod(hh_rem, 60)
    a = jd + 32044
    b = _div(4 * a + 3, 146097)
    c = a - _div(146097 * b, 4)
    d = _div(4 * c + 3, 1461)
    e = c - _div(1461 * d, 4)
    m = _div(5 * e + 2, 153)
    day = e - _div(153 * m + 2, 5) + 1
    month = m + 3 - 12 * _div(m, 10)
    year = 100 * b + d - 4800 + _div(m, 10)
    return (year, month, day), (hh, mm, ss)


def ptime_to_string(t):  # pp zipc.ml:89-91
    (y, mo, d), (hh, mm, ss) = ptime_to_date_time(t)
    return "%04d-%02d-%02d %02d:%02d:%02dZ" % (y, mo, d, hh, mm, ss)


def ptime_of_dos_date_time(dos_date, dos_time):  # zipc.ml:97-113
    if dos_date < 0x21:
        return DOS_EPOCH
    hh = dos_time >> 11
    mm = (dos_time >> 5) & 0x3F
    ss = (dos_time & 0x1F) * 2
    year = ((dos_date >> 9) & 0x7F) + 1980
    month = (dos_date >> 5) & 0xF
    day = dos_date & 0x1F
    a = _div(14 - month, 12)
    y = year + 4800 - a
    m = month + 12 * a - 3
    jd = day + _div(153 * m + 2, 5) + 365 * y + _div(y, 4) - _div(y, 100) + _div(y, 400) - 32045
    return (jd - JD_POSIX_EPOCH) * 86400 + hh * 3600 + mm * 60 + ss


def ptime_to_dos_date_time(t):  # zipc.ml:115-124
    (y, mo, d), (hh, mm, ss) = ptime_to_date_time(t)
    if y < 1980:
        (y, mo, d), (hh, mm, ss) = (1980, 1, 1), (0, 0, 0)
    elif y > 2107:
        (y, mo, d), (hh, mm, ss) = (2107, 12, 31), (23, 59, 59)
    return d | (mo << 5) | ((y - 1980) << 9), (ss // 2) | (mm << 5) | (hh << 11)


# ---- File zipc.ml:127-226
GP_ENCRYPTED, GP_UTF8 = 0x1, 0x800
GP_DEFAULT = GP_UTF8
VERSION_MADE_BY_DEFAULT = (3 << 8) | 20
VERSION_NEEDED_DEFAULT = 20
ERR_ENCRYPTED = "Encrypted file not supported"


def err_format(c):
    return "Compression %s not supported" % compression_to_string(c)


def err_size(cs, ds):
    return ("Maximum ZIP byte size 4294967295 exceeded by compressed (%d) or decompressed (%d) file size"
            % (cs, ds))


def crc_error(expect, found):  # zipc_deflate.ml:103-104
    return "Checksum mismatch, expected %x found %x)" % (expect, found)


def file_make(compression, compressed_bytes, decompressed_size, decompressed_crc_32, start=0, compressed_size=None,
              version_made_by=VERSION_MADE_BY_DEFAULT, version_needed_to_extract=VERSION_NEEDED_DEFAULT,
              gp_flags=GP_DEFAULT):
    """File.make zipc.ml:157-170 -> (file dict, None) or (None, error)"""
    if compressed_size is None:
        compressed_size = len(compressed_bytes) - start
    if compressed_size < 0 or decompressed_size < 0:
        raise ValueError("negative size")  # Invalid_argument in the reference
    if compressed_size > UINT32_MAX or decompressed_size > UINT32_MAX:
        return None, err_size(compressed_size, decompressed_size)
    return dict(version_made_by=version_made_by, version_needed_to_extract=version_needed_to_extract,
                gp_flags=gp_flags, compression=compression, start=start, compressed_size=compressed_size,
                compressed_bytes=compressed_bytes, decompressed_size=decompressed_size,
                decompressed_crc_32=decompressed_crc_32), None


def file_stored_of_binary_string(s, codec):  # zipc.ml:172-178
    return file_make(STORED, s, len(s), codec.crc32(s), start=0, compressed_size=len(s))


def file_deflate_of_binary_string(s, codec, level=None):  # zipc.ml:180-186; default level is `Best (Q2)
    st, comp, crc = codec.deflate(s, level=codec.LEVEL_BEST if level is None else level, crc_op=codec.CRC_CRC32)
    if st != 0:
        return None, codec.MESSAGES[st]
    return file_make(DEFLATE, comp, len(s), crc)


def file_is_encrypted(f):
    return (f["gp_flags"] & GP_ENCRYPTED) != 0


def file_can_extract(f):  # zipc.ml:203-206
    return not file_is_encrypted(f) and f["compression"] in (STORED, DEFLATE)


def file_compressed_bytes_to_binary_string(f):
    return f["compressed_bytes"][f["start"]:f["start"] + f["compressed_size"]]


def file_to_binary_string_no_crc_check(f, codec):  # zipc.ml:208-221 -> ((bytes, crc), None) or (None, error)
    if file_is_encrypted(f):
        return None, ERR_ENCRYPTED
    if f["compression"] == STORED:
        s = file_compressed_bytes_to_binary_string(f)
        return (s, codec.crc32(s)), None
    if f["compression"] == DEFLATE:
        st, out, crc = codec.inflate(file_compressed_bytes_to_binary_string(f),
                                     decompressed_size=f["decompressed_size"], crc_op=codec.CRC_CRC32)
        if st != 0:
            return None, "deflate: " + codec.MESSAGES[st]
        return (out, crc), None
    return None, err_format(f["compression"])


def file_to_binary_string(f, codec):  # zipc.ml:223-231
    r, e = file_to_binary_string_no_crc_check(f, codec)
    if e is not None:
        return None, e
    s, found = r
    if found != f["decompressed_crc_32"]:
        return None, crc_error(f["decompressed_crc_32"], found)
    return s, None


# ---- Member zipc.ml:233-290
MEMBER_MAX = 0xFFFF
MAX_PATH_LENGTH = 0xFFFF


def member_make(path, kind, mode=None, mtime=DOS_EPOCH):
    """kind: None for Dir, a file dict for File.  -> (member dict, None) or (None, error)"""
    path = fpath_ensure_unix(path)
    if kind is None:
        path = fpath_ensure_directoryness(path)
    if len(path) > MAX_PATH_LENGTH:
        return None, "Maximum ZIP path length %d exceeded (%d)" % (MAX_PATH_LENGTH, len(path))
    if mode is None:
        mode = 0o755 if kind is None else 0o644
    if mtime < DOS_EPOCH:
        mtime = DOS_EPOCH
    return dict(path=path, kind=kind, mode=mode, mtime=mtime), None


def member_to_string(m, crc=False):  # _pp zipc.ml:263-289
    f = m["kind"]
    is_dir = "d" if f is None else "-"
    comp = "none" if f is None else "%4s" % compression_to_string(f["compression"])
    enc = "X" if f is not None and file_is_encrypted(f) else " "
    size = 0 if f is None else f["decompressed_size"]
    if f is None:
        pct = "    "
    else:
        if f["decompressed_size"] == 0:  # float division by zero: nan (0/0) or inf (n/0) -> Float.to_int gives 0
            pct = "%3d%%" % 0
        else:
            pct = "%3d%%" % int(float(f["compressed_size"]) / float(f["decompressed_size"]) * 100.0)
    crcs = "" if not crc else ("        " if f is None else "%08x" % f["decompressed_crc_32"])
    return "%s%s %s%s%s %8d %s %s %s" % (is_dir, fpath_mode_string(m["mode"]), comp, enc, crcs, size, pct,
                                         ptime_to_string(m["mtime"]), m["path"].decode("utf-8", "replace"))


# ---- decoding zipc.ml:314-433
LFH_SIG, CDFH_SIG, EOCD_SIG = 0x04034b50, 0x02014b50, 0x06054b50
LFH_MIN, CDFH_MIN, EOCD_MIN = 30, 46, 22


class ZipError(Exception):
    pass


def _u16(s, i):
    return struct.unpack_from("<H", s, i)[0]


def _u32(s, i):
    return struct.unpack_from("<I", s, i)[0]


def _data_start_of_lfh(s, i, compressed_size):  # zipc.ml:327-336
    if i + LFH_MIN > len(s) or _u32(s, i) != LFH_SIG:
        raise ZipError("Corrupted local file header")
    data_start = i + LFH_MIN + _u16(s, i + 26) + _u16(s, i + 28)
    if data_start + compressed_size > len(s):
        raise ZipError("Corrupted local file header")
    return data_start


def _member_of_cd(s, cd_max, i):  # zipc.ml:344-391
    if i + CDFH_MIN - 1 > cd_max or _u32(s, i) != CDFH_SIG:
        raise ZipError("Corrupted central directory file header")
    path_len = _u16(s, i + 28)
    n = i + CDFH_MIN + path_len + _u16(s, i + 30) + _u16(s, i + 32)
    if n - 1 > cd_max:
        raise ZipError("Corrupted central directory file header")
    # String.sub raises Invalid_argument past the end of s; n - 1 <= cd_max < len(s) keeps it inside
    path = s[i + 46:i + 46 + path_len]
    mtime = ptime_of_dos_date_time(_u16(s, i + 14), _u16(s, i + 12))
    hi = _u16(s, i + 40)
    if hi != 0:
        is_dir, mode = (hi & 0o70000) == 0o40000, hi & 0o07777
    elif s[i + 38] & 0x10:
        is_dir, mode = True, 0o755
    else:
        is_dir, mode = False, 0o644
    if is_dir:
        kind = None
    else:
        compressed_size = _u32(s, i + 20)
        crc = _u32(s, i + 16)
        start_local = _u32(s, i + 42)
        if start_local >= len(s):
            raise ZipError("Corrupted central directory file header")
        start = _data_start_of_lfh(s, start_local, compressed_size)
        if crc == 0:
            crc = _u32(s, start_local + 14)
        kind = dict(version_made_by=_u16(s, i + 4), version_needed_to_extract=_u16(s, i + 6),
                    gp_flags=_u16(s, i + 8), compression=_u16(s, i + 10), compressed_bytes=s, start=start,
                    compressed_size=compressed_size, decompressed_size=_u32(s, i + 24), decompressed_crc_32=crc)
    return n, dict(path=path, mtime=mtime, mode=mode, kind=kind)


def _find_cd_info(s):  # zipc.ml:401-432
    n = len(s)
    start = n - EOCD_MIN
    if start < 0:
        raise ZipError("File too short to be a ZIP archive")
    min_start = n - 65535 - EOCD_MIN
    while True:
        if start < min_start or start < 0:
            raise ZipError("Likely not a ZIP archive: no end of central directory record found")
        if _u32(s, start) == EOCD_SIG:
            break
        start -= 1
    i = start
    disk_num, disk_cd = _u16(s, i + 4), _u16(s, i + 6)
    if disk_num == 0xFFFF:
        raise ZipError("ZIP64 archives are not supported")
    if disk_num != 0 or disk_cd != 0:
        raise ZipError("Multipart archives are not supported")
    count, cd_size, cd_start = _u16(s, i + 10), _u32(s, i + 12), _u32(s, i + 16)
    if cd_start + cd_size > n:
        raise ZipError("Corrupted end of central directory record")
    return cd_start, cd_size, count


def string_has_magic(s):  # zipc.ml:434-437
    return len(s) >= 4 and _u32(s, 0) in (LFH_SIG, EOCD_SIG)


def of_binary_string(s):
    """zipc.ml:439-446 -> ({path: member}, None) or (None, error message)"""
    try:
        cd_start, cd_size, count = _find_cd_info(s)
        cd_max = cd_start + cd_size - 1
        z, i = {}, cd_start
        for _ in range(count):
            if i > cd_max:
                raise ZipError("Truncated central directory")
            i, m = _member_of_cd(s, cd_max, i)
            z[m["path"]] = m  # add: a later member with the same path replaces the earlier one
        return z, None
    except ZipError as e:
        return None, str(e)


# ---- encoding zipc.ml:448-588
def encoding_size(z):  # zipc.ml:455-463
    n = EOCD_MIN
    for m in z.values():
        data = 0 if m["kind"] is None else m["kind"]["compressed_size"]
        n += LFH_MIN + len(m["path"]) + data + CDFH_MIN + len(m["path"])
    return n


def _write_order(z, first):  # zipc.ml:568-581: `first` leads, the rest in byte order of the paths
    paths = sorted(z)
    if first in z:
        paths.remove(first)
        paths.insert(0, first)
    return paths


def to_binary_string(z, first=b"mimetype"):
    """zipc.ml:583-588 -> (bytes, None) or (None, error)"""
    out = bytearray(encoding_size(z))
    if not z:
        struct.pack_into("<IHHHHIIH", out, 0, EOCD_SIG, 0, 0, 0, 0, 0, 0, 0)
        return bytes(out), None
    if len(z) > MEMBER_MAX:
        return None, "Maximum ZIP member count %d exceeded (%d)" % (MEMBER_MAX, len(z))
    pos, placed = 0, []
    for p in _write_order(z, first):  # encode_member zipc.ml:465-506
        m = z[p]
        f = m["kind"]
        date, time = ptime_to_dos_date_time(m["mtime"])
        if f is None:
            needed, flags, comp, crc, cs, ds = VERSION_NEEDED_DEFAULT, GP_DEFAULT, STORED, 0, 0, 0
        else:
            needed, flags, comp = f["version_needed_to_extract"], f["gp_flags"] & ~(1 << 3) & 0xFFFF, f["compression"]
            crc, cs, ds = f["decompressed_crc_32"], f["compressed_size"], f["decompressed_size"]
        struct.pack_into("<IHHHHHIIIHH", out, pos, LFH_SIG, needed, flags, comp, time, date, crc,
                         cs & UINT32_MAX, ds & UINT32_MAX, len(p), 0)
        out[pos + 30:pos + 30 + len(p)] = p
        placed.append((pos, m))
        pos += 30 + len(p)
        if f is not None:
            out[pos:pos + cs] = f["compressed_bytes"][f["start"]:f["start"] + cs]
            pos += cs
    cd_start = pos
    for lfh, m in placed:  # encode_cd_member zipc.ml:508-560
        f, p = m["kind"], m["path"]
        date, time = ptime_to_dos_date_time(m["mtime"])
        if f is None:
            made, needed, flags, comp, crc, cs, ds = (VERSION_MADE_BY_DEFAULT, VERSION_NEEDED_DEFAULT, GP_DEFAULT,
                                                      STORED, 0, 0, 0)
            hi, lo = 0o040000 | (m["mode"] & 0o7777), 0x10
        else:
            made, needed = f["version_made_by"], f["version_needed_to_extract"]
            flags, comp = f["gp_flags"] & ~(1 << 3) & 0xFFFF, f["compression"]
            crc, cs, ds = f["decompressed_crc_32"], f["compressed_size"], f["decompressed_size"]
            hi, lo = 0o100000 | (m["mode"] & 0o7777), 0
        struct.pack_into("<IHHHHHHIIIHHHHHHHI", out, pos, CDFH_SIG, made, needed, flags, comp, time, date, crc,
                         cs & UINT32_MAX, ds & UINT32_MAX, len(p), 0, 0, 0, 0, lo, hi & 0xFFFF, lfh & UINT32_MAX)
        out[pos + 46:pos + 46 + len(p)] = p
        pos += 46 + len(p)
    cd_size = pos - cd_start
    if cd_start > UINT32_MAX:  # encode_eocd zipc.ml:562-582
        return None, "Maximum ZIP central directory offset 4294967295 exceeded (%d)" % cd_start
    if cd_size > UINT32_MAX:
        return None, "Maximum ZIP central directory size 4294967295 exceeded (%d)" % cd_size
    struct.pack_into("<IHHHHIIH", out, pos, EOCD_SIG, 0, 0, len(z), len(z), cd_size, cd_start, 0)
    return bytes(out), None
