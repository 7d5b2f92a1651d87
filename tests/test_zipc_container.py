"""SURVEY 8(f) rows 1-2 -- member glue and ZIP container -- on the CPU:

* the container oracle (oracle/zipc_container.py) is pinned against the reference's own
  container test (test/test.ml:57-118, the zip-docs.zip fixture) and against Python's
  zipfile and Info-ZIP `unzip -t` as independent readers;
* the C++ host layer (zipc_amd/host, through include/zipc_host.h) is compared with the
  oracle byte for byte on everything that needs no codec call: parsing, metadata,
  encoding, dates, paths, error messages.

Codec-backed operations (deflate_of_binary_string, to_binary_string ...) run on the GPU:
tests/test_gpu_zipc.py."""
import io
import os
import random
import shutil
import struct
import subprocess
import zipfile
import zlib

import pytest

import util
from oracle import zipc_container as zc

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zip-docs.zip")


@pytest.fixture(scope="module")
def host():
    from zipc_amd import zipc_host

    zipc_host.lib()
    return zipc_host


def fixture_bytes():
    return open(FIXTURE, "rb").read()


# ---------------------------------------------------------------- oracle pins
def assert_zip_docs(z, original, oracle):
    """assert_zip of test/test.ml:76-105"""
    d = z[b"zip-docs/"]
    assert d["kind"] is None
    assert zc.ptime_to_date_time(d["mtime"]) == ((2023, 10, 21), (15, 6, 50))
    assert d["mode"] == 0o755
    for path, when, size, crc in ((b"zip-docs/rfc1951.txt", (15, 6, 24), 36944, 0xfb4f3400),
                                  (b"zip-docs/APPNOTE.TXT", (15, 6, 50), 174585, 0x39b029c4)):
        m = z[path]
        f = m["kind"]
        assert zc.ptime_to_date_time(m["mtime"]) == ((2023, 10, 21), when)
        assert m["mode"] == 0o644
        if original:
            assert f["compression"] == zc.DEFLATE and f["decompressed_size"] == size
        assert f["decompressed_crc_32"] == crc
        s, e = zc.file_to_binary_string(f, oracle)
        assert e is None and len(s) == size


def test_oracle_reads_the_reference_fixture_like_the_reference_test(oracle):
    z, e = zc.of_binary_string(fixture_bytes())
    assert e is None and sorted(z) == [b"zip-docs/", b"zip-docs/APPNOTE.TXT", b"zip-docs/rfc1951.txt"]
    assert_zip_docs(z, True, oracle)
    # redeflate_recode of test/test.ml:58-74: every extractable file re-deflated (default level), re-encoded
    z2 = dict(z)
    for p, m in z.items():
        f = m["kind"]
        if f is None or not zc.file_can_extract(f):
            continue
        s, _ = zc.file_to_binary_string(f, oracle)
        nf, e = zc.file_deflate_of_binary_string(s, oracle)
        assert e is None
        z2[p], e = zc.member_make(p, nf, mode=m["mode"], mtime=m["mtime"])
        assert e is None
    enc, e = zc.to_binary_string(z2)
    assert e is None and len(enc) == zc.encoding_size(z2)
    z3, e = zc.of_binary_string(enc)
    assert e is None
    assert_zip_docs(z3, True, oracle)
    # independent readers agree with what was written
    with zipfile.ZipFile(io.BytesIO(enc)) as zf:
        assert zf.testzip() is None
        assert sorted(i.filename for i in zf.infolist()) == ["zip-docs/", "zip-docs/APPNOTE.TXT", "zip-docs/rfc1951.txt"]
        info = zf.getinfo("zip-docs/rfc1951.txt")
        assert (info.file_size, info.CRC, info.date_time) == (36944, 0xfb4f3400, (2023, 10, 21, 15, 6, 24))
        assert (info.external_attr >> 16) == 0o100644 and info.create_system == 3


def test_oracle_agrees_with_zipfile_on_the_fixture():
    z, _ = zc.of_binary_string(fixture_bytes())
    with zipfile.ZipFile(FIXTURE) as zf:
        for i in zf.infolist():
            m = z[i.filename.encode()]
            (y, mo, d), (hh, mm, ss) = zc.ptime_to_date_time(m["mtime"])
            assert (y, mo, d, hh, mm, ss) == i.date_time
            assert (m["kind"] is None) == i.is_dir()
            if m["kind"] is not None:
                f = m["kind"]
                assert (f["compressed_size"], f["decompressed_size"], f["decompressed_crc_32"]) == \
                       (i.compress_size, i.file_size, i.CRC)
                assert f["compression"] == i.compress_type


@pytest.mark.skipif(shutil.which("unzip") is None, reason="Info-ZIP unzip not installed")
def test_oracle_output_passes_unzip_t(tmp_path):
    z = random_archive(random.Random(5), 12)
    enc, e = zc.to_binary_string(z)
    assert e is None
    p = tmp_path / "a.zip"
    p.write_bytes(enc)
    r = subprocess.run(["unzip", "-t", str(p)], capture_output=True, text=True)
    assert r.returncode == 0 and "No errors detected" in r.stdout, r.stdout + r.stderr


# ---------------------------------------------------------------- helpers
def py_file(data, method, r=None):
    """a file dict built like File.make from bytes compressed by Python's zlib"""
    if method == zc.STORED:
        comp = data
    else:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
    f, e = zc.file_make(method, comp, len(data), zlib.crc32(data))
    assert e is None
    return f


def random_archive(r, n, with_mimetype=False):
    z = {}
    if with_mimetype:
        z[b"mimetype"], _ = zc.member_make(b"mimetype", py_file(b"application/epub+zip", zc.STORED))
    for k in range(n):
        depth = r.randrange(1, 4)
        path = b"/".join(bytes(r.choice(b"abcXYZ_-.09") for _ in range(r.randrange(1, 9))) for _ in range(depth))
        mtime = r.choice([None, 0, 315532800, 315532801, 1697900810, 4354819199, 4354819200, 5000000000,
                          r.randrange(315532800, 4354819200)])
        mode = r.choice([None, 0o644, 0o755, 0o600, 0o7777, 0o100644, 0])
        kw = {}
        if mtime is not None:
            kw["mtime"] = mtime
        if mode is not None:
            kw["mode"] = mode
        if r.random() < 0.25:
            m, e = zc.member_make(path + r.choice([b"", b"/", b"\\"]), None, **kw)
        else:
            data = bytes(r.choice(b"abcdef \n") for _ in range(r.randrange(0, 3000)))
            f = py_file(data, r.choice([zc.STORED, zc.DEFLATE]))
            if r.random() < 0.2:
                f["gp_flags"] |= 1 << 3  # a data-descriptor flag the writer must clear
            m, e = zc.member_make(path.replace(b"/", r.choice([b"/", b"\\"])), f, **kw)
        assert e is None
        z[m["path"]] = m
    return z


def host_archive_of(host, z):
    a = host.Archive()
    for p, m in z.items():
        f = m["kind"]
        if f is None:
            a.add_dir(p, mtime=m["mtime"], mode=m["mode"])
        else:
            a.add_file_made(p, f["compression"], f["compressed_bytes"], f["decompressed_size"], f["decompressed_crc_32"],
                            start=f["start"], compressed_size=f["compressed_size"], gp_flags=f["gp_flags"],
                            version_made_by=f["version_made_by"],
                            version_needed_to_extract=f["version_needed_to_extract"], mtime=m["mtime"], mode=m["mode"])
    return a


def same_members(host_arch, z):
    ms = host_arch.members()
    assert [m["path"] for m in ms] == sorted(z)
    for m in ms:
        o = z[m["path"]]
        assert (m["mode"], m["mtime"], bool(m["is_dir"])) == (o["mode"], o["mtime"], o["kind"] is None), m["path"]
        f = o["kind"]
        if f is not None:
            for k in ("compression", "gp_flags", "version_made_by", "version_needed_to_extract", "start",
                      "compressed_size", "decompressed_size", "decompressed_crc_32"):
                assert m[k] == f[k], (m["path"], k)
            assert bool(m["can_extract"]) == zc.file_can_extract(f) and bool(m["is_encrypted"]) == zc.file_is_encrypted(f)


# ---------------------------------------------------------------- host layer vs oracle
def test_host_ptime_and_fpath_equal_oracle(host):
    r = random.Random(1)
    times = [0, 1, 86399, 86400, 315532799, 315532800, 951782400, 1697900810, 4102444800, 4354819199, 4354819200,
             2 ** 33] + [r.randrange(0, 2 ** 33) for _ in range(2000)]
    for t in times:
        assert host.ptime_to_date_time(t) == zc.ptime_to_date_time(t)
        assert host.ptime_to_dos_date_time(t) == zc.ptime_to_dos_date_time(t)
        assert host.ptime_pp(t) == zc.ptime_to_string(t)
    for _ in range(3000):
        d, tm = r.randrange(0, 1 << 16), r.randrange(0, 1 << 16)
        assert host.ptime_of_dos_date_time(d, tm) == zc.ptime_of_dos_date_time(d, tm)
    # the two directions agree on every even second of the DOS range
    for t in (315532800, 1697900810, 4354819198):
        assert zc.ptime_of_dos_date_time(*zc.ptime_to_dos_date_time(t)) == t
    paths = [b"", b"a", b"a/", b"a\\b", b"../a/./b//c/..", b"\\\\x\\..\\y", b".", b"..", b"/abs/path", b"a//b\\\\c",
             b"dir/", b"./"]
    for p in paths:
        assert host.fpath_ensure_unix(p) == zc.fpath_ensure_unix(p)
        assert host.fpath_ensure_directoryness(p) == zc.fpath_ensure_directoryness(p)
        assert host.fpath_sanitize(p) == zc.fpath_sanitize(p)
    for m in (0, 0o644, 0o755, 0o777, 0o421, 0o7777):
        assert host.fpath_pp_mode(m) == zc.fpath_mode_string(m)


def test_host_parses_the_reference_fixture_like_the_oracle(host):
    s = fixture_bytes()
    z, _ = zc.of_binary_string(s)
    a = host.Archive.of_binary_string(s)
    assert a.member_count() == 3 and host.string_has_magic(s)
    same_members(a, z)
    for i, m in enumerate(a.members()):
        assert a.pp(i) == zc.member_to_string(z[m["path"]]) and a.pp(i, True) == zc.member_to_string(z[m["path"]], True)
    assert a.find(b"zip-docs/rfc1951.txt") == 2 and a.find(b"nope") is None
    # decode -> encode is what the oracle writes for the same members
    enc, _ = zc.to_binary_string(z)
    assert a.encoding_size() == len(enc) and a.to_binary_string() == enc


def test_host_encodes_random_archives_like_the_oracle(host):
    r = random.Random(2)
    for case in range(40):
        z = random_archive(r, r.randrange(0, 25), with_mimetype=case % 3 == 0)
        a = host_archive_of(host, z)
        same_members(a, z)
        for first in (None, b"mimetype", sorted(z)[-1] if z else b"x", b"absent"):
            enc, e = zc.to_binary_string(z, first if first is not None else b"mimetype")
            assert e is None
            got = a.to_binary_string(first)
            assert got == enc, (case, first)
        enc, _ = zc.to_binary_string(z)
        # and back: both parsers see the same members in what was written
        z2, e = zc.of_binary_string(enc)
        assert e is None
        same_members(host.Archive.of_binary_string(enc), z2)
        with zipfile.ZipFile(io.BytesIO(enc)) as zf:
            assert zf.testzip() is None
            assert sorted(i.filename.encode() for i in zf.infolist()) == sorted(z)
        if z:
            assert enc[:4] == (b"PK\x03\x04") and host.string_has_magic(enc)
            if b"mimetype" in z:
                assert enc[30:38] == b"mimetype"


def test_host_empty_archive_and_member_rules(host):
    a = host.Archive()
    assert a.to_binary_string() == zc.to_binary_string({})[0] == b"PK\x05\x06" + b"\0" * 18
    # Member.make: unix separators, directoryness, default modes, mtime floor (zipc.ml:247-260)
    a.add_dir(b"d\\e")
    a.add_dir(b"")
    a.add_file_made(b"f\\g", 0, b"hi", 2, zlib.crc32(b"hi"), mtime=5)
    ms = {m["path"]: m for m in a.members()}
    assert sorted(ms) == [b"./", b"d/e/", b"f/g"]
    assert ms[b"d/e/"]["mode"] == 0o755 and ms[b"f/g"]["mode"] == 0o644
    assert ms[b"f/g"]["mtime"] == zc.DOS_EPOCH == ms[b"./"]["mtime"]
    a.remove(b"./")
    assert a.member_count() == 2
    # a later add with the same path replaces (String_map.add)
    a.add_file_made(b"f/g", 0, b"ho!", 3, zlib.crc32(b"ho!"))
    assert a.member_count() == 2 and a.member(a.find(b"f/g"))["decompressed_size"] == 3
    # errors: the reference's messages
    with pytest.raises(host.ZipcError) as ei:
        a.add_dir(b"x" * 70000)
    assert ei.value.code == host.ERROR and ei.value.msg == zc.member_make(b"x" * 70000, None)[1]
    with pytest.raises(host.ZipcError) as ei:
        a.add_file_made(b"big", 8, b"abc", 5, 0, compressed_size=1 << 32)
    assert ei.value.code == host.ERROR and ei.value.msg == zc.err_size(1 << 32, 5)
    with pytest.raises(host.ZipcError) as ei:
        a.add_file_made(b"neg", 8, b"abc", -1, 0)
    assert ei.value.code == host.INVALID


def test_host_rejects_damaged_archives_with_the_reference_messages(host):
    r = random.Random(3)
    base, _ = zc.to_binary_string(random_archive(r, 6))
    good = fixture_bytes()
    cases = [b"", b"PK", base[:21], base[:-1], base[1:], base[:len(base) // 2]]
    eocd = len(base) - 22
    for off, val in ((4, 0xFFFF), (4, 1), (6, 1)):  # ZIP64 marker, multipart (this disk / directory disk)
        b = bytearray(base)
        struct.pack_into("<H", b, eocd + off, val)
        cases.append(bytes(b))
    for off, fmt, val in ((12, "<I", 0xFFFFFF), (16, "<I", len(base)), (10, "<H", 500), (12, "<I", 3), (16, "<I", 0)):
        b = bytearray(base)  # directory size / offset / count that do not fit
        struct.pack_into(fmt, b, eocd + off, val)
        cases.append(bytes(b))
    one = zc.to_binary_string({b"f": zc.member_make(b"f", py_file(b"payload" * 9, zc.DEFLATE))[0]})[0]
    for off, fmt, val in ((0, "<I", 0x04034b51), (26, "<H", 60000), (28, "<H", 65535)):  # local header damage
        b = bytearray(one)
        struct.pack_into(fmt, b, off, val)
        cases.append(bytes(b))
    for _ in range(300):  # byte smashes and truncations anywhere
        b = bytearray(r.choice([base, good]))
        k = r.randrange(3)
        if k == 0:
            for _ in range(r.randrange(1, 4)):
                b[r.randrange(len(b))] = r.randrange(256)
        elif k == 1:
            i = r.randrange(len(b))
            b[i] ^= 1 << r.randrange(8)
        else:
            b = b[:r.randrange(len(b))]
        cases.append(bytes(b))
    seen = set()
    for s in cases:
        z, e = zc.of_binary_string(s)
        if e is None:
            same_members(host.Archive.of_binary_string(s), z)
        else:
            with pytest.raises(host.ZipcError) as ei:
                host.Archive.of_binary_string(s)
            assert ei.value.code == host.ERROR and ei.value.msg == e, s[:8]
            seen.add(e)
    assert {"File too short to be a ZIP archive", "ZIP64 archives are not supported",
            "Multipart archives are not supported", "Corrupted end of central directory record",
            "Likely not a ZIP archive: no end of central directory record found", "Truncated central directory",
            "Corrupted central directory file header", "Corrupted local file header"} <= seen, seen


def test_host_decode_rules_crc_from_local_header_and_dos_directories(host):
    # zipc.ml:382-385: CRC 0 in the directory -> the local header's; zipc.ml:360-369: no unix mode -> DOS dir bit
    data = b"streamed member"
    f = py_file(data, zc.DEFLATE)
    z = {b"s.bin": zc.member_make(b"s.bin", f)[0], b"d/": zc.member_make(b"d", None)[0]}
    enc = bytearray(zc.to_binary_string(z)[0])
    cd = struct.unpack_from("<I", enc, len(enc) - 22 + 16)[0]
    pos = cd
    while enc[pos:pos + 4] == b"PK\x01\x02":
        n = struct.unpack_from("<H", enc, pos + 28)[0]
        name = bytes(enc[pos + 46:pos + 46 + n])
        struct.pack_into("<H", enc, pos + 40, 0)  # made by DOS: no unix permissions
        if name == b"s.bin":
            struct.pack_into("<I", enc, pos + 16, 0)
        pos += 46 + n
    enc = bytes(enc)
    zo, e = zc.of_binary_string(enc)
    assert e is None and zo[b"s.bin"]["kind"]["decompressed_crc_32"] == zlib.crc32(data)
    assert zo[b"d/"]["kind"] is None and zo[b"d/"]["mode"] == 0o755 and zo[b"s.bin"]["mode"] == 0o644
    same_members(host.Archive.of_binary_string(enc), zo)
