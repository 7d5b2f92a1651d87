"""SURVEY 8(f) rows 1-2 on the GPU: the host layer's codec-backed operations
(File.deflate_of_binary_string, stored_of_binary_string, to_binary_string[_no_crc_check],
and the batch forms) against the container oracle + deflate oracle, through the C
interface of include/zipc_host.h.  Reads like the reference's test_crunched_trip
(test/test.ml:57-118)."""
import io
import os
import random
import shutil
import struct
import subprocess
import zipfile
import zlib

import numpy as np
import pytest

import util
from oracle import zipc_container as zc
from test_zipc_container import FIXTURE, assert_zip_docs, fixture_bytes, py_file, same_members

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host(gpu_ctx):
    from zipc_amd import zipc_host

    zipc_host.lib()
    return zipc_host


def oracle_recode(z, oracle, level=None):
    z2 = dict(z)
    for p, m in z.items():
        f = m["kind"]
        if f is None or not zc.file_can_extract(f):
            continue
        s, e = zc.file_to_binary_string(f, oracle)
        assert e is None
        nf, e = zc.file_deflate_of_binary_string(s, oracle, level)
        assert e is None
        z2[p], _ = zc.member_make(p, nf, mode=m["mode"], mtime=m["mtime"])
    return z2


def test_crunched_trip_like_reference(host, oracle):  # test/test.ml:57-118
    s = fixture_bytes()
    a = host.Archive.of_binary_string(s)
    z, _ = zc.of_binary_string(s)
    same_members(a, z)
    # unzip: every file member, CRC checked
    for i, m in enumerate(a.members()):
        if m["is_dir"]:
            continue
        data, crc = a.member_to_binary_string(i)
        want, e = zc.file_to_binary_string(z[m["path"]]["kind"], oracle)
        assert e is None and data == want and crc == m["decompressed_crc_32"]
    # redeflate_recode: File.to_binary_string |> File.deflate_of_binary_string (default level), same mtime/mode
    b = host.Archive()
    for i, m in enumerate(a.members()):
        if m["is_dir"]:
            b.add_dir(m["path"], mtime=m["mtime"], mode=m["mode"])
        else:
            b.add_file_deflate(m["path"], a.member_to_binary_string(i)[0], mtime=m["mtime"], mode=m["mode"])
    enc = b.to_binary_string()
    z2 = oracle_recode(z, oracle)
    assert enc == zc.to_binary_string(z2)[0]  # compressed bytes included: the GPU deflate is the reference's
    z3, e = zc.of_binary_string(enc)
    assert e is None
    assert_zip_docs(z3, True, oracle)
    with zipfile.ZipFile(io.BytesIO(enc)) as zf:
        assert zf.testzip() is None


def test_stored_and_deflate_members_at_every_level(host, oracle):
    r = random.Random(11)
    datas = [b"", b"a", b"hello hello hello hello", bytes(r.randrange(256) for _ in range(5000)),
             bytes(r.choice(b"ab") for _ in range(70000)), bytes(r.choice(b"0123456789abcdef") for _ in range(140000))]
    for level in (None, 0, 1, 2, 3):
        a = host.Archive()
        z = {}
        for k, d in enumerate(datas):
            p = b"m/%d.bin" % k
            a.add_file_deflate(p, d, level=level, mtime=1697900810 + k)
            f, e = zc.file_deflate_of_binary_string(d, oracle, level)
            assert e is None
            z[p], _ = zc.member_make(p, f, mtime=1697900810 + k)
            ps = b"s/%d.bin" % k
            a.add_file_stored(ps, d, mode=0o600)
            z[ps], _ = zc.member_make(ps, zc.file_stored_of_binary_string(d, oracle)[0], mode=0o600)
        same_members(a, z)
        enc = a.to_binary_string()
        assert enc == zc.to_binary_string(z)[0], level
        back = host.Archive.of_binary_string(enc)
        for i, m in enumerate(back.members()):
            data, crc = back.member_to_binary_string(i)
            assert data == datas[int(m["path"].split(b"/")[1].split(b".")[0])] and crc == zlib.crc32(data)
            assert back.member_to_binary_string(i, check_crc=False) == (data, crc)


def test_batch_forms_equal_single_member_forms(host, oracle):
    r = random.Random(12)
    files = []
    for k in range(97):
        n = r.choice([0, 1, 100, 4096, 65534, 65535, 65536, 70000, 200000]) if k < 20 else r.randrange(0, 30000)
        bits = r.choice([1, 2, 4, 8])
        files.append((b"batch/%03d" % k, bytes(r.randrange(1 << bits) for _ in range(n))))
    for level in (None, 2, 0):
        a, b = host.Archive(), host.Archive()
        a.add_files_deflate(files, level=level)  # one launch of the batch kernels
        for p, d in files:
            b.add_file_deflate(p, d, level=level)
        assert a.to_binary_string() == b.to_binary_string()
        z = {}
        for p, d in files[:12]:
            z[p] = zc.member_make(p, zc.file_deflate_of_binary_string(d, oracle, level)[0])[0]
        for m in a.members()[:12]:
            f = z[m["path"]]["kind"]
            assert (m["compressed_size"], m["decompressed_crc_32"]) == (f["compressed_size"], f["decompressed_crc_32"])
        got = a.extract_all()  # every member inflated + CRC-checked as one batch
        assert [p for p, _ in got] == sorted(p for p, _ in files)
        assert dict(got) == dict(files)
    with zipfile.ZipFile(io.BytesIO(a.to_binary_string())) as zf:
        assert zf.testzip() is None and len(zf.infolist()) == len(files)


def test_member_errors_are_the_references(host, oracle):
    r13 = random.Random(13)
    data = bytes(r13.choice(b"abcdefgh") for _ in range(20000))
    f = py_file(data, zc.DEFLATE)
    good = zc.member_make(b"ok", f)[0]
    wrong_crc = dict(f, decompressed_crc_32=f["decompressed_crc_32"] ^ 0x1234)
    damaged = dict(f, compressed_bytes=f["compressed_bytes"][:50] + b"\xff\xff\xff" + f["compressed_bytes"][53:])
    too_long = dict(f, decompressed_size=f["decompressed_size"] - 1)
    encrypted = dict(f, gp_flags=f["gp_flags"] | 1)
    bz2 = dict(f, compression=zc.BZIP2)
    other = dict(f, compression=99)
    z = {b"ok": good}
    for name, ff in ((b"wrong_crc", wrong_crc), (b"damaged", damaged), (b"too_long", too_long),
                     (b"encrypted", encrypted), (b"bz2", bz2), (b"other", other)):
        z[name] = zc.member_make(name, ff)[0]
    a = host.Archive.of_binary_string(zc.to_binary_string(z)[0])
    seen = {}
    for i, m in enumerate(a.members()):
        want, e = zc.file_to_binary_string(z[m["path"]]["kind"], oracle)
        if e is None:
            assert a.member_to_binary_string(i)[0] == want
        else:
            with pytest.raises(host.ZipcError) as ei:
                a.member_to_binary_string(i)
            assert ei.value.code == host.ERROR and ei.value.msg == e, m["path"]
            seen[m["path"]] = e
    assert seen[b"wrong_crc"].startswith("Checksum mismatch, expected ")
    assert seen[b"damaged"] in ("deflate: Corrupted data stream", "deflate: Expected decompression size exceeded") or \
        seen[b"damaged"].startswith("Checksum mismatch")
    assert seen[b"too_long"] == "deflate: Expected decompression size exceeded"
    assert seen[b"encrypted"] == "Encrypted file not supported"
    assert seen[b"bz2"] == "Compression bz2 not supported" and seen[b"other"] == "Compression 0099 not supported"
    # to_binary_string_no_crc_check returns the bytes and the CRC it found
    i = a.find(b"wrong_crc")
    got, crc = a.member_to_binary_string(i, check_crc=False)
    assert got == data and crc == zlib.crc32(data)
    # the batch form reports the same per member
    for p, r in a.extract_all():
        want, e = zc.file_to_binary_string(z[p]["kind"], oracle)
        if e is None:
            assert r == want
        else:
            assert isinstance(r, host.ZipcError) and r.msg == e, p


@pytest.mark.skipif(shutil.which("unzip") is None, reason="Info-ZIP unzip not installed")
def test_config4_like_archive_validates_with_unzip(host, tmp_path):
    """BASELINE config C4 in small: members of 3-bit symbols compressed as one batch at level
    `Default, assembled into a ZIP on the host, checked by Info-ZIP and zipfile"""
    from zipc_amd import synth

    n, size = 256, 1 << 18
    files = [(b"m/%05d.bin" % j, synth.stream_bytes_np(4, j, size, 3).tobytes()) for j in range(n)]
    a = host.Archive()
    a.add_files_deflate(files, level=2)
    enc = a.to_binary_string()
    ratio = sum(m["compressed_size"] for m in a.members()) / (n * size)
    assert 0.40 < ratio < 0.46  # ~0.43
    p = tmp_path / "c4.zip"
    p.write_bytes(enc)
    r = subprocess.run(["unzip", "-tq", str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    with zipfile.ZipFile(io.BytesIO(enc)) as zf:
        assert zf.read("m/00017.bin") == files[17][1]
    assert dict(host.Archive.of_binary_string(enc).extract_all()) == dict(files)


def test_members_sharded_over_devices_equal_one_device(host):
    """The member loop over several devices (zipc_host_set_devices; SURVEY 8(e): contiguous ranges of about
    equal bytes, one context and one host thread per device, nothing exchanged).  A box with one GPU names it
    three times -- three contexts, three threads, the same code path as three GPUs -- and the archive must be
    the one a single device writes, byte for byte; extraction likewise."""
    from zipc_amd import synth

    n, size = 192, 1 << 18  # 48 MiB: above the size from which a batch is spread
    files = [(b"m/%05d.bin" % j, synth.stream_bytes_np(4, j, size if j % 7 else size // 3, 3).tobytes()) for j in range(n)]
    try:
        host.set_devices([0])
        one = host.Archive()
        one.add_files_deflate(files, level=2)
        enc_one = one.to_binary_string()
        host.set_devices([0, 0, 0])
        assert host.devices() == [0, 0, 0]
        parts = host.partition([len(d) for _, d in files], 3)
        assert parts[0][0] == 0 and parts[-1][1] == n and all(hi > lo for lo, hi in parts)
        three = host.Archive()
        three.add_files_deflate(files, level=2)
        assert three.to_binary_string() == enc_one
        assert dict(host.Archive.of_binary_string(enc_one).extract_all()) == dict(files)
    finally:
        host.set_devices([])
    assert host.devices() == list(range(max(1, len(host.devices()))))


def test_command_line_like_the_reference_tool(host, oracle, tmp_path):
    """zipc-hip (zipc_amd/host/zipc_tool.cpp, after test/zipc_tool.ml): crc, compress / decompress,
    zip, list, unzip -t / extract, recode -- each checked against the oracle or an independent tool"""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zipc_amd", "bin", "zipc-hip")
    assert os.path.exists(tool), "run `make -C zipc_amd/host`"

    def run(*args, stdin=None):
        r = subprocess.run([tool, *args], input=stdin, capture_output=True)
        assert r.returncode == 0, (args, r.stderr.decode())
        return r.stdout

    r = random.Random(21)
    tree = tmp_path / "tree"
    (tree / "sub" / "deep").mkdir(parents=True)
    files = {"a.txt": b"hello zipc\n" * 500, "sub/b.bin": bytes(r.randrange(256) for _ in range(30000)),
             "sub/deep/c.dat": bytes(r.choice(b"acgt") for _ in range(200000)), "sub/empty": b""}
    for p, d in files.items():
        (tree / p).write_bytes(d)
    # crc / adler
    assert run("crc", str(tree / "a.txt")).strip() == b"%x" % zlib.crc32(files["a.txt"])
    assert run("crc", "-a", stdin=files["sub/b.bin"]).strip() == b"%x" % oracle.adler32(files["sub/b.bin"])
    # compress (default level = `Best like the reference) / decompress, raw and zlib
    comp = run("compress", str(tree / "sub/deep/c.dat"))
    assert comp == oracle.deflate(files["sub/deep/c.dat"], level=oracle.LEVEL_BEST)[1]
    assert run("decompress", stdin=comp) == files["sub/deep/c.dat"]
    z = run("compress", "--zlib", "--level", "fast", stdin=files["a.txt"])
    assert zlib.decompress(z) == files["a.txt"] and run("decompress", "--zlib", stdin=z) == files["a.txt"]
    # zip -> Info-ZIP / zipfile read it; list; unzip -t; extract
    arc = tmp_path / "t.zip"
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        run("zip", "--level", "default", "-o", str(arc), "tree")
    finally:
        os.chdir(cwd)
    with zipfile.ZipFile(arc) as zf:
        assert zf.testzip() is None
        assert {i.filename for i in zf.infolist() if not i.is_dir()} == {"tree/" + p for p in files}
        assert zf.read("tree/sub/b.bin") == files["sub/b.bin"]
    if shutil.which("unzip"):
        assert subprocess.run(["unzip", "-tq", str(arc)], capture_output=True).returncode == 0
    zo, e = zc.of_binary_string(arc.read_bytes())
    assert e is None
    assert run("list", "-l", str(arc)).decode().splitlines() == [zc.member_to_string(zo[p], True) for p in sorted(zo)]
    assert run("list", "-s", str(arc)).split() == sorted(zo)
    assert b"No errors detected" in run("unzip", "-t", str(arc))  # (Archive::test_all: only the verdicts come back from the GPU)
    out = tmp_path / "out"
    run("unzip", "-d", str(out), str(arc))
    for p, d in files.items():
        assert (out / "tree" / p).read_bytes() == d
    # recode at another level: same members, same bytes back (the default, --as-is, writes the members as they are)
    rec = tmp_path / "r.zip"
    run("recode", "--deflate", "--level", "fast", "-o", str(rec), str(arc))
    with zipfile.ZipFile(rec) as zf:
        assert zf.testzip() is None and zf.read("tree/a.txt") == files["a.txt"]
    assert os.path.getsize(rec) != os.path.getsize(arc)
    same = tmp_path / "same.zip"
    run("recode", "-o", str(same), str(arc))
    assert same.read_bytes() == arc.read_bytes()
    st = tmp_path / "stored.zip"
    run("recode", "-u", "-o", str(st), str(arc))
    with zipfile.ZipFile(st) as zf:
        assert zf.testzip() is None and all(i.compress_type == zipfile.ZIP_STORED for i in zf.infolist())
    # recode -t: nothing written, the recoded archive decoded again -- in memory, or by a command (test/zipc_tool.ml:485-545)
    assert run("recode", "--deflate", "-t", str(arc)) == b""
    if shutil.which("unzip"):
        assert run("recode", "--deflate", "--level", "default", "-t", "--check-cmd=unzip -P '' -q -q -t", str(arc)) == b""
    rr = subprocess.run([tool, "recode", "--deflate", "-t", "--check-cmd", "false", str(arc)], capture_output=True)
    assert rr.returncode == 123 and b"check command returned 1" in rr.stderr
    # a damaged archive member is reported by unzip -t: exit 2, "corrupted" (test/zipc_tool.ml:282-311)
    b = bytearray(arc.read_bytes())
    off = zo[b"tree/sub/deep/c.dat"]["kind"]["start"]
    b[off + 100] ^= 0x55
    bad = tmp_path / "bad.zip"
    bad.write_bytes(bytes(b))
    rr = subprocess.run([tool, "unzip", "-t", str(bad)], capture_output=True)
    assert rr.returncode == 2 and b"tree/sub/deep/c.dat" in rr.stderr
    # a member in a format the codec does not decode: exit 3 unless --skip (test/zipc_tool.ml:650-656)
    other = tmp_path / "bz.zip"
    with zipfile.ZipFile(other, "w") as zf:
        zf.writestr("plain.txt", files["a.txt"], compress_type=zipfile.ZIP_DEFLATED)
        zf.writestr("packed.bz2", files["a.txt"], compress_type=zipfile.ZIP_BZIP2)
    rr = subprocess.run([tool, "unzip", "-t", str(other)], capture_output=True)
    assert rr.returncode == 3 and b"packed.bz2: Cannot decompress format bz2" in rr.stderr
    rr = subprocess.run([tool, "unzip", "-t", "--skip", "-v", str(other)], capture_output=True)
    assert rr.returncode == 0 and b"[ OK ] plain.txt" in rr.stderr and b"[ ?? ] packed.bz2" in rr.stderr
    run("recode", "--deflate", "-t", str(other))  # the bzip2 member is kept as it is
    # sniff: the ZIP files below a directory, NUL separated; a file without the magic is exit 4 (test/zipc_tool.ml:556-579)
    got = run("sniff", "-0", "-r", "-P", str(tmp_path)).split(b"\0")
    assert got[-1] == b"" and {os.path.basename(g) for g in got[:-1]} == {b"t.zip", b"r.zip", b"same.zip", b"stored.zip", b"bad.zip", b"bz.zip"}
    assert run("sniff", str(arc)).strip() == str(arc).encode()
    assert run("sniff", str(tree)) == b""  # (no -r: a directory's own files only; the tree holds no archive)
    rr = subprocess.run([tool, "sniff", str(tree / "a.txt"), str(arc)], capture_output=True)
    assert rr.returncode == 4 and b"Not a ZIP archive" in rr.stderr and rr.stdout.strip() == str(arc).encode()


def test_real_files_at_every_level_equal_oracle(host, oracle):
    """text, source code, an ELF shared object and an already compressed file: what real
    archives hold (long matches, all 286 symbols in use, multi-block members with Q1
    active, stored/fixed/dynamic decisions) -- compressed bytes equal the oracle's"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["DESIGN.md", "SURVEY.md", "include/zipc_hip.h", "zipc_amd/csrc/deflate.hip", "oracle/libzd_oracle.so",
             "tests/golden/zip-docs.zip", "tests/golden/zlib_streams.json"]
    files = [(n.encode(), open(os.path.join(root, n), "rb").read()) for n in names]
    assert sum(len(d) for _, d in files) > 500000
    for level in (1, 2, 3):
        a = host.Archive()
        a.add_files_deflate(files, level=level)
        z = {p: zc.member_make(p, zc.file_deflate_of_binary_string(d, oracle, level)[0])[0] for p, d in files}
        assert a.to_binary_string() == zc.to_binary_string(z)[0], level
        assert dict(a.extract_all()) == dict(files)
