"""zipc_hip_deflate_many / zipc_hip_inflate_many (include/zipc_hip.h) on batches big and
ragged enough that the call runs as a pipeline of several sub-batches (api.hip many_streams:
4 of them from 4096 streams on, the first and the last half as large as the others, each
through gather, copy in, kernels, the way back -- the outputs end to end, written into the
pinned buffer by a kernel -- and scatter on its own queues, events and host threads; a
stream of more than a MiB is moved in pieces).  Every stream's bytes, length, checksum and
status against the oracle; guard bytes behind every destination.  The archive-level tests (test_gpu_zipc.py) only reach these entry points
with a handful of members."""
import ctypes as C
import random
import zlib

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

GUARD = 16
ERR_DST_TOO_SMALL = 16


def _ragged_inputs(n, seed):
    rnd = random.Random(seed)
    big_at = {rnd.randrange(n): rnd.randrange(100_000, 300_000) for _ in range(6)}  # uneven chunks
    big_at[rnd.randrange(n // 2, n)] = 2_400_000 + rnd.randrange(1000)  # (several pieces for the host threads)
    out = []
    for i in range(n):
        if i in big_at:
            ln = big_at[i]
        elif i % 50 == 7:
            ln = 0
        else:
            ln = rnd.randrange(1, 2000)
        kind = rnd.randrange(4)
        if kind == 0:
            d = util.rand_bytes(ln, seed * 100003 + i)
        elif kind == 1:
            d = util.rand_bytes(ln, seed * 100003 + i, 3)
        elif kind == 2:
            k = rnd.randrange(1, 40)
            d = (util.rand_bytes(k, i) * (ln // k + 1))[:ln]
        else:
            d = util.text(ln, i)
        assert len(d) == ln
        out.append(bytes(d))
    return out


class _Bufs:
    """n caller-owned destination buffers with guard bytes behind their capacity."""

    def __init__(self, caps):
        self.caps = list(caps)
        self.arr = [np.full(c + GUARD, 0xA5, np.uint8) for c in self.caps]
        n = len(self.caps)
        self.ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in self.arr])
        self.cap = (C.c_size_t * n)(*self.caps)

    def bytes(self, i, ln):
        return self.arr[i][:ln].tobytes()

    def guards_intact(self):
        return all(bool((a[c:] == 0xA5).all()) for a, c in zip(self.arr, self.caps))


def _srcs(datas):
    keep = [np.frombuffer(d, np.uint8) if d else np.zeros(1, np.uint8) for d in datas]
    n = len(datas)
    return keep, (C.c_void_p * n)(*[a.ctypes.data for a in keep]), (C.c_size_t * n)(*[len(d) for d in datas])


def test_many_streams_chunked_staging_equals_oracle(gpu_ctx, oracle):
    from zipc_amd import _lib

    lib = _lib.lib()
    n = 4500
    assert n // 4 >= 1024  # many_streams() keeps its default of 4 sub-batches only if each holds >= 1024 streams
    plain = _ragged_inputs(n, 5)
    level = 2

    # ---- deflate: one stream gets a destination that cannot hold its output
    small = next(i for i, d in enumerate(plain) if 500 < len(d) < 2000 and i > 2300)
    caps = [int(lib.zipc_hip_deflate_bound(len(d))) for d in plain]
    caps[small] = 3
    dst = _Bufs(caps)
    keep, sp, sl = _srcs(plain)
    res = (_lib.StreamResult * n)()
    assert lib.zipc_hip_deflate_many(gpu_ctx.handle, n, sp, sl, level, 1, dst.ptrs, dst.cap, res) == 0
    assert dst.guards_intact()
    comp = []
    for i, d in enumerate(plain):
        if i == small:
            assert (int(res[i].status), int(res[i].out_len)) == (ERR_DST_TOO_SMALL, 0), i
            assert bool((dst.arr[i] == 0xA5).all())  # nothing is copied for a failed stream
            comp.append(oracle.deflate(d, level=level)[1])
            continue
        st, want, crc = oracle.deflate(d, level=level, crc_op=oracle.CRC_CRC32)
        assert st == 0
        assert int(res[i].status) == 0 and int(res[i].out_len) == len(want), (i, len(d))
        assert dst.bytes(i, len(want)) == want, (i, len(d))
        assert int(res[i].checksum) == crc == zlib.crc32(d), i
        comp.append(want)

    # ---- inflate: a corrupted stream, a truncated one, a stream whose limit is one byte short.
    # The victims are streams that really compressed (Huffman blocks: flipping bytes of a
    # stored block only changes the data), and the oracle must reject what was made of them.
    rnd = random.Random(9)
    huff = [i for i in range(n) if len(comp[i]) > 120 and len(comp[i]) < 0.8 * len(plain[i])]
    bad = next(i for i in huff if 300 < i < 600)       # in the first sub-batch (a sixth of the bytes)
    for attempt in range(50):
        c = bytearray(comp[bad])
        for k in range(4, min(60, len(c))):
            c[k] ^= rnd.randrange(1, 256)
        if oracle.inflate(bytes(c), decompressed_size=len(plain[bad]))[0] != 0:
            break
    comp[bad] = bytes(c)
    trunc = next(i for i in huff if 1700 < i < 2100)   # in a middle one
    comp[trunc] = comp[trunc][: len(comp[trunc]) // 2]
    short = next(i for i, d in enumerate(plain) if len(d) > 100 and i > 4400)  # in the last
    limits = [len(d) for d in plain]
    limits[short] -= 1
    expect_fail = {bad, trunc, short}
    for i in expect_fail:  # the failure paths are really exercised
        assert oracle.inflate(comp[i], decompressed_size=limits[i])[0] != 0, i
    out = _Bufs(limits)
    keep2, cp, cl = _srcs(comp)
    lim = (C.c_size_t * n)(*limits)
    ires = (_lib.StreamResult * n)()
    assert lib.zipc_hip_inflate_many(gpu_ctx.handle, n, cp, cl, lim, 1, out.ptrs, out.cap, ires) == 0
    assert out.guards_intact()
    for i, d in enumerate(plain):
        st, want, crc = oracle.inflate(comp[i], decompressed_size=limits[i], crc_op=oracle.CRC_CRC32)
        assert int(ires[i].status) == st, (i, st, int(ires[i].status))
        assert (st != 0) == (i in expect_fail), i
        if st != 0:
            assert int(ires[i].out_len) == 0, i
            continue
        assert int(ires[i].out_len) == len(want) and out.bytes(i, len(want)) == want == d, i
        assert int(ires[i].checksum) == crc, i
    assert int(ires[short].status) == 2  # "Expected decompression size exceeded"
    # ---- the same call for its results alone (zipc_hip_inflate_many_check: `zipc unzip -t`): the same statuses, lengths and
    # checksums, no destination at all
    cres = (_lib.StreamResult * n)()
    assert lib.zipc_hip_inflate_many_check(gpu_ctx.handle, n, cp, cl, lim, 1, out.cap, cres) == 0
    for i in range(n):
        assert (int(cres[i].status), int(cres[i].out_len), int(cres[i].checksum)) == (int(ires[i].status), int(ires[i].out_len), int(ires[i].checksum)), i


def test_a_few_long_members_inflate_by_blocks(gpu_ctx, oracle):
    """zipc_hip_inflate_many with a handful of long members (an archive of a few big files): each goes by a wave per
    block (api.hip zipc_hip_inflate_batch), the damaged one and the one over its limit through the stream's one wave
    -- bytes, lengths, CRC-32 and statuses against the oracle, guard bytes intact"""
    from zipc_amd import _lib

    lib = _lib.lib()
    rng = np.random.default_rng(3)
    text = util.text(1_300_000, 4)
    plain = [bytes(text), (rng.integers(0, 16, 1_500_000, dtype=np.uint8) * 17).astype(np.uint8).tobytes(),
             (rng.integers(0, 1 << 14, 300_000, dtype=np.uint32) * np.uint32(0x10001)).tobytes(), bytes(text[::-1])]
    comp = []
    for i, d in enumerate(plain):
        c = zlib.compressobj(6 if i % 2 else 1, zlib.DEFLATED, -15)
        comp.append(c.compress(d) + c.flush())
    c3 = bytearray(comp[3]); c3[len(c3) // 2] ^= 0x10; comp[3] = bytes(c3)  # damaged (or merely different: the oracle says)
    limits = [len(d) for d in plain]
    limits[2] -= 1
    n = len(plain)
    out = _Bufs(limits)
    keep, cp, cl = _srcs(comp)
    lim = (C.c_size_t * n)(*limits)
    ires = (_lib.StreamResult * n)()
    assert lib.zipc_hip_inflate_many(gpu_ctx.handle, n, cp, cl, lim, 1, out.ptrs, out.cap, ires) == 0
    assert out.guards_intact()
    for i in range(n):
        st, want, crc = oracle.inflate(comp[i], decompressed_size=limits[i], crc_op=oracle.CRC_CRC32)
        assert int(ires[i].status) == st, (i, st, int(ires[i].status))
        if st == 0:
            assert int(ires[i].out_len) == len(want) and out.bytes(i, len(want)) == want and int(ires[i].checksum) == crc, i
        else:
            assert int(ires[i].out_len) == 0, i
    assert int(ires[0].status) == 0 and int(ires[1].status) == 0 and int(ires[2].status) == 2


def test_a_few_long_members_deflate(gpu_ctx, oracle):
    """zipc_hip_deflate_many with a handful of long members: one sub-batch, every member gathered and scattered in
    pieces of a MiB, outputs of megabytes end to end on the way back -- bytes, lengths and CRC-32 against the oracle"""
    from zipc_amd import _lib

    lib = _lib.lib()
    rng = np.random.default_rng(11)
    text = util.text(2_200_000, 6)
    plain = [bytes(text), (rng.integers(0, 8, 3_000_001, dtype=np.uint8) * 31).astype(np.uint8).tobytes(), b"",
             rng.integers(0, 256, 1_100_000, dtype=np.uint8).tobytes(), bytes(text[:77])]
    n = len(plain)
    caps = [int(lib.zipc_hip_deflate_bound(len(d))) for d in plain]
    dst = _Bufs(caps)
    keep, sp, sl = _srcs(plain)
    res = (_lib.StreamResult * n)()
    assert lib.zipc_hip_deflate_many(gpu_ctx.handle, n, sp, sl, 2, 1, dst.ptrs, dst.cap, res) == 0
    assert dst.guards_intact()
    for i, d in enumerate(plain):
        st, want, crc = oracle.deflate(d, level=2, crc_op=oracle.CRC_CRC32)
        assert st == 0 and int(res[i].status) == 0 and int(res[i].out_len) == len(want), (i, len(d))
        assert dst.bytes(i, len(want)) == want and int(res[i].checksum) == crc, i


def test_the_copy_engine_as_the_way_back():
    """ZIPC_HIP_HOST_PACK=0 (csrc/tuning.h: a sub-batch's whole destination slots come back by the copy engine): the
    same tests in a process of its own, since the switch is read once per process"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, ZIPC_HIP_HOST_PACK="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k", "not way_back"], env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_one_huge_member_among_hundreds_of_small_ones(gpu_ctx, oracle):
    """An archive's ragged members in one call: 400 files of a few KiB to 100 KiB and ONE of 256 MiB (a library directory:
    tools/corpus_box.py met it on /opt/rocm/lib).  The many-wave parse sized its segment buffers by streams x the LONGEST
    stream's segments -- 401 x 15 625 x 66 KiB = 425 GB -- and the call failed with "out of memory"; the slots are compact
    now (deflate.hip ParseSegs::slot).  Every stream round-trips through zlib with the right CRC-32; the small ones and the
    giant's length are the oracle's."""
    from zipc_amd import _lib

    lib = _lib.lib()
    rnd = random.Random(41)
    plain = []
    for i in range(400):
        ln = rnd.randrange(2000, 100_000)
        plain.append(util.text(ln, i) if i % 3 else util.rand_bytes(ln, i, 4))
    piece = util.text(1 << 20, 77) + util.rand_bytes(1 << 20, 78, 3) + bytes(1 << 19) + util.rand_bytes(1 << 19, 79)
    giant = (piece * ((256 << 20) // len(piece) + 1))[: (256 << 20) + 12345]
    at = 137
    plain.insert(at, giant)
    n = len(plain)
    caps = [int(lib.zipc_hip_deflate_bound(len(d))) for d in plain]
    dst = _Bufs(caps)
    keep, sp, sl = _srcs(plain)
    res = (_lib.StreamResult * n)()
    st = lib.zipc_hip_deflate_many(gpu_ctx.handle, n, sp, sl, 2, 1, dst.ptrs, dst.cap, res)
    assert st == 0, (st, lib.zipc_hip_last_error(gpu_ctx.handle))
    assert dst.guards_intact()
    for i, d in enumerate(plain):
        assert int(res[i].status) == 0, i
        c = dst.bytes(i, int(res[i].out_len))
        assert zlib.decompress(c, -15) == d and int(res[i].checksum) == zlib.crc32(d), i
        if i % 57 == 0 and i != at:
            assert c == oracle.deflate(d, level=2)[1], i
    st0, c0, _ = oracle.deflate(giant, level=2)
    assert st0 == 0 and dst.bytes(at, int(res[at].out_len)) == c0
