"""Runs of one match -- zeros, a short word repeated: 258 bytes per symbol -- which inflate_batch copies
as one periodic copy (match_run / wave_copy_match in inflate.hip) instead of a symbol at a time.
Bytes, accept/reject and checksums against the oracle, with the size limit falling inside, at the
end of, and just behind a run (Buf's "Expected decompression size exceeded", zd.ml:27-29), periods on
both sides of the run code's own threshold, and streams zlib wrote (other encoders, other codes)."""
import random
import zlib

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _period(word: bytes, n: int) -> bytes:
    return (word * (n // len(word) + 1))[:n]


def _inputs():
    rng = random.Random(11)
    out = {"zeros_1m": bytes(1 << 20), "zeros_70000": bytes(70000), "ff_5000": b"\xff" * 5000}
    for p in (1, 2, 3, 5, 15, 16, 17, 31, 63, 64, 65, 79, 80, 81, 100, 257, 258, 259, 300, 1000, 4097):
        out["period_%d" % p] = _period(bytes(rng.randrange(256) for _ in range(p)), 40000 + p)
    for p in (5000, 32767, 32768, 32769):  # (the last one: no match reaches back that far)
        out["period_%d" % p] = _period(bytes(rng.randrange(256) for _ in range(p)), 110000)
    # records that repeat with one byte changing: runs of one match broken by literals
    rec = bytes(rng.randrange(256) for _ in range(300))
    out["records"] = b"".join(rec[:150] + bytes([i & 255]) + rec[151:] for i in range(200))
    # runs between other things: literals, far matches, a second period
    mix = bytearray()
    for i in range(40):
        mix += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
        mix += _period(bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9))), rng.randrange(300, 9000))
        if i % 7 == 3:
            mix += mix[len(mix) // 3:len(mix) // 3 + 500]
    out["mixed"] = bytes(mix)
    return out


def _streams(oracle):
    for name, data in _inputs().items():
        for level in (1, 2, 3):
            st, comp, _ = oracle.deflate(data, level=level, crc_op=0)
            assert st == 0
            yield "%s/L%d" % (name, level), data, comp
        for zl in (1, 6, 9):
            co = zlib.compressobj(zl, zlib.DEFLATED, -15)
            yield "%s/zlib%d" % (name, zl), data, co.compress(data) + co.flush()


def test_runs_inflate_to_the_same_bytes(gpu_ctx, oracle):
    import torch

    from zipc_amd import batch

    cases = list(_streams(oracle))
    srcs = [c for _, _, c in cases]
    caps = [len(d) for _, d, _ in cases]
    for crc_op in (1, 2):
        got = util.gpu_inflate_batch(gpu_ctx, srcs, caps, [True] * len(cases), caps, crc_op)
        for i, (name, data, comp) in enumerate(cases):
            st0, d0, c0 = oracle.inflate(comp, decompressed_size=len(data), crc_op=crc_op)
            assert st0 == 0 and d0 == data, name
            assert got[i] == (0, data, c0), (name, crc_op, got[i][0])


def test_the_size_limit_inside_a_run(gpu_ctx, oracle):
    cases = [(n, d, c) for n, d, c in _streams(oracle) if n.split("/")[0] in ("zeros_70000", "period_3", "period_63", "mixed")
             and n.split("/")[1] in ("L2", "zlib6")]
    srcs, caps, lims, names = [], [], [], []
    for name, data, comp in cases:
        n = len(data)
        for lim in (n, n - 1, n + 1, n - 257, n - 258, n - 259, n // 2, 600, 259, 258, 257, 3, 1, 0):
            srcs.append(comp); lims.append(lim); caps.append(n + 64); names.append("%s limit %d" % (name, lim))
    got = util.gpu_inflate_batch(gpu_ctx, srcs, caps, [True] * len(srcs), lims, 2)
    for i, name in enumerate(names):
        st0, d0, c0 = oracle.inflate(srcs[i], decompressed_size=lims[i], crc_op=2)
        assert got[i][0] == st0, (name, got[i][0], st0)
        if st0 == 0:
            assert got[i][1] == d0 and got[i][2] == c0, name


def test_a_small_destination_under_a_run(gpu_ctx, oracle):
    # no limit given: the caller's dst_cap is the boundary's DST_TOO_SMALL, wherever in a run it falls
    data = bytes(20000)
    st, comp, _ = oracle.deflate(data, level=2, crc_op=0)
    caps = [20000, 19999, 19743, 19742, 258, 257, 8, 7, 1]
    got = util.gpu_inflate_batch(gpu_ctx, [comp] * len(caps), caps, [False] * len(caps), [0] * len(caps), 1)
    assert got[0][0] == 0 and got[0][1] == data
    for g in got[1:]:
        assert g[0] != 0


def test_more_holes_than_a_tile_lists(gpu_ctx, oracle):
    # a span's tile lists at most 1024 holes (SPAN_LIST_MAX): 1365 three-byte matches fit in 4 KiB
    cases = []
    for length, dist, n, seed in ((3, 3, 30000, None), (3, 1, 30000, None), (4, 3, 20000, None), (3, 4, 5000, None), (5, 2, 12000, None),
                                  (3, 4, 30000, 1), (4, 4, 30000, 2), (4, 2, 20000, 3), (10, 4, 20000, 4)):
        comp, plain = util.fixed_block_of_short_matches(n, length, dist, seed=seed)
        assert zlib.decompress(comp, -15) == plain
        st0, d0, c0 = oracle.inflate(comp, decompressed_size=len(plain), crc_op=2)
        assert st0 == 0 and d0 == plain
        cases.append((comp, plain, c0))
    got = util.gpu_inflate_batch(gpu_ctx, [c for c, _, _ in cases], [len(p) for _, p, _ in cases], [True] * len(cases),
                                 [len(p) for _, p, _ in cases], 2)
    for i, (comp, plain, c0) in enumerate(cases):
        assert got[i] == (0, plain, c0), i


def test_random_hand_made_blocks(gpu_ctx, oracle):
    """blocks of random literals and matches at every mix of short and long lengths, near and far
    distances (util.random_fixed_block): dense near matches, chains of them, runs -- the shapes the
    hole filling and the run copy have special paths for, at random"""
    cases = []
    for seed in range(96):
        comp, plain = util.random_fixed_block(seed, 12000, max_dist=[1, 3, 4, 16, 64, 300, 5000, 32768][seed % 8],
                                              max_len=[3, 4, 10, 40, 258][(seed // 8) % 5], lit_share=[0.0, 0.05, 0.5][seed % 3])
        assert zlib.decompress(comp, -15) == plain
        cases.append((comp, plain))
    got = util.gpu_inflate_batch(gpu_ctx, [c for c, _ in cases], [len(p) for _, p in cases], [True] * len(cases),
                                 [len(p) for _, p in cases], 1)
    for i, (comp, plain) in enumerate(cases):
        assert got[i][0] == 0 and got[i][1] == plain and got[i][2] == zlib.crc32(plain), i
    # and with the size limit inside: same accept / reject as the oracle
    lims = [len(p) - 1 - (i * 37) % 300 for i, (_, p) in enumerate(cases)]
    got = util.gpu_inflate_batch(gpu_ctx, [c for c, _ in cases], [len(p) for _, p in cases], [True] * len(cases), lims, 1)
    for i, (comp, plain) in enumerate(cases):
        st0, _, _ = oracle.inflate(comp, decompressed_size=lims[i])
        assert got[i][0] == st0 != 0, i


def test_a_repeated_header_longer_than_what_stays_staged(gpu_ctx, oracle):
    """Calls of up to 256 streams go on with the tables there are when a block repeats the header bits of the short
    block before it (inflate.hip REUSE): the header's bits are compared in memory and skipped, so the position may
    pass what the input ring holds -- with a header of a few hundred bits it does, and the ring must start over
    there (round 3's kernel decoded stale slots).  Headers of several lengths, short and long blocks, batches of 1,
    3, 64 and 256 streams, all three checksum modes."""
    cases = []
    for seed, (n_blocks, n_used, per_block) in enumerate([(6, 2, 40), (5, 20, 10), (9, 60, 3), (4, 100, 200), (12, 200, 1),
                                                          (7, 255, 30), (3, 150, 1500), (20, 90, 7)]):
        comp, plain = util.repeated_literal_blocks(1000 + seed, n_blocks, n_used, per_block)
        assert zlib.decompress(comp, -15) == plain
        cases.append((comp, plain))
    for n in (1, 3, 64, 256):
        srcs = [cases[i % len(cases)][0] for i in range(n)]
        plains = [cases[i % len(cases)][1] for i in range(n)]
        for crc_op in (0, 1, 2):
            got = util.gpu_inflate_batch(gpu_ctx, srcs, [len(p) for p in plains], [True] * n, [len(p) for p in plains], crc_op)
            for i in range(n):
                st0, d0, c0 = oracle.inflate(srcs[i], decompressed_size=len(plains[i]), crc_op=crc_op)
                assert st0 == 0 and d0 == plains[i]
                assert got[i] == (0, plains[i], c0), (n, i, crc_op, got[i][0])


def _one_length_streams():
    """payloads whose literals nearly all have ONE code length (base64: 64 letters, 6 bits; hex: 16, 4 bits) -- what the
    strided turn of the one-wave inflate takes (inflate.hip strided_turn, round 5) -- cut into many blocks so that strides
    run up to, across and from block boundaries: sync flushes every few KiB, Huffman-only and default strategies, and
    stretches of text and zeros between them (a stride must stop at the first symbol of another length)."""
    import base64

    rnd = random.Random(61)
    out = []
    for k, (n, flush, strategy) in enumerate([(60_000, 3000, zlib.Z_HUFFMAN_ONLY), (48_000, 0, zlib.Z_DEFAULT_STRATEGY),
                                              (200_000, 7000, zlib.Z_DEFAULT_STRATEGY), (33_000, 500, zlib.Z_HUFFMAN_ONLY),
                                              (150_000, 4096, zlib.Z_FIXED), (90_000, 11_000, zlib.Z_HUFFMAN_ONLY)]):
        raw = rnd.randbytes(n)
        b64, hx = base64.b64encode(raw), raw.hex().encode()
        plain = b64[: n // 2] + util.text(700 + k, k) + hx[: n // 3] + bytes(300) + b64[n // 2:n] + b"=" * (k % 3)
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, strategy)
        comp = b""
        if flush:
            for i in range(0, len(plain), flush):
                comp += c.compress(plain[i:i + flush]) + c.flush(zlib.Z_SYNC_FLUSH if (i // flush) % 3 else zlib.Z_FULL_FLUSH)
        else:
            comp = c.compress(plain)
        out.append((plain, comp + c.flush()))
    return out


def test_strides_of_one_code_length_across_blocks_limits_and_cuts(gpu_ctx, oracle):
    """the review of round 5: "strided_turn is a new decode path that exists only in the device build ... add a directed GPU
    test with base64 or hex payloads that cross block boundaries in each mode the function is compiled for (IM_REAL, IM_TOKEN
    via the block path, limit / cap cuts mid-stride, input ending inside a stride)"."""
    import ctypes as C

    from zipc_amd import _lib
    from zipc_amd import zipc_deflate as Z

    lib = _lib.lib()
    streams = _one_length_streams()
    for plain, comp in streams:
        assert zlib.decompress(comp, -15) == plain
    # ---- ONE stream per call: the long ones go by blocks (IM_DRY, IM_TOKEN), the short ones by their one wave (IM_REAL)
    by_blocks = 0
    for plain, comp in streams:
        for crc_op in (oracle.CRC_CRC32, oracle.CRC_ADLER32):
            st, want, k = oracle.inflate(comp, decompressed_size=len(plain), crc_op=crc_op)
            got, kk = (Z.inflate_and_crc_32 if crc_op == oracle.CRC_CRC32 else Z.inflate_and_adler_32)(comp, decompressed_size=len(plain)).get_ok()
            assert st == 0 and got == want == plain and kk == k
        by_blocks += int(gpu_ctx.last_inflate_blocks() >= 2)
    assert by_blocks >= 2
    # ---- calls of many streams (inflate_batch_kernel: more than 256; the few-streams form: 40): whole streams, limits that
    # end a stream inside a stride, inputs that end inside one, destinations a byte short
    rnd = random.Random(62)
    for n in (300, 40):
        comps, limits, caps = [], [], []
        for i in range(n):
            plain, comp = streams[i % len(streams)]
            kind = i % 5
            if kind == 1:
                limits.append(rnd.randrange(1, len(plain)))  # the size limit falls somewhere inside
                comps.append(comp)
            elif kind == 2:
                limits.append(len(plain))
                comps.append(comp[: rnd.randrange(8, len(comp) - 1)])  # the input ends early
            else:
                limits.append(len(plain))
                comps.append(comp)
            caps.append(limits[-1])
        keep = [np.frombuffer(c, np.uint8) for c in comps]
        outs = [np.full(c + 16, 0xA5, np.uint8) for c in caps]
        P, S = C.c_void_p * n, C.c_size_t * n
        res = (_lib.StreamResult * n)()
        assert lib.zipc_hip_inflate_many(gpu_ctx.handle, n, P(*[a.ctypes.data for a in keep]), S(*[len(c) for c in comps]), S(*limits), 1,
                                         P(*[a.ctypes.data for a in outs]), S(*caps), res) == 0
        failed = 0
        for i in range(n):
            st, want, crc = oracle.inflate(comps[i], decompressed_size=limits[i], crc_op=oracle.CRC_CRC32)
            assert int(res[i].status) == st, (n, i, st, int(res[i].status))
            assert bool((outs[i][caps[i]:] == 0xA5).all()), (n, i)
            if st != 0:
                failed += 1
                assert int(res[i].out_len) == 0
                continue
            assert int(res[i].out_len) == len(want) and outs[i][:len(want)].tobytes() == want and int(res[i].checksum) == crc, (n, i)
        assert failed >= n // 4
