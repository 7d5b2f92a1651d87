"""The lane-serial halves of the kernels (zipc_amd/csrc/*_lane.h) compiled with
g++ and driven on the CPU, against the oracle.  This checks the kernel LOGIC on
the build box; the kernels themselves are checked on the GPU (test_gpu_*.py).
No GPU, and nothing here is a product path."""
import ctypes as C
import random

import pytest

import util
from host_sim import lib as sim_lib


@pytest.fixture(scope="module")
def sim():
    return sim_lib()


def sim_inflate(sim, raw, cap, limit=None, crc_op=0, budget=512):
    dst = C.create_string_buffer(max(cap, 1) + 64)
    ol, ck = C.c_uint64(), C.c_uint32()
    st = sim.sim_inflate(raw, len(raw), dst, cap, int(limit is not None), limit or 0, crc_op,
                         C.byref(ol), C.byref(ck), budget)
    return st, dst.raw[:ol.value], ck.value


def sim_deflate(sim, oracle, data, level):
    cap = oracle.deflate_bound(len(data))
    dst = C.create_string_buffer(cap)
    ol, ad, kinds, nk = C.c_uint64(), C.c_uint32(), (C.c_int * 64)(), C.c_int()
    st = sim.sim_deflate(data, len(data), level, dst, cap, C.byref(ol), C.byref(ad), kinds, 64, C.byref(nk))
    return st, dst.raw[:ol.value], ad.value, list(kinds[:min(nk.value, 64)])


def test_inflate_lane_golden_streams(sim, oracle):
    for s in util.zlib_streams():
        st0, d0, a0 = oracle.inflate(s["raw"], crc_op=oracle.CRC_ADLER32)
        for budget in (512, 7):
            st, d, a = sim_inflate(sim, s["raw"], s["plain_len"] + 100, crc_op=2, budget=budget)
            assert (st, d, a) == (st0, d0, a0), s["name"]
        for lim in (s["plain_len"], s["plain_len"] + 1, max(0, s["plain_len"] - 1)):
            st0, d0, _ = oracle.inflate(s["raw"], decompressed_size=lim)
            st, d, _ = sim_inflate(sim, s["raw"], lim, limit=lim)
            assert (st, d) == (st0, d0), (s["name"], lim)


def test_inflate_lane_accept_reject_fuzz(sim, oracle):
    seen = {}
    for i, s in enumerate(util.zlib_streams()):
        if len(s["raw"]) > 40000:
            continue
        cap = s["plain_len"] * 2 + 1000
        for r in util.corrupt_variants(s["raw"], i, 60):
            st0, d0, a0 = oracle.inflate(r, decompressed_size=cap, crc_op=oracle.CRC_ADLER32)
            st, d, a = sim_inflate(sim, r, cap, limit=cap, crc_op=2)
            assert st == st0 and d == d0 and (st != 0 or a == a0), s["name"]
            seen[st] = seen.get(st, 0) + 1
    assert seen.get(0, 0) > 50 and seen.get(1, 0) > 50


def test_inflate_wide_turn_model(sim, oracle, monkeypatch):
    """The kernel's wide turn (64 speculative symbol decodes, chain by pointer
    doubling, parallel commit), modelled on the host: same bytes, same accept /
    reject, on the golden streams and on damaged ones."""
    monkeypatch.setenv("SIM_INFLATE_WIDE", "1")
    seen = {}
    for i, s in enumerate(util.zlib_streams()):
        st0, d0, a0 = oracle.inflate(s["raw"], crc_op=oracle.CRC_ADLER32)
        for budget in (24, 3, 400):
            st, d, a = sim_inflate(sim, s["raw"], s["plain_len"] + 100, crc_op=2, budget=budget)
            assert (st, d, a) == (st0, d0, a0), s["name"]
        lim = max(0, s["plain_len"] - 1)
        assert sim_inflate(sim, s["raw"], lim, limit=lim)[0] == oracle.inflate(s["raw"], decompressed_size=lim)[0]
        if len(s["raw"]) > 40000:
            continue
        cap = s["plain_len"] * 2 + 1000
        for r in util.corrupt_variants(s["raw"], i, 40):
            st0, d0, a0 = oracle.inflate(r, decompressed_size=cap, crc_op=oracle.CRC_ADLER32)
            st, d, a = sim_inflate(sim, r, cap, limit=cap, crc_op=2)
            assert st == st0 and d == d0 and (st != 0 or a == a0), s["name"]
            seen[st] = seen.get(st, 0) + 1
    assert seen.get(0, 0) > 30 and seen.get(1, 0) > 30
    for name, data in util.deflate_cases(small=True).items():
        for lvl in (0, 2):
            c = oracle.deflate(data, level=lvl)[1]
            assert sim_inflate(sim, c, len(data), limit=len(data))[:2] == (0, data), name


def test_deflate_lane_logic_bytes_equal_oracle(sim, oracle):
    for name, data in util.deflate_cases().items():
        for lvl in (0, 1, 2, 3):
            st0, c0, a0, blocks = oracle.deflate_trace(data, level=lvl, crc_op=oracle.CRC_ADLER32)
            st, c, a, kinds = sim_deflate(sim, oracle, data, lvl)
            assert st == 0 and c == c0 and a == a0, (name, lvl)
            assert kinds == [b.kind for b in blocks][:64], (name, lvl)


def test_parse_tiles_model_bytes_equal_oracle(sim, oracle, monkeypatch):
    """lz_parse_kernel's algorithm (pointer doubling + marking per 64-position
    tile instead of a serial walk), modelled on the host, gives the same bytes."""
    monkeypatch.setenv("SIM_PARSE_TILES", "1")
    for name, data in util.deflate_cases().items():
        for lvl in (1, 2, 3):
            st0, c0, a0 = oracle.deflate(data, level=lvl, crc_op=oracle.CRC_ADLER32)
            st, c, a, _ = sim_deflate(sim, oracle, data, lvl)
            assert st == 0 and c == c0 and a == a0, (name, lvl)


def _sim_cases_for_segments():
    """the deflate cases plus what the segment stitch is about: runs of one byte and short periods (paths that stay
    out of phase), long enough for several segments and blocks"""
    cases = dict(util.deflate_cases())
    rnd = random.Random(5)
    for k in (1, 2, 3, 7, 64, 257, 258, 259, 260, 515, 1000):
        pat = bytes(rnd.randrange(256) for _ in range(k))
        cases["period_%d" % k] = (pat * (150000 // k + 1))[:150000 + k]
    t = bytearray(util.text(200000, 9))
    t[50000:120000] = b"ab" * 35000
    t[140000:170000] = bytes(30000)
    cases["text_with_runs"] = bytes(t)
    return cases


@pytest.mark.parametrize("seg", [64, 4096, 8192])
def test_parse_segments_model_bytes_equal_oracle(sim, oracle, monkeypatch, seg):
    """lz_parse by segments (lz_parse_spec / _stitch / _gather_kernel): the algorithm on the host -- every segment
    parsed from its first position, stitched where the true path meets it within a tile, runs of equal steps, the
    block cut at the step that holds the block's byte 65534 -- gives the reference's bytes (64: a segment per tile,
    every boundary case many times over)."""
    monkeypatch.setenv("SIM_PARSE_SEGMENTS", str(seg))
    for name, data in _sim_cases_for_segments().items():
        for lvl in ((1, 2, 3) if seg != 64 else (2,)):
            st0, c0, a0 = oracle.deflate(data, level=lvl, crc_op=oracle.CRC_ADLER32)
            st, c, a, _ = sim_deflate(sim, oracle, data, lvl)
            assert st == 0 and c == c0 and a == a0, (name, lvl, seg)


def test_blocks_coded_independently_model_bytes_equal_oracle(sim, oracle, monkeypatch):
    """deflate_plan / _counts / _codelen / _scan_kernel's split of the block coder: what a block's bits owe to the
    blocks before it is the bit it starts at, the counts of the code-length symbols (never reset, Q1) and nothing
    else; the sizes the scan goes by are the real ones (a stored block's estimate runs 8 high when its type bits end
    a byte, Q3; a dynamic block's counts every block's code-length symbols)."""
    monkeypatch.setenv("SIM_EMIT_BLOCKS", "1")
    monkeypatch.setenv("SIM_PARSE_SEGMENTS", "4096")
    for name, data in _sim_cases_for_segments().items():
        for lvl in (1, 2, 3):
            st0, c0, a0 = oracle.deflate(data, level=lvl, crc_op=oracle.CRC_ADLER32)
            st, c, a, _ = sim_deflate(sim, oracle, data, lvl)
            assert st == 0 and c == c0 and a == a0, (name, lvl)


def test_deflate_lane_logic_fuzz(sim, oracle):
    rnd = random.Random(11)
    for t in range(600):
        n = rnd.randrange(0, 600)
        kind = rnd.randrange(4)
        if kind == 0:
            data = util.rand_bytes(n, t)
        elif kind == 1:
            data = util.rand_bytes(n, t, 2)
        elif kind == 2:
            data = (util.rand_bytes(rnd.randrange(1, 20), t) * (n // 2 + 1))[:n]
        else:
            data = util.text(n, t)
        lvl = rnd.randrange(1, 4)
        st0, c0, a0 = oracle.deflate(data, level=lvl, crc_op=oracle.CRC_ADLER32)
        st, c, a, _ = sim_deflate(sim, oracle, data, lvl)
        assert st == 0 and c == c0 and a == a0, (t, n, kind, lvl)


def test_chain_round_model_equals_serial_chain(sim):
    """lz_chain_kernel's round algorithm, with the surviving LDS store picked at
    random, yields the reference's insert_hash links."""
    cases = {
        "zeros": bytes(70000), "rand": util.rand_bytes(70000, 1), "nib": util.rand_bytes(70000, 2, 4),
        "nib3": util.rand_bytes(100000, 3, 3), "p2": b"ab" * 20000, "p9": b"abcdefghi" * 6000,
        "p300": util.rand_bytes(300, 4) * 200, "text": util.text(80000, 5), "bin": util.rand_bytes(50000, 6, 1),
        "short": b"abcabcabcab", "tiny": b"abcd", "n1027": util.rand_bytes(1027, 7),
        "old": util.rand_bytes(100, 8) + bytes(70000) + util.rand_bytes(100, 8) + bytes(70000),
    }
    for name, d in cases.items():
        n = len(d)
        for seed in (1, 2):
            a, b, mt = (C.c_uint16 * (n + 8))(), (C.c_uint16 * (n + 8))(), C.c_int()
            sim.sim_chain(d, n, a, seed, C.byref(mt))
            sim.sim_chain_serial(d, n, b)
            assert list(a[:max(0, n - 3)]) == list(b[:max(0, n - 3)]), name


def test_chain_links_by_segments_equal_serial_chain(sim):
    """lz_chain_segments_kernel's claim: a link is a function of the 32 KiB before its position, so a stream's links
    can be made a segment at a time, each from an empty table 32 768 positions before its first one."""
    cases = {
        "text": util.text(200000, 5), "nib": util.rand_bytes(150000, 2, 4), "nib3": util.rand_bytes(150000, 3, 3),
        "zeros": bytes(120000), "p300": util.rand_bytes(300, 4) * 500,
        "old": util.rand_bytes(100, 8) + bytes(70000) + util.rand_bytes(100, 8) + bytes(70000),
        "far": util.rand_bytes(40000, 9) + util.text(33000, 1) + util.rand_bytes(40000, 9) + util.text(70000, 2),
    }
    for name, d in cases.items():
        n = len(d)
        b = (C.c_uint16 * (n + 8))()
        sim.sim_chain_serial(d, n, b)
        for seg in (16384, 32768, 131072):
            a = (C.c_uint16 * (n + 8))()
            sim.sim_chain_segments(d, n, a, 3, seg)
            assert list(a[:n - 3]) == list(b[:n - 3]), (name, seg)


def test_symbol_value_closed_forms_equal_rfc_tables(sim):
    assert sim.sim_sym_values_match_tables() == 0


def test_crc_combination_rule(sim, oracle):
    rnd = random.Random(5)
    for _ in range(100):
        a, b = util.rand_bytes(rnd.randrange(0, 5000), rnd.random()), util.rand_bytes(rnd.randrange(0, 5000), rnd.random())
        raw_b = oracle.crc32_update(0, b)
        st_a = oracle.crc32_update(0xFFFFFFFF, a)
        assert sim.sim_crc_advance(st_a, raw_b, len(b)) == oracle.crc32_update(0xFFFFFFFF, a + b)


def test_inflate_models_header_fuzz(sim, oracle, monkeypatch):
    """random complete / perturbed Huffman codes in dynamic headers: plain and wide model vs oracle"""
    streams = util.header_fuzz_streams(78, 150, 450)
    for wide in (False, True):
        if wide:
            monkeypatch.setenv("SIM_INFLATE_WIDE", "1")
        for s in streams:
            st0, d0, c0 = oracle.inflate(s, decompressed_size=1 << 16, crc_op=2)
            st, d, a = sim_inflate(sim, s, 1 << 16, limit=1 << 16, crc_op=2, budget=24)
            assert st == st0
            if st0 == 0:
                assert d == d0 and a == c0


def test_inflate_span_model(sim, oracle, monkeypatch):
    """inflate_span.h (regions per lane, self-synchronising walks, LDS output tiles) -- the very
    code the kernel runs, on the emulated wave (host_sim/wave_emu.h), lanes resumed in ascending
    and in descending order: same bytes, checksums and statuses as the oracle on every input
    shape, under size limits that fall inside a span, and on damaged streams."""
    monkeypatch.setenv("SIM_INFLATE_WIDE", "1")
    stats = (C.c_uint64 * 8).in_dll(sim, "sim_span_stats")
    r = random.Random(2024)
    long_random = []
    while len(long_random) < 24:
        try:
            long_random.append(util.random_dynamic_stream(r, 0, max_symbols=9000))
        except (KeyError, IndexError, ValueError):
            pass
    for order in ("a", "d"):
        monkeypatch.setenv("SIM_INFLATE_SPAN", order)
        for i in range(8):
            stats[i] = 0
        for name, data in util.deflate_cases().items():
            for lvl in (0, 1, 2, 3) if order == "a" else (2,):
                st0, c, a0 = oracle.deflate(data, level=lvl, crc_op=oracle.CRC_ADLER32)
                st, d, a = sim_inflate(sim, c, len(data), limit=len(data), crc_op=2, budget=24)
                assert (st, d, a) == (0, data, a0), (name, lvl, order)
        assert stats[0] > 20 and stats[2] > 900000, list(stats)  # spans ran and produced most of the bytes
        # limits inside the data: the reference's "Expected decompression size exceeded", same prefix rules
        for name in ("nib64k", "text150k", "mixed", "zip-docs/rfc1951.txt"):
            data = util.deflate_cases()[name]
            c = oracle.deflate(data, level=2)[1]
            for lim in (len(data) - 1, len(data) // 2, 5000, len(data) - 300):
                st0, d0, _ = oracle.inflate(c, decompressed_size=lim)
                st, d, _ = sim_inflate(sim, c, lim, limit=lim)
                assert (st, d) == (st0, d0), (name, lim, order)
        for s in util.zlib_streams():
            st0, d0, a0 = oracle.inflate(s["raw"], crc_op=oracle.CRC_ADLER32)
            st, d, a = sim_inflate(sim, s["raw"], s["plain_len"] + 100, crc_op=2, budget=24)
            assert (st, d, a) == (st0, d0, a0), s["name"]
        # binary-like: stretches a span cannot take (zeros, short periods: granules too rich for a tile) between
        # stretches it can -- the span steps aside for the wide turns and comes back (SPAN_LATER), again and again
        rb = random.Random(77)
        for shape in range(3):
            parts = []
            while sum(map(len, parts)) < 150000:
                parts.append(bytes(rb.randrange(256 if shape else 16) for _ in range(rb.randrange(200, 6000))))
                parts.append(bytes([0, 0xFF, 0x90][shape]) * rb.randrange(300, 9000) if rb.random() < 0.7
                             else bytes(rb.randrange(256) for _ in range(rb.randrange(1, 9))) * rb.randrange(100, 1500))
                parts.append(util.deflate_cases()["text150k"][rb.randrange(0, 100000):][:rb.randrange(500, 8000)])
            data = b"".join(parts)
            spans_before = stats[0]
            st0, c, a0 = oracle.deflate(data, level=2, crc_op=oracle.CRC_ADLER32)
            st, d, a = sim_inflate(sim, c, len(data), limit=len(data), crc_op=2, budget=24)
            assert (st, d, a) == (0, data, a0), (shape, order)
            assert stats[0] - spans_before > 6, (shape, list(stats))  # spans kept coming back after the runs
        # more holes than a tile lists (SPAN_LIST_MAX): thousands of 3-byte matches in a row, hand-made
        for length, dist, n, seed in ((3, 3, 6000, None), (3, 1, 3000, None), (3, 4, 8000, 1), (4, 4, 8000, 2), (4, 2, 5000, 3)):
            c, data = util.fixed_block_of_short_matches(n, length, dist, seed=seed)
            st0, d0, a0 = oracle.inflate(c, decompressed_size=len(data), crc_op=2)
            st, d, a = sim_inflate(sim, c, len(data), limit=len(data), crc_op=2, budget=24)
            assert (st0, d0) == (0, data) and (st, d, a) == (0, data, a0), (length, dist, order)
        # random literals and matches at every mix of lengths and distances, hand-made
        for seed in range(8 if order == "a" else 3):
            c, data = util.random_fixed_block(seed, 5000, max_dist=[1, 4, 64, 300, 5000, 3, 16, 32768][seed], max_len=[3, 4, 10, 40, 258][seed % 5],
                                              lit_share=[0.0, 0.05, 0.5][seed % 3])
            st0, d0, a0 = oracle.inflate(c, decompressed_size=len(data), crc_op=2)
            st, d, a = sim_inflate(sim, c, len(data), limit=len(data), crc_op=2, budget=24)
            assert (st0, d0) == (0, data) and (st, d, a) == (0, data, a0), (seed, order)
        # random complete codes (long codes, odd alphabets), thousands of symbols; then damaged
        seen = {}
        for k, s in enumerate(long_random):
            st0, d0, a0 = oracle.inflate(s, decompressed_size=1 << 21, crc_op=2)
            st, d, a = sim_inflate(sim, s, 1 << 21, limit=1 << 21, crc_op=2, budget=24)
            assert st0 == 0 and (st, d, a) == (st0, d0, a0), k
            for v in util.corrupt_variants(s, k, 6 if order == "a" else 2):
                st0, d0, a0 = oracle.inflate(v, decompressed_size=1 << 21, crc_op=2)
                st, d, a = sim_inflate(sim, v, 1 << 21, limit=1 << 21, crc_op=2, budget=24)
                assert st == st0 and (st != 0 or (d == d0 and a == a0)), k
                seen[st] = seen.get(st, 0) + 1
        for i, s in enumerate(util.zlib_streams()):
            if len(s["raw"]) < 2000:
                continue
            cap = s["plain_len"] * 2 + 1000
            for v in util.corrupt_variants(s["raw"], i, 12 if order == "a" else 4):
                if len(v) > 300 and r.random() < 0.5:  # damage past the header too
                    b = bytearray(v)
                    b[r.randrange(200, len(b))] ^= 1 << r.randrange(8)
                    v = bytes(b)
                st0, d0, a0 = oracle.inflate(v, decompressed_size=cap, crc_op=oracle.CRC_ADLER32)
                st, d, a = sim_inflate(sim, v, cap, limit=cap, crc_op=2)
                assert st == st0 and (st != 0 or (d == d0 and a == a0)), s["name"]
                seen[st] = seen.get(st, 0) + 1
        assert seen.get(0, 0) > 5 and seen.get(1, 0) > 20, seen


def test_huffman_two_queues_equal_the_heap(sim):
    """Huffman.lengths_of_freqs (zd.ml:404-473) two ways: the reference's heap (huff_lengths_of_freqs, what the
    oracle restates) and the two queues deflate_emit runs (huff_lengths_of_freqs_tq).  The tree is a function
    of the order of the keys (freq << 10) | link, so frequency ties -- merged nodes before leaves, later merged
    nodes first -- and the flatten-and-retry path are what must agree."""
    import numpy as np

    r = random.Random(20261003)

    def fib(n):
        a, b, out = 1, 1, []
        for _ in range(n):
            out.append(a)
            a, b = b, a + b
        return out

    for it in range(20000):
        kind = it % 9
        max_sym, max_len = ((285, 15), (29, 15), (18, 7))[it % 3]
        n = max_sym + 1
        if kind == 0: fr = [r.choice([0, 1]) for _ in range(n)]
        elif kind == 1: fr = [r.choice([0, 1, 2]) for _ in range(n)]
        elif kind == 2: fr = [r.choice([0, 1, 2, 4, 8, 16]) for _ in range(n)]
        elif kind == 3: fr = [r.randrange(0, 5) for _ in range(n)]
        elif kind == 4: fr = [r.randrange(0, 70000) if r.random() < 0.5 else 0 for _ in range(n)]
        elif kind == 5:
            fr = [0] * n
            fb = fib(min(n, 30))
            for i, v in zip(r.sample(range(n), len(fb)), fb):
                fr[i] = min(v, 65535)
        elif kind == 6:
            fr = [0] * n
            for i in r.sample(range(n), r.randrange(0, min(n, 6) + 1)):
                fr[i] = r.randrange(1, 4)
        elif kind == 7: fr = [int(2 ** r.uniform(0, 16)) if r.random() < 0.7 else 0 for _ in range(n)]
        else:
            base = r.randrange(1, 1000)
            fr = [base * r.choice([0, 1, 1, 2, 3]) for _ in range(n)]
        f = np.array(fr, np.uint32)
        a = np.zeros(n, np.uint32)
        b = np.zeros(n, np.uint32)
        assert sim.sim_huff_lengths(f.ctypes.data, max_sym, max_len, a.ctypes.data, b.ctypes.data) == 0, (it, fr)


def _find_sources():
    import os
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = (open(os.path.join(root, "SURVEY.md"), "rb").read() + open(os.path.join(root, "BASELINE.md"), "rb").read() * 40)[:150000]
    assert len(text) == 150000
    rnd = random.Random(11)
    symbols = bytes(rnd.randrange(16) * 17 for _ in range(150000))
    records = b"".join(bytes([rnd.randrange(256), rnd.randrange(64)]) * 2 for _ in range(40000))
    for name, data in (("text", text), ("symbols", symbols), ("records", records)):
        yield name, data, zlib


def test_block_header_search_finds_every_dynamic_block(sim, oracle):
    """inflate_find.h (the search of inflate.hip's one-stream path) over every bit offset of streams of the
    reference's encoder (the oracle) and of zlib: every block with a dynamic header is found where the inflate model
    says it starts, and next to nothing else passes."""
    sim.sim_find_candidates.restype = C.c_uint64
    sim.sim_inflate_block_starts.restype = C.c_uint64
    for name, data, zlib in _find_sources():
        streams = [("oracle-%d" % lv, oracle.deflate(data, level=lv)[1]) for lv in (1, 2)]
        for lv in (1, 6):
            c = zlib.compressobj(lv, zlib.DEFLATED, -15)
            streams.append(("zlib-%d" % lv, c.compress(data) + c.flush()))
        for enc, raw in streams:
            st, d, _ = sim_inflate(sim, raw, len(data), limit=len(data))
            assert st == 0 and d == data, (name, enc)
            bits, types = (C.c_uint64 * 4096)(), (C.c_int * 4096)()
            nb = sim.sim_inflate_block_starts(bits, types, 4096)
            starts = {int(bits[i]) for i in range(nb) if types[i] == 2}
            assert nb >= 2 and len(starts) >= 1, (name, enc, nb)
            cand, nf = (C.c_uint64 * 65536)(), C.c_uint64()
            nc = sim.sim_find_candidates(raw, len(raw), cand, 65536, C.byref(nf))
            found = {int(cand[i]) for i in range(nc)}
            assert 0 in found
            assert starts <= found, (name, enc, sorted(starts - found))
            assert len(found - starts - {0}) <= 1, (name, enc, sorted(found - starts))
            assert nf.value < len(raw) // 32  # what the first test lets through to the second


def test_inflate_token_form_resolves_to_the_plain_decode(sim, oracle):
    """inflate.hip's IM_TOKEN form of the span decoder on the emulated wave (literals stored, a match's bytes written
    down as the positions they copy -- directly, or "following": what those positions copy -- tiles that are refused
    leave nothing behind), spans cut short like those of a wave that leaves at a checkpoint; the copies resolved
    afterwards must give the stream's bytes."""
    import zlib
    rnd = random.Random(5)
    datas = []
    for name, data, _ in _find_sources():
        datas.append((name, data[:90000]))
    datas.append(("runs", b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 700) + bytes(rnd.randrange(256) for _ in range(rnd.randrange(40)))
                                   for _ in range(300))))
    for name, data in datas:
        streams = [("oracle-2", oracle.deflate(data, level=2)[1])]
        for lv, stg in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED)):
            c = zlib.compressobj(lv, zlib.DEFLATED, -15, 8, stg)
            streams.append(("zlib-%d-%d" % (lv, stg), c.compress(data) + c.flush()))
        for enc, raw in streams:
            for follow, desc, cut in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 20000), (1, 0, 9000)):
                dst = C.create_string_buffer(len(data) + 64)
                ol = C.c_uint64()
                st = sim.sim_inflate_token(raw, len(raw), dst, len(data), follow, desc, cut, C.byref(ol))
                assert st == 0 and ol.value == len(data) and dst.raw[:len(data)] == data, (name, enc, follow, desc, cut)
