"""Pins the CPU oracle (oracle/zd_oracle.c) to everything the reference's own
tests hold for this path and to Python zlib as an independent
decoder / checksum.  No GPU."""
import os
import sys
import zlib

import pytest

import util

LEVELS = {"none": 0, "fast": 1, "default": 2, "best": 3}
KIND = {0: "stored", 1: "fixed", 2: "dynamic"}


def test_crc32_kat(oracle):  # test/test.ml:16-20
    assert oracle.crc32(b"") == 0
    assert oracle.crc32(util.FOX) == 0x414FA339


def test_adler32_kat(oracle):  # test/test.ml:22-26
    assert oracle.adler32(b"") == 1
    assert oracle.adler32(util.FOX) == 0x5BDC0FDA


def test_crc32_matches_zlib(oracle):
    for n in (0, 1, 3, 4, 5, 255, 4096, 70001):
        d = util.rand_bytes(n, n)
        assert oracle.crc32(d) == zlib.crc32(d)
    a, b = util.rand_bytes(1000, 1), util.rand_bytes(777, 2)
    st = oracle.crc32_update(oracle.crc32_update(0xFFFFFFFF, a), b) ^ 0xFFFFFFFF
    assert st == zlib.crc32(a + b)  # chaining across calls is exact


def test_adler32_matches_zlib_when_q6_does_not_fire(oracle):
    for d in (b"", b"a", util.text(100000, 1), bytes(70000), util.rand_bytes(50000, 3, 4)):
        assert oracle.adler32(d) == zlib.adler32(d)


def test_adler32_signed_remainder_q6(oracle):
    # 0xFF x 4200 differs from RFC 1950
    d = b"\xff" * 4200
    assert oracle.adler32(d) == 0xA2045889
    assert zlib.adler32(d) == 0xA2D65889


def test_adler32_chunking_q7(oracle):
    # the value depends on how the bytes are split over update calls
    d = util.rand_bytes(200000, 42)
    whole = oracle.adler32(d)
    st = 1
    for i in range(0, len(d), 65534):
        st = oracle.adler32_update(st, d[i:i + 65534])
    assert whole != zlib.adler32(d)
    assert st != whole


@pytest.mark.parametrize("level", ["default", "fast", "best", "none"])
def test_deflate_trip_and_block_kinds(oracle, level):  # test/test.ml:28-43,123-126
    for s, kind in util.trip_strings():
        st, c, _, blocks = oracle.deflate_trace(s, level=LEVELS[level])
        assert st == 0
        st, d, _ = oracle.inflate(c)
        assert st == 0 and d == s
        assert zlib.decompress(c, -15) == s  # independent decoder
        if level != "none":
            assert [KIND[b.kind] for b in blocks] == [kind]
        else:
            assert all(b.kind == 0 for b in blocks)


def test_decompression_size_limits(oracle):  # test/test.ml:45-55
    src = util.kat()["limits"].encode()
    st, c, _ = oracle.deflate(src)  # default level = best
    assert st == 0
    assert oracle.inflate(c)[1] == src
    assert oracle.inflate(c, decompressed_size=len(src))[:2] == (0, src)
    assert oracle.inflate(c, decompressed_size=len(src) + 1)[:2] == (0, src)
    assert oracle.inflate(c, decompressed_size=len(src) - 1)[0] == oracle.ERR_SIZE_EXCEEDED


def test_zip_docs_fixture(oracle):  # test/test.ml:57-118
    z = util.zip_docs()
    assert len(z) == 56924
    for m, raw in util.zip_docs_members():
        st, d, crc = oracle.inflate(raw, decompressed_size=m["decompressed_size"],
                                    crc_op=oracle.CRC_CRC32)
        assert st == 0 and len(d) == m["decompressed_size"] and crc == m["crc32"]
        assert d == zlib.decompress(raw, -15)
        # re-deflate with the default level and re-check (redeflate_recode, test/test.ml:58-73)
        st, c, crc2 = oracle.deflate(d, crc_op=oracle.CRC_CRC32)
        assert st == 0 and crc2 == m["crc32"]
        st, d2, crc3 = oracle.inflate(c, decompressed_size=len(d), crc_op=oracle.CRC_CRC32)
        assert st == 0 and d2 == d and crc3 == m["crc32"]
        assert zlib.decompress(c, -15) == d


def test_zip_docs_fixture_was_not_made_by_the_reference_encoder(oracle):
    """Why the fixture pins inflate and the checksums but NOT the deflate bytes (DESIGN.md row (c),
    oracle/zd_oracle.h): its members were compressed by Info-ZIP / zlib.  zlib level 6 reproduces the
    stored bytes of rfc1951.txt exactly; no level of the restated reference encoder does.  If this
    ever fails the other way round, the fixture has become a pin for deflate and the docs must say so."""
    members = dict((m["path"], (m, raw)) for m, raw in util.zip_docs_members())
    m, raw = members["zip-docs/rfc1951.txt"]
    plain = zlib.decompress(raw, -15)
    assert len(raw) == 11132 and len(plain) == 36944
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    assert co.compress(plain) + co.flush() == raw
    for _, (mm, rr) in members.items():
        d = zlib.decompress(rr, -15)
        for level in (oracle.LEVEL_NONE, oracle.LEVEL_FAST, oracle.LEVEL_DEFAULT, oracle.LEVEL_BEST):
            assert oracle.deflate(d, level=level)[1] != rr, (mm["path"], level)


def test_inflate_zlib_made_streams(oracle):
    n = 0
    for s in util.zlib_streams():
        st, d, crc = oracle.inflate(s["raw"], crc_op=oracle.CRC_CRC32)
        assert st == 0 and len(d) == s["plain_len"] and crc == s["plain_crc32"], s["name"]
        assert d == zlib.decompress(s["raw"], -15)
        n += 1
    assert n >= 30


def test_inflate_rejects(oracle):
    # BTYPE 3, LEN/NLEN mismatch, truncated input, distance before start
    assert oracle.inflate(b"\x07")[0] == oracle.ERR_CORRUPTED
    assert oracle.inflate(b"\x01\x01\x00\xff\xff\x00")[0] == oracle.ERR_CORRUPTED
    assert oracle.inflate(b"\x01\x02\x00\xfd\xffa")[0] == oracle.ERR_CORRUPTED
    assert oracle.inflate(b"")[0] == oracle.ERR_CORRUPTED
    # fixed block: literal 'a' then a match with distance 2 > 1 byte produced
    import struct
    bad = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    ok = bad.compress(b"aaaaaaaaaaaaaaaa") + bad.flush()
    assert oracle.inflate(ok)[:2] == (0, b"a" * 16)
    # trailing garbage after the final block is ignored (zipc_deflate.ml:704)
    assert oracle.inflate(ok + b"garbage")[:2] == (0, b"a" * 16)


def test_deflate_matches_independent_decoder_on_all_cases(oracle):
    for name, data in util.deflate_cases().items():
        for lvl in (0, 1, 2, 3):
            st, c, crc = oracle.deflate(data, level=lvl, crc_op=oracle.CRC_CRC32)
            assert st == 0 and crc == zlib.crc32(data), name
            assert zlib.decompress(c, -15) == data, (name, lvl)
            assert len(c) <= oracle.deflate_bound(len(data))


def test_survey_anchors(oracle):
    # SURVEY.md Appendix B (from an independent scratch model of the reference)
    assert oracle.deflate(b"", level=2)[1] == bytes.fromhex("0300")
    assert oracle.deflate(b"a", level=2)[1] == bytes.fromhex("4b0400")
    assert oracle.deflate(b"hellohello", level=2)[1] == bytes.fromhex("cb48cdc9c9071300")
    assert oracle.deflate(b"", level=0)[1] == bytes.fromhex("010000ffff")
    st, c, _, blocks = oracle.deflate_trace(bytes(1 << 20), level=1)
    assert len(c) == 1224 and len(blocks) == 17
    assert c[:16] == bytes.fromhex("ecc0810000000080a0fda917a9000000")
    assert [b.kind for b in blocks] == [2] * 16 + [1]
    st, c, _, blocks = oracle.deflate_trace(util.rand_bytes(65536, 5), level=2)
    assert len(c) == 65543 and [b.kind for b in blocks] == [0, 1]


def test_zlib_container(oracle):
    for lvl, hdr in ((0, "7801"), (1, "785e"), (2, "789c"), (3, "78da")):
        d = util.text(5000, lvl)
        st, c, adler = oracle.zlib_compress(d, level=lvl)
        assert st == 0 and c[:2].hex() == hdr and adler == zlib.adler32(d)
        assert zlib.decompress(c) == d
        st, d2, a2, _, _ = oracle.zlib_decompress(c)
        assert st == 0 and d2 == d and a2 == adler
    c = bytearray(oracle.zlib_compress(b"hello world", level=2)[1])
    c[-1] ^= 1
    st, _, _, expect, found = oracle.zlib_decompress(bytes(c))
    assert st == oracle.ERR_CHECKSUM and expect != found
    assert oracle.zlib_decompress(b"\x78\x9c\x03\x00")[0] == oracle.ERR_CORRUPTED  # < 6 bytes
    assert oracle.zlib_decompress(b"\x79\x9c\x03\x00\x00\x00\x00\x01")[0] == oracle.ERR_CORRUPTED
    assert oracle.zlib_decompress(b"\x78\xbb\x03\x00\x00\x00\x00\x01")[0] == oracle.ERR_ZLIB_DICT


def test_huffman_length_limit_retry(oracle):
    # Fibonacci-like frequencies force codes longer than the limit: the
    # reference halves freq_cap and retries (zipc_deflate.ml:470-473)
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    lens = oracle.huffman_lengths(fib[:30], 15)
    assert max(lens) <= 15 and min(lens) >= 1
    assert sum(2.0 ** -l for l in lens) <= 1.0 + 1e-12
    lens7 = oracle.huffman_lengths(fib[:19], 7)
    assert max(lens7) <= 7
    assert oracle.huffman_lengths([0, 5, 0], 15) == [0, 1, 0]  # single symbol -> length 1
    assert oracle.huffman_lengths([0, 0, 0], 15) == [0, 0, 0]


# ---- double entry for the encode side (round 6) --------------------------------------------------------------------
# tests/golden/deflate_vectors.json is what a SECOND reading of src/zipc_deflate.ml:166-206,404-528,742-1277 -- naive Python
# written from the .ml without looking at oracle/zd_oracle.c (tests/golden/zd_second_reading.py, run by
# make_deflate_vectors.py in the build container) -- makes of 50 named inputs at `Fast / `Default / `Best: the reference
# itself holds no vector for compressed bytes (test/test.ml:33-36 only round-trips).  Two readings that must agree.

def _vectors():
    import json

    return json.load(open(os.path.join(util.GOLDEN, "deflate_vectors.json")))


def test_oracle_equals_the_second_reading_of_the_encoder(oracle):
    import hashlib

    doc = _vectors()
    kinds = {0: "none", 1: "fixed", 2: "dynamic"}
    checked = 0
    for name, v in doc["vectors"].items():
        data = util.vector_input(name)
        assert len(data) == v["len"] and hashlib.sha256(data).hexdigest() == v["sha256_plain"], name
        for level, want in v["levels"].items():
            lv = oracle.LEVELS[level]
            oracle.huffman_retries(reset=True)
            st, c, crc, blocks = oracle.deflate_trace(data, level=lv, crc_op=oracle.CRC_CRC32)
            retries = sum(oracle.huffman_retries())
            assert st == 0 and len(c) == want["clen"] and hashlib.sha256(c).hexdigest() == want["sha256"], (name, level)
            assert crc == want["crc32"] == zlib.crc32(data), (name, level)
            # the kind of every block (the chooser, zd.ml:1094-1104) and how often lengths_of_freqs retried (zd.ml:470-473)
            got_kinds = " ".join(k if n == 1 else "%s*%d" % (k, n) for k, n in _runs([kinds[b.kind] for b in blocks]))
            assert got_kinds == want["blocks"], (name, level, got_kinds)
            assert retries == want["huffman_retries"], (name, level, retries)
            # the fused Adler-32: one update per block (Q7), signed remainder (Q6)
            st, c2, adler = oracle.deflate(data, level=lv, crc_op=oracle.CRC_ADLER32)
            assert c2 == c and adler == want["adler32_fused"], (name, level)
            checked += 1
    assert checked >= 130
    # Adler_32.string over whole buffers: the values no standard library gives (0xFF x 4200 -> a2045889)
    for name, want in doc["adler32_whole"].items():
        assert oracle.adler32(util.vector_input(name)) == want, name
    assert doc["adler32_whole"]["ff4200"] == 0xA2045889 and zlib.adler32(b"\xff" * 4200) != 0xA2045889


def _runs(xs):
    out = []
    for x in xs:
        if out and out[-1][0] == x:
            out[-1][1] += 1
        else:
            out.append([x, 1])
    return out


def test_the_second_reading_is_live_and_reproduces_its_vectors():
    """the generator's own code on the quick inputs: the committed file is what this code makes (not a stale artefact)"""
    import hashlib

    sys.path.insert(0, util.GOLDEN)
    import zd_second_reading as Z2

    doc = _vectors()
    for name in ("trip0", "trip1", "trip2", "trip3", "trip4", "fib_codelen", "ff4200", "zipdocs_rfc1951", "c2_stream0"):
        data = util.vector_input(name)
        for level, want in doc["vectors"][name]["levels"].items():
            crc, comp, stats = Z2.crc_and_deflate(data, level, Z2.CRC_32)
            assert (len(comp), hashlib.sha256(comp).hexdigest(), crc) == (want["clen"], want["sha256"], want["crc32"]), (name, level)
    # the block kinds the reference's own test comments expect (test/test.ml:38-42): Fixed, Fixed, Fixed, Dynamic, None
    assert [doc["vectors"]["trip%d" % i]["levels"]["default"]["blocks"] for i in range(5)] == ["fixed", "fixed", "fixed", "dynamic", "none"]
    assert Z2.adler_32_string(b"\xff" * 4200) == doc["adler32_whole"]["ff4200"]
    assert Z2.zlib_compress(b"hello world", "default")[1][:2].hex() == "789c"
