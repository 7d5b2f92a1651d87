"""Parity of the HIP path against the CPU oracle, through the C ABI
(include/zipc_hip.h), on a real MI355X.  Bar: bit-exact bytes, identical
accept/reject status, identical checksums (integer/byte work: no tolerance)."""
import random
import zlib

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

LEVELS = {"none": 0, "fast": 1, "default": 2, "best": 3}


# ---------------------------------------------------------------- host forms


def test_checksum_kats_like_reference(gpu_ctx):  # test/test.ml:16-26
    from zipc_amd.zipc_deflate import Adler_32, Crc_32

    assert Crc_32.string(b"") == 0
    assert Crc_32.string(util.FOX) == 0x414FA339
    assert Adler_32.string(b"") == 1
    assert Adler_32.string(util.FOX) == 0x5BDC0FDA
    assert Crc_32.string(b"xx" + util.FOX + b"yy", start=2, len=len(util.FOX)) == 0x414FA339
    assert Crc_32.check(1, 1).is_ok() and Crc_32.check(1, 2).is_error()
    assert Crc_32.check(0xABC, 0x1).error == "Checksum mismatch, expected abc found 1)"


def test_checksums_match_oracle(gpu_ctx, oracle):
    from zipc_amd.zipc_deflate import Adler_32, Crc_32

    sizes = [0, 1, 3, 4, 5, 255, 256, 257, 4200, 5551, 5552, 5553, 11104, 65535, 65536, 65537,
             200000, (1 << 20) + 3, 3 * 5552 * 100]
    for n in sizes:
        for kind in ("rand", "ff", "text"):
            d = util.rand_bytes(n, n) if kind == "rand" else (b"\xff" * n if kind == "ff" else util.text(n, n))
            assert Crc_32.string(d) == oracle.crc32(d), (n, kind)
            assert Adler_32.string(d) == oracle.adler32(d), (n, kind)  # incl. Q6 signed remainder
    assert Adler_32.string(b"\xff" * 4200) == 0xA2045889  # not RFC 1950's a2d65889


def test_both_checksums_in_one_pass_match_oracle(gpu_ctx, oracle):
    """checksum_device with both checksums wanted reads the bytes once (crc32_adler_segments_kernel): every
    length around the grids involved (16-byte units, 128-byte pieces, 5552-byte chunks, 32 KiB segments), at
    odd starting addresses, on bytes that make the largest sums."""
    import numpy as np
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    big = 3 * 32768 + 5552 * 7 + 300
    pools = {"rand": rng.integers(0, 256, big + 64, dtype=np.uint8), "ff": np.full(big + 64, 255, np.uint8),
             "text": np.frombuffer(util.text(big + 64, 5), np.uint8).copy()}
    d_pools = {k: torch.from_numpy(v).to(dev) for k, v in pools.items()}
    lens = set(range(0, 300)) | {big}
    for base in (5552, 2 * 5552, 32768, 32768 + 5552, 65536, 3 * 32768, 6 * 5552):
        lens |= set(range(base - 20, base + 21))
    lens |= {int(x) for x in rng.integers(300, big, 60)}
    for n in sorted(lens):
        for kind in (("ff", "rand") if n % 3 else ("ff", "rand", "text")):
            off = int(rng.integers(0, 64))
            crc, adler = batch.checksum_device(gpu_ctx, d_pools[kind][off:off + n])
            d = pools[kind][off:off + n].tobytes()
            assert crc == oracle.crc32(d), (n, kind, off)
            assert adler == oracle.adler32(d), (n, kind, off)


def _chunk_with_s2(target_s2, n=5552):
    """n bytes whose position-weighted sum S2 = sum (n - i) b_i equals target_s2"""
    full = n * (n + 1) // 2
    v = min(target_s2 // full, 254)
    b = [v] * n
    rem = target_s2 - v * full
    i = 0
    while rem >= n - i and i < n:
        w = n - i
        k = min(rem // w, 255 - b[i])
        b[i] += k
        rem -= k * w
        i += 1
    while rem > 0:  # weights n-i for i >= current index are all <= rem's reach
        w = min(rem, n - i)
        j = n - w
        if b[j] < 255:
            b[j] += 1
            rem -= w
        else:
            i += 1
    return bytes(b)


def test_adler_chain_ambiguous_chunks(gpu_ctx, oracle):
    """Buffers built so that n*s1 + S2 of a 5552-byte chunk lands right at 0 and
    right at 2^31, where the reference's signed remainder (zipc_deflate.ml:95,196)
    depends on the sign of the running s2: the parallel chain must replay those
    chunks exactly."""
    import torch

    from zipc_amd import batch
    from zipc_amd.zipc_deflate import Adler_32

    n = 5552
    for deltas in ([-30000, -1, 0, 1, 30000], [0, 0, 0], [65000, -65000, 5, -5, 0, 1]):
        for lead in (b"", b"\xff" * n, util.rand_bytes(3 * n, 9)):
            data = bytearray(lead)
            for d in deltas:
                for target in (1 << 31, 40000):
                    st = oracle.adler32_update(1, bytes(data)) if len(data) % n == 0 else None
                    s1 = oracle.adler32(bytes(data)) & 0xFFFF if len(data) else 1
                    want = target + d - n * s1
                    if want < 0:
                        want = 0
                    data += _chunk_with_s2(want, n)
                    data += b"\xff" * n  # pushes s2 negative (C >= 2^31) before the next crafted chunk
            data = bytes(data)
            assert len(data) % n == 0
            assert Adler_32.string(data) == oracle.adler32(data), (deltas, len(lead))
            assert Adler_32.string(data + b"xyz") == oracle.adler32(data + b"xyz")
            both = batch.checksum_device(gpu_ctx, torch.frombuffer(bytearray(data + b"xyz"), dtype=torch.uint8).cuda())
            assert both == (oracle.crc32(data + b"xyz"), oracle.adler32(data + b"xyz"))  # the one-pass form


def test_deflate_trip_like_reference(gpu_ctx, oracle):  # test/test.ml:28-43,123-126
    from zipc_amd import zipc_deflate as Z

    for level in ("default", "fast", "best", "none"):
        for s, _kind in util.trip_strings():
            cs = Z.deflate(s, level=level).get_ok()
            assert Z.inflate(cs).get_ok() == s
            assert cs == oracle.deflate(s, level=LEVELS[level])[1]  # bit-exact vs the reference algorithm


def test_decompression_size_limits_like_reference(gpu_ctx):  # test/test.ml:45-55
    from zipc_amd import zipc_deflate as Z

    src = util.kat()["limits"].encode()
    limit = len(src)
    csrc = Z.deflate(src).get_ok()
    assert Z.inflate(csrc).get_ok() == src
    assert Z.inflate(csrc, decompressed_size=limit).get_ok() == src
    assert Z.inflate(csrc, decompressed_size=limit + 1).get_ok() == src
    r = Z.inflate(csrc, decompressed_size=limit - 1)
    assert r.is_error() and r.error == "Expected decompression size exceeded"


def test_zip_docs_fixture_like_reference(gpu_ctx, oracle):  # test/test.ml:57-118
    from zipc_amd import zipc_deflate as Z

    z = util.zip_docs()
    for m, raw in util.zip_docs_members():
        # decoded straight out of the archive string with ~start ~len, like Zipc.File does
        data, crc = Z.inflate_and_crc_32(z, decompressed_size=m["decompressed_size"],
                                         start=m["data_start"], len=m["compressed_size"]).get_ok()
        assert len(data) == m["decompressed_size"] and crc == m["crc32"]
        assert data == zlib.decompress(raw, -15)
        crc2, cs = Z.crc_32_and_deflate(data).get_ok()  # default level (= best), re-encode
        assert crc2 == m["crc32"]
        assert cs == oracle.deflate(data)[1]
        assert Z.inflate(cs, decompressed_size=len(data)).get_ok() == data


def test_deflate_bytes_equal_oracle_all_cases(gpu_ctx, oracle):
    from zipc_amd import zipc_deflate as Z

    for name, data in util.deflate_cases().items():
        for level, lv in LEVELS.items():
            st, c0, a0 = oracle.deflate(data, level=lv, crc_op=oracle.CRC_ADLER32)
            adler, cs = Z.adler_32_and_deflate(data, level=level).get_ok()
            assert cs == c0, (name, level)
            assert adler == a0, (name, level)  # per-block chaining (Q7)
            crc, cs2 = Z.crc_32_and_deflate(data, level=level).get_ok()
            assert cs2 == c0 and crc == zlib.crc32(data), (name, level)


def test_inflate_golden_streams_and_limits(gpu_ctx, oracle):
    from zipc_amd import zipc_deflate as Z

    for s in util.zlib_streams():
        data, crc = Z.inflate_and_crc_32(s["raw"]).get_ok()
        assert len(data) == s["plain_len"] and crc == s["plain_crc32"], s["name"]
        data2, adler = Z.inflate_and_adler_32(s["raw"]).get_ok()
        assert data2 == data and adler == oracle.inflate(s["raw"], crc_op=oracle.CRC_ADLER32)[2]
        for lim in (s["plain_len"], max(0, s["plain_len"] - 1)):
            st0, d0, _ = oracle.inflate(s["raw"], decompressed_size=lim)
            r = Z.inflate(s["raw"], decompressed_size=lim)
            assert r.is_ok() == (st0 == 0), (s["name"], lim)
            if st0 == 0:
                assert r.get_ok() == d0
            else:
                assert r.error == oracle.MESSAGES[st0]


def test_zlib_container(gpu_ctx, oracle):
    from zipc_amd import zipc_deflate as Z

    for level, lv in LEVELS.items():
        d = util.rand_bytes(200000, lv)  # Q6/Q7 fire on random data
        adler, cs = Z.zlib_compress(d, level=level).get_ok()
        st, c0, a0 = oracle.zlib_compress(d, level=lv)
        assert cs == c0 and adler == a0
        out, a2 = Z.zlib_decompress(cs).get_ok()
        assert out == d and a2 == adler
    cs = bytearray(Z.zlib_compress(b"hello world", level="default").get_ok()[1])
    cs[-1] ^= 1
    r = Z.zlib_decompress(bytes(cs))
    assert r.is_error() and r.error[0] is not None and r.error[1].startswith("Checksum mismatch")
    assert Z.zlib_decompress(b"\x78\x9c\x03\x00").error == (None, "Corrupted data stream")
    assert Z.zlib_decompress(b"\x78\xbb\x03\x00\x00\x00\x00\x01").error == (None, "Preset dictionary unsupported")
    assert Z.zlib_decompress(b"\x77\x85\x03\x00\x00\x00\x00\x01").error[1].startswith("Unknown compression method")


# ---------------------------------------------------------------- batch forms


def test_inflate_batch_ragged_with_errors(gpu_ctx, oracle):
    streams, expect = [], []
    for i, s in enumerate(util.zlib_streams()):
        streams.append(s["raw"])
        if len(s["raw"]) < 40000:
            streams.extend(util.corrupt_variants(s["raw"], i, 6))
    streams += [b"", b"\x07", b"\x01\x00\x00\xff\xff", b"\x03\x00"]
    caps = []
    for s in streams:
        st, d, _ = oracle.inflate(s, decompressed_size=1 << 20)
        caps.append(max(len(d) + 8, 1024))
    for crc_op in (0, 1, 2):
        # limit = cap: same accept/reject as the oracle with ?decompressed_size
        import torch

        from zipc_amd import batch

        dev = torch.device("cuda", 0)
        src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
        slots = [(c + 255) // 256 * 256 + 256 for c in caps]
        dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
        descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps, limit=caps)
        src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
        dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
        d_descs = batch.to_device(descs, dev)
        d_res = torch.zeros(len(streams) * 16, dtype=torch.uint8, device=dev)
        batch.inflate_batch(gpu_ctx, src, dst, d_descs, d_res, len(streams), max(caps), crc_op)
        res = batch.results_from_device(d_res)
        out = dst.cpu().numpy()
        n_ok = 0
        for i, s in enumerate(streams):
            st0, d0, c0 = oracle.inflate(s, decompressed_size=caps[i], crc_op=crc_op)
            assert res["status"][i] == st0, (i, crc_op)
            if st0 == 0:
                n_ok += 1
                assert res["out_len"][i] == len(d0)
                o = int(dst_off[i])
                assert out[o:o + len(d0)].tobytes() == d0, i
                assert res["checksum"][i] == c0, (i, crc_op)
        assert n_ok > 40 and n_ok < len(streams)


def test_inflate_header_fuzz_equals_oracle(gpu_ctx, oracle):
    """accept/reject and output of damaged and random dynamic headers: the kernel
    builds its decode tables wave-parallel, the oracle the reference's way"""
    import torch

    from zipc_amd import batch

    streams = util.header_fuzz_streams(77, 500, 1500)
    cap = 1 << 16
    dev = torch.device("cuda", 0)
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    dst_off = (np.arange(len(streams), dtype=np.uint64) * (cap + 256))
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, [cap] * len(streams), limit=[cap] * len(streams))
    src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(len(streams) * (cap + 256) + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(len(streams) * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(gpu_ctx, src, dst, batch.to_device(descs, dev), d_res, len(streams), cap, 2)
    res = batch.results_from_device(d_res)
    out = dst.cpu().numpy()
    kinds = {}
    for i, s in enumerate(streams):
        st0, d0, c0 = oracle.inflate(s, decompressed_size=cap, crc_op=2)
        kinds[st0] = kinds.get(st0, 0) + 1
        assert res["status"][i] == st0, (i, st0, int(res["status"][i]))
        if st0 == 0:
            o = int(dst_off[i])
            assert res["out_len"][i] == len(d0) and out[o:o + len(d0)].tobytes() == d0, i
            assert res["checksum"][i] == c0, i
    assert kinds.get(0, 0) > 500 and kinds.get(1, 0) > 500, kinds


def test_deflate_batch_ragged_equals_oracle(gpu_ctx, oracle):
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    cases = util.deflate_cases()
    names = list(cases)
    streams = [cases[n] for n in names]
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(len(s)) for s in streams]
    slots = [(c + 255) // 256 * 256 for c in caps]
    dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
    src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
    d_descs = batch.to_device(descs, dev)
    total = int(sum(len(s) for s in streams))
    for level in (0, 1, 2, 3):
        for crc_op in (1, 2):
            dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
            d_res = torch.zeros(len(streams) * 16, dtype=torch.uint8, device=dev)
            batch.deflate_batch(gpu_ctx, src, dst, d_descs, d_res, len(streams), max(len(s) for s in streams),
                                total, level, crc_op)
            res = batch.results_from_device(d_res)
            out = dst.cpu().numpy()
            for i, s in enumerate(streams):
                st0, c0, k0 = oracle.deflate(s, level=level, crc_op=crc_op)
                assert res["status"][i] == 0, (names[i], level)
                o = int(dst_off[i])
                assert int(res["out_len"][i]) == len(c0), (names[i], level)
                assert out[o:o + len(c0)].tobytes() == c0, (names[i], level)
                assert res["checksum"][i] == k0, (names[i], level, crc_op)


def _long_streams_for_segments(seed):
    """Streams of many parse segments (4096 positions each): text, symbols of several entropies, and the data on
    which a parse started at a segment boundary does NOT fall in with the true one -- runs of one byte and
    periods around the longest match -- with lengths around the segment and block sizes."""
    from zipc_amd import synth

    rnd = random.Random(seed)
    out = []
    lens = [4096 * 8, 4096 * 8 + 1, 65534, 65535, 65536 + 4095, 2 * 65534 + 3, 200000, 300001, (1 << 20) + 77]
    for i, ln in enumerate(lens):
        out.append(util.text(ln, seed + i))
        out.append(synth.stream_bytes_np(9, seed + i, ln, (1, 2, 3, 4, 8)[i % 5]).tobytes())
    for k in (1, 2, 3, 7, 64, 257, 258, 259, 260, 515, 1000, 4095, 4096, 4097, 5000):
        ln = rnd.choice([70000, 131072 + 5, 300000])
        pat = synth.stream_bytes_np(9, seed + k, k, 8).tobytes()
        out.append((pat * (ln // k + 1))[:ln])
    out.append(b"\0" * ((1 << 20) + 5))
    # a periodic stretch inside text, so that paths part and meet again
    t = bytearray(util.text(400000, seed + 99))
    t[100000:180000] = b"ab" * 40000
    t[250000:300000] = bytes(50000)
    out.append(bytes(t))
    return out


def test_deflate_long_streams_by_segments_equal_oracle(gpu_ctx, oracle):
    """Few long streams: lz_parse runs as a wave per segment with a stitch (deflate.hip lz_parse_spec_kernel ...);
    bytes, block by block, against the oracle at every level."""
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    streams = _long_streams_for_segments(5)
    src_off = np.cumsum([0] + [(len(s) + 63) // 64 * 64 for s in streams[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(len(s)) for s in streams]
    slots = [(c + 255) // 256 * 256 for c in caps]
    dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
    arena = bytearray(int(src_off[-1]) + len(streams[-1]) + 64)
    for o, st in zip(src_off, streams):
        arena[int(o):int(o) + len(st)] = st
    src = torch.from_numpy(np.frombuffer(bytes(arena), dtype=np.uint8).copy()).to(dev)
    d_descs = batch.to_device(descs, dev)
    total = int(sum(len(s) for s in streams))
    for level in (1, 2, 3):
        dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(len(streams) * 16, dtype=torch.uint8, device=dev)
        batch.deflate_batch(gpu_ctx, src, dst, d_descs, d_res, len(streams), max(len(s) for s in streams), total, level, 1)
        res = batch.results_from_device(d_res)
        out = dst.cpu().numpy()
        for i, st in enumerate(streams):
            st0, c0, k0 = oracle.deflate(st, level=level, crc_op=1)
            assert res["status"][i] == 0, (i, level)
            o = int(dst_off[i])
            assert int(res["out_len"][i]) == len(c0), (i, level, len(st))
            assert out[o:o + len(c0)].tobytes() == c0, (i, level, len(st))
            assert res["checksum"][i] == k0, (i, level)


def _grouped_tile_streams(n, max_len, seed):
    """Ragged streams for the window kernel's grouped form: lengths around the multiples of
    its 16 384-position tile, data of several entropies (numpy-generated: tens of MB)."""
    from zipc_amd import synth

    rnd = random.Random(seed)
    specials = [16384 * k + d for k in range(1, max_len // 16384 + 1) for d in (-5, -4, -3, -1, 0, 1, 3, 4, 5)]
    specials = [x for x in specials if 0 < x <= max_len] + [0, 1, 3, 4, 5, max_len]
    out = []
    for i in range(n):
        ln = specials[i] if i < len(specials) else rnd.randrange(4, max_len + 1)
        kind = i % 7
        if kind < 5:
            d = synth.stream_bytes_np(9, i, ln, (1, 2, 3, 4, 8)[kind]).tobytes()
        elif kind == 5:  # periodic: long matches, l == maxlen exits, chains at the cap
            k = rnd.randrange(1, 300)
            d = (synth.stream_bytes_np(9, i, k, 8).tobytes() * (ln // k + 1))[:ln]
        else:  # two alternating regimes: chain lengths differ a lot between tiles of one stream
            a = synth.stream_bytes_np(9, i, ln, 2).tobytes()
            b = synth.stream_bytes_np(9, i, ln, 8).tobytes()
            d = b"".join((a if (q // 20000) % 2 else b)[q:q + 20000] for q in range(0, ln, 20000))
        assert len(d) == ln
        out.append(d)
    return out


def test_deflate_grouped_tiles_ragged_equals_oracle(gpu_ctx, oracle):
    """lz_match_window_kernel with several tiles per workgroup (deflate.hip: taken when
    n_streams x tiles_per_stream / 2048 > 1) on a ragged batch: streams that end inside a
    group, groups that start behind a stream's end, a group size that does not divide the
    tile count.  Every stream's bytes against the oracle.  (The full-size tests sample the
    grouped form on uniform streams only.)"""
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    n, max_len = 700, 150000
    tps = (max_len + 16383) // 16384
    tpg = max(1, min(tps, n * tps // 2048))  # the host's rule (launch_deflate)
    assert tpg > 1 and tps % tpg != 0, (tps, tpg)
    streams = _grouped_tile_streams(n, max_len, 21)
    assert max(len(s) for s in streams) == max_len
    src_off = np.cumsum([0] + [(len(s) + 15) // 16 * 16 for s in streams[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(len(s)) for s in streams]
    slots = [(c + 255) // 256 * 256 for c in caps]
    dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
    arena = np.zeros(int(src_off[-1]) + len(streams[-1]) + 64, np.uint8)
    for o, s in zip(src_off, streams):
        arena[int(o):int(o) + len(s)] = np.frombuffer(s, np.uint8)
    src = torch.from_numpy(arena).to(dev)
    d_descs = batch.to_device(descs, dev)
    total = int(sum(len(s) for s in streams))
    for level in (2, 1):
        dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        batch.deflate_batch(gpu_ctx, src, dst, d_descs, d_res, n, max_len, total, level, 1)
        res = batch.results_from_device(d_res)
        out = dst.cpu().numpy()
        for i, s in enumerate(streams):
            st0, c0, k0 = oracle.deflate(s, level=level, crc_op=oracle.CRC_CRC32)
            assert res["status"][i] == 0, (i, len(s), level)
            o = int(dst_off[i])
            assert int(res["out_len"][i]) == len(c0), (i, len(s), level)
            assert out[o:o + len(c0)].tobytes() == c0, (i, len(s), level)
            assert res["checksum"][i] == k0, (i, len(s), level)


def test_batch_declared_sizes_are_checked(gpu_ctx, oracle):
    """What include/zipc_hip.h promises about max_src_len / total_src_len / max_dst_cap and about
    descriptors beyond the format's range: the library reports ZIPC_HIP_ERR_INVALID_ARG -- it
    never compresses a stream in part or sums a part of it."""
    import torch

    from zipc_amd import _lib, batch

    INVALID = 18
    dev = torch.device("cuda", 0)
    plain = [util.text(5000, 1), util.rand_bytes(3000, 2, 3), util.text(200000, 3), util.rand_bytes(7000, 4)]
    lens = [len(p) for p in plain]
    src_off = np.cumsum([0] + [(l + 15) // 16 * 16 for l in lens[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(l) for l in lens]
    dst_off = np.cumsum([0] + [(c + 255) // 256 * 256 for c in caps[:-1]]).astype(np.uint64)
    arena = np.zeros(int(src_off[-1]) + lens[-1] + 64, np.uint8)
    for o, pl in zip(src_off, plain):
        arena[int(o):int(o) + len(pl)] = np.frombuffer(pl, np.uint8)
    src = torch.from_numpy(arena).to(dev)
    dst = torch.zeros(int(dst_off[-1]) + caps[-1] + 256, dtype=torch.uint8, device=dev)
    n = len(plain)

    def deflate(descs, max_len, total, crc_op=1):
        d_res = torch.full((n * 16,), 0xEE, dtype=torch.uint8, device=dev)
        batch.deflate_batch(gpu_ctx, src, dst, batch.to_device(descs, dev), d_res, n, max_len, total, 2, crc_op)
        return batch.results_from_device(d_res), dst.cpu().numpy()

    honest = batch.make_descs(src_off, lens, dst_off, caps)
    want = [oracle.deflate(pl, level=2, crc_op=oracle.CRC_CRC32) for pl in plain]

    def exact(res, out, which):
        for i in which:
            st0, c0, k0 = want[i]
            assert res["status"][i] == 0 and int(res["out_len"][i]) == len(c0) and res["checksum"][i] == k0, i
            assert out[int(dst_off[i]):int(dst_off[i]) + len(c0)].tobytes() == c0, i

    # honest declarations: every stream exact (the control for what follows)
    res, out = deflate(honest, max(lens), sum(lens))
    exact(res, out, range(n))
    # a stream longer than the declared max_src_len: the whole batch is refused
    for crc_op in (0, 1, 2):
        res, _ = deflate(honest, 10000, sum(lens), crc_op)
        assert (res["status"] == INVALID).all() and (res["out_len"] == 0).all(), crc_op
    # the sum longer than the declared total_src_len: the same
    res, _ = deflate(honest, max(lens), 50000)
    assert (res["status"] == INVALID).all() and (res["out_len"] == 0).all()
    # ... and the next honest call on the same context is exact again (nothing sticks)
    res, out = deflate(honest, max(lens), sum(lens))
    exact(res, out, range(n))
    # one descriptor beyond the format's range (its bytes are never touched): only that stream fails
    for too_long in (1 << 32, 0xFFFF0001):  # ZIPC_HIP_MAX_STREAM_LEN is 0xFFFF0000 (4 GiB - 64 KiB)
        lying = honest.copy()
        lying["src_len"][1] = too_long
        res, out = deflate(lying, max(lens), sum(lens) - lens[1])
        assert res["status"][1] == INVALID and res["out_len"][1] == 0, too_long
        exact(res, out, (0, 2, 3))
    lying = honest.copy()
    lying["dst_cap"][3] = 1 << 33
    res, out = deflate(lying, max(lens), sum(lens))
    assert res["status"][3] == INVALID and res["out_len"][3] == 0
    exact(res, out, (0, 1, 2))
    # max_src_len itself out of range: the call fails, nothing is launched
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    assert _lib.lib().zipc_hip_deflate_batch(gpu_ctx.handle, src.data_ptr(), dst.data_ptr(),
                                             batch.to_device(honest, dev).data_ptr(), d_res.data_ptr(), n, 1 << 32,
                                             sum(lens), 2, 1) == INVALID

    # ---- inflate side
    comp = [w[1] for w in want]
    c_off = np.cumsum([0] + [(len(c) + 15) // 16 * 16 for c in comp[:-1]]).astype(np.uint64)
    carena = np.zeros(int(c_off[-1]) + len(comp[-1]) + 64, np.uint8)
    for o, c in zip(c_off, comp):
        carena[int(o):int(o) + len(c)] = np.frombuffer(c, np.uint8)
    csrc = torch.from_numpy(carena).to(dev)
    o_off = np.cumsum([0] + [(l + 255) // 256 * 256 + 256 for l in lens[:-1]]).astype(np.uint64)
    odst = torch.zeros(int(o_off[-1]) + lens[-1] + 512, dtype=torch.uint8, device=dev)

    def inflate(descs, max_cap, crc_op=1):
        d_res = torch.full((n * 16,), 0xEE, dtype=torch.uint8, device=dev)
        batch.inflate_batch(gpu_ctx, csrc, odst, batch.to_device(descs, dev), d_res, n, max_cap, crc_op)
        return batch.results_from_device(d_res), odst.cpu().numpy()

    def plain_back(res, out, which):
        for i in which:
            assert res["status"][i] == 0 and int(res["out_len"][i]) == lens[i], i
            assert out[int(o_off[i]):int(o_off[i]) + lens[i]].tobytes() == plain[i], i
            assert res["checksum"][i] == zlib.crc32(plain[i]), i

    ihonest = batch.make_descs(c_off, [len(c) for c in comp], o_off, lens, limit=lens)
    res, out = inflate(ihonest, max(lens))
    plain_back(res, out, range(n))
    # an output longer than the declared max_dst_cap, CRC-32 asked for: that stream is refused, not half-summed
    res, out = inflate(ihonest, 40000)
    assert res["status"][2] == INVALID
    plain_back(res, out, (0, 1, 3))
    # descriptors beyond the range
    for field, too_long in (("src_len", 1 << 32), ("dst_cap", 1 << 32), ("src_len", 0xFFFF0001), ("dst_cap", 0xFFFF0001)):
        lying = ihonest.copy()
        lying[field][0] = too_long
        res, out = inflate(lying, max(lens))
        assert res["status"][0] == INVALID and res["out_len"][0] == 0, field
        plain_back(res, out, (1, 2, 3))


def test_deflate_dst_too_small_is_reported(gpu_ctx):
    import ctypes as C

    from zipc_amd import _lib

    d = util.rand_bytes(5000, 1)
    dst = C.create_string_buffer(100)
    ol, ck = C.c_size_t(), C.c_uint32()
    st = _lib.lib().zipc_hip_deflate(gpu_ctx.handle, d, len(d), 2, 0, dst, 100, C.byref(ol), C.byref(ck))
    assert st == _lib.ERR_DST_TOO_SMALL


def test_deflate_dst_too_small_is_reported_for_a_long_stream(gpu_ctx, oracle):
    """the same through the forms a long stream takes (blocks coded by a wave each, deflate_scan_kernel decides):
    a capacity below the output fails the stream and leaves the bytes behind the capacity alone; the exact size fits
    when the reference's own estimates let it (they run high: the code-length counts are never reset)"""
    import ctypes as C

    from zipc_amd import _lib

    d = util.text(300000, 3)
    st0, c0, k0 = oracle.deflate(d, level=2, crc_op=1)
    for cap, ok in ((len(c0) // 2, False), (len(c0) - 1, False), (len(c0) + 600, True)):
        dst = C.create_string_buffer(b"\xA5" * (cap + 64), cap + 64)
        ol, ck = C.c_size_t(), C.c_uint32()
        st = _lib.lib().zipc_hip_deflate(gpu_ctx.handle, d, len(d), 2, 1, dst, cap, C.byref(ol), C.byref(ck))
        assert dst.raw[cap:] == b"\xA5" * 64, cap
        if ok:
            assert st == 0 and dst.raw[:ol.value] == c0 and ck.value == k0
        else:
            assert st == _lib.ERR_DST_TOO_SMALL, cap


def test_deflate_start_offset_is_the_requested_range(gpu_ctx, oracle):
    """`?start ?len` on the encode side: the reference's Lz77.compress mixes absolute and relative
    indices when start > 0 (zipc_deflate.ml:1206,1215,1220: it would encode [start, len) instead of
    [start, start + len); no caller passes start > 0 to deflate) -- the boundary deliberately
    compresses exactly the requested range, and this pins it."""
    from zipc_amd import zipc_deflate as Z

    data = util.text(40000, 5) + util.rand_bytes(30000, 6, 4)
    for start, n in ((0, len(data)), (100, 5000), (39990, 20000), (69999, 1), (70000, 0), (12345, 57655)):
        for level, lv in LEVELS.items():
            want = oracle.deflate(data[start:start + n], level=lv)[1]
            assert Z.deflate(data, level=level, start=start, len=n).get_ok() == want, (start, n, level)
        adler, z = Z.zlib_compress(data, level="default", start=start, len=n).get_ok()
        assert zlib.decompress(z) == data[start:start + n]  # (text and 4-bit bytes: the two Adler-32s agree)
        assert adler == zlib.adler32(data[start:start + n])


def _one_stream_sources(n):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "SURVEY.md"), "rb").read() + open(os.path.join(root, "BASELINE.md"), "rb").read()
    rng = np.random.default_rng(21)
    yield "text", (text * (n // len(text) + 1))[:n]
    yield "symbols", (rng.integers(0, 16, n, dtype=np.uint8) * 17).astype(np.uint8).tobytes()
    yield "records", (rng.integers(0, 1 << 14, n // 4 + 1, dtype=np.uint32) * np.uint32(0x10001)).tobytes()[:n]


def _raw_zlib(data, level, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""  # (a full flush ends the block and puts an empty stored block behind it: stored blocks inside the chain)
    for i in range(0, len(data), flush_every):
        out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)
    return out + c.flush()


def test_one_stream_inflates_by_a_wave_per_block(gpu_ctx, oracle):
    """ONE stream per call -- the shape of the reference's inflate (zipc_deflate.mli:79-97) -- long enough to be
    decoded by a wave per block (inflate.hip: find, dry, chain, token, resolve): streams of the reference's encoder
    (dynamic blocks, a fixed block at the end) and of zlib (other block sizes; with full flushes: empty stored
    blocks in the chain), bytes and CRC-32 equal to the source's, and the path really taken."""
    from zipc_amd import zipc_deflate as Z

    n = 1200000
    for name, data in _one_stream_sources(n):
        streams = [("oracle-2", oracle.deflate(data, level=2)[1]), ("zlib-1", _raw_zlib(data, 1)), ("zlib-6", _raw_zlib(data, 6)),
                   ("zlib-9-flushes", _raw_zlib(data, 9, 200000))]
        for enc, raw in streams:
            got, crc = Z.inflate_and_crc_32(raw, decompressed_size=n).get_ok()
            assert gpu_ctx.last_inflate_blocks() >= 6, (name, enc, gpu_ctx.last_inflate_blocks())
            assert got == data and crc == zlib.crc32(data), (name, enc)
            # Adler-32: block by block, every block's bytes in 5552-byte chunks of their own (zd.ml:682-690)
            got2, adler = Z.inflate_and_adler_32(raw, decompressed_size=n).get_ok()
            assert gpu_ctx.last_inflate_blocks() >= 6, (name, enc)
            assert got2 == data and adler == oracle.inflate(raw, decompressed_size=n, crc_op=oracle.CRC_ADLER32)[2], (name, enc)
            # no size given: the first tries are too small (the stream's one wave says so), the last one fits
            assert Z.inflate(raw).get_ok() == data, (name, enc)


def test_a_call_of_long_and_short_streams(gpu_ctx, oracle):
    """zipc_hip_inflate_batch with streams of every kind in ONE call: long ones of the reference's encoder and of zlib (by
    blocks, side by side), a long one of fixed blocks only and one of stored blocks only (the block path may refuse
    them), short ones, an empty stored stream, a damaged long one, a cut one and one whose limit is too small (their
    one waves own the message) -- status, length, bytes and checksum of every stream as the oracle has them, for
    CRC-32, Adler-32 and none; the bytes behind every stream's output untouched."""
    import torch
    from zipc_amd import batch

    n = 1100000
    plain = {k: v for k, v in _one_stream_sources(n)}
    text, symbols, records = plain["text"], plain["symbols"], plain["records"]
    fixed = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    stored = zlib.compressobj(0, zlib.DEFLATED, -15)
    good = _raw_zlib(text, 6)
    at = len(good) // 2
    cases = [("oracle-2 text", oracle.deflate(text, level=2)[1], n), ("zlib-6 text", good, n), ("short", _raw_zlib(text[:3000], 6), 3000),
             ("zlib-1 symbols", _raw_zlib(symbols, 1), n), ("fixed", fixed.compress(records) + fixed.flush(), n),
             ("empty", _raw_zlib(b"", 6), 0), ("stored", stored.compress(symbols[:400000]) + stored.flush(), 400000),
             ("damaged", good[:at] + bytes([good[at] ^ 0x10]) + good[at + 1:], n), ("zlib-9 records, flushes", _raw_zlib(records, 9, 300000), n),
             ("cut", good[:len(good) * 3 // 4], n), ("limit short", _raw_zlib(symbols, 6), n - 1), ("oracle-1 records", oracle.deflate(records, level=1)[1], n)]
    dev = torch.device("cuda", 0)
    caps = [max(lim, 8) + 64 for _, _, lim in cases]
    src_off = np.cumsum([0] + [(len(raw) + 7) & ~7 for _, raw, _ in cases])
    dst_off = np.cumsum([0] + [c + 13 for c in caps])  # (outputs at odd addresses)
    arena = np.zeros(int(src_off[-1]) + 64, np.uint8)
    for (_, raw, _), o in zip(cases, src_off):
        arena[o:o + len(raw)] = np.frombuffer(raw, np.uint8)
    descs = batch.make_descs(src_off[:-1], [len(raw) for _, raw, _ in cases], dst_off[:-1], caps, limit=[lim for _, _, lim in cases])
    src = torch.from_numpy(arena).to(dev)
    d_descs = batch.to_device(descs, dev)
    for crc_op in (oracle.CRC_CRC32, oracle.CRC_ADLER32, oracle.CRC_NOP):
        dst = torch.full((int(dst_off[-1]) + 64,), 0xA5, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(16 * len(cases), dtype=torch.uint8, device=dev)
        batch.inflate_batch(gpu_ctx, src, dst, d_descs, d_res, len(cases), max(caps), crc_op)
        assert gpu_ctx.last_inflate_blocks() >= 60, gpu_ctx.last_inflate_blocks()
        res = batch.results_from_device(d_res)
        out = dst.cpu().numpy()
        for i, (name, raw, lim) in enumerate(cases):
            st0, d0, c0 = oracle.inflate(raw, decompressed_size=lim, crc_op=crc_op)
            assert int(res["status"][i]) == st0, (name, crc_op, res[i])
            if st0 == 0:
                assert int(res["out_len"][i]) == len(d0) and out[dst_off[i]:dst_off[i] + len(d0)].tobytes() == d0, (name, crc_op)
                assert crc_op == oracle.CRC_NOP or int(res["checksum"][i]) == c0, (name, crc_op)
            assert bool((out[dst_off[i] + caps[i]:dst_off[i + 1]] == 0xA5).all()), (name, crc_op)


def test_the_long_members_among_thousands_of_short_ones(gpu_ctx, oracle):
    """An archive's shape: 4 300 short members and six long ones in ONE call.  The long ones go by blocks, the short
    ones by their one waves in two slices on two queues over the marked copy of the descriptors, one CRC-32 pass over
    all of them -- every member's bytes and CRC-32 as zlib has them, the damaged long one as the oracle has it."""
    import torch
    from zipc_amd import batch

    rnd = random.Random(31)
    plain = {k: v for k, v in _one_stream_sources(1300000)}
    longs = [plain["text"], plain["records"], plain["symbols"][:1100000], plain["text"][200000:1250000], plain["records"][:1048576]]
    members = []
    for i in range(4300):
        n = rnd.choice((0, 1, 17, 300, 2000, 5000))
        o = rnd.randrange(1000000)
        members.append(plain["text"][o:o + n] if i % 3 else plain["symbols"][o:o + n])
    at = [700, 701, 2222, 4000, 4299]
    for k, a in enumerate(at):
        members[a] = longs[k]
    raws = [_raw_zlib(m, 6) for m in members]
    bad = 3000  # a long member with a flipped bit
    members[bad] = plain["text"][:1200000]
    good = _raw_zlib(members[bad], 6)
    raws[bad] = good[:len(good) // 2] + bytes([good[len(good) // 2] ^ 4]) + good[len(good) // 2 + 1:]
    n = len(members)
    dev = torch.device("cuda", 0)
    caps = [max(len(m), 8) for m in members]
    src_off = np.cumsum([0] + [(len(r) + 3) & ~3 for r in raws])
    dst_off = np.cumsum([0] + [(c + 15) & ~15 for c in caps])
    arena = np.zeros(int(src_off[-1]) + 64, np.uint8)
    for r, o in zip(raws, src_off):
        arena[o:o + len(r)] = np.frombuffer(r, np.uint8)
    descs = batch.make_descs(src_off[:-1], [len(r) for r in raws], dst_off[:-1], caps, limit=[len(m) for m in members])
    src = torch.from_numpy(arena).to(dev)
    dst = torch.zeros(int(dst_off[-1]) + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
    batch.inflate_batch(gpu_ctx, src, dst, batch.to_device(descs, dev), d_res, n, max(caps), oracle.CRC_CRC32)
    assert gpu_ctx.last_inflate_blocks() >= 40, gpu_ctx.last_inflate_blocks()
    res = batch.results_from_device(d_res)
    out = dst.cpu().numpy()
    st_bad, d_bad, c_bad = oracle.inflate(raws[bad], decompressed_size=len(members[bad]), crc_op=oracle.CRC_CRC32)
    assert int(res["status"][bad]) == st_bad  # (a flipped bit may leave a stream that still decodes, to other bytes)
    if st_bad == 0:
        assert int(res["out_len"][bad]) == len(d_bad) and int(res["checksum"][bad]) == c_bad
        assert out[dst_off[bad]:dst_off[bad] + len(d_bad)].tobytes() == d_bad
    for i in range(n):
        if i == bad:
            continue
        assert int(res["status"][i]) == 0 and int(res["out_len"][i]) == len(members[i]), (i, res[i])
        assert int(res["checksum"][i]) == zlib.crc32(members[i]), i
        if i in at or i % 97 == 0:
            assert out[dst_off[i]:dst_off[i] + len(members[i])].tobytes() == members[i], i


def test_one_stream_by_blocks_leaves_errors_to_the_streams_wave(gpu_ctx, oracle):
    """what the block path must not decide: damaged streams, cut streams, sizes that do not fit -- status and
    message of the oracle, whatever the blocks before the damage were decoded by"""
    from zipc_amd import zipc_deflate as Z

    n = 600000
    rnd = random.Random(8)
    for name, data in _one_stream_sources(n):
        raw = _raw_zlib(data, 6)
        cases = [("cut", raw[:len(raw) * 2 // 3], n), ("one byte short", raw[:-1], n), ("limit one short", raw, n - 1),
                 ("limit far short", raw, n // 2), ("garbage behind", raw + bytes(rnd.randrange(256) for _ in range(999)), n)]
        for k in range(6):
            at = rnd.randrange(len(raw) // 8, len(raw))
            cases.append(("flipped bit at %d" % at, raw[:at] + bytes([raw[at] ^ (1 << rnd.randrange(8))]) + raw[at + 1:], n))
        for what, s, lim in cases:
            st0, d0, c0 = oracle.inflate(s, decompressed_size=lim, crc_op=oracle.CRC_CRC32)
            r = Z.inflate_and_crc_32(s, decompressed_size=lim)
            assert r.is_ok() == (st0 == 0), (name, what)
            if st0 == 0:
                assert r.get_ok() == (d0, c0), (name, what)
            else:
                assert r.error == oracle.MESSAGES[st0], (name, what)


def test_one_stream_by_blocks_every_kind_of_block(gpu_ctx, oracle):
    """the block path over streams of every make zlib has: fixed blocks only (nothing to search for: explorers),
    stored only, Huffman-only and run-length strategies, small blocks by the thousand (memLevel 1), full flushes --
    bytes and CRC-32 of the source whichever way a stream ends up being decoded, and the block path taken where its
    blocks can be found"""
    from zipc_amd import zipc_deflate as Z

    n = 700000
    for name, data in _one_stream_sources(n):
        makes = [("fixed", dict(level=6, strategy=zlib.Z_FIXED), True), ("stored", dict(level=0), False),
                 ("huffman-only", dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), True), ("rle", dict(level=6, strategy=zlib.Z_RLE), True),
                 ("filtered", dict(level=9, strategy=zlib.Z_FILTERED), True), ("memlevel-1", dict(level=6, memLevel=1), False),
                 ("memlevel-3", dict(level=1, memLevel=3), False)]
        for make, kw, by_blocks in makes:
            c = zlib.compressobj(kw.get("level", 6), zlib.DEFLATED, -15, kw.get("memLevel", 8), kw.get("strategy", zlib.Z_DEFAULT_STRATEGY))
            raw = b"".join(c.compress(data[i:i + 150000]) + c.flush(zlib.Z_FULL_FLUSH if make == "rle" else zlib.Z_NO_FLUSH)
                           for i in range(0, n, 150000)) + c.flush()
            got, crc = Z.inflate_and_crc_32(raw, decompressed_size=n).get_ok()
            assert got == data and crc == zlib.crc32(data), (name, make)
            if by_blocks:
                assert gpu_ctx.last_inflate_blocks() >= 4, (name, make, gpu_ctx.last_inflate_blocks())
            st0, d0, _ = oracle.inflate(raw[:len(raw) // 2], decompressed_size=n)
            r = Z.inflate(raw[:len(raw) // 2], decompressed_size=n)
            assert (not r.is_ok()) and st0 != 0 and r.error == oracle.MESSAGES[st0], (name, make)


def test_unknown_descriptor_flags_are_an_invalid_argument(gpu_ctx, oracle):
    """include/zipc_hip.h: bits of a descriptor's flags other than ZIPC_HIP_STREAM_HAS_LIMIT must be zero.  Bit 31 is the
    mark the library sets in its OWN copy of a call's descriptors (a stream that went by blocks); in a caller's descriptor
    it used to skip the stream and leave its result slot as it was.  Now the stream reports ZIPC_HIP_ERR_INVALID_ARG, in
    calls of a few streams (the kernel that shares tables) and of many, and its neighbours are decoded as ever."""
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    data = util.text(20000, 5)
    comp = oracle.deflate(data, level=2)[1]
    for n in (3, 300):
        src = torch.from_numpy(np.frombuffer(comp * n + b"\0" * 64, np.uint8).copy()).to(dev)
        descs = batch.make_descs(np.arange(n, dtype=np.uint64) * len(comp), [len(comp)] * n, np.arange(n, dtype=np.uint64) * 20480, [20000] * n)
        descs["flags"][1] = 1 << 31
        descs["flags"][2] = 2
        dst = torch.zeros(n * 20480, dtype=torch.uint8, device=dev)
        d_res = torch.full((n * 16,), 0xAB, dtype=torch.uint8, device=dev)
        batch.inflate_batch(gpu_ctx, src, dst, batch.to_device(descs, dev), d_res, n, 20000, 1)
        res = batch.results_from_device(d_res)
        assert int(res["status"][1]) == 18 and int(res["status"][2]) == 18, (n, res["status"][:4])
        assert int(res["status"][0]) == 0 and int(res["out_len"][0]) == len(data)
        assert dst[:len(data)].cpu().numpy().tobytes() == data
        assert not bool(dst[20480:20480 * 3].any())
        if n > 3:
            assert (res["status"][3:] == 0).all() and dst[20480 * (n - 1):20480 * (n - 1) + len(data)].cpu().numpy().tobytes() == data


# ---- double entry for the encode side (round 6): the GPU against tests/golden/deflate_vectors.json -------------------
# The file is what a second, independent reading of src/zipc_deflate.ml's encoder makes of 50 named inputs (see
# tests/test_oracle_pins.py, tests/golden/make_deflate_vectors.py): no oracle in these two tests -- the kernels' bytes are held
# directly against the hashes, through the one-stream host forms and through one ragged call of the batch form.

def _vectors():
    import json
    import os

    return json.load(open(os.path.join(util.GOLDEN, "deflate_vectors.json")))


def test_deflate_host_forms_equal_the_second_readings_vectors(gpu_ctx):
    import hashlib

    from zipc_amd import zipc_deflate as Z

    doc = _vectors()
    checked = 0
    for name, v in doc["vectors"].items():
        data = util.vector_input(name)
        for level, want in v["levels"].items():
            crc, cs = Z.crc_32_and_deflate(data, level=level).get_ok()
            assert (len(cs), hashlib.sha256(cs).hexdigest(), crc) == (want["clen"], want["sha256"], want["crc32"]), (name, level)
            adler, cs2 = Z.adler_32_and_deflate(data, level=level).get_ok()
            assert cs2 == cs and adler == want["adler32_fused"], (name, level)  # per-block chaining, signed remainder (Q6/Q7)
            checked += 1
    assert checked >= 130
    for name, want in doc["adler32_whole"].items():
        assert Z.Adler_32.string(util.vector_input(name)) == want, name


def test_deflate_batch_form_equals_the_second_readings_vectors(gpu_ctx):
    import hashlib

    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    doc = _vectors()
    for level, lv in (("fast", 1), ("default", 2), ("best", 3)):
        names = [n for n, v in doc["vectors"].items() if level in v["levels"]]
        streams = [util.vector_input(n) for n in names]
        src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
        caps = [batch.deflate_bound(len(s)) for s in streams]
        slots = [(c + 255) // 256 * 256 for c in caps]
        dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
        descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
        src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
        d_descs = batch.to_device(descs, dev)
        total = int(sum(len(s) for s in streams))
        for crc_op, key in ((1, "crc32"), (2, "adler32_fused")):
            dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
            d_res = torch.zeros(len(streams) * 16, dtype=torch.uint8, device=dev)
            batch.deflate_batch(gpu_ctx, src, dst, d_descs, d_res, len(streams), max(len(s) for s in streams), total, lv, crc_op)
            res = batch.results_from_device(d_res)
            out = dst.cpu().numpy()
            for i, n in enumerate(names):
                want = doc["vectors"][n]["levels"][level]
                assert res["status"][i] == 0 and int(res["out_len"][i]) == want["clen"], (n, level)
                o = int(dst_off[i])
                assert hashlib.sha256(out[o:o + want["clen"]].tobytes()).hexdigest() == want["sha256"], (n, level)
                assert int(res["checksum"][i]) == want[key], (n, level, key)
