"""BASELINE.json's full-size configurations, checked through size-independent
properties on the device plus oracle samples (the oracle would need minutes for
the whole GiB)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _c2_round_trip(gpu_ctx, oracle, n_streams, bits, level, config, sample):
    import torch

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    L = 65536
    src = synth.batch_bytes_torch(config, 0, n_streams, L, bits, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n_streams, L, cap)
    slot = int(descs["dst_off"][1]) if n_streams > 1 else cap
    comp = torch.zeros(n_streams * slot + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n_streams * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(gpu_ctx, src, comp, d_descs, d_res, n_streams, L, n_streams * L, level, 1)
    res = batch.results_from_device(d_res)
    assert (res["status"] == 0).all()
    # inverse: compressed slots -> fresh output arena, limit = original length
    idescs = batch.compact_descs(res, descs, L)
    out = torch.zeros(n_streams * L + 256, dtype=torch.uint8, device=dev)
    d_idescs = batch.to_device(idescs, dev)
    d_ires = torch.zeros(n_streams * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(gpu_ctx, comp, out, d_idescs, d_ires, n_streams, L, 1)
    ires = batch.results_from_device(d_ires)
    assert (ires["status"] == 0).all() and (ires["out_len"] == L).all()
    assert torch.equal(out[:n_streams * L], src)              # encode -> decode round trip, every byte
    assert (ires["checksum"] == res["checksum"]).all()        # CRC of output == CRC of input, per stream
    # oracle samples: compressed bytes and CRC bit-exact
    comp_h = None
    rng = np.random.default_rng(1)
    for j in sorted(set([0, n_streams - 1] + rng.integers(0, n_streams, size=sample).tolist())):
        plain = synth.stream_bytes_np(config, j, L, bits).tobytes()
        st, c0, crc0 = oracle.deflate(plain, level=level, crc_op=oracle.CRC_CRC32)
        o = int(descs["dst_off"][j])
        got = comp[o:o + int(res["out_len"][j])].cpu().numpy().tobytes()
        assert got == c0, j
        assert int(res["checksum"][j]) == crc0
    return res


def test_config2_full_1gib_default_round_trip(gpu_ctx, oracle):
    """C2: 16 384 x 64 KiB of 4-bit symbols, level `Default, deflate then inflate."""
    res = _c2_round_trip(gpu_ctx, oracle, 16384, 4, 2, 2, sample=6)
    ratio = res["out_len"].sum() / (16384 * 65536)
    assert 0.5 < ratio < 0.6  # SURVEY.md 8d expects ~0.545


def test_config2_full_1gib_level_best(gpu_ctx, oracle):
    """The same 16 384 streams at `Best -- the level a caller of the seam gets who passes none
    (make_encoder ?(level = `Best), zd.ml:817): K = 4096 chain steps, good_match 32."""
    res = _c2_round_trip(gpu_ctx, oracle, 16384, 4, 3, 2, sample=6)
    ratio = res["out_len"].sum() / (16384 * 65536)
    assert 0.5 < ratio < 0.6


def test_config5_stored_heavy(gpu_ctx, oracle):
    """C5 shape at 1/16 scale: uniform random bytes -> stored 65534 + 2-literal fixed block."""
    res = _c2_round_trip(gpu_ctx, oracle, 8192, 8, 2, 5, sample=3)
    assert (res["out_len"] == 65543).all()


def test_config3_checksum_large_buffer(gpu_ctx, oracle):
    """C3 shape at 1/16 scale (256 MiB): CRC-32 + Adler-32 of one random buffer; the
    oracle takes ~1 s for this size.  Linearity check: CRC of halves combines."""
    import torch
    import zlib

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    n = 256 << 20
    buf = synth.batch_bytes_torch(3, 0, 1, n, 8, dev)
    crc, adler = batch.checksum_device(gpu_ctx, buf)
    host = buf.cpu().numpy()
    assert crc == zlib.crc32(host)
    assert crc == oracle.crc32(host)
    assert adler == oracle.adler32(host)  # Q6 fires in about half of the 48k chunks
    # odd length / unaligned start
    crc2, adler2 = batch.checksum_device(gpu_ctx, buf[3:n - 5])
    assert crc2 == oracle.crc32(host[3:n - 5]) and adler2 == oracle.adler32(host[3:n - 5])


def test_config4_one_mib_members_default(gpu_ctx, oracle):
    """C4 shape at 1/8 scale: 1 024 members x 1 MiB of 3-bit symbols, crc_32_and_deflate
    level `Default (17 blocks per member, Q1 carries across blocks); samples bit-exact
    vs the oracle, every member round-trips."""
    import torch

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    n, L = 1024, 1 << 20
    src = synth.batch_bytes_torch(4, 0, n, L, 3, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1])
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(gpu_ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 1)
    res = batch.results_from_device(d_res)
    assert (res["status"] == 0).all()
    ratio = res["out_len"].sum() / (n * L)
    assert 0.38 < ratio < 0.48  # SURVEY.md 8d estimates ~0.43
    for j in (0, 511, n - 1):
        plain = synth.stream_bytes_np(4, j, L, 3).tobytes()
        st, c0, crc0, blocks = oracle.deflate_trace(plain, level=2, crc_op=oracle.CRC_CRC32)
        assert len(blocks) == 17
        o = int(descs["dst_off"][j])
        assert comp[o:o + int(res["out_len"][j])].cpu().numpy().tobytes() == c0, j
        assert int(res["checksum"][j]) == crc0
    idescs = batch.compact_descs(res, descs, L)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_idescs = batch.to_device(idescs, dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(gpu_ctx, comp, out, d_idescs, d_ires, n, L, 1)
    ires = batch.results_from_device(d_ires)
    assert (ires["status"] == 0).all() and torch.equal(out[:n * L], src)
    assert (ires["checksum"] == res["checksum"]).all()


def test_config3_full_4gib_checksums(gpu_ctx, oracle):
    """C3 at full size: CRC-32 + Adler-32 of one 4 GiB random buffer (773 590 Adler
    chunks, the signed remainder fires in about half of them)."""
    import torch

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    n = 4 << 30
    buf = synth.batch_bytes_torch(3, 0, 1, n, 8, dev)
    crc, adler = batch.checksum_device(gpu_ctx, buf)
    host = buf.cpu().numpy()
    assert crc == oracle.crc32(host)
    assert adler == oracle.adler32(host)
    # linearity: the halves' CRCs combine to the whole (chained update on the oracle side)
    half = n // 2
    c1, _ = batch.checksum_device(gpu_ctx, buf[:half], want_adler32=False)
    st = oracle.crc32_update(c1 ^ 0xFFFFFFFF, host[half:]) ^ 0xFFFFFFFF
    assert st == crc


def _full_round_trip(gpu_ctx, oracle, config, n, L, bits, level, samples, blocks_per_member=None):
    """deflate + inflate of n streams of L bytes, arenas laid out back to back (offsets pass 2^32 when
    n * L does); every byte round-trips, per-stream CRCs agree, sampled streams are the oracle's bytes"""
    import torch

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    src = synth.batch_bytes_torch(config, 0, n, L, bits, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1])
    assert int(descs["dst_off"][-1]) > 1 << 32 and int(descs["src_off"][-1]) > 1 << 32
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(gpu_ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
    res = batch.results_from_device(d_res)
    assert (res["status"] == 0).all()
    for j in samples:
        plain = synth.stream_bytes_np(config, j, L, bits).tobytes()
        st, c0, crc0, blocks = oracle.deflate_trace(plain, level=level, crc_op=oracle.CRC_CRC32)
        if blocks_per_member:
            assert len(blocks) == blocks_per_member
        o = int(descs["dst_off"][j])
        assert comp[o:o + int(res["out_len"][j])].cpu().numpy().tobytes() == c0, j
        assert int(res["checksum"][j]) == crc0
    idescs = batch.compact_descs(res, descs, L)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(idescs, dev), d_ires, n, L, 1)
    ires = batch.results_from_device(d_ires)
    assert (ires["status"] == 0).all() and (ires["out_len"] == L).all()
    assert torch.equal(out[:n * L], src)                  # every byte of the 8 GiB
    assert (ires["checksum"] == res["checksum"]).all()    # CRC-32 of output == CRC-32 of source, per stream
    return res


def test_config4_full_8192_members(gpu_ctx, oracle):
    """C4 at full size on one GPU: 8192 members x 1 MiB of 3-bit symbols (8 GiB arenas, offsets past
    2^32), crc_32_and_deflate level `Default, then inflate; members from the start, the 4 GiB line and
    the end are the oracle's bytes (17 blocks each, Q1 carries across them)."""
    res = _full_round_trip(gpu_ctx, oracle, 4, 8192, 1 << 20, 3, 2, samples=(0, 4095, 4096, 4097, 8191), blocks_per_member=17)
    ratio = res["out_len"].sum() / (8192 * (1 << 20))
    assert 0.38 < ratio < 0.48


def test_config5_full_131072_streams(gpu_ctx, oracle):
    """C5 at full size: 131 072 streams x 64 KiB of uniform random bytes (8 GiB): each deflates to a
    stored block of 65 534 bytes + a fixed block of 2 literals, and inflates back."""
    res = _full_round_trip(gpu_ctx, oracle, 5, 131072, 65536, 8, 2, samples=(0, 65535, 65536, 65537, 131071))
    # (a stream in a few hundred holds a 4-byte repeat whose match tips a block's cost: not all are 65 543)
    assert (res["out_len"] == 65543).mean() > 0.98 and (np.abs(res["out_len"].astype(np.int64) - 65543) < 64).all()


def _stored_stream(src, block=65534):
    """`src` (uint8 cuda tensor) as ONE deflate stream of stored blocks of `block` bytes (the last one
    shorter and final): what the reference's encoder makes of incompressible data"""
    import torch

    n = src.numel()
    J = n // block
    rest = n - J * block
    out = torch.empty(n + 5 * (J + 1), dtype=torch.uint8, device=src.device)
    body = out[:J * (block + 5)].view(J, block + 5)
    body[:, 0] = 0
    body[:, 1] = block & 255
    body[:, 2] = block >> 8
    body[:, 3] = (~block) & 255
    body[:, 4] = ((~block) >> 8) & 255
    body[:, 5:] = src[:J * block].view(J, block)
    tail = out[J * (block + 5):]
    tail[0] = 1
    tail[1] = rest & 255
    tail[2] = rest >> 8
    tail[3] = (~rest) & 255
    tail[4] = ((~rest) >> 8) & 255
    tail[5:] = src[J * block:]
    return out, J


def test_config5_single_8gib_stream_of_stored_blocks(gpu_ctx, oracle):
    """C5's secondary form: the same 8 GiB as ONE stream, 131 076 stored blocks of 65 534 bytes and a
    short final one -- beyond the 32-bit positions of the batch kernel, so the chain of equal stored
    blocks is found and copied with 64-bit offsets (inflate.hip, stored_chain_*).  Every byte comes
    back, the CRC-32 of the output is the source's, the reference's errors are reported where the
    chain is damaged near its end or the limit is a byte short; the first blocks, cut out as a stream
    of their own, are what the oracle makes of them."""
    import torch

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    n = 8 << 30
    src = synth.batch_bytes_torch(5, 0, 1, n, 8, dev)
    comp, J = _stored_stream(src)
    assert J == 131076
    out = torch.zeros(n + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(16, dtype=torch.uint8, device=dev)

    def run(limit, cap=n):
        descs = batch.make_descs([0], [comp.numel()], [0], [cap], limit=None if limit is None else [limit])
        batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(descs, dev), d_res, 1, cap, 1)
        return batch.results_from_device(d_res)

    res = run(n)
    assert int(res["status"][0]) == 0 and int(res["out_len"][0]) == n
    assert torch.equal(out[:n], src)
    crc, _ = batch.checksum_device(gpu_ctx, src, want_adler32=False)
    assert int(res["checksum"][0]) == crc
    assert int(run(None)["status"][0]) == 0                       # no ?decompressed_size
    assert int(run(n - 1)["status"][0]) == 2                      # "Expected decompression size exceeded"
    # the oracle on the stream's first three blocks (made final), and the GPU's ordinary path on the same
    head = bytearray(comp[:3 * 65539].cpu().numpy().tobytes())
    head[2 * 65539] |= 1
    st0, d0, c0 = oracle.inflate(bytes(head), crc_op=oracle.CRC_CRC32)
    assert st0 == 0 and d0 == src[:3 * 65534].cpu().numpy().tobytes()
    # a damaged NLEN ten blocks before the end: the chain stops there, the rest is the kernel's and is refused
    at = (J - 10) * 65539 + 3
    saved = int(comp[at])
    comp[at] = saved ^ 0x40
    res = run(n)
    assert int(res["status"][0]) == 1 and int(res["out_len"][0]) == 0  # "Corrupted data stream"
    comp[at] = saved
    # damaged early: the reference's message here too (the walk over the headers finds it; round 2 refused the
    # stream as an argument because what follows the damage is beyond the batch kernel's range)
    at = 1000 * 65539 + 3
    saved = int(comp[at])
    comp[at] = saved ^ 0x40
    res = run(n)
    assert int(res["status"][0]) == 1 and int(res["out_len"][0]) == 0
    comp[at] = saved
    # a compressed block early in a stream this long cannot be taken (32-bit positions): refused as an argument
    at = 1000 * 65539
    saved = int(comp[at])
    comp[at] = saved | 2  # BTYPE 01
    assert int(run(n)["status"][0]) == 18
    comp[at] = saved


def _stored_stream_of(src, lens):
    """src as stored blocks of the given lengths (their sum = len(src)), the last one final, on the device"""
    import torch

    parts = []
    at = 0
    for k, L in enumerate(lens):
        h = torch.tensor([1 if k + 1 == len(lens) else 0, L & 255, L >> 8, (~L) & 255, ((~L) >> 8) & 255], dtype=torch.uint8, device=src.device)
        parts += [h, src[at:at + L]]
        at += L
    assert at == src.numel()
    return torch.cat(parts)


def test_huge_stream_of_stored_blocks_of_any_lengths(gpu_ctx, oracle):
    """Beyond 4 GiB with blocks of MIXED lengths: runs of equal blocks of several lengths, single odd blocks,
    empty blocks, a short final one -- the walk over the headers (inflate.hip, stored_walk_kernel) lists what the
    all-at-once check of equal blocks cannot take.  Bytes, CRC-32, the reference's statuses; the first blocks as
    a stream of their own against the oracle."""
    import random

    import torch

    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    r = random.Random(9)
    lens = []
    total = 0
    target = (1 << 32) + (1 << 27)
    while total < target:
        kind = r.randrange(5)
        run = [65534] * r.randrange(200, 6000) if kind < 2 else [r.choice([65535, 40000, 1, 17, 65534])] * r.randrange(1, 900) if kind < 4 \
            else [r.randrange(0, 65536) for _ in range(r.randrange(1, 30))] + [0, 0]
        lens += run
        total += sum(run)
    while total % 8:
        lens.append(1)
        total += 1
    src = synth.batch_bytes_torch(5, 11, 1, total, 8, dev)
    comp = _stored_stream_of(src, lens)
    out = torch.zeros(total + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(16, dtype=torch.uint8, device=dev)

    def run_(limit, cap=total):
        descs = batch.make_descs([0], [comp.numel()], [0], [cap], limit=None if limit is None else [limit])
        batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(descs, dev), d_res, 1, cap, 1)
        return batch.results_from_device(d_res)

    res = run_(total)
    assert int(res["status"][0]) == 0 and int(res["out_len"][0]) == total
    assert torch.equal(out[:total], src)
    crc, _ = batch.checksum_device(gpu_ctx, src, want_adler32=False)
    assert int(res["checksum"][0]) == crc
    assert int(run_(total - 1)["status"][0]) == 2   # "Expected decompression size exceeded"
    assert int(run_(None, cap=total - 1)["status"][0]) == 16  # no ?decompressed_size: the caller's buffer is what is short
    # a damaged NLEN in a block of an odd length, far into the stream
    pos = 0
    k = len(lens) * 2 // 3
    for L in lens[:k]:
        pos += 5 + L
    saved = int(comp[pos + 4])
    comp[pos + 4] = saved ^ 1
    res = run_(total)
    assert int(res["status"][0]) == 1 and int(res["out_len"][0]) == 0
    comp[pos + 4] = saved
    # cut short inside the last block's bytes: "Corrupted data stream" (zd.ml:676)
    descs = batch.make_descs([0], [comp.numel() - 1], [0], [total], limit=[total])
    batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(descs, dev), d_res, 1, total, 1)
    assert int(batch.results_from_device(d_res)["status"][0]) == 1
    # the oracle on the first blocks (made final)
    n_head = 40
    end = sum(5 + L for L in lens[:n_head])
    head = bytearray(comp[:end].cpu().numpy().tobytes())
    head[end - 5 - lens[n_head - 1]] |= 1
    st0, d0, _ = oracle.inflate(bytes(head))
    assert st0 == 0 and d0 == src[:sum(lens[:n_head])].cpu().numpy().tobytes()


def test_huge_stored_stream_through_the_host_form(gpu_ctx):
    """The same kind of stream -- equal stored blocks beyond 4 GiB, a short final block -- through the HOST form
    zipc_hip_inflate with a CRC-32: the host forms keep the stream's descriptor and result in the context's
    staging slots, which the remainder's pass through the batch kernel must not reuse (it did: the CRC-32 then
    covered the wrong range of the output)."""
    import ctypes as C

    import torch

    from zipc_amd import _lib, batch, synth

    dev = torch.device("cuda", 0)
    n = (1 << 32) + 3 * 65534 + 22  # a multiple of 8 (the generator), a 26-byte final block
    src = synth.batch_bytes_torch(5, 7, 1, n, 8, dev)
    comp, J = _stored_stream(src)
    crc, _ = batch.checksum_device(gpu_ctx, src, want_adler32=False)
    plain = src.cpu().numpy()
    comp_h = comp.cpu().numpy()
    del src, comp
    torch.cuda.empty_cache()
    out = np.zeros(n + 64, np.uint8)
    out_len, ck = C.c_size_t(0), C.c_uint32(0)
    st = _lib.lib().zipc_hip_inflate(gpu_ctx.handle, comp_h.ctypes.data, comp_h.size, 1, n, 1, out.ctypes.data, n,
                                     C.byref(out_len), C.byref(ck))
    assert st == 0 and out_len.value == n
    assert ck.value == crc
    assert np.array_equal(out[:n], plain)
    # a limit one byte short: the reference's message, from the remainder's pass
    st = _lib.lib().zipc_hip_inflate(gpu_ctx.handle, comp_h.ctypes.data, comp_h.size, 1, n - 1, 1, out.ctypes.data, n,
                                     C.byref(out_len), C.byref(ck))
    assert st == 2


def test_one_long_stream_deflates_on_many_waves_bytes_equal_oracle(gpu_ctx, oracle):
    """ONE 40 MiB stream -- text, runs of one byte, short periods, symbols of several entropies, random bytes, in
    stretches of odd lengths -- through the forms a long stream takes (chain links by 128 Ki-position workgroups,
    parse by segments with the stitch and its runs of equal steps, ~650 blocks coded by a wave each): every byte
    of the output against the oracle at `Fast and `Default, zlib inflates it, and the GPU inflates it back."""
    import zlib

    import torch

    import util
    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    rnd = np.random.default_rng(404)
    parts, total, want = [], 0, 40 << 20
    k = 0
    while total < want:
        ln = int(rnd.integers(1, 700000))
        kind = k % 7
        if kind == 0:
            d = util.text(ln, k)
        elif kind == 1:
            d = bytes([int(rnd.integers(0, 256))]) * ln
        elif kind == 2:
            per = int(rnd.choice([2, 3, 7, 257, 258, 259, 1000, 4097]))
            pat = synth.stream_bytes_np(9, k, per, 8).tobytes()
            d = (pat * (ln // per + 1))[:ln]
        elif kind == 6:
            d = rnd.integers(0, 256, ln, dtype=np.uint8).tobytes()
        else:
            d = synth.stream_bytes_np(9, k, ln, (1, 2, 3, 4)[kind - 2 if kind - 2 < 4 else 3]).tobytes()
        parts.append(d)
        total += ln
        k += 1
    plain = b"".join(parts)[:want]
    n = len(plain)
    src = torch.from_numpy(np.frombuffer(plain, np.uint8).copy()).to(dev)
    cap = batch.deflate_bound(n)
    descs = batch.uniform_layout(1, n, cap)
    comp = torch.zeros(cap + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(16, dtype=torch.uint8, device=dev)
    for level in (1, 2):
        batch.deflate_batch(gpu_ctx, src, comp, d_descs, d_res, 1, n, n, level, 1)
        res = batch.results_from_device(d_res)
        assert int(res["status"][0]) == 0
        got = comp[:int(res["out_len"][0])].cpu().numpy().tobytes()
        st0, c0, k0 = oracle.deflate(plain, level=level, crc_op=1)
        assert len(got) == len(c0) and got == c0, level
        assert int(res["checksum"][0]) == k0 == zlib.crc32(plain)
        assert zlib.decompress(got, -15) == plain
    out = torch.zeros(n + 256, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(16, dtype=torch.uint8, device=dev)
    d_id = batch.to_device(batch.compact_descs(res, descs, n), dev)
    batch.inflate_batch(gpu_ctx, comp, out, d_id, d_ires, 1, n, 1)
    ires = batch.results_from_device(d_ires)
    assert int(ires["status"][0]) == 0 and int(ires["checksum"][0]) == k0 and torch.equal(out[:n], src)
    # ... by a wave per block (inflate.hip): ~650 blocks, most of them fixed ones that only the explorers find
    assert gpu_ctx.last_inflate_blocks() >= 500, gpu_ctx.last_inflate_blocks()
    # the same bytes as zlib codes them (other blocks, other matches), with the reference's Adler-32 this time
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    zraw = c.compress(plain) + c.flush()
    d_z = torch.cat([torch.from_numpy(np.frombuffer(zraw, np.uint8).copy()).to(dev), torch.zeros(256, dtype=torch.uint8, device=dev)])
    zdesc = batch.uniform_layout(1, len(zraw), n)
    out.fill_(0x5A)
    batch.inflate_batch(gpu_ctx, d_z, out, batch.to_device(zdesc, dev), d_ires, 1, n, 2)
    ires = batch.results_from_device(d_ires)
    assert int(ires["status"][0]) == 0 and int(ires["out_len"][0]) == n and torch.equal(out[:n], src)
    st0, d0, a0 = oracle.inflate(zraw, decompressed_size=n, crc_op=oracle.CRC_ADLER32)  # (block by block: not adler32(plain))
    assert st0 == 0 and int(ires["checksum"][0]) == a0 and bool((out[n:] == 0x5A).all())
    assert gpu_ctx.last_inflate_blocks() >= 500, gpu_ctx.last_inflate_blocks()


def test_real_text_at_16384_streams(gpu_ctx, oracle):
    """The reference's own documents (tests/golden/zip-docs.zip: APPNOTE.TXT, rfc1951.txt) as 16 384
    chunks of 64 KiB: long hash chains, long matches, long Huffman codes, many dynamic blocks per
    stream -- what the synthetic configs do not have.  Round trip of every byte, per-stream
    checksums, and every distinct chunk's compressed bytes against the oracle at `Fast, `Default and `Best."""
    import zipfile
    import zlib

    import torch

    import util
    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    import io

    z = zipfile.ZipFile(io.BytesIO(util.zip_docs()))
    app, rfc = z.read("zip-docs/APPNOTE.TXT"), z.read("zip-docs/rfc1951.txt")
    L, n = 65536, 16384
    pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L], (app[2 * L:] + rfc)[:L], rfc[1000:31000] + bytes(L - 30000)]
    assert all(len(p) == L for p in pieces)
    host = np.frombuffer(b"".join(pieces[i % len(pieces)] for i in range(n)), np.uint8).copy()
    src = torch.from_numpy(host).to(dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1])
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    for level in (1, 2, 3):  # 3 = `Best, the seam's default: up to 4096 candidates per position (about 300 on this text)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        batch.deflate_batch(gpu_ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
        res = batch.results_from_device(d_res)
        assert (res["status"] == 0).all()
        for j in list(range(len(pieces))) + [n - 1]:
            st, c0, crc0 = oracle.deflate(pieces[j % len(pieces)], level=level, crc_op=oracle.CRC_CRC32)
            o = int(descs["dst_off"][j])
            assert comp[o:o + int(res["out_len"][j])].cpu().numpy().tobytes() == c0, (level, j)
            assert int(res["checksum"][j]) == crc0 == zlib.crc32(pieces[j % len(pieces)])
        d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        out.fill_(0xA5)
        batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(batch.compact_descs(res, descs, L), dev), d_ires, n, L, 1)
        ires = batch.results_from_device(d_ires)
        assert (ires["status"] == 0).all() and (ires["out_len"] == L).all()
        assert torch.equal(out[:n * L], src)
        assert (ires["checksum"] == res["checksum"]).all()


def test_corpus_every_chunk_against_the_oracle(gpu_ctx, oracle):
    """tools/corpus.py: the reference's documents and this repository's own files (text, code, ELF and gfx950
    binaries, fixtures) as distinct 64 KiB streams -- every one of them against the oracle's bytes and CRC-32 at
    `Default and `Best, and back through inflate."""
    import os
    import sys
    import zlib

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import corpus
    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    L = 65536
    chunks = corpus.chunks(L)
    n = len(chunks)
    assert n >= 40
    src = torch.from_numpy(np.frombuffer(b"".join(chunks), np.uint8).copy()).to(dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1])
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    for level in (2, 3):
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        batch.deflate_batch(gpu_ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
        res = batch.results_from_device(d_res)
        assert (res["status"] == 0).all()
        host = comp.cpu().numpy()
        for j, c in enumerate(chunks):
            st, c0, crc0 = oracle.deflate(c, level=level, crc_op=oracle.CRC_CRC32)
            o = int(descs["dst_off"][j])
            assert host[o:o + int(res["out_len"][j])].tobytes() == c0, (level, j)
            assert int(res["checksum"][j]) == crc0 == zlib.crc32(c)
        d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        out.fill_(0x5A)
        batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(batch.compact_descs(res, descs, L), dev), d_ires, n, L, 1)
        ires = batch.results_from_device(d_ires)
        assert (ires["status"] == 0).all() and torch.equal(out[:n * L], src)
        assert (ires["checksum"] == res["checksum"]).all()


def test_one_stream_of_48_mib_inflates_by_blocks_with_following(gpu_ctx, oracle):
    """48 MiB of text and 4-bit symbols in ONE stream -- long enough for the form in which a block's wave writes down
    what its sources are copies of (api.hip: from 32 MiB of output on, output at least 1.5 x the input) -- as zlib
    codes it and as the reference's encoder does (most of its blocks fixed: explorers): the source's bytes, CRC-32."""
    import zlib

    import torch

    import util
    from zipc_amd import batch, synth

    dev = torch.device("cuda", 0)
    n = 48 << 20
    text = util.text(4 << 20, 12)
    parts = []
    for k in range(8):
        parts.append(text[(k * 37777) % (1 << 20):][:3 << 20])
        parts.append(synth.stream_bytes_np(2, 50 + k, 3 << 20, 4).tobytes())
    plain = b"".join(parts)[:n]
    assert len(plain) == n
    src = torch.from_numpy(np.frombuffer(plain, np.uint8).copy()).to(dev)
    out = torch.zeros(n + 256, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(16, dtype=torch.uint8, device=dev)
    # zlib's coding
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    zraw = c.compress(plain) + c.flush()
    assert n >= 1.5 * len(zraw)
    d_z = torch.cat([torch.from_numpy(np.frombuffer(zraw, np.uint8).copy()).to(dev), torch.zeros(256, dtype=torch.uint8, device=dev)])
    out.fill_(0x5A)
    batch.inflate_batch(gpu_ctx, d_z, out, batch.to_device(batch.uniform_layout(1, len(zraw), n), dev), d_ires, 1, n, 1)
    ires = batch.results_from_device(d_ires)
    assert int(ires["status"][0]) == 0 and int(ires["out_len"][0]) == n and torch.equal(out[:n], src)
    assert int(ires["checksum"][0]) == zlib.crc32(plain) and bool((out[n:] == 0x5A).all())
    assert gpu_ctx.last_inflate_blocks() >= 300, gpu_ctx.last_inflate_blocks()
    # the reference's encoder (this library's deflate: bytes checked against the oracle elsewhere)
    cap = batch.deflate_bound(n)
    descs = batch.uniform_layout(1, n, cap)
    comp = torch.zeros(cap + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(gpu_ctx, src, comp, batch.to_device(descs, dev), d_res, 1, n, n, 2, 1)
    res = batch.results_from_device(d_res)
    assert int(res["status"][0]) == 0 and int(res["checksum"][0]) == zlib.crc32(plain)
    out.fill_(0x5A)
    batch.inflate_batch(gpu_ctx, comp, out, batch.to_device(batch.compact_descs(res, descs, n), dev), d_ires, 1, n, 1)
    ires = batch.results_from_device(d_ires)
    assert int(ires["status"][0]) == 0 and int(ires["out_len"][0]) == n and torch.equal(out[:n], src)
    assert int(ires["checksum"][0]) == zlib.crc32(plain) and gpu_ctx.last_inflate_blocks() >= 700, gpu_ctx.last_inflate_blocks()
