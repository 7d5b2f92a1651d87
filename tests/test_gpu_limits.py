"""The corners of the encoder the synthetic configs never reach, on the GPU against the oracle:
Huffman codes that outgrow their length limit (Huffman.lengths_of_freqs' flatten-and-retry branch,
zd.ml:470-473 -- on the GPU a wave-parallel coder of its own, wave_lengths_of_freqs in deflate.hip),
matches at the very edge of the 32 KiB window (zd.ml:1143) next to the chain builder's sweep
boundaries (zd.ml:1187)."""
import zlib

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _deflate_batch(gpu_ctx, streams, level, crc_op=2):
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    n = len(streams)
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(len(s)) for s in streams]
    slots = [(c + 255) // 256 * 256 for c in caps]
    dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
    src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(gpu_ctx, src, dst, batch.to_device(descs, dev), d_res, n, max(len(s) for s in streams),
                        int(sum(len(s) for s in streams)), level, crc_op)
    res = batch.results_from_device(d_res)
    out = dst.cpu().numpy()
    return [(int(res["status"][i]), out[int(dst_off[i]):int(dst_off[i]) + int(res["out_len"][i])].tobytes(),
             int(res["checksum"][i])) for i in range(n)]


def test_huffman_length_limit_retry_reached_and_bytes_equal(gpu_ctx, oracle):
    cases = util.deflate_cases()
    want = {"fib_litlen": (False, True), "fib_codelen": (True, False), "fib_both": (True, True), "fib_multi": (True, True)}
    # more shapes of the same kind: other alphabet sizes and paddings, several seeds
    extra = {"fib_%d_%d_%d" % (k, pad, seed): util.fib_block(k, pad, seed)
             for k in (14, 16, 17, 18, 19) for pad in (2, 3, 5) for seed in (2, 3)}
    streams = {**{n: cases[n] for n in want}, **extra}
    names = list(streams)
    fired = [0, 0]
    for level in (1, 2, 3):
        got = _deflate_batch(gpu_ctx, [streams[n] for n in names], level)
        for name, (st, comp, adler) in zip(names, got):
            oracle.huffman_retries(reset=True)
            st0, c0, a0 = oracle.deflate(streams[name], level=level, crc_op=oracle.CRC_ADLER32)
            cl, ll = oracle.huffman_retries()
            if name in want:  # the input does reach the branch, for the code-length code and / or litlen
                assert (cl > 0, ll > 0) == want[name], (name, level, cl, ll)
            fired[0] += cl > 0
            fired[1] += ll > 0
            assert st == 0 and comp == c0 and adler == a0, (name, level)
            assert zlib.decompress(comp, -15) == streams[name]
    assert fired[0] >= 12 and fired[1] >= 12, fired


def test_matches_at_the_edge_of_the_window(gpu_ctx, oracle):
    from zipc_amd import zipc_deflate as Z

    for seed in (21, 22, 23):
        data = util.far_match_data(seed)
        for level, lv in (("fast", 1), ("default", 2), ("best", 3)):
            st0, c0, _ = oracle.deflate(data, level=lv)
            assert len(c0) < len(data) + 6 * 3  # some of the planted repeats were found (all-stored would be len + 15)
            assert Z.deflate(data, level=level).get_ok() == c0, (seed, level)
            assert Z.inflate(c0, decompressed_size=len(data)).get_ok() == data
    got = _deflate_batch(gpu_ctx, [util.far_match_data(s) for s in range(30, 40)], 2)
    for s, (st, comp, adler) in zip(range(30, 40), got):
        st0, c0, a0 = oracle.deflate(util.far_match_data(s), level=2, crc_op=oracle.CRC_ADLER32)
        assert st == 0 and comp == c0 and adler == a0, s


@pytest.mark.parametrize("shape", ["c2-16384x64KiB-4bit", "c4-1024x1MiB-3bit", "text-16384x64KiB", "ragged"])
def test_both_chain_kernels_make_the_same_links(gpu_ctx, shape):
    """insert_hash (zd.ml:1150-1152) two ways over WHOLE batches: lz_chain_xchg_kernel, whose order of equal hashes
    inside one LDS exchange is a property of the hardware that the context's probe vouches for, against lz_chain_kernel,
    which orders them itself -- every link of the benchmark's batch, of 1024 members of C4's shape, of the text batch
    and of a batch of ragged lengths (the fuzz runs and the full-size tests compare a few streams with the oracle; a
    rare reordering under load would only show in a stream nobody sampled)."""
    import torch

    from zipc_amd import batch, synth

    assert gpu_ctx.lds_exchange_ordered()
    dev = torch.device("cuda", 0)
    if shape.startswith("c2"):
        n, L = 16384, 65536
        src = synth.batch_bytes_torch(2, 0, n, L, 4, dev)
        lens = [L] * n
    elif shape.startswith("c4"):
        n, L = 1024, 1 << 20
        src = synth.batch_bytes_torch(4, 0, n, L, 3, dev)
        lens = [L] * n
    elif shape.startswith("text"):
        import io
        import zipfile

        z = zipfile.ZipFile(io.BytesIO(util.zip_docs()))
        app, rfc = z.read("zip-docs/APPNOTE.TXT"), z.read("zip-docs/rfc1951.txt")
        n, L = 16384, 65536
        pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
        src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()).to(dev)
        lens = [L] * n
    else:  # lengths from 0 to 300 000, text, symbols, runs and random bytes: the window's edge, short streams, streams of one tile and of many
        rng = np.random.default_rng(77)
        lens = [0, 1, 3, 4, 5, 63, 64, 65, 1023, 8192, 8193, 32767, 32768, 32769, 65535, 65536, 65537, 100000, 262144, 300000]
        lens += [int(x) for x in rng.integers(0, 200000, 400)]
        n = len(lens)
        parts = []
        for i, ln in enumerate(lens):
            kind = i % 4
            if kind == 0:
                parts.append(rng.integers(0, 1 << (1 + i % 8), ln, dtype=np.uint8).tobytes())
            elif kind == 1:
                parts.append((util.zip_docs() * (ln // 50000 + 1))[:ln])
            elif kind == 2:
                parts.append((bytes([i & 255]) * 700 + bytes(range(256)) * 3)[: max(1, min(ln, 1468))] * (ln // 1400 + 1))
                parts[-1] = parts[-1][:ln]
            else:
                parts.append(rng.integers(0, 256, ln, dtype=np.uint8).tobytes())
        src = torch.from_numpy(np.frombuffer(b"".join(parts) + b"\0" * 64, np.uint8).copy()).to(dev)
    off = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    descs = batch.make_descs(off, lens, np.zeros(n, np.uint64), [0] * n)
    d_descs = batch.to_device(descs, dev)
    total, longest = int(sum(lens)), int(max(lens))
    a, base_a = batch.debug_chain_links(gpu_ctx, src, d_descs, n, longest, total, 0)
    b, base_b = batch.debug_chain_links(gpu_ctx, src, d_descs, n, longest, total, 1)
    assert torch.equal(base_a, base_b)
    if not torch.equal(a, b):
        bad = torch.nonzero(a != b).flatten()
        assert False, "%d links differ, the first at slot %d (exchange %d, peel %d)" % (
            bad.numel(), int(bad[0]), int(a[bad[0]]) & 0xFFFF, int(b[bad[0]]) & 0xFFFF)
    assert int(torch.count_nonzero(a)) > total // 8 or shape == "ragged"  # (the links are there: most positions of these inputs have one)


def test_the_exchange_order_guards_have_run_and_agree():
    """lz_chain_xchg_kernel is exact only if one LDS exchange serves same-address lanes in lane order.  Round 6's guards:
    the context's create-time probe runs THE KERNEL (whole and by segments) on a stream of runs, short periods and
    few-symbol alphabets and compares every link with the ordering kernel's; the context's first deflate batch has its
    first streams chained by both kernels under that batch's load.  A fresh context: both have run, nothing differed, and
    the batch's bytes are the oracle's."""
    import oracle
    import torch
    import zipc_amd
    from zipc_amd import batch, synth

    ctx = zipc_amd.Context(0)
    assert ctx.lds_exchange_ordered()
    probed, bad = ctx.chain_check()
    assert probed > 2 * 200000 and bad == 0  # the probe's stream twice: whole and by segments
    dev = torch.device("cuda", 0)
    n, L = 4096, 65536
    src = synth.batch_bytes_torch(2, 0, n, L, 4, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    comp = torch.zeros(n * int(descs["dst_off"][1]) + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    for rep in range(2):  # (the second batch is not checked again: once per context)
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 1)
        res = batch.results_from_device(d_res)
        assert (res["status"] == 0).all()
        after, bad = ctx.chain_check()
        assert bad == 0 and after == probed + 32 * (L - 3) and ctx.lds_exchange_ordered()
    for j in (0, 31, 32, 4095):
        st, c0, crc0 = oracle.deflate(synth.stream_bytes_np(2, j, L, 4).tobytes(), level=2, crc_op=oracle.CRC_CRC32)
        o = int(descs["dst_off"][j])
        assert comp[o:o + int(res["out_len"][j])].cpu().numpy().tobytes() == c0 and int(res["checksum"][j]) == crc0
