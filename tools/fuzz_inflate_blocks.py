#!/usr/bin/env python3
"""Randomized parity of the one-stream inflate (a wave per block, inflate.hip) against the oracle: sources of mixed
content, the reference's encoder and zlib with random levels, memory levels, strategies and flushes, some streams
damaged or cut or given too small a limit.  TRIALS (default 200), SEED; SIZES=a,b,c: source sizes to draw from.  BATCH=k: the same streams k to a call of the
batch form (zipc_hip_inflate_batch: the call's long streams go by blocks side by side, the others by their one waves),
with short streams mixed in."""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle, zipc_amd
from zipc_amd import zipc_deflate as Z

text = b"".join(open(os.path.join(ROOT, f), "rb").read() for f in ("SURVEY.md", "DESIGN.md", "BASELINE.md"))
lib = open(os.path.join(ROOT, "oracle", "libzd_oracle.so"), "rb").read()


def run(trials, seed, sizes=(120000, 300000, 700000, 1500000, 3000000), verbose=True):
    """-> (trials that went by blocks, mismatches)"""
    rnd = random.Random(seed); rng = np.random.default_rng(seed)

    def piece(n):
        k = rnd.randrange(6)
        if k == 0: o = rnd.randrange(len(text)); return (text[o:] + text)[:n] if n <= len(text) else (text * (n // len(text) + 1))[:n]
        if k == 1: return (rng.integers(0, 1 << rnd.choice((2, 3, 4, 6)), n, dtype=np.uint8)).tobytes()
        if k == 2: return (rng.integers(0, 1 << 14, n // 4 + 1, dtype=np.uint32) * np.uint32(0x10001)).tobytes()[:n]
        if k == 3: return bytes([rnd.randrange(256)]) * n if rnd.random() < 0.5 else bytes(rnd.randrange(256) for _ in range(rnd.randrange(2, 9))) * (n // 2 + 1)
        if k == 4: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        o = rnd.randrange(max(1, len(lib) - n)); return (lib[o:o + n] + lib)[:n]

    def source():
        n = rnd.choice(sizes) + rnd.randrange(50000)
        out = b""
        while len(out) < n: out += piece(min(n - len(out), rnd.choice((3000, 40000, 200000, 1 << 20))))[:n - len(out)]
        return out[:n]

    def encode(data):
        if rnd.random() < 0.3:
            lv = rnd.choice((1, 2, 3)); return "oracle-%d" % lv, oracle.deflate(data, level=lv)[1]
        lv, ml, stg = rnd.randrange(0, 10), rnd.randrange(4, 10), rnd.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED))
        c = zlib.compressobj(lv, zlib.DEFLATED, -15, ml, stg); out = b""; i = 0
        while i < len(data):
            step = rnd.choice((5000, 70000, 400000, len(data))); out += c.compress(data[i:i + step]); i += step
            if rnd.random() < 0.3: out += c.flush(rnd.choice((zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH)))
        return "zlib-%d/%d/%d" % (lv, ml, stg), out + c.flush()

    ctx = zipc_amd.default_context(0)
    by_blocks = bad = 0
    for t in range(trials):
        data = source(); enc, raw = encode(data); lim = len(data); what = "whole"
        r = rnd.random()
        if r < 0.12: at = rnd.randrange(len(raw)); raw = raw[:at] + bytes([raw[at] ^ (1 << rnd.randrange(8))]) + raw[at + 1:]; what = "bit flipped at %d" % at
        elif r < 0.2: raw = raw[:rnd.randrange(len(raw) // 2, len(raw))]; what = "cut"
        elif r < 0.28: lim = rnd.choice((len(data) - 1, len(data) // 2, len(data) + 5)); what = "limit %d" % lim
        if len(raw) == 0: continue
        adler = rnd.random() < 0.3  # (Adler-32: per block with the reference's chunking)
        st0, d0, c0 = oracle.inflate(raw, decompressed_size=lim, crc_op=oracle.CRC_ADLER32 if adler else oracle.CRC_CRC32)
        g = Z.inflate_and_adler_32(raw, decompressed_size=lim) if adler else Z.inflate_and_crc_32(raw, decompressed_size=lim)
        nb = ctx.last_inflate_blocks(); by_blocks += nb > 0
        ok = g.is_ok() == (st0 == 0) and ((g.get_ok() == (d0, c0)) if st0 == 0 else g.error == oracle.MESSAGES[st0])
        if not ok:
            bad += 1
            if verbose:
                print("MISMATCH trial %d: %d B, %s, %s, %s, %d B of input, blocks %d, oracle status %d" % (t, len(data), enc, what, "adler" if adler else "crc", len(raw), nb, st0), flush=True)
    return by_blocks, bad


def run_batches(calls, seed, k, sizes=(120000, 300000, 700000, 1100000, 1500000, 3000000), verbose=True):
    """-> (blocks that went by blocks, mismatched streams)"""
    import torch
    from zipc_amd import batch
    rnd = random.Random(seed); rng = np.random.default_rng(seed)
    ctx = zipc_amd.default_context(0); dev = torch.device("cuda:0")
    pieces = [text, lib, rng.integers(0, 16, 1 << 20, dtype=np.uint8).tobytes(),
              (rng.integers(0, 1 << 14, 1 << 18, dtype=np.uint32) * np.uint32(0x10001)).tobytes(), rng.integers(0, 256, 1 << 18, dtype=np.uint8).tobytes()]

    def source():
        n = rnd.choice(sizes + (500, 20000)) + rnd.randrange(50000)
        out = b""
        while len(out) < n:
            pc = rnd.choice(pieces); o = rnd.randrange(len(pc)); out += pc[o:o + rnd.choice((3000, 40000, 200000, 1 << 20))]
        return out[:n]

    def encode(data):
        if rnd.random() < 0.3: return oracle.deflate(data, level=rnd.choice((1, 2, 3)))[1]
        c = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, -15, rnd.randrange(4, 10), rnd.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED)))
        return c.compress(data) + c.flush()

    blocks = bad = 0
    for call in range(calls):
        n = rnd.randrange(2, k + 1); streams = []; lims = []; caps = []
        for i in range(n):
            data = source(); raw = encode(data); lim = len(data); r = rnd.random()
            if r < 0.1: at = rnd.randrange(len(raw)); raw = raw[:at] + bytes([raw[at] ^ (1 << rnd.randrange(8))]) + raw[at + 1:]
            elif r < 0.15: raw = raw[:rnd.randrange(len(raw) // 2, len(raw))]
            elif r < 0.22: lim = rnd.choice((len(data) - 1, len(data) // 2, len(data) + 5))
            streams.append(raw); lims.append(lim); caps.append(max(lim, 8) + rnd.choice((0, 0, 7, 4096)))
        crc_op = rnd.choice((oracle.CRC_CRC32, oracle.CRC_CRC32, oracle.CRC_ADLER32, oracle.CRC_NOP))
        src_off = np.cumsum([0] + [(len(s) + 3) & ~3 for s in streams]); dst_off = np.cumsum([0] + [(c + 255) & ~255 for c in caps])
        arena = np.zeros(int(src_off[-1]) + 64, np.uint8)
        for s, o in zip(streams, src_off): arena[o:o + len(s)] = np.frombuffer(s, np.uint8)
        descs = batch.make_descs(src_off[:-1], [len(s) for s in streams], dst_off[:-1], caps, limit=lims)
        src = torch.from_numpy(arena).to(dev); dst = torch.zeros(int(dst_off[-1]) + 64, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(16 * n, dtype=torch.uint8, device=dev)
        batch.inflate_batch(ctx, src, dst, batch.to_device(descs, dev), d_res, n, max(caps), crc_op)
        blocks += ctx.last_inflate_blocks()
        res = batch.results_from_device(d_res); out = dst.cpu().numpy()
        for i in range(n):
            st0, d0, c0 = oracle.inflate(streams[i], decompressed_size=lims[i], crc_op=crc_op)
            ok = int(res["status"][i]) == st0
            if ok and st0 == 0:
                ok = int(res["out_len"][i]) == len(d0) and out[dst_off[i]:dst_off[i] + len(d0)].tobytes() == d0 and (crc_op == oracle.CRC_NOP or int(res["checksum"][i]) == c0)
            if not ok:
                bad += 1
                if verbose: print("MISMATCH call %d stream %d of %d: %d B in, limit %d, crc_op %d, oracle status %d, got %s" % (call, i, n, len(streams[i]), lims[i], crc_op, st0, res[i]), flush=True)
    return blocks, bad


if __name__ == "__main__":
    TRIALS = int(os.environ.get("TRIALS", "200")); SEED = int(os.environ.get("SEED", "1"))
    if os.environ.get("BATCH"):
        blocks, bad = run_batches(TRIALS, SEED, int(os.environ["BATCH"]))
        print("fuzz_inflate_blocks: %d calls of up to %s streams (seed %d), %d blocks went by blocks, %d mismatches" % (TRIALS, os.environ["BATCH"], SEED, blocks, bad))
        sys.exit(1 if bad else 0)
    sizes = tuple(int(x) for x in os.environ["SIZES"].split(",")) if os.environ.get("SIZES") else None
    by_blocks, bad = run(TRIALS, SEED, sizes) if sizes else run(TRIALS, SEED)
    print("fuzz_inflate_blocks: %d trials (seed %d), %d went by blocks, %d mismatches" % (TRIALS, SEED, by_blocks, bad))
    sys.exit(1 if bad else 0)
