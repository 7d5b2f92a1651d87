#!/usr/bin/env python3
"""Randomized parity of the one-stream inflate (a wave per block, inflate.hip) against the oracle: sources of mixed
content, the reference's encoder and zlib with random levels, memory levels, strategies and flushes, some streams
damaged or cut or given too small a limit.  TRIALS (default 200), SEED."""
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


if __name__ == "__main__":
    TRIALS = int(os.environ.get("TRIALS", "200")); SEED = int(os.environ.get("SEED", "1"))
    by_blocks, bad = run(TRIALS, SEED)
    print("fuzz_inflate_blocks: %d trials (seed %d), %d went by blocks, %d mismatches" % (TRIALS, SEED, by_blocks, bad))
    sys.exit(1 if bad else 0)
