#!/usr/bin/env python3
"""Randomized parity of the many-wave deflate forms (parse by segments, blocks by a wave each, chain links by
segments) on long streams of mixed content: stretches of text, symbols of several entropies, random bytes, runs of one
byte and short periods, at odd lengths, a few streams per call -- bytes and checksums against the oracle, zlib inflates.
Usage: fuzz_long.py [seed0] [n_seeds]   (ZIPC_HIP_PARSE_SEG picks the segment size: run it with several)."""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle, util, zipc_amd
from zipc_amd import batch, synth

dev = torch.device("cuda", 0)
ctx = zipc_amd.Context(0)


def stretch(r, k):
    ln = r.choice([1, 5, 63, 64, 65, 257, 258, 259, 4095, 4096, 4097]) if r.random() < 0.2 else r.randrange(1, 300000)
    kind = r.randrange(8)
    if kind == 0:
        return util.text(ln, k)
    if kind == 1:
        return bytes([r.randrange(256)]) * ln
    if kind == 2:
        per = r.choice([2, 3, 4, 7, 64, 255, 256, 257, 258, 259, 260, 516, 1000, 4096, 4097, 30000])
        pat = bytes(r.randrange(256) for _ in range(per))
        return (pat * (ln // per + 1))[:ln]
    if kind == 3:
        return bytes(r.getrandbits(8) for _ in range(min(ln, 20000)))
    if kind == 4:  # a period with a defect now and then
        per = r.choice([3, 258, 259, 1000])
        b = bytearray((bytes(r.randrange(256) for _ in range(per)) * (ln // per + 1))[:ln])
        for _ in range(ln // 5000):
            b[r.randrange(ln)] ^= 1
        return bytes(b)
    return synth.stream_bytes_np(9, k, ln, (1, 2, 3, 4)[kind - 5 if kind - 5 < 4 else 3]).tobytes()


def gen(r, k):
    want = r.choice([33000, 65534, 65535, 131072, 200000]) if r.random() < 0.3 else r.randrange(33000, 3000000)
    parts, total = [], 0
    while total < want:
        d = stretch(r, k + len(parts))
        parts.append(d)
        total += len(d)
    return b"".join(parts)[:want]


def run_seed(seed):
    r = random.Random(seed)
    n = r.randrange(1, 6)
    plains = [gen(r, seed * 100 + i * 10) for i in range(n)]
    level = r.randrange(1, 4)
    crc_op = r.choice([1, 2])
    src_off = np.cumsum([0] + [(len(p) + 255) // 256 * 256 for p in plains[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(len(p)) for p in plains]
    dst_off = np.cumsum([0] + [(c + 255) // 256 * 256 for c in caps[:-1]]).astype(np.uint64)
    arena = bytearray(int(src_off[-1]) + len(plains[-1]) + 64)
    for o, p in zip(src_off, plains):
        arena[int(o):int(o) + len(p)] = p
    src = torch.from_numpy(np.frombuffer(bytes(arena), np.uint8).copy()).to(dev)
    dst = torch.zeros(int(dst_off[-1]) + caps[-1] + 256, dtype=torch.uint8, device=dev)
    descs = batch.make_descs(src_off, [len(p) for p in plains], dst_off, caps)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, dst, batch.to_device(descs, dev), d_res, n, max(len(p) for p in plains),
                        sum(len(p) for p in plains), level, crc_op)
    res, out = batch.results_from_device(d_res), dst.cpu().numpy()
    bad = 0
    for i, p in enumerate(plains):
        st0, c0, k0 = oracle.deflate(p, level=level, crc_op=crc_op)
        got = out[int(dst_off[i]):int(dst_off[i]) + int(res["out_len"][i])].tobytes()
        if res["status"][i] != 0 or got != c0 or res["checksum"][i] != k0 or zlib.decompress(got, -15) != p:
            bad += 1
            print("MISMATCH seed", seed, "stream", i, "level", level, "len", len(p), "out", len(got), len(c0), flush=True)
    return n, bad


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    total = bad = 0
    for seed in range(seed0, seed0 + n_seeds):
        t, b = run_seed(seed)
        total += t
        bad += b
    print("FUZZ", "ok" if bad == 0 else "FAILED", total, "streams", "seg", os.environ.get("ZIPC_HIP_PARSE_SEG", "default"), flush=True)
    sys.exit(1 if bad else 0)


main()
