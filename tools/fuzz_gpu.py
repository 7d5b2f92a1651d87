#!/usr/bin/env python3
"""Longer randomized parity run than the test-suite carries: inflate and deflate of
assorted streams on the GPU against the oracle.  Usage: fuzz_gpu.py [seed0] [n_seeds]."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle, util, zipc_amd
from zipc_amd import batch

dev = torch.device("cuda", 0)
ctx = None  # made on first use (tests/test_gpu_fuzz.py imports this file for run_seed)


def _ctx():
    global ctx
    if ctx is None:
        ctx = zipc_amd.Context(0)
    return ctx


def gen_plain(r):
    kind = r.randrange(7)
    n = r.choice([0, 1, 2, 3, 4, 5, 17, 100, 1000, 5000, 65533, 65534, 65535, 65536, 70000, 140000]) if r.random() < 0.4 \
        else r.randrange(0, 60000)
    if kind == 0:
        return bytes(r.randrange(1 << r.randrange(1, 9)) for _ in range(n))
    if kind == 1:
        return bytes([r.randrange(256)]) * n
    if kind == 2:
        pat = bytes(r.randrange(256) for _ in range(r.randrange(1, 40)))
        return (pat * (n // len(pat) + 1))[:n]
    if kind == 3:
        words = [bytes(r.randrange(97, 123) for _ in range(r.randrange(2, 9))) for _ in range(r.randrange(2, 200))]
        out = bytearray()
        while len(out) < n:
            out += r.choice(words) + b" "
        return bytes(out[:n])
    if kind == 4:
        b = bytearray(r.randrange(4) for _ in range(n))
        for _ in range(n // 50):
            if n > 300:
                i, j, l = r.randrange(n - 258), r.randrange(n - 258), r.randrange(3, 258)
                b[i:i + l] = b[j:j + l]
        return bytes(b)
    if kind == 5:
        return bytes((i * i >> 3) & 0xFF for i in range(n))
    return bytes(r.getrandbits(8) if r.random() < 0.1 else 65 for _ in range(n))


def run_inflate(streams, cap, crc_op):
    n = len(streams)
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    dst_off = np.arange(n, dtype=np.uint64) * (cap + 256)
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, [cap] * n, limit=[cap] * n)
    src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(n * (cap + 256) + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(_ctx(), src, dst, batch.to_device(descs, dev), d_res, n, cap, crc_op)
    return batch.results_from_device(d_res), dst.cpu().numpy(), dst_off


def run_seed(seed, n_plain=120, n_headers=(100, 300), say=print):
    """one seed's batch: (checks, mismatches)"""
    total = bad = 0
    if True:
        r = random.Random(seed)
        plains = [gen_plain(r) for _ in range(n_plain)]
        # deflate on the GPU at a random level per batch, bytes against the oracle
        level = r.randrange(4)
        cap = batch.deflate_bound(max(len(p) for p in plains))
        n = len(plains)
        src_off = np.cumsum([0] + [len(p) for p in plains[:-1]]).astype(np.uint64)
        dst_off = np.arange(n, dtype=np.uint64) * (cap + 256)
        descs = batch.make_descs(src_off, [len(p) for p in plains], dst_off, [cap] * n)
        src = torch.from_numpy(np.frombuffer(b"".join(plains) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
        dst = torch.zeros(n * (cap + 256) + 256, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        batch.deflate_batch(_ctx(), src, dst, batch.to_device(descs, dev), d_res, n, max(len(p) for p in plains),
                            sum(len(p) for p in plains), level, 2)
        res, out = batch.results_from_device(d_res), dst.cpu().numpy()
        comps = []
        for i, p in enumerate(plains):
            st0, c0, a0 = oracle.deflate(p, level=level, crc_op=2)
            got = out[int(dst_off[i]):int(dst_off[i]) + int(res["out_len"][i])].tobytes()
            total += 1
            if res["status"][i] != 0 or got != c0 or res["checksum"][i] != a0:
                bad += 1
                say("DEFLATE MISMATCH seed", seed, "stream", i, "level", level, len(p))
            comps.append(c0)
        # inflate: the valid streams, damaged copies, random headers
        streams = list(comps)
        for i, c in enumerate(comps):
            streams += util.corrupt_variants(c, seed * 1000 + i, 2)
        streams += util.header_fuzz_streams(seed, n_headers[0], n_headers[1])
        capi = max(max(len(p) for p in plains), 1 << 16) + 64
        for crc_op in (r.choice([0, 1]), 2):
            res, out, doff = run_inflate(streams, capi, crc_op)
            for i, s in enumerate(streams):
                st0, d0, c0 = oracle.inflate(s, decompressed_size=capi, crc_op=crc_op)
                total += 1
                ok = res["status"][i] == st0
                if ok and st0 == 0:
                    o = int(doff[i])
                    ok = res["out_len"][i] == len(d0) and out[o:o + len(d0)].tobytes() == d0 and res["checksum"][i] == c0
                if not ok:
                    bad += 1
                    say("INFLATE MISMATCH seed", seed, "stream", i, "crc_op", crc_op, st0, int(res["status"][i]))
    return total, bad


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    total = bad = 0
    for seed in range(seed0, seed0 + n_seeds):
        t, b = run_seed(seed)
        total += t
        bad += b
        print("seed", seed, "done; checks so far", total, "mismatches", bad, flush=True)
    print("FUZZ", "FAILED" if bad else "ok", total, "checks")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
