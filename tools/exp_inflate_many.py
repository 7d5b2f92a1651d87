#!/usr/bin/env python3
"""A call of N long streams (default 64 x 1 MiB of text, zlib -6 and this library's own encoder): the batch form with
its streams by blocks side by side (inflate.hip) against their one waves (ZIPC_HIP_INFLATE_BLOCKS=0 in a second
process); checked against the sources, wall time per call.  N, LEN, REPS; KIND, ENC pick one case."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch

dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
N = int(os.environ.get("N", "64")); L = int(os.environ.get("LEN", str(1 << 20))); REPS = int(os.environ.get("REPS", "7"))
text = b"".join(open(os.path.join(ROOT, f), "rb").read() for f in ("SURVEY.md", "DESIGN.md", "BASELINE.md"))
rng = np.random.default_rng(3)


def source(i, kind):
    if kind.startswith("xx"):  # text with one stretch of it said twice in a row, xx<bytes>: matches of 258 at one distance
        n = int(kind[2:]); base = ((text[i * 7919 % 1000:] + text) * (L // len(text) + 2))
        return (base[:300000] + base[300000 - n:300000] + base[300000:])[:L]
    if kind == "text":
        o = (i * 7919) % len(text); return ((text[o:] + text) * (L // len(text) + 2))[:L]
    return (rng.integers(0, 1 << 14, L // 4, dtype=np.uint32) * np.uint32(0x10001)).tobytes()


def raw_zlib(data, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15); return c.compress(data) + c.flush()


KINDS = os.environ.get("KIND", "text,binary").split(","); ENCS = os.environ.get("ENC", "zlib-6,own").split(",")
for kind in KINDS:
    plain = [source(i + int(os.environ.get("FIRST", "0")), kind) for i in range(N)]
    for enc in ENCS:
        if enc == "own":
            # (one call of the batch form for all of them)
            src = torch.from_numpy(np.frombuffer(b"".join(plain), np.uint8).copy()).to(dev)
            cap = batch.deflate_bound(L); descs = batch.uniform_layout(N, L, cap)
            comp = torch.zeros(int(descs["dst_off"][-1]) + cap + 256, dtype=torch.uint8, device=dev); d_res = torch.zeros(16 * N, dtype=torch.uint8, device=dev)
            batch.deflate_batch(ctx, src, comp, batch.to_device(descs, dev), d_res, N, L, N * L, 2, 0)
            r = batch.results_from_device(d_res); hc = comp.cpu().numpy()
            streams = [hc[int(descs["dst_off"][i]):int(descs["dst_off"][i]) + int(r["out_len"][i])].tobytes() for i in range(N)]
        else:
            streams = [raw_zlib(p) for p in plain]
        src_off = np.cumsum([0] + [(len(s) + 255) & ~255 for s in streams])
        arena = np.zeros(int(src_off[-1]) + 256, np.uint8)
        for s, o in zip(streams, src_off): arena[o:o + len(s)] = np.frombuffer(s, np.uint8)
        descs = batch.make_descs(src_off[:-1], [len(s) for s in streams], np.arange(N) * L, [L] * N, limit=[L] * N)
        d_src = torch.from_numpy(arena).to(dev); d_descs = batch.to_device(descs, dev)
        out = torch.zeros(N * L + 256, dtype=torch.uint8, device=dev); d_res = torch.zeros(16 * N, dtype=torch.uint8, device=dev)
        want = torch.from_numpy(np.frombuffer(b"".join(plain), np.uint8).copy()).to(dev)
        for crc_op in (0, 1):
            ts = []
            for rep in range(REPS):
                out.zero_(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                batch.inflate_batch(ctx, d_src, out, d_descs, d_res, N, L, crc_op, sync=False); ctx.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            r = batch.results_from_device(d_res)
            ok = bool((r["status"] == 0).all()) and bool((r["out_len"] == L).all()) and torch.equal(out[:N * L], want)
            if crc_op == 1: ok = ok and all(int(r["checksum"][i]) == (zlib.crc32(plain[i]) & 0xFFFFFFFF) for i in range(N))
            print("%-7s %-7s %d x %d B, crc_op %d: %s  blocks %d  ms min %.2f median %.2f  (%.1f GiB/s of output)" % (
                kind, enc, N, L, crc_op, "ok" if ok else "MISMATCH", ctx.last_inflate_blocks(), min(ts), sorted(ts)[len(ts) // 2],
                N * L / 2**30 / (min(ts) / 1e3)), flush=True)
