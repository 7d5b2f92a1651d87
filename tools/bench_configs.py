#!/usr/bin/env python3
"""Device-resident timings of the other BASELINE.json configs (C3, C4, C5) -- one
JSON line each.  bench.py stays the headline (C2); these feed DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zipc_amd
from zipc_amd import batch, synth

GIB = float(1 << 30)
dev = torch.device("cuda", 0)
ctx = zipc_amd.Context(0)

def timed(fn, reps=3):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps

def codec(cfg, n, L, bits, level, label):
    src = synth.batch_bytes_torch(cfg, 0, n, L, bits, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1])
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
    res = batch.results_from_device(d_res)
    idescs = batch.compact_descs(res, descs, L)
    d_idescs = batch.to_device(idescs, dev)
    batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1)
    ok = bool(torch.equal(out[:n * L], src))
    td = timed(lambda: batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1, sync=False))
    ti = timed(lambda: batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1, sync=False))
    ctx.set_profiling(True); ctx.reset_kernel_times()
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
    batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1)
    k = {a: round(b[1] / b[0], 3) for a, b in ctx.kernel_times().items()}
    ctx.set_profiling(False)
    N, C = n * L, int(res["out_len"].sum())
    print(json.dumps({"config": label, "streams": n, "stream_len": L, "ratio": C / N, "round_trip_ok": ok,
                      "deflate_gib_s": N / GIB / td, "inflate_gib_s": N / GIB / ti,
                      "inflate_hbm_gb_s": (N + C) / ti / 1e9, "kernels_ms": k}))

which = sys.argv[1:] or ["c3", "c4", "c5"]
if "c3" in which:
    n = 4 << 30
    buf = synth.batch_bytes_torch(3, 0, 1, n, 8, dev)
    t = timed(lambda: batch.checksum_device(ctx, buf))
    ctx.set_profiling(True); ctx.reset_kernel_times(); batch.checksum_device(ctx, buf)
    k = {a: round(b[1] / b[0], 3) for a, b in ctx.kernel_times().items()}; ctx.set_profiling(False)
    print(json.dumps({"config": "C3: CRC-32 + Adler-32 of one 4 GiB random buffer (one pass over the bytes: crc32_adler_segments)",
                      "gib_s": n / GIB / t, "hbm_gb_s": n / t / 1e9, "kernels_ms": k}))
    del buf
if "c4" in which:
    codec(4, 8192, 1 << 20, 3, 2, "C4: 8192 members x 1 MiB of 3-bit symbols, level default (1 GPU)")
if "c5" in which:
    codec(5, 131072, 65536, 8, 2, "C5: 131072 streams x 64 KiB uniform random (stored 65534 + fixed 2), inflate roofline run")
if "c5single" in which or not sys.argv[1:]:
    # C5's secondary form: the same 8 GiB as ONE stream of stored blocks (tests/test_gpu_fullsize.py builds it the same way)
    import numpy as np
    n, block = 8 << 30, 65534
    src = synth.batch_bytes_torch(5, 0, 1, n, 8, dev)
    J = n // block; rest = n - J * block
    comp = torch.empty(n + 5 * (J + 1), dtype=torch.uint8, device=dev)
    body = comp[:J * (block + 5)].view(J, block + 5)
    for col, val in enumerate((0, block & 255, block >> 8, (~block) & 255, ((~block) >> 8) & 255)): body[:, col] = val
    body[:, 5:] = src[:J * block].view(J, block)
    tail = comp[J * (block + 5):]
    for col, val in enumerate((1, rest & 255, rest >> 8, (~rest) & 255, ((~rest) >> 8) & 255)): tail[col] = val
    tail[5:] = src[J * block:]
    out = torch.zeros(n + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(batch.make_descs([0], [comp.numel()], [0], [n], limit=[n]), dev)
    d_res = torch.zeros(16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(ctx, comp, out, d_descs, d_res, 1, n, 0)
    ok = bool(torch.equal(out[:n], src)) and int(batch.results_from_device(d_res)["status"][0]) == 0
    ti = timed(lambda: batch.inflate_batch(ctx, comp, out, d_descs, d_res, 1, n, 0, sync=False))
    ctx.set_profiling(True); ctx.reset_kernel_times(); batch.inflate_batch(ctx, comp, out, d_descs, d_res, 1, n, 0)
    k = {a: round(b[1] / b[0], 3) for a, b in ctx.kernel_times().items()}; ctx.set_profiling(False)
    print(json.dumps({"config": "C5 as ONE stream: 8 GiB in %d stored blocks of 65534 bytes + a short final one, inflate (no checksum)" % J,
                      "round_trip_ok": ok, "inflate_gib_s": n / GIB / ti, "inflate_hbm_gb_s": (n + comp.numel()) / ti / 1e9, "kernels_ms": k}))
