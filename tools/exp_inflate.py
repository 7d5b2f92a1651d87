#!/usr/bin/env python3
"""Experiment driver: times the inflate / deflate kernels on the C2 workload with
whatever libzipc_hip.so ZIPC_HIP_LIB points at (per-kernel HIP-event times)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zipc_amd
from zipc_amd import batch, synth

def main():
    n = int(os.environ.get("N_STREAMS", "16384")); L = 65536; bits = int(os.environ.get("BITS", "4"))
    level = int(os.environ.get("LEVEL", "2"))
    reps = int(os.environ.get("REPS", "3"))
    crc = int(os.environ.get("CRC_OP", "0"))  # 0 none, 1 CRC-32, 2 Adler-32 (fused per-stream checksums)
    dev = torch.device("cuda", 0)
    ctx = zipc_amd.Context(0)
    if os.environ.get("TEXT"):  # 64 KiB chunks of the reference's zip-docs texts instead of synthetic symbols
        import zipfile
        import numpy as np
        z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
        app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
        pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
        src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()).to(dev)
    else:
        src = synth.batch_bytes_torch(2, 0, n, L, bits, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1])
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc)
    res = batch.results_from_device(d_res)
    idescs = batch.compact_descs(res, descs, L)
    d_idescs = batch.to_device(idescs, dev)
    batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, crc)
    ok = bool(torch.equal(out[:n * L], src))
    if crc:  # the checksum of the source (deflate side) must equal the one of the output (inflate side)
        ires = batch.results_from_device(d_ires)
        ok = ok and bool((ires["checksum"] == res["checksum"]).all()) and bool((ires["status"] == 0).all())
    ctx.set_profiling(True); ctx.reset_kernel_times()
    for _ in range(reps):
        if os.environ.get("DEFLATE", "1") == "1":
            batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc)
        batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, crc)
    t = ctx.kernel_times()
    print(json.dumps({"lib": os.environ.get("ZIPC_HIP_LIB", "default"), "roundtrip_ok": ok,
                      "ratio": float(res["out_len"].sum()) / (n * L),
                      "ms": {k: round(v[1] / v[0], 3) for k, v in t.items()}}))

main()
