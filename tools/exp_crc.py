#!/usr/bin/env python3
"""times the checksum kernels on one big buffer"""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zipc_amd
from zipc_amd import batch, synth
n = int(os.environ.get("N", str(1 << 30)))
dev = torch.device("cuda", 0)
ctx = zipc_amd.Context(0)
buf = synth.batch_bytes_torch(3, 0, 1, n, 8, dev)
crc, adler = batch.checksum_device(ctx, buf)
ctx.set_profiling(True); ctx.reset_kernel_times()
for _ in range(3): batch.checksum_device(ctx, buf)
t = ctx.kernel_times()
ok = crc == zlib.crc32(buf[: 1 << 26].cpu().numpy()) if n == (1 << 26) else None
print(json.dumps({"n": n, "crc": hex(crc), "adler": hex(adler), "ms": {k: round(v[1] / v[0], 3) for k, v in t.items()}}))
