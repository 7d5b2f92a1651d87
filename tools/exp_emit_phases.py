#!/usr/bin/env python3
"""Phase times inside deflate_emit_kernel, from a timing-only build of the library
(-DZD_EMIT_PHASES: the per-stream results carry s_memtime deltas, shader-clock ticks, instead of lengths).
ZIPC_HIP_LIB must point at that build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch, synth
n = int(os.environ.get("N_STREAMS", "16384")); L = 65536; bits = int(os.environ.get("BITS", "4"))
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
src = synth.batch_bytes_torch(2, 0, n, L, bits, dev)
descs = batch.uniform_layout(n, L, batch.deflate_bound(L))
comp = torch.zeros(n * int(descs["dst_off"][1]) + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
for _ in range(2): batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 0)
res = batch.results_from_device(d_res)
tot = res["checksum"].astype(np.float64) * 16
hist = (res["out_len"] & 0xFFFFFFFF).astype(np.float64) * 16
code = (res["out_len"] >> 32).astype(np.float64) * 16
MHZ = 2381.0  # s_memtime is the shader clock; calibrated against s_memrealtime in tools/exp_match_phases.py
us = lambda t: t / MHZ
print("per stream, us (s_memtime at %.0f MHz): total %.1f  histogram %.1f  codes %.1f  pack+rest %.1f   (streams %d)"
      % (MHZ, us(tot.mean()), us(hist.mean()), us(code.mean()), us((tot - hist - code).mean()), n))
print("slowest / mean stream: %.2f ; p99 / mean: %.2f" % (tot.max() / tot.mean(), np.percentile(tot, 99) / tot.mean()))
