#!/usr/bin/env python3
"""Inflate of hand-made worst cases, one stream alone: thousands of 3-byte matches each copying the one
before it (a dependence chain as long as the tile has holes), with and without a period in the bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle, util, zipc_amd
from zipc_amd import batch
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
for name, (comp, plain) in (("30000 x (3, 3): a period in the bits", util.fixed_block_of_short_matches(30000, 3, 3)),
                            ("30000 x (3..4, 1..4) at random", util.fixed_block_of_short_matches(30000, 4, 4, seed=2)),
                            ("30000 x (3..10, 1..4) at random", util.fixed_block_of_short_matches(30000, 10, 4, seed=4))):
    L = len(plain)
    src = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
    out = torch.zeros(L + 256, dtype=torch.uint8, device=dev)
    d = batch.to_device(batch.make_descs([0], [len(comp)], [0], [L], limit=[L]), dev); res = torch.zeros(16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(ctx, src, out, d, res, 1, L, 1)
    assert out[:L].cpu().numpy().tobytes() == plain
    t0 = time.perf_counter()
    for _ in range(3): batch.inflate_batch(ctx, src, out, d, res, 1, L, 1)
    t = (time.perf_counter() - t0) / 3
    a = time.perf_counter(); oracle.inflate(comp, decompressed_size=L); b = time.perf_counter()
    print("%s: %d -> %d bytes, gpu inflate %.3f ms, cpu oracle %.2f ms" % (name, len(comp), L, t * 1e3, (b - a) * 1e3))
