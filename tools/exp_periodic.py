#!/usr/bin/env python3
"""One 1 MiB stream of exactly periodic data (random period of 1000 / 20000 bytes) inflated alone:
what the reference's encoder makes of it is one match repeated, which inflate_batch copies as one
periodic copy per run (match_run / wave_copy_match, inflate.hip).  Device-resident, next to the oracle."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, random
import oracle, zipc_amd
from zipc_amd import batch
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
L = 1 << 20
r = random.Random(5)
for P in (1000, 20000):
    pat = bytes(r.randrange(256) for _ in range(P)); host = (pat * (L // P + 1))[:L]
    st, comp, _ = oracle.deflate(host, level=2)
    src = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
    out = torch.zeros(L + 256, dtype=torch.uint8, device=dev)
    d = batch.to_device(batch.make_descs([0], [len(comp)], [0], [L], limit=[L]), dev); res = torch.zeros(16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(ctx, src, out, d, res, 1, L, 1)
    assert out[:L].cpu().numpy().tobytes() == host
    t0 = time.perf_counter()
    for _ in range(5): batch.inflate_batch(ctx, src, out, d, res, 1, L, 1)
    t = (time.perf_counter() - t0) / 5
    a = time.perf_counter(); oracle.inflate(comp, decompressed_size=L); b = time.perf_counter()
    print("period", P, "comp", len(comp), "gpu inflate ms %.3f" % (t * 1e3), "cpu oracle ms %.2f" % ((b - a) * 1e3))
