#!/usr/bin/env python3
"""One-wave inflate of N copies of a few FIXED 64 KiB inputs (slices of tests/golden/zlib_streams.json -- base64 --, base64 and
hex of random bytes, a table of records, text): ms per batch, for A/B runs of builds (ZIPC_HIP_LIB) that must not depend on what
the corpus of tools/corpus.py holds today (it is made of this repository's files)."""
import base64, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import zipc_amd, util
from zipc_amd import batch
n = int(os.environ.get("N_STREAMS", "4096")); L = 65536
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
js = open(os.path.join(ROOT, "tests/golden/zlib_streams.json"), "rb").read()
r = random.Random(5)
cases = [("json@%d" % o, js[o:o + L]) for o in ([int(x) for x in os.environ["JSON_AT"].split(",")] if os.environ.get("JSON_AT") else (0, 200000, 400000))]
cases.append(("b64 random", base64.b64encode(bytes(r.randrange(256) for _ in range(49152)))))
cases.append(("hex random", bytes(r.randrange(256) for _ in range(32768)).hex().encode()))
cases.append(("record table", util.record_table(2730)[:L]))
cases.append(("text", util.text(L, 3)))
cap = batch.deflate_bound(L); descs = batch.uniform_layout(n, L, cap); slot = int(descs["dst_off"][1])
comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev); out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev); d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
for name, c in cases:
    c = c + bytes(L - len(c))
    src = torch.from_numpy(np.frombuffer(c * n, np.uint8).copy()).to(dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 1)
    res = batch.results_from_device(d_res)
    d_id = batch.to_device(batch.compact_descs(res, descs, L), dev)
    batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1)
    ok = bool(torch.equal(out[:n * L], src))
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); ctx.synchronize()
        a = time.perf_counter(); batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1, sync=False); ctx.synchronize()
        best = min(best, (time.perf_counter() - a) * 1e3)
    print("%-14s comp %6d  inflate %7.3f ms  ok %s" % (name, int(res["out_len"][0]), best, ok))
