#!/usr/bin/env python3
"""Which chunks of the corpus (tools/corpus.py) are slow: every distinct chunk as a batch of N copies of itself,
deflate (LEVEL) and inflate timed per chunk.  One line per chunk, slowest inflate first."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import zipc_amd
from zipc_amd import batch
from tools import corpus

n = int(os.environ.get("N_STREAMS", "1024")); L = 65536; level = int(os.environ.get("LEVEL", "2"))
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
chunks = corpus.chunks(L)
# where each chunk comes from
names = []
off = 0
import zipfile
z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
spans = [("APPNOTE.TXT", len(z.read("zip-docs/APPNOTE.TXT"))), ("rfc1951.txt", len(z.read("zip-docs/rfc1951.txt")))]
spans += [(os.path.relpath(f, ROOT), os.path.getsize(f)) for f in corpus.files()]
starts = np.cumsum([0] + [s for _, s in spans])
def origin(j):
    a = j * L
    k = int(np.searchsorted(starts, a, side="right") - 1)
    return spans[min(k, len(spans) - 1)][0]
cap = batch.deflate_bound(L); descs = batch.uniform_layout(n, L, cap); slot = int(descs["dst_off"][1])
comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev); out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev); d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
rows = []
only = [int(x) for x in os.environ.get("ONLY", "").split(",") if x]
for j, c in enumerate(chunks):
    if only and j not in only: continue
    src = torch.from_numpy(np.frombuffer(c * n, np.uint8).copy()).to(dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
    res = batch.results_from_device(d_res)
    d_id = batch.to_device(batch.compact_descs(res, descs, L), dev)
    batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1)
    ok = bool(torch.equal(out[:n * L], src))
    t = []
    for f in (lambda: batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1, sync=False),
              lambda: batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1, sync=False)):
        torch.cuda.synchronize(); ctx.synchronize()
        a = time.perf_counter(); f(); ctx.synchronize(); t.append((time.perf_counter() - a) * 1e3)
    km = ""
    if os.environ.get("KERNELS", "0") == "1":  # the deflate direction's kernels, ms
        ctx.set_profiling(True); ctx.reset_kernel_times()
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
        ctx.synchronize()
        km = "  " + " ".join("%s %.2f" % (k.replace("deflate_", "").replace("lz_", ""), v[1] / v[0]) for k, v in ctx.kernel_times().items() if v[1] / v[0] >= 0.05)
        ctx.set_profiling(False)
    rows.append((t[1], t[0], j, origin(j) + km, int(res["out_len"][0]), ok))
rows.sort(reverse=True, key=lambda r: r[1] if os.environ.get("SORT", "inflate") == "deflate" else r[0])
for r in rows:
    print("chunk %3d inflate %8.3f ms deflate %8.3f ms  comp %6d  ok %s  %s" % (r[2], r[0], r[1], r[4], r[5], r[3]))
