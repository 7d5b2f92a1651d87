#!/usr/bin/env python3
"""Lane utilisation of lz_match_window_kernel's chain walks, from a counting build
(tools/build_timing_lib.sh ZD_MATCH_UTIL; ZIPC_HIP_LIB must point at it): wave steps, lane steps
that walked a candidate, long compares and their lengths, handouts.  DATA=c2|c4|text."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch, synth, _lib
data = os.environ.get("DATA", "c2")
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
if data == "c4":
    n, L = 512, 1 << 20
    src = synth.batch_bytes_torch(2, 0, n, L, 3, dev)
elif data == "text":
    import zipfile
    n, L = 4096, 65536
    z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
    app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
    pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
    src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()).to(dev)
else:
    n, L = 4096, 65536
    src = synth.batch_bytes_torch(2, 0, n, L, 4, dev)
cap = batch.deflate_bound(L); descs = batch.uniform_layout(n, L, cap)
comp = torch.zeros(n * int(descs["dst_off"][1]) + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
dbg = C.CDLL(_lib.LIB_PATH).zipc_hip_debug_match_util
dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()
batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 0)
torch.cuda.synchronize()
assert dbg(out, 1) == 0
ctx.set_profiling(True); ctx.reset_kernel_times()
batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 0)
torch.cuda.synchronize()
assert dbg(out, 0) == 0
ws, walk, cmp_, cmp8, hand, fin = [int(out[i]) for i in range(6)]
ms = {k: v[1] / v[0] for k, v in ctx.kernel_times().items()}["lz_match"]
pos = n * L
print(json.dumps({"data": data, "lz_match_ms (counting build)": round(ms, 2), "positions": pos,
                  "wave steps": ws, "walked lane steps per position": round(walk / pos, 2),
                  "lane utilisation (walked / (wave steps x 64))": round(walk / (ws * 64.0), 3),
                  "long compares per position": round(cmp_ / pos, 3), "mean long compare, bytes": round(8.0 * cmp8 / max(cmp_, 1), 1),
                  "handout iterations / wave iterations": round(hand / (ws / 2.0), 3), "positions per handout": round(fin / max(hand, 1), 2)}))
