#!/usr/bin/env python3
"""Where a tile of lz_tile_kernel spends its time, from a timing-only build of the library (-DZD_TILE_PHASES: shader-clock
deltas of wave 0 between the kernel's phases, summed over tiles; data paths untouched).  ZIPC_HIP_LIB must point at that build
(tools/build_variant.sh phases -DZD_TILE_PHASES lz_tile).  DATA c2 (default) | text."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch, synth, _lib
n = int(os.environ.get("N_STREAMS", "16384")); L = 65536; bits = int(os.environ.get("BITS", "4"))
reps = int(os.environ.get("REPS", "3")); level = int(os.environ.get("LEVEL", "2"))
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
if os.environ.get("DATA", "c2") == "text":
    import zipfile
    z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
    app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
    pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
    src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()).to(dev)
else:
    src = synth.batch_bytes_torch(2, 0, n, L, bits, dev)
descs = batch.uniform_layout(n, L, batch.deflate_bound(L))
comp = torch.zeros(n * int(descs["dst_off"][1]) + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
dbg = C.CDLL(_lib.LIB_PATH).zipc_hip_debug_tile_phases
dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 0)  # warm-up
torch.cuda.synchronize()
out = (C.c_ulonglong * 10)()
assert dbg(out, 1) == 0
ctx.set_profiling(True); ctx.reset_kernel_times()
for _ in range(reps):
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 0)
torch.cuda.synchronize()
assert dbg(out, 0) == 0
ms = {k: v[1] / v[0] for k, v in ctx.kernel_times().items()}["lz_tile"]
tiles = reps * n * ((L + 8191) // 8192)
names = ["search 1 (+barrier)", "search 2 pool", "barrier (+hi_exc)", "prefetch issue + parse A", "barrier + parse B",
         "parse C1", "barrier", "parse C2", "slide + stores + barriers (not the last tile's)"]
tot = sum(int(out[i]) for i in range(9))
# the shader clock's rate: ticks of a tile's phases against the CU time a tile takes (one workgroup per CU)
tile_us = ms * 1e3 * 256 / (tiles / reps)
print(json.dumps({"kernel_ms": round(ms, 3), "tiles": tiles // reps, "CU us per tile": round(tile_us, 2),
                  "ticks per tile": round(tot / tiles, 1),
                  "share": {names[i]: round(int(out[i]) / tot, 3) for i in range(9)},
                  "ticks": {names[i]: round(int(out[i]) / tiles, 1) for i in range(9)}}))
