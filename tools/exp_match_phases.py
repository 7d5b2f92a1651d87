#!/usr/bin/env python3
"""Where a tile of lz_match_window_kernel spends its time, from a timing-only build of the
library (-DZD_MATCH_PHASES: s_memtime deltas accumulated in a __device__ array, data paths
untouched).  ZIPC_HIP_LIB must point at that build.  s_memtime ticks at 100 MHz."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zipc_amd
from zipc_amd import batch, synth, _lib
n = int(os.environ.get("N_STREAMS", "16384")); L = int(os.environ.get("LEN", "65536")); bits = int(os.environ.get("BITS", "4"))
reps = int(os.environ.get("REPS", "3"))
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
if os.environ.get("TEXT"):  # 64 KiB chunks of the reference's zip-docs texts instead of synthetic symbols
    import zipfile, numpy as np
    z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
    app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
    pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
    src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()).to(dev)
else:
    src = synth.batch_bytes_torch(4 if L > 65536 else 2, 0, n, L, bits, dev)
descs = batch.uniform_layout(n, L, batch.deflate_bound(L))
comp = torch.zeros(n * int(descs["dst_off"][1]) + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
dbg = C.CDLL(_lib.LIB_PATH).zipc_hip_debug_match_phases
dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 0)  # warm-up
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
assert dbg(out, 1) == 0
ctx.set_profiling(True); ctx.reset_kernel_times()
for _ in range(reps):
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 0)
torch.cuda.synchronize()
assert dbg(out, 0) == 0
stage, loop, waves, loop_wall, wgs, wg_wall, wg_wall_rt = [int(out[i]) for i in range(7)]
ms = {k: v[1] / v[0] for k, v in ctx.kernel_times().items()}["lz_match"]
mhz = wg_wall / (wg_wall_rt / 100.0)  # s_memtime ticks per us, calibrated on s_memrealtime (100 MHz)
us = lambda ticks: ticks / mhz
res = batch.results_from_device(d_res)
tile_cu_us = ms * 1e3 * 256 / (wgs / reps)  # one workgroup per CU at a time; slot [4] counts tiles
wg = us(wg_wall / wgs)
print(json.dumps({
    "kernel_ms (timing build; compare with the product kernel's time from tools/exp_inflate.py to see the perturbation)": round(ms, 3), "tiles per launch": wgs // reps,
    "all streams ok": bool((res["status"] == 0).all()), "s_memtime MHz (calibrated)": round(mhz, 1),
    "per tile, us": {
        "CU time per tile (kernel time x 256 CUs / tiles)": round(tile_cu_us, 2),
        "inside a workgroup (tile start -> all waves done)": round(wg, 2),
        "staging, mean over waves (tile start -> window ready)": round(us(stage / waves), 2),
        "next-window load issue + match loop, mean over waves": round(us(loop / waves), 2),
        "the same, slowest wave of the workgroup": round(us(loop_wall / wgs), 2),
        "not inside any workgroup (launch, drain)": round(tile_cu_us - wg, 2),
    }}))
