#!/usr/bin/env python3
"""A call of N long streams deflated (default 64 x 1 MiB; KIND text | binary | symbols, LEVEL 1-3): wall time per call,
round trip checked by zlib.  Under rocprofv3 --kernel-trace --stats: the kernels of the few-long-streams forms."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch, synth
import bench

dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
N = int(os.environ.get("N", "64")); L = int(os.environ.get("LEN", str(1 << 20))); REPS = int(os.environ.get("REPS", "5"))
LEVEL = int(os.environ.get("LEVEL", "2")); KIND = os.environ.get("KIND", "text")
rng = np.random.default_rng(3)
if KIND == "text": plain = [bench.leg_plain(("textlong",), j, L) for j in range(N)]
elif KIND == "symbols": plain = [synth.stream_bytes_np(2, j, L, 4).tobytes() for j in range(N)]
else: plain = [(rng.integers(0, 1 << 14, L // 4, dtype=np.uint32) * np.uint32(0x10001)).tobytes() for j in range(N)]
src = torch.from_numpy(np.frombuffer(b"".join(plain), np.uint8).copy()).to(dev)
cap = batch.deflate_bound(L); descs = batch.uniform_layout(N, L, cap)
comp = torch.zeros(int(descs["dst_off"][-1]) + cap + 256, dtype=torch.uint8, device=dev); d_res = torch.zeros(16 * N, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev)
ts = []
for rep in range(REPS + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, N, L, N * L, LEVEL, 1, sync=False); ctx.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
r = batch.results_from_device(d_res); hc = comp.cpu().numpy()
ok = all(zlib.decompress(hc[int(descs["dst_off"][i]):int(descs["dst_off"][i]) + int(r["out_len"][i])].tobytes(), -15) == plain[i] for i in (0, N - 1))
print("%s level %d, %d x %d B: %s  ratio %.3f  ms min %.2f median %.2f  (%.1f GiB/s)" % (KIND, LEVEL, N, L, "ok" if ok else "MISMATCH",
      float(r["out_len"].sum()) / (N * L), min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2], N * L / 2**30 / (min(ts[1:]) / 1e3)), flush=True)
