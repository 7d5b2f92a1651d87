#!/usr/bin/env python3
"""C3 (both checksums of one buffer): wall ms per call and the kernels' own times, for the build and
environment the process was started with (ZIPC_HIP_CHECKSUM_FUSED=0: the two passes).  N bytes (default 4 GiB)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zipc_amd
from zipc_amd import batch, synth
from zipc_amd._lib import lib
n = int(os.environ.get("N", str(4 << 30)))
dev = torch.device("cuda", 0)
ctx = zipc_amd.Context(0)
buf = synth.batch_bytes_torch(3, 0, 1, n, 8, dev)
out = torch.zeros(2, dtype=torch.int32, device=dev)
first = batch.checksum_device(ctx, buf)
def call():
    st = lib().zipc_hip_checksum_device(ctx.handle, buf.data_ptr(), n, 1, 1, out.data_ptr())
    assert st == 0
ts = []
for _ in range(int(os.environ.get("REPS", "7"))):
    ctx.synchronize(); a = time.perf_counter(); call(); ctx.synchronize(); ts.append((time.perf_counter() - a) * 1e3)
ctx.set_profiling(True); ctx.reset_kernel_times()
for _ in range(3): call()
ctx.synchronize()
k = {name: round(v[1] / v[0], 3) for name, v in ctx.kernel_times().items()}
ctx.set_profiling(False)
again = batch.checksum_device(ctx, buf)
ms = sorted(ts)[len(ts) // 2]
print(json.dumps({"n": n, "fused": os.environ.get("ZIPC_HIP_CHECKSUM_FUSED", "1"), "crc": hex(first[0]), "adler": hex(first[1]),
                  "same": first == again, "ms": round(ms, 3), "best_ms": round(min(ts), 3), "gib_s": round(n / 2**30 / ms * 1e3, 1), "kernels_ms": k}))
