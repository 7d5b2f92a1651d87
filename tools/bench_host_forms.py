#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch forms (zipc_hip_deflate_many /
zipc_hip_inflate_many): C2's 16 384 x 64 KiB streams (N_STREAMS x STREAM_LEN) held in pageable host memory,
staged H2D, run, copied back D2H.  Context for DESIGN.md; bench.py's value is device-resident."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zipc_amd
from zipc_amd import _lib, synth

n = int(os.environ.get("N_STREAMS", "16384")); L = int(os.environ.get("STREAM_LEN", "65536"))
lib = _lib.lib()
ctx = zipc_amd.Context(0)
plain = [synth.stream_bytes_np(2, j, L, 4) for j in range(n)]
cap = lib.zipc_hip_deflate_bound(L)
comp = [np.zeros(cap, np.uint8) for _ in range(n)]
P = C.c_void_p * n; S = C.c_size_t * n
src = P(*[a.ctypes.data for a in plain]); slen = S(*([L] * n))
dst = P(*[a.ctypes.data for a in comp]); dcap = S(*([cap] * n))
res = (_lib.StreamResult * n)()
def deflate():
    assert lib.zipc_hip_deflate_many(ctx.handle, n, src, slen, 2, 1, dst, dcap, res) == 0
REPS = int(os.environ.get("REPS", "5"))
def timed(f):
    f()  # warm-up: buffer growth, first-touch of the pinned memory
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return ts
ts_def = timed(deflate); t_def = min(ts_def)
med = lambda ts: sorted(ts)[len(ts) // 2]
clen = S(*[int(res[i].out_len) for i in range(n)])
out = [np.zeros(L, np.uint8) for _ in range(n)]
odst = P(*[a.ctypes.data for a in out]); ocap = S(*([L] * n)); lim = S(*([L] * n))
ires = (_lib.StreamResult * n)()
def inflate():
    assert lib.zipc_hip_inflate_many(ctx.handle, n, dst, clen, lim, 1, odst, ocap, ires) == 0
ts_inf = timed(inflate); t_inf = min(ts_inf)
ok = all(int(ires[i].status) == 0 for i in range(n)) and all(np.array_equal(out[i], plain[i]) for i in range(0, n, 97))
gib = n * L / float(1 << 30)
print(json.dumps({"streams": n, "stream_len": L, "round_trip_ok": bool(ok), "deflate_many_gib_s": gib / t_def, "inflate_many_gib_s": gib / t_inf,
                  "round_trip_gib_s": gib / (t_def + t_inf), "reps": REPS, "rate_is": "best of reps",
                  "deflate_many_gib_s_median": gib / med(ts_def), "inflate_many_gib_s_median": gib / med(ts_inf),
                  "deflate_ms_all": [round(t * 1e3, 2) for t in ts_def], "inflate_ms_all": [round(t * 1e3, 2) for t in ts_inf],
                  "host_threads": int(os.environ.get("ZIPC_HIP_HOST_THREADS", min(8, os.cpu_count() or 1))), "sub_batches": int(os.environ.get("ZIPC_HIP_HOST_CHUNKS", 3)),
                  "note": "caller buffers pageable; library gathers them into one pinned buffer on host threads; the call runs as `sub_batches` pipelined sub-batches (gather, H2D, kernels, D2H, scatter overlap across them); host-side memcpy time varies from call to call, so read the median"}))
