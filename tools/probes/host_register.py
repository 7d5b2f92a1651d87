#!/usr/bin/env python3
"""What pinning a caller's pageable buffer in place costs (hipHostRegister / hipHostUnregister) beside a copy of it
through pinned staging memory: the question behind "register members above a few MiB instead of staging them".
Prints one JSON line per buffer size."""
import ctypes as C, json, time
import numpy as np
import torch  # (its libamdhip64 first: zipc_amd/_lib.py says why)

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
assert torch.cuda.is_available()
torch.zeros(1, device="cuda")
for mib in (1, 8, 64, 512):
    n = mib << 20
    buf = np.random.default_rng(mib).integers(0, 255, n, dtype=np.uint8)  # touched pages
    dev, pin = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), n) == 0 and hip.hipHostMalloc(C.byref(pin), n, 0) == 0
    C.memset(pin, 0, n)
    row = {"MiB": mib}
    for rep in range(3):
        t0 = time.perf_counter(); r = hip.hipHostRegister(buf.ctypes.data, n, 0); t1 = time.perf_counter()
        assert r == 0, r
        hip.hipMemcpy(dev, buf.ctypes.data, n, 1); t2 = time.perf_counter()
        hip.hipHostUnregister(buf.ctypes.data); t3 = time.perf_counter()
        C.memmove(pin, buf.ctypes.data, n); t4 = time.perf_counter()
        hip.hipMemcpy(dev, pin, n, 1); t5 = time.perf_counter()
        hip.hipMemcpy(dev, buf.ctypes.data, n, 1); t6 = time.perf_counter()
        row["rep%d ms" % rep] = {"register": round((t1 - t0) * 1e3, 3), "h2d registered": round((t2 - t1) * 1e3, 3),
                                 "unregister": round((t3 - t2) * 1e3, 3), "memcpy to pinned (1 thread)": round((t4 - t3) * 1e3, 3),
                                 "h2d pinned": round((t5 - t4) * 1e3, 3), "h2d pageable (hipMemcpy)": round((t6 - t5) * 1e3, 3)}
    print(json.dumps(row))
