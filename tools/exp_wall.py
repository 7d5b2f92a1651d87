#!/usr/bin/env python3
"""Experiment driver: WALL time (host clock around enqueue + synchronize, no per-kernel events) of
the device-resident batch forms on the shapes the round is judged on, with whatever library and
environment the process was started with (ZIPC_HIP_LIB, ZIPC_HIP_SLICES, ...).

  DATA   c2 (default: 16 384 x 64 KiB, 4-bit symbols) | text (16 384 x 64 KiB chunks of the
         zip-docs texts) | c4 (N_STREAMS x 1 MiB, 3-bit symbols; default 2048) | corpus (tools/corpus.py)
  LEVEL  1 fast, 2 default (default), 3 best        REPS  timed repetitions (default 5)
  CHECK  1: compare sampled streams with the oracle (default 0: round trip + CRC only)
One JSON line: ms per call of deflate, inflate, and of the two back to back (bench.py's step)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import zipc_amd
from zipc_amd import batch, synth


def make_src(data, n, L, dev):
    if data == "text":
        import zipfile
        z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
        app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
        pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
        host = np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()
        return torch.from_numpy(host).to(dev), (lambda j: pieces[j % 4])
    if data == "corpus":
        from tools import corpus
        chunks = corpus.chunks(L)
        host = np.frombuffer(b"".join(chunks[i % len(chunks)] for i in range(n)), np.uint8).copy()
        return torch.from_numpy(host).to(dev), (lambda j: chunks[j % len(chunks)])
    if data == "c1":  # BASELINE C1: one stream, 1 MiB of zeros
        return torch.zeros(n * L, dtype=torch.uint8, device=dev), (lambda j: bytes(L))
    cfg, bits = (4, 3) if data == "c4" else (2, int(os.environ.get("BITS", "4")))
    return synth.batch_bytes_torch(cfg, 0, n, L, bits, dev), (lambda j: synth.stream_bytes_np(cfg, j, L, bits).tobytes())


def main():
    data = os.environ.get("DATA", "c2")
    L = (1 << 20) if data in ("c4", "c1") else 65536
    n = int(os.environ.get("N_STREAMS", "2048" if data == "c4" else "1" if data == "c1" else "16384"))
    level = int(os.environ.get("LEVEL", "1" if data == "c1" else "2"))
    reps = int(os.environ.get("REPS", "5"))
    crc = int(os.environ.get("CRC_OP", "1"))
    dev = torch.device("cuda", 0)
    ctx = zipc_amd.Context(0)
    src, plain_of = make_src(data, n, L, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1]) if n > 1 else cap
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.reserve(ctx, n, L, n * L)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc)
    res = batch.results_from_device(d_res)
    ok = bool((res["status"] == 0).all())
    d_id = batch.to_device(batch.compact_descs(res, descs, L), dev)
    batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, crc)
    ires = batch.results_from_device(d_ires)
    ok = ok and bool((ires["status"] == 0).all()) and bool(torch.equal(out[:n * L], src))
    if crc:
        ok = ok and bool((ires["checksum"] == res["checksum"]).all())
    exact = None
    if os.environ.get("CHECK", "0") == "1":
        import oracle
        exact = True
        for j in sorted(set([0, 1, 2, 3, n // 2, n - 1])):
            st, c0, k0 = oracle.deflate(plain_of(j), level=level, crc_op=crc)
            o = int(descs["dst_off"][j])
            exact = exact and comp[o:o + int(res["out_len"][j])].cpu().numpy().tobytes() == c0 and (not crc or int(res["checksum"][j]) == k0)

    def timed(f):
        torch.cuda.synchronize(); ctx.synchronize()
        best, tot = 1e9, 0.0
        for _ in range(reps):
            a = time.perf_counter()
            f()
            ctx.synchronize()
            d = time.perf_counter() - a
            best = min(best, d); tot += d
        return round(tot / reps * 1e3, 3), round(best * 1e3, 3)

    td = timed(lambda: batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc, sync=False))
    ti = timed(lambda: batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, crc, sync=False))

    def step():
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc, sync=False)
        batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, crc, sync=False)

    ts = timed(step)
    res2 = batch.results_from_device(d_res)
    same = bool((res2["out_len"] == res["out_len"]).all() and (res2["checksum"] == res["checksum"]).all())
    gib = n * L / float(1 << 30)
    line = {"data": data, "n": n, "L": L, "level": level, "lib": os.path.basename(os.environ.get("ZIPC_HIP_LIB", "default")),
            "slices": os.environ.get("ZIPC_HIP_SLICES", "default"), "ok": ok and same, "exact": exact,
            "ratio": round(float(res["out_len"].sum()) / (n * L), 4),
            "deflate_ms": td[0], "inflate_ms": ti[0], "step_ms": ts[0], "best": [td[1], ti[1], ts[1]],
            "deflate_gib_s": round(gib / td[0] * 1e3, 2), "inflate_gib_s": round(gib / ti[0] * 1e3, 2),
            "step_gib_s": round(gib / ts[0] * 1e3, 2)}
    if os.environ.get("MATCH_COUNTS", "0") == "1":  # a -DZD_MATCH_COUNTS build (tools/build_timing_lib.sh ZD_MATCH_COUNTS)
        import ctypes as C
        from zipc_amd import _lib
        dbg = C.CDLL(_lib.LIB_PATH).zipc_hip_debug_match_counts
        dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        cnt = (C.c_ulonglong * 16)()
        assert dbg(cnt, 1) == 0
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc)
        assert dbg(cnt, 0) == 0
        names = ["waves", "outer_iterations", "rounds", "walking_slots", "compare_phases", "hit_slots", "handouts", "fin_slots", "alive_slots_first_form"]
        line["match_counts"] = {k: int(cnt[i]) for i, k in enumerate(names)}
    if os.environ.get("PARSE_COUNTS", "0") == "1":  # a -DZD_PARSE_COUNTS build
        import ctypes as C
        from zipc_amd import _lib
        dbg = C.CDLL(_lib.LIB_PATH).zipc_hip_debug_parse_counts
        dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        cnt = (C.c_ulonglong * 8)()
        assert dbg(cnt, 1) == 0
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, crc)
        assert dbg(cnt, 0) == 0
        line["parse_counts"] = {k: int(cnt[i]) for i, k in enumerate(["tiles", "turns", "tiles_with_a_turn", "chaining_lanes"])}
        line["parse_phases (a -DZD_PARSE_PHASES build: tiles, then shader clocks to macro steps / visited / symbols / tile end)"] = [int(cnt[i]) for i in range(5)]
    if os.environ.get("KERNELS", "0") == "1":
        ctx.set_profiling(True); ctx.reset_kernel_times()
        for _ in range(2):
            step()
        ctx.synchronize()
        line["kernels_ms"] = {k: round(v[1] / v[0], 3) for k, v in ctx.kernel_times().items()}
        ctx.set_profiling(False)
    print(json.dumps(line))


main()
