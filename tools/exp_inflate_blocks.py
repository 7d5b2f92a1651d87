#!/usr/bin/env python3
"""One stream inflated by a wave per block (inflate.hip) against the stream's one wave (ZIPC_HIP_INFLATE_BLOCKS=0):
streams of the reference's encoder (this library's deflate) and of zlib at three levels, of symbols, text and binary
data; checked against the source, wall time and per-kernel time.  LEN = bytes of source (default 1 MiB)."""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch, synth

dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
L = int(os.environ.get("LEN", str(1 << 20)))
REPS = int(os.environ.get("REPS", "5"))

def sources():
    yield "4-bit symbols", synth.stream_bytes_np(2, 0, L, 4).tobytes()
    text = open(os.path.join(ROOT, "SURVEY.md"), "rb").read() + open(os.path.join(ROOT, "DESIGN.md"), "rb").read()
    yield "text", (text * (L // len(text) + 1))[:L]
    rng = np.random.default_rng(5)
    words = rng.integers(0, 1 << 14, L // 4, dtype=np.uint32) * np.uint32(0x10001)  # binary: records with repeats
    yield "binary", words.tobytes()[:L]
    yield "zeros", bytes(L)

def raw_zlib(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush()

for name, host in sources():
    src = torch.from_numpy(np.frombuffer(host, np.uint8).copy()).to(dev)
    cap = batch.deflate_bound(L); descs = batch.uniform_layout(1, L, cap)
    comp = torch.zeros(cap + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev); d_res = torch.zeros(16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, 1, L, L, 2, 0)
    n0 = int(batch.results_from_device(d_res)["out_len"][0])
    streams = [("own", comp[:n0].cpu().numpy().tobytes())] + [("zlib-%d" % lv, raw_zlib(host, lv)) for lv in (1, 6, 9)]
    for enc, c in streams:
        n = len(c)
        d_c = torch.from_numpy(np.frombuffer(c, np.uint8).copy()).to(dev)
        d_c = torch.cat([d_c, torch.zeros(256, dtype=torch.uint8, device=dev)])
        idesc = batch.uniform_layout(1, n, L); idesc["src_len"][0] = n; idesc["dst_cap"][0] = L
        d_id = batch.to_device(idesc, dev); d_ires = torch.zeros(16, dtype=torch.uint8, device=dev)
        out = torch.full((L + 256,), 0xA5, dtype=torch.uint8, device=dev)
        for crc_op in (0, 1):
            out.fill_(0xA5)
            batch.inflate_batch(ctx, d_c, out, d_id, d_ires, 1, L, crc_op); ctx.synchronize()
            r = batch.results_from_device(d_ires)
            ok = int(r["status"][0]) == 0 and int(r["out_len"][0]) == L and torch.equal(out[:L], src) and bool((out[L:] == 0xA5).all())
            if crc_op == 1: ok = ok and int(r["checksum"][0]) == (zlib.crc32(host) & 0xFFFFFFFF)
            if not ok:
                bad = (out[:L] != src).nonzero()
                print("MISMATCH", name, enc, crc_op, dict(status=int(r["status"][0]), out_len=int(r["out_len"][0]), blocks=ctx.last_inflate_blocks(),
                      first_bad=int(bad[0]) if len(bad) else -1, n_bad=len(bad)))
        blocks = ctx.last_inflate_blocks()
        ctx.set_profiling(True); ctx.reset_kernel_times()
        torch.cuda.synchronize(); a = time.perf_counter()
        for _ in range(REPS): batch.inflate_batch(ctx, d_c, out, d_id, d_ires, 1, L, 0)
        ctx.synchronize(); ms = (time.perf_counter() - a) / REPS * 1e3
        ks = {k: round(v[1] / REPS, 3) for k, v in ctx.kernel_times().items()}
        ctx.set_profiling(False)
        print(json.dumps({"input": "%d B of %s" % (L, name), "encoder": enc, "src_len": n, "blocks": blocks, "inflate_ms": round(ms, 3), "kernels_ms": ks}))
