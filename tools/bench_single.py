#!/usr/bin/env python3
"""One stream alone (the shape every value of the reference's .mli has: one stream per call),
device-resident: 1 MiB of zeros (BASELINE C1's input) and 1 MiB of i.i.d. 4-bit symbols, deflate
and inflate, per kernel, next to the CPU oracle on the same input."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle, zipc_amd
from zipc_amd import batch, synth

dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
L = int(os.environ.get("LEN", str(1 << 20)))
for name, host in (("zeros", bytes(L)), ("4-bit symbols", synth.stream_bytes_np(2, 0, L, 4).tobytes()),
                   ("text", (open(os.path.join(ROOT, "SURVEY.md"), "rb").read() * (L // 50000 + 1))[:L])):
    src = torch.from_numpy(np.frombuffer(host, np.uint8).copy()).to(dev)
    cap = batch.deflate_bound(L); descs = batch.uniform_layout(1, L, cap)
    comp = torch.zeros(cap + 256, dtype=torch.uint8, device=dev); out = torch.zeros(L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev); d_res = torch.zeros(16, dtype=torch.uint8, device=dev); d_ires = torch.zeros(16, dtype=torch.uint8, device=dev)
    line = {"input": "%d B of %s" % (L, name)}
    for level in (1, 2):
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, 1, L, L, level, 0)
        res = batch.results_from_device(d_res)
        st0, c0, _ = oracle.deflate(host, level=level)
        assert comp[:int(res["out_len"][0])].cpu().numpy().tobytes() == c0
        d_id = batch.to_device(batch.compact_descs(res, descs, L), dev)
        out.fill_(0xA5)
        batch.inflate_batch(ctx, comp, out, d_id, d_ires, 1, L, 0)
        assert int(batch.results_from_device(d_ires)["status"][0]) == 0 and torch.equal(out[:L], src)
        ctx.set_profiling(True); ctx.reset_kernel_times()
        td = ti = 0.0
        for _ in range(5):
            a = time.perf_counter(); batch.deflate_batch(ctx, src, comp, d_descs, d_res, 1, L, L, level, 0)
            b = time.perf_counter(); batch.inflate_batch(ctx, comp, out, d_id, d_ires, 1, L, 0)
            td += b - a; ti += time.perf_counter() - b
        ks = {k: round(v[1] / v[0], 3) for k, v in ctx.kernel_times().items()}
        ctx.set_profiling(False)
        a = time.perf_counter(); oracle.deflate(host, level=level); b = time.perf_counter(); oracle.inflate(c0, decompressed_size=L); c = time.perf_counter()
        line["level %d" % level] = {"gpu_deflate_ms": round(td / 5 * 1e3, 3), "gpu_inflate_ms": round(ti / 5 * 1e3, 3), "kernels_ms": ks,
                                    "cpu_oracle_deflate_ms": round((b - a) * 1e3, 2), "cpu_oracle_inflate_ms": round((c - b) * 1e3, 2), "ratio": round(len(c0) / L, 5)}
    print(json.dumps(line))
