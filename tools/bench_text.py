#!/usr/bin/env python3
"""The whole path on 1 GiB of REAL TEXT -- 64 KiB chunks of the two documents in the reference's own
zip-docs.zip fixture (APPNOTE.TXT, rfc1951.txt), repeated to 16 384 streams -- device-resident, per
kernel.  The BASELINE configs are i.i.d. symbols, whose hash chains are 1.2 (C2) or 8.5 (C4)
candidates long; text has 34 per position, a quarter of them agreeing in 8 bytes and more, and
stresses lz_match and lz_chain in ways those do not.  Sampled streams are compared with the oracle."""
import os, sys, json, zipfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch, zipc_amd, oracle
from zipc_amd import batch
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
z = zipfile.ZipFile("tests/golden/zip-docs.zip")
app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
L = 65536; n = 16384
chunks = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
for name, pieces in (("real text (APPNOTE / rfc1951, 4 distinct 64 KiB chunks repeated)", chunks),):
    host = np.frombuffer(b"".join(pieces[i % len(pieces)] for i in range(n)), np.uint8).copy()
    src = torch.from_numpy(host).to(dev)
    cap = batch.deflate_bound(L); descs = batch.uniform_layout(n, L, cap); slot = int(descs["dst_off"][1])
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev); out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev); d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 1)
    res = batch.results_from_device(d_res)
    d_id = batch.to_device(batch.compact_descs(res, descs, L), dev)
    batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1)
    ok = bool(torch.equal(out[:n * L], src))
    exact = all(comp[int(descs["dst_off"][j]):int(descs["dst_off"][j]) + int(res["out_len"][j])].cpu().numpy().tobytes() == oracle.deflate(pieces[j % len(pieces)], level=2)[1] for j in (0, 1, 2, 3, n - 1))
    ctx.set_profiling(True); ctx.reset_kernel_times()
    for _ in range(3):
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 1)
        batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1)
    ks = {k: round(v[1] / v[0], 3) for k, v in ctx.kernel_times().items()}
    tot = sum(v[1] for v in ctx.kernel_times().values()) / 3
    print(json.dumps({"data": name, "round_trip": ok, "sampled_bytes_equal_oracle": exact, "ratio": round(float(res["out_len"].sum()) / (n * L), 4),
                      "step_ms_sum_of_kernels": round(tot, 2), "gib_s": round(1.0 / (tot / 1e3), 2), "kernels_ms": ks}))
