import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
L = 16 << 20
rng = np.random.default_rng(5)
host = (rng.integers(0, 1 << 14, L // 4, dtype=np.uint32) * np.uint32(0x10001)).tobytes()[:L]
c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(host) + c.flush()
n = len(comp)
src = torch.from_numpy(np.frombuffer(host, np.uint8).copy()).to(dev)
d_c = torch.cat([torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).to(dev), torch.zeros(256, dtype=torch.uint8, device=dev)])
idesc = batch.uniform_layout(1, n, L); idesc["src_len"][0] = n; idesc["dst_cap"][0] = L
d_id = batch.to_device(idesc, dev); d_ires = torch.zeros(16, dtype=torch.uint8, device=dev)
for rep in range(3):
    out = torch.full((L + 256,), 0xA5, dtype=torch.uint8, device=dev)
    batch.inflate_batch(ctx, d_c, out, d_id, d_ires, 1, L, 0); ctx.synchronize()
    bad = (out[:L] != src).nonzero().flatten().cpu().numpy()
    print("rep", rep, "blocks", ctx.last_inflate_blocks(), "n_bad", len(bad))
    if len(bad):
        runs = np.split(bad, np.where(np.diff(bad) != 1)[0] + 1)
        print(" runs", len(runs), [(int(r[0]), len(r)) for r in runs[:12]])
        p = int(bad[0]); o = out[:L].cpu().numpy(); s = np.frombuffer(host, np.uint8)
        print(" got ", o[p - 8:p + 24].tolist()); print(" want", s[p - 8:p + 24].tolist())
