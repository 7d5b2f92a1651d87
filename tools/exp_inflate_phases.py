#!/usr/bin/env python3
"""Phase times inside inflate_batch_kernel, from a timing-only build of the library
(-DZD_INFLATE_PHASES: the per-stream results carry s_memtime deltas instead of lengths).
Compressed input is produced with the default library; ZIPC_HIP_PHASE_LIB is the timing build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zipc_amd
from zipc_amd import batch, synth, _lib
n = int(os.environ.get("N_STREAMS", "16384")); L = int(os.environ.get("LEN", "65536")); bits = int(os.environ.get("BITS", "4"))
dev = torch.device("cuda", 0); ctx = zipc_amd.Context(0)
if os.environ.get("ZEROS"):
    src = torch.zeros(n * L, dtype=torch.uint8, device=dev)
elif os.environ.get("DOC"):  # tools/bench_single.py's text: this repo's SURVEY.md over and over
    doc = open(os.path.join(ROOT, "SURVEY.md"), "rb").read()
    src = torch.from_numpy(np.frombuffer((doc * (n * L // len(doc) + 1))[:n * L], np.uint8).copy()).to(dev)
elif os.environ.get("JSON_AT"):  # a 64 KiB slice of tests/golden/zlib_streams.json (base64), n copies
    js = open(os.path.join(ROOT, "tests/golden/zlib_streams.json"), "rb").read()
    o = int(os.environ["JSON_AT"]); c = js[o:o + L]; c = c + bytes(L - len(c))
    src = torch.from_numpy(np.frombuffer(c * n, np.uint8).copy()).to(dev)
elif os.environ.get("CORPUS_CHUNK"):  # one chunk of tools/corpus.py, n copies
    from tools import corpus
    c = corpus.chunks(L)[int(os.environ["CORPUS_CHUNK"])]
    src = torch.from_numpy(np.frombuffer(c * n, np.uint8).copy()).to(dev)
elif os.environ.get("TEXT"):  # 64 KiB chunks of the reference's zip-docs texts instead of synthetic symbols
    import zipfile
    z = zipfile.ZipFile(os.path.join(ROOT, "tests/golden/zip-docs.zip"))
    app = z.read("zip-docs/APPNOTE.TXT"); rfc = z.read("zip-docs/rfc1951.txt")
    pieces = [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]
    src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(n)), np.uint8).copy()).to(dev)
else:
    src = synth.batch_bytes_torch(2, 0, n, L, bits, dev)
descs = batch.uniform_layout(n, L, batch.deflate_bound(L))
comp = torch.zeros(n * (int(descs["dst_off"][1]) if n > 1 else batch.deflate_bound(L)) + 256, dtype=torch.uint8, device=dev)
out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
d_descs = batch.to_device(descs, dev); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, 2, 0)
res = batch.results_from_device(d_res)
d_idescs = batch.to_device(batch.compact_descs(res, descs, L), dev)
# second library instance: the timing build
P = ctypes.CDLL(os.environ["ZIPC_HIP_PHASE_LIB"])
P.zipc_hip_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
h = ctypes.c_void_p(); assert P.zipc_hip_create(ctypes.byref(h), 0) == 0
P.zipc_hip_inflate_batch.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
for _ in range(2):
    assert P.zipc_hip_inflate_batch(h, comp.data_ptr(), out.data_ptr(), d_idescs.data_ptr(), d_ires.data_ptr(), n, L, 0) == 0
P.zipc_hip_synchronize.argtypes = [ctypes.c_void_p]; P.zipc_hip_synchronize(h)
r = batch.results_from_device(d_ires)
tot = r["status"].astype(np.float64) * 64; hdr = r["checksum"].astype(np.float64) * 64
wide = (r["out_len"] & 0xFFFFFFFF).astype(np.float64) * 64; plain = (r["out_len"] >> 32).astype(np.float64) * 64
print("per stream, clocks: total %.0f  headers %.0f (%.1f%%)  wide turns %.0f (%.1f%%)  plain steps %.0f (%.1f%%)  services+rest %.0f (%.1f%%)"
      % (tot.mean(), hdr.mean(), 100 * hdr.mean() / tot.mean(), wide.mean(), 100 * wide.mean() / tot.mean(), plain.mean(),
         100 * plain.mean() / tot.mean(), (tot - hdr - wide - plain).mean(), 100 * (tot - hdr - wide - plain).mean() / tot.mean()))
sp = out[:n * L].view(torch.int64).reshape(n, L // 8)[:, :8].cpu().numpy().astype(np.float64)
names = ["A walk", "A stitch", "probe+chain+tile setup", "B decode", "B far holes", "B near holes", "B flush"]
m = sp.mean(axis=0)
print("span (booked under 'plain steps'), clocks per stream: " + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, m[:7])) + "  tiles %.1f" % m[7])
