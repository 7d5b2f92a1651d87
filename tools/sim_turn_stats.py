#!/usr/bin/env python3
"""Wide-turn statistics of the inflate host model (tests/host_sim) on one synthetic
stream: symbols and bits per turn and why turns end.  CPU only; test tooling."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIM_INFLATE_WIDE"] = "1"
import numpy as np
import oracle
from tests import host_sim
from zipc_amd import synth

def main():
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = 65536
    if bits == 0:
        plain = (open(os.path.join(ROOT, "SURVEY.md"), "rb").read() * 4)[:n]
    else:
        plain = synth.stream_bytes_np(2, 0, n, bits).tobytes()
    st, comp, _ = oracle.deflate(plain, level=level, crc_op=0)
    sim = host_sim.lib()
    stats = (C.c_uint64 * 16).in_dll(sim, "sim_stats")
    dst = C.create_string_buffer(n + 64)
    out_len = C.c_uint64(); ck = C.c_uint32()
    r = sim.sim_inflate(comp, len(comp), dst, n + 64, 0, 0, 0, C.byref(out_len), C.byref(ck), 12)
    assert r == 0 and dst.raw[:n] == plain
    t, syms, m, b, sink, stop, late = [int(stats[i]) for i in range(7)]
    print("ratio %.4f  turns %d  symbols/turn %.2f  matches/turn %.2f  bits/turn %.1f  bits/symbol %.2f"
          % (len(comp) / n, t, syms / t, m / t, b / t, b / max(syms, 1)))
    print("turn ends: sink %.1f%%  stop entry %.1f%%  late %.1f%%" % (100 * sink / t, 100 * stop / t, 100 * late / t))

main()
