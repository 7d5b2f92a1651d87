#!/usr/bin/env python3
"""How the one-wave inflate walks a 64 KiB chunk, on the CPU model (tests/host_sim: the kernel's own lane and span code on
an emulated wave): spans run, what they committed, how they ended, and the wide turns / single symbols that took the rest.
  python3 tools/exp_span_stats.py [corpus chunk numbers ...]     (no arguments: a C2 stream, a text chunk, corpus chunks 47 50 51)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SIM_INFLATE_WIDE"] = "1"; os.environ["SIM_INFLATE_SPAN"] = "a"
import oracle
from host_sim import lib as sim_lib
from tools import corpus
from zipc_amd import synth
sim = sim_lib()
span = (C.c_uint64 * 8).in_dll(sim, "sim_span_stats"); turn = (C.c_uint64 * 16).in_dll(sim, "sim_stats")
chunks = corpus.chunks(65536)
cases = [("c2 stream 0", synth.stream_bytes_np(2, 0, 65536, 4).tobytes()), ("text (APPNOTE 0)", chunks[0])]
for j in ([int(a) for a in sys.argv[1:]] or [47, 50, 51]):
    cases.append(("corpus chunk %d" % j, chunks[j]))
for name, data in cases:
    st, comp, _ = oracle.deflate(data, level=2)
    for i in range(8): span[i] = 0
    for i in range(16): turn[i] = 0
    dst = C.create_string_buffer(len(data) + 64); ol, ck = C.c_uint64(), C.c_uint32()
    st = sim.sim_inflate(comp, len(comp), dst, len(data), 1, len(data), 0, C.byref(ol), C.byref(ck), 24)
    assert st == 0 and dst.raw[:ol.value] == data
    print("%-18s comp %6d  spans %3d (again %d later %d off %d)  span bits %7d of %7d  span bytes %6d of %6d | wide turns %5d symbols %6d, others %s" % (
        name, len(comp), span[0], span[4], span[5], span[6], span[1], len(comp) * 8, span[2], len(data), turn[0], turn[1], list(turn[2:10])))
