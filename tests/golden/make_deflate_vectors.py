#!/usr/bin/env python3
"""Makes tests/golden/deflate_vectors.json: what the SECOND reading of the reference's encoder (zd_second_reading.py,
written from src/zipc_deflate.ml without looking at oracle/zd_oracle.c) makes of a fixed list of inputs -- length and
sha256 of the compressed bytes, CRC-32 and Adler-32 as the fused forms return them, the kind of every block, how often
Huffman.lengths_of_freqs retried -- at `Fast, `Default and `Best (and `None where it says something).

The reference holds no vector for compressed bytes (test/test.ml:33-36 only round-trips) and cannot be run here (no
OCaml): this file does not turn the oracle into the reference, it turns one reading of zd.ml into two that must agree.
tests/test_oracle_pins.py holds oracle/zd_oracle.c against it, tests/test_gpu_parity.py the GPU.

Run in the build container:   python3 tests/golden/make_deflate_vectors.py [-j 8]      (pure Python: a few minutes)
Inputs are rebuilt by name in the tests (tests/util.py vector_input): nothing but numbers and hashes is stored.
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import zd_second_reading as Z  # noqa: E402

import util  # noqa: E402  (tests/util.py: the named inputs)

LEVELS3 = ("fast", "default", "best")

# name -> levels.  The inputs themselves: util.vector_input(name)
PLAN = [(n, ("none",) + LEVELS3) for n in ("trip0", "trip1", "trip2", "trip3", "trip4")] + [
    ("zipdocs_rfc1951", LEVELS3), ("zipdocs_appnote", LEVELS3),
    ("zeros1M", LEVELS3),
    ("c2_stream0", LEVELS3), ("c2_stream16383", ("default",)),
    ("c4_stream0", LEVELS3), ("c4_stream0_80k", ("best",)), ("c4_stream8191_200k", ("default",)), ("c4_stream4095", ("default",)),
    ("fib_litlen", LEVELS3), ("fib_codelen", LEVELS3), ("fib_both", LEVELS3), ("fib_multi", LEVELS3),
    ("far_match", LEVELS3),
    ("mixed", ("fast", "default")), ("len65534", ("default",)), ("len65535", ("default",)), ("len65537", ("default",)),
    ("record_table", LEVELS3), ("rand70k", ("none", "default")),
    ("ff4200", ("none", "default")), ("rand200k", ("none", "default")),
]
# ... and every named case of the parity tests (tests/util.py deflate_cases), at the three levels
PLAN += [(n, LEVELS3) for n in sorted(util.deflate_cases()) if n not in dict(PLAN)]


def one(job):
    name, level = job
    data = util.vector_input(name)
    t = time.time()
    crc, comp, stats = Z.crc_and_deflate(data, level, Z.CRC_32)
    # the Adler-32 of the fused form: one update per block over the block's source bytes (zd.ml:1081-1086).  The block
    # cuts are the parse's, so it is replayed over them instead of running the parse twice.
    adler = 1
    pos = 0
    for ln in stats["block_src_lens"]:
        adler = Z.adler_32_string_update(adler, data, pos, ln)
        pos += ln
    assert pos == len(data)
    return name, level, {
        "clen": len(comp), "sha256": hashlib.sha256(comp).hexdigest(), "crc32": crc, "adler32_fused": adler & 0xFFFFFFFF,
        "blocks": compact(stats["blocks"]), "huffman_retries": stats.get("huffman_retries", 0), "seconds": round(time.time() - t, 1)}


def compact(kinds):
    """['dynamic', 'dynamic', 'fixed'] -> 'dynamic*2 fixed'"""
    out = []
    for k in kinds:
        if out and out[-1][0] == k:
            out[-1][1] += 1
        else:
            out.append([k, 1])
    return " ".join(k if n == 1 else "%s*%d" % (k, n) for k, n in out)


def main():
    jobs_n = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else (os.cpu_count() or 1)
    only = [a for a in sys.argv[1:] if not a.startswith("-") and not a.isdigit()]
    jobs = [(n, lv) for n, lvs in PLAN for lv in lvs if not only or n in only]
    # the slowest first
    weight = {"best": 30, "default": 3, "fast": 1, "none": 0}
    jobs.sort(key=lambda j: -len(util.vector_input(j[0])) * weight[j[1]])
    out_path = os.path.join(HERE, "deflate_vectors.json")
    doc = json.load(open(out_path)) if only and os.path.exists(out_path) else {
        "made_by": "tests/golden/make_deflate_vectors.py: tests/golden/zd_second_reading.py, a reading of "
                   "src/zipc_deflate.ml:166-206,404-528,742-1277 independent of oracle/zd_oracle.c",
        "adler32_whole": {}, "vectors": {}}
    t0 = time.time()
    with mp.Pool(jobs_n) as pool:
        for name, level, rec in pool.imap_unordered(one, jobs):
            data = util.vector_input(name)
            v = doc["vectors"].setdefault(name, {"len": len(data), "sha256_plain": hashlib.sha256(data).hexdigest(), "levels": {}})
            seconds = rec.pop("seconds")  # (the log's, not the file's: the file is the same from run to run)
            v["levels"][level] = rec
            sys.stderr.write("%-22s %-8s %8d -> %8d  %-28s %6.1fs (elapsed %.0fs)\n" % (name, level, len(data), rec["clen"], rec["blocks"][:28],
                                                                                        seconds, time.time() - t0))
    # Adler_32.string over whole buffers (one string_update: first chunk len mod 5552, signed remainder: SURVEY Q6/Q7)
    for name in ("ff4200", "rand200k", "rand70k", "trip2", "zipdocs_rfc1951", "c2_stream0", "fox", "empty"):
        doc["adler32_whole"][name] = Z.adler_32_string(util.vector_input(name))
    for v in doc["vectors"].values():
        v["levels"] = dict(sorted(v["levels"].items()))
    doc["vectors"] = dict(sorted(doc["vectors"].items()))
    json.dump(doc, open(out_path, "w"), indent=1)
    sys.stderr.write("wrote %s (%d inputs)\n" % (out_path, len(doc["vectors"])))


if __name__ == "__main__":
    main()
