#!/usr/bin/env python3
"""Randomized parity of the many-stream host forms on RAGGED calls -- what an archive of real files looks like, and
what the suite's equal-sized batches do not reach: a call of 1 .. 2500 members whose lengths follow a heavy tail (most
of a few hundred bytes to tens of KiB, some of MiB, now and then ONE of tens or hundreds of MiB), every kind of content,
a random level, CRC-32 or Adler-32; then the same members back through inflate_many, some damaged, cut or under-limited,
and through inflate_many_check (results alone).  Every stream: status, length, checksum and bytes against zlib's round
trip and, where the stream is small enough for it, byte for byte against the oracle.  (Round 6: a library directory with
one 233 MB file among 491 others asked the many-wave parse for 475 GB of scratch; nothing in the suite had that shape.)

  fuzz_many_ragged.py [seed0] [calls]          BIG=1: let a call hold a member of up to 384 MiB
"""
import ctypes as C
import os
import random
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import zipc_amd  # noqa: E402
from zipc_amd import _lib  # noqa: E402

ORACLE_MAX = 600_000  # streams up to this length are held byte for byte against the oracle


def content(r, n):
    kind = r.randrange(8)
    if n == 0:
        return b""
    if kind == 0:
        return r.randbytes(n)
    if kind == 1:
        bits = r.randrange(1, 8)
        return (np.frombuffer(r.randbytes(n), np.uint8) & ((1 << bits) - 1)).astype(np.uint8).tobytes()
    if kind == 2:
        return bytes([r.randrange(256)]) * n
    if kind == 3:
        pat = r.randbytes(r.randrange(1, 400))
        return (pat * (n // len(pat) + 1))[:n]
    if kind == 4:
        words = [bytes(r.randrange(97, 123) for _ in range(r.randrange(2, 10))) for _ in range(r.randrange(3, 300))]
        out = bytearray()
        while len(out) < min(n, 1 << 16):
            out += r.choice(words) + b" "
        out = bytes(out)
        return (out * (n // len(out) + 1))[:n]
    if kind == 5:  # records
        rec = bytearray(r.randbytes(24))
        out = bytearray()
        while len(out) < min(n, 1 << 18):
            rec[r.randrange(24)] = r.randrange(256)
            out += rec
        out = bytes(out)
        return (out * (n // len(out) + 1))[:n]
    if kind == 6:  # stretches of everything
        parts, left = [], n
        while left > 0:
            k = min(left, r.randrange(1, 1 + max(1, n // 3)))
            parts.append(content(r, k) if k < n else r.randbytes(k))
            left -= k
        return b"".join(parts)
    elf = open(os.path.join(ROOT, "zipc_amd", "lib", "libzipc_hip.so"), "rb").read()  # a real binary, from a place of its own
    o = r.randrange(len(elf))
    return ((elf[o:] + elf[:o]) * (n // len(elf) + 1))[:n]


def length(r, big):
    u = r.random()
    if u < 0.05:
        return r.choice([0, 1, 2, 3, 4, 5, 63, 64, 65, 65533, 65534, 65535, 65536, 65537])
    if u < 0.75:
        return int(2 ** r.uniform(4, 15))
    if u < 0.95:
        return int(2 ** r.uniform(15, 20))
    if u < 0.995:
        return int(2 ** r.uniform(20, 23))
    return int(2 ** r.uniform(23, 28.5 if big else 25.5))


def one_call(lib, ctx, r, big):
    n = r.choice([1, 2, 3, 17, 255, 256, 257]) if r.random() < 0.2 else r.randrange(1, 2500 if r.random() < 0.3 else 400)
    budget = (3 << 30) if big else (1 << 30)
    plain, total = [], 0
    for _ in range(n):
        ln = length(r, big)
        if total + ln > budget:
            ln = r.randrange(0, 2000)
        plain.append(content(r, ln))
        total += len(plain[-1])
    level = r.choice([0, 1, 2, 2, 2, 3]) if total < (64 << 20) else r.choice([1, 2])
    crc_op = r.choice([1, 2])
    P, S = C.c_void_p * n, C.c_size_t * n
    caps = [int(lib.zipc_hip_deflate_bound(len(d))) for d in plain]
    small = r.randrange(n) if r.random() < 0.3 else -1
    if small >= 0 and len(plain[small]) > 40:
        caps[small] = r.randrange(0, 8)
    else:
        small = -1
    keep = [np.frombuffer(d, np.uint8) if d else np.zeros(1, np.uint8) for d in plain]
    outs = [np.full(c + 16, 0xA5, np.uint8) for c in caps]
    res = (_lib.StreamResult * n)()
    st = lib.zipc_hip_deflate_many(ctx.handle, n, P(*[a.ctypes.data for a in keep]), S(*[len(d) for d in plain]), level, crc_op,
                                   P(*[a.ctypes.data for a in outs]), S(*caps), res)
    assert st == 0, (st, lib.zipc_hip_last_error(ctx.handle))
    checks = 0
    comps = []
    for i, d in enumerate(plain):
        assert bool((outs[i][caps[i]:] == 0xA5).all()), ("guard", i)
        if i == small:
            assert (int(res[i].status), int(res[i].out_len)) == (16, 0), ("too small", i, int(res[i].status))
            comps.append(zlib.compress(d, 6)[2:-4])
            continue
        assert int(res[i].status) == 0, (i, int(res[i].status), len(d))
        c = outs[i][: int(res[i].out_len)].tobytes()
        assert zlib.decompress(c, -15) == d, ("round trip", i, len(d))
        want_sum = zlib.crc32(d) if crc_op == 1 else None
        if len(d) <= ORACLE_MAX:
            st0, c0, k0 = oracle.deflate(d, level=level, crc_op=crc_op)
            assert c == c0 and int(res[i].checksum) == k0, ("oracle", i, len(d), level)
        elif want_sum is not None:
            assert int(res[i].checksum) == want_sum, ("crc", i)
        comps.append(c)
        checks += 1
    # ---- back: some damaged, cut, under-limited
    limits = [len(d) for d in plain]
    expect = [0] * n
    for i in range(n):
        u = r.random()
        if len(comps[i]) > 30 and u < 0.08:
            b = bytearray(comps[i])
            for _ in range(r.randrange(1, 6)):
                b[r.randrange(len(b))] ^= 1 << r.randrange(8)
            comps[i] = bytes(b)
            expect[i] = None  # whatever the oracle says
        elif len(comps[i]) > 30 and u < 0.14:
            comps[i] = comps[i][: r.randrange(1, len(comps[i]))]
            expect[i] = None
        elif len(plain[i]) > 10 and u < 0.2:
            limits[i] = r.randrange(0, len(plain[i]))
            expect[i] = None
    keep2 = [np.frombuffer(c, np.uint8) if c else np.zeros(1, np.uint8) for c in comps]
    back = [np.full(l + 16, 0xA5, np.uint8) for l in limits]
    ires, cres = (_lib.StreamResult * n)(), (_lib.StreamResult * n)()
    srcp, srcl, lim = P(*[a.ctypes.data for a in keep2]), S(*[len(c) for c in comps]), S(*limits)
    st = lib.zipc_hip_inflate_many(ctx.handle, n, srcp, srcl, lim, crc_op, P(*[a.ctypes.data for a in back]), S(*limits), ires)
    assert st == 0, (st, lib.zipc_hip_last_error(ctx.handle))
    st = lib.zipc_hip_inflate_many_check(ctx.handle, n, srcp, srcl, lim, crc_op, S(*limits), cres)
    assert st == 0, (st, lib.zipc_hip_last_error(ctx.handle))
    for i, d in enumerate(plain):
        assert bool((back[i][limits[i]:] == 0xA5).all()), ("guard back", i)
        assert (int(cres[i].status), int(cres[i].out_len), int(cres[i].checksum)) == (int(ires[i].status), int(ires[i].out_len), int(ires[i].checksum)), ("check form", i)
        if expect[i] is None or len(d) <= ORACLE_MAX:
            if len(comps[i]) > 4 * ORACLE_MAX:  # (a damaged giant: zlib's verdict is enough)
                try:
                    ok = zlib.decompressobj(-15).decompress(comps[i], limits[i] + 1)
                    good = len(ok) <= limits[i]
                except zlib.error:
                    good = False
                if not good:
                    assert int(ires[i].status) != 0 or back[i][: int(ires[i].out_len)].tobytes() == d[: int(ires[i].out_len)], ("giant verdict", i)
                checks += 1
                continue
            st0, want, k0 = oracle.inflate(comps[i], decompressed_size=limits[i], crc_op=crc_op)
            assert int(ires[i].status) == st0, ("status", i, st0, int(ires[i].status), len(d))
            if st0 == 0:
                assert int(ires[i].out_len) == len(want) and back[i][: len(want)].tobytes() == want and int(ires[i].checksum) == k0, ("inflate", i)
            else:
                assert int(ires[i].out_len) == 0
        else:
            assert int(ires[i].status) == 0 and back[i][: len(d)].tobytes() == d, ("inflate big", i)
        checks += 1
    return n, total, checks


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    big = os.environ.get("BIG", "0") == "1"
    lib = _lib.lib()
    ctx = zipc_amd.Context(0)
    t0 = time.time()
    tot_n = tot_b = tot_c = 0
    for k in range(calls):
        r = random.Random(seed0 * 1000 + k)
        n, b, c = one_call(lib, ctx, r, big)
        tot_n += n
        tot_b += b
        tot_c += c
        print("call %d (seed %d): %d members, %.1f MiB, %d checks ok" % (k, seed0 * 1000 + k, n, b / 2 ** 20, c), flush=True)
    print("fuzz_many_ragged: %d calls, %d members, %.1f GiB, %d checks, 0 mismatches (%.0f s)" % (calls, tot_n, tot_b / 2 ** 30, tot_c, time.time() - t0))


if __name__ == "__main__":
    main()
