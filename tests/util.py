"""Shared inputs for the parity tests (CPU oracle tests and GPU tests use the same)."""
import base64
import json
import os
import random
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
FOX = b"The quick brown fox jumps over the lazy dog"  # test/test.ml:14
WORDS = [b"deflate", b"inflate", b"huffman", b"window", b"stream", b"block", b" ", b" the ", b"zip",
         b"archive", b"\n", b"checksum", b"0123456789"]


def kat():
    return json.load(open(os.path.join(GOLDEN, "kat.json")))


def zlib_streams():
    js = json.load(open(os.path.join(GOLDEN, "zlib_streams.json")))
    for s in js["streams"]:
        s = dict(s)
        s["raw"] = base64.b64decode(s["raw_b64"])
        yield s


def zip_docs():
    return open(os.path.join(GOLDEN, "zip-docs.zip"), "rb").read()


def zip_docs_members():
    z = zip_docs()
    for m in kat()["zip_docs"]["members"]:
        yield m, z[m["data_start"]:m["data_start"] + m["compressed_size"]]


def rand_bytes(n, seed, bits=8):
    r = random.Random(seed)
    if bits == 8:
        return r.randbytes(n)
    return bytes(r.randrange(1 << bits) for _ in range(n))


def text(n, seed):
    r = random.Random(seed)
    out = bytearray()
    while len(out) < n:
        out += r.choice(WORDS)
    return bytes(out[:n])


def trip_strings():
    """the reference's round-trip strings and expected block kinds (test/test.ml:38-42)"""
    return [(base64.b64decode(t["s_b64"]), t["block"]) for t in kat()["trip"]]


def fib_block(k, pad, seed=1):
    """a block whose Huffman codes outgrow their length limits: k rare byte values with Fibonacci
    counts, every occurrence followed by `pad` random bytes of the other 256 - k values (no 4 bytes
    repeat, so everything stays a literal).  With the right (k, pad) the litlen code wants more than
    15 bits, or the code-length code more than 7, and Huffman.lengths_of_freqs takes its
    flatten-and-retry branch (zd.ml:470-473) -- oracle.huffman_retries() tells."""
    r = random.Random(seed)
    f = [1, 1]
    while len(f) < k:
        f.append(f[-1] + f[-2])
    occ = [i for i, c in enumerate(f[:k]) for _ in range(c)]
    r.shuffle(occ)
    out = bytearray()
    for sym in occ:
        out.append(sym)
        out += bytes(r.randrange(k, 256) for _ in range(pad))
    return bytes(out)


def far_match_data(seed=21):
    """random bytes with short repeats planted at distances 32766 .. 32770 (the window is 32768:
    zd.ml:1143) whose second copy lies within a few bytes of a multiple of 16384 -- where the chain
    builder sweeps entries that have left the window (zd.ml:1187)"""
    r = random.Random(seed)
    b = bytearray(r.randbytes(150000))
    k = 0
    for base in (32768, 49152, 65536, 81920, 98304, 114688, 131072):
        for delta in (-3, -1, 0, 1, 2, 5):
            pos = base + delta + 40 * k
            dist = 32766 + k % 5
            n = 6 + k % 13
            b[pos:pos + n] = b[pos - dist:pos - dist + n]
            k += 1
    return bytes(b)


def record_table(n_records, seed=31):
    """A table of 24-byte records, most fields of a record the one before's (an ELF symbol table): its matches are a few
    long chains side by side, each link a copy of the record before -- the one-wave inflate's hole rounds took a round a
    link (corpus chunks of libhost_sim.so: 55 % of their time, round 5)"""
    r = random.Random(seed)
    out = bytearray()
    name, value = 1, 0x401000
    for _ in range(n_records):
        name += r.randrange(4, 40)
        value += r.choice([16, 16, 32, 48, 64, 176, 4096])
        out += struct.pack("<IBBHQQ", name, r.choice([0x12, 0x12, 0x12, 0x11, 0x22]), 0, r.choice([14, 14, 14, 16, 25]), value,
                           r.choice([0, 0, 8, 16, 16, 43, 176]))
    return bytes(out)


_VECTOR_INPUTS = {}


def vector_input(name):
    """the named inputs of tests/golden/deflate_vectors.json (made by tests/golden/make_deflate_vectors.py from a second,
    independent reading of the reference's encoder); rebuilt here, nothing but hashes is stored"""
    if name in _VECTOR_INPUTS:
        return _VECTOR_INPUTS[name]
    import sys

    import numpy as np

    sys.path.insert(0, os.path.dirname(HERE))
    from zipc_amd import synth

    if name.startswith("trip"):
        v = trip_strings()[int(name[4:])][0]
    elif name.startswith("zipdocs_"):
        m, raw = [(m, raw) for m, raw in zip_docs_members() if name[8:] in m["path"].lower().replace(".txt", "")][0]
        v = zlib.decompress(raw, -15)
        assert len(v) == m["decompressed_size"] and zlib.crc32(v) == m["crc32"]
    elif name.startswith("c2_stream"):
        v = synth.stream_bytes_np(2, int(name[9:]), 65536, 4).tobytes()
    elif name.startswith("c4_stream"):
        parts = name[9:].split("_")
        v = synth.stream_bytes_np(4, int(parts[0]), 1 << 20, 3).tobytes()
        if len(parts) > 1:
            v = v[:int(parts[1][:-1]) * 1024]
    elif name == "zeros1M":
        v = bytes(1 << 20)
    elif name == "ff4200":
        v = b"\xff" * 4200
    elif name == "rand200k":
        v = rand_bytes(200000, 77)
    elif name == "fox":
        v = FOX
    elif name == "empty":
        v = b""
    else:
        v = deflate_cases()[name]
    _VECTOR_INPUTS[name] = v
    return v


def deflate_cases(small=False):
    """name -> plaintext: the edge cases the reference tests plus multi-block, run,
    period, incompressible and mixed inputs"""
    c = {
        "empty": b"", "a": b"a", "abc": b"abc", "abcd": b"abcd", "hellohello": b"hellohello",
        "palindrome": b"abcdefghijklmnopqrstuvwxyzzyxwvutsrqponmlkjihgfedcba",
        "ramp": bytes((i + 1) % 255 for i in range(256)),
        "limits": b"Keep it to the limits.",
        "fox": FOX,
        "zeros5k": bytes(5000),
        "nib20k": rand_bytes(20000, 1, 4),
        "rand3k": rand_bytes(3000, 2),
        "text30k": text(30000, 3),
        "period2": b"ab" * 3000, "period9": b"abcdefghi" * 700, "period300": rand_bytes(300, 9) * 40,
    }
    if not small:
        c.update({
            "zeros70k": bytes(70000), "ff200k": b"\xff" * 200000,
            "nib64k": rand_bytes(65536, 11, 4), "nib3_200k": rand_bytes(200000, 12, 3),
            "rand70k": rand_bytes(70000, 13), "text150k": text(150000, 14),
            "mixed": text(30000, 4) + rand_bytes(5000, 15) + bytes(40000) + rand_bytes(70000, 5, 4)
            + text(66000, 6),
            "len65534": rand_bytes(65534, 7, 4), "len65535": rand_bytes(65535, 8, 4),
            "len65537": rand_bytes(65537, 9, 5),
            "zeros1M": bytes(1 << 20),
            # Huffman length limits (litlen > 15 bits, code-length code > 7 bits, both, and both in a
            # multi-block stream where the code-length counts carry over: Q1) and the window's edge
            "fib_litlen": fib_block(16, 5), "fib_codelen": fib_block(14, 2), "fib_both": fib_block(18, 2),
            "fib_multi": fib_block(18, 2) * 4 + fib_block(17, 5), "far_match": far_match_data(),
            "record_table": record_table(3000),
        })
        for m, raw in zip_docs_members():
            c[m["path"]] = zlib.decompress(raw, -15)
    return c


def corrupt_variants(raw, seed, count):
    """deterministic damaged copies of a deflate stream (bit flips near the
    header, truncations, byte smashes)"""
    r = random.Random(seed)
    out = []
    for _ in range(count):
        b = bytearray(raw)
        k = r.randrange(3)
        if k == 0 and b:
            i = r.randrange(min(len(b), 200))
            b[i] ^= 1 << r.randrange(8)
        elif k == 1 and len(b) > 1:
            b = b[:r.randrange(len(b))]
        else:
            for _ in range(3):
                if b:
                    b[r.randrange(len(b))] = r.randrange(256)
        out.append(bytes(b))
    return out


class _Bits:
    def __init__(self):
        self.bits = []

    def put(self, v, n):  # LSB first (extra bits, header fields)
        for i in range(n):
            self.bits.append((v >> i) & 1)

    def code(self, c, n):  # Huffman codes go MSB first
        for i in range(n - 1, -1, -1):
            self.bits.append((c >> i) & 1)

    def bytes(self):
        b = bytearray((len(self.bits) + 7) // 8)
        for i, v in enumerate(self.bits):
            b[i >> 3] |= v << (i & 7)
        return bytes(b)


def _random_code_lengths(r, n_used, max_len):
    """lengths of a random complete prefix code over n_used symbols (n_used >= 2)"""
    lens = [1, 1]
    while len(lens) < n_used:
        cands = [i for i, l in enumerate(lens) if l < max_len]
        i = r.choice(cands)
        l = lens.pop(i) + 1
        lens += [l, l]
    r.shuffle(lens)
    return lens


def _canonical(lengths):
    codes, code = {}, 0
    for l in range(1, 16):
        for sym, sl in enumerate(lengths):
            if sl == l:
                codes[sym] = code
                code += 1
        code <<= 1
    return codes


def random_dynamic_stream(r, damage, max_symbols=120):
    """one final dynamic block with a random complete litlen / dist / codelen code,
    a few literals and matches (up to max_symbols of them); damage > 0 perturbs that
    many code lengths (the stream then is usually, not always, rejected)"""
    n_lit = r.choice([1, 2, 5, 20, 60, 200])
    used = set(r.sample(range(256), n_lit)) | {256} | set(r.sample(range(257, 286), r.choice([0, 1, 3, 10])))
    used = sorted(used)
    ll = [0] * 286
    for sym, l in zip(used, _random_code_lengths(r, max(len(used), 2), 15) if len(used) > 1 else [1]):
        ll[sym] = l
    n_dist = r.choice([1, 1, 2, 6, 30])
    dl = [0] * 30
    dsyms = sorted(r.sample(range(30), n_dist))
    if n_dist == 1:
        dl[dsyms[0]] = 1
    else:
        for sym, l in zip(dsyms, _random_code_lengths(r, n_dist, 15)):
            dl[sym] = l
    for _ in range(damage):
        which = r.choice([ll, dl])
        which[r.randrange(len(which))] = r.randrange(16)
    hlit = max(257, max([i for i, l in enumerate(ll) if l] + [256]) + 1)
    hdist = max(1, max([i for i, l in enumerate(dl) if l] + [0]) + 1)
    seq = ll[:hlit] + dl[:hdist]
    # code-length alphabet: plain lengths, zero runs as 17/18, repeats as 16 now and then
    items, i = [], 0
    while i < len(seq):
        run = 1
        while i + run < len(seq) and seq[i + run] == seq[i]:
            run += 1
        if seq[i] == 0 and run >= 3 and r.random() < 0.8:
            n = min(run, 138)
            items.append((17, n - 3, 3) if n <= 10 else (18, n - 11, 7))
            i += n
        elif run >= 4 and r.random() < 0.6:
            items.append((seq[i], 0, 0))
            n = min(run - 1, 6)
            items.append((16, n - 3, 2))
            i += 1 + n
        else:
            items.append((seq[i], 0, 0))
            i += 1
    cl_used = sorted({it[0] for it in items})
    cl = [0] * 19
    if len(cl_used) == 1:
        cl[cl_used[0]] = 1
    else:
        for sym, l in zip(cl_used, _random_code_lengths(r, len(cl_used), 7)):
            cl[sym] = l
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    hclen = max(4, max(i for i, s in enumerate(order) if cl[s]) + 1)
    w = _Bits()
    w.put(1, 1); w.put(2, 2); w.put(hlit - 257, 5); w.put(hdist - 1, 5); w.put(hclen - 4, 4)
    for s_ in order[:hclen]:
        w.put(cl[s_], 3)
    clc = _canonical(cl)
    for sym, extra, nb in items:
        w.code(clc[sym], cl[sym]); w.put(extra, nb)
    lc, dc = _canonical(ll), _canonical(dl)
    lits = [s_ for s_ in used if s_ < 256 and ll[s_]]
    lens_ = [s_ for s_ in used if s_ > 256 and ll[s_]]
    produced = 0
    lbase = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    lext = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
    dbase = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    dext = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]
    for _ in range(r.randrange(1, max_symbols)):
        if lens_ and lits and produced > 0 and r.random() < 0.3:
            ok_d = [d for d in range(30) if dl[d] and dbase[d] <= produced]
            if ok_d:
                ls = r.choice(lens_); d = r.choice(ok_d)
                w.code(lc[ls], ll[ls]); e = r.randrange(1 << lext[ls - 257]); w.put(e, lext[ls - 257])
                de = r.randrange(1 << dext[d])
                while dbase[d] + de > produced:
                    de = r.randrange(1 << dext[d]) if dbase[d] + de > produced and dext[d] else 0
                    if dbase[d] + de <= produced:
                        break
                    de = 0
                w.code(dc[d], dl[d]); w.put(de, dext[d])
                produced += lbase[ls - 257] + e
                continue
        if lits:
            s_ = r.choice(lits)
            w.code(lc[s_], ll[s_])
            produced += 1
    w.code(lc[256], ll[256]) if ll[256] else None
    w.put(0, 7)
    return w.bytes()


def header_fuzz_streams(seed, n_random, n_flips):
    """deflate streams that stress the dynamic block header (read_dynamic_block
    zd.ml:623-669, Huffman.init_decoder zd.ml:355-391): random bits behind a
    "final, dynamic" block start with small code-length-code lengths (so that a fair
    share survives the first decoder and reaches the next checks), and single bit
    flips inside the header of valid dynamic streams."""
    r = random.Random(seed)
    out = []
    for _ in range(n_random):
        bits = [1, 0, 1]  # BFINAL = 1, BTYPE = 2
        def put(v, n):
            for i in range(n):
                bits.append((v >> i) & 1)
        put(r.randrange(32), 5); put(r.randrange(32), 5)
        hclen = r.randrange(16); put(hclen, 4)
        for _ in range(hclen + 4):
            put(r.choice([0, 0, 1, 2, 2, 3, 3, 3, 4, 4, 5, 7]), 3)
        while len(bits) < 8 * 96:
            bits.append(r.randrange(2))
        b = bytearray(len(bits) // 8)
        for i, v in enumerate(bits):
            b[i >> 3] |= v << (i & 7)
        out.append(bytes(b))
    for k in range(n_flips):  # valid random codes, and the same with a few lengths perturbed
        try:
            out.append(random_dynamic_stream(r, 0 if k % 2 == 0 else r.randrange(1, 3)))
        except (KeyError, IndexError, ValueError):
            pass  # a perturbed table the writer itself cannot encode with
    return out


def gpu_inflate_batch(ctx, streams, caps, has_limit, limits, crc_op):
    """One inflate_batch launch over `streams` through the C ABI: [(status, bytes, checksum)].
    limits[i] is the ?decompressed_size of stream i where has_limit[i]; caps[i] its dst_cap."""
    import numpy as np
    import torch

    from zipc_amd import batch

    dev = torch.device("cuda", 0)
    n = len(streams)
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    slots = [(c + 255) // 256 * 256 + 256 for c in caps]
    dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
    lim = [limits[i] if has_limit[i] else None for i in range(n)]
    if all(l is None for l in lim):
        descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
    else:
        assert all(l is not None for l in lim), "mixed limit / no limit: two launches"
        descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps, limit=lim)
    src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
    dst = torch.full((int(sum(slots)) + 256,), 0xA5, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(ctx, src, dst, batch.to_device(descs, dev), d_res, n, max(caps), crc_op)
    res = batch.results_from_device(d_res)
    out = dst.cpu().numpy()
    return [(int(res["status"][i]), out[int(dst_off[i]):int(dst_off[i]) + int(res["out_len"][i])].tobytes(),
             int(res["checksum"][i])) for i in range(n)]


class BitWriter:
    """deflate's bit order: fields least significant bit first, Huffman codes most significant bit first"""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def field(self, v, nbits):
        self.acc |= v << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, v, nbits):
        self.field(int(format(v, "0%db" % nbits)[::-1], 2), nbits)

    def bytes(self):
        return bytes(self.out) + (bytes([self.acc]) if self.n else b"")


def fixed_block_of_short_matches(n_matches, length, dist, lead=b"abcdefgh", seed=None):
    """one fixed-Huffman block: some literals, then n_matches times (length, dist), all legal but
    nothing an encoder would write: thousands of 3-byte matches in a row.  With a seed the lengths
    (3 .. length) and distances (1 .. dist) vary from match to match -- a stream of one repeated
    symbol has a period, and walks that start inside it never fall into step."""
    assert 3 <= length <= 10 and 1 <= dist <= 4
    r = random.Random(seed) if seed is not None else None
    w = BitWriter()
    w.field(1, 1); w.field(1, 2)              # BFINAL, BTYPE = fixed
    for c in lead:
        w.code(0x30 + c, 8)                   # literals 0..143
    plain = bytearray(lead)
    for _ in range(n_matches):
        ln, d = (r.randrange(3, length + 1), r.randrange(1, dist + 1)) if r else (length, dist)
        w.code(ln - 2, 7)                     # length symbols 257..264 (3..10): 7-bit codes 1..8
        w.code(d - 1, 5)                      # distances 1..4: codes 0..3, no extra bits
        for _ in range(ln):
            plain.append(plain[-d])
    w.code(0, 7)                              # end of block
    return w.bytes(), bytes(plain)


_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
              8193, 12289, 16385, 24577]
_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]


def random_fixed_block(seed, n_symbols, max_dist=300, max_len=20, lit_share=0.3):
    """one fixed-Huffman block of random literals and matches (any legal length up to max_len, any
    distance up to max_dist that the output so far allows): streams no encoder writes -- dense near
    matches, chains of them -- but every decoder must read.  Returns (compressed, plain)."""
    r = random.Random(seed)
    w = BitWriter()
    w.field(1, 1); w.field(1, 2)
    plain = bytearray()

    def lit(c):
        if c < 144: w.code(0x30 + c, 8)
        else: w.code(0x190 + c - 144, 9)
        plain.append(c)

    for c in b"seed":
        lit(c)
    for _ in range(n_symbols):
        if r.random() < lit_share:
            lit(r.randrange(256))
            continue
        ln = r.randrange(3, max_len + 1)
        d = r.randrange(1, min(max_dist, len(plain)) + 1)
        li = max(i for i, b in enumerate(_LEN_BASE) if b <= ln)
        if li == 28 and ln != 258: li = 27
        sym = 257 + li
        if sym < 280: w.code(sym - 256, 7)
        else: w.code(0xC0 + sym - 280, 8)
        ebits = 0 if li < 8 or li == 28 else (li - 4) // 4
        w.field(ln - _LEN_BASE[li], ebits)
        di = max(i for i, b in enumerate(_DIST_BASE) if b <= d)
        w.code(di, 5)
        w.field(d - _DIST_BASE[di], 0 if di < 4 else (di - 2) // 2)
        for _ in range(ln):
            plain.append(plain[-d])
    w.code(0, 7)
    return w.bytes(), bytes(plain)


def repeated_literal_blocks(seed, n_blocks, n_used, syms_per_block):
    """A deflate stream of n_blocks + 1 dynamic blocks with the very SAME bits (but the last one's "final" bit): a
    random complete code over n_used literals and the end-of-block symbol, syms_per_block random literals per block.
    Returns (stream, plain).  (What inflate_batch_few_kernel's shortcut for a repeated header is tested with: the
    header is a few hundred bits long, far more than stays staged behind an end of block.)"""
    import random

    r = random.Random(seed)
    lits = sorted(r.sample(range(256), n_used))
    ll = [0] * 286
    for sym, l in zip(lits + [256], _random_code_lengths(r, n_used + 1, 15)):
        ll[sym] = l
    seq = ll[:257] + [1]  # HLIT = 257 codes, HDIST = 1 code: distance symbol 0 with length 1, never used
    items, i = [], 0
    while i < len(seq):
        run = 1
        while i + run < len(seq) and seq[i + run] == seq[i]:
            run += 1
        if seq[i] == 0 and run >= 3:
            n = min(run, 138)
            items.append((17, n - 3, 3) if n <= 10 else (18, n - 11, 7))
            i += n
        else:
            items.append((seq[i], 0, 0))
            i += 1
    cl_used = sorted({it[0] for it in items})
    cl = [0] * 19
    for sym, l in zip(cl_used, _random_code_lengths(r, len(cl_used), 7)):
        cl[sym] = l
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    hclen = max(4, max(i for i, s in enumerate(order) if cl[s]) + 1)
    body = [r.choice(lits) for _ in range(syms_per_block)]

    def block(w, final):
        w.put(final, 1); w.put(2, 2); w.put(0, 5); w.put(0, 5); w.put(hclen - 4, 4)
        for s_ in order[:hclen]:
            w.put(cl[s_], 3)
        clc = _canonical(cl)
        for sym, extra, nb in items:
            w.code(clc[sym], cl[sym]); w.put(extra, nb)
        lc = _canonical(ll)
        for s_ in body:
            w.code(lc[s_], ll[s_])
        w.code(lc[256], ll[256])

    w = _Bits()
    for _ in range(n_blocks):
        block(w, 0)
    block(w, 1)
    return w.bytes(), bytes(body) * (n_blocks + 1)
