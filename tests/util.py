"""Shared inputs for the parity tests (CPU oracle tests and GPU tests use the same)."""
import base64
import json
import os
import random
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
