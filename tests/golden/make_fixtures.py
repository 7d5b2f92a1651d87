#!/usr/bin/env python3
"""Regenerates tests/golden/* (DATA only: inputs and expected outputs).

Run in the build container, where /root/reference exists:
    python tests/golden/make_fixtures.py

1. zip-docs.zip: the binary test fixture the reference's own test embeds as an
   OCaml string literal (test/test.ml:131-3294).  Only
   the data bytes are extracted; no reference source text is kept.
2. zlib_streams.json: raw deflate streams made with Python zlib (levels
   0/1/6/9 -> stored / fixed / dynamic blocks) with their expected plaintext
   CRC-32s: an independent inflate oracle.
3. kat.json: the reference's known-answer values (test/test.ml:14-26,38-42,46)
   and the fixture's member table (test/test.ml:84-108).
"""
import base64
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TEST = "/root/reference/test/test.ml"


def extract_zip_docs():
    text = open(REF_TEST, "rb").read().decode("latin-1")
    key = "let () = zip_docs_zip :="
    at = text.index(key) + len(key)
    i = text.index('"', at) + 1
    out = bytearray()
    while True:
        c = text[i]
        if c == '"':
            break
        if c == "\\":
            n = text[i + 1]
            if n == "x":
                out.append(int(text[i + 2:i + 4], 16))
                i += 4
            elif n == "\n":  # line continuation: skip following blanks
                i += 2
                while text[i] in " \t":
                    i += 1
            else:
                raise ValueError("unexpected escape %r" % n)
        else:
            out.append(ord(c))
            i += 1
    return bytes(out)


def splitmix64(seed):
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def gen_bytes(seed, n, bits):
    g = splitmix64(seed)
    out = bytearray()
    mask = (1 << bits) - 1
    while len(out) < n:
        v = next(g)
        for k in range(8):
            out.append((v >> (8 * k)) & mask)
    return bytes(out[:n])


def main():
    z = extract_zip_docs()
    assert len(z) == 56924 and z[:4] == b"PK\x03\x04", len(z)
    open(os.path.join(HERE, "zip-docs.zip"), "wb").write(z)

    words = [b"deflate", b"inflate", b"huffman", b"window", b"stream", b"block", b" ", b" the ",
             b"zip", b"archive", b"\n", b"checksum", b"0123456789"]
    g = splitmix64(7)
    text = b"".join(words[next(g) % len(words)] for _ in range(6000))
    plains = {
        "empty": b"",
        "a": b"a",
        "hellohello": b"hellohello",
        "text": text,
        "nibbles": gen_bytes(1, 70000, 4),
        "random": gen_bytes(2, 20000, 8),
        "zeros": bytes(100000),
        "ramp": bytes((i * 7) & 0xFF for i in range(3000)),
    }
    streams = []
    for name, p in plains.items():
        for lvl in (0, 1, 6, 9):
            for strat, sname in ((zlib.Z_DEFAULT_STRATEGY, "default"), (zlib.Z_FIXED, "fixed")):
                if sname == "fixed" and lvl != 6:
                    continue
                co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat)
                raw = co.compress(p) + co.flush()
                streams.append({
                    "name": "%s-l%d-%s" % (name, lvl, sname),
                    "raw_b64": base64.b64encode(raw).decode(),
                    "plain_len": len(p),
                    "plain_crc32": zlib.crc32(p),
                    "plain_adler32_rfc1950": zlib.adler32(p),
                    "plain_b64": base64.b64encode(p).decode() if len(p) <= 64 else None,
                    "gen": None if len(p) <= 64 else name,
                })
    json.dump({"streams": streams}, open(os.path.join(HERE, "zlib_streams.json"), "w"))

    kat = {
        "fox": "The quick brown fox jumps over the lazy dog",
        "crc32": {"": 0, "fox": 0x414FA339},
        "adler32": {"": 1, "fox": 0x5BDC0FDA},
        "trip": [
            {"s_b64": base64.b64encode(b"").decode(), "block": "fixed"},
            {"s_b64": base64.b64encode(b"a").decode(), "block": "fixed"},
            {"s_b64": base64.b64encode(b"hellohello").decode(), "block": "fixed"},
            {"s_b64": base64.b64encode(
                b"abcdefghijklmnopqrstuvwxyzzyxwvutsrqponmlkjihgfedcba").decode(),
             "block": "dynamic"},
            {"s_b64": base64.b64encode(bytes((i + 1) % 255 for i in range(256))).decode(),
             "block": "stored"},
        ],
        "limits": "Keep it to the limits.",
        "zip_docs": {
            "size": 56924,
            "members": [
                {"path": "zip-docs/rfc1951.txt", "data_start": 145, "compressed_size": 11132,
                 "decompressed_size": 36944, "crc32": 0xFB4F3400},
                {"path": "zip-docs/APPNOTE.TXT", "data_start": 11355, "compressed_size": 45288,
                 "decompressed_size": 174585, "crc32": 0x39B029C4},
            ],
        },
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)
    print("ok", len(z), len(streams))


if __name__ == "__main__":
    sys.exit(main())
