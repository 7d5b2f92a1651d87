#!/usr/bin/env python3
"""A small real-data corpus for the benchmark's corpus leg and the tests: the reference's two documents
(tests/golden/zip-docs.zip: APPNOTE.TXT, rfc1951.txt) and this repository's own files -- sources, documents,
fixtures and the built libraries (ELF + gfx950 code objects) -- concatenated in a fixed order and cut into
chunks of L bytes, every chunk a stream of its own.  The reference's procedure for real data is `unzip -t` over
a corpus archive (DEVEL.md:41-53, B0.ml:133-151, Silesia); there is no network here, so the corpus is what
travels with the repository: a few hundred distinct 64 KiB chunks of text, code and binaries instead of 4."""
import io
import os
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRS = ["zipc_amd", "oracle", "tests", "tools", "include", "bindings", "profiles"]
SKIP_DIRS = {"__pycache__", "build", "build_san", ".pytest_cache", "gpurun_out", ".git"}  # (build_san: the sanitizer builds of tests/test_sanitizers.py)
TOP = ["SURVEY.md", "DESIGN.md", "INTEGRATION.md", "README.md", "BASELINE.md", "bench.py", "__graft_entry__.py"]


def files():
    out = [os.path.join(ROOT, f) for f in TOP if os.path.exists(os.path.join(ROOT, f))]
    for d in DIRS:
        for base, dirs, names in os.walk(os.path.join(ROOT, d)):
            dirs[:] = sorted(x for x in dirs if x not in SKIP_DIRS)
            for n in sorted(names):
                if n.endswith((".pyc", ".o")) or "libzipc_hip_" in n or "_asan" in n or "_tsan" in n:  # (variant and sanitizer builds are not part of it)
                    continue
                out.append(os.path.join(base, n))
    return out


def blob():
    z = zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", "zip-docs.zip"))
    parts = [z.read("zip-docs/APPNOTE.TXT"), z.read("zip-docs/rfc1951.txt")]
    for f in files():
        try:
            with open(f, "rb") as h:
                parts.append(h.read())
        except OSError:
            pass
    return b"".join(parts)


def chunks(L=65536):
    b = blob()
    return [b[i:i + L] for i in range(0, len(b) - L + 1, L)]


if __name__ == "__main__":
    c = chunks()
    import zlib
    print(len(c), "chunks of 64 KiB;", "zlib -6 ratio %.3f" % (sum(len(zlib.compress(x, 6)) for x in c) / (65536.0 * len(c))))
