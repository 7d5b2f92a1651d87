"""A bounded run of tools/fuzz_gpu.py's randomized parity loop inside the -m gpu suite: deflate of
assorted streams at a random level (bytes and Adler-32 against the oracle), then inflate of the valid
streams, of damaged copies and of random block headers (status, bytes, checksum).  Fresh seeds every
day of the year; the long runs stay with the tool.  Every assertion names the seeds it ran with, and
ZIPC_TEST_DAY=<day of the year> runs the suite with another day's seeds: a red run can be repeated."""
import datetime
import importlib.util
import os
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _day():
    """what the seeds are made of: the day of the year, or ZIPC_TEST_DAY to repeat another day's run"""
    return int(os.environ.get("ZIPC_TEST_DAY", datetime.date.today().timetuple().tm_yday))


def test_randomized_parity_bounded(gpu_ctx, oracle):
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tools", "fuzz_gpu.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    fuzz.ctx = gpu_ctx
    day = _day()
    t0 = time.time()
    total = bad = 0
    lines = []
    for k in range(12):
        t, b = fuzz.run_seed(50000 + 100 * day + k, n_plain=60, n_headers=(50, 150), say=lambda *a: lines.append(" ".join(map(str, a))))
        total += t
        bad += b
        if time.time() - t0 > 25:  # about 20 s of GPU and oracle time
            break
    assert bad == 0, ("seeds %d + k, ZIPC_TEST_DAY=%d" % (50000 + 100 * day, day), lines[:10])
    assert total > 1000


@pytest.mark.parametrize("env", [
    {"ZIPC_HIP_MATCH_FORM": "2"},                                   # the byte-first walk on every tile (default: chosen per tile)
    {"ZIPC_HIP_MATCH_FORM": "1"},                                   # the first walk on every tile
    {"ZIPC_HIP_SLICES": "3", "ZIPC_HIP_SLICE_MIN": "1"},            # batches cut into slices on side queues
    {"ZIPC_HIP_MATCH_FORM": "2", "ZIPC_HIP_MATCH_TILES_PER_GROUP": "3", "ZIPC_HIP_SLICES": "2", "ZIPC_HIP_SLICE_MIN": "1"},
    {"ZIPC_HIP_PARSE_SEGMENTS": "1"},                               # lz_parse by a wave per segment whenever a stream has two
    {"ZIPC_HIP_PARSE_SEGMENTS": "1", "ZIPC_HIP_SLICES": "2", "ZIPC_HIP_SLICE_MIN": "1"},
    {"ZIPC_HIP_DEFLATE_GROUP_BYTES": "300000"},                     # a batch in groups of a few streams through one scratch (batches beyond 8 GiB)
    {"ZIPC_HIP_DEFLATE_GROUP_BYTES": "500000", "ZIPC_HIP_PARSE_SEGMENTS": "1"},
    {"ZIPC_HIP_CHAIN": "peel"},                                     # hash chains by the kernel that orders equal hashes itself (default: ordered LDS exchange)
    {"ZIPC_HIP_CHAIN": "peel", "ZIPC_HIP_PARSE_SEGMENTS": "0"},
    {"ZIPC_HIP_PARSE_SEGMENTS": "0"},                               # a wave per stream for parse and blocks (what 16 384 streams take)
], ids=["scan-walk", "first-walk", "slices", "scan-walk+groups+slices", "parse-segments", "parse-segments+slices",
        "batch-groups", "batch-groups+parse-segments", "chain-peel", "chain-peel+one-wave-forms", "one-wave-forms"])
def test_randomized_parity_under_overrides(env):
    """The same loop in a process of its own under the library's overrides (read once per process), so that
    the paths a 120-stream batch would not reach by itself are compared with the oracle too."""
    import subprocess
    import sys

    day = _day()
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), str(70000 + 10 * day), "2"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-600:]
    assert r.returncode == 0 and "FUZZ ok" in tail, ("tools/fuzz_gpu.py %d 2 under %s (ZIPC_TEST_DAY=%d)" % (70000 + 10 * day, env, day), tail)


@pytest.mark.parametrize("seg", ["default", "8192", "32768"])
def test_long_mixed_streams_bounded(seg):
    """tools/fuzz_long.py, a few seeds: long streams of mixed content (text, symbols, random bytes, runs of one byte,
    short periods with and without defects) through the many-wave deflate forms at several segment sizes, bytes and
    checksums against the oracle."""
    import subprocess
    import sys

    day = _day()
    e = dict(os.environ)
    if seg != "default":
        e["ZIPC_HIP_PARSE_SEG"] = seg
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_long.py"), str(9000 + 10 * day), "8"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-600:]
    assert r.returncode == 0 and "FUZZ ok" in tail, ("tools/fuzz_long.py %d 8, ZIPC_HIP_PARSE_SEG=%s (ZIPC_TEST_DAY=%d)" % (9000 + 10 * day, seg, day), tail)


@pytest.mark.parametrize("form", ["default", "following", "explorers-everywhere"])
def test_one_stream_inflate_by_blocks_bounded(form):
    """tools/fuzz_inflate_blocks.py, a few dozen streams: sources of mixed content, the reference's encoder and zlib
    with random levels, memory levels, strategies and flushes, some damaged, cut or given too small a limit, CRC-32 or
    Adler-32 -- status, bytes and checksum against the oracle, and most of them decoded by a wave per block.  Also with
    the form long streams take forced on these short ones (sources written down as what they copy, a wave per block),
    and with an explorer every KiB and few hops a resolve round."""
    import subprocess
    import sys

    day = _day()
    e = dict(os.environ)
    e["TRIALS"] = "40" if form == "default" else "25"
    e["SEED"] = str(500 + day)
    if form == "following":
        e["ZIPC_HIP_INFLATE_FOLLOW"] = "1"
    elif form == "explorers-everywhere":
        e["ZIPC_HIP_EXPLORE_STRIDE"] = "1024"
        e["ZIPC_HIP_RESOLVE_HOPS0"] = "3"
        e["ZIPC_HIP_RESOLVE_HOPS1"] = "5"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_inflate_blocks.py")], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-800:]
    assert r.returncode == 0 and "0 mismatches" in tail, ("tools/fuzz_inflate_blocks.py SEED=%s TRIALS=%s form %s (ZIPC_TEST_DAY=%d)" % (e["SEED"], e["TRIALS"], form, day), tail)
    went = int(tail.split(" went by blocks")[0].split(", ")[-1])
    assert went >= (15 if form == "default" else 8), tail


@pytest.mark.parametrize("form", ["default", "following", "few-hops"])
def test_calls_of_long_streams_inflate_by_blocks_side_by_side(form):
    """tools/fuzz_inflate_blocks.py BATCH=10: calls of 2 to 10 streams through zipc_hip_inflate_batch -- long ones, which go
    by blocks side by side (the kernels' grids over stream x block, inflate.hip), short ones and ones the block path
    refuses, which go by their one waves in the same call; damaged, cut and under-limited streams among them; CRC-32,
    Adler-32 or none.  Every stream's status, bytes and checksum against the oracle."""
    import subprocess
    import sys

    day = _day()
    e = dict(os.environ)
    e["TRIALS"] = "12"
    e["BATCH"] = "10"
    e["SEED"] = str(900 + day)
    if form == "following":
        e["ZIPC_HIP_INFLATE_FOLLOW"] = "1"
    elif form == "few-hops":
        e["ZIPC_HIP_EXPLORE_STRIDE"] = "2048"
        e["ZIPC_HIP_RESOLVE_HOPS0"] = "3"
        e["ZIPC_HIP_RESOLVE_HOPS1"] = "5"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_inflate_blocks.py")], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-800:]
    assert r.returncode == 0 and "0 mismatches" in tail, ("tools/fuzz_inflate_blocks.py SEED=%s TRIALS=12 BATCH=10 form %s (ZIPC_TEST_DAY=%d)" % (e["SEED"], form, day), tail)
    went = int(tail.split(" blocks went by blocks")[0].split(", ")[-1])
    assert went >= 200, tail


def test_chains_are_built_by_ordered_exchange(gpu_ctx):
    """The context's probe (deflate.hip xchg_order_probe) passes on gfx950, so the suite's deflate tests run
    lz_chain_xchg_kernel; the peel kernel is compared with the oracle by the overrides above."""
    assert gpu_ctx.lds_exchange_ordered()


@pytest.mark.parametrize("env", [{}, {"ZIPC_HIP_SLICES": "3", "ZIPC_HIP_SLICE_MIN": "1"}, {"ZIPC_HIP_HOST_CHUNKS": "6", "ZIPC_HIP_HOST_CHUNK_MIN": "8"}],
                         ids=["default", "slices", "many-sub-batches"])
def test_ragged_calls_of_the_many_stream_forms_bounded(env):
    """tools/fuzz_many_ragged.py, bounded: calls of 1 .. 2500 members with heavy-tailed lengths (bytes to tens of MiB) through
    zipc_hip_deflate_many, inflate_many and inflate_many_check against zlib and the oracle -- the shape of a directory of real
    files, which round 6's corpus run showed the suite did not reach (one 233 MB member among 491: 475 GB of parse scratch)."""
    import subprocess
    import sys

    day = _day()
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_many_ragged.py"), str(300 + day), "3"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-800:]
    assert r.returncode == 0 and "0 mismatches" in tail, ("tools/fuzz_many_ragged.py %d 3 under %s (ZIPC_TEST_DAY=%d)" % (300 + day, env, day), tail)
