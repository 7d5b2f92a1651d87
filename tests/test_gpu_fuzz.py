"""A bounded run of tools/fuzz_gpu.py's randomized parity loop inside the -m gpu suite: deflate of
assorted streams at a random level (bytes and Adler-32 against the oracle), then inflate of the valid
streams, of damaged copies and of random block headers (status, bytes, checksum).  Fresh seeds every
day of the year; the long runs stay with the tool."""
import datetime
import importlib.util
import os
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomized_parity_bounded(gpu_ctx, oracle):
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tools", "fuzz_gpu.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    fuzz.ctx = gpu_ctx
    day = datetime.date.today().timetuple().tm_yday
    t0 = time.time()
    total = bad = 0
    lines = []
    for k in range(12):
        t, b = fuzz.run_seed(50000 + 100 * day + k, n_plain=60, n_headers=(50, 150), say=lambda *a: lines.append(" ".join(map(str, a))))
        total += t
        bad += b
        if time.time() - t0 > 25:  # about 20 s of GPU and oracle time
            break
    assert bad == 0, lines[:10]
    assert total > 1000
