"""The reference's corpus procedure (DEVEL.md:7-31, 41-53) on this box's own files, bounded: tools/corpus_box.py over a few
dozen of the ZIP-format files the image holds (.npz, .zip ...) and 48 MiB of this repository's built libraries and files -- every archive decoded on
the GPU and by Info-ZIP with the same verdict, recoded on the GPU and read back by Info-ZIP, sampled members of freshly zipped
binaries byte for byte against the oracle.  The full run (every archive below /usr and /opt, 8 GiB of binaries) is
profiles/r06_box_corpus.json."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_corpus_procedure_bounded(tmp_path):
    # (archives: the image's numpy test data; binaries and text: this repository's own built libraries and files -- a fresh box
    # pages the image in on first touch, so the bounded run stays off /opt/rocm/lib's gigabytes)
    roots = [r for r in ("/usr/lib/python3", "/usr/local/lib/python3.10/dist-packages/numpy", "/usr/local/lib/python3.10/dist-packages/scipy") if os.path.isdir(r)]
    roots.append(os.path.join(ROOT, "tests", "golden"))
    trees = [os.path.join(ROOT, d) for d in ("zipc_amd", "oracle", "tests", "tools")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "corpus_box.py"), "--roots", *roots, "--max-archives", "12",
                        "--max-archive-mib", "16", "--sniff-timeout-s", "120", "--tree", *trees, "--bytes-gib", "0.046875", "--per-archive-gib", "0.03125",
                        "--max-file-mib", "16", "--read-budget-s", "60", "--workdir", str(tmp_path)], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:] + r.stdout.decode()[-3000:]
    doc = json.loads(r.stdout.decode())
    if roots and doc["sniff"]["found"]:
        a = doc["archives"]
        assert a["checked"] > 0 and a["different_verdict"] == [] and a["recode_t_failed"] == []
        assert a["recode_t_ok"] == a["checked"]
    t = doc["tree"]
    assert t["members"] > 0 and t["sampled"] > 0 and t["sampled"] == t["sampled_equal_oracle"]
    assert all(x["gpu_unzip_t"]["exit"] == 0 and x.get("infozip_tq", {"exit": 0})["exit"] == 0 for x in t["archives"])
