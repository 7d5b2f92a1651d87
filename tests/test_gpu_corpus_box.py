"""The reference's corpus procedure (DEVEL.md:7-31, 41-53) on this box's own files, bounded: tools/corpus_box.py over a few
dozen of the ZIP-format files the image holds (wheels, jars, .npz ...) and 96 MiB of /opt/rocm/lib -- every archive decoded on
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


def test_corpus_procedure_bounded(gpu_ctx, tmp_path):
    roots = [r for r in ("/usr/lib/python3", "/usr/local/lib", "/opt/rocm/share") if os.path.isdir(r)]
    tree = "/opt/rocm/lib" if os.path.isdir("/opt/rocm/lib") else os.path.join(ROOT, "zipc_amd")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "corpus_box.py"), "--roots", *roots, "--max-archives", "40",
                        "--max-archive-mib", "48", "--tree", tree, "--bytes-gib", "0.09375", "--per-archive-gib", "0.0625",
                        "--workdir", str(tmp_path)], capture_output=True, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-2000:] + r.stdout.decode()[-3000:]
    doc = json.loads(r.stdout.decode())
    if roots and doc["sniff"]["found"]:
        a = doc["archives"]
        assert a["checked"] > 0 and a["different_verdict"] == [] and a["recode_t_failed"] == []
        assert a["recode_t_ok"] == a["checked"]
    t = doc["tree"]
    assert t["members"] > 0 and t["sampled"] > 0 and t["sampled"] == t["sampled_equal_oracle"]
    assert all(x["gpu_unzip_t"]["exit"] == 0 and x.get("infozip_tq", {"exit": 0})["exit"] == 0 for x in t["archives"])
