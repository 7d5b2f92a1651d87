#!/usr/bin/env python3
"""The reference's corpus procedure (DEVEL.md:7-31, 41-53; B0.ml:54, 72-86) on whatever box this runs on, through
zipc_amd/bin/zipc-hip (zipc_amd/host/zipc_tool.cpp, after test/zipc_tool.ml):

  (i)  `zipc-hip sniff -0 -P -r ROOTS` lists every file below ROOTS that begins with a ZIP magic number (.zip, .whl, .jar,
       .npz, .egg ...); every archive found goes through
           zipc-hip unzip --skip -t ARCHIVE                      (the GPU decodes and CRC-checks every member)
           unzip -P '' -q -q -t ARCHIVE                          (Info-ZIP does the same on the host)
       and the two must give the same verdict; then
           zipc-hip recode --deflate --level default -t --check-cmd="unzip -P '' -q -q -t" ARCHIVE
       -- every member inflated and deflated again on the GPU, the recoded archive read back by Info-ZIP.
  (ii) files below a directory of real binaries (/opt/rocm/lib: ELF, gfx code objects) are zipped on the GPU at `Default into
       archives of about a GiB of source each until BYTES of source are in (the format has no ZIP64: an archive stays under
       4 GiB and 65535 members), a sample of members is held byte for byte against the oracle, and `zipc-hip unzip -t` is timed
       beside `unzip -tq` on the same files (the reference's time-inflate: DEVEL.md:41-53).

One JSON document on stdout (profiles/rNN_box_corpus.json is a run of it on an MI355X box).  The oracle is imported here as
the checker of sampled members only.
  python3 tools/corpus_box.py [--roots /usr /opt] [--max-archives N] [--max-archive-mib M] [--tree /opt/rocm/lib] [--bytes-gib G]
"""
import argparse
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "zipc_amd", "bin", "zipc-hip")
UNZIP_CHECK = "unzip -P '' -q -q -t"


def run(cmd, **kw):
    t = time.perf_counter()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    return r.returncode, r.stdout, r.stderr, time.perf_counter() - t


def sniff(roots, timeout_s):
    """zipc-hip sniff -0 -P -r ROOTS -> paths (DEVEL.md:9); a walk that outlasts timeout_s is cut and what it found so far is used"""
    t = time.perf_counter()
    p = subprocess.Popen([TOOL, "sniff", "-0", "-P", "-r"] + list(roots), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        out, _ = p.communicate(timeout=timeout_s)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        p.kill()
        out, _ = p.communicate()
        rc = -1
    return [q.decode("utf-8", "surrogateescape") for q in out.split(b"\0")[:None if rc != -1 else -1] if q], rc, time.perf_counter() - t


def verdict_ours(rc):
    return {0: "ok", 2: "corrupted", 3: "unsupported"}.get(rc, "error %d" % rc)


def verdict_infozip(rc):
    # unzip(1): 0 fine, 1 warnings (an empty archive among them), 2/3 errors in the archive, 9 nothing found, 81 unsupported
    # compression or encryption, 82 bad password
    return {0: "ok", 1: "ok", 2: "corrupted", 3: "corrupted", 51: "corrupted", 81: "unsupported", 82: "unsupported"}.get(rc, "error %d" % rc)


def check_archives(paths, max_archives, max_bytes):
    """part (i): every archive through the GPU and through Info-ZIP; the same verdicts; recoded and read back"""
    have_unzip = shutil.which("unzip") is not None
    rows, skipped = [], 0
    for p in paths:
        try:
            size = os.path.getsize(p)
        except OSError:
            continue
        if size > max_bytes or size < 22:
            skipped += 1
            continue
        if len(rows) >= max_archives:
            skipped += 1
            continue
        row = {"path": p, "bytes": size}
        rc, out, err, dt = run([TOOL, "unzip", "--skip", "-t", p])
        row["unzip_t"] = {"exit": rc, "verdict": verdict_ours(rc), "s": round(dt, 3)}
        if rc not in (0, 2, 3):
            row["unzip_t"]["stderr"] = err.decode("utf-8", "replace")[-300:]
        if have_unzip:
            rc2, _, err2, dt2 = run("%s %s" % (UNZIP_CHECK, shell_quote(p)), shell=True)
            row["infozip"] = {"exit": rc2, "verdict": verdict_infozip(rc2), "s": round(dt2, 3)}
            # --skip passes over what neither tool decodes: Info-ZIP's "unsupported" is our "ok" then
            ours = row["unzip_t"]["verdict"]
            theirs = "ok" if row["infozip"]["verdict"] == "unsupported" else row["infozip"]["verdict"]
            row["same_verdict"] = ours == theirs
        cmd = [TOOL, "recode", "--deflate", "--level", "default", "-t"] + (["--check-cmd=" + UNZIP_CHECK] if have_unzip else []) + [p]
        rc3, _, err3, dt3 = run(cmd)
        row["recode_t"] = {"exit": rc3, "s": round(dt3, 3)}
        if rc3 != 0:
            row["recode_t"]["stderr"] = err3.decode("utf-8", "replace")[-300:]
        rows.append(row)
    return rows, skipped


def shell_quote(p):
    return "'" + p.replace("'", "'\\''") + "'"


def tree_files(trees, want_bytes, max_file, read_budget_s):
    """Regular files below `trees` (sorted walk) until want_bytes are in -- each READ once here, so that what is timed later
    is not the image's pages coming in from wherever a fresh box keeps them (the first read of 1 GB of /opt/rocm/lib took
    minutes on one box, seconds on the next); the walk stops when the reads have taken read_budget_s."""
    out, total, t_read = [], 0, 0.0
    for tree in trees:
        for base, dirs, names in os.walk(tree):
            dirs.sort()
            for n in sorted(names):
                p = os.path.join(base, n)
                try:
                    st = os.lstat(p)
                    if not os.path.isfile(p) or os.path.islink(p) or st.st_size == 0 or st.st_size > max_file:
                        continue
                    t = time.perf_counter()
                    with open(p, "rb") as f:
                        while f.read(1 << 24):
                            pass
                    t_read += time.perf_counter() - t
                except OSError:
                    continue
                out.append((p, st.st_size))
                total += st.st_size
                if total >= want_bytes or t_read > read_budget_s:
                    return out, total, t_read
    return out, total, t_read


def member_compressed_bytes(zf, info, blob_path):
    """the raw compressed bytes of a member: behind its local file header"""
    with open(blob_path, "rb") as f:
        f.seek(info.header_offset)
        h = f.read(30)
        sig, _, _, _, _, _, _, _, _, nlen, xlen = struct.unpack("<IHHHHHIIIHH", h)
        assert sig == 0x04034B50
        f.seek(info.header_offset + 30 + nlen + xlen)
        return f.read(info.compress_size)


def zip_tree(trees, want_bytes, per_archive, workdir, max_file, read_budget_s, sample_per_archive=6, sample_max=8 << 20):
    """part (ii): zip on the GPU, sample against the oracle, time unzip -t both ways"""
    sys.path.insert(0, ROOT)
    import oracle

    files, total, t_read = tree_files(trees, want_bytes, max_file, read_budget_s)
    groups, cur, cur_b = [], [], 0
    for p, sz in files:
        if cur and (cur_b + sz > per_archive or len(cur) >= 60000):
            groups.append((cur, cur_b))
            cur, cur_b = [], 0
        cur.append(p)
        cur_b += sz
    if cur:
        groups.append((cur, cur_b))
    have_unzip = shutil.which("unzip") is not None
    rows = []
    sums = {"first_read_s": round(t_read, 2), "source_bytes": 0, "archive_bytes": 0, "members": 0, "zip_s": 0.0, "gpu_unzip_t_s": 0.0, "infozip_tq_s": 0.0, "sampled": 0, "sampled_equal_oracle": 0}
    for k, (paths, nbytes) in enumerate(groups):
        arc = os.path.join(workdir, "tree%02d.zip" % k)
        rc, _, err, dt_zip = run([TOOL, "zip", "--level", "default", "-o", arc] + paths)
        row = {"archive": os.path.basename(arc), "members": len(paths), "source_bytes": nbytes, "zip": {"exit": rc, "s": round(dt_zip, 3)}}
        if rc != 0:
            row["zip"]["stderr"] = err.decode("utf-8", "replace")[-300:]
            rows.append(row)
            continue
        row["archive_bytes"] = os.path.getsize(arc)
        # a sample of members byte for byte against the oracle (level `Default, CRC-32)
        with zipfile.ZipFile(arc) as zf:
            infos = [i for i in zf.infolist() if not i.is_dir() and 0 < i.file_size <= sample_max]
            step = max(1, len(infos) // sample_per_archive)
            equal = 0
            picked = infos[::step][:sample_per_archive]
            by_name = {p.lstrip("/"): p for p in paths}  # (the tool names a member by its sanitized path)
            for i in picked:
                plain = open(by_name[i.filename], "rb").read()
                st, c0, crc0 = oracle.deflate(plain, level=2, crc_op=oracle.CRC_CRC32)
                got = member_compressed_bytes(zf, i, arc)
                equal += int(st == 0 and got == c0 and i.CRC == crc0 and i.compress_type == zipfile.ZIP_DEFLATED)
            row["sampled"] = len(picked)
            row["sampled_equal_oracle"] = equal
        rc1, out1, err1, dt1 = run([TOOL, "unzip", "-t", arc])
        row["gpu_unzip_t"] = {"exit": rc1, "s": round(dt1, 3)}
        if have_unzip:
            rc2, _, _, dt2 = run(["unzip", "-tq", arc])
            row["infozip_tq"] = {"exit": rc2, "s": round(dt2, 3)}
            sums["infozip_tq_s"] += dt2
        rows.append(row)
        sums["source_bytes"] += nbytes
        sums["archive_bytes"] += row["archive_bytes"]
        sums["members"] += len(paths)
        sums["zip_s"] += dt_zip
        sums["gpu_unzip_t_s"] += dt1
        sums["sampled"] += row["sampled"]
        sums["sampled_equal_oracle"] += row["sampled_equal_oracle"]
        os.unlink(arc)
    gib = sums["source_bytes"] / 2.0 ** 30
    sums["gib"] = gib
    if sums["zip_s"]:
        sums["zip_gib_s_wall"] = gib / sums["zip_s"]
    if sums["gpu_unzip_t_s"]:
        sums["gpu_unzip_t_gib_s_wall"] = gib / sums["gpu_unzip_t_s"]
    if sums["infozip_tq_s"]:
        sums["infozip_tq_gib_s_wall"] = gib / sums["infozip_tq_s"]
        sums["gpu_over_infozip"] = sums["infozip_tq_s"] / sums["gpu_unzip_t_s"] if sums["gpu_unzip_t_s"] else None
    sums["is"] = ("wall time of whole command lines, process start, reading the files and making the GPU context included: `zipc-hip zip --level default`, "
                  "`zipc-hip unzip -t` and Info-ZIP `unzip -tq` over the same archives (DEVEL.md:41-53)")
    return rows, sums


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--roots", nargs="*", default=["/usr", "/opt"])
    ap.add_argument("--max-archives", type=int, default=400)
    ap.add_argument("--max-archive-mib", type=int, default=512)
    ap.add_argument("--tree", nargs="*", default=["/opt/rocm/lib"])
    ap.add_argument("--max-file-mib", type=int, default=256)
    ap.add_argument("--read-budget-s", type=float, default=900.0)
    ap.add_argument("--sniff-timeout-s", type=float, default=900.0)
    ap.add_argument("--bytes-gib", type=float, default=8.0)
    ap.add_argument("--per-archive-gib", type=float, default=1.0)
    ap.add_argument("--workdir", default=None)
    a = ap.parse_args()
    assert os.path.exists(TOOL), "run `make -C zipc_amd/host`"
    doc = {"tool": "zipc_amd/bin/zipc-hip", "procedure": "DEVEL.md:7-31 (sniff, unzip --skip -t, recode --deflate -t --check-cmd) and :41-53 (unzip -t timed beside Info-ZIP)"}
    roots = [r for r in a.roots if os.path.isdir(r)]
    if roots:
        paths, rc, dt = sniff(roots, a.sniff_timeout_s)
        rows, skipped = check_archives(paths, a.max_archives, a.max_archive_mib << 20)
        by_ext = {}
        for r in rows:
            by_ext[os.path.splitext(r["path"])[1].lower() or "(none)"] = by_ext.get(os.path.splitext(r["path"])[1].lower() or "(none)", 0) + 1
        doc["sniff"] = {"roots": roots, "found": len(paths), "exit": rc, "s": round(dt, 2), "checked": len(rows), "passed_over": skipped, "by_extension": by_ext}
        doc["archives"] = {
            "checked": len(rows),
            "bytes": sum(r["bytes"] for r in rows),
            "gpu_unzip_t": {v: sum(1 for r in rows if r["unzip_t"]["verdict"] == v) for v in sorted({r["unzip_t"]["verdict"] for r in rows})},
            "infozip": {v: sum(1 for r in rows if r.get("infozip", {}).get("verdict") == v) for v in sorted({r.get("infozip", {}).get("verdict", "absent") for r in rows})},
            "same_verdict": sum(1 for r in rows if r.get("same_verdict")),
            "different_verdict": [r for r in rows if r.get("same_verdict") is False],
            "recode_t_ok": sum(1 for r in rows if r["recode_t"]["exit"] == 0),
            "recode_t_failed": [r for r in rows if r["recode_t"]["exit"] != 0],
            "gpu_unzip_t_s": round(sum(r["unzip_t"]["s"] for r in rows), 2),
            "infozip_s": round(sum(r.get("infozip", {}).get("s", 0) for r in rows), 2),
            "recode_t_s": round(sum(r["recode_t"]["s"] for r in rows), 2),
        }
    trees = [t for t in a.tree if os.path.isdir(t)]
    if trees and a.bytes_gib > 0:
        work = a.workdir or tempfile.mkdtemp(prefix="zipc_box_")
        try:
            rows, sums = zip_tree(trees, int(a.bytes_gib * 2 ** 30), int(a.per_archive_gib * 2 ** 30), work, a.max_file_mib << 20, a.read_budget_s)
            doc["tree"] = {"roots": trees, "archives": rows, **sums}
        finally:
            if not a.workdir:
                shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(doc, indent=1))
    bad = doc.get("archives", {}).get("different_verdict") or doc.get("archives", {}).get("recode_t_failed")
    t = doc.get("tree", {})
    bad = bad or (t and t.get("sampled") != t.get("sampled_equal_oracle"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
