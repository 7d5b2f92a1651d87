#!/usr/bin/env python3
"""The tables of DESIGN.md section 6 from the round's committed profiles (profiles/<TAG>_*), so that every number in the
text is one a reader can find in a file:   python3 tools/design_tables.py r06 [r05]   (markdown on stdout; the second
tag adds the round before for comparison)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(tag, name):
    f = os.path.join(P, "%s_%s" % (tag, name))
    if not os.path.exists(f):
        return None
    if name.endswith(".jsonl"):
        return [json.loads(l) for l in open(f) if l.strip().startswith("{")]
    return json.load(open(f))


def f1(x, nd=1):
    return "-" if x is None else ("%." + str(nd) + "f") % x


def kernel_stats(tag):
    f = os.path.join(P, "%s_kernel_stats.csv" % tag)
    out = {}
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            name = r["Name"].split("(")[0].replace("zd::", "").replace("_kernel", "").replace("_window", "")
            out[name] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["Percentage"]))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    old = sys.argv[2] if len(sys.argv) > 2 else None
    b, a, bo = load(tag, "bench.json"), load(tag, "bench_alone.json"), load(old, "bench.json") if old else None
    if b:
        print("### C2 (bench.py, %s)\n" % tag)
        print("value %.2f GiB/s, %.2f ms per step; deflate alone %.1f, inflate alone %.1f GiB/s" % (b["value"], b["ms_per_step"], b["deflate_gib_s"], b["inflate_gib_s"]))
        r = b["roofline"]
        print("dominant kernel `%s`: %.1f GB/s of %d = frac %.4f (launch %.3f ms over %.0f MB of the path's bytes)" % (
            r["kernel"], r["achieved"], r["peak"], r["frac"], r["launch_ms"], r["algorithmic_bytes"] / 1e6))
        for k in ("deflate_frac", "inflate_frac", "deflate_gb_s", "inflate_gb_s", "deflate_traffic_over_algorithmic_raw", "deflate_traffic_over_algorithmic_corrected",
                  "inflate_traffic_over_algorithmic_raw", "inflate_traffic_over_algorithmic_corrected", "traffic_over_algorithmic_raw", "traffic_over_algorithmic_corrected",
                  "issue_bound_frac"):
            if k in r:
                print("  roofline.%s = %.4f" % (k, r[k]))
        ks = kernel_stats(tag)
        print("\n| kernel | ms per step (HIP events, both slices) | alone: ms per launch over the whole batch | rocprofv3: calls x avg ms | %s |" % (old or ""))
        print("|---|---|---|---|---|")
        al = (a or {}).get("roofline", {})
        alone = {al.get("kernel"): al.get("alone", {}).get("launch_ms")}
        for k, v in (al.get("others") or {}).items():
            alone[k] = (v.get("alone") or {}).get("launch_ms")
        for k, v in sorted(b["kernels_ms_per_step"].items(), key=lambda x: -x[1]):
            rp = ks.get(k) or ks.get(k + "_xchg") or ks.get(k + "_streams")
            print("| `%s` | %.2f | %s | %s | %s |" % (k, v, f1(alone.get(k), 2), ("%d x %.3f" % (rp[0], rp[1])) if rp else "-",
                                                 f1(bo["kernels_ms_per_step"].get(k), 2) if bo else ""))
        print("\n| leg (device-resident unless said) | deflate GiB/s | inflate GiB/s | %s |" % (old or ""))
        print("|---|---|---|---|")

        def leg(name, d, o=None):
            if not isinstance(d, dict) or "deflate" not in d:
                return
            print("| %s | %s | %s | %s |" % (name, f1(d.get("deflate")), f1(d.get("inflate")), ("%s / %s" % (f1(o.get("deflate")), f1(o.get("inflate")))) if isinstance(o, dict) and "deflate" in o else ""))
        g = lambda d, *ks_: (lambda x: x)(__import__("functools").reduce(lambda acc, k: (acc or {}).get(k) if isinstance(acc, dict) else None, ks_, d))
        leg("real text, 16 384 x 64 KiB, `Default", b.get("text_gib_s"), (bo or {}).get("text_gib_s"))
        leg("`Best, C2 symbols (4096)", g(b, "best_gib_s", "c2"), g(bo, "best_gib_s", "c2") if bo else None)
        leg("`Best, text (2048)", g(b, "best_gib_s", "text"), g(bo, "best_gib_s", "text") if bo else None)
        leg("corpus, 4096 x 64 KiB, `Default", g(b, "corpus_gib_s", "default"), g(bo, "corpus_gib_s", "default") if bo else None)
        leg("corpus, `Best (2048)", g(b, "corpus_gib_s", "best"), g(bo, "corpus_gib_s", "best") if bo else None)
        leg("C4's shape, 4096 x 1 MiB", b.get("c4_leg"), (bo or {}).get("c4_leg"))
        leg("64 x 1 MiB of text in one call", b.get("long_members"), (bo or {}).get("long_members"))
        leg("host forms, 4096 x 64 KiB, PCIe-inclusive", b.get("e2e_gib_s"), (bo or {}).get("e2e_gib_s"))
        one = b.get("one_stream_ms") or {}
        for k, v in one.items():
            if isinstance(v, dict):
                print("| ONE stream per call: %s | %.2f ms | %.2f ms | |" % (k, v["deflate_ms"], v["inflate_ms"]))
        cb = b.get("cpu_baseline") or {}
        if cb:
            print("\ncpu_baseline (%s): %.4f GiB/s on %d core (deflate %.4f, inflate %.4f); all %s threads: %s GiB/s" % (
                cb.get("kind"), cb["value"], cb["cores"], cb.get("deflate_gib_s", 0), cb.get("inflate_gib_s", 0), (cb.get("nproc") or {}).get("cores"),
                f1((cb.get("nproc") or {}).get("value"), 2)))
    c4 = load(tag, "bench_c4.json")
    if c4:
        print("\n### `--config c4` (%s): %.1f GiB/s deflate, %.1f ms per step" % (tag, c4["value"], c4["ms_per_step"]))
        r = c4["roofline"]
        print("dominant `%s` frac %.4f; " % (r["kernel"], r["frac"]) + ", ".join("%s %.4g" % (k, r[k]) for k in ("deflate_gib_s", "deflate_frac", "deflate_traffic_over_algorithmic_raw", "deflate_traffic_over_algorithmic_corrected", "issue_bound_frac") if k in r))
        print({k: round(v, 2) for k, v in c4["kernels_ms_per_step"].items()})
    cf = load(tag, "configs.jsonl")
    if cf:
        print("\n### other configs (%s)" % tag)
        for r in cf:
            print("* %s: %s" % (r["config"][:70], ", ".join("%s %.1f" % (k, r[k]) for k in ("gib_s", "hbm_gb_s", "deflate_gib_s", "inflate_gib_s", "inflate_hbm_gb_s") if k in r)))
    hf = load(tag, "host_forms.json")
    if hf:
        print("\nhost forms 16 384 x 64 KiB: deflate_many %.1f, inflate_many %.1f GiB/s" % (hf["deflate_many_gib_s"], hf["inflate_many_gib_s"]))
    bx = load(tag, "box_corpus.json")
    if bx:
        print("\n### corpus procedure on the box (%s_box_corpus.json)" % tag)
        if "sniff" in bx:
            print("sniff:", bx["sniff"])
            print({k: v for k, v in bx["archives"].items() if not isinstance(v, list)})
        if "tree" in bx:
            print({k: v for k, v in bx["tree"].items() if k not in ("archives", "is")})


if __name__ == "__main__":
    main()
