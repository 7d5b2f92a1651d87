#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric on its config C2 (default), or C4 (--config c4).

A "step" is one pass of the hot path over one batch of synthetic input:
  c2  deflate (level `Default, CRC-32 of the source by a separate pass) of 16 384 x 64 KiB
      streams of i.i.d. 4-bit symbols (1 GiB), then inflate (CRC-32 of the output likewise) of
      the 16 384 compressed streams, both through the device-resident batch forms of
      include/zipc_hip.h.  value = uncompressed GiB round-tripped per second.
  c4  crc_32_and_deflate of the members of a ZIP archive, 8192 x 1 MiB of i.i.d. 3-bit symbols,
      member-sharded: rank r takes a contiguous range of the members in the order Zipc writes
      them (src/zipc.ml:575-581), deflates it into its own arena, and the ranks all-gather one
      16-byte record per member (size, CRC-32, arena offset; RCCL) from which every rank can lay
      out the archive.  value = uncompressed GiB deflated per second, all ranks together.
Inputs are generated on the device and stay in HBM.  With N > 1 ranks the streams are
member-sharded, one process per GPU, no data-path collective (c2: weak scaling, every rank
runs the full 1 GiB config on its own streams; c4: strong scaling, the archive is fixed).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process the launcher:
before anything touches torch or the GPU it starts N workers (this script again, one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays rank 0's JSON
line and exits non-zero if any worker failed.  Under torch.distributed.run the environment's
WORLD_SIZE must equal --gpus.

Extra objects in the JSON line: "roofline" (dominant kernel: the algorithmic bytes of the streams ONE launch covers --
a step is two slices on two queues, a launch half the batch -- over its HIP-event duration, against the 8 TB/s HBM peak;
its "others" lists the step's other kernels the same way, "deflate_pipeline" / "inflate" the directions over their wall
time; --alone-pass adds every kernel launched alone over the whole batch; SCALARS inside it -- and again inside "config" --
carry the line's claims for a record that keeps nothing else: deflate_gib_s / inflate_gib_s, deflate_frac / inflate_frac (N + C
over the direction's wall time over 8 TB/s), *_traffic_over_algorithmic_raw / _corrected and issue_bound_frac, the last two kinds
only from counter passes of THIS workload: profiles/rNN_hbm_traffic[_c4].json, rNN_sq_counters[_c4].json), "cpu_baseline" (the oracle's C port timed
on this host on a bounded sample: 1 thread and all host threads, the headline's workload and the legs'; with the digests
of sampled streams the legs check their bytes against; rank 0 / N=1 only), "kernels_ms_per_step", "inflate_gib_s" /
"deflate_gib_s", and the untimed legs, each with its parity sample and its share of the roofline: "e2e_gib_s"
(PCIe-inclusive host forms), "text_gib_s", "best_gib_s", "corpus_gib_s", "one_stream_ms", "long_members" (a call of 64
members of 1 MiB), "c4_leg".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "GiB/s deflate+inflate on 1 GiB synthetic; bit-exact vs Zipc_deflate"
LEVELS = ["none", "fast", "default", "best"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE or 1")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["c2", "c4"], default="c2")
    ap.add_argument("--streams", type=int, default=None, help="c2: streams per rank (16384); c4: members of the archive (8192)")
    ap.add_argument("--stream-len", type=int, default=None, help="c2: 65536; c4: 1048576")
    ap.add_argument("--bits", type=int, default=None, help="entropy bits per byte (c2: 4, c4: 3)")
    ap.add_argument("--level", type=int, default=2, help="0 none, 1 fast, 2 default, 3 best")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the untimed legs (host forms, real text)")
    ap.add_argument("--no-archive-check", action="store_true",
                    help="c4: skip the untimed check of the whole archive (unzip -tq, zipfile, N-rank == 1-rank): counter passes only")
    ap.add_argument("--alone-pass", action="store_true",
                    help="a second profiled pass with ONE slice: every kernel launched over the whole batch with nothing beside it "
                         "(roofline.*.alone).  Off by default so that a rocprofv3 summary of the default command holds only the "
                         "launches the timed region makes")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)  # the cpu_baseline leg's own process
    return ap.parse_args(argv)


# ---- the launcher (no torch, no GPU in this process) ------------------------------------------------

def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_workers(n, argv):
    """Start n workers (this script, one per GPU) and relay rank 0's JSON line.  Returns the exit code.
    The workers are polled: when one exits non-zero the others -- which would sit in the rendezvous or
    in a barrier until the RCCL timeout, holding their GPUs -- are terminated and the launcher fails at once."""
    import tempfile
    import threading

    port = free_port()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # fresh child processes (nothing in this process has touched a GPU); a group of their own, so that a
        # worker's own children (the cpu_baseline leg) end with it
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, start_new_session=True))
    bad = []
    deadline = time.time() + float(os.environ.get("ZIPC_BENCH_LAUNCH_TIMEOUT_S", "3600"))
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            bad = [(-1, "timeout")]
            break
        time.sleep(0.1)
    if bad:
        import signal

        for p in procs:  # the exact process groups started above
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()
        sys.stderr.write("bench.py: workers failed (rank, exit code): %s; the others were stopped\n" % bad)
        return 1
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return 0


# ---- pieces of the JSON line -------------------------------------------------------------------------

DEFLATE_KERNELS = ("deflate_offsets", "lz_chain", "lz_match", "lz_parse", "deflate_emit", "deflate_stored",
                   # few long streams (at most 2048 of 32 KiB and more, 4096 of 512 KiB and more): the parse by segments, the blocks by a wave each
                   "lz_parse_spec", "lz_parse_meet", "lz_parse_stitch", "lz_parse_gather", "deflate_plan", "deflate_counts",
                   "deflate_codelen", "deflate_scan", "deflate_bits", "deflate_pack", "deflate_seal")


def algorithmic_bytes(kernel, N, C):
    """the MINIMUM HBM bytes of the path a kernel belongs to, for N uncompressed and C
    compressed bytes per step -- deflate N + C (every kernel of the deflate pipeline is priced against the
    pipeline's minimum: what it moves beyond that is this design's choice, see design_bytes), inflate C + N,
    a CRC-32 pass N."""
    if kernel in DEFLATE_KERNELS:
        return N + C
    return {"inflate_batch": C + N, "crc32_segments": N}.get(kernel, 0)


def design_bytes(kernel, N, C):
    """What one launch of `kernel` moves at minimum IN THIS DESIGN (its own inputs and outputs, intermediates
    included: 2-byte chain links, 4-byte match records, 4-byte symbols; DESIGN.md 'Kernels')."""
    return {
        "inflate_batch": C + N,            # read compressed, write plain
        "crc32_segments": N,               # one pass over the checked bytes
        "lz_chain": N + 2 * N,             # read source, write 2-byte links
        "lz_match": N + 2 * N + 4 * N,     # source + links in, 4-byte match records out (round 5; the second table only where K/4 differs)
        "lz_parse": 4 * N + N + 4 * N,     # match records + literals in, <= 4 B/symbol out
        "deflate_emit": 4 * N + C,         # symbols in, compressed out
        "lz_parse_spec": 4 * N + N + 4 * N,    # as lz_parse, symbols to the segments' buffers
        "lz_parse_gather": 4 * N + 4 * N,      # ... and from there to the stream's symbol array
        "deflate_plan": 4 * N,                 # symbols in (histograms), a 3 KiB record per block out
        "deflate_pack": 4 * N + C,             # symbols in, compressed out
    }.get(kernel, 0)


# Which workload's counter passes the line may quote: "c2" (profiles/rNN_hbm_traffic.json, rNN_sq_counters.json: passes of
# the default command), "c4" (rNN_hbm_traffic_c4.json, rNN_sq_counters_c4.json: passes of --config c4) or None -- a shape
# of the caller's own has no counters of its own, and one workload's bytes over another's time are not evidence.
WORKLOAD = None


def latest_profile(kind):
    """the newest committed counter file of `kind` ("hbm_traffic" | "sq_counters") for WORKLOAD, or None"""
    import glob

    if WORKLOAD is None:
        return None
    suffix = "" if WORKLOAD == "c2" else "_" + WORKLOAD
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s%s.json" % (kind, suffix))))
    return files[-1] if files else None


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of THIS workload (profiles/rNN_hbm_traffic[_c4].json,
    made by tools/pmc_report.py --hbm-json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command), or None."""
    f = latest_profile("hbm_traffic")
    if not f:
        return None, None
    try:
        k = json.load(open(f))["kernels"]
        # (the timers call both chain kernels "lz_chain"; the counters know them by their own names, and since round 6 the
        # ordering kernel also runs as the guard of the other: the step's chain kernel is the exchange one where it ran)
        name = "lz_chain_xchg" if kernel == "lz_chain" and "lz_chain_xchg" in k else kernel
        return k[name], os.path.basename(f)
    except Exception:
        return None, None


SHADER_CLOCK_HZ = 2.38e9  # s_memtime against s_memrealtime under load, profiles/r01_match_phases.txt


VALU_CLOCKS = 4.0  # clocks of its SIMD per wave64 vector instruction of these kernels' mix (tools/probes/valu_costs.hip, below)


def issue_bound(kernel, launch_ms, share=None):
    """What instruction issue allows for `kernel`, from the committed SQ counter pass (profiles/*sq_counters.json,
    tools/exp_sq_counters.sh).  Round 5 measured the vector side on the hardware (tools/probes/valu_costs.hip, four
    waves per SIMD): v_add / sub / and / or / xor / lshrrev_b32 and v_fma_f32 take 2.4-2.7 clocks of their SIMD per
    wave64 instruction -- the guide's SIMD-32 figure -- but everything else these kernels are made of (v_cmp,
    v_cndmask, v_lshlrev, v_lshl_or, v_and_or, v_bfe, v_min / max, v_alignbit, v_ffbl, v_mbcnt, DPP moves, v_readlane)
    takes 4.2-4.4, and a scalar instruction between vector ones about 2 of the same SIMD's issue clocks.  So the
    vector side is priced at VALU_CLOCKS = 4 again (rounds 3 and 4 priced it at 2 and called the kernels scalar-bound:
    they are vector-issue bound -- lz_match alone: 2.56 G vector instructions x 4 / 1024 SIMDs / clock = its time).
    share: the part of the batch the priced launch covers; the counters are per launch of the counter run
    ("launch_share" in the file, 0.5: two slices) and are scaled to it."""
    f = latest_profile("sq_counters")
    if not f:
        return None
    try:
        d = json.load(open(f))
        k = d["kernels"]["lz_chain_xchg" if kernel == "lz_chain" and "lz_chain_xchg" in d["kernels"] else kernel]
        scale = (share / d.get("launch_share", 0.5)) if share else 1.0
        vec = k["SQ_INSTS_VALU"] * scale * VALU_CLOCKS / (1024 * SHADER_CLOCK_HZ) * 1e3
        sca = (k["SQ_INSTS_SALU"] + k["SQ_INSTS_BRANCH"] + k.get("SQ_INSTS_SMEM", 0)) * scale / (256 * SHADER_CLOCK_HZ) * 1e3
        bound = max(vec, sca)
        return {"vector_ms": vec, "scalar_ms": sca, "launch_ms": launch_ms, "frac": bound / launch_ms if launch_ms else None,
                # the same with every vector instruction priced as the cheap class (adds, logic, right shifts: 2.5 clocks): the two
                # fractions bracket a kernel's mix -- inflate_batch, mostly shifts and masks, sits at ~1.0 on THIS one and above 1 on `frac`
                "frac_at_2_5_clocks": max(vec * 2.5 / VALU_CLOCKS, sca) / launch_ms if launch_ms else None,
                "clock_hz": SHADER_CLOCK_HZ, "valu_clocks": VALU_CLOCKS, "counters_scaled_by": scale, "source": os.path.basename(f),
                "is": "max(vector, scalar) issue time / measured launch time at the nominal clock (the kernels run at 1.7-2.1 GHz under "
                      "load: the true fraction is higher); vector instructions priced at 4 clocks of their SIMD (compares, selects, left shifts, "
                      "three-operand forms, cross-lane moves: tools/probes/valu_costs.hip) -- a mix of the 2.5-clock class prices above 1 here and "
                      "is bracketed by frac_at_2_5_clocks; counters from a separate profiled run, scaled to this launch's share of the batch; "
                      "a launch beside the other slice's kernels shares its SIMDs with them"}
    except Exception:
        return None


_PIECES = {}


def text_pieces(L):
    if L not in _PIECES:
        _PIECES[L] = _text_pieces(L)
    return _PIECES[L]


def _text_pieces(L):
    """the reference's own texts (tests/golden/zip-docs.zip) as the four chunks of L bytes the text legs repeat"""
    import zipfile

    z = zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", "zip-docs.zip"))
    app = z.read("zip-docs/APPNOTE.TXT")
    rfc = z.read("zip-docs/rfc1951.txt")
    return [app[0:L], app[L:2 * L], (rfc + rfc)[:L], app[100000:100000 + L]]


def leg_plain(source, j, stream_len):
    """stream j of a leg, on the host: ("synth", config, bits) | ("text",) | ("corpus",) | ("zeros",)"""
    if source[0] == "synth":
        from zipc_amd import synth

        return synth.stream_bytes_np(source[1], j, stream_len, source[2]).tobytes()
    if source[0] == "text":
        return text_pieces(stream_len)[j % 4]
    if source[0] == "textlong":  # a long member: the four 64 KiB text chunks over and over, begun at a place of its own
        four = b"".join(text_pieces(65536))
        o = (j * 7919) % len(four)
        return ((four[o:] + four[:o]) * (stream_len // len(four) + 1))[:stream_len]
    if source[0] == "zeros":
        return bytes(stream_len)
    if ("corpus", stream_len) not in _PIECES:
        from tools import corpus

        _PIECES[("corpus", stream_len)] = corpus.chunks(stream_len)
    chunks = _PIECES[("corpus", stream_len)]
    return chunks[j % len(chunks)]


# The legs bench.py reports beside the headline: what each is made of, so that the cpu_baseline process (the only place
# the oracle runs) can time the SAME streams on the host and hand over what the oracle makes of a few of them.
# name: (source, level, stream_len, sampled streams, do_inflate, 1-thread budget in seconds)
LEGS = {
    "c2_default": (("synth", 2, 4), 2, 65536, (0, 1, 16383), True, 12.0),
    "text_default": (("text",), 2, 65536, (0, 1, 2, 3), True, 3.0),
    "text_best": (("text",), 3, 65536, (0, 1, 2, 3), True, 4.0),
    "c2_best": (("synth", 2, 4), 3, 65536, (0, 1), True, 1.5),
    "corpus_default": (("corpus",), 2, 65536, (0, 1, 2), True, 2.0),
    "corpus_best": (("corpus",), 3, 65536, (0, 1), True, 2.0),
    "c4_default": (("synth", 4, 3), 2, 1 << 20, (0, 1), False, 3.0),
    "long_members": (("textlong",), 2, 1 << 20, (0, 63), True, 2.0),
    "one_c1_zeros_1mib": (("zeros",), 1, 1 << 20, (0,), True, 0.5),
    "one_symbols_1mib": (("synth", 2, 4), 2, 1 << 20, (0,), True, 0.5),
}


def _cpu_worker(job):
    """one host thread of the cpu_baseline leg: streams j0, j0 + stride, ... until the deadline"""
    source, level, stream_len, j0, stride, budget_s, do_inflate = job
    import oracle

    t_def = t_inf = 0.0
    n = 0
    j = j0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 65536:
        plain = leg_plain(source, j, stream_len)
        a = time.perf_counter()
        st, comp, _ = oracle.deflate(plain, level=level, crc_op=oracle.CRC_CRC32)
        b = time.perf_counter()
        t_def += b - a
        assert st == 0
        if do_inflate:
            st2, out, _ = oracle.inflate(comp, decompressed_size=stream_len, crc_op=oracle.CRC_CRC32)
            t_inf += time.perf_counter() - b
            assert st2 == 0 and out == plain
        n += 1
        j += stride
    return n, t_def, t_inf, time.perf_counter() - t0


def cpu_leg(name, all_threads=True):
    """The oracle (C port of the reference algorithm) on this host on a bounded sample of one leg's workload: 1 thread
    (like the reference), then every host thread, member-sharded."""
    import multiprocessing as mp

    source, level, stream_len, _, do_inflate, budget_s = LEGS[name]
    n, t_def, t_inf, _ = _cpu_worker((source, level, stream_len, 0, 1, budget_s, do_inflate))
    what = "deflate+inflate" if do_inflate else "deflate"
    line = {
        "value": n * stream_len / GIB / (t_def + t_inf),
        "unit": "GiB/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d streams x %d B of the same workload (%s level %s); timed: oracle/zd_oracle.c, this repository's C restatement of "
                  "src/zipc_deflate.ml (gcc -O2, 1 thread) -- NOT the OCaml reference, which cannot be built on this box"
                  % (n, stream_len, what, LEVELS[level]),
        "deflate_gib_s": n * stream_len / GIB / t_def,
    }
    if do_inflate:
        line["inflate_gib_s"] = n * stream_len / GIB / t_inf
    if not all_threads:
        return line
    threads = os.cpu_count() or 1
    try:
        with mp.get_context("fork").Pool(threads) as pool:
            res = pool.map(_cpu_worker, [(source, level, stream_len, k, threads, budget_s / 2, do_inflate) for k in range(threads)])
        total = sum(r[0] for r in res)
        wall = max(r[3] for r in res)
        line["nproc"] = {"value": total * stream_len / GIB / wall, "unit": "GiB/s", "cores": threads,
                         "deflate_gib_s": total * stream_len / GIB / (sum(r[1] for r in res) / threads) if sum(r[1] for r in res) else None,
                         "sample": "%d streams, member-sharded over %d host threads" % (total, threads)}
    except Exception as e:  # a host that cannot fork: the 1-thread figure stands alone
        line["nproc"] = {"value": None, "error": repr(e)}
    return line


def cpu_samples():
    """What the oracle makes of the sampled streams of every leg: sha256 of the compressed bytes and the CRC-32,
    for the GPU process to hold its own output against (the oracle itself stays in this process)."""
    import hashlib

    import oracle

    out = {}
    for name, (source, level, stream_len, sampled, _, _) in LEGS.items():
        d = {}
        for j in sampled:
            st, comp, crc = oracle.deflate(leg_plain(source, j, stream_len), level=level, crc_op=oracle.CRC_CRC32)
            assert st == 0
            d[str(j)] = [hashlib.sha256(comp).hexdigest()[:24], int(crc), len(comp)]
        out[name] = d
    return out


def cpu_baseline(config):
    """the cpu_baseline object of the JSON line (this process only: host threads, the oracle, no GPU)"""
    if config == "c4":
        line = cpu_leg("c4_default")
        line["samples"] = {"c4_default": cpu_samples()["c4_default"]}
        return line
    line = cpu_leg("c2_default")
    # the other legs the line reports GPU numbers for: a GPU-over-CPU ratio is context, not credit
    line["legs"] = {k: cpu_leg(k) for k in ("text_default", "text_best", "c4_default", "long_members")}
    line["samples"] = cpu_samples()
    return line


def timed_steps(step, barrier, steps, warmup, world, dev):
    import torch
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    mine = elapsed / steps * 1e3
    info = {"rccl_ranks": 1, "ms_per_step_by_rank": [mine]}
    if world > 1:
        # every rank's own time, gathered on DEVICE tensors: the collective goes through RCCL (backend "nccl"), and the
        # number of ranks that answered is what "rccl_ranks" reports
        t = torch.tensor([mine, float(dist.get_rank())], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        per = sorted((int(g[1].item()), float(g[0].item())) for g in got)
        assert [r for r, _ in per] == list(range(world)), "a rank did not answer the all-gather"
        info = {"rccl_ranks": len(per), "rccl_backend": dist.get_backend(), "ms_per_step_by_rank": [m for _, m in per]}
        elapsed = max(m for _, m in per) * steps / 1e3
    return elapsed, info


def roofline_of(ctx, per_step_fn, psteps, N, C, wall=None, alone_pass=False):
    """Per-kernel durations from HIP events on the launch stream (separate, untimed steps).  A step of the batch forms
    is cut into slices on side queues (two by default, api.hip batch_slices): a kernel is then launched once per slice,
    over that slice's streams, beside the other slice's kernels.  `achieved` prices a LAUNCH: the bytes of the
    streams it covers over its duration -- the duration rocprofv3 reports for it under the same command; `alone`
    is the same kernel launched once over the whole batch with nothing beside it (--alone-pass: a second pass with one slice).
    wall: {"def": s, "inf": s} filled by per_step_fn, the two directions' wall time over the psteps steps."""
    from zipc_amd import _lib

    def one_pass():
        ctx.set_profiling(True)
        ctx.reset_kernel_times()
        if wall is not None:
            wall["def"] = wall["inf"] = 0.0
        for _ in range(psteps):
            per_step_fn()
        times = ctx.kernel_times()
        ctx.set_profiling(False)
        return times, (dict(wall) if wall is not None else None)

    times, walls = one_pass()
    alone_times = {}
    if alone_pass:
        _lib.lib().zipc_hip_debug_set_slices(1)
        try:
            alone_times, _ = one_pass()
        finally:
            _lib.lib().zipc_hip_debug_set_slices(0)
    if wall is not None:
        wall.update(walls)  # (the caller reads the default form's)
    kernels = {k: v[1] / v[0] for k, v in times.items()}  # ms per launch
    per_step = {k: v[1] / psteps for k, v in times.items()}  # ms per step, summed over its launches (which overlap other kernels')
    launches = {k: v[0] / psteps for k, v in times.items()}  # launches per step
    alone = {k: v[1] / v[0] for k, v in alone_times.items()}
    alone_launches = {k: v[0] / psteps for k, v in alone_times.items()}
    has_inflate = "inflate_batch" in per_step
    # passes over the batch per step: the CRC-32 kernels run once over the source and once over the output
    passes = {k: (2.0 if k.startswith("crc32") and has_inflate else 1.0) for k in per_step}

    def entry(k, brief):
        share = passes[k] / launches[k] if launches[k] else 1.0  # of the batch, per launch
        e = _roofline_entry(k, kernels[k], N, C, brief=brief, share=share)
        e["launches_per_step"] = launches[k]
        if k in alone:
            a_share = passes[k] / alone_launches[k] if alone_launches.get(k) else 1.0
            a = _roofline_entry(k, alone[k], N, C, brief=True, share=a_share)
            e["alone"] = {"launch_ms": alone[k], "achieved": a["achieved"], "frac": a["frac"], "launches_per_step": alone_launches[k],
                          "is": "one slice: the kernel launched over the whole batch, nothing beside it"}
            if "issue_bound" in a:
                e["alone"]["issue_bound"] = a["issue_bound"]
        return e

    dom = max(per_step, key=per_step.get)
    roof = entry(dom, False)
    # the other kernels of the step, the same way: "roofline" stays the longest launch's, the rest are
    # listed so that a kernel that stops being the longest (inflate_batch in round 2) stays in the line
    roof["others"] = {k: entry(k, True) for k in sorted(per_step, key=per_step.get, reverse=True)
                      if k != dom and per_step[k] >= 0.05 * per_step[dom]}
    # the two directions as wholes, against section 8(d)'s bytes: everything deflate launches (its CRC-32 pass
    # over the source included) for N + C, everything inflate launches for C + N -- over the direction's WALL time
    # per step where it was measured (the kernels of two slices overlap: their durations do not add up to it)
    crc_ms = sum(per_step.get(k, 0.0) for k in ("crc32_segments", "crc32_finish"))
    defl_sum = sum(per_step.get(k, 0.0) for k in DEFLATE_KERNELS) + (crc_ms / 2 if has_inflate else crc_ms)
    defl_ms = walls["def"] / psteps * 1e3 if walls and walls.get("def") else defl_sum
    if defl_ms > 0:
        roof["deflate_pipeline"] = _path_entry(N + C, defl_ms, [k for k in DEFLATE_KERNELS if k in per_step] + ["crc32 (source)"],
                                               sum_traffic([k for k in DEFLATE_KERNELS if k in per_step], launches))
        roof["deflate_pipeline"]["kernels_ms_sum"] = defl_sum
        roof["deflate_pipeline"]["ms_is"] = "wall time of the direction per step" if walls and walls.get("def") else "sum of its kernels' durations"
    if has_inflate:
        inf_sum = per_step["inflate_batch"] + crc_ms / 2
        inf_ms = walls["inf"] / psteps * 1e3 if walls and walls.get("inf") else inf_sum
        roof["inflate"] = _path_entry(C + N, inf_ms, ["inflate_batch", "crc32 (output)"], sum_traffic(["inflate_batch"], launches))
        roof["inflate"]["kernels_ms_sum"] = inf_sum
        roof["inflate"]["ms_is"] = roof["deflate_pipeline"]["ms_is"] if defl_ms > 0 else "sum of its kernels' durations"
    return roof, per_step


def headline_scalars(roof, N, C, deflate_gib_s, inflate_gib_s=None):
    """The claims of the line as SCALARS (a record that keeps only the scalar fields of `roofline` and the `config` object
    still holds them): each direction's GiB/s of uncompressed bytes on its own, its fraction of the HBM roofline on
    SURVEY 8(d)'s bytes (N + C over the direction's wall time / 8 TB/s), and -- where this workload has committed
    counter passes -- the counters' traffic of the direction over those bytes, raw and with the guide's 2 x FETCH."""
    h = {"uncompressed_bytes": N, "compressed_bytes": C, "algorithmic_bytes_per_direction": N + C, "deflate_gib_s": deflate_gib_s}
    if inflate_gib_s is not None:
        h["inflate_gib_s"] = inflate_gib_s
    for key, name in (("deflate_pipeline", "deflate"), ("inflate", "inflate")):
        e = roof.get(key)
        if not e:
            continue
        h[name + "_ms"] = e["ms_per_step"]
        h[name + "_gb_s"] = e["achieved"]
        h[name + "_frac"] = e["frac"]
        if "traffic_over_algorithmic" in e:
            h[name + "_traffic_over_algorithmic_raw"], h[name + "_traffic_over_algorithmic_corrected"] = e["traffic_over_algorithmic"]
            h[name + "_traffic_source"] = e["traffic_source"]
    if isinstance(roof.get("traffic_over_algorithmic"), list):  # the dominant kernel's, as two scalars beside the list
        h["traffic_over_algorithmic_raw"], h["traffic_over_algorithmic_corrected"] = roof["traffic_over_algorithmic"]
    if isinstance(roof.get("issue_bound"), dict):
        h["issue_bound_frac"] = roof["issue_bound"]["frac"]
        h["issue_bound_source"] = roof["issue_bound"]["source"]
    return h


def sum_traffic(kernels, launches=None):
    """the kernels' counter traffic per STEP: the PMC passes give bytes per launch (profiles/*_hbm_traffic.json)"""
    tot = {"bytes": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0}
    src = None
    for k in kernels:
        t, src_k = measured_traffic(k)
        if not t:
            continue
        src = src_k
        per = (launches or {}).get(k, 1.0)
        for f in tot:
            tot[f] += t[f] * per
    return (tot, src) if src else (None, None)


def _path_entry(alg, ms, kernels, traffic):
    achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    e = {"bound": "hbm", "kernels": kernels, "algorithmic_bytes": alg, "ms_per_step": ms, "achieved": achieved,
         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    t, src = traffic
    if t:
        e["traffic"] = t["bytes"]
        e["traffic_corrected"] = 2 * t["fetch_bytes"] + t["write_bytes"]
        e["traffic_over_algorithmic"] = [t["bytes"] / alg, e["traffic_corrected"] / alg]
        e["traffic_source"] = src
    return e


def _roofline_entry(dom, dom_ms, N, C, brief=False, share=1.0):
    """share: the part of the batch one launch covers (1 / slices; 1 with one slice)"""
    alg = algorithmic_bytes(dom, N, C) * share
    des = design_bytes(dom, N, C) * share
    achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic, traffic_src = measured_traffic(dom)
    roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic["bytes"] if traffic else None,
            "traffic_source": traffic_src, "algorithmic_bytes": alg, "launch_ms": dom_ms,
            "algorithmic_is": "SURVEY section 8(d): deflate N + C, inflate C + N, CRC-32 N (the path's minimum, not this kernel's own I/O), of the streams ONE launch covers",
            "batch_share_per_launch": share,
            "design_bytes": des, "design_gb_s": des / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0}
    if traffic:
        # "traffic" is the raw FETCH_SIZE + WRITE_SIZE.  Calibrated on the inflate kernel's stored-block
        # path (profiles/r01_inflate_traffic.txt): WRITE_SIZE exact, FETCH_SIZE tallies every request at
        # 64 B and reads 0.504x a wide streamed read, so raw <= true fetch <= 2 x raw (the guide's gfx950
        # correction is the upper bound); Infinity-Cache hits are included, so HBM-side bytes can be lower.
        roof["traffic_fetch_bounds"] = [traffic["fetch_bytes"], 2 * traffic["fetch_bytes"]]
        roof["traffic_write"] = traffic["write_bytes"]
        roof["traffic_corrected"] = 2 * traffic["fetch_bytes"] + traffic["write_bytes"]
        roof["traffic_over_algorithmic"] = [traffic["bytes"] / alg if alg else None,
                                            roof["traffic_corrected"] / alg if alg else None]
        roof["traffic_is"] = "raw FETCH_SIZE+WRITE_SIZE per launch (L2->fabric); corrected = 2 x fetch + write"
    ib = issue_bound(dom, dom_ms, share if dom not in ("crc32_segments", "crc32_finish") else None)
    if ib:
        roof["issue_bound"] = ib
    if brief:
        roof = {k: roof[k] for k in ("achieved", "frac", "algorithmic_bytes", "design_bytes", "launch_ms", "traffic", "traffic_corrected", "issue_bound") if k in roof}
    return roof


# ---- C2 ----------------------------------------------------------------------------------------------

def run_c2(args, rank, local_rank, world, dev, cpu=None):
    import torch
    import torch.distributed as dist

    import zipc_amd
    from zipc_amd import batch, synth

    global WORKLOAD
    ctx = zipc_amd.Context(local_rank)
    n = args.streams or 16384
    L = args.stream_len or 65536
    bits = args.bits or 4
    N = n * L
    WORKLOAD = "c2" if (n, L, bits, args.level) == (16384, 65536, 4, 2) else None  # whose committed counter passes the line may quote
    # member-sharded: rank r owns streams [r*n, (r+1)*n) of the synthetic archive
    src = synth.batch_bytes_torch(2, rank * n, n, L, bits, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1]) if n > 1 else cap
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(N + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.reserve(ctx, n, L, N)

    # one untimed pass fixes the inflate descriptors (compressed sizes are data
    # dependent but identical every step) and checks the round trip
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1)
    res = batch.results_from_device(d_res)
    assert (res["status"] == 0).all(), "deflate failed"
    C = int(res["out_len"].sum())
    idescs = batch.compact_descs(res, descs, L)
    d_idescs = batch.to_device(idescs, dev)
    batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1)
    ires = batch.results_from_device(d_ires)
    assert (ires["status"] == 0).all() and torch.equal(out[:N], src), "round trip failed"
    assert (ires["checksum"] == res["checksum"]).all()
    # a few of the timed streams byte for byte against what the oracle made of them in the cpu_baseline process
    default_shape = (n, L, bits, args.level) == (16384, 65536, 4, 2) and rank == 0
    same = sampled_equal(comp, descs, res, ((cpu or {}).get("samples") or {}).get("c2_default")) if default_shape else None
    assert same is not False, "a sampled stream of the timed batch differs from the oracle's output"

    def step():
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1, sync=False)
        batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1, sync=False)

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, ranks = timed_steps(step, barrier, args.steps, args.warmup, world, dev)

    psteps = max(1, min(args.steps, 3))
    t = {"def": 0.0, "inf": 0.0}

    def profiled():
        a = time.perf_counter()
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1)
        b = time.perf_counter()
        batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1)
        t["def"] += b - a
        t["inf"] += time.perf_counter() - b

    roof, per_step = roofline_of(ctx, profiled, psteps, N, C, wall=t, alone_pass=args.alone_pass)
    if rank != 0:
        return None
    headline = headline_scalars(roof, N, C, N / GIB * psteps / t["def"], N / GIB * psteps / t["inf"])
    roof.update(headline)
    line = {
        "metric": METRIC,
        "value": world * N / GIB * args.steps / elapsed,
        "unit": "GiB/s",
        # the two directions by themselves (one rank's, over the profiled steps): at the head of the line, where a cut tail keeps them
        "deflate_gib_s": N / GIB * psteps / t["def"],
        "inflate_gib_s": N / GIB * psteps / t["inf"],
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "C2: %d independent streams x %d B of i.i.d. %d-bit symbols per GPU, deflate level %s + "
                        "inflate, CRC-32 of source and output by a separate pass over the bytes, device-resident"
                        % (n, L, bits, LEVELS[args.level]),
            "streams_per_gpu": n, "stream_len": L, "level": args.level,
            "compressed_ratio": C / N, "parallelism": "member-shard x%d" % world,
            "sampled_bytes_equal_oracle": same,
            **headline,
        },
        **ranks,
        "roofline": roof,
        "kernels_ms_per_step": per_step,
    }
    if world == 1 and not args.no_extra_legs:
        line.update(extra_legs(ctx, dev, n, L, cpu))
    return line


def extra_legs(ctx, dev, n, L, cpu):
    """Untimed legs beside the headline (each bounded to a second or two of GPU time): the PCIe-inclusive host forms on
    the same workload, the whole path on real text, a corpus, level `Best, one stream per call, and C4's shape.  Every
    device-resident leg carries its own parity sample (a few streams against the oracle's bytes, worked out in the
    cpu_baseline process), its fraction of the HBM roofline and -- where the cpu_baseline process timed the same
    streams -- the host's figures."""
    import numpy as np
    import torch

    import zipc_amd
    from zipc_amd import batch, synth

    samples = (cpu or {}).get("samples") or {}
    cpu_legs = (cpu or {}).get("legs") or {}
    out = {}
    try:  # e2e: pageable host buffers in, host buffers out (zipc_hip_{deflate,inflate}_many)
        import ctypes as C

        from zipc_amd import _lib

        lib = _lib.lib()
        m = min(n, 4096)
        plain = [synth.stream_bytes_np(2, j, L, 4) for j in range(m)]
        cap = lib.zipc_hip_deflate_bound(L)
        comp = [np.zeros(cap, np.uint8) for _ in range(m)]
        back = [np.zeros(L, np.uint8) for _ in range(m)]
        P, S = C.c_void_p * m, C.c_size_t * m
        srcp, slen = P(*[a.ctypes.data for a in plain]), S(*([L] * m))
        dstp, dcap = P(*[a.ctypes.data for a in comp]), S(*([cap] * m))
        res, ires = (_lib.StreamResult * m)(), (_lib.StreamResult * m)()
        best = [1e9, 1e9]
        for rep in range(3):  # the first call pins the staging buffers
            a = time.perf_counter()
            assert lib.zipc_hip_deflate_many(ctx.handle, m, srcp, slen, 2, 1, dstp, dcap, res) == 0
            b = time.perf_counter()
            clen = S(*[int(res[i].out_len) for i in range(m)])
            backp, bcap = P(*[a_.ctypes.data for a_ in back]), S(*([L] * m))
            c = time.perf_counter()
            assert lib.zipc_hip_inflate_many(ctx.handle, m, dstp, clen, bcap, 1, backp, bcap, ires) == 0
            d = time.perf_counter()
            best = [min(best[0], b - a), min(best[1], d - c)]
        assert all(int(ires[i].status) == 0 for i in range(m)) and np.array_equal(back[m - 1], plain[m - 1])
        out["e2e_gib_s"] = {"deflate": m * L / GIB / best[0], "inflate": m * L / GIB / best[1],
                            "is": "PCIe-inclusive host forms (zipc_hip_*_many) on %d of the streams, pageable host memory in and out, best of 3" % m}
    except Exception as e:
        out["e2e_gib_s"] = {"error": repr(e)}
    pieces = None
    try:  # the reference's own texts (tests/golden/zip-docs.zip), 64 KiB chunks
        pieces = text_pieces(L)
        m = min(n, 16384)
        src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(m)), np.uint8).copy()).to(dev)
        leg = device_round_trip(ctx, dev, src, m, L, 2, expect=samples.get("text_default"))
        leg["cpu_baseline"] = cpu_legs.get("text_default")
        leg["is"] = "%d x 64 KiB chunks of APPNOTE.TXT / rfc1951.txt, level `Default, device-resident" % m
        out["text_gib_s"] = leg
    except Exception as e:
        out["text_gib_s"] = {"error": repr(e)}
    try:  # level `Best, the seam's default (make_encoder ?(level = `Best), zd.ml:817): K = 4096 candidates per position
        m = min(n, 4096)
        src = synth.batch_bytes_torch(2, 0, m, L, 4, dev)
        best = {"c2": device_round_trip(ctx, dev, src, m, L, 3, expect=samples.get("c2_best"))}
        m = min(n, 2048)
        src = torch.from_numpy(np.frombuffer(b"".join(pieces[i % 4] for i in range(m)), np.uint8).copy()).to(dev)
        best["text"] = device_round_trip(ctx, dev, src, m, L, 3, reps=1, expect=samples.get("text_best"))
        best["text"]["cpu_baseline"] = cpu_legs.get("text_best")
        best["is"] = "level `Best on the C2 symbols and on the text chunks, device-resident (the text walks ~300 candidates per position where `Default walks ~33)"
        out["best_gib_s"] = best
    except Exception as e:
        out["best_gib_s"] = {"error": repr(e)}
    try:  # a corpus of distinct real chunks (tools/corpus.py): the reference's documents + this repository's files
        from tools import corpus

        chunks = corpus.chunks(L)
        k = len(chunks)
        m = min(n, 4096)
        src = torch.from_numpy(np.frombuffer(b"".join(chunks[i % k] for i in range(m)), np.uint8).copy()).to(dev)
        out["corpus_gib_s"] = {"default": device_round_trip(ctx, dev, src, m, L, 2, expect=samples.get("corpus_default")),
                               "best": device_round_trip(ctx, dev, src[:2048 * L], min(m, 2048), L, 3, reps=1, expect=samples.get("corpus_best")),
                               "is": "%d distinct 64 KiB chunks (APPNOTE.TXT, rfc1951.txt, this repository's sources, documents, fixtures "
                                     "and built libraries) repeated to %d streams (`Best: 2048), device-resident" % (k, m)}
    except Exception as e:
        out["corpus_gib_s"] = {"error": repr(e)}
    try:  # one stream per call (what every value of the reference's .mli takes): C1's 1 MiB of zeros, 1 MiB and 16 MiB of the C2 symbols
        one = {}
        for name, ML, bits in (("c1_zeros_1mib", 1 << 20, 0), ("symbols_1mib", 1 << 20, 4), ("symbols_16mib", 16 << 20, 4)):
            src = torch.zeros(ML, dtype=torch.uint8, device=dev) if bits == 0 else synth.batch_bytes_torch(2, 0, 1, ML, bits, dev)
            leg = device_round_trip(ctx, dev, src, 1, ML, 1 if bits == 0 else 2, reps=3, expect=samples.get("one_" + name))
            one[name] = {"deflate_ms": ML / GIB / leg["deflate"] * 1e3, "inflate_ms": ML / GIB / leg["inflate"] * 1e3,
                         "sampled_bytes_equal_oracle": leg["sampled_bytes_equal_oracle"]}
        one["is"] = "ONE stream per call, device-resident: `Fast on 1 MiB of zeros (BASELINE C1's input), `Default on the C2 symbols; " \
                    "deflate of a long stream runs as segments and blocks on many waves, inflate as a wave per block (block starts " \
                    "searched for; copies resolved by pointer jumping) -- zeros are one block: the stream's one wave"
        out["one_stream_ms"] = one
    except Exception as e:
        out["one_stream_ms"] = {"error": repr(e)}
    try:  # an archive's few long members in one call: 64 x 1 MiB of text -- too few for a wave each to fill the device
        m, ML = 64, 1 << 20
        src = torch.from_numpy(np.frombuffer(b"".join(leg_plain(("textlong",), j, ML) for j in range(m)), np.uint8).copy()).to(dev)
        leg = device_round_trip(ctx, dev, src, m, ML, 2, reps=3, expect=samples.get("long_members"))
        leg["cpu_baseline"] = cpu_legs.get("long_members")
        leg["deflate_ms"], leg["inflate_ms"] = m * ML / GIB / leg["deflate"] * 1e3, m * ML / GIB / leg["inflate"] * 1e3
        leg["inflate_blocks"] = ctx.last_inflate_blocks()
        leg["is"] = "%d members x 1 MiB of text in one call of the batch forms, device-resident: deflate runs them as segments on many " \
                    "waves, inflate by blocks side by side (inflate_blocks: blocks decoded that way in the last call)" % m
        out["long_members"] = leg
    except Exception as e:
        out["long_members"] = {"error": repr(e)}
    try:  # C4 shape on this GPU: 4096 members x 1 MiB of 3-bit symbols (a wave per member: half of them leaves the GPU half empty)
        m, ML = 4096, 1 << 20
        src = synth.batch_bytes_torch(4, 0, m, ML, 3, dev)
        leg = device_round_trip(ctx, dev, src, m, ML, 2, expect=samples.get("c4_default"))
        leg["cpu_baseline"] = cpu_legs.get("c4_default")
        out["c4_deflate_gib_s"] = leg["deflate"]
        out["c4_inflate_gib_s"] = leg["inflate"]
        out["c4_leg"] = leg
        out["c4_leg_is"] = "%d members x 1 MiB of 3-bit symbols on this GPU, device-resident (bench.py --config c4 runs all 8192)" % m
    except Exception as e:
        out["c4_deflate_gib_s"] = {"error": repr(e)}
    return out


def sampled_equal(comp, descs, res, expect):
    """Do the sampled streams' compressed bytes (device tensor comp, layout descs, results res) and CRC-32s equal what
    the oracle made of them in the cpu_baseline process (expect: stream -> [sha256 prefix, crc, length])?  None: no
    expectation at hand (--no-cpu-baseline, N > 1)."""
    import hashlib

    if not expect:
        return None
    ok = True
    for j, (sha, crc, ln) in expect.items():
        j = int(j)
        if j >= len(res):
            continue
        o, k = int(descs["dst_off"][j]), int(res["out_len"][j])
        got = comp[o:o + k].cpu().numpy().tobytes()
        ok = ok and k == ln and int(res["checksum"][j]) == crc and hashlib.sha256(got).hexdigest()[:24] == sha
    return ok


def device_round_trip(ctx, dev, src, n, L, level, reps=2, expect=None):
    """One leg: n streams of L bytes held in `src`, device-resident, deflate (CRC-32) then inflate, checked by the
    round trip and -- a few sampled streams -- byte for byte against the oracle's output (expect).  Returns GiB/s and
    the two directions against the HBM roofline on SURVEY 8(d)'s bytes (N + C each way)."""
    import torch

    from zipc_amd import batch

    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1]) if n > 1 else cap
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
    res = batch.results_from_device(d_res)
    assert (res["status"] == 0).all()
    same = sampled_equal(comp, descs, res, expect)
    assert same is not False, "a sampled stream differs from the oracle's output"
    d_id = batch.to_device(batch.compact_descs(res, descs, L), dev)
    batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1)
    assert torch.equal(out[:n * L], src)
    td = ti = 0.0
    for _ in range(reps):
        a = time.perf_counter()
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, n * L, level, 1)
        b = time.perf_counter()
        batch.inflate_batch(ctx, comp, out, d_id, d_ires, n, L, 1)
        td += b - a
        ti += time.perf_counter() - b
    N, C = n * L, int(res["out_len"].sum())
    gd, gi = N / GIB * reps / td, N / GIB * reps / ti
    roof = {d: {"achieved": (N + C) / (t / reps) / 1e9, "frac": (N + C) / (t / reps) / 1e9 / HBM_PEAK_GBS}
            for d, t in (("deflate", td), ("inflate", ti))}
    return {"deflate": gd, "inflate": gi, "streams": n, "compressed_ratio": C / N, "sampled_bytes_equal_oracle": same,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes": N + C, **roof}}


# ---- C4 ----------------------------------------------------------------------------------------------

def run_c4(args, rank, local_rank, world, dev, cpu=None):
    import numpy as np
    import torch
    import torch.distributed as dist

    import zipc_amd
    from zipc_amd import batch, shard, synth

    global WORKLOAD
    ctx = zipc_amd.Context(local_rank)
    members = args.streams or 8192
    L = args.stream_len or (1 << 20)
    bits = args.bits or 3
    WORKLOAD = "c4" if (members, L, bits, args.level, world) == (8192, 1 << 20, 3, 2, 1) else None  # counters of a C4 pass on one GPU, or none
    paths = shard.member_paths(members)           # m/%05d.bin: already in the order Zipc writes them
    parts = shard.partition([L] * members, world)
    lo, hi = parts[rank]
    n = hi - lo
    N = n * L
    src = synth.batch_bytes_torch(4, lo, n, L, bits, dev)
    cap = batch.deflate_bound(L)
    descs = batch.uniform_layout(n, L, cap)
    slot = int(descs["dst_off"][1]) if n > 1 else cap
    comp = torch.zeros(n * slot + 256, dtype=torch.uint8, device=dev)
    d_descs = batch.to_device(descs, dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.reserve(ctx, n, L, N)
    counts = [b - a for a, b in parts]
    state = {}

    def step():
        # crc_32_and_deflate of this rank's members, then the one exchange of the path: the records
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1, sync=False)
        ctx.synchronize()
        res = batch.results_from_device(d_res)
        local = np.zeros(n, dtype=shard.RECORD_DTYPE)
        local["compressed_size"] = res["out_len"].astype(np.uint32)
        local["crc32"] = res["checksum"]
        local["arena_offset"] = descs["dst_off"]
        state["res"] = res
        state["records"] = shard.gather_records(local, counts)

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, ranks = timed_steps(step, barrier, args.steps, args.warmup, world, dev)
    res, records = state["res"], state["records"]
    assert (res["status"] == 0).all(), "deflate failed"
    assert len(records) == members
    C_all = int(records["compressed_size"].astype(np.uint64).sum())
    check = None if args.no_archive_check else c4_check(ctx, dev, rank, world, parts, paths, records, comp, descs, L, bits, args.level)

    t4 = {"def": 0.0, "inf": 0.0}

    def profiled4():
        a = time.perf_counter()
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1)
        t4["def"] += time.perf_counter() - a

    roof, per_step = roofline_of(ctx, profiled4, 1, N, int(res["out_len"].sum()), wall=t4)
    if rank != 0:
        return None
    headline = headline_scalars(roof, N, int(res["out_len"].sum()), N / GIB / t4["def"])
    roof.update(headline)
    line = {
        "metric": METRIC,
        "value": members * L / GIB * args.steps / elapsed,
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "C4: ZIP archive of %d members x %d B of i.i.d. %d-bit symbols, crc_32_and_deflate level %s, "
                        "member-sharded over %d GPU(s), records all-gathered (deflate only: value counts source bytes)"
                        % (members, L, bits, LEVELS[args.level], world),
            "members": members, "member_len": L, "level": args.level, "compressed_ratio": C_all / (members * L),
            "parallelism": "member-shard x%d" % world, "members_per_rank": counts,
            **{k: v for k, v in headline.items()},  # rank 0's own shard (N = its members' bytes)
        },
        **ranks,
        "roofline": roof,
        "kernels_ms_per_step": per_step,
        "archive_check": check,
    }
    return line


def c4_check(ctx, dev, rank, world, parts, paths, records, comp, descs, L, bits, level):
    """Untimed: the WHOLE archive.  Every rank copies its members' compressed bytes off its GPU and sends them
    to rank 0 (one payload gather); rank 0 lays out the ZIP of all members with the host layer
    (zipc_amd/host: src/zipc.ml:568-588) from the gathered records and payloads, lets Info-ZIP test the whole
    file (`unzip -tq`: every member inflated and CRC-checked by an independent implementation) and -- with more
    than one rank -- deflates ALL members itself and requires the archive of the N ranks to equal, byte for
    byte, the archive one rank writes."""
    import hashlib
    import shutil
    import tempfile

    import numpy as np
    import torch

    from zipc_amd import batch, shard, synth

    lo, hi = parts[rank]
    n = hi - lo
    members = len(paths)
    counts = [b - a for a, b in parts]
    host = comp.cpu().numpy()
    mine = b"".join(host[int(descs["dst_off"][k]):int(descs["dst_off"][k]) + int(records["compressed_size"][lo + k])].tobytes()
                    for k in range(n))
    del host
    gathered = shard.gather_payloads(mine)
    if rank != 0:
        return None
    t0 = time.perf_counter()
    blob = shard.assemble_archive(paths, records, gathered, counts, L)
    digest = hashlib.sha256(blob).hexdigest()
    unzip_ok = None
    if shutil.which("unzip"):
        with tempfile.NamedTemporaryFile(suffix=".zip") as f:
            f.write(blob)
            f.flush()
            unzip_ok = subprocess.run(["unzip", "-tq", f.name], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
    # Python's zipfile as the second independent reader (the only one where Info-ZIP is not installed): the names in the
    # order Zipc writes them, every member inflated and CRC-checked
    import io
    import zipfile

    with zipfile.ZipFile(io.BytesIO(blob)) as zf:
        zipfile_ok = zf.namelist() == [q.decode() for q in paths] and zf.testzip() is None
    same = None
    if world > 1:
        # the archive one rank writes: rank 0 deflates every member itself, range by range through its own arenas
        del gathered
        recs1 = np.zeros(members, dtype=shard.RECORD_DTYPE)
        parts1 = []
        cap = batch.deflate_bound(L)
        step = max(n, 1)
        for a0 in range(0, members, step):
            m = min(step, members - a0)
            src = synth.batch_bytes_torch(4, a0, m, L, bits, dev)
            d = batch.uniform_layout(m, L, cap)
            d_res = torch.zeros(m * 16, dtype=torch.uint8, device=dev)
            batch.deflate_batch(ctx, src, comp, batch.to_device(d, dev), d_res, m, L, m * L, level, 1)
            r1 = batch.results_from_device(d_res)
            assert (r1["status"] == 0).all()
            h = comp.cpu().numpy()
            parts1.append(b"".join(h[int(d["dst_off"][k]):int(d["dst_off"][k]) + int(r1["out_len"][k])].tobytes() for k in range(m)))
            recs1["compressed_size"][a0:a0 + m] = r1["out_len"].astype(np.uint32)
            recs1["crc32"][a0:a0 + m] = r1["checksum"]
            del src, h
        blob1 = shard.assemble_archive(paths, recs1, parts1, [min(step, members - a0) for a0 in range(0, members, step)], L)
        same = hashlib.sha256(blob1).hexdigest() == digest and len(blob1) == len(blob)
    ok = unzip_ok is not False and zipfile_ok and same is not False
    assert ok, "C4 archive check failed: equal to the one-rank archive: %s, unzip -tq: %s, zipfile: %s" % (same, unzip_ok, zipfile_ok)
    return {"members": members, "archive_bytes": len(blob), "sha256": digest, "unzip_tq_ok": unzip_ok, "zipfile_ok": zipfile_ok,
            "bytes_equal_one_rank_archive": same, "check_s": time.perf_counter() - t0,
            "is": "all members gathered to rank 0, laid out by zipc_amd/host, the whole file tested by Info-ZIP; "
                  "with N > 1 also compared with the archive rank 0 writes alone"}


# ---- main --------------------------------------------------------------------------------------------

def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        return launch_workers(args.gpus, argv)
    world = int(env_world) if env_world is not None else 1
    if args.gpus is not None and args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.cpu_baseline_only:  # a process of its own: host threads only, nothing of the GPU in it
        if args.config == "c2" and (args.bits or 4) == 4 and args.level == 2 and (args.stream_len or 65536) == 65536:
            line = cpu_baseline("c2")
        elif args.config == "c4" and (args.bits or 3) == 3 and args.level == 2 and (args.stream_len or (1 << 20)) == (1 << 20):
            line = cpu_baseline("c4")
        else:  # a workload of the caller's own: its one leg, no samples
            LEGS["custom"] = (("synth", 4 if args.config == "c4" else 2, args.bits or (3 if args.config == "c4" else 4)), args.level,
                              args.stream_len or ((1 << 20) if args.config == "c4" else 65536), (), args.config != "c4", 8.0)
            line = cpu_leg("custom")
        print(json.dumps(line))
        return 0
    sys.stderr.write("[bench] rank %d of %d (local rank %d), config %s\n" % (rank, world, local_rank, args.config))
    sys.stderr.flush()
    if os.environ.get("ZIPC_BENCH_TEST_HANG_RANK") == str(rank):  # tests/test_bench_launch.py: a rank that never arrives
        time.sleep(3600)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        # the oracle on this host's cores, timed before this process touches the GPU (its worker pool forks)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"] + argv,
                           stdout=subprocess.PIPE)
        try:
            cpu = json.loads(r.stdout.decode().strip().splitlines()[-1])
        except Exception:
            cpu = {"error": "cpu baseline leg failed (exit %d)" % r.returncode}

    import torch

    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: ZIPC_HIP_ERR_NO_DEVICE: no GPU visible to rank %d (there is no CPU path)\n" % rank)
        return 3
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    line = (run_c4 if args.config == "c4" else run_c2)(args, rank, local_rank, world, dev, cpu)
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
