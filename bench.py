#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric on its config C2.

A "step" is one pass of the hot path over one batch of synthetic input:
  deflate (level `Default, fused CRC-32) of 16 384 x 64 KiB streams of i.i.d.
  4-bit symbols (1 GiB), then inflate (fused CRC-32) of the 16 384 compressed
  streams, both through the device-resident batch forms of include/zipc_hip.h.
Inputs are generated on the device and stay in HBM; value = uncompressed GiB
round-tripped per second (whole job over all ranks).  With N > 1 ranks the
streams are member-sharded, one process per GPU, no data-path collective
(weak scaling: every rank runs the full 1 GiB config on its own streams).

Extra objects in the JSON line: "roofline" (dominant kernel: algorithmic bytes /
HIP-event duration vs the 8 TB/s HBM peak), "cpu_baseline" (the oracle's C port
timed on this host on a bounded sample, rank 0 / N=1 only), "kernels"
(per-kernel ms per step from HIP events), "inflate_gib_s" / "deflate_gib_s".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(kernel, N, C):
    """HBM bytes one launch of `kernel` must move at minimum, for N uncompressed
    and C compressed bytes per step (DESIGN.md 'Kernels')."""
    return {
        "inflate_batch": C + N,            # read compressed, write plain
        "crc32_segments": N,               # one pass over the checked bytes
        "lz_chain": N + 2 * N,             # read source, write 2-byte links
        "lz_match": N + 2 * N + 8 * N,     # source + links in, 8-byte macro steps out
        "lz_parse": 8 * N + N + 4 * N,     # macro steps + literals in, <= 4 B/symbol out
        "deflate_emit": 4 * N + C,         # symbols in, compressed out
    }.get(kernel, 0)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*hbm_traffic.json,
    made by tools/pmc_report.py --hbm-json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command), or None."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        k = d["kernels"][kernel]
        return k, os.path.basename(files[-1])
    except Exception:
        return None, None


SHADER_CLOCK_HZ = 2.38e9  # s_memtime against s_memrealtime under load, profiles/r01_match_phases.txt


def issue_bound(kernel, launch_ms):
    """What the instruction-issue ports allow for `kernel`, from the committed SQ counter pass
    (profiles/*sq_counters.json, tools/exp_sq_counters.sh): a SIMD issues one wave64 vector
    instruction per 4 clocks (1024 SIMDs), a CU one scalar instruction per clock for all its
    waves (256 CUs).  None of the path's kernels is bound by HBM or MFMA; this is the bound
    that is close."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"][kernel]
        vec = k["SQ_INSTS_VALU"] * 4 / (1024 * SHADER_CLOCK_HZ) * 1e3
        sca = (k["SQ_INSTS_SALU"] + k["SQ_INSTS_BRANCH"] + k.get("SQ_INSTS_SMEM", 0)) / (256 * SHADER_CLOCK_HZ) * 1e3
        bound = max(vec, sca)
        return {"vector_ms": vec, "scalar_ms": sca, "launch_ms": launch_ms, "frac": bound / launch_ms if launch_ms else None,
                "clock_hz": SHADER_CLOCK_HZ, "source": os.path.basename(files[-1]),
                "is": "max(vector, scalar) issue time / measured launch time; counters from a separate profiled run"}
    except Exception:
        return None


def cpu_baseline(config_id, bits, level, stream_len, budget_s=15.0):
    """The oracle (C port of the reference algorithm) on this host, 1 thread, on a
    bounded sample of the same workload."""
    import oracle
    from zipc_amd import synth

    t_def = t_inf = 0.0
    nbytes = 0
    j = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and j < 65536:
        plain = synth.stream_bytes_np(config_id, j, stream_len, bits).tobytes()
        a = time.perf_counter()
        st, comp, _ = oracle.deflate(plain, level=level, crc_op=oracle.CRC_CRC32)
        b = time.perf_counter()
        st2, out, _ = oracle.inflate(comp, decompressed_size=stream_len, crc_op=oracle.CRC_CRC32)
        c = time.perf_counter()
        assert st == 0 and st2 == 0 and out == plain
        t_def += b - a
        t_inf += c - b
        nbytes += stream_len
        j += 1
    return {
        "value": nbytes / GIB / (t_def + t_inf),
        "unit": "GiB/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d streams x %d B of the same workload (deflate+inflate, oracle/zd_oracle.c, gcc -O2)" % (j, stream_len),
        "deflate_gib_s": nbytes / GIB / t_def,
        "inflate_gib_s": nbytes / GIB / t_inf,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=16384, help="streams per rank (config C2: 16384)")
    ap.add_argument("--stream-len", type=int, default=65536)
    ap.add_argument("--bits", type=int, default=4, help="entropy bits per byte (C2: 4)")
    ap.add_argument("--level", type=int, default=2, help="0 none, 1 fast, 2 default, 3 best")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import zipc_amd
    from zipc_amd import batch, synth

    ctx = zipc_amd.Context(local_rank)
    n, L = args.streams, args.stream_len
    N = n * L
    # member-sharded: rank r owns streams [r*n, (r+1)*n) of the synthetic archive
    src = synth.batch_bytes_torch(2, rank * n, n, L, args.bits, dev)
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

    def step(sync_each=False):
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1, sync=sync_each)
        batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1, sync=sync_each)

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations: HIP events on the launch stream, separate (untimed)
    # steps so that the event records do not perturb the headline number
    ctx.set_profiling(True)
    ctx.reset_kernel_times()
    psteps = max(1, min(args.steps, 3))
    t_def = t_inf = 0.0
    for _ in range(psteps):
        a = time.perf_counter()
        batch.deflate_batch(ctx, src, comp, d_descs, d_res, n, L, N, args.level, 1)
        b = time.perf_counter()
        batch.inflate_batch(ctx, comp, out, d_idescs, d_ires, n, L, 1)
        c = time.perf_counter()
        t_def += b - a
        t_inf += c - b
    times = ctx.kernel_times()
    ctx.set_profiling(False)
    kernels = {k: v[1] / v[0] for k, v in times.items()}  # ms per launch
    per_step = {k: v[1] / psteps for k, v in times.items()}  # ms per step
    dom = max(per_step, key=per_step.get)
    dom_ms = kernels[dom]
    alg = algorithmic_bytes(dom, N, C)
    achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic, traffic_src = measured_traffic(dom)

    if rank == 0:
        line = {
            "metric": "GiB/s deflate+inflate on 1 GiB synthetic; bit-exact vs Zipc_deflate",
            "value": world * N / GIB * args.steps / elapsed,
            "unit": "GiB/s",
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
                "workload": "C2: %d independent streams x %d B of i.i.d. %d-bit symbols per GPU, "
                            "deflate level %s + inflate, CRC-32 fused, device-resident"
                            % (n, L, args.bits, ["none", "fast", "default", "best"][args.level]),
                "streams_per_gpu": n, "stream_len": L, "level": args.level,
                "compressed_ratio": C / N, "parallelism": "member-shard x%d" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic_src, "algorithmic_bytes": alg, "launch_ms": dom_ms,
            },
            "deflate_gib_s": N / GIB * psteps / t_def,
            "inflate_gib_s": N / GIB * psteps / t_inf,
            "kernels_ms_per_step": per_step,
        }
        if traffic:
            # what "traffic" is: raw FETCH_SIZE + WRITE_SIZE. Calibrated on this kernel's stored-block
            # path (profiles/r01_inflate_traffic.txt): WRITE_SIZE exact, FETCH_SIZE tallies every
            # request at 64 B and reads 0.504x a wide streamed read, so raw <= true fetch <= 2 x raw;
            # Infinity-Cache hits are included, so HBM-side bytes can be lower.
            line["roofline"]["traffic_fetch_bounds"] = [traffic["fetch_bytes"], 2 * traffic["fetch_bytes"]]
            line["roofline"]["traffic_write"] = traffic["write_bytes"]
            line["roofline"]["traffic_is"] = "raw FETCH_SIZE+WRITE_SIZE per launch (L2->fabric, lower bound on reads)"
        ib = issue_bound(dom, dom_ms)
        if ib:
            line["roofline"]["issue_bound"] = ib
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(2, args.bits, args.level, L)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
