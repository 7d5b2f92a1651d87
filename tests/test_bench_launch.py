"""bench.py as the launcher of its own ranks (no GPU here): `--gpus N` starts N worker processes
with the torch.distributed environment, each fails cleanly without a GPU, and the launcher
reports that with a non-zero exit; a --gpus that contradicts WORLD_SIZE is refused."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout)


def test_gpus_n_spawns_n_ranks_and_fails_cleanly_without_a_gpu():
    for n, config in ((2, "c2"), (3, "c4")):
        r = run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--config", config])
        err = r.stderr.decode()
        ranks = sorted(int(m.group(1)) for m in re.finditer(r"\[bench\] rank (\d+) of %d " % n, err))
        assert ranks == list(range(n)), err  # the spawned world size equals --gpus
        assert r.returncode != 0 and "workers failed" in err
        # the rank that fails first says why (the launcher stops the others at once); nothing falls back to the CPU
        assert 1 <= err.count("ZIPC_HIP_ERR_NO_DEVICE") <= n
        assert r.stdout.decode().strip() == ""  # and no JSON line is made up


def test_a_dead_rank_stops_the_others_at_once():
    """Rank 1 never arrives (it sleeps for an hour), rank 0 dies without a GPU: the launcher must not wait for
    rank 1 -- on a GPU node the survivors would sit in the rendezvous until the RCCL timeout -- but stop it."""
    import time

    t0 = time.time()
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"ZIPC_BENCH_TEST_HANG_RANK": "1"},
            timeout=120)
    took = time.time() - t0
    err = r.stderr.decode()
    assert r.returncode != 0 and "workers failed" in err and "the others were stopped" in err
    assert took < 60, took
    # nothing of the run is left behind
    ps = subprocess.run(["ps", "-eo", "pid,args"], stdout=subprocess.PIPE).stdout.decode()
    assert not [l for l in ps.splitlines() if "bench.py" in l and "--gpus 2" in l and str(os.getpid()) not in l.split()[0:1]], ps


def test_gpus_must_agree_with_world_size():
    r = run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and b"WORLD_SIZE=4" in r.stderr


def test_cpu_baseline_leg_reports_one_thread_and_all_threads():
    import bench

    bench.LEGS["quick"] = (("synth", 2, 4), 2, 65536, (0, 1), True, 1.0)
    line = bench.cpu_leg("quick")
    assert line["cores"] == 1 and line["kind"] == "port" and line["value"] > 0
    assert line["nproc"]["cores"] == (os.cpu_count() or 1) and line["nproc"]["value"] > line["value"] * 0.5


def test_cpu_baseline_leg_hands_over_the_oracles_samples():
    """the legs' parity samples: what the oracle makes of a few streams of every leg, worked out in the cpu_baseline
    process (the only place bench.py runs the oracle) -- digest, CRC-32 and length per sampled stream"""
    import hashlib

    import bench
    import oracle

    keep = dict(bench.LEGS)
    try:
        for k in list(bench.LEGS):
            if k not in ("text_default", "one_c1_zeros_1mib"):
                del bench.LEGS[k]
        s = bench.cpu_samples()
    finally:
        bench.LEGS.clear()
        bench.LEGS.update(keep)
    assert sorted(s) == ["one_c1_zeros_1mib", "text_default"] and sorted(s["text_default"]) == ["0", "1", "2", "3"]
    st, comp, crc = oracle.deflate(bench.text_pieces(65536)[2], level=2, crc_op=oracle.CRC_CRC32)
    assert s["text_default"]["2"] == [hashlib.sha256(comp).hexdigest()[:24], crc, len(comp)]
