"""Member sharding (zipc_amd/shard.py): partition properties and the N > 1 record
all-gather on CPU (gloo, world_size 2)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_contiguous_complete_and_balanced():
    from zipc_amd.shard import partition

    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 5, 8192):
            sizes = rng.integers(1, 1 << 20, size=n)
            parts = partition(sizes, world)
            assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and a <= b
            if n >= 64 * world:
                loads = [int(sizes[a:b].sum()) for a, b in parts]
                assert max(loads) - min(loads) <= 2 * int(sizes.max())
    assert partition([1 << 20] * 8192, 8) == [(k * 1024, (k + 1) * 1024) for k in range(8)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from zipc_amd.shard import RECORD_DTYPE, archive_offsets, gather_records, partition

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    sizes = np.arange(1, 12, dtype=np.uint64) * 1000  # 11 members
    parts = partition(sizes, world)
    lo, hi = parts[rank]
    local = np.zeros(hi - lo, dtype=RECORD_DTYPE)
    local["compressed_size"] = (sizes[lo:hi] // 2).astype(np.uint32)
    local["crc32"] = np.arange(lo, hi, dtype=np.uint32) * 7 + 1
    local["arena_offset"] = np.arange(hi - lo, dtype=np.uint64) * 4096
    allrec = gather_records(local, [b - a for a, b in parts])
    offs = archive_offsets(allrec, np.full(len(allrec), 30))
    q.put((rank, allrec["crc32"].tolist(), allrec["compressed_size"].tolist(), offs.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_record_allgather_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_crc = [i * 7 + 1 for i in range(11)]
    want_size = [((i + 1) * 1000) // 2 for i in range(11)]
    for rank, crc, size, offs in got:
        assert crc == want_crc and size == want_size  # every rank sees all members, in member order
        assert offs[0] == 0 and offs[1] == want_size[0] + 30
