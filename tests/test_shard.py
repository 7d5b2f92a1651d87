"""Member sharding (zipc_amd/shard.py): partition properties and the N > 1 record
all-gather on CPU (gloo, world_size 2 and 4)."""
import os
import socket
import sys

import numpy as np
import pytest
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


@pytest.mark.parametrize("world", [2, 4])
def test_record_allgather_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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


def test_host_layer_cuts_members_like_the_ranks():
    """zipc_amd/host (Archive::add_deflated_files / extract_all over several devices) and bench.py's ranks use one
    rule: zipc_host_partition against shard.partition."""
    import random

    from zipc_amd import shard, zipc_host

    r = random.Random(5)
    for _ in range(500):
        n, w = r.randrange(0, 60), r.randrange(1, 9)
        sizes = [r.choice([0, 1, 7, 4096, 1 << 20, r.randrange(0, 1 << 22)]) for _ in range(n)]
        assert zipc_host.partition(sizes, w) == [(int(a), int(b)) for a, b in shard.partition(sizes, w)]



def _member(j, L):
    import random

    r = random.Random(1000 + j)
    return bytes(r.randrange(8) for _ in range(L))


def _raw_deflate(data):
    import zlib

    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush()


def _archive_worker(rank, world, port, members, L, q):
    """One rank of the C4 flow without a GPU: its members compressed (zlib stands in for the kernels -- the
    layout is what is under test), the records all-gathered, the payloads gathered to rank 0, the archive laid
    out there (bench.py c4_check does the same with the GPU's bytes)."""
    import zlib

    import torch.distributed as dist

    from zipc_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    paths = shard.member_paths(members)
    parts = shard.partition([L] * members, world)
    counts = [b - a for a, b in parts]
    lo, hi = parts[rank]
    comp = [_raw_deflate(_member(j, L)) for j in range(lo, hi)]
    local = np.zeros(hi - lo, dtype=shard.RECORD_DTYPE)
    local["compressed_size"] = [len(c) for c in comp]
    local["crc32"] = [zlib.crc32(_member(j, L)) for j in range(lo, hi)]
    records = shard.gather_records(local, counts)
    gathered = shard.gather_payloads(b"".join(comp))
    if rank == 0:
        q.put(shard.assemble_archive(paths, records, gathered, counts, L))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_archive_of_n_ranks_equals_the_archive_of_one(world):
    """The N-rank archive is the 1-rank archive: members spread over 2 and over 4 ranks (gloo), payloads gathered, laid out
    by the host layer -- against the same members laid out in one process, and read back by zipfile."""
    import io
    import zipfile
    import zlib

    from zipc_amd import shard

    members, L = 23, 3000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_archive_worker, args=(r, world, port, members, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob2 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    comp = [_raw_deflate(_member(j, L)) for j in range(members)]
    rec = np.zeros(members, dtype=shard.RECORD_DTYPE)
    rec["compressed_size"] = [len(c) for c in comp]
    rec["crc32"] = [zlib.crc32(_member(j, L)) for j in range(members)]
    blob1 = shard.assemble_archive(shard.member_paths(members), rec, [b"".join(comp)], [members], L)
    assert blob2 == blob1
    with zipfile.ZipFile(io.BytesIO(blob2)) as z:
        assert z.testzip() is None
        assert z.namelist() == [p.decode() for p in shard.member_paths(members)]
        assert z.read("m/00007.bin") == _member(7, L)
