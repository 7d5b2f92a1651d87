"""Member sharding over the GPUs of one node.

The path shards by independent units -- archive members / 64 KiB chunks, each its
own deflate stream (make_encoder / make_decoder are per call, zipc_deflate.ml:548,
817) -- so there is NO collective on the data path: rank r compresses a contiguous
range of members into its own arena.  The only exchange is an all-gather of one
fixed-size record per member {compressed_size, crc32, arena_offset} so that every
rank can lay out the archive (prefix sum of sizes, the order Zipc writes members,
src/zipc.ml:575-581).
"""
from __future__ import annotations

import numpy as np

RECORD_DTYPE = np.dtype([("compressed_size", "<u4"), ("crc32", "<u4"), ("arena_offset", "<u8")])


def member_paths(n: int):
    """C4's member names m/00000.bin ... (bytes).  Zipc writes members in the byte order of their
    paths, `mimetype` first (src/zipc.ml:568-583): zero-padded numbers are already in that order,
    so member index = position in the archive."""
    return [b"m/%05d.bin" % j for j in range(n)]


def partition(sizes, world_size: int):
    """Contiguous ranges [lo, hi) per rank, balanced by uncompressed bytes.

    Boundaries are the member indices where the running byte count crosses
    k/world_size of the total; every member belongs to exactly one rank and the
    ranges keep member order."""
    sizes = np.asarray(sizes, dtype=np.uint64)
    n = len(sizes)
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    if n == 0:
        return [(0, 0)] * world_size
    csum = np.concatenate([[0], np.cumsum(sizes, dtype=np.float64)])
    total = csum[-1]
    bounds = [0]
    for k in range(1, world_size):
        target = total * k / world_size
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), n)
        bounds.append(b)
    bounds.append(n)
    return [(bounds[k], bounds[k + 1]) for k in range(world_size)]


def gather_records(local: np.ndarray, counts, group=None) -> np.ndarray:
    """All-gather of the per-member records (RCCL on GPU ranks, gloo on CPU).

    `local`: RECORD_DTYPE array of this rank's members; `counts[r]`: members of
    rank r (known to every rank from `partition`).  Returns all records in member
    order."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.copy()
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    width = max(int(max(counts)), 1) * RECORD_DTYPE.itemsize
    mine = torch.zeros(width, dtype=torch.uint8)
    raw = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    mine[:raw.size] = torch.from_numpy(raw.copy())
    mine = mine.to(dev)
    out = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    parts = []
    for r in range(world):
        nbytes = int(counts[r]) * RECORD_DTYPE.itemsize
        parts.append(out[r][:nbytes].cpu().numpy().view(RECORD_DTYPE))
    return np.concatenate(parts) if parts else np.zeros(0, RECORD_DTYPE)


def archive_offsets(records: np.ndarray, header_sizes) -> np.ndarray:
    """Local-file-header offsets of the members in archive order: exclusive prefix
    sum of (header + compressed payload) sizes."""
    sizes = records["compressed_size"].astype(np.uint64) + np.asarray(header_sizes, dtype=np.uint64)
    return np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)


def gather_payloads(local: bytes, dst: int = 0, group=None):
    """Every rank's compressed members (its part of the arena, members back to back) to rank `dst`: one
    point-to-point payload gather, SURVEY 8(e)'s CG2 (RCCL over xGMI on GPU ranks -- about 0.43 GiB per peer
    at C4 -- gloo on CPU).  Returns the list of parts in rank order on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(local)], dtype=torch.int64, device=dev), group=group)
    sizes = [int(t.item()) for t in sizes]
    width = max(max(sizes), 1)
    mine = torch.zeros(width, dtype=torch.uint8)
    if local:
        mine[:len(local)] = torch.from_numpy(np.frombuffer(local, np.uint8).copy())
    mine = mine.to(dev)
    if rank == dst:
        out = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(mine, out, dst=dst, group=group)
        return [out[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world)]
    dist.gather(mine, None, dst=dst, group=group)
    return None


def assemble_archive(paths, records: np.ndarray, parts, counts, member_len) -> bytes:
    """The ZIP archive of the sharded members on the rank that holds all parts (zipc_amd/host: Zipc.add +
    Zipc.to_binary_string, src/zipc.ml:568-588): member j is `paths[j]`, Deflate, its compressed bytes the
    next records['compressed_size'][j] bytes of its rank's part, its CRC-32 and size from the records.  The
    archive depends on the members only, not on how they were spread over ranks."""
    from . import zipc_host

    a = zipc_host.Archive()
    j = 0
    for r, part in enumerate(parts):
        at = 0
        for _ in range(int(counts[r])):
            size = int(records["compressed_size"][j])
            a.add_file_made(paths[j], 8, part[at:at + size], int(member_len), int(records["crc32"][j]))  # 8: Deflate
            at += size
            j += 1
        assert at == len(part), "part %d: %d bytes of members, %d bytes received" % (r, at, len(part))
    assert j == len(paths)
    return a.to_binary_string()

