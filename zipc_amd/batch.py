"""Device-resident batch forms (zipc_hip_inflate_batch / _deflate_batch / _checksum_device).

torch is used only as the owner of device memory (uint8 arenas, descriptor and
result arrays); the work itself is enqueued by libzipc_hip.so on the context's
HIP stream.  Descriptors are numpy structured arrays with the layout of
zipc_hip_stream_desc / zipc_hip_stream_result (include/zipc_hip.h).
"""
from __future__ import annotations

import numpy as np

from ._lib import CRC_NOP, OK, STREAM_HAS_LIMIT, Context, lib

DESC_DTYPE = np.dtype([("src_off", "<u8"), ("src_len", "<u8"), ("dst_off", "<u8"),
                       ("dst_cap", "<u8"), ("limit", "<u8"), ("flags", "<u4"),
                       ("reserved", "<u4")])
RESULT_DTYPE = np.dtype([("status", "<u4"), ("checksum", "<u4"), ("out_len", "<u8")])
assert DESC_DTYPE.itemsize == 48 and RESULT_DTYPE.itemsize == 16


def make_descs(src_off, src_len, dst_off, dst_cap, limit=None) -> np.ndarray:
    n = len(src_off)
    d = np.zeros(n, dtype=DESC_DTYPE)
    d["src_off"], d["src_len"], d["dst_off"], d["dst_cap"] = src_off, src_len, dst_off, dst_cap
    if limit is not None:
        d["limit"] = limit
        d["flags"] = STREAM_HAS_LIMIT
    return d


def to_device(arr: np.ndarray, device):
    import torch

    return torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(device)


def results_from_device(t) -> np.ndarray:
    return t.cpu().numpy().view(RESULT_DTYPE).copy()


def _sync_torch(t):
    import torch

    torch.cuda.current_stream(t.device).synchronize()


def inflate_batch(ctx: Context, src, dst, descs_dev, results_dev, n_streams: int, max_dst_cap: int,
                  crc_op: int = CRC_NOP, sync: bool = True):
    """src/dst: uint8 cuda tensors (arenas); descs_dev/results_dev: uint8 cuda tensors."""
    if sync:
        _sync_torch(src)
    st = lib().zipc_hip_inflate_batch(ctx.handle, src.data_ptr(), dst.data_ptr(), descs_dev.data_ptr(),
                                      results_dev.data_ptr(), n_streams, max_dst_cap, crc_op)
    ctx.check(st)
    if sync:
        ctx.synchronize()


def deflate_batch(ctx: Context, src, dst, descs_dev, results_dev, n_streams: int, max_src_len: int,
                  total_src_len: int, level: int, crc_op: int = CRC_NOP, sync: bool = True):
    if sync:
        _sync_torch(src)
    st = lib().zipc_hip_deflate_batch(ctx.handle, src.data_ptr(), dst.data_ptr(), descs_dev.data_ptr(),
                                      results_dev.data_ptr(), n_streams, max_src_len, total_src_len,
                                      level, crc_op)
    ctx.check(st)
    if sync:
        ctx.synchronize()


def debug_chain_links(ctx: Context, src, descs_dev, n_streams: int, max_src_len: int, total_src_len: int, which: int):
    """Tests only (include/zipc_hip.h zipc_hip_debug_chain_links): the hash-chain links of a device-resident batch as the
    library's chain kernel `which` makes them (0: ordered LDS exchange, 1: the kernel that orders equal hashes itself):
    (links, pos_base) as an int16 and an int64 cuda tensor."""
    import torch

    _sync_torch(src)
    cap = int(lib().zipc_hip_debug_chain_positions(n_streams, total_src_len))
    links = torch.empty(cap, dtype=torch.int16, device=src.device)
    base = torch.empty(n_streams, dtype=torch.int64, device=src.device)
    torch.cuda.current_stream(src.device).synchronize()
    ctx.check(lib().zipc_hip_debug_chain_links(ctx.handle, src.data_ptr(), descs_dev.data_ptr(), n_streams, max_src_len,
                                               total_src_len, which, links.data_ptr(), cap, base.data_ptr()))
    ctx.synchronize()
    return links, base


def checksum_device(ctx: Context, buf, want_crc32=True, want_adler32=True):
    """(crc32, adler32) of a uint8 cuda tensor, computed on the device."""
    import torch

    _sync_torch(buf)
    out = torch.zeros(2, dtype=torch.int32, device=buf.device)
    torch.cuda.current_stream(buf.device).synchronize()
    st = lib().zipc_hip_checksum_device(ctx.handle, buf.data_ptr(), buf.numel(), int(want_crc32),
                                        int(want_adler32), out.data_ptr())
    ctx.check(st)
    ctx.synchronize()
    v = out.cpu().numpy().view(np.uint32)
    return int(v[0]), int(v[1])


def reserve(ctx: Context, n_streams: int, max_src_len: int, total_src_len: int):
    ctx.check(lib().zipc_hip_reserve(ctx.handle, n_streams, max_src_len, total_src_len))


def deflate_bound(n: int) -> int:
    return lib().zipc_hip_deflate_bound(n)


def uniform_layout(n_streams: int, src_len: int, dst_cap: int, limit=None) -> np.ndarray:
    """n equal streams laid back to back in both arenas (dst slots 256-byte aligned)."""
    slot = (dst_cap + 255) // 256 * 256
    i = np.arange(n_streams, dtype=np.uint64)
    return make_descs(i * np.uint64(src_len), np.full(n_streams, src_len, np.uint64),
                      i * np.uint64(slot), np.full(n_streams, dst_cap, np.uint64),
                      None if limit is None else np.full(n_streams, limit, np.uint64))


def compact_descs(results: np.ndarray, descs: np.ndarray, dst_cap_of_out, limit_exact=True) -> np.ndarray:
    """Descriptors for the inverse operation: the outputs of one batch (at their
    dst offsets, with their out_len) become the inputs of the next."""
    n = len(results)
    slot = (int(dst_cap_of_out) + 255) // 256 * 256
    i = np.arange(n, dtype=np.uint64)
    d = make_descs(descs["dst_off"], results["out_len"], i * np.uint64(slot),
                   np.full(n, dst_cap_of_out, np.uint64),
                   descs["src_len"] if limit_exact else None)
    return d


__all__ = ["DESC_DTYPE", "RESULT_DTYPE", "make_descs", "to_device", "results_from_device",
           "inflate_batch", "deflate_batch", "checksum_device", "reserve", "deflate_bound",
           "uniform_layout", "compact_descs", "OK"]
