"""Synthetic inputs of the BASELINE.json configs.

Byte generator = splitmix64: one 64-bit draw per 8 output bytes, byte k of a
draw = (draw >> 8k) & mask.  Stream j of config c is seeded
0x5EED_0000_0000_0000 + (c << 32) + j, so any shard can regenerate its own data.
numpy (host) and torch (device) versions produce identical bytes.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x5EED000000000000
_GOLD = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB
_M64 = (1 << 64) - 1


def stream_seed(config: int, j: int) -> int:
    return (SEED_BASE + (config << 32) + j) & _M64


def splitmix64_np(seed: int, n_draws: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = np.arange(1, n_draws + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * np.uint64(_GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def stream_bytes_np(config: int, j: int, n: int, bits: int) -> np.ndarray:
    """n bytes of stream j: i.i.d. uniform `bits`-bit symbols (bits = 8: raw bytes)."""
    draws = splitmix64_np(stream_seed(config, j), (n + 7) // 8)
    b = draws.view(np.uint8)[:n]  # little endian: byte k of a draw is (draw >> 8k) & 0xFF
    if bits < 8:
        b = b & np.uint8((1 << bits) - 1)
    return np.ascontiguousarray(b)


def _to_i64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def batch_bytes_torch(config: int, first_stream: int, n_streams: int, stream_len: int, bits: int,
                      device):
    """uint8 tensor [n_streams * stream_len] on `device`: streams first_stream.. back to back.

    int64 arithmetic wraps like uint64; logical shifts are emulated with masks."""
    import torch

    assert stream_len % 8 == 0
    d = stream_len // 8
    out = torch.empty(n_streams * d, dtype=torch.int64, device=device)
    step = max(1, (1 << 24) // max(d, 1))  # ~16M draws per slab
    gold, c1, c2 = _to_i64(_GOLD), _to_i64(_C1), _to_i64(_C2)
    k = torch.arange(1, d + 1, dtype=torch.int64, device=device) * gold
    for s0 in range(0, n_streams, step):
        s1 = min(n_streams, s0 + step)
        seeds = torch.tensor([_to_i64(stream_seed(config, first_stream + j)) for j in range(s0, s1)],
                             dtype=torch.int64, device=device)
        z = seeds[:, None] + k[None, :]
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * c1
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * c2
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        out[s0 * d:s1 * d] = z.reshape(-1)
    b = out.view(torch.uint8)
    if bits < 8:
        b &= (1 << bits) - 1
    return b
