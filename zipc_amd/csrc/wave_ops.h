// wave_ops.h -- wave64 cooperative primitives shared by the HIP kernels (device only).
#pragma once

#include "zd_common.h"

namespace zd {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte accesses at any byte alignment (gfx950 global memory allows them;
// these compile to one global_load/store_dwordx4)
__device__ __forceinline__ u32x4 load16_unaligned(const uint8_t *p) {
  u32x4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
__device__ __forceinline__ void store16_unaligned(uint8_t *p, u32x4 v) { __builtin_memcpy(p, &v, 16); }

// all 64 lanes copy len bytes, 16 B per lane per step (coalesced 1 KiB rows)
__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t len, int lane) {
  const uint32_t body = len & ~15u;
  for (uint32_t i = (uint32_t)lane * 16u; i < body; i += 1024u)
    store16_unaligned(dst + i, load16_unaligned(src + i));
  if ((uint32_t)lane < (len & 15u)) dst[body + lane] = src[body + lane];
}

// inclusive wave scan: Kogge-Stone inside the rows of 16 on DPP row shifts,
// then the row totals broadcast down (row_bcast:15 to rows 1 and 3,
// row_bcast:31 to rows 2 and 3).
// (bound_ctrl on the row shifts -- a lane without a source reads 0 -- is what lets the compiler fold each step into ONE
// v_add_u32_dpp: with "old = 0, bound_ctrl off", the same values, it emitted v_mov 0 + v_mov_dpp + v_add per step, 18 vector
// instructions a scan where this is 6; round 6, found in lz_parse's assembly)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return x;
}

// value of lane `addr / 4`
__device__ __forceinline__ uint32_t lane_value(uint32_t addr, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)v);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One Adler chunk c[0, len), len <= 5552, by all 64 lanes:
// S1 = sum b_i, S2 = sum (len - i) b_i  (both exact in u32), reduced over the wave.
__device__ __forceinline__ void wave_adler_chunk_sums(const uint8_t *c, uint32_t len, int lane,
                                                      uint32_t &S1, uint32_t &S2) {
  uint32_t a1 = 0, a2 = 0;
  for (uint32_t o = (uint32_t)lane * 16u; o < len; o += 1024u) {
    uint32_t s = 0, w = 0;
    if (o + 16u <= len) {
      u32x4 v = load16_unaligned(c + o);
      s = __builtin_amdgcn_udot4(v.x, 0x01010101u, s, false);
      s = __builtin_amdgcn_udot4(v.y, 0x01010101u, s, false);
      s = __builtin_amdgcn_udot4(v.z, 0x01010101u, s, false);
      s = __builtin_amdgcn_udot4(v.w, 0x01010101u, s, false);
      w = __builtin_amdgcn_udot4(v.x, 0x03020100u, w, false);
      w = __builtin_amdgcn_udot4(v.y, 0x07060504u, w, false);
      w = __builtin_amdgcn_udot4(v.z, 0x0b0a0908u, w, false);
      w = __builtin_amdgcn_udot4(v.w, 0x0f0e0d0cu, w, false);
    } else {
      for (uint32_t m = 0; o + m < len; m++) { uint32_t b = c[o + m]; s += b; w += m * b; }
    }
    a1 += s;
    a2 += (len - o) * s - w;
  }
  S1 = wave_sum(a1);
  S2 = wave_sum(a2);
}

// Adler_32.string_update (src/zipc_deflate.ml:175-198) over p[0, n) by all 64
// lanes: the FIRST chunk is n mod 5552 bytes (possibly empty), then 5552 each;
// per chunk every lane applies the reference's signed-remainder step.
__device__ __forceinline__ uint32_t wave_adler_update(uint32_t a, const uint8_t *p, uint32_t n, int lane, bool rfc = false) {
  uint32_t s1, s2;
  adler_unpack(a, s1, s2);
  uint32_t start = 0, block_len = n % ADLER_CHUNK;
  while (start < n) {
    uint32_t S1, S2;
    wave_adler_chunk_sums(p + start, block_len, lane, S1, S2);
    adler_chunk_step(s1, s2, block_len, S1, S2, rfc);
    start += block_len;
    block_len = ADLER_CHUNK;
  }
  return adler_pack(s1, s2);
}

}  // namespace zd
