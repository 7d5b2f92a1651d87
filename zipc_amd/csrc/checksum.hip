// checksum.hip -- CRC-32 and Adler-32 kernels for gfx950.
//
// CRC-32 (Crc_32.string_update, src/zipc_deflate.ml:137-156): the reference's
// slice-by-4 table walk is a serial chain over the bytes.  Here a range is cut
// into 64 KiB segments (one 256-thread workgroup each) and every thread walks a
// 256-byte piece with the same 4x256 tables held in LDS; pieces and segments
// are merged with the CRC combination rule (zd_common.h: gf2_mul), the range
// being RIGHT-aligned on the piece grid so that all pieces have equal length
// (leading zero bytes do not change a raw CRC).  Chaining across the
// reference's per-block update calls is exact for CRC-32, so the fused forms
// (inflate_and_crc_32, crc_32_and_deflate) run this once over the whole
// produced / consumed range of each stream.
//
// Adler-32 (Adler_32.string_update, src/zipc_deflate.ml:175-198): one wave per
// 5552-byte chunk of the reference's chunk grid (FIRST chunk = len mod 5552)
// reduces (S1, S2); the chunk chain with the signed 32-bit remainder
// (src/zipc_deflate.ml:95,196) is then applied in order by adler_chain_kernel.
#include "kernels.h"
#include "wave_ops.h"

namespace zd {

constexpr uint32_t CRC_PIECE = 256;                // bytes per thread
constexpr uint32_t CRC_THREADS = 256;
constexpr uint32_t CRC_SEG = CRC_PIECE * CRC_THREADS;  // 64 KiB per workgroup
static_assert(CRC_SEG == CRC_SEG_BYTES, "kernels.h");

__device__ __forceinline__ void get_range(int mode, uint32_t i, const StreamDesc *descs,
                                          const StreamResult *results, uint64_t single_off,
                                          uint64_t single_len, uint64_t &off, uint64_t &len) {
  if (mode == RANGE_SINGLE) { off = single_off; len = single_len; return; }
  const StreamDesc d = descs[i];
  if (mode == RANGE_DEFLATE_SRC) { off = d.src_off; len = d.src_len; return; }
  const StreamResult r = results[i];
  off = d.dst_off;
  len = r.status == ST_OK ? r.out_len : 0;
}

// Crc_32.table (src/zipc_deflate.ml:114-133) built in LDS by the workgroup
__device__ __forceinline__ void build_crc_tables(uint32_t (*T)[256], int t) {
  uint32_t c = (uint32_t)t;
#pragma unroll
  for (int k = 0; k < 8; k++) c = (c & 1u) ? (CRC_POLY ^ (c >> 1)) : (c >> 1);
  T[0][t] = c;
  __syncthreads();
  uint32_t v = c;
#pragma unroll
  for (int k = 1; k < 4; k++) {
    v = (v >> 8) ^ T[0][v & 0xFF];
    T[k][t] = v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(CRC_THREADS) void crc32_segments_kernel(
    const uint8_t *__restrict__ base, int mode, const StreamDesc *__restrict__ descs,
    const StreamResult *__restrict__ results, uint64_t single_off, uint64_t single_len,
    uint32_t segs_per_range, CrcConsts K, uint32_t *__restrict__ partials) {
  __shared__ uint32_t T[4][256];
  __shared__ uint32_t wave_part[CRC_THREADS / 64];
  const int t = threadIdx.x;
  const uint32_t range = blockIdx.x / segs_per_range;
  const uint32_t seg = blockIdx.x % segs_per_range;
  uint64_t off, len;
  get_range(mode, range, descs, results, single_off, single_len, off, len);
  const uint64_t nseg = (len + CRC_SEG - 1) / CRC_SEG;
  if (seg >= nseg) return;  // uniform per workgroup
  build_crc_tables(T, t);

  // right-aligned piece grid: `pad` virtual zero bytes in front of the range
  const uint64_t pad = nseg * CRC_SEG - len;
  const int64_t p0 = (int64_t)((uint64_t)seg * CRC_SEG + (uint64_t)t * CRC_PIECE) - (int64_t)pad;
  const int64_t lo = p0 < 0 ? 0 : p0;
  const int64_t hi = p0 + (int64_t)CRC_PIECE;  // <= len by construction
  const uint8_t *p = base + off;
  uint32_t c = 0;
  int64_t i = lo;
  if (hi > lo) {
    // word loop (src/zipc_deflate.ml:141-150)
    for (; i + 4 <= hi; i += 4) {
      uint32_t u = c ^ load_u32_le(p + i);
      c = T[3][u & 0xFF] ^ T[2][(u >> 8) & 0xFF] ^ T[1][(u >> 16) & 0xFF] ^ T[0][u >> 24];
    }
    for (; i < hi; i++) c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xFF];  // byte tail (:151-155)
  }
  // merge the 256 equal-length pieces: tree over lanes, then over waves
#pragma unroll
  for (int k = 0; k < 6; k++) {
    uint32_t other = __shfl_down(c, 1u << k, 64);
    if ((t & ((2 << k) - 1)) == 0) c = gf2_mul(c, K.xpiece[k]) ^ other;
  }
  if ((t & 63) == 0) wave_part[t >> 6] = c;
  __syncthreads();
  if (t == 0) {
    uint32_t a = gf2_mul(wave_part[0], K.xpiece[6]) ^ wave_part[1];
    uint32_t b = gf2_mul(wave_part[2], K.xpiece[6]) ^ wave_part[3];
    partials[(uint64_t)range * segs_per_range + seg] = gf2_mul(a, K.xpiece[7]) ^ b;
  }
}

// One workgroup per range: folds the segment partials (Horner runs per thread,
// then a tree), applies init/finish (src/zipc_deflate.ml:135-136) and stores
// the checksum.
__global__ __launch_bounds__(256) void crc32_finish_kernel(
    int mode, const StreamDesc *__restrict__ descs, StreamResult *__restrict__ results,
    uint64_t single_len, uint32_t segs_per_range, CrcConsts K,
    const uint32_t *__restrict__ partials, uint32_t *__restrict__ single_out) {
  __shared__ uint32_t sh[256];
  const int t = threadIdx.x;
  const uint32_t range = blockIdx.x;
  uint64_t off, len;
  get_range(mode, range, descs, results, 0, single_len, off, len);
  if (mode == RANGE_INFLATE_OUT && results[range].status != ST_OK) return;
  const uint64_t nseg = (len + CRC_SEG - 1) / CRC_SEG;
  const uint32_t *P = partials + (uint64_t)range * segs_per_range;
  uint32_t raw = 0;
  if (nseg <= 1) {
    raw = nseg ? P[0] : 0;
  } else {
    // right-aligned grid of 256 runs of R partials each
    const uint64_t R = (nseg + 255) / 256;
    const uint64_t padp = 256 * R - nseg;
    uint32_t c = 0;
    for (uint64_t j = 0; j < R; j++) {
      const int64_t idx = (int64_t)((uint64_t)t * R + j) - (int64_t)padp;
      const uint32_t v = idx >= 0 ? P[idx] : 0u;
      c = gf2_mul(c, K.xseg) ^ v;
    }
    sh[t] = c;
    __syncthreads();
    uint32_t xr = gf2_xpow8n((uint64_t)CRC_SEG * R);  // shift of one run
    for (int s = 1; s < 256; s <<= 1) {
      if ((t & (2 * s - 1)) == 0) sh[t] = gf2_mul(sh[t], xr) ^ sh[t + s];
      xr = gf2_mul(xr, xr);
      __syncthreads();
    }
    raw = sh[0];
  }
  if (t == 0) {
    const uint32_t state = crc_state_advance(0xFFFFFFFFu, raw, gf2_xpow8n(len));
    const uint32_t crc = state ^ 0xFFFFFFFFu;
    if (mode == RANGE_SINGLE) single_out[0] = crc;
    else results[range].checksum = crc;
  }
}

// ---- Adler-32 over one buffer ---------------------------------------------------

// chunk k of the reference's grid for a buffer of n bytes: chunk 0 is
// [0, n mod 5552) (possibly empty), chunk k >= 1 is 5552 bytes
__global__ __launch_bounds__(256) void adler_chunks_kernel(const uint8_t *__restrict__ p, uint64_t n,
                                                           uint64_t n_chunks,
                                                           uint2 *__restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= n_chunks) return;
  const uint64_t r = n % ADLER_CHUNK;
  const uint64_t start = chunk == 0 ? 0 : r + (chunk - 1) * ADLER_CHUNK;
  const uint32_t len = chunk == 0 ? (uint32_t)r : ADLER_CHUNK;
  uint32_t S1, S2;
  wave_adler_chunk_sums(p + start, len, lane, S1, S2);
  if (lane == 0) sums[chunk] = make_uint2(S1, S2);
}

// the chunk chain, in order (src/zipc_deflate.ml:196): s1/s2 := signed rem
__global__ void adler_chain_kernel(const uint2 *__restrict__ sums, uint64_t n, uint64_t n_chunks,
                                   uint32_t *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t s1, s2;
  adler_unpack(1u, s1, s2);  // Adler_32.init
  const uint32_t r = (uint32_t)(n % ADLER_CHUNK);
  for (uint64_t k = 0; k < n_chunks; k++) {
    const uint2 s = sums[k];
    adler_chunk_step(s1, s2, k == 0 ? r : ADLER_CHUNK, s.x, s.y);
  }
  out[0] = adler_pack(s1, s2);
}

}  // namespace zd
