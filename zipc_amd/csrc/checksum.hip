// checksum.hip -- CRC-32 and Adler-32 kernels for gfx950.
//
// CRC-32 (Crc_32.string_update, src/zipc_deflate.ml:137-156): the reference's
// slice-by-4 table walk is a serial chain over the bytes.  Here a range is cut
// into 32 KiB segments (one 256-thread workgroup each) and every thread walks a
// 128-byte piece with the same 4x256 tables held in LDS; pieces and segments
// are merged with the CRC combination rule (zd_common.h: gf2_mul), the range
// being RIGHT-aligned on the piece grid so that all pieces have equal length
// (leading zero bytes do not change a raw CRC).  A segment goes through LDS on
// its way to the threads: it is read from memory as coalesced 16-byte units
// (every byte fetched once) and written so that piece t starts at 144 t -- 16
// bytes of padding per piece make the threads' 16-byte reads of their own
// pieces conflict free.  (Threads reading their pieces straight from memory,
// 64 lanes x 16 B at a 256 B stride, re-fetched every line several times: the
// lines in flight on a CU did not fit the caches.)  Chaining across the
// reference's per-block update calls is exact for CRC-32, so the fused forms
// (inflate_and_crc_32, crc_32_and_deflate) run this once over the whole
// produced / consumed range of each stream.
//
// Adler-32 (Adler_32.string_update, src/zipc_deflate.ml:175-198): one wave per
// 5552-byte chunk of the reference's chunk grid (FIRST chunk = len mod 5552)
// reduces (S1, S2); the chunk chain with the signed 32-bit remainder
// (src/zipc_deflate.ml:95,196) is then applied in order by the adler_runs / adler_scan /
// adler_replay kernels.
#include "kernels.h"
#include "wave_ops.h"

namespace zd {

constexpr uint32_t CRC_PIECE = CRC_PIECE_BYTES;      // bytes per thread
constexpr uint32_t CRC_THREADS = 256;
constexpr uint32_t CRC_SEG = CRC_PIECE * CRC_THREADS;  // 32 KiB per workgroup
constexpr uint32_t CRC_PIECE_STRIDE = CRC_PIECE + 16;  // of a piece in LDS
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

// ADLER: the same pass also leaves the Adler-32 chunk sums of the range (RANGE_SINGLE only): the chunk grid
// (first chunk = len mod 5552, then 5552 each) and the right-aligned piece grid agree mod 16 -- 5552 = 16 * 347
// and both are congruent to len -- so a chunk boundary falls between two 16-byte units of a thread's piece,
// never inside one.  A thread sums its 8 units (S1 = sum b, S2 = sum (end - i) b, two dot products per word),
// splits them at the boundary if its piece holds one, and the wave adds what it has of at most three chunks
// to `adler_sums` (zeroed by the caller; integer adds: any order gives the same sums).
template <bool ADLER>
__device__ __forceinline__ void crc32_segment(
    const uint8_t *__restrict__ base, int mode, const StreamDesc *__restrict__ descs,
    const StreamResult *__restrict__ results, uint64_t single_off, uint64_t single_len,
    uint32_t segs_per_range, const uint32_t *__restrict__ nib, uint32_t *__restrict__ partials,
    uint32_t *__restrict__ adler_sums, uint64_t n_chunks) {
  __shared__ uint32_t T[4][256];
  static_assert(8 * GF2_NIB_WORDS == 4 * CRC_THREADS, "one 16-byte load per thread");
  __shared__ __attribute__((aligned(16))) uint8_t stage[CRC_THREADS * CRC_PIECE_STRIDE];
  // xpiece[0..7] as nibble tables (4 KiB) and the waves' partials: in the staged segment's place once every thread has walked
  // its piece -- 40 KiB of LDS a workgroup, so that FOUR fit a CU (three with tables of their own: the table walk of one
  // workgroup hides behind the loads of the others, and C3 is the one kernel here that waits for memory)
  uint32_t *N = (uint32_t *)stage;
  uint32_t *wave_part = (uint32_t *)(stage + 8 * GF2_NIB_WORDS * 4);
  const int t = threadIdx.x;
  const uint32_t range = blockIdx.x / segs_per_range;
  const uint32_t seg = blockIdx.x % segs_per_range;
  uint64_t off, len;
  get_range(mode, range, descs, results, single_off, single_len, off, len);
  const uint64_t nseg = (len + CRC_SEG - 1) / CRC_SEG;
  if (seg >= nseg) return;  // uniform per workgroup
  build_crc_tables(T, t);

  // right-aligned piece grid: `pad` virtual zero bytes in front of the range;
  // seg0 = range position of the segment's first byte (negative inside the pad)
  const uint64_t pad = nseg * CRC_SEG - len;
  const int64_t seg0 = (int64_t)((uint64_t)seg * CRC_SEG) - (int64_t)pad;
  const uint8_t *p = base + off;
  constexpr int UNITS = CRC_SEG / 16 / CRC_THREADS;  // 16-byte units per thread
  u32x4 v[UNITS];
  if (len >= 16) {
    // unit u = i * 256 + t: consecutive threads, consecutive 16 bytes; all loads in flight together
#pragma unroll
    for (int i = 0; i < UNITS; i++) {
      const int64_t q = seg0 + (int64_t)((uint32_t)i * CRC_THREADS + (uint32_t)t) * 16;
      v[i] = load16_unaligned(p + (q > 0 ? q : 0));  // q + 16 <= len by construction
    }
#pragma unroll
    for (int i = 0; i < UNITS; i++) {
      const int64_t q = seg0 + (int64_t)((uint32_t)i * CRC_THREADS + (uint32_t)t) * 16;
      if (q < 0) {  // only in the first segment of a range: pad bytes are zero
        uint8_t b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = q + k >= 0 ? p[q + k] : (uint8_t)0;
        __builtin_memcpy(&v[i], b, 16);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < UNITS; i++) {
      const int64_t q = seg0 + (int64_t)((uint32_t)i * CRC_THREADS + (uint32_t)t) * 16;
      uint8_t b[16];
#pragma unroll
      for (int k = 0; k < 16; k++) b[k] = (q + k >= 0 && q + k < (int64_t)len) ? p[q + k] : (uint8_t)0;
      __builtin_memcpy(&v[i], b, 16);
    }
  }
#pragma unroll
  for (int i = 0; i < UNITS; i++) {
    const uint32_t u = (uint32_t)i * CRC_THREADS + (uint32_t)t;
    *(u32x4 *)(stage + (u / (CRC_PIECE / 16)) * CRC_PIECE_STRIDE + (u % (CRC_PIECE / 16)) * 16) = v[i];
  }
  __syncthreads();
  // my piece (Crc_32.string_update's word loop, src/zipc_deflate.ml:141-150; leading
  // zero bytes leave c = 0)
  uint32_t c = 0;
  const u32x4 *mine = (const u32x4 *)(stage + (uint32_t)t * CRC_PIECE_STRIDE);
  // Adler: the chunk that holds the piece's first byte, and the unit at which the next chunk starts (8: none)
  uint32_t cut = 8, s1 = 0, s2 = 0, A1 = 0, A2 = 0;
  uint64_t kA = 0;
  int64_t to_end = 0;  // from the piece's start to the end of chunk kA: a positive multiple of 16
  if constexpr (ADLER) {
    const int64_t ps = seg0 + (int64_t)((uint32_t)t * CRC_PIECE);
    const uint64_t r = len % ADLER_CHUNK;
    const uint64_t pos0 = ps > 0 ? (uint64_t)ps : 0;
    uint64_t eA = r;
    if (pos0 >= r) { kA = (pos0 - r) / ADLER_CHUNK + 1; eA = r + kA * ADLER_CHUNK; }
    to_end = (int64_t)eA - ps;
    cut = to_end < (int64_t)CRC_PIECE ? (uint32_t)to_end / 16u : 8u;
  }
#pragma unroll
  for (int j = 0; j < (int)(CRC_PIECE / 16); j++) {
    const u32x4 w = mine[j];
    if constexpr (ADLER) {
      A1 = cut == (uint32_t)j ? s1 : A1;  // the sums of the units before the boundary
      A2 = cut == (uint32_t)j ? s2 : A2;
      uint32_t u1 = __builtin_amdgcn_udot4(w.x, 0x01010101u, 0u, false);
      u1 = __builtin_amdgcn_udot4(w.y, 0x01010101u, u1, false);
      u1 = __builtin_amdgcn_udot4(w.z, 0x01010101u, u1, false);
      u1 = __builtin_amdgcn_udot4(w.w, 0x01010101u, u1, false);
      uint32_t u2 = __builtin_amdgcn_udot4(w.x, 0x0D0E0F10u, 0u, false);  // byte i of the unit weighs 16 - i
      u2 = __builtin_amdgcn_udot4(w.y, 0x090A0B0Cu, u2, false);
      u2 = __builtin_amdgcn_udot4(w.z, 0x05060708u, u2, false);
      u2 = __builtin_amdgcn_udot4(w.w, 0x01020304u, u2, false);
      s2 += 16u * s1 + u2;
      s1 += u1;
    }
#ifdef ZD_CRC_FAKE  // timing only (wrong checksums): what the pass costs without the table walk
    c ^= w.x ^ w.y ^ w.z ^ w.w;
#else
    uint32_t u = c ^ w.x;
    c = T[3][u & 0xFF] ^ T[2][(u >> 8) & 0xFF] ^ T[1][(u >> 16) & 0xFF] ^ T[0][u >> 24];
    u = c ^ w.y;
    c = T[3][u & 0xFF] ^ T[2][(u >> 8) & 0xFF] ^ T[1][(u >> 16) & 0xFF] ^ T[0][u >> 24];
    u = c ^ w.z;
    c = T[3][u & 0xFF] ^ T[2][(u >> 8) & 0xFF] ^ T[1][(u >> 16) & 0xFF] ^ T[0][u >> 24];
    u = c ^ w.w;
    c = T[3][u & 0xFF] ^ T[2][(u >> 8) & 0xFF] ^ T[1][(u >> 16) & 0xFF] ^ T[0][u >> 24];
#endif
  }
  // merge the 256 equal-length pieces: tree over lanes, then over waves.  The
  // multipliers are constants: nibble-table products (the bit-serial gf2_mul here was
  // two thirds of the kernel's instructions)
  const u32x4 nib_mine = ((const u32x4 *)nib)[t];  // (requested here, not beside the data: four more registers held through the walk spill others)
  __syncthreads();  // every piece has been read: the segment's place takes the tables
  ((u32x4 *)N)[t] = nib_mine;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const uint32_t other = __shfl_down(c, 1u << k, 64);
    c = gf2_mul_nib(c, N + k * GF2_NIB_WORDS) ^ other;  // only lanes with (t & ((2 << k) - 1)) == 0 are used further
  }
  if ((t & 63) == 0) wave_part[t >> 6] = c;
  __syncthreads();
  if (t == 0) {
    const uint32_t a = gf2_mul_nib(wave_part[0], N + 6 * GF2_NIB_WORDS) ^ wave_part[1];
    const uint32_t b = gf2_mul_nib(wave_part[2], N + 6 * GF2_NIB_WORDS) ^ wave_part[3];
    partials[(uint64_t)range * segs_per_range + seg] = gf2_mul_nib(a, N + 7 * GF2_NIB_WORDS) ^ b;
  }
  if constexpr (ADLER) {
    if (cut == 8u) { A1 = s1; A2 = s2; }
    // what the piece adds to chunk kA (units before the boundary) and to chunk kA + 1 (the rest): S2 counts a
    // byte by its distance to the end of ITS CHUNK, the piece sums count it to the end of the part
    const uint32_t B1 = s1 - A1, B2 = s2 - A2 - (CRC_PIECE - 16u * cut) * A1;
    const uint32_t a2 = A2 + (cut == 8u ? (uint32_t)(to_end - (int64_t)CRC_PIECE) * A1 : 0u);
    const uint32_t b2 = B2 + ((uint32_t)to_end + ADLER_CHUNK - CRC_PIECE) * B1;  // (B1 = 0 when there is no boundary)
    // the wave's 8 KiB touch at most three chunks: three sums of each kind, lane 63 of a scan holds the total
    const uint64_t k0 = ((uint64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kA >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kA);
    const uint32_t dk = (uint32_t)(kA - k0);
    uint32_t tot[6];
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
      const uint32_t v1 = (dk == k ? A1 : 0u) + (dk + 1u == k ? B1 : 0u);
      const uint32_t v2 = (dk == k ? a2 : 0u) + (dk + 1u == k ? b2 : 0u);
      tot[2 * k] = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v1), 63);
      tot[2 * k + 1] = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v2), 63);
    }
    const uint32_t ln = (uint32_t)t & 63u;
    uint32_t mine_v = tot[0];
#pragma unroll
    for (uint32_t m = 1; m < 6; m++) mine_v = ln == m ? tot[m] : mine_v;
    const uint64_t kc = k0 + (ln >> 1);
    if (ln < 6u && kc < n_chunks && mine_v) atomicAdd(adler_sums + 2 * kc + (ln & 1u), mine_v);
  }
}

__global__ __launch_bounds__(CRC_THREADS, 4) void crc32_segments_kernel(
    const uint8_t *__restrict__ base, int mode, const StreamDesc *__restrict__ descs,
    const StreamResult *__restrict__ results, uint64_t single_off, uint64_t single_len,
    uint32_t segs_per_range, const uint32_t *__restrict__ nib, uint32_t *__restrict__ partials) {
  crc32_segment<false>(base, mode, descs, results, single_off, single_len, segs_per_range, nib, partials, nullptr, 0);
}

// CRC-32 partials and Adler-32 chunk sums of ONE buffer in one pass over its bytes
__global__ __launch_bounds__(CRC_THREADS, 4) void crc32_adler_segments_kernel(
    const uint8_t *__restrict__ p, uint64_t len, uint32_t n_segs, const uint32_t *__restrict__ nib,
    uint32_t *__restrict__ partials, uint2 *__restrict__ adler_sums, uint64_t n_chunks) {
  crc32_segment<true>(p, RANGE_SINGLE, nullptr, nullptr, 0, len, n_segs, nib, partials, (uint32_t *)adler_sums, n_chunks);
}

// One workgroup per range: folds the segment partials (Horner runs per thread,
// then a tree), applies init/finish (src/zipc_deflate.ml:135-136) and stores
// the checksum.
// x^(8 * nbytes) mod P from the context's table of x^(8 * 2^k): one multiply per set
// bit of nbytes (gf2_xpow8n squares once per bit position on top of that)
__device__ __forceinline__ uint32_t xpow8n_tab(const CrcConsts &K, uint64_t nbytes) {
  uint32_t r = 0x80000000u;  // x^0
  for (int k = 0; k < 48 && nbytes; k++, nbytes >>= 1)
    if (nbytes & 1) r = gf2_mul(r, K.xbyte[k]);
  if (nbytes) {  // beyond 2^48 bytes: keep squaring
    uint32_t sq = gf2_mul(K.xbyte[47], K.xbyte[47]);
    while (nbytes) {
      if (nbytes & 1) r = gf2_mul(r, sq);
      sq = gf2_mul(sq, sq);
      nbytes >>= 1;
    }
  }
  return r;
}

// The finish of a batch of SHORT ranges (at most 16 segments each, the batch forms'
// streams): one range per thread.  (One workgroup per range left 255 threads waiting
// for thread 0's ~20 dependent gf2_mul: 0.19 ms for C2's 16 384 streams.)
__global__ __launch_bounds__(256) void crc32_finish_streams_kernel(
    int mode, const StreamDesc *__restrict__ descs, StreamResult *__restrict__ results, uint32_t n_ranges,
    uint32_t segs_per_range, CrcConsts K, const uint32_t *__restrict__ partials) {
  const uint32_t range = blockIdx.x * 256u + threadIdx.x;
  if (range >= n_ranges) return;
  uint64_t off, len;
  get_range(mode, range, descs, results, 0, 0, off, len);
  if (mode == RANGE_INFLATE_OUT && results[range].status != ST_OK) return;
  const uint64_t nseg = (len + CRC_SEG - 1) / CRC_SEG;
  if (nseg > segs_per_range) {  // longer than the caller declared (max_src_len / max_dst_cap): only part of it was summed
    results[range].status = ST_INVALID_ARG;
    results[range].checksum = 0;
    return;
  }
  const uint32_t *P = partials + (uint64_t)range * segs_per_range;
  uint32_t raw = 0;
  for (uint32_t j = 0; j < (uint32_t)nseg; j++) raw = gf2_mul(raw, K.xseg) ^ P[j];
  const uint32_t state = crc_state_advance(0xFFFFFFFFu, raw, xpow8n_tab(K, len));
  results[range].checksum = state ^ 0xFFFFFFFFu;
}

// (NT = blockDim.x threads, a power of two up to 1024: the launch takes 1024 for ranges of many segments -- C3's 131 072
// partials as 256 Horner runs of 512 dependent steps took 0.118 ms, a tenth of the pass over the 4 GiB themselves)
__global__ __launch_bounds__(1024) void crc32_finish_kernel(
    int mode, const StreamDesc *__restrict__ descs, StreamResult *__restrict__ results,
    uint64_t single_len, uint32_t segs_per_range, CrcConsts K, const uint32_t *__restrict__ nib,
    const uint32_t *__restrict__ partials, uint32_t *__restrict__ single_out) {
  __shared__ uint32_t sh[1024];
  const uint32_t NT = blockDim.x;
  __shared__ uint32_t NS[GF2_NIB_WORDS];  // xseg as a nibble table
  const int t = threadIdx.x;
  const uint32_t range = blockIdx.x;
  uint64_t off, len;
  get_range(mode, range, descs, results, 0, single_len, off, len);
  if (mode == RANGE_INFLATE_OUT && results[range].status != ST_OK) return;
  const uint64_t nseg = (len + CRC_SEG - 1) / CRC_SEG;
  if (nseg > segs_per_range) {  // see crc32_finish_streams_kernel (uniform per workgroup)
    if (t == 0 && mode != RANGE_SINGLE) { results[range].status = ST_INVALID_ARG; results[range].checksum = 0; }
    return;
  }
  const uint32_t *P = partials + (uint64_t)range * segs_per_range;
  uint32_t raw = 0;
  if (nseg <= 1) {
    raw = nseg ? P[0] : 0;
  } else if (nseg <= 16) {  // short ranges (the batch forms' streams): Horner by the one thread that stores
    if (t == 0)
      for (uint32_t j = 0; j < (uint32_t)nseg; j++) raw = gf2_mul(raw, K.xseg) ^ P[j];
  } else {
    // right-aligned grid of R rows of NT partials: thread t takes column t -- consecutive threads, consecutive words of every row
    // (a run of consecutive partials per thread read a cache line per thread and step: C3's finish took 0.11 ms, a tenth
    // of the pass over the 4 GiB themselves) -- so its Horner step is a shift by a whole ROW, x^(8 * segment * NT), whose
    // nibble table the workgroup makes first; the tree then shifts by a segment, two, four ...
    const uint64_t R = (nseg + NT - 1) / NT;
    const uint64_t padp = NT * R - nseg;
    const uint32_t xrow = xpow8n_tab(K, (uint64_t)CRC_SEG * NT);
    if (t < GF2_NIB_WORDS) NS[t] = gf2_mul(((uint32_t)t & 15u) << (4 * (t >> 4)), xrow);  // (gf2_nib_table's entries, one per thread)
    __syncthreads();
    uint32_t c = 0;
    for (uint64_t j0 = 0; j0 < R; j0 += 8) {  // eight rows' words requested together (clamped index, value selected afterwards: one
      uint32_t v[8];                          // load at a time behind a branch cost a memory round trip per row)
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int64_t idx = (int64_t)((j0 + (uint64_t)u) * NT + (uint64_t)t) - (int64_t)padp;
        const uint32_t w = P[idx >= 0 && j0 + (uint64_t)u < R ? idx : 0];
        v[u] = idx >= 0 ? w : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (j0 + (uint64_t)u < R) c = gf2_mul_nib(c, NS) ^ v[u];
    }
    sh[t] = c;
    __syncthreads();
    uint32_t xr = K.xseg;  // shift of one segment
    for (int s = 1; s < (int)NT; s <<= 1) {
      if ((t & (2 * s - 1)) == 0) sh[t] = gf2_mul(sh[t], xr) ^ sh[t + s];
      xr = gf2_mul(xr, xr);
      __syncthreads();
    }
    raw = sh[0];
  }
  if (t == 0) {
    const uint32_t state = crc_state_advance(0xFFFFFFFFu, raw, xpow8n_tab(K, len));
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

// RFC 1950's Adler-32 of the buffer from its chunk sums: with the unsigned remainder the chunk
// steps are an affine map mod 65521 in (s1, s2), so chunks combine in any grouping: every thread
// folds a run of chunks starting from (0, 0), then the runs are chained by one thread
// (s1' = s1 + A1, s2' = s2 + bytes_of_run * s1 + A2 for a run that maps (0, 0) to (A1, A2)).
__global__ __launch_bounds__(1024) void adler_rfc_finish_kernel(const uint2 *__restrict__ sums, uint64_t n,
                                                                uint64_t n_chunks, uint32_t *__restrict__ out) {
  __shared__ uint32_t a1[1024], a2[1024];
  __shared__ uint64_t nb[1024];
  const uint32_t t = threadIdx.x;
  const uint64_t per = (n_chunks + 1023) / 1024;
  const uint64_t lo = (uint64_t)t * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
  const uint64_t r = n % ADLER_CHUNK;
  uint32_t s1 = 0, s2 = 0;
  uint64_t bytes = 0;
  for (uint64_t k = lo; k < hi; k++) {
    const uint32_t len = k == 0 ? (uint32_t)r : ADLER_CHUNK;
    adler_chunk_step(s1, s2, len, sums[k].x, sums[k].y, true);
    bytes += len;
  }
  a1[t] = s1; a2[t] = s2; nb[t] = bytes;
  __syncthreads();
  if (t == 0) {
    uint64_t c1 = 1, c2 = 0;  // Adler-32 starts at (1, 0)
    for (int i = 0; i < 1024; i++) {
      c2 = (c2 + (nb[i] % ADLER_BASE) * c1 + a2[i]) % ADLER_BASE;
      c1 = (c1 + a1[i]) % ADLER_BASE;
    }
    out[0] = (uint32_t)((c2 << 16) | c1);
  }
}

// The chunk chain (src/zipc_deflate.ml:196: s1/s2 := SIGNED rem after every chunk)
// for hundreds of thousands of chunks, without walking them one by one.
//   s1 never goes negative, so s1 before chunk k is (1 + sum of S1) mod 65521: a scan.
//   s2: with x = s2 before the chunk (|x| < 65521) the reference computes
//   srem32(wrap32(x + C)), C = n * s1 + S2 < 2^32.  Unless C lies within 65521 of 0
//   or of 2^31 the branch taken depends on C alone: below 2^31 the result is
//   (x + C) mod p >= 0, above it is congruent to x + C - 225 (2^32 mod 65521 = 225)
//   with a non-positive representative.  So the residues follow from a second scan
//   of a_k = C_k - 225 * hi_k; the few chunks whose branch does depend on x
//   ("ambiguous", about 6e-5 of them on random data) are then replayed exactly, in
//   order, by one thread, each replay shifting all later residues by a constant.
constexpr uint32_t CHAIN_THREADS_A = 1024;

// (a + b) mod p for a, b <= p
__device__ __forceinline__ uint32_t addmod(uint32_t a, uint32_t b) {
  const uint32_t s = a + b;
  return s >= ADLER_BASE ? s - ADLER_BASE : s;
}

__device__ __forceinline__ uint32_t block_excl_scan_mod(uint32_t v, uint32_t *sh, int t) {
  // exclusive scan of per-thread values (each < 65521) over the 1024 threads, mod p
  sh[t] = v;
  __syncthreads();
  for (int o = 1; o < (int)CHAIN_THREADS_A; o <<= 1) {
    const uint32_t u = t >= o ? sh[t - o] : 0;
    __syncthreads();
    sh[t] = addmod(sh[t], u);
    __syncthreads();
  }
  const uint32_t incl = sh[t];
  __syncthreads();
  return incl >= v ? incl - v : incl + ADLER_BASE - v;
}

// what pass 2 already knows about an ambiguous chunk, kept for the replay
struct AmbRecord {
  uint32_t k;          // chunk index
  uint32_t s1;         // s1 before the chunk
  uint32_t a_partial;  // sum of a_j over the chunks of its run before it (mod p)
  uint32_t prev_hi;    // branch of chunk k-1 as C_{k-1} alone decides it
};
static_assert(sizeof(AmbRecord) == 16, "kernels.h sizes the list with 16 bytes per entry");

// The chain runs as five small launches over R runs of `per` consecutive chunks
// (R a multiple of 1024, up to 64 Ki: a thread per run, hundreds of workgroups):
//   adler_runs_s1    sum of S1 per run
//   adler_scan_runs  exclusive scan (mod p) of the run sums by one workgroup -> s1 before each run
//   adler_runs_a     a_k summed per run from that s1; ambiguous chunks recorded
//   adler_scan_runs  scan of the a sums -> residue of s2 before each run
//   adler_replay     one thread: ambiguous chunks replayed in order, final value
// AdlerRuns: the per-run arrays in device scratch.
__global__ __launch_bounds__(256) void adler_runs_s1_kernel(const uint2 *__restrict__ sums, uint64_t n_chunks,
                                                            uint64_t per, AdlerRuns R) {
  const uint64_t run = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (run >= R.n_runs) return;
  const uint64_t lo = run * per < n_chunks ? run * per : n_chunks;
  const uint64_t hi = lo + per < n_chunks ? lo + per : n_chunks;
  uint64_t acc = 0;
  for (uint64_t k = lo; k < hi; k++) acc += sums[k].x;
  R.sum[run] = (uint32_t)(acc % ADLER_BASE);
}

// out[run] = (first + exclusive prefix of in[0 .. run)) mod p; the inputs are < p.  A thread owns `each`
// consecutive runs (1, 2, 4 .. 64 of them: 16-byte loads from 4 on) and everything is 32-bit: a run of 64
// values sums below 2^22, and a sum of two residues needs one conditional subtraction.
__global__ __launch_bounds__(CHAIN_THREADS_A) void adler_scan_runs_kernel(const uint32_t *__restrict__ in,
                                                                          uint32_t *__restrict__ out,
                                                                          uint32_t n_runs, uint32_t first) {
  __shared__ uint32_t sh[CHAIN_THREADS_A];
  const int t = threadIdx.x;
  const uint32_t each = n_runs / CHAIN_THREADS_A;  // n_runs is 1024 times a power of two
  const uint32_t *mine = in + (uint32_t)t * each;
  uint32_t *mine_out = out + (uint32_t)t * each;
  uint32_t acc = 0;
  if (each >= 4) {
    for (uint32_t j = 0; j < each; j += 4) {
      const u32x4 v = *(const u32x4 *)(mine + j);
      acc += v.x + v.y + v.z + v.w;
    }
  } else {
    for (uint32_t j = 0; j < each; j++) acc += mine[j];
  }
  uint32_t pre = addmod(first % ADLER_BASE, block_excl_scan_mod(acc % ADLER_BASE, sh, t));
  if (each >= 4) {
    for (uint32_t j = 0; j < each; j += 4) {
      const u32x4 v = *(const u32x4 *)(mine + j);
      u32x4 o;
      o.x = pre; pre = addmod(pre, v.x);
      o.y = pre; pre = addmod(pre, v.y);
      o.z = pre; pre = addmod(pre, v.z);
      o.w = pre; pre = addmod(pre, v.w);
      *(u32x4 *)(mine_out + j) = o;
    }
  } else {
    for (uint32_t j = 0; j < each; j++) {
      const uint32_t v = mine[j];
      mine_out[j] = pre;
      pre = addmod(pre, v);
    }
  }
}

__global__ __launch_bounds__(256) void adler_runs_a_kernel(const uint2 *__restrict__ sums, uint64_t n,
                                                           uint64_t n_chunks, uint64_t per, AdlerRuns R,
                                                           uint32_t *__restrict__ amb_raw, uint32_t amb_cap) {
  AmbRecord *amb = (AmbRecord *)amb_raw;
  const uint64_t run = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (run >= R.n_runs) return;
  const uint32_t r = (uint32_t)(n % ADLER_CHUNK);
  const uint64_t lo = run * per < n_chunks ? run * per : n_chunks;
  const uint64_t hi = lo + per < n_chunks ? lo + per : n_chunks;
  uint64_t s1 = R.s1_before[run], a_acc = 0;
  uint32_t last_hi = 0xFFFFFFFFu;  // branch of the previous chunk (unknown for the first of the run)
  for (uint64_t k = lo; k < hi; k++) {
    const uint2 sm = sums[k];
    const uint32_t len = k == 0 ? r : ADLER_CHUNK;
    const uint64_t C = (uint64_t)len * s1 + sm.y;  // < 2^32
    const bool hi_k = C >= 0x80000000ull;
    const bool ambiguous = C < ADLER_BASE || (C > 0x80000000ull - ADLER_BASE && C < 0x80000000ull + ADLER_BASE);
    if (ambiguous) {
      const uint32_t slot = atomicAdd(R.amb_count, 1u);
      if (slot < amb_cap) {
        AmbRecord rec;
        rec.k = (uint32_t)k;
        rec.s1 = (uint32_t)s1;
        rec.a_partial = (uint32_t)a_acc;
        rec.prev_hi = last_hi;
        amb[slot] = rec;
      }
    }
    a_acc = (a_acc + C % ADLER_BASE + (hi_k ? ADLER_BASE - 225u : 0u)) % ADLER_BASE;
    s1 = (s1 + sm.x) % ADLER_BASE;
    last_hi = hi_k ? 1u : 0u;
  }
  R.sum[run] = (uint32_t)a_acc;       // the run's a sum (the S1 sums are no longer needed)
  R.last_hi[run] = last_hi;
  R.s1_after[run] = (uint32_t)s1;
}

constexpr uint32_t REPLAY_MAX = 4096;  // ambiguous chunks replayed out of LDS; more: plain walk
__global__ __launch_bounds__(CHAIN_THREADS_A) void adler_replay_kernel(const uint2 *__restrict__ sums, uint64_t n,
                                                                       uint64_t n_chunks, uint64_t per, AdlerRuns R,
                                                                       uint32_t *__restrict__ amb_raw,
                                                                       uint32_t amb_cap, uint32_t *__restrict__ out) {
  // everything the serial replay reads is gathered into LDS by the whole
  // workgroup first: one thread walking global memory pays a full memory latency
  // per access
  // -- and everything that does not depend on the chunks before a record is worked out by the thread that
  // ranks it: the one thread's loop is a handful of 32-bit operations per record.
  __shared__ uint32_t keys[REPLAY_MAX];
  __shared__ uint32_t r_k[REPLAY_MAX];     // sorted by chunk index: the chunk,
  __shared__ uint32_t r_res[REPLAY_MAX];   // the predicted residue of s2 before it,
  __shared__ uint32_t r_C[REPLAY_MAX];     // C = n * s1 + S2 (< 2^32),
  __shared__ uint32_t r_pc[REPLAY_MAX];    // what the prediction adds for the chunk: (C - 225 * hi) mod p,
  __shared__ uint8_t r_prev[REPLAY_MAX];   // the branch of the chunk before it as C alone decides it
  const AmbRecord *amb = (const AmbRecord *)amb_raw;
  const int t = threadIdx.x;
  const uint32_t r = (uint32_t)(n % ADLER_CHUNK);
  const uint32_t *run_res = R.res_before, *run_a = R.sum, *run_last_hi = R.last_hi;
  const uint32_t n_amb = R.amb_count[0];
  if (n_amb > amb_cap || n_amb > REPLAY_MAX) {
    // adversarial input: more ambiguous chunks than worth sorting -- plain walk
    if (t != 0) return;
    uint32_t a1, a2;
    adler_unpack(1u, a1, a2);
    for (uint64_t k = 0; k < n_chunks; k++) adler_chunk_step(a1, a2, k == 0 ? r : ADLER_CHUNK, sums[k].x, sums[k].y);
    out[0] = adler_pack(a1, a2);
    return;
  }
  // branch of the chunk before position k when k opens a run: last chunk of the
  // nearest earlier non-empty run
  auto prev_branch = [&](uint64_t k, uint32_t rec_prev_hi) -> bool {
    if (rec_prev_hi != 0xFFFFFFFFu) return rec_prev_hi != 0;
    if (k == 0) return false;
    int64_t run = (int64_t)((k - 1) / per);
    while (run >= 0 && run_last_hi[run] == 0xFFFFFFFFu) run--;
    return run >= 0 && run_last_hi[run] != 0;
  };
  for (uint32_t i = (uint32_t)t; i < n_amb; i += CHAIN_THREADS_A) keys[i] = amb[i].k;
  __syncthreads();
  for (uint32_t i = (uint32_t)t; i < n_amb; i += CHAIN_THREADS_A) {  // rank sort (chunk indices are distinct)
    const AmbRecord rec = amb[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n_amb; j++) rank += keys[j] < rec.k ? 1u : 0u;
    const uint2 sm = sums[rec.k];
    const uint32_t len = rec.k == 0 ? r : ADLER_CHUNK;
    const uint32_t C = len * rec.s1 + sm.y;  // < 2^32
    r_k[rank] = rec.k;
    r_res[rank] = addmod(run_res[rec.k / per], rec.a_partial);
    r_C[rank] = C;
    r_pc[rank] = addmod(C % ADLER_BASE, C >= 0x80000000u ? ADLER_BASE - 225u : 0u);
    r_prev[rank] = prev_branch(rec.k, rec.prev_hi) ? 1 : 0;
  }
  __syncthreads();
  if (t != 0) return;
  // ---- replay of the ambiguous chunks, in order, by one thread
  uint32_t delta = 0;            // correction (mod p) of every predicted residue from here on
  int32_t exact_next = 0;        // exact s2 after the last replayed chunk ...
  uint64_t exact_at = ~0ull;     // ... valid as the input of chunk `exact_at`
  for (uint32_t i = 0; i < n_amb; i++) {
    const uint32_t k = r_k[i];
    const uint32_t rr = addmod(r_res[i], delta);
    const int32_t x = k == exact_at ? exact_next : (r_prev[i] ? (rr == 0 ? 0 : (int32_t)rr - (int32_t)ADLER_BASE) : (int32_t)rr);
    const uint32_t t2 = r_C[i] + (uint32_t)x;                 // wraps like the reference's int32
    const int32_t outv = (int32_t)t2 % (int32_t)ADLER_BASE;  // the reference's signed rem
    const uint32_t ro = (uint32_t)(outv < 0 ? outv + (int32_t)ADLER_BASE : outv);
    const uint32_t predicted = addmod(rr, r_pc[i]);
    delta = addmod(addmod(delta, ro), ADLER_BASE - predicted);
    exact_next = outv;
    exact_at = (uint64_t)k + 1;
  }
  // state after the last chunk
  uint32_t final_s2;
  if (exact_at == n_chunks) final_s2 = (uint32_t)(int32_t)exact_next;
  else {
    const uint64_t last_run = n_chunks ? (n_chunks - 1) / per : 0;
    const uint64_t total_res = n_chunks ? ((uint64_t)run_res[last_run] + run_a[last_run]) % ADLER_BASE : 0;
    const uint64_t rr = (total_res + delta) % ADLER_BASE;
    const bool ph = n_chunks ? prev_branch(n_chunks, 0xFFFFFFFFu) : false;
    final_s2 = (uint32_t)(int32_t)(ph ? (rr == 0 ? 0 : (int64_t)rr - ADLER_BASE) : (int64_t)rr);
  }
  const uint64_t lr = n_chunks ? (n_chunks - 1) / per : 0;
  const uint64_t s1_all = n_chunks ? R.s1_after[lr] : 1;
  out[0] = adler_pack((uint32_t)s1_all, final_s2);
}

}  // namespace zd
