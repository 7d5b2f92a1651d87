// deflate_pipeline.h -- what the kernels of the deflate pipeline share (deflate.hip): the layout of the
// context's scratch and the barrier that waits for LDS only.
#pragma once

#include "ctx.h"
#include "deflate_lane.h"
#include "wave_ops.h"

namespace zd {

constexpr uint32_t POS_PAD = 256;            // scratch slack per stream, in positions
constexpr uint32_t PARSE_PAD = 200;          // table entries behind the last position the parse may load (3 tiles of 64 + 3)
static_assert(PARSE_PAD <= POS_PAD, "inside the stream's scratch");
constexpr uint32_t MIN_BLOCK_SRC = 65277;    // a non-final block holds > 65534 - 258 source bytes

// streams with an out-of-range length are rejected by every kernel and take no scratch
__host__ __device__ inline uint64_t padded_positions(uint64_t src_len) {
  if (src_len > MAX_STREAM_LEN) src_len = 0;
  return ((src_len + 255) & ~255ull) + POS_PAD;
}
__host__ __device__ inline uint64_t max_blocks_of(uint64_t src_len) {
  if (src_len > MAX_STREAM_LEN) src_len = 0;
  return src_len / MIN_BLOCK_SRC + 2;
}

constexpr uint32_t MATCH_SNAP = 1u << 31;
// both answers of a position as the 64-bit word the parse works on (best of K | best of K/4 << 32) from the two tables
__device__ __forceinline__ uint64_t match_pair(const uint32_t *__restrict__ match, const uint32_t *__restrict__ snap, uint64_t i) {
  const uint32_t lo = match[i];
  const uint32_t hi = (lo & MATCH_SNAP) ? snap[i] : lo;
  return (uint64_t)(lo & ~MATCH_SNAP) | ((uint64_t)hi << 32);
}

struct DeflateScratch {
  uint64_t *pos_base;   // [n] first position slot of stream i
  uint64_t *blk_base;   // [n] first BlockDesc slot of stream i
  uint32_t *n_blocks;   // [n]
  uint32_t *error;      // [1] != 0: the batch does not fit what the caller declared (total_src_len too small,
                        //     or a stream longer than max_src_len: the grids are sized from it)
  uint16_t *prev;       // [P] chain links
  uint32_t *match;      // [P] lz_match_position's best of the first K candidates (dist << 9 | len, 0: none) | MATCH_SNAP when the best of the
                        //     first K/4 is another: that one is then in snap[] (round 5: 4 bytes a position where 8 were written and read)
  uint32_t *snap;       // [P] the best of the first K/4, written for positions with MATCH_SNAP only
  uint32_t *snap_used;  // [n] != 0: the stream has such positions (its parse then reads both tables side by side; zeroed by deflate_offsets_kernel)
  uint32_t *syms;       // [P]
  BlockDesc *blocks;    // [Bk]
  uint64_t cap_positions, cap_blocks;
};

// The threads of a workgroup that exchange data through LDS only wait for the LDS counter alone:
// __syncthreads() would also wait (vmcnt) for every global load and store the wave has in flight --
// source words requested ahead, the stores of results.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

}  // namespace zd
