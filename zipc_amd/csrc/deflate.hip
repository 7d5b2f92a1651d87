// deflate.hip -- batch deflate pipeline for gfx950 (CDNA4, wave64).
//
// The reference encoder (src/zipc_deflate.ml:742-1277) is one sequential loop per
// stream.  It is restated as a handful of kernels over a batch of independent streams
// (deflate_lane.h explains why each split preserves the output bit for bit):
//
//   deflate_offsets_kernel   per-stream bases into the scratch arrays (one scan)
//   lz_chain_kernel          hash + chain links: prev[p] = distance to the previous
//                            position with the same 15-bit hash4 (insert_hash,
//                            zd.ml:1145-1152).  One 1024-thread workgroup per
//                            stream, the 32 Ki-entry head table as u16 in LDS
//                            (64 KiB), 1024 positions inserted per round.
//   lz_match_window_kernel   best match of every position over the first K and the first
//                            K/4 chain candidates (find_backref, zd.ml:1176-1201) -> 8
//                            bytes per position.  A 1024-thread workgroup takes several
//                            consecutive 16 Ki-position tiles of one stream; a tile's 32 KiB
//                            window + tile of source and links sits in LDS (144 KiB), the
//                            next tile's is loaded into registers meanwhile.  A lane walks
//                            its 16 positions of a tile with 2 run slots on one cursor.
//   lz_match_kernel          the same for streams of up to 8 KiB, out of global memory,
//                            4 positions per lane (a whole-CU window would sit idle).
//   lz_parse_kernel          one wave per stream: the "macro step" of every position
//                            (deflate_lane.h), then which positions the lazy parse
//                            visits (Lz77.compress zd.ml:1203-1244), found per
//                            64-position tile by pointer doubling + marking
//                            instead of a serial walk; symbol emission and block
//                            cut (zd.ml:1118-1123).
//   deflate_emit_kernel      one wave per stream, blocks in order: histogram (LDS
//                            atomics), Huffman codes + stored/fixed/dynamic choice
//                            (write_block zd.ml:1094-1104) by the whole wave, then all 64
//                            lanes pack bits: wave prefix-scan of the per-symbol bit
//                            lengths, scatter-OR into an LDS staging row, coalesced
//                            flush of the completed bytes.
//   deflate_stored_kernel    level `None (write_all_non_compressed zd.ml:1106-1116).
#include <cstring>
#include <type_traits>

#include "deflate_pipeline.h"
#include "tuning.h"

namespace zd {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void scratch_caps(size_t n, size_t total_src_len, uint64_t &P, uint64_t &Bk) {
  P = total_src_len + (uint64_t)(POS_PAD + 256) * n + 256;
  Bk = total_src_len / MIN_BLOCK_SRC + 2 * (uint64_t)n + 16;
}

// A batch of more than DEFLATE_GROUP_BYTES of source (8 GiB) goes through the pipeline in groups of
// consecutive streams, each group of at most that much, one after the other through the same
// scratch: 14 bytes of scratch per source byte of a GROUP (2-byte links, 8-byte match entries,
// 4-byte symbols), 112 GiB at most however large the batch.  Smaller groups were measured on C4
// (8192 x 1 MiB): 4 GiB groups deflate 9 % slower and 2 GiB groups 23 % slower than one group --
// lz_parse and deflate_emit are one wave per stream, and 2048 streams leave 2 waves per SIMD where
// 8192 leave 8 -- so the scratch is bounded, not halved.  ZIPC_HIP_DEFLATE_GROUP_BYTES overrides
// the size (tests run with groups of a few streams).
static size_t deflate_group_bytes() { return tuning().deflate_group_bytes; }
// streams per group, and the source bytes a group can hold at most
static void deflate_grouping(size_t n, size_t max_src_len, size_t total_src_len, size_t &per_group, size_t &group_total) {
  per_group = n;
  group_total = total_src_len;
  if (total_src_len > deflate_group_bytes() && max_src_len > 0 && n > 1) {
    per_group = deflate_group_bytes() / max_src_len;
    per_group = per_group < 1 ? 1 : (per_group > n ? n : per_group);
    const unsigned __int128 bound = (unsigned __int128)per_group * max_src_len;
    group_total = bound < total_src_len ? (size_t)bound : total_src_len;
  }
}

size_t deflate_scratch_bytes(size_t n_all, size_t max_src_len, size_t total_all, int level) {
  size_t n, total_src_len;
  deflate_grouping(n_all, max_src_len, total_all, n, total_src_len);
  uint64_t P, Bk;
  scratch_caps(n, total_src_len, P, Bk);
  size_t b = 0;
  b += align_up(n * 8, 256) * 2 + align_up(n * 4, 256) * 2 + 256;
  if (level != LEVEL_NONE) {
    b += align_up(P * 2, 256) + 2 * align_up(P * 4, 256) + align_up(P * 4, 256);
    b += align_up(Bk * sizeof(BlockDesc), 256);
  }
  return b + 1024;
}

static DeflateScratch carve(void *base, size_t n, size_t total_src_len, int level) {
  DeflateScratch s;
  uint64_t P, Bk;
  scratch_caps(n, total_src_len, P, Bk);
  uint8_t *p = (uint8_t *)base;
  s.pos_base = (uint64_t *)p; p += align_up(n * 8, 256);
  s.blk_base = (uint64_t *)p; p += align_up(n * 8, 256);
  s.n_blocks = (uint32_t *)p; p += align_up(n * 4, 256);
  s.snap_used = (uint32_t *)p; p += align_up(n * 4, 256);
  s.error = (uint32_t *)p; p += 256;
  s.prev = nullptr; s.match = nullptr; s.snap = nullptr;
  s.syms = nullptr; s.blocks = nullptr;
  if (level != LEVEL_NONE) {
    s.prev = (uint16_t *)p; p += align_up(P * 2, 256);
    s.match = (uint32_t *)p; p += align_up(P * 4, 256);
    s.snap = (uint32_t *)p; p += align_up(P * 4, 256);
    s.syms = (uint32_t *)p; p += align_up(P * 4, 256);
    s.blocks = (BlockDesc *)p; p += align_up(Bk * sizeof(BlockDesc), 256);
  }
  s.cap_positions = P;
  s.cap_blocks = Bk;
  return s;
}

// ---------------------------------------------------------------------------------
// Exclusive scan of the per-stream scratch needs (single workgroup).
__global__ __launch_bounds__(1024) void deflate_offsets_kernel(const StreamDesc *__restrict__ descs,
                                                               uint32_t n, DeflateScratch S, uint64_t max_src_len) {
  __shared__ uint64_t part_p[1024], part_b[1024];
  __shared__ uint32_t too_long;
  const uint32_t t = threadIdx.x;
  if (t == 0) too_long = 0;
  __syncthreads();
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = t * per, hi = lo + per < n ? lo + per : n;
  uint64_t sp = 0, sb = 0;
  bool over = false;
  for (uint32_t i = lo; i < hi; i++) {
    const uint64_t l = descs[i].src_len;
    sp += padded_positions(l);
    sb += max_blocks_of(l);
    // a stream beyond the format's range is rejected on its own (above); one beyond what the
    // caller declared would be matched and checksummed only in part: the whole batch is refused
    over |= l <= MAX_STREAM_LEN && l > max_src_len;
  }
  part_p[t] = sp;
  part_b[t] = sb;
  if (over) too_long = 1;
  __syncthreads();
  if (t == 0) {
    uint64_t ap = 0, ab = 0;
    for (int i = 0; i < 1024; i++) {
      uint64_t vp = part_p[i], vb = part_b[i];
      part_p[i] = ap; part_b[i] = ab;
      ap += vp; ab += vb;
    }
    S.error[0] = (ap > S.cap_positions || ab > S.cap_blocks || too_long) ? 1u : 0u;
  }
  __syncthreads();
  sp = part_p[t];
  sb = part_b[t];
  for (uint32_t i = lo; i < hi; i++) {
    S.pos_base[i] = sp;
    S.blk_base[i] = sb;
    S.snap_used[i] = 0;
    sp += padded_positions(descs[i].src_len);
    sb += max_blocks_of(descs[i].src_len);
  }
}

// ---------------------------------------------------------------------------------
// Hash-chain links.  Round = 1024 consecutive positions, one per thread.  The
// reference inserts positions one by one (head/prev arrays, zd.ml:1150-1152); a
// round reproduces the same links for all its positions at once:
//   * every thread first reads head[h] (latest earlier position with its hash);
//   * members of the round that share a hash are ordered by a "peel": the not yet
//     ordered ones all store their position to head[h], exactly one store lands,
//     everybody with that hash reads back who it was -- after as many turns as
//     the largest group has members every thread knows its nearest smaller
//     member.  Threads with an equal hash at most 8 positions to their left take
//     that neighbour directly, and only threads with no equal hash within 8 to
//     their right take part in the peel, so runs and short periods cost one turn;
//   * the largest member of each group leaves its position in head[h].
// head holds positions mod 2^16; every 16384 positions entries older than 32768
// are replaced by a marker that decodes as "none" until the next sweep.
constexpr uint32_t CHAIN_THREADS = 1024;
constexpr int CHAIN_PPT = 1;                                // positions per thread and round
constexpr uint32_t CHAIN_ROUND = CHAIN_THREADS * CHAIN_PPT;  // positions inserted per round
constexpr uint32_t SWEEP_PERIOD = 16384;
constexpr uint32_t SWEEP_MARK = 20000;
constexpr int NEAR = 8;
constexpr int PLAIN_TURNS = 4;  // peel turns before the neighbour short-cut is worth its LDS reads
static_assert(SWEEP_PERIOD % CHAIN_ROUND == 0, "sweeps fall on round boundaries");

// A link is the distance to the nearest earlier position with the same hash if that is within 32768, else 0:
// a function of the 32 KiB before the position and nothing else.  So a long stream's links can be made by
// several workgroups (SEG): each takes seg_positions of them, starts with an empty table at the sweep boundary at
// least 32768 before its first one, and stores the links of its own positions only.
// (seg_positions: a multiple of the sweep period; 32 or 64 Ki while that leaves the chip workgroups to spare -- twice
// or half again the work for four times or twice the workgroups -- else 128 Ki)
constexpr uint32_t CHAIN_SEG_MIN = 2 * SWEEP_PERIOD, CHAIN_SEG_MAX = 8 * SWEEP_PERIOD;
template <bool SEG>
__device__ __forceinline__ void lz_chain_workgroup(const uint8_t *__restrict__ src_arena,
                                                   const StreamDesc *__restrict__ descs,
                                                   DeflateScratch S, uint32_t stream, uint32_t seg,
                                                   uint32_t seg_positions) {
  __shared__ uint16_t head[32768];
  __shared__ uint16_t hs[CHAIN_ROUND + 2 * NEAR];
  __shared__ uint32_t peel_more[2];
  __shared__ uint32_t head_spare;
  if (S.error[0]) return;
  const uint32_t t = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) return;
  const uint32_t len = (uint32_t)sd.src_len;
  const uint8_t *s = src_arena + sd.src_off;
  uint16_t *prev = S.prev + S.pos_base[stream];
  const uint32_t stream_max_pos = len - 4;
  // SEG: links of [own_lo, own_hi], rounds from B_first on
  const uint32_t own_lo = SEG ? seg * seg_positions : 0u;
  if (SEG && own_lo > stream_max_pos) return;
  const uint32_t own_hi = SEG ? (stream_max_pos - own_lo >= seg_positions ? own_lo + seg_positions - 1u : stream_max_pos) : stream_max_pos;
  const uint32_t B_first = SEG ? (own_lo > 2u * SWEEP_PERIOD ? own_lo - 2u * SWEEP_PERIOD : 0u) : 0u;  // (own_lo is a multiple of the period)
  const uint32_t max_pos = own_hi;  // the last position inserted
  if (t < NEAR) { hs[t] = 0xFFFF; hs[CHAIN_ROUND + NEAR + t] = 0xFFFF; }
  if (t < 2) peel_more[t] = 0;

  // the 4 source bytes of a round's positions are requested one round ahead (the
  // barriers below do not wait for them)
  uint32_t word_next[CHAIN_PPT];
#pragma unroll
  for (int i = 0; i < CHAIN_PPT; i++) {
    const uint32_t p = B_first + t + CHAIN_THREADS * (uint32_t)i;
    word_next[i] = load_u32_le(s + (p <= max_pos ? p : max_pos));  // unconditional: the wait counts stay exact
  }
  for (uint32_t B = B_first; B <= max_pos; B += CHAIN_ROUND) {
    uint32_t word[CHAIN_PPT];
#pragma unroll
    for (int i = 0; i < CHAIN_PPT; i++) {
      word[i] = word_next[i];
      const uint64_t pn = (uint64_t)B + CHAIN_ROUND + t + CHAIN_THREADS * (uint32_t)i;
      word_next[i] = load_u32_le(s + (pn <= max_pos ? pn : (uint64_t)max_pos));
    }
    if ((B % SWEEP_PERIOD) == 0) {
      const uint16_t mark = (uint16_t)(B + SWEEP_MARK);
      for (uint32_t i = t; i < 32768; i += CHAIN_THREADS) {
        bool keep = false;
        if (B != B_first) {
          const uint32_t d = (B - head[i]) & 0xFFFFu;
          keep = d >= 1 && d <= 32768;
        }
        if (!keep) head[i] = mark;
      }
      lds_barrier();
    }
    // local index of my i-th position: t + 1024 * i (coalesced loads and stores)
    uint32_t h[CHAIN_PPT], e_old[CHAIN_PPT];
    bool active[CHAIN_PPT];
#pragma unroll
    for (int i = 0; i < CHAIN_PPT; i++) {
      const uint32_t li = t + CHAIN_THREADS * (uint32_t)i;
      const uint32_t p = B + li;
      active[i] = p <= max_pos;
      h[i] = active[i] ? hash4(word[i]) : 0xFFFFu;
      hs[NEAR + li] = (uint16_t)h[i];
      e_old[i] = active[i] ? head[h[i]] : 0;
    }
    lds_barrier();

    // The peel starts with every position both reading and writing (exact on its
    // own).  Only if it is still running after PLAIN_TURNS turns -- runs, short
    // periods -- are the neighbour hashes consulted: a position with an equal hash
    // within 8 to its left takes that one as predecessor and stops reading, and
    // one with an equal hash within 8 to its right stops writing (it is never the
    // predecessor of a position that still reads, nor the group's last).
    // Branch-light on purpose: the workgroup's 16 waves share the CU's one scalar issue
    // per clock, so a turn is select-and-arithmetic on per-lane integers, stores of
    // lanes that have nothing to store go to a spare LDS slot, and only the neighbour
    // pass and the vote branch.
    uint32_t near_pred[CHAIN_PPT];
    uint32_t reader[CHAIN_PPT], writer[CHAIN_PPT], pending[CHAIN_PPT];  // 0 / 1
    int notmax[CHAIN_PPT];      // > 0: a later member of my group landed
    int pred_local[CHAIN_PPT];  // nearest earlier member seen so far, -1: none
    uint32_t h2[CHAIN_PPT];     // byte offset of my head entry (entry 0 for positions past the end)
#pragma unroll
    for (int i = 0; i < CHAIN_PPT; i++) {
      near_pred[i] = 0;
      reader[i] = writer[i] = pending[i] = active[i] ? 1u : 0u;
      notmax[i] = 0;
      pred_local[i] = -1;
      h2[i] = active[i] ? h[i] * 2u : 0u;
    }
    uint8_t *const head_bytes = (uint8_t *)head;
    const uint32_t spare = (uint32_t)((uint8_t *)&head_spare - head_bytes);  // where idle lanes store
    for (int turn = 0;; turn++) {
      if (turn == PLAIN_TURNS) {
#pragma unroll
        for (int i = 0; i < CHAIN_PPT; i++) {
          const uint32_t li = t + CHAIN_THREADS * (uint32_t)i;
          bool has_succ = false;
          if (active[i]) {
#pragma unroll
            for (int k = NEAR; k >= 1; k--)
              if (hs[NEAR + li - k] == h[i]) near_pred[i] = (uint32_t)k;  // ends with the nearest
#pragma unroll
            for (int k = 1; k <= NEAR; k++) has_succ |= hs[NEAR + li + k] == h[i];
          }
          reader[i] = (active[i] && near_pred[i] == 0) ? 1u : 0u;
          if (has_succ) { writer[i] = 0; pending[i] = 0; }
        }
      }
#pragma unroll
      for (int i = 0; i < CHAIN_PPT; i++)
        *(uint16_t *)(head_bytes + (pending[i] ? h2[i] : spare)) = (uint16_t)(B + t + CHAIN_THREADS * (uint32_t)i);
      lds_barrier();
      uint32_t any_pending = 0;
#pragma unroll
      for (int i = 0; i < CHAIN_PPT; i++) {
        // who landed in my bucket, relative to me: 0 me, < 0 an earlier member, > 0 a later one.
        // (Lanes that neither read nor write any more -- neighbour pass -- and positions past
        // the end compute values nobody uses.)
        const uint32_t li = t + CHAIN_THREADS * (uint32_t)i;
        const uint32_t r_local = ((uint32_t)*(const uint16_t *)(head_bytes + h2[i]) - B) & 0xFFFFu;
        const int rel = (int)r_local - (int)li;
        pending[i] = rel != 0 ? pending[i] : 0u;
        const int cand = rel < 0 ? (int)r_local : -1;
        pred_local[i] = cand > pred_local[i] ? cand : pred_local[i];
        notmax[i] = rel > notmax[i] ? rel : notmax[i];
        any_pending |= pending[i];
      }
      // does anybody still peel?  One flag per turn parity: set by the waves that do,
      // read behind the barrier that also keeps this turn's read-backs ahead of the
      // next turn's stores, cleared for the turn after next.
      if (__builtin_amdgcn_ballot_w64(any_pending != 0) && (t & 63u) == 0) peel_more[turn & 1] = 1;
      lds_barrier();
      const uint32_t more = peel_more[turn & 1];
      if (t == 0) peel_more[(turn + 1) & 1] = 0;
      if (!more) break;
    }
#pragma unroll
    for (int i = 0; i < CHAIN_PPT; i++) {
      const uint32_t li = t + CHAIN_THREADS * (uint32_t)i;
      const uint32_t p = B + li;
      // the largest member of each group leaves its position in head[h]
      *(uint16_t *)(head_bytes + ((writer[i] && notmax[i] <= 0) ? h2[i] : spare)) = (uint16_t)p;
      if (active[i]) {
        uint32_t d;
        if (near_pred[i]) d = near_pred[i];
        else if (pred_local[i] >= 0) d = li - (uint32_t)pred_local[i];
        else {
          d = (p - e_old[i]) & 0xFFFFu;
          if (d > 32768) d = 0;
        }
        if (!SEG || p >= own_lo) prev[p] = (uint16_t)d;
      }
    }
    lds_barrier();
  }
}

__global__ __launch_bounds__(CHAIN_THREADS) void lz_chain_kernel(const uint8_t *__restrict__ src_arena,
                                                                 const StreamDesc *__restrict__ descs,
                                                                 DeflateScratch S) {
  lz_chain_workgroup<false>(src_arena, descs, S, blockIdx.x, 0, 0);
}

__global__ __launch_bounds__(CHAIN_THREADS) void lz_chain_segments_kernel(const uint8_t *__restrict__ src_arena,
                                                                          const StreamDesc *__restrict__ descs,
                                                                          DeflateScratch S, uint32_t segs_per_stream,
                                                                          uint32_t seg_positions) {
  lz_chain_workgroup<true>(src_arena, descs, S, blockIdx.x / segs_per_stream, blockIdx.x % segs_per_stream, seg_positions);
}

// ---------------------------------------------------------------------------------
// Hash-chain links by ORDERED LDS EXCHANGE (round 4).  insert_hash (zd.ml:1150-1152) is "prev[pos] <- head[hash];
// head[hash] <- pos", position after position.  One ds_wrxchg_rtn_b32 does exactly that for 64 positions at once
// IF the lanes that hit one address are served in ascending lane order -- which gfx950's LDS does (measured:
// tools/probes/lds_xchg_order.hip, any mix of addresses, banks and EXEC masks; every context repeats the probe when it
// is created, xchg_order_probe below, and keeps the peel kernel above where it fails) -- and a wave's LDS
// operations execute in the order they were issued.  So ONE wave inserts a stream: lane = position, 64 consecutive
// positions per exchange, the table 32 Ki words of 32 bits (positions as they are: no wrap, no sweeps; "none" is a
// value 64 Ki below zero, so that its distance from any position is beyond the window).  No barrier, no peel, ~10
// instructions per 64 positions where the peel kernel issues 67 + 60 -- and what bounds it is one wave's issue rate:
// the table takes 128 of the CU's 160 KiB, so a CU works on one stream at a time.
constexpr uint32_t XCHG_NONE = 0xFFFF0000u;  // p - XCHG_NONE = p + 65536 (p <= MAX_STREAM_LEN: no wrap) > MAX_MATCH_DIST
static_assert(MAX_STREAM_LEN <= XCHG_NONE, "no position looks like the table's empty entry or wraps past it");
// ONE wave's exchanges are ordered; to give a stream more than one wave, XCHG_WAVES waves take TURNS: turn t -- XCHG_U
// rounds of 64 positions -- belongs to wave t mod XCHG_WAVES, which requests its source words and hashes them while the
// others have their turns, waits until the word `turn_now` in LDS says t, issues its exchanges and hands over
// (turn_now <- t + 1, behind its exchanges: a wave's LDS operations are executed in the order they were issued, and
// the next wave issues its exchanges only after it has READ the new value), then works out its links and stores them.
// Measured (C2, ms per GiB; tools/ab_wall.sh on one box): 4 waves x 16 rounds 1.42-1.45; 8 waves 1.41; 2 waves 1.97;
// turns of 8 / 32 rounds 1.83 / 1.41; polling without s_sleep 1.63; the hand-over NOT waiting for the exchanges' answers
// 1.42 -- neither the hand-over nor a wave's own work is what a stream waits for: 1024 exchanges of 64 random addresses
// take the LDS ~45 clocks each (text, where lanes of one exchange share addresses: 1.59).  One wave alone: 3.4.
#ifndef ZD_XCHG_WAVES
#define ZD_XCHG_WAVES 4
#endif
#ifndef ZD_XCHG_U
#define ZD_XCHG_U 16
#endif
#ifndef ZD_XCHG_SLEEP
#define ZD_XCHG_SLEEP 1
#endif
constexpr uint32_t XCHG_WAVES = ZD_XCHG_WAVES;
constexpr int XCHG_U = ZD_XCHG_U;             // rounds of 64 positions per turn
constexpr uint32_t XCHG_TURN = 64u * XCHG_U;  // positions per turn
static_assert(MAX_MATCH_DIST % XCHG_TURN == 0 && SWEEP_PERIOD % XCHG_TURN == 0, "segments start on turn boundaries");

template <bool SEG>
__device__ __forceinline__ void lz_chain_xchg_workgroup(const uint8_t *__restrict__ src_arena,
                                                        const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                        uint32_t stream, uint32_t seg, uint32_t seg_positions) {
  __shared__ __attribute__((aligned(16))) uint32_t head[32768];
  __shared__ uint32_t turn_now;
  if (S.error[0]) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const StreamDesc sd = descs[stream];
  if (sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) return;
  const uint32_t len = (uint32_t)sd.src_len;
  const uint8_t *s = src_arena + sd.src_off;
  uint16_t *prev = S.prev + S.pos_base[stream];
  const uint32_t stream_max_pos = len - 4;
  // SEG: links of [own_lo, own_hi]; the table is warmed up with the 32 Ki positions before (a link is the distance to
  // the nearest earlier position with the same hash if within 32768, else 0: a function of those and nothing else)
  const uint32_t own_lo = SEG ? seg * seg_positions : 0u;
  if (SEG && own_lo > stream_max_pos) return;
  const uint32_t max_pos = SEG ? (stream_max_pos - own_lo >= seg_positions ? own_lo + seg_positions - 1u : stream_max_pos) : stream_max_pos;
  const uint32_t first = SEG ? (own_lo > (uint32_t)MAX_MATCH_DIST ? own_lo - (uint32_t)MAX_MATCH_DIST : 0u) : 0u;
  const uint32_t t_first = first / XCHG_TURN, t_last = max_pos / XCHG_TURN;  // turns [t_first, t_last]
  {
    const u32x4 none = {XCHG_NONE, XCHG_NONE, XCHG_NONE, XCHG_NONE};
    for (uint32_t i = tid * 4u; i < 32768u; i += 4u * 64u * XCHG_WAVES) *(u32x4 *)(head + i) = none;
    if (tid == 0) turn_now = t_first;
  }
  __syncthreads();
  auto load = [&](uint32_t t, uint32_t *wd) {  // (clamped, never predicated: the wait counts stay exact)
    const uint32_t tc = t <= t_last ? t : t_last;
#pragma unroll
    for (int u = 0; u < XCHG_U; u++) {
      const uint32_t p = tc * XCHG_TURN + 64u * (uint32_t)u + lane;  // (64 KiB of headroom below 2^32: no wrap)
      wd[u] = load_u32_le(s + (p <= max_pos ? p : max_pos));
    }
  };
  // my turn t: wait for it, all its exchanges one behind the other, hand over, then the links
  auto insert = [&](uint32_t t, const uint32_t *wd, auto WHOLE) {
    uint32_t old[XCHG_U], hb[XCHG_U];
#pragma unroll
    for (int u = 0; u < XCHG_U; u++) hb[u] = hash4(wd[u]);
    while (__hip_atomic_load(&turn_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != t) {
#if ZD_XCHG_SLEEP
      __builtin_amdgcn_s_sleep(ZD_XCHG_SLEEP);
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int u = 0; u < XCHG_U; u++) {
      const uint32_t p = t * XCHG_TURN + 64u * (uint32_t)u + lane;
      old[u] = XCHG_NONE;
      if (decltype(WHOLE)::value || p <= max_pos)
        old[u] = __hip_atomic_exchange(&head[hb[u]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (release: the exchanges above are complete before the next wave sees its turn)
    if (lane == 0) __hip_atomic_store(&turn_now, t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int u = 0; u < XCHG_U; u++) {
      const uint32_t p = t * XCHG_TURN + 64u * (uint32_t)u + lane;
      const uint32_t d = p - old[u];
      if ((decltype(WHOLE)::value || p <= max_pos) && (!SEG || p >= own_lo)) prev[p] = (uint16_t)(d <= (uint32_t)MAX_MATCH_DIST ? d : 0u);
    }
  };
  uint32_t wa[XCHG_U], wb[XCHG_U];
  uint32_t t = t_first + w;
  if (t > t_last) return;
  load(t, wa);
  for (;;) {  // (two sets of registers that swap roles by name: nothing waits for the words just requested)
    load(t + XCHG_WAVES, wb);
    if (t == t_last) { insert(t, wa, std::false_type{}); break; }
    insert(t, wa, std::true_type{});
    t += XCHG_WAVES;
    if (t > t_last) break;
    load(t + XCHG_WAVES, wa);
    if (t == t_last) { insert(t, wb, std::false_type{}); break; }
    insert(t, wb, std::true_type{});
    t += XCHG_WAVES;
    if (t > t_last) break;
  }
}

__global__ __launch_bounds__(64 * XCHG_WAVES) void lz_chain_xchg_kernel(const uint8_t *__restrict__ src_arena,
                                                                        const StreamDesc *__restrict__ descs, DeflateScratch S) {
  lz_chain_xchg_workgroup<false>(src_arena, descs, S, blockIdx.x, 0, 0);
}
__global__ __launch_bounds__(64 * XCHG_WAVES) void lz_chain_xchg_segments_kernel(const uint8_t *__restrict__ src_arena,
                                                                                 const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                                                 uint32_t segs_per_stream, uint32_t seg_positions) {
  lz_chain_xchg_workgroup<true>(src_arena, descs, S, blockIdx.x / segs_per_stream, blockIdx.x % segs_per_stream, seg_positions);
}

// The probe every context runs once (api.hip zipc_hip_create): 256 exchanges of 64 lanes on a few addresses, under EXEC
// masks, against the order the reference inserts in.  out[0] = mismatches.
__global__ __launch_bounds__(64) void xchg_order_probe_kernel(uint32_t *__restrict__ out) {
  __shared__ uint32_t head[64];
  const uint32_t lane = threadIdx.x;
  head[lane] = XCHG_NONE;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t bad = 0, x = 0x9E3779B1u * (lane + 1u);
  for (uint32_t r = 0; r < 256u; r++) {
    x = x * 1664525u + 1013904223u;
    const uint32_t kind = r & 3u;
    const uint32_t a = kind == 0 ? 5u : kind == 1 ? (lane & 1u) : kind == 2 ? ((x >> 13) & 7u) : ((x >> 9) & 63u);
    const bool active = (r % 3u == 0) || ((x >> 20) & 1u);
    // expected: the nearest lower ACTIVE lane with my address in this round, else the address's value before the round
    uint32_t before = head[a];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t want = before;
    for (uint32_t l = 0; l < 64u; l++) {
      const uint32_t al = (uint32_t)__builtin_amdgcn_readlane((int)a, (int)l);
      const bool actl = (__builtin_amdgcn_ballot_w64(active) >> l) & 1ull;
      if (actl && al == a && l < lane) want = r * 64u + l;
    }
    uint32_t got = want;
    if (active) got = __hip_atomic_exchange(&head[a], r * 64u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bad += got != want ? 1u : 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  bad = wave_sum(bad);
  if (lane == 0) out[0] = bad;
}

// ---- the guard in the kernel's own shape (round 6) ---------------------------------------------------------------
// The toy above is one wave on 64 words.  lz_chain_xchg_kernel is four waves taking turns on a 128 KiB table, 16 exchanges
// a turn, with loads in flight beside them: a chip (or a firmware) that served same-address lanes out of order only
// under that load would pass the toy and write valid streams with the wrong bytes.  So the context also runs THE KERNEL
// on a stream made for it and compares its links with lz_chain_kernel's, which orders equal hashes itself:
//   runs of one byte (every lane of every round on one address, 16 rounds a turn, turn after turn), periods of 2, 3, 5
//   and 7 bytes (a few addresses a round, each hit by many lanes), symbols of 1 and 2 bits (a few dozen addresses,
//   chains thousands deep), nibbles, bytes, and the same runs again more than a window later (links beyond 32768).
// And every context checks the first streams of its first real batch the same way, under the load of that batch
// (chain_check below).
static std::vector<uint8_t> xchg_probe_stream() {
  std::vector<uint8_t> v;
  uint32_t x = 0x2545F491u;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
  auto run = [&](size_t n, uint8_t b) { v.insert(v.end(), n, b); };
  auto period = [&](size_t n, int k) { for (size_t i = 0; i < n; i++) v.push_back((uint8_t)('a' + i % k)); };
  auto symbols = [&](size_t n, int bits) { for (size_t i = 0; i < n; i++) v.push_back((uint8_t)(rnd() >> 9 & ((1u << bits) - 1))); };
  run(20000, 0); period(9000, 2); period(9000, 3); period(6000, 5); period(6000, 7);
  symbols(24000, 1); symbols(24000, 2); symbols(20000, 4); symbols(12000, 8);
  run(3000, 0xFF); run(5000, 0); symbols(9000, 1);            // (these zeros: 130 000 positions behind the first ones)
  run(70000, 7); period(5000, 2); symbols(8000, 3); run(1031, 0);  // a run longer than two windows, an odd end
  return v;
}

__global__ __launch_bounds__(256) void chain_links_save_kernel(const StreamDesc *__restrict__ descs, DeflateScratch S, uint16_t *__restrict__ saved) {
  const StreamDesc sd = descs[blockIdx.x];
  if (S.error[0] || sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) return;
  const uint64_t base = S.pos_base[blockIdx.x], base0 = S.pos_base[0];
  for (uint64_t p = threadIdx.x; p < sd.src_len - 3; p += 256) saved[base - base0 + p] = S.prev[base + p];
}
// out[0] += positions whose saved link differs from the one in the scratch now, out[1] += positions compared; a difference
// also sets the batch's error word to 2: every stream of the call then reports ZIPC_HIP_ERR_HIP instead of bytes nobody checked.
__global__ __launch_bounds__(256) void chain_links_compare_kernel(const StreamDesc *__restrict__ descs, DeflateScratch S, const uint16_t *__restrict__ saved,
                                                                  unsigned long long *__restrict__ out) {
  const StreamDesc sd = descs[blockIdx.x];
  if (S.error[0] == 1u || sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) return;
  const uint64_t base = S.pos_base[blockIdx.x], base0 = S.pos_base[0];
  uint32_t bad = 0;
  for (uint64_t p = threadIdx.x; p < sd.src_len - 3; p += 256) bad += saved[base - base0 + p] != S.prev[base + p] ? 1u : 0u;
  bad = wave_sum(bad);
  if ((threadIdx.x & 63u) == 0 && bad) {
    __hip_atomic_fetch_add(&out[0], (unsigned long long)bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    S.error[0] = 2u;
  }
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&out[1], (unsigned long long)(sd.src_len - 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Streams [0, k) of a batch whose links lz_chain_xchg(_segments)_kernel has just written (S.prev): keep a copy, let
// lz_chain_kernel write them again -- exact links either way from here on -- and count the differences into `counts`
// (device-visible host memory: nothing waits for it).  Enqueued on ctx->cur.
static hipError_t chain_check_enqueue(zipc_hip_ctx *ctx, const uint8_t *d_src, const StreamDesc *dd, DeflateScratch Q, size_t k, size_t max_src_len,
                                      unsigned long long *counts) {
  const size_t positions = k * (size_t)padded_positions(max_src_len);
  const hipError_t e = ctx->ensure(ctx->chain_check_links, positions * 2);
  if (e != hipSuccess) return e;
  uint16_t *saved = (uint16_t *)ctx->chain_check_links.p;
  ZD_LAUNCH(ctx, "chain_check", chain_links_save_kernel, dim3((unsigned)k), dim3(256), 0, dd, Q, saved);
  ZD_LAUNCH(ctx, "chain_check", lz_chain_kernel, dim3((unsigned)k), dim3(CHAIN_THREADS), 0, d_src, dd, Q);
  ZD_LAUNCH(ctx, "chain_check", chain_links_compare_kernel, dim3((unsigned)k), dim3(256), 0, dd, Q, (const uint16_t *)saved, counts);
  return hipGetLastError();
}

bool xchg_order_probe(zipc_hip_ctx *ctx) {
  if (ctx->ensure(ctx->io_small, 256) != hipSuccess) return false;
  uint32_t *d = (uint32_t *)ctx->io_small.p;
  uint32_t h = 1;
  if (hipMemcpyAsync(d, &h, 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return false;
  hipLaunchKernelGGL(xchg_order_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
  if (hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return false;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
  if (h != 0) return false;
  // ... and the kernel itself, whole and by segments, against the kernel that orders equal hashes itself
  if (!ctx->chain_check_host) {
    if (hipHostMalloc((void **)&ctx->chain_check_host, 4 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { ctx->chain_check_host = nullptr; return false; }
    memset(ctx->chain_check_host, 0, 4 * sizeof(unsigned long long));
  }
  const std::vector<uint8_t> v = xchg_probe_stream();
  const size_t len = v.size();
  StreamDesc sd;
  memset(&sd, 0, sizeof sd);
  sd.src_len = len; sd.dst_cap = len;
  zipc_hip_ctx::Buf src, desc;
  bool ok = false;
  do {
    if (ctx->ensure(src, len + 64) != hipSuccess || ctx->ensure(desc, sizeof sd) != hipSuccess) break;
    if (ctx->ensure(ctx->deflate_scratch, deflate_scratch_bytes(1, len, len, LEVEL_DEFAULT)) != hipSuccess) break;
    if (hipMemcpyAsync(src.p, v.data(), len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) break;
    if (hipMemcpyAsync(desc.p, &sd, sizeof sd, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) break;
    DeflateScratch S = carve(ctx->deflate_scratch.p, 1, len, LEVEL_DEFAULT);
    unsigned long long *counts = ctx->chain_check_host + 2;  // [2], [3]: the probe's; [0], [1]: the first batch's
    counts[0] = counts[1] = 0;
    bool launched = true;
    for (int form = 0; form < 2 && launched; form++) {
      hipLaunchKernelGGL(deflate_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const StreamDesc *)desc.p, 1u, S, (uint64_t)len);
      if (form == 0)
        hipLaunchKernelGGL(lz_chain_xchg_kernel, dim3(1), dim3(64 * XCHG_WAVES), 0, ctx->stream, (const uint8_t *)src.p, (const StreamDesc *)desc.p, S);
      else {
        const uint32_t xseg = 96u << 10, xsegs = (uint32_t)((len + xseg - 1) / xseg);
        hipLaunchKernelGGL(lz_chain_xchg_segments_kernel, dim3(xsegs), dim3(64 * XCHG_WAVES), 0, ctx->stream, (const uint8_t *)src.p,
                           (const StreamDesc *)desc.p, S, xsegs, xseg);
      }
      launched = chain_check_enqueue(ctx, (const uint8_t *)src.p, (const StreamDesc *)desc.p, S, 1, len, counts) == hipSuccess;
    }
    if (!launched || hipStreamSynchronize(ctx->stream) != hipSuccess) break;
    ok = counts[0] == 0 && counts[1] == 2 * (unsigned long long)(len - 3);
  } while (0);
  if (src.p) (void)hipFree(src.p);
  if (desc.p) (void)hipFree(desc.p);
  return ok;
}

// ---------------------------------------------------------------------------------
constexpr uint32_t MATCH_THREADS = 256;
constexpr int MATCH_NP = 4;  // positions per lane, their chain walks interleaved
constexpr uint32_t MATCH_TILE = MATCH_THREADS * MATCH_NP;

// A position's two answers into the tables; true when the second one went to snap[]
__device__ __forceinline__ bool match_store(uint32_t *__restrict__ match, uint32_t *__restrict__ snap, uint32_t p, uint32_t best, uint32_t first) {
  const bool two = first != best;
  match[p] = best | (two ? MATCH_SNAP : 0u);
  if (two) snap[p] = first;
  return two;
}

__global__ __launch_bounds__(MATCH_THREADS) void lz_match_kernel(const uint8_t *__restrict__ src_arena,
                                                                 const StreamDesc *__restrict__ descs,
                                                                 DeflateScratch S, uint32_t n_streams,
                                                                 uint32_t chunks_per_stream, int K, int Kq) {
  if (S.error[0]) return;
  // XCD-aware order: workgroups b and b + 8 share an XCD (and its L2), so the
  // grid is cut into 8 contiguous slabs, one per XCD: all tiles of a stream --
  // which re-read the same 32 KiB window and its chain links -- hit one L2.
  const uint32_t nb = gridDim.x;
  const uint32_t per_xcd = (nb + 7) / 8;
  const uint32_t logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const uint32_t stream = logical / chunks_per_stream;
  const uint32_t chunk = logical % chunks_per_stream;
  if (stream >= n_streams) return;  // grid is padded to a multiple of 8
  const StreamDesc sd = descs[stream];
  if (sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) return;
  const uint32_t len = (uint32_t)sd.src_len;
  const uint32_t p0 = chunk * MATCH_TILE + threadIdx.x;
  if (chunk * MATCH_TILE > len - 4) return;
  const uint64_t base = S.pos_base[stream];
  uint32_t p[MATCH_NP];
  bool act[MATCH_NP];
  uint64_t out[MATCH_NP];
#pragma unroll
  for (int i = 0; i < MATCH_NP; i++) {
    p[i] = p0 + MATCH_THREADS * (uint32_t)i;
    act[i] = p[i] <= len - 4;
    if (!act[i]) p[i] = 0;
  }
  if ((uint64_t)chunk * MATCH_TILE + MATCH_TILE > (uint64_t)len - 4 && threadIdx.x < PARSE_PAD)
    S.match[base + (len - 3) + threadIdx.x] = 0;  // what the parse may read behind the last position
  lz_match_positions<MATCH_NP>(src_arena + sd.src_off, len, p, act, S.prev + base, K, Kq, out);
  bool any = false;
#pragma unroll
  for (int i = 0; i < MATCH_NP; i++)
    if (act[i]) any |= match_store(S.match + base, S.snap + base, p[i], (uint32_t)out[i], (uint32_t)(out[i] >> 32));
  if (any) S.snap_used[stream] = 1;  // (every writer writes the same word)
}

// The same search for longer streams, out of LDS: a workgroup takes a tile of
// 16 Ki positions and first stages everything their chain walks can touch -- the
// 32 KiB window in front of the tile, the tile itself and the 258 bytes behind it,
// with the chain links of those positions: 48 KiB + 96 KiB of the CU's 160 KiB.
// The walks are chains of dependent, data-dependent 2- and 8-byte reads; from
// global memory they are bound by the rate at which the vector memory pipeline
// takes divergent addresses, out of LDS a wave's 64 addresses go in a few clocks.
constexpr uint32_t MATCHW_THREADS = 1024;
constexpr uint32_t MATCHW_TILE = 16384;
constexpr int MATCHW_NP = 2;  // run slots per lane.  3 and 4 measured on C2 with the shared cursor: +3 % and +7.5 % (and
                               // +3 % / +8 % on 1 MiB streams of 3-bit symbols); 3 again under the tile-wide pool: +9 % on C2,
                               // +6 % on real text: the loop is nearer its vector bound than latency-bound
#ifndef ZD_SCAN_NP
#define ZD_SCAN_NP 2
#endif
constexpr int MATCHW_SCAN_NP = ZD_SCAN_NP;  // run slots per lane of the second form of the walk
constexpr uint32_t MATCHW_LINKS = MAX_MATCH_DIST + MATCHW_TILE;           // u16 each
constexpr uint32_t MATCHW_SRC_BYTES = MAX_MATCH_DIST + MATCHW_TILE + 272;  // + MAX_MATCH_LEN + an 8-byte read, 16-aligned
#ifndef ZD_SCAN_STEPS
#define ZD_SCAN_STEPS 4
#endif
// run-slot steps per position (iterations x lanes x slots / positions) from which the second form of the walk
// pays: 4-bit symbols take 2.7 of the first form's (1.56 chain steps at 0.58 lane use), 3-bit symbols ~12, text ~48
constexpr uint32_t MATCHW_SCAN_STEPS = ZD_SCAN_STEPS;
#ifndef ZD_PROBE_DEEP
#define ZD_PROBE_DEEP 192
#endif
// of the 1024 chains a workgroup probes in its first tile, those that hold three candidates: above this many the tile takes the
// second form.  Round 5's second form costs a third less than round 4's, and the rule moved with it (it was 512, and 6 steps
// a position): the benchmark's symbols probe 143 +- 11 deep and keep the first form (1.57 ms per 4096 streams against 1.75 in the
// second); the corpus' binaries -- a few long chains among many empty ones -- now take the second: 9.1 -> 3.6 ms per 64 MiB of
// such chunks, the corpus' lz_match 17.1 -> 12.6 ms per 256 MiB (tools/sweep_form_rule.sh)
constexpr uint32_t MATCHW_PROBE_DEEP = ZD_PROBE_DEEP;
#ifndef ZD_MATCH_GROUPS_PER_WG
#define ZD_MATCH_GROUPS_PER_WG 8
#endif
constexpr size_t MATCHW_SMALL = 8192;  // streams up to this long keep the global-memory kernel
static_assert(MATCHW_TILE % MATCHW_THREADS == 0 && MATCHW_SRC_BYTES % 16 == 0, "tile shape");
static_assert(MATCHW_SRC_BYTES + 2 * MATCHW_LINKS <= 160 * 1024, "LDS of one CU");

#ifdef ZD_MATCH_PHASES  // timing-only build (tools/exp_match_phases.py): clock deltas, data paths untouched
// Slot = workgroup index mod ZD_PH_SLOTS (plain atomics on one address from every workgroup
// cost more than the kernel), 8 words each, s_memtime (shader clock) ticks unless noted:
// [0] sum over waves of staging (entry -> staging barrier)   [1] sum over waves of the match loop
// [2] waves   [3] slowest wave's loop end - staging barrier   [4] workgroups
// [5] all waves done - entry   [6] the same in s_memrealtime ticks (100 MHz): calibrates [0..5]
constexpr int ZD_PH_SLOTS = 1024;
__device__ unsigned long long zd_match_phases[ZD_PH_SLOTS * 8];
#endif

// A stream's FIRST tile has no window in front of it, so the LDS that holds a window and a tile holds three tiles' worth of
// the stream's start: tile 0 is 48 Ki positions, the others 16 Ki.  Every tile ends with its workgroup's waves running dry
// one after the other -- the pool is empty, a wave's last walks go on with few lanes, and the longest chain of the last
// positions sets how long: about 5 of a wave's 17 iterations of a 16 Ki tile on the benchmark's symbols (its lane use of
// 0.58 is mostly that) -- and a 64 KiB stream now has two such ends where it had four (round 5).
constexpr uint32_t MATCHW_TILE0 = MAX_MATCH_DIST + MATCHW_TILE;  // positions of a stream's first tile
__host__ __device__ inline uint32_t match_tile_start(uint32_t tile) { return tile == 0 ? 0u : MATCHW_TILE0 + (tile - 1u) * MATCHW_TILE; }
// tiles of a stream of len bytes (len >= 4): a tile exists when its first position can start a match
__host__ __device__ inline uint64_t match_tiles_of(uint64_t len) {
  const uint64_t last = len - 4;  // the last position that can
  return last < MATCHW_TILE0 ? 1 : 2 + (last - MATCHW_TILE0) / MATCHW_TILE;
}
// What one tile stages: source bytes [w0, src_end) and links [w0, link_end) of its stream
struct MatchTile {
  uint32_t t0, w0;            // first position of the tile, first staged position
  uint32_t t1;                // one past the tile's last position (not cut at the stream's end)
  uint32_t src_end;           // one past the last staged source byte
  uint32_t n_src, n_links;    // staged in whole 16-byte units: source bytes, links
};
__device__ __forceinline__ MatchTile match_tile(uint32_t tile, uint32_t len) {
  MatchTile g;
  g.t0 = match_tile_start(tile);
  g.t1 = match_tile_start(tile + 1u);  // (< 2^32: streams end 64 KiB below it)
  g.w0 = g.t0 > MAX_MATCH_DIST ? g.t0 - MAX_MATCH_DIST : 0;
  const uint64_t want = (uint64_t)g.t1 + 264;
  g.src_end = want < len ? (uint32_t)want : len;
  g.n_src = (g.src_end - g.w0) & ~15u;  // whole 16-byte units; the rest byte by byte
  const uint32_t link_end = g.t1 < len ? g.t1 : len;
  g.n_links = (link_end - g.w0 + 7u) & ~7u;  // the scratch is padded past len
  return g;
}

// Eight links on their way into the window: "no link", 0 in the table, is 0xFFFF there (LinkNone<WinLinks>): x - 1 wraps
// a 0 to 0xFFFF and the saturating + 1 leaves it there (two packed 16-bit instructions a word)
__device__ __forceinline__ u32x4 links_for_window(u32x4 v) {
  typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
  const u16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
  return __builtin_bit_cast(u32x4, __builtin_elementwise_add_sat((u16x8)(__builtin_bit_cast(u16x8, v) - one), one));
}

// A workgroup takes tiles_per_group consecutive tiles of one stream: between workgroups
// a CU sat idle for 4.4 us of a 29.3 us tile on C2 (tools/exp_match_phases.py).  A wave
// requests its share of the next tile's window into registers as soon as it has walked
// its own positions -- the faster waves' loads are then in flight while the workgroup
// waits for its slowest wave -- and the window goes to LDS between two barriers.
// (Requesting it BEFORE the walk, to hide the whole latency, was the first version and
// measured slower, 6.02-6.05 against 5.62-5.65 ms on C2 with the same unconditional loads:
// the walk phase of a tile ran 15.5 instead of 13.5 us.  About 1 us of that is the issue
// work itself; the rest was not explained.  That first version also had a branch around
// the source loads, whose join made every wave wait for them -- see issue() below.  Measured
// once more under the tile-wide pool, where no slow wave is left to hide the loads behind:
// before the walk is still slower, 5.82-5.85 against 5.35-5.45 ms on C2, 154.5 against 149.9 ms
// on real text.)
__global__ __launch_bounds__(MATCHW_THREADS) void lz_match_window_kernel(const uint8_t *__restrict__ src_arena,
                                                                         const StreamDesc *__restrict__ descs,
                                                                         DeflateScratch S, uint32_t n_streams,
                                                                         uint32_t tiles_per_stream,
                                                                         uint32_t tiles_per_group, uint32_t groups_per_wg, int K, int Kq,
                                                                         int form) {
  __shared__ __attribute__((aligned(16))) uint8_t win_src[MATCHW_SRC_BYTES];
  __shared__ __attribute__((aligned(16))) uint16_t win_prev[MATCHW_LINKS];
  __shared__ uint32_t pool_next;  // positions of the tile handed out to waves so far
  __shared__ uint32_t tile_iters[2];  // loop iterations of the workgroup's waves in the tile they walk (by tile parity)
  __shared__ uint32_t probe_deep;     // probed chains of a group's first tile that hold three candidates
#ifdef ZD_MATCH_PHASES
  __shared__ unsigned long long ph_acc[4];  // stage sum, loop sum, waves, latest loop end
  unsigned long long ph0 = __builtin_readcyclecounter();
  unsigned long long pr0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x < 4) ph_acc[threadIdx.x] = 0;
#endif
  if (S.error[0]) return;
  // XCD-aware order as in lz_match_kernel: the groups of a stream re-read each
  // other's windows, so they go to one XCD's L2
  const uint32_t nb = gridDim.x;
  const uint32_t per_xcd = (nb + 7) / 8;
  const uint32_t logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const uint32_t groups_per_stream = (tiles_per_stream + tiles_per_group - 1) / tiles_per_group;
  // A workgroup takes groups_per_wg GROUPS one behind the other -- consecutive ones: the next group of its stream, or the next
  // stream's first -- and a group's first window is requested like any next tile's: while the waves finish the tile before.
  // (A workgroup per group paid for that window's memory round trip with all its waves waiting, 5 of a tile's 39 us on the
  // benchmark's streams -- tools/exp_match_phases.py -- and for a CU's turn from one workgroup to the next.)
  const uint64_t n_groups = (uint64_t)n_streams * groups_per_stream;
  uint64_t item = (uint64_t)logical * groups_per_wg;
  const uint64_t item_end = item + groups_per_wg < n_groups ? item + groups_per_wg : n_groups;
  struct Group {  // a group of tiles of one stream
    uint32_t stream, len, tile, tile_end;
    uint64_t base;
    const uint8_t *s;
  };
  // the next group from `item` on that has work (a stream too short or too long has none), or false
  auto open = [&](Group &c) -> bool {
    for (; item < item_end; item++) {
      c.stream = (uint32_t)(item / groups_per_stream);
      const uint32_t group = (uint32_t)(item % groups_per_stream);
      const StreamDesc sd = descs[c.stream];
      if (sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) continue;
      c.len = (uint32_t)sd.src_len;
      // tiles [tile, tile_end) of the stream; a tile exists when its first position can start a match
      const uint32_t stream_tiles = (uint32_t)match_tiles_of(c.len);
      c.tile = group * tiles_per_group;
      if (c.tile >= stream_tiles) continue;
      c.tile_end = c.tile + tiles_per_group < stream_tiles ? c.tile + tiles_per_group : stream_tiles;
      c.base = S.pos_base[c.stream];
      c.s = src_arena + sd.src_off;
      item++;
      return true;
    }
    return false;
  };
  const uint32_t tid = threadIdx.x;
  constexpr int SRC_ROUNDS = (MATCHW_SRC_BYTES / 16 + MATCHW_THREADS - 1) / MATCHW_THREADS;
  constexpr int LINK_ROUNDS = (MATCHW_LINKS / 8 + MATCHW_THREADS - 1) / MATCHW_THREADS;
  u32x4 vs[SRC_ROUNDS], vl[LINK_ROUNDS];
  // every thread issues all its 16-byte loads of a window at once (clamped indices) ...
  auto issue = [&](const Group &c, uint32_t w0, uint32_t n_src, uint32_t n_links) {
    const uint16_t *pv = S.prev + c.base + w0;
    const uint32_t last_src = n_src ? n_src - 16u : 0u;
    // A stream shorter than one unit (in a batch of long ones) has no source to load here; its
    // lanes read scratch instead and the values are never stored.  NOT a branch around the
    // loads: at the join the compiler waits for every load issued so far (vmcnt(0)), which
    // put one full memory latency between the source and the link loads of every tile.
    const uint8_t *sp = n_src ? c.s + w0 : (const uint8_t *)pv;
#pragma unroll
    for (int j = 0; j < SRC_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MATCHW_THREADS) * 16u;
      vs[j] = load16_unaligned(sp + (o < n_src ? o : last_src));
    }
#pragma unroll
    for (int j = 0; j < LINK_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MATCHW_THREADS) * 8u;
      vl[j] = *(const u32x4 *)(pv + (o < n_links ? o : n_links - 8u));
    }
  };
  // ... and stores them to the window later
  auto store = [&](const Group &c, const MatchTile &g) {
#pragma unroll
    for (int j = 0; j < SRC_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MATCHW_THREADS) * 16u;
      if (o < g.n_src) *(u32x4 *)(win_src + o) = vs[j];
    }
#pragma unroll
    for (int j = 0; j < LINK_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MATCHW_THREADS) * 8u;
      if (o < g.n_links) *(u32x4 *)(win_prev + o) = links_for_window(vl[j]);
    }
    if (tid < ((g.src_end - g.w0) & 15u)) win_src[g.n_src + tid] = c.s[g.w0 + g.n_src + tid];
  };
  Group c;
  if (!open(c)) return;
  MatchTile g = match_tile(c.tile, c.len);
  issue(c, g.w0, g.n_src, g.n_links);
  store(c, g);
  if (tid == 0) { pool_next = 0; tile_iters[0] = 0; tile_iters[1] = 0; probe_deep = 0; }
  __syncthreads();
  // Two forms of the walk (deflate_lane.h): the first reads every candidate's 8 bytes and is the
  // faster one where chains hold a candidate or two (the benchmark's 4-bit symbols: 1.2 per
  // position); the second looks at two bytes first and compares in bulk, and is the faster one on
  // longer chains (3-bit symbols 8.5, text 34 per position).  Same results; a workgroup takes the second form for a tile
  // when the tile before it took more than MATCHW_SCAN_STEPS run-slot steps per position.  A group's FIRST tile has no tile
  // before it (and is three quarters of a 64 KiB stream since round 5): every thread follows the links of one position of
  // the tile through three candidates -- three dependent reads of the window the barrier above has just completed -- and the
  // tile takes the second form when more than MATCHW_PROBE_DEEP of the probed chains get that far (the benchmark's symbols:
  // one in seven; 3-bit symbols and text: nine in ten).  Only speed depends on it.
  // form: 0 that rule, 1 / 2 always the first / second form (ZIPC_HIP_MATCH_FORM: tests and tuning).
  bool scan_form = form == 2;
  bool first_of_group = true;
  for (;;) {
    if (first_of_group && form == 0) {
      const uint32_t t_last = (uint64_t)g.t1 < (uint64_t)c.len - 3 ? g.t1 : c.len - 3;  // one past the tile's last position
      const uint32_t n_pos = t_last - g.t0;                                                // (>= 1)
      uint32_t p = g.t0 + (uint32_t)(((uint64_t)tid * n_pos) >> 10), hops = 0;
      const uint16_t *wp0 = win_prev - g.w0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const uint32_t d = wp0[p];
        const bool on = d != LinkNone<WinLinks>::value && p - d >= g.w0 && hops == (uint32_t)k;
        p = on ? p - d : p;
        hops += on ? 1u : 0u;
      }
      const unsigned long long deep = __builtin_amdgcn_ballot_w64(hops == 3u);
      if ((tid & 63u) == 0) atomicAdd(&probe_deep, (uint32_t)__builtin_popcountll(deep));
      __syncthreads();
      scan_form = probe_deep > MATCHW_PROBE_DEEP;
    }
#ifdef ZD_MATCH_PHASES
    const unsigned long long ph1 = __builtin_readcyclecounter();
#endif
    const bool has_next = c.tile + 1 < c.tile_end;  // (of the same group; uniform over the workgroup)
    const uint8_t *ws = win_src - g.w0;  // indexed by stream position
    const WinLinks wp{win_prev - g.w0};
    // The tile's positions are ONE pool for the workgroup's 16 waves (lz_match_runs_pool): a wave
    // fetches chunks of 256 from pool_next and hands them to its run slots as they finish.  The
    // first schedule gave every wave a fixed 1 Ki positions and every lane every 64th of them: a
    // tile then took as long as its slowest wave (a fifth of a tile on the benchmark's symbols by
    // the phase timers, more on text) and a wave as long as its slowest lane.  Same box, the pool
    // against that: 5.46 vs 5.83-5.87 ms on C2, 171.8 vs 174.7 ms on C4, 153 vs 241 ms on real text.
    // the parse reads up to PARSE_PAD entries behind the last position without a range test
    if ((uint64_t)g.t1 > (uint64_t)c.len - 4 && tid < PARSE_PAD) S.match[c.base + (c.len - 3) + tid] = 0;
    {
      const uint64_t tend64 = (uint64_t)g.t1 < (uint64_t)c.len - 3 ? (uint64_t)g.t1 : (uint64_t)c.len - 3;
      uint32_t two = 0;  // some position of mine left a second answer
      auto sink = [&](uint32_t p, uint32_t best, uint32_t first) { two |= match_store(S.match + c.base, S.snap + c.base, p, best, first) ? 1u : 0u; };
      // (the second form in coordinates of its own: the LDS addresses of the window's bytes -- its range test is signed, and
      // its reads take no base)
      typedef __attribute__((address_space(3))) uint8_t lds_u8;
      const uint32_t lds_src = (uint32_t)(uintptr_t)(lds_u8 *)win_src, lds_links = (uint32_t)(uintptr_t)(lds_u8 *)win_prev;
      const uint32_t off_w = g.w0 - lds_src;  // coordinate + off_w = stream position
      auto sink_w = [&](uint32_t p, uint32_t best, uint32_t first) { sink(p + off_w, best, first); };
      TilePool pool{&pool_next, g.t0, (uint32_t)tend64, tid & 63u, 0u}, pool_w{&pool_next, g.t0 - off_w, (uint32_t)tend64 - off_w, tid & 63u, POOL_TAPER};
      const uint32_t iters = scan_form ? lz_match_scan_pool<MATCHW_SCAN_NP>((const uint8_t *)win_src - lds_src, c.len - off_w, pool_w, (uint32_t)tend64 - off_w, tid & 63u,
                                                                          WinLinks{win_prev - lds_src}, lds_links - 2u * lds_src, K, Kq, sink_w)
                                       : lz_match_runs_pool<MATCHW_NP>(ws, c.len, pool, (uint32_t)tend64, tid & 63u, wp, K, Kq, sink);
      if (two) S.snap_used[c.stream] = 1;  // (every writer writes the same word; read by the parse, a kernel later)
      if (has_next && (tid & 63u) == 0) atomicAdd(&tile_iters[c.tile & 1u], iters);
    }
#ifdef ZD_MATCH_PHASES
    {
      const unsigned long long ph2 = __builtin_readcyclecounter();
      if ((tid & 63u) == 0) {
        atomicAdd(&ph_acc[0], ph1 - ph0);
        atomicAdd(&ph_acc[1], ph2 - ph1);
        atomicAdd(&ph_acc[2], 1ull);
        atomicMax(&ph_acc[3], ph2);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned long long *slot = zd_match_phases + (size_t)(blockIdx.x % ZD_PH_SLOTS) * 8;
        atomicAdd(slot + 0, ph_acc[0]);
        atomicAdd(slot + 1, ph_acc[1]);
        atomicAdd(slot + 2, ph_acc[2]);
        atomicAdd(slot + 3, ph_acc[3] - ph1);
        atomicAdd(slot + 4, 1ull);  // tiles
        atomicAdd(slot + 5, (unsigned long long)__builtin_readcyclecounter() - ph0);
        atomicAdd(slot + 6, (unsigned long long)__builtin_amdgcn_s_memrealtime() - pr0);
        ph_acc[0] = 0; ph_acc[1] = 0; ph_acc[2] = 0; ph_acc[3] = 0;
      }
      ph0 = __builtin_readcyclecounter();
      pr0 = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // what comes next: the group's next tile, or the next group's first
    Group n = c;
    if (has_next) n.tile++;
    else if (!open(n)) break;
    const MatchTile gn = match_tile(n.tile, n.len);
    issue(n, gn.w0, gn.n_src, gn.n_links);  // in flight while the slower waves finish
    __syncthreads();  // every wave is done with this tile's window
    // run-slot steps the tile took per position (a tile with a successor is a full one): chain steps / lane use
    if (has_next && form == 0) scan_form = (uint64_t)tile_iters[c.tile & 1u] * (64u * (scan_form ? MATCHW_SCAN_NP : MATCHW_NP)) > (uint64_t)MATCHW_SCAN_STEPS * (g.t1 - g.t0);
    store(n, gn);
    // The next tile's counters: nobody touches them now.  NOT the counter the line above reads -- a wave may still be on its way
    // to that read (round 5 cleared both here: a slow wave could then read 0 and take another form than the rest of its
    // workgroup; the results were the same, the form rule and its timings were not deterministic).  With a successor the next
    // tile's parity is the other one, and this tile's counter is cleared a tile later; without one nobody reads either.
    if (tid == 0) {
      pool_next = 0;
      tile_iters[n.tile & 1u] = 0;
      if (!has_next) tile_iters[(n.tile & 1u) ^ 1u] = 0;
      probe_deep = 0;
    }
    __syncthreads();
    first_of_group = !has_next;
    c = n;
    g = gn;
  }
}

// ---------------------------------------------------------------------------------
// The lazy parse proper (Lz77.compress zd.ml:1203-1244 + write_block_symbol
// zd.ml:1118-1123): which positions does the parse visit, what symbols do they
// emit, where are blocks cut.  One wave per stream walks the stream in tiles of
// 64 positions (one per lane).  Inside a tile the serial walk "p -> p + advance"
// is replaced by pointer doubling: J_k[t] = position reached from t after 2^k
// steps (shuffles; indices >= 64 are exits into later tiles); then every lane
// decides whether it is on the path from the tile's entry position by a
// descending search through those tables.  Visited lanes get their symbol index
// from a wave scan and write their symbols.  The only serial dependency left
// between tiles is the entry position.
constexpr int PARSE_TILE = 64;
#ifdef ZD_PARSE_PHASES  // timing-only build: shader clocks of a wave's tiles, summed: [0] tiles [1] loads issued -> macro steps done [2] -> visited
// known (doubling, search) [3] -> symbols written [4] -> tile done (block cut, next entry, skipped tiles' loads)
static __device__ unsigned long long zd_parse_phases[8];
#endif
#ifdef ZD_PARSE_COUNTS  // counting-only build: [0] tiles, [1] turns of the lazy chains' loop, [2] tiles with a turn, [3] chaining lanes over the turns
static __device__ unsigned long long zd_parse_counts[8];
#define ZD_PCOUNT(i, v) do { if (lane == 0) atomicAdd(&zd_parse_counts[i], (unsigned long long)(v)); } while (0)
#else
#define ZD_PCOUNT(i, v) ((void)0)
#endif

// The macro step of the 64 positions of a tile, one per lane (lz_macro_position, with the following
// positions' entries taken from the neighbouring lanes: m_cur is this tile's entry of the lane, m_nxt the
// next tile's).  br: the match the step ends with (0: the position is a literal), lits: the literals in front of it
// (0 for a literal position), adv: the step's advance, literals + match length (1 for a literal position, 0 for a lane behind the
// stream's end).  valid_m: the lanes whose position lies in the stream, as a mask.
__device__ __forceinline__ void parse_tile_macro(int lane, uint32_t p, unsigned long long valid_m, bool has_match, uint32_t max_pos,
                                               uint32_t len, uint64_t m_cur, uint64_t m_nxt,
                                               const uint32_t *__restrict__ match, const uint32_t *__restrict__ snap, int good_match, uint32_t &br,
                                               uint32_t &adv, uint32_t &lits) {
#ifndef ZD_PARSE_CHAIN_INTS
  // The lanes whose lazy chain goes on are a LANE MASK (round 6).  Rounds 3-5 kept the flag as an integer in a vector
  // register -- a select to make it, a compare to read it back, twice a turn -- because the kernel was thought to wait for
  // scalar issue; it waits for vector issue, and a mask's and / or are scalar instructions.
  // A turn: 29 -> 20 vector instructions (-DZD_PARSE_CHAIN_INTS keeps the old form for A/B runs).
  // Round 6, second pass over the loop's assembly (22 -> 13 vector instructions a turn):
  //  * a chaining lane's j is p + the turn: every lane of chain_m has taken every turn so far (a lane that does not take one
  //    leaves the mask for good), so j is not a register -- and "j <= max_pos" follows from "pl < maxlen" (a pending length is
  //    at least 3, so len - j > 3);
  //  * the literals are counted by an add-with-carry whose carry is the take mask (one instruction; the compiler's form is a
  //    select of 0 / 1 and an add, and it has no builtin for this one);
  //  * the shifts are in place: wave_shl with "a lane without a source reads 0" and lane 63 written by v_writelane (with an
  //    "old" operand the compiler copies the register every turn).
  auto ballot = [](bool b) { return (unsigned long long)__builtin_amdgcn_ballot_w64(b); };
  auto mine = [](unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); };
  const unsigned long long ok_m = has_match ? valid_m & ballot(p <= max_pos) : 0ull;
  uint32_t pend = mine(ok_m) ? (uint32_t)m_cur : 0u;
  unsigned long long chain_m = ok_m & ballot((pend & 0x1FF) != 0);
  uint32_t n_lits = 0;
  const uint32_t lenp = len - p;  // (lanes behind the end: never in chain_m)
  const uint32_t cur_hi = (uint32_t)(m_cur >> 32), nxt_hi = (uint32_t)(m_nxt >> 32);
  // one step of the lanes' lazy chains with the entry c of position j = p + ahead (zd.ml:1224-1240): which lanes take it
  auto chain_step = [&](uint32_t c, uint32_t rem) {
    const uint32_t pl = pend & 0x1FF;
    const uint32_t maxlen = rem < (uint32_t)MAX_MATCH_LEN ? rem : (uint32_t)MAX_MATCH_LEN;
    const unsigned long long take_m = chain_m & ballot(pl < maxlen) & ballot((c & 0x1FF) > pl);
    pend = mine(take_m) ? c : pend;
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(n_lits), "=s"(carry_out) : "v"(n_lits), "s"(take_m));  // n_lits += take
    chain_m = take_m;
  };
  uint32_t sh_cur = (uint32_t)m_cur, sh_nxt = (uint32_t)m_nxt;  // after k turns: the words of positions p + k (this tile's lanes), p + 64 + k
  ZD_PCOUNT(0, 1); ZD_PCOUNT(2, chain_m ? 1 : 0);
  uint32_t ahead = 1;
  for (; ahead < 128u - 63u && chain_m; ahead++) {
    ZD_PCOUNT(1, 1); ZD_PCOUNT(3, __builtin_popcountll(chain_m));
    {  // the two tiles' best-of-K words shifted down a lane (wave_shl:1: lane i takes lane i + 1's, lane 63 the next tile's lane 0)
      // (by hand: four instructions, in place.  The leading s_nop: a DPP read wants two wait states behind the register's
      // writer, and the compiler does not look into an asm; the two shifts are the wait states between v_readlane and the
      // v_writelane that reads its scalar)
      uint32_t n0;
      asm("s_nop 1\n\t"
          "v_readlane_b32 %2, %1, 0\n\t"
          "v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_writelane_b32 %0, %2, 63"
          : "+v"(sh_cur), "+v"(sh_nxt), "=&s"(n0));
    }
    uint32_t c = sh_cur;
    // best-of-K/4 only where a pending match is that long (and only where the stream has second answers at all: else hi = lo)
    const unsigned long long want_hi = chain_m & ballot((pend & 0x1FF) >= (uint32_t)good_match);
    if (want_hi) {
      const uint32_t off = (uint32_t)lane + ahead;
      const uint32_t addr = (off & 63u) * 4u;
      const uint32_t a_hi = lane_value(addr, cur_hi), b_hi = lane_value(addr, nxt_hi);
      c = mine(want_hi) ? (off < 64u ? a_hi : b_hi) : c;
    }
    chain_step(c, lenp - ahead);
  }
  // a chain of 64 strictly growing matches and more: straight from the table
  // (kept out of the loop above: its load would make that loop wait for memory)
  while (chain_m) {
    const uint32_t j = p + ahead;
    const uint64_t mj = mine(chain_m) && j <= max_pos ? match_pair(match, snap, j) : 0ull;
    chain_step((pend & 0x1FF) >= (uint32_t)good_match ? (uint32_t)(mj >> 32) : (uint32_t)mj, lenp - ahead);
    ahead++;
  }
  br = pend;  // (0 where no match is pending: the lanes outside ok_m, and the positions without a match)
  lits = n_lits;
  const uint32_t step = n_lits + (pend & 0x1FF);
  adv = mine(valid_m) ? (step ? step : 1u) : 0u;
#else
  // (flags as integers and selects instead of branches: a loop-carried bool lives in a scalar mask that
  // costs three scalar instructions per update, and the CU's ONE scalar issue per clock is what this
  // kernel's 32 waves per CU queue for)
  const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
  uint32_t pend = (valid && has_match && p <= max_pos) ? (uint32_t)m_cur : 0u;
  uint32_t chaining = (pend & 0x1FF) != 0 ? 1u : 0u;
  uint32_t n_lits = 0, j = p + 1;
  const uint32_t cur_lo = (uint32_t)m_cur, cur_hi = (uint32_t)(m_cur >> 32);
  const uint32_t nxt_lo = (uint32_t)m_nxt, nxt_hi = (uint32_t)(m_nxt >> 32);
  // one step of a lane's lazy chain with the entry mj of position j
  auto chain_step = [&](uint32_t mj_lo, uint32_t mj_hi, uint32_t use_hi) {
    const uint32_t pl = pend & 0x1FF;
    const uint32_t rem = len - j;  // (j > max_pos: the value is not used)
    const uint32_t maxlen = rem < (uint32_t)MAX_MATCH_LEN ? rem : (uint32_t)MAX_MATCH_LEN;
    const uint32_t c = use_hi ? mj_hi : mj_lo;
    const uint32_t take = (chaining != 0 && j <= max_pos && pl < maxlen && (c & 0x1FF) > pl) ? 1u : 0u;
    n_lits += take;
    pend = take ? c : pend;
    j += take;
    chaining = take;
  };
  // chains within the staged tiles (all but pathological ones).  Every chaining lane looks at the same distance ahead,
  // lane + ahead (< 128), so the two tiles' best-of-K words are SHIFTED down a lane per turn (v_mov_b32_dpp wave_shl:1:
  // lane i takes lane i + 1's, lane 63 the next tile's lane 0 -- tools/probes/dpp_wave_shl.hip) where round 3 fetched
  // them by two ds_bpermute a turn: no LDS operation and no address -- and no measurable difference (C2 lz_parse 2.90-2.93
  // against 2.86-2.97 ms beside the other slice's kernels): the shuffles' round trips are not what a tile waits for.
  uint32_t sh_cur = cur_lo, sh_nxt = nxt_lo;  // after k turns: the words of positions p + k (this tile's lanes), p + 64 + k
  ZD_PCOUNT(0, 1); ZD_PCOUNT(2, __builtin_amdgcn_ballot_w64(chaining != 0) ? 1 : 0);
  for (uint32_t ahead = 1; ahead < 128u - 63u && __builtin_amdgcn_ballot_w64(chaining != 0); ahead++) {
    ZD_PCOUNT(1, 1); ZD_PCOUNT(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(chaining != 0)));
    const uint32_t off = (uint32_t)lane + ahead;
    const uint32_t addr = (off & 63u) * 4u;
    const bool in_cur = off < 64u;
    // best-of-K of position j; best-of-K/4 only if a pending match is that long
    {
      const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)sh_nxt, 0);
      sh_cur = (uint32_t)__builtin_amdgcn_update_dpp((int)n0, (int)sh_cur, 0x130, 0xf, 0xf, false);
      sh_nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sh_nxt, 0x130, 0xf, 0xf, false);
    }
    const uint32_t mj_lo = sh_cur;
    uint32_t mj_hi = 0;
    const uint32_t want_hi = (chaining != 0 && (pend & 0x1FF) >= (uint32_t)good_match) ? 1u : 0u;
    if (__builtin_amdgcn_ballot_w64(want_hi != 0)) {
      const uint32_t a_hi = lane_value(addr, cur_hi), b_hi = lane_value(addr, nxt_hi);
      mj_hi = in_cur ? a_hi : b_hi;
    }
    chain_step(mj_lo, mj_hi, want_hi);
  }
  // a chain of 64 strictly growing matches and more: straight from the table
  // (kept out of the loop above: its load would make that loop wait for memory)
  if (__builtin_amdgcn_ballot_w64(chaining != 0)) {
    while (chaining) {
      const uint64_t mj = j <= max_pos ? match_pair(match, snap, j) : 0ull;
      chain_step((uint32_t)mj, (uint32_t)(mj >> 32), (pend & 0x1FF) >= (uint32_t)good_match ? 1u : 0u);
    }
  }
  br = 0; lits = 0; adv = valid ? 1u : 0u;
  if ((pend & 0x1FF) != 0) { br = pend; lits = n_lits; adv = n_lits + (pend & 0x1FF); }
#endif
}

// Streams parsed by several waves (lz_parse_spec_kernel / lz_parse_stitch_kernel / lz_parse_gather_kernel below):
// what the waves leave for each other, per stream at the stream's position base (tiles: base / 64, segments: seg_base).
// positions per segment (ParseSegs::seg_positions): a multiple of the tile, far above the longest step (63 + 512);
// chosen per call -- the stitch takes a couple of microseconds per segment, one after the other, the waves of
// lz_parse_spec_kernel a third of a microsecond per tile, side by side
constexpr uint32_t PARSE_SEG_MIN = 4096;
constexpr uint32_t PARSE_SEG_SLACK = 576;  // symbols of a segment at most: one per position before its last step, and that step's
struct ParseSegs {
  uint32_t *spec_syms;            // [segments * seg_syms] a segment's symbols as parsed from its first position
  unsigned long long *vis;        // [P / 64] per tile: the positions on the path (zeroed per call: a tile jumped over has none)
  uint32_t *tile_sym0;            // [P / 64] per tile: symbols of its segment before it
  uint32_t *seg_exit, *seg_total; // [segments] where the segment's parse left it (>= its end), and its symbols
  uint32_t *seg_dst, *seg_from, *seg_n;  // [segments] the stitch's verdict: spec symbols [from, from + n) go to the stream's symbols at dst
  unsigned long long *vis2;       // [P / 64] per tile that was parsed again: the true path, and
  uint32_t *sym02;                // [P / 64]   the symbols parsed again before the tile
  uint32_t *meet_syms;            // [segments * MEET_CAP] lz_parse_meet_kernel: the symbols parsed again from the expected entry
  uint32_t *meet_f, *meet_from, *meet_exit, *meet_end;  // [segments] ... and what parse_again returned (f = ~0: no room)
  uint32_t *fix_dst, *fix_n;      // [segments] the stitch's verdict on them: meet_syms[0, n) go to the stream's symbols at dst
  uint32_t segs_per_stream;       // of the longest stream: the grids are streams x segs_per_stream (a block whose segment lies behind its stream's end leaves at once)
  // Segment slot of (stream, k): COMPACT, not stream * segs_per_stream + k -- a batch of ragged members (492 files of a library
  // directory, one of them 233 MB: 14 247 segments) sized by its longest stream asked for 475 GB of symbols (round 6:
  // tools/corpus_box.py met it).  pos_base is the prefix sum of the streams' padded positions, a stream has at most
  // padded / seg_positions + 1 segments, so pos_base / seg_positions + stream + k never collides and stays below
  // cap_positions / seg_positions + n_streams.
  __device__ __forceinline__ size_t slot(const DeflateScratch &S, uint32_t stream, uint32_t k) const {
    return (size_t)(S.pos_base[stream] / seg_positions) + stream + k;
  }
  uint32_t seg_positions, seg_syms;  // positions per segment; symbol slots per segment (seg_positions + PARSE_SEG_SLACK)
};

// MODE 0: one wave parses a whole stream and cuts its blocks.  MODE 1: one wave parses ONE SEGMENT of a stream as if
// a symbol started at the segment's first position, into ParseSegs (no blocks: lz_parse_stitch_kernel).
template <int MODE, bool SNAP>
__device__ __forceinline__ void lz_parse_wave(const uint8_t *__restrict__ src_arena,
                                              const StreamDesc *__restrict__ descs,
                                              DeflateScratch S, int good_match, uint32_t stream, uint32_t seg,
                                              ParseSegs G) {
  if (S.error[0]) return;
  const int lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (sd.src_len > MAX_STREAM_LEN) {
    if (MODE == 0 && lane == 0) S.n_blocks[stream] = 0;
    return;
  }
  const uint32_t len = (uint32_t)sd.src_len;
  const uint8_t *s = src_arena + sd.src_off;
  const uint64_t base = S.pos_base[stream];
  const uint32_t *match = S.match + base, *snap = S.snap + base;
  const bool has_match = len >= (uint32_t)MIN_MATCH_LEN;
  const uint32_t max_pos = has_match ? len - MIN_MATCH_LEN : 0;  // positions <= max_pos have a match entry
  if (MODE == 1 && (uint64_t)seg * G.seg_positions >= len && !(len == 0 && seg == 0)) return;
  const uint32_t B0 = MODE == 1 ? seg * G.seg_positions : 0u;  // where this wave starts,
  const uint32_t lim = MODE == 1 ? (len - B0 > G.seg_positions ? B0 + G.seg_positions : len) : len;  // and the tiles it takes: those below lim
  const size_t seg_slot = MODE == 1 ? G.slot(S, stream, seg) : 0;
  uint32_t *syms = MODE == 1 ? G.spec_syms + seg_slot * G.seg_syms : S.syms + base;
  BlockDesc *blocks = MODE == 1 ? nullptr : S.blocks + S.blk_base[stream];

  if (MODE == 1 && len < 4u) {  // (the stitch handles streams this short itself)
    if (lane == 0) { G.seg_exit[seg_slot] = len; G.seg_total[seg_slot] = 0; }
    return;
  }
  if (MODE == 0 && len != 0 && !has_match) {  // 1 to 3 bytes: literals, one block (and the loop below may load whole words of source)
    if ((uint32_t)lane < len) syms[lane] = s[lane];
    if (lane == 0) {
      BlockDesc b;
      b.src_start = 0; b.src_len = len; b.sym_start = 0; b.n_syms = len;
      blocks[0] = b;
      S.n_blocks[stream] = 1;
    }
    return;
  }
  uint32_t entry = B0, nsym = 0, blk_start = 0, blk_sym_start = 0, nblk = 0;  // wave-uniform
  uint32_t B = B0;
  // Table entries and source bytes of four tiles are kept in registers: the current
  // one, the next one (the lazy chains look into it) and the two after, the last of which is
  // requested while the current tile is worked on -- three tiles ahead of its first use
  // (two ahead: 1 % slower; the waves wait for memory a little less).
  // The loads are unconditional (clamped index, value selected afterwards) so that
  // the wait counters stay exact and nothing waits for the newest requests.
  // (lz_match zeroes the PARSE_PAD table entries behind the last position, so
  // the table is read without a range test; source bytes past the end are never used)
  // (addresses are a wave-uniform base, made scalar explicitly, plus a small lane part:
  // per-lane 64-bit address arithmetic costs vector instructions this kernel is bound by)
  // (no branch and no select around a load: the compiler waits for EVERYTHING in flight at the join of a
  // conditional load, and a select on the loaded value is an instruction it may place early)
  // (SNAP: the stream has positions whose best of the first K/4 is not their best of the first K -- S.snap_used -- and both
  // tables are read side by side, unconditionally like every load here; else one word a position is all there is)
  auto load_match = [&](uint32_t tile) -> uint64_t {
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
    const uint32_t lo = (match + t)[lane];  // the tile's entries
    if (!SNAP) return (uint64_t)lo | ((uint64_t)lo << 32);
    const uint32_t second = (snap + t)[lane];
    return (uint64_t)(lo & ~MATCH_SNAP) | ((uint64_t)((lo & MATCH_SNAP) ? second : lo) << 32);
  };
  // A lane's source byte travels as the 4-byte word it was loaded in and is extracted where it is used
  // (lit_byte) -- NOT loaded as a byte: a byte load is zero-extended by an instruction the compiler put at
  // the end of the iteration that issued the load (where the value moves into the next iteration's variable),
  // and with it an s_waitcnt vmcnt(0): every tile waited for the loads it had just requested for two tiles
  // ahead, and for its own symbol stores.
  // The lane's byte is byte i = min(tile + lane, len - 1), its word starts at c = min(tile + lane, len - 4) (an in-bounds word
  // that holds byte i; len >= 4 here, len <= MAX_STREAM_LEN: no wrap) -- formed as a SCALAR part min(tile, len - 4) and a lane
  // part min(lane, len - 4 - that): one vector instruction per load.  And in every tile but a stream's last (tile + 67 <= len)
  // c = i: the byte is the word's lowest, found by a mask where the general form is seven instructions (round 6: 14 of a tile's
  // ~130 vector instructions were these two extractions).
  auto lit_word = [&](uint32_t t, uint32_t &tcl) -> uint32_t {
    tcl = t < len - 4u ? t : len - 4u;
    const uint32_t room = len - 4u - tcl;
    return (uint32_t)lane < room ? (uint32_t)lane : room;
  };
  auto load_lit = [&](uint32_t tile) -> uint32_t {
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
    uint32_t tcl;
    const uint32_t v = lit_word(t, tcl);
    return load_u32_le((s + tcl) + v);
  };
  auto lit_byte = [&](uint32_t raw, uint32_t tile) -> uint32_t {
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
    if (t + 67u <= len) return raw & 0xFFu;  // (wave-uniform)
    uint32_t tcl;
    const uint32_t c = lit_word(t, tcl) + tcl;
    uint32_t i = (t < len ? t : len - 1u) + (uint32_t)lane;
    i = i < len ? i : len - 1u;
    return (raw >> 8u * (i - c)) & 0xFFu;
  };
  // One tile.  The three sets of registers -- current tile, next tile, the one requested now -- change
  // roles from tile to tile BY NAME (the loop below calls this three times with the sets rotated): moved from
  // variable to variable at the end of an iteration, a value that had just been requested had to arrive first
  // (s_waitcnt vmcnt(0) before the v_mov), and every tile waited a full memory round trip for loads it would
  // only need two tiles later.
#ifdef ZD_PARSE_PHASES
  unsigned long long pp[5] = {0, 0, 0, 0, 0};
#define ZD_PP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pp[i] += now_ - pp_t; pp_t = now_; } while (0)
#else
#define ZD_PP(i) ((void)0)
#endif
  auto tile_step = [&](uint64_t &m_cur, uint64_t &m_nxt, uint64_t &m_nx2, uint64_t &m_nx3, uint32_t &lit_cur,
                       uint32_t &lit_nxt, uint32_t &lit_nx2, uint32_t &lit_nx3) {
#ifdef ZD_PARSE_PHASES
    unsigned long long pp_t = __builtin_readcyclecounter();
    pp[0] += 1;
#endif
    uint32_t Bn = B + PARSE_TILE;
    m_nx3 = load_match(Bn + 2u * PARSE_TILE);
    lit_nx3 = load_lit(Bn + 2u * PARSE_TILE);

    const uint32_t p = B + (uint32_t)lane;
    // (the lanes inside the stream as a mask, taken where the compare is: a ballot of a flag that comes from elsewhere goes
    // through a 0 / 1 in a vector register and a second compare)
    const unsigned long long valid_m = __builtin_amdgcn_ballot_w64(p < len);
    // macro step of every position of the tile (lz_macro_position, with the
    // following positions' matches taken from the neighbouring lanes)
    uint32_t br, adv, lits;
    parse_tile_macro(lane, p, valid_m, has_match, max_pos, len, m_cur, m_nxt, match, snap, good_match, br, adv, lits);
    ZD_PP(1);
    const uint32_t cnt = lits + 1u;  // (of a visited lane: a literal position's one symbol, or the literals and the match)
    // J[k]: position reached after 2^k steps as a ds_bpermute address; a step that
    // leaves the tile points to itself, so the tables need no range tests
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const uint32_t j0 = (uint32_t)lane + adv;  // <= 63 + 512
    // (six tables, 1 .. 32 steps: a path has at most 63 steps inside a tile -- every step advances -- and 32 + 16 + .. + 1
    // of them reach every one of its nodes from the entry; a seventh table and search step were there until round 5)
    uint32_t J[6];
    J[0] = j0 < (uint32_t)PARSE_TILE ? j0 * 4u : lane4;
#pragma unroll
    for (int k = 1; k < 6; k++) J[k] = lane_value(J[k - 1], J[k - 1]);
    // Is lane t on the path from the entry?  Every lane searches the path for the
    // largest element <= t, descending through the 2^k-step tables (the path is
    // strictly increasing): 6 shuffles, no memory.
    uint32_t v = (entry - B) * 4u;  // < 64: tiles the parse jumps over are skipped below
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      const uint32_t y = lane_value(v, J[k]);
      if (y <= lane4) v = y;
    }
    // (the visited lanes as a mask first: every ballot below is a compare's own -- a compound condition's ballot goes through a
    // 0 / 1 in a vector register and a second compare)
    const unsigned long long vis_m = valid_m & __builtin_amdgcn_ballot_w64(v == lane4);
    const bool visited = __builtin_amdgcn_inverse_ballot_w64(vis_m);
    ZD_PP(2);
    if (MODE == 1) {
      if (lane == 0) { G.vis[(base + B) >> 6] = vis_m; G.tile_sym0[(base + B) >> 6] = nsym; }
    }
    // the path leaves the tile after the last visited position (lane 63's answer)
    const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    uint32_t next_entry = B + (uint32_t)__builtin_amdgcn_readlane((int)j0, (int)(last >> 2));
    if (next_entry > len) next_entry = len;
    // symbol indices: exclusive scan of cnt over the visited lanes
    // (the scan's result goes through an empty asm: seeing through it, the compiler forms "incl - x" as the sum of the six
    // shifted pieces, keeps each in a register of its own and spends 15 vector instructions where the scan is 6 and the difference 1)
    const uint32_t mycnt = visited ? cnt : 0u;
    uint32_t incl = wave_scan_incl(mycnt);
    asm("" : "+v"(incl));
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);  // scalar: nsym stays in an SGPR
    const uint32_t first_rel = incl - mycnt;  // within the tile's symbols (< 64 * 513)
    const uint32_t first = nsym + first_rel;
    uint32_t *tsyms = syms + nsym;  // scalar base of the tile's symbols
    // symbols (lz_emit_position): a literal position writes its byte, a match
    // position its deferral literals -- the bytes of the next positions, taken
    // from their lanes -- then its match
    {
      const uint32_t byte_cur = lit_byte(lit_cur, B), byte_nxt = lit_byte(lit_nxt, Bn);
      // (a ballot of a compound condition makes the compiler turn its mask into 0 / 1 in a vector register and compare that
      // again: two vector instructions; the compare's own ballot and a scalar `and` are one -- round 6, like the chains' masks)
      const unsigned long long lit_m = vis_m & __builtin_amdgcn_ballot_w64(lits != 0);  // (literals in front of a match)
      uint32_t k = 0;
      while (lit_m & __builtin_amdgcn_ballot_w64(k < lits)) {  // rarely more than one turn
        const uint32_t at = (uint32_t)lane + k;
        const uint32_t a = lane_value((at & 63u) * 4u, byte_cur), b2 = lane_value((at & 63u) * 4u, byte_nxt);
        // (the third alternative's load stands behind a branch, and at its join the compiler waits for everything in flight
        // whenever a lane's literals reach into the next tile; moving it out of the loop -- a flag, a loop of its own behind a
        // ballot -- was measured: lz_parse 3.35 -> 3.58 ms on the benchmark's streams, lz_parse_spec 8.5 -> 9.4 on 1 MiB members)
        if (__builtin_amdgcn_inverse_ballot_w64(lit_m) && k < lits) tsyms[first_rel + k] = at < 64u ? a : at < 128u ? b2 : (uint32_t)s[p + k];
        k++;
      }
      if (visited) tsyms[first_rel + lits] = br ? br : byte_cur;
    }
    ZD_PP(3);
    // block cut: the first visited node that ends past blk_start + 65534
    // (a visited position is at or behind blk_start, so the test runs on 32-bit
    // distances from it; lanes that are not visited are masked out)
    const uint32_t rel = p - blk_start;
    const unsigned long long cb = MODE == 0 ? vis_m & __builtin_amdgcn_ballot_w64(rel + adv > (uint32_t)MAX_BLOCK_SRC_LEN) : 0ull;
    if (MODE == 0 && cb) {
      const int c = __ffsll((long long)cb) - 1;
      uint32_t cutpos, symidx;
      if (br == 0) { cutpos = p; symidx = first; }
      else if (rel + lits > (uint32_t)MAX_BLOCK_SRC_LEN) { const uint32_t i = (uint32_t)MAX_BLOCK_SRC_LEN - rel; cutpos = p + i; symidx = first + i; }
      else { cutpos = p + lits; symidx = first + lits; }
      cutpos = (uint32_t)__builtin_amdgcn_readlane((int)cutpos, c);  // scalar: the block bookkeeping stays in SGPRs
      symidx = (uint32_t)__builtin_amdgcn_readlane((int)symidx, c);
      if (lane == 0) {
        BlockDesc b;
        b.src_start = blk_start; b.src_len = cutpos - blk_start;
        b.sym_start = blk_sym_start; b.n_syms = symidx - blk_sym_start;
        blocks[nblk] = b;
      }
      nblk++;
      blk_start = cutpos;
      blk_sym_start = symidx;
    }
    nsym += total;
    entry = next_entry;
    // tiles the parse jumps over entirely are skipped
    const uint32_t Be = entry & ~63u;
    if (Be > Bn) {  // (into the sets that are "current" and "next" for the call that follows)
      B = Be;
      m_nxt = load_match(Be);
      lit_nxt = load_lit(Be);
      m_nx2 = load_match(Be + PARSE_TILE);
      lit_nx2 = load_lit(Be + PARSE_TILE);
      m_nx3 = load_match(Be + 2u * PARSE_TILE);
      lit_nx3 = load_lit(Be + 2u * PARSE_TILE);
    } else {
      B = Bn;
    }
    ZD_PP(4);
  };
  uint64_t ma = 0, mb = 0, mc = 0, md = 0;
  uint32_t la = 0, lb = 0, lc = 0, ld = 0;
  if (len) {
    ma = load_match(B0);
    la = load_lit(B0);
    mb = load_match(B0 + PARSE_TILE);
    lb = load_lit(B0 + PARSE_TILE);
    mc = load_match(B0 + 2u * PARSE_TILE);
    lc = load_lit(B0 + 2u * PARSE_TILE);
  }
  while (B < lim) {
    tile_step(ma, mb, mc, md, la, lb, lc, ld);
    if (B >= lim) break;
    tile_step(mb, mc, md, ma, lb, lc, ld, la);
    if (B >= lim) break;
    tile_step(mc, md, ma, mb, lc, ld, la, lb);
    if (B >= lim) break;
    tile_step(md, ma, mb, mc, ld, la, lb, lc);
  }
#ifdef ZD_PARSE_PHASES
  if (lane == 0) for (int i = 0; i < 5; i++) atomicAdd(&zd_parse_phases[i], pp[i]);
#endif
  if (MODE == 1) {
    if (lane == 0) { G.seg_exit[seg_slot] = entry; G.seg_total[seg_slot] = nsym; }
    return;
  }
  if (lane == 0) {
    BlockDesc b;  // the final block, always present (zd.ml:1216)
    b.src_start = blk_start; b.src_len = len - blk_start;
    b.sym_start = blk_sym_start; b.n_syms = nsym - blk_sym_start;
    blocks[nblk] = b;
    S.n_blocks[stream] = nblk + 1;
  }
}

__global__ __launch_bounds__(64) void lz_parse_kernel(const uint8_t *__restrict__ src_arena,
                                                      const StreamDesc *__restrict__ descs,
                                                      DeflateScratch S, int good_match) {
  if (S.snap_used[blockIdx.x]) lz_parse_wave<0, true>(src_arena, descs, S, good_match, blockIdx.x, 0, ParseSegs{});
  else lz_parse_wave<0, false>(src_arena, descs, S, good_match, blockIdx.x, 0, ParseSegs{});
}

// ---------------------------------------------------------------------------------
// One stream parsed by many waves.  Between two positions where the reference has no pending match the parse is
// a function of the position alone (above), so a wave can start anywhere -- at a position the true parse may
// never visit -- and once the two paths share ONE position they are the same from there on; they do within a
// few symbols on anything but periodic data.
//   lz_parse_spec_kernel    a wave per segment of 4096 positions: the parse from the segment's first position
//                           (true for segment 0), symbols to the segment's own buffer, per tile the positions
//                           visited and the symbol count before it, per segment where the path left it;
//   lz_parse_meet_kernel    a wave per segment again: from the entry the segment has IF the one before is left where
//                           its own parse left it, tiles are parsed again until the path meets the segment's own
//                           (a common position in a tile); the rest of the segment is then a range of its buffer,
//                           and the segment is left where its own parse left it.  No meeting before the segment
//                           ends: the whole segment was parsed again and is left where that parse says;
//   lz_parse_stitch_kernel  a wave per stream, segment by segment: is the segment entered where the meet kernel
//                           assumed?  Almost always -- then its numbers stand.  Else (behind a segment whose path
//                           never met: a long run of one byte parsed at the wrong phase) it is parsed again here,
//                           from the true entry; at worst that is the serial order.  Blocks are cut here, at the
//                           step that holds source byte 65534 of the block;
//   lz_parse_gather_kernel  a wave per segment copies the symbols parsed again and the segment's range to their
//                           place in the symbol array.
__global__ __launch_bounds__(64) void lz_parse_spec_kernel(const uint8_t *__restrict__ src_arena,
                                                           const StreamDesc *__restrict__ descs,
                                                           DeflateScratch S, int good_match, ParseSegs G) {
  if (S.snap_used[blockIdx.x / G.segs_per_stream])
    lz_parse_wave<1, true>(src_arena, descs, S, good_match, blockIdx.x / G.segs_per_stream, blockIdx.x % G.segs_per_stream, G);
  else
    lz_parse_wave<1, false>(src_arena, descs, S, good_match, blockIdx.x / G.segs_per_stream, blockIdx.x % G.segs_per_stream, G);
}

// What the waves that re-parse parts of a stream share (lz_parse_meet_kernel, lz_parse_stitch_kernel).
struct ParseStream {
  int lane;
  uint32_t len, max_pos;
  const uint8_t *s;
  const uint32_t *match, *snap;
  int good_match;
  const unsigned long long *own_vis;  // per tile: the path of the segment's own parse (read only after lz_parse_spec_kernel),
  const uint32_t *own_sym0;           //   and the segment's symbols before the tile
  unsigned long long *vis2;           // per tile that was parsed again: the true path,
  uint32_t *sym02;                    //   and the symbols parsed again before the tile
};
// A tile's steps, counts and scan: the path from `entry` (use_mask false) or the positions of `mask`.
struct ParseTile {
  uint32_t br, adv, lits, cnt, first_rel, total, next_entry;
  bool visited;
  unsigned long long vm;
};
__device__ __forceinline__ ParseTile parse_eval_tile(const ParseStream &P, uint32_t B, uint32_t entry, bool use_mask,
                                                     unsigned long long mask) {
  ParseTile t;
  const int lane = P.lane;
  const uint32_t p = B + (uint32_t)lane;
  const bool valid = p < P.len;
  const uint64_t m_cur = match_pair(P.match, P.snap, p), m_nxt = match_pair(P.match, P.snap, p + PARSE_TILE);  // (PARSE_PAD zero entries behind the last position)
  parse_tile_macro(lane, p, __builtin_amdgcn_ballot_w64(p < P.len), true, P.max_pos, P.len, m_cur, m_nxt, P.match, P.snap, P.good_match, t.br,
                   t.adv, t.lits);
  t.cnt = valid ? t.lits + 1u : 0u;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  const uint32_t j0 = (uint32_t)lane + t.adv;
  if (use_mask) {
    t.visited = valid && ((mask >> lane) & 1ull);
    t.next_entry = 0;
  } else {
    uint32_t J[6];  // (as in lz_parse_wave's tile_step: six tables reach every node of a tile's path)
    J[0] = j0 < (uint32_t)PARSE_TILE ? j0 * 4u : lane4;
#pragma unroll
    for (int k = 1; k < 6; k++) J[k] = lane_value(J[k - 1], J[k - 1]);
    uint32_t v = (entry - B) * 4u;
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      const uint32_t y = lane_value(v, J[k]);
      if (y <= lane4) v = y;
    }
    t.visited = valid && v == lane4;
    const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    t.next_entry = B + (uint32_t)__builtin_amdgcn_readlane((int)j0, (int)(last >> 2));
    if (t.next_entry > P.len) t.next_entry = P.len;
  }
  t.vm = __builtin_amdgcn_ballot_w64(t.visited);
  const uint32_t incl = wave_scan_incl(t.visited ? t.cnt : 0u);
  t.total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  t.first_rel = incl - (t.visited ? t.cnt : 0u);
  return t;
}

// A segment parsed again from its true entry e (seg_start <= e < seg_end) until the path meets the segment's own --
// a common position in a tile -- or the segment ends.  Symbols to out[0 .. f), the path's marks to vis2 / sym02.
// A path that leaves segment k without having met goes on into k + 1 and so on, up to segment k_limit - 1: a long run
// of equal steps crosses segments 64 steps at a time instead of being cut at every boundary.
struct ParseAgain {
  uint32_t k_end;      // the segment the path met in, or was stopped at the end of
  uint32_t f;          // symbols parsed again
  uint32_t from;       // segment k_end's own symbols from this one on follow them (segments before it: none do)
  uint32_t exit;       // where segment k_end is left
  uint32_t again_end;  // tiles in [tile of e, again_end) were parsed again (their marks are vis2 / sym02)
  bool ok;             // false: out[] was too small (cap) -- nothing of the above holds
};
__device__ __forceinline__ ParseAgain parse_again(const ParseStream &P, uint32_t seg_positions, const uint32_t *seg_total,
                                                  const uint32_t *seg_exit, uint32_t k, uint32_t k_limit, uint32_t e,
                                                  uint32_t *__restrict__ out, uint32_t cap) {
  const int lane = P.lane;
  auto end_of = [&](uint32_t j) -> uint32_t {
    const uint64_t x = (uint64_t)(j + 1) * seg_positions;
    return x < P.len ? (uint32_t)x : P.len;
  };
  const uint32_t ext_end = end_of(k_limit - 1);  // nothing behind this is touched
  uint32_t j = k, seg_end = end_of(k);
  ParseAgain r;
  r.k_end = k; r.f = 0; r.from = 0; r.exit = e; r.again_end = seg_end; r.ok = true;
  uint32_t f = 0;
  uint32_t B = e & ~63u, entry = e;
  uint32_t rest = 0;  // tiles until a run of equal steps is tried again
  // the path is at tile Bx, behind segment j's end: on into the next segment, or the end (true: stop here)
  auto leave = [&](uint32_t Bx) -> bool {
    while (Bx >= seg_end) {
      if (j + 1 >= k_limit || seg_end >= P.len) {
        r.k_end = j; r.again_end = Bx; r.from = seg_total[j]; r.exit = entry;
        return true;
      }
      j++;
      seg_end = end_of(j);
    }
    return false;
  };
  for (;;) {
    if (cap - f < 64u + 576u) { r.ok = false; return r; }  // (a tile's steps make at most a symbol per position they cover)
    const ParseTile t = parse_eval_tile(P, B, entry, false, 0ull);
    const unsigned long long own = P.own_vis[B >> 6];  // the segment's own path through this tile (0: it jumped over it)
    // symbols (lz_emit_position)
    {
      const uint32_t p = B + (uint32_t)lane;
      if (t.visited) {
        for (uint32_t i = 0; i < t.lits; i++) out[f + t.first_rel + i] = P.s[p + i];
        out[f + t.first_rel + t.lits] = t.br ? t.br : (uint32_t)P.s[p];
      }
    }
    if (lane == 0) { P.vis2[B >> 6] = t.vm; P.sym02[B >> 6] = f; }
    f += t.total;
    entry = t.next_entry;
    const uint32_t Be = entry & ~63u;
    const uint32_t Bn = Be > B + PARSE_TILE ? Be : B + PARSE_TILE;
    const bool met = (t.vm & own) != 0ull;
    // tiles this path jumps over are not on it (the block cut looks for the last visited position at or before
    // a given one)
    if (Bn > B + PARSE_TILE) {
      const uint32_t tz = B + PARSE_TILE * (1u + (uint32_t)lane);
      if (tz < Bn && tz < ext_end) P.vis2[tz >> 6] = 0ull;
    }
    if (met) {
      r.k_end = j; r.again_end = Bn;
      r.from = Bn < seg_end ? P.own_sym0[Bn >> 6] : seg_total[j];  // (the segment's own parse went on with tile Bn too)
      r.exit = seg_exit[j];
      break;
    }
    if (leave(Bn)) break;
    // One long step through the tile and no meeting: a run of one byte, or a short period -- the data on which
    // this path and the segment's own stay out of phase to the segment's end.  Then the next steps are probably
    // the same: every lane takes the position i steps on and works out ITS step (lz_macro_position); the lanes
    // before the first one that steps differently are on the path -- up to 64 steps for the price of a tile.
    const uint32_t node_pos = B + (uint32_t)__builtin_ctzll(t.vm | (1ull << 63));
    const uint32_t stride = entry - node_pos;
    if (rest != 0) rest--;
    else if (__builtin_popcountll(t.vm) == 1 && stride >= (uint32_t)PARSE_TILE) {
      const uint64_t ahead = (uint64_t)lane * stride;
      const bool in = ahead < (uint64_t)(ext_end - entry);
      const uint32_t p = in ? entry + (uint32_t)ahead : entry;
      const uint32_t *match = P.match, *snap = P.snap;
      const MacroStep m = lz_macro_position(p, P.len, P.good_match, [&](uint32_t j) -> uint64_t { return match_pair(match, snap, j); });
      const uint32_t adv = m.bref ? macro_advance(m.step) : 1u;
      const unsigned long long go = __builtin_amdgcn_ballot_w64(in && adv == stride);
      const uint32_t lead = ~go ? (uint32_t)__builtin_ctzll(~go) : 64u;  // lanes [0, lead) are on the path, and step alike
      if (lead == 0) rest = 8;
      else {
        const bool on = (uint32_t)lane < lead;
        const uint32_t cnt = on ? macro_sym_count(m) : 0u;
        const uint32_t incl = wave_scan_incl(cnt);
        const uint32_t first_rel = incl - cnt;
        const uint32_t made = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (cap - f < made) { r.ok = false; return r; }
        if (on) {
          lz_emit_position(P.s, p, m, out + f, first_rel);
          // the path's marks for the block cut: this step's tile holds it alone, the tiles it jumps over nothing
          P.vis2[p >> 6] = 1ull << (p & 63u);
          P.sym02[p >> 6] = f + first_rel;
          for (uint32_t tz = (p & ~63u) + PARSE_TILE; tz < ((p + stride) & ~63u) && tz < ext_end; tz += PARSE_TILE) P.vis2[tz >> 6] = 0ull;
        }
        f += made;
        entry += lead * stride;  // (< ext_end + stride: lane lead - 1 was inside)
        if (entry > P.len) entry = P.len;
        const uint32_t Bs = entry & ~63u;
        if (leave(Bs)) break;
        B = Bs;
        continue;
      }
    }
    B = Bn;
  }
  r.f = f;
  return r;
}

// lz_parse_meet_kernel: a wave per segment k >= 1 does that for the entry the segment has if the one before is left
// where its OWN parse left it -- true of all segments but the few whose path never meets -- into the segment's
// own small buffer (MEET_CAP symbols; more: left to the stitch).  The stitch then only checks the chain of exits.
constexpr uint32_t MEET_CAP = 2048;
__global__ __launch_bounds__(64) void lz_parse_meet_kernel(const uint8_t *__restrict__ src_arena,
                                                           const StreamDesc *__restrict__ descs,
                                                           DeflateScratch S, int good_match, ParseSegs G) {
  if (S.error[0]) return;
  const uint32_t stream = blockIdx.x / G.segs_per_stream, k = blockIdx.x % G.segs_per_stream;
  const StreamDesc sd = descs[stream];
  if (sd.src_len > MAX_STREAM_LEN || sd.src_len < (uint64_t)MIN_MATCH_LEN) return;
  const uint32_t len = (uint32_t)sd.src_len;
  if (k == 0 || (uint64_t)k * G.seg_positions >= len) return;
  const uint64_t base = S.pos_base[stream];
  const size_t slot = G.slot(S, stream, k);
  ParseStream P;
  P.lane = threadIdx.x; P.len = len; P.max_pos = len - MIN_MATCH_LEN; P.s = src_arena + sd.src_off;
  P.match = S.match + base; P.snap = S.snap + base; P.good_match = good_match;
  P.own_vis = G.vis + (base >> 6); P.own_sym0 = G.tile_sym0 + (base >> 6);
  P.vis2 = G.vis2 + (base >> 6); P.sym02 = G.sym02 + (base >> 6);
  const uint32_t seg_start = k * G.seg_positions;
  const uint32_t seg_end = len - seg_start > G.seg_positions ? seg_start + G.seg_positions : len;
  const uint32_t e = G.seg_exit[slot - 1];
  const uint32_t spec_total = G.seg_total[slot];
  ParseAgain a;
  if (e >= seg_end) {  // (a step over a short last segment)
    a.k_end = k; a.f = 0; a.from = spec_total; a.exit = e; a.again_end = seg_end; a.ok = true;
  } else {
    a = parse_again(P, G.seg_positions, G.seg_total + (slot - k), G.seg_exit + (slot - k), k, k + 1, e,
                    G.meet_syms + slot * MEET_CAP, MEET_CAP);
  }
  if (P.lane == 0) {
    G.meet_f[slot] = a.ok ? a.f : 0xFFFFFFFFu;
    G.meet_from[slot] = a.from;
    G.meet_exit[slot] = a.exit;
    G.meet_end[slot] = a.again_end;
  }
}

__global__ __launch_bounds__(64) void lz_parse_stitch_kernel(const uint8_t *__restrict__ src_arena,
                                                             const StreamDesc *__restrict__ descs,
                                                             DeflateScratch S, int good_match, ParseSegs G) {
  if (S.error[0]) return;
  const uint32_t stream = blockIdx.x;
  const int lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (sd.src_len > MAX_STREAM_LEN) {
    if (lane == 0) S.n_blocks[stream] = 0;
    return;
  }
  const uint32_t len = (uint32_t)sd.src_len;
  const uint8_t *s = src_arena + sd.src_off;
  const uint64_t base = S.pos_base[stream];
  uint32_t *syms = S.syms + base;
  BlockDesc *blocks = S.blocks + S.blk_base[stream];
  if (len < (uint32_t)MIN_MATCH_LEN) {  // 0 to 3 bytes: literals, one block
    if ((uint32_t)lane < len) syms[lane] = s[lane];
    if (lane == 0) {
      BlockDesc b;
      b.src_start = 0; b.src_len = len; b.sym_start = 0; b.n_syms = len;
      blocks[0] = b;
      S.n_blocks[stream] = 1;
    }
    return;
  }
  ParseStream P;
  P.lane = lane; P.len = len; P.max_pos = len - MIN_MATCH_LEN; P.s = s;
  P.match = S.match + base; P.snap = S.snap + base; P.good_match = good_match;
  P.own_vis = G.vis + (base >> 6); P.own_sym0 = G.tile_sym0 + (base >> 6);
  P.vis2 = G.vis2 + (base >> 6); P.sym02 = G.sym02 + (base >> 6);
  const size_t slot0 = G.slot(S, stream, 0);
  const uint32_t PARSE_SEG = G.seg_positions;
  const uint32_t nseg = (uint32_t)(((uint64_t)len + PARSE_SEG - 1) / PARSE_SEG);

  uint32_t off = 0;        // symbols of the stream so far
  uint32_t exit_prev = 0;  // where the segment before was left: this segment's true entry
  uint32_t spec_exit_prev = 0;  // ... and where its own parse left it: the entry lz_parse_meet_kernel went by
  uint32_t blk_start = 0, blk_sym_start = 0, nblk = 0;
  for (uint32_t c0 = 0; c0 < nseg; c0 += 64) {
    // 64 segments a turn: the lanes fetch their segment's numbers together, the turn's segments are gone through
    // in order on wave-uniform values
    const uint32_t mine = c0 + (uint32_t)lane < nseg ? c0 + (uint32_t)lane : nseg - 1;
    const uint32_t my_total = G.seg_total[slot0 + mine], my_exit = G.seg_exit[slot0 + mine];
    const uint32_t my_mf = mine ? G.meet_f[slot0 + mine] : 0u, my_mfrom = mine ? G.meet_from[slot0 + mine] : 0u;
    const uint32_t my_mexit = mine ? G.meet_exit[slot0 + mine] : 0u, my_mend = mine ? G.meet_end[slot0 + mine] : 0u;
    uint32_t out_dst = 0, out_from = 0, out_n = 0, out_fix_dst = 0, out_fix_n = 0;  // my segment's verdict, stored after the turn
    const uint32_t turn = nseg - c0 < 64u ? nseg - c0 : 64u;
    for (uint32_t i = 0; i < turn; i++) {
      uint32_t k = c0 + i;
      auto of_lane = [&](uint32_t v, uint32_t l) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
      uint32_t spec_total = of_lane(my_total, i), spec_exit = of_lane(my_exit, i);
      const uint32_t e = exit_prev;
      ParseAgain a;
      a.k_end = k; a.f = 0; a.from = 0; a.exit = spec_exit; a.ok = true;  // segment 0: its own parse is the true one
      a.again_end = k * PARSE_SEG;
      bool in_buffer = false;  // the symbols parsed again are in the segment's meet buffer (else at syms[off ..))
      if (k != 0) {
        const uint32_t mf = of_lane(my_mf, i);
        const uint32_t seg_end_k = len - k * PARSE_SEG > PARSE_SEG ? (k + 1) * PARSE_SEG : len;
        if (e == spec_exit_prev && mf != 0xFFFFFFFFu) {  // as lz_parse_meet_kernel assumed
          a.f = mf; a.from = of_lane(my_mfrom, i); a.exit = of_lane(my_mexit, i); a.again_end = of_lane(my_mend, i);
          in_buffer = true;
        } else if (e >= seg_end_k) {  // (a step over a short last segment)
          a.from = spec_total; a.exit = e; a.again_end = seg_end_k;
        } else {  // from the true entry, through the turn's segments if need be
          a = parse_again(P, PARSE_SEG, G.seg_total + slot0, G.seg_exit + slot0, k, c0 + turn, e, syms + off, 0xFFFFFFFFu);
          if (a.k_end != k) {  // the segments the path went through without meeting: nothing of their own is used
            const uint32_t i2 = a.k_end - c0;
            if ((uint32_t)lane >= i && (uint32_t)lane < i2) { out_dst = off; out_from = my_total; out_n = 0; out_fix_dst = off; out_fix_n = 0; }
            i = i2;
            k = a.k_end;
            spec_total = of_lane(my_total, i);
            spec_exit = of_lane(my_exit, i);
          }
        }
      }
      const uint32_t seg_end = len - k * PARSE_SEG > PARSE_SEG ? (k + 1) * PARSE_SEG : len;
      const uint32_t f = a.f, from = a.from, exit_k = a.exit, again_end = a.again_end;
      const uint32_t n_own = spec_total - from;
      if ((uint32_t)lane == i) {
        out_dst = off + f; out_from = from; out_n = n_own;
        out_fix_dst = off; out_fix_n = in_buffer ? f : 0u;
      }
      // block cut (write_block_symbol zd.ml:1118-1123): the step that holds source byte blk_start + 65534 --
      // the first one that ends behind it -- closes the block
      while (exit_k - blk_start > (uint32_t)MAX_BLOCK_SRC_LEN) {
        const uint32_t T = blk_start + (uint32_t)MAX_BLOCK_SRC_LEN;
        // the last visited position <= T; it lies in this segment (the step that leaves this segment started
        // inside it).  A tile's marks: the true path's where the tile was parsed again, else the segment's own.
        auto again_tile = [&](uint32_t Bt) -> bool { return k != 0 && Bt >= (e & ~63u) && Bt < again_end; };
        auto marks = [&](uint32_t Bt) -> unsigned long long { return again_tile(Bt) ? P.vis2[Bt >> 6] : P.own_vis[Bt >> 6]; };
        const uint32_t Ts = T < seg_end ? T : seg_end - 1u;
        uint32_t Bt = Ts & ~63u;
        unsigned long long w = marks(Bt) & (~0ull >> (63u - (Ts & 63u)));
        while (w == 0ull) { Bt -= PARSE_TILE; w = marks(Bt); }  // (the step started in an earlier tile: at most 9 back)
        const int c = 63 - __builtin_clzll(w);
        const ParseTile t = parse_eval_tile(P, Bt, 0, true, marks(Bt));
        const uint32_t tile0 = again_tile(Bt) ? off + P.sym02[Bt >> 6] : off + f - from + P.own_sym0[Bt >> 6];
        const uint32_t p = Bt + (uint32_t)lane;
        const uint32_t rel = p - blk_start, first = tile0 + t.first_rel;
        uint32_t cutpos, symidx;
        if (t.br == 0) { cutpos = p; symidx = first; }
        else if (rel + t.lits > (uint32_t)MAX_BLOCK_SRC_LEN) { const uint32_t j = (uint32_t)MAX_BLOCK_SRC_LEN - rel; cutpos = p + j; symidx = first + j; }
        else { cutpos = p + t.lits; symidx = first + t.lits; }
        cutpos = (uint32_t)__builtin_amdgcn_readlane((int)cutpos, c);
        symidx = (uint32_t)__builtin_amdgcn_readlane((int)symidx, c);
        if (lane == 0) {
          BlockDesc b;
          b.src_start = blk_start; b.src_len = cutpos - blk_start;
          b.sym_start = blk_sym_start; b.n_syms = symidx - blk_sym_start;
          blocks[nblk] = b;
        }
        nblk++;
        blk_start = cutpos;
        blk_sym_start = symidx;
      }
      off += f + n_own;
      exit_prev = exit_k;
      spec_exit_prev = spec_exit;
    }
    if (c0 + (uint32_t)lane < nseg) {
      const size_t slot = slot0 + c0 + (uint32_t)lane;
      G.seg_dst[slot] = out_dst; G.seg_from[slot] = out_from; G.seg_n[slot] = out_n;
      G.fix_dst[slot] = out_fix_dst; G.fix_n[slot] = out_fix_n;
    }
  }
  if (lane == 0) {
    BlockDesc b;  // the final block, always present (zd.ml:1216)
    b.src_start = blk_start; b.src_len = len - blk_start;
    b.sym_start = blk_sym_start; b.n_syms = off - blk_sym_start;
    blocks[nblk] = b;
    S.n_blocks[stream] = nblk + 1;
  }
}

__global__ __launch_bounds__(64) void lz_parse_gather_kernel(const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                             ParseSegs G) {
  if (S.error[0]) return;
  const uint32_t stream = blockIdx.x / G.segs_per_stream, seg = blockIdx.x % G.segs_per_stream;
  const uint64_t len = descs[stream].src_len;
  if (len > MAX_STREAM_LEN || len < (uint64_t)MIN_MATCH_LEN || (uint64_t)seg * G.seg_positions >= len) return;
  const size_t slot = G.slot(S, stream, seg);
  uint32_t *syms = S.syms + S.pos_base[stream];
  {  // the symbols lz_parse_meet_kernel parsed again, if the stitch took them
    const uint32_t n = G.fix_n[slot];
    const uint32_t *from = G.meet_syms + slot * MEET_CAP;
    uint32_t *to = syms + G.fix_dst[slot];
    for (uint32_t i = threadIdx.x; i < n; i += 64u) to[i] = from[i];
  }
  const uint32_t n = G.seg_n[slot];
  const uint32_t *from = G.spec_syms + slot * G.seg_syms + G.seg_from[slot];
  uint32_t *to = syms + G.seg_dst[slot];
  for (uint32_t i = threadIdx.x; i < n; i += 64u) to[i] = from[i];
}

// ---------------------------------------------------------------------------------
// Bit packing by the whole wave.  Items (block header, dynamic-header fields,
// symbols) are taken 64 at a time: each lane turns one item into (value, nbits),
// an inclusive wave scan gives its bit offset, lanes OR their bits into the LDS
// staging row, and the completed bytes are flushed with coalesced stores.
// (measured, same box, emit on C2 / text with 1 / 2 / 3 / 4 / 6 / 8 tiles: 3.05 / 2.61 / 2.41 / 2.38 / 2.31 / 2.56 and 1.70 / 1.57 / 1.55 / 1.51 /
// 1.49 / 1.70 ms: fewer flushes per item, until the staging row costs the wave its place on the CU)
constexpr int PACK_TILES = 6;  // tiles of 64 items packed between two flushes of the staging row
constexpr int STAGE_WORDS = (7 + PACK_TILES * 64 * 48 + 31) / 32 + 7;  // 7 carried bits + 48 bits per item

// The emit kernel runs one wave per workgroup: LDS operations of one wave execute
// in order, so lanes only need the COMPILER to keep LDS accesses in program order
// around an exchange -- not __syncthreads(), whose s_waitcnt vmcnt(0) would also
// wait for every global store of the previous tile.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct BitOut {
  uint8_t *dst;        // stream output base
  uint32_t out_pos;    // bytes already in global memory
  uint32_t acc;        // pending bits (< 8), dst_bits of the reference (zd.ml:784)
  int acc_bits;        // dst_bits_len (zd.ml:785)
};

__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// pack PACK_TILES tiles: lane holds (value, nbits) of one item per tile (nbits = 0
// for idle lanes); tile u's items follow tile u-1's in the bit stream
__device__ __forceinline__ void pack_tiles(BitOut &bo, uint32_t *stage, const uint64_t *value, const int *nbits,
                                           int lane) {
  // stage[] is zero except stage[0] = pending bits
  uint32_t at = (uint32_t)bo.acc_bits;  // bit offset of the tile in the staging row
#pragma unroll
  for (int u = 0; u < PACK_TILES; u++) {
    const uint32_t incl = wave_scan_incl((uint32_t)nbits[u]);
    if (nbits[u]) {
      const uint32_t o = at + incl - (uint32_t)nbits[u];
      const uint32_t wi = o >> 5, sh = o & 31;
      const uint64_t lo = value[u] << sh;
      atomicOr(&stage[wi], (uint32_t)lo);
      const uint32_t mid = (uint32_t)(lo >> 32);
      if (mid) atomicOr(&stage[wi + 1], mid);
      if (sh && nbits[u] + (int)sh > 64) atomicOr(&stage[wi + 2], (uint32_t)(value[u] >> (64 - sh)));
    }
    at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  const uint32_t total = at;
  wave_sync();
  const uint32_t full_bytes = total >> 3;
  const uint32_t full_words = full_bytes >> 2;
  uint8_t *o = bo.dst + bo.out_pos;
  for (uint32_t w = (uint32_t)lane; w < full_words; w += 64) store_u32_unaligned(o + 4 * w, stage[w]);
  const uint32_t tail = full_bytes & 3u;
  if ((uint32_t)lane < tail) o[4 * full_words + lane] = (uint8_t)(stage[full_words] >> (8 * lane));
  const uint32_t rem_bits = total & 7u;
  const uint32_t last = (stage[full_bytes >> 2] >> (8 * (full_bytes & 3u))) & ((1u << rem_bits) - 1u);
  wave_sync();
  // reset the part of the staging row that was used
  for (uint32_t w = (uint32_t)lane; w <= (total >> 5) + 1 && w < (uint32_t)STAGE_WORDS; w += 64) stage[w] = 0;
  wave_sync();
  if (lane == 0) stage[0] = last;
  bo.out_pos += full_bytes;
  bo.acc = last;
  bo.acc_bits = (int)rem_bits;
  wave_sync();
}

// ---- the block coder by the whole wave (serial forms: deflate_lane.h, which the
// host model runs).  What is order sensitive in the reference -- the heap that
// builds the tree, with its tie behaviour, and the run-length scan of the code
// lengths -- stays a serial algorithm that every lane runs redundantly (wave-
// uniform, hence on the scalar unit); the loops over symbols around them are
// spread over the lanes.
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Huffman.lengths_of_freqs zd.ml:404-473 on two queues (huff_tree_two_queues, deflate_lane.h, has
// why that is the reference's tree).  The leaves are keyed and sorted by all lanes (rank by counting:
// the keys are distinct), ONE lane merges -- n - 1 steps of a few LDS reads each, where the heap
// took 2 (n - 1) sift-downs of log n dependent levels -- and all lanes climb from their leaves.
// heap: 2 n + 2 words of LDS (n <= max_sym + 1): n queue words, then the parents as u16.
__device__ __forceinline__ void wave_lengths_of_freqs(uint32_t *heap, uint32_t *e, const uint32_t *freqs, int max_sym,
                                                      int max_code_len, int lane) {
  uint32_t freq_cap = 65535;
  for (;;) {
    // keys of the used symbols in symbol order, at tmp[rank - 1]
    int n = 0;
#pragma unroll 1
    for (int c = 0; c <= max_sym; c += 64) {
      const int sym = c + lane;
      const uint32_t f = sym <= max_sym ? freqs[sym] : 0u;
      n += __popcll(__builtin_amdgcn_ballot_w64(f != 0));
    }
    if (n < 2) {  // trivial_codeword_lengths zd.ml:462-466
      for (int sym = lane; sym <= max_sym; sym += 64) e[sym] = freqs[sym] == 0 ? 0u : 1u;
      return;
    }
    uint32_t *q = heap, *tmp = heap + n;          // tmp: where the parents go later
    uint16_t *par = (uint16_t *)(heap + n);
    int rank0 = 0;
#pragma unroll 1
    for (int c = 0; c <= max_sym; c += 64) {
      const int sym = c + lane;
      uint32_t f = sym <= max_sym ? freqs[sym] : 0u;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(f != 0);
      if (f != 0) {
        if (f > freq_cap) f = freq_cap;
        const int rank = rank0 + 1 + (int)lanes_below(m);
        tmp[rank - 1] = (f << 10) | (uint32_t)rank;
      }
      rank0 += __popcll(m);
    }
    wave_sync();
    // sorted place of a key = number of smaller keys; a lane's keys (up to 5) take one pass together
    {
      constexpr int MAXC = (LITLEN_SYM_MAX + 64) / 64;
      uint32_t key[MAXC], below[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; c++) {
        const int i = c * 64 + lane;
        key[c] = i < n ? tmp[i] : 0u;  // 0: below every real key, counts nothing
        below[c] = 0;
      }
      const int chunks = (n + 63) / 64;  // wave-uniform
#pragma unroll 2
      for (int j = 0; j < n; j++) {
        const uint32_t kj = tmp[j];
#pragma unroll
        for (int c = 0; c < MAXC; c++)
          if (c < chunks) below[c] += kj < key[c] ? 1u : 0u;
      }
      wave_sync();  // (q does not overlap tmp)
#pragma unroll
      for (int c = 0; c < MAXC; c++)
        if (c * 64 + lane < n) q[below[c]] = key[c];
    }
    wave_sync();
    if (lane == 0) huff_tree_two_queues(q, n, par);
    wave_sync();
    bool overflow = false;  // code_lengths_of_tree zd.ml:446-461: every leaf climbs to the root
    int rank = 0;
#pragma unroll 1
    for (int c = 0; c <= max_sym; c += 64) {
      const int sym = c + lane;
      const bool nz = sym <= max_sym && freqs[sym] != 0;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(nz);
      uint32_t l = 0;
      if (nz) {
        uint32_t p = par[n + rank + 1 + (int)lanes_below(m)];
        l = 1;
        while (p != 2) { l++; p = par[p]; }
      }
      if (sym <= max_sym) e[sym] = l;
      overflow |= l > (uint32_t)max_code_len;
      rank += __popcll(m);
    }
    if (__builtin_amdgcn_ballot_w64(overflow) == 0) return;
    freq_cap >>= 1;  // flatten and retry zd.ml:470-473
    wave_sync();
  }
}

// Huffman.init_with_lengths zd.ml:477-506 (huff_init_with_lengths); scratch: 32 words
__device__ __forceinline__ void wave_init_with_lengths(uint32_t *e, int max_sym, uint32_t *scratch, int lane) {
  uint32_t *count = scratch, *next = scratch + 16;
  if (lane < 16) count[lane] = 0;
  wave_sync();
  for (int sym = lane; sym <= max_sym; sym += 64) atomicAdd(&count[e[sym] & 0x1F], 1u);
  wave_sync();
  uint32_t code = 0;
#pragma unroll 1
  for (int l = 1; l <= 15; l++) {  // every lane, same values
    code = (code + (l == 1 ? 0u : count[l - 1])) << 1;
    if (lane == 0) next[l] = code;
  }
  wave_sync();
#pragma unroll 1
  for (int c = 0; c <= max_sym; c += 64) {
    const int sym = c + lane;
    const uint32_t l = sym <= max_sym ? e[sym] & 0x1F : 0u;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(l != 0);
#pragma unroll 1
    while (todo) {
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)l, __ffsll((long long)todo) - 1);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(l == cur);
      const uint32_t first = next[cur];
      if (l == cur) e[sym] = (bitrev(first + lanes_below(m), (int)cur) << 5) | cur;
      wave_sync();
      if (lane == 0) next[cur] = first + (uint32_t)__popcll(m);
      wave_sync();
      todo &= ~m;
    }
  }
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
  return v;
}

// bit_length_of_block_symbols zd.ml:1049-1064 (coder_symbols_bits)
__device__ __forceinline__ uint64_t wave_symbols_bits(const BlockCoder &c, const uint32_t *hlit, const uint32_t *hdist,
                                                      int lane) {
  uint64_t acc = 0;
  for (int sym = lane; sym <= LITLEN_SYM_MAX; sym += 64)
    acc += (uint64_t)c.lit_freq[sym] * ((hlit[sym] & 0x1F) + (uint32_t)length_extra_bits(sym));
  if (lane <= DIST_SYM_MAX) acc += (uint64_t)c.dist_freq[lane] * ((hdist[lane] & 0x1F) + (uint32_t)dist_extra_bits(lane));
  return wave_sum64(acc);
}

// make_dynamic_huffman + make_dynamic_huffman_encoding zd.ml:953-1043 (coder_make_dynamic), in two halves:
// the codes of the block's own symbols and their code-length symbols (counted into c.codelen_freq) ...
__device__ __forceinline__ void wave_make_dynamic_syms(BlockCoder &c, uint32_t *scratch, int lane) {
  wave_lengths_of_freqs(c.heap, c.dyn_lit, c.lit_freq, LITLEN_SYM_MAX, 15, lane);
  wave_sync();
  wave_init_with_lengths(c.dyn_lit, LITLEN_SYM_MAX, scratch, lane);
  wave_lengths_of_freqs(c.heap, c.dyn_dist, c.dist_freq, DIST_SYM_MAX, 15, lane);
  wave_sync();
  wave_init_with_lengths(c.dyn_dist, DIST_SYM_MAX, scratch, lane);
  wave_sync();
  // gather_dynamic_huffman_code_lengths zd.ml:963-988
  int litlen_count = 0;
#pragma unroll 1
  for (int cbase = 0; cbase <= LITLEN_SYM_MAX; cbase += 64) {
    const int sym = cbase + lane;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(sym <= LITLEN_SYM_MAX && (c.dyn_lit[sym] & 0x1F) != 0);
    if (m) litlen_count = cbase + 64 - __clzll((long long)m);
  }
  const unsigned long long dm = __builtin_amdgcn_ballot_w64(lane <= DIST_SYM_MAX && (c.dyn_dist[lane] & 0x1F) != 0);
  int dist_count = dm ? 64 - __clzll((long long)dm) : 0;
  if (dist_count == 0) {  // HDIST 0 means 1: symbol 0 gets length 1, code 0 (zd.ml:974-979)
    if (lane == 0) c.dyn_dist[0] = 1;
    dist_count = 1;
    wave_sync();
  }
  c.hlit = litlen_count - 257;
  c.hdist = dist_count - 1;
  uint32_t *l = c.codelen_syms;
  for (int i = lane; i < litlen_count; i += 64) l[i] = c.dyn_lit[i] & 0x1F;
  if (lane < dist_count) l[litlen_count + lane] = c.dyn_dist[lane] & 0x1F;
  wave_sync();
  // compute_codelen_syms zd.ml:989-1030 (in place: the encoding never expands)
  // (one lane: the scan is serial, and 64 lanes bumping the same counter were 64 stores to one address)
  const int len_max = litlen_count + dist_count - 1;
  // where the lengths are not zero, as one bit each (a ballot per 64): the end of a run of zeros -- up to 138 of
  // them, and most of a sparse alphabet's array -- is then the next set bit, not a read per entry
  constexpr int LC = 5;
  static_assert(LITLEN_SYM_MAX + 1 + DIST_SYM_MAX + 1 <= LC * 64, "the lengths fit the masks");
  unsigned long long NZ[LC];
#pragma unroll
  for (int cc = 0; cc < LC; cc++) {
    const int idx = cc * 64 + lane;
    NZ[cc] = __builtin_amdgcn_ballot_w64(idx > len_max || l[idx] != 0);  // (behind the array: a run ends there)
  }
  auto zeros_end = [&](int from) -> int {  // the first entry at or behind `from` that is not zero
    int r = LC * 64;
#pragma unroll
    for (int cc = LC - 1; cc >= 0; cc--) {
      const int c0 = from >> 6, b0 = from & 63;
      const unsigned long long m = cc == c0 ? (NZ[cc] >> b0) << b0 : cc > c0 ? NZ[cc] : 0ull;
      if (m) r = cc * 64 + __builtin_ctzll(m);
    }
    return r;
  };
  int k = 0, i = 0;
  if (lane == 0) {
#pragma unroll 1
  while (i <= len_max) {
    if (l[i] == 0) {
      const int ze = zeros_end(i + 1);
      const int j = ze < i + 138 ? ze : i + 138;
      const int zcount = j - i;
      if (zcount < 3) { l[k] = 0; c.codelen_freq[0]++; i = i + 1; }
      else if (zcount <= 10) { l[k] = ((uint32_t)(zcount - 3) << 8) | 17; c.codelen_freq[17]++; i = j; }
      else { l[k] = ((uint32_t)(zcount - 11) << 8) | 18; c.codelen_freq[18]++; i = j; }
      k++;
    } else {
      const uint32_t sym = l[i];
      const int mx = len_max < i + 6 ? len_max : i + 6;
      int j = i + 1;
      while (j <= mx && l[j] == sym) j++;
      const int scount = j - i;
      l[k] = sym;
      c.codelen_freq[sym]++;
      if (scount <= 3) { k++; i = i + 1; }
      else {
        l[k + 1] = ((uint32_t)(scount - 3 - 1) << 8) | 16;
        c.codelen_freq[16]++;
        k += 2;
        i = j;
      }
    }
  }
  }
  c.codelen_syms_len = __builtin_amdgcn_readfirstlane(k);
  wave_sync();
}
// ... and the code of the code-length symbols, from counts that are never reset between the blocks of a
// stream (Q1, zd.ml:849-854): it depends on every block before
__device__ __forceinline__ void wave_make_dynamic_codelen(BlockCoder &c, uint32_t *scratch, int lane) {
  wave_lengths_of_freqs(c.heap, c.dyn_codelen, c.codelen_freq, CODELEN_SYM_MAX, 7, lane);
  wave_sync();
  wave_init_with_lengths(c.dyn_codelen, CODELEN_SYM_MAX, scratch, lane);
  wave_sync();
  int o = CODELEN_SYM_MAX;  // codelen_length_count zd.ml:1032-1036
  while (o > 0 && (c.dyn_codelen[k_codelen_order[o]] & 0x1F) == 0) o--;
  c.hclen = (o + 1) - 4;
}

__device__ __forceinline__ void wave_make_dynamic(BlockCoder &c, uint32_t *scratch, int lane) {
  wave_make_dynamic_syms(c, scratch, lane);
  wave_make_dynamic_codelen(c, scratch, lane);
}

// write_block's three estimates and its choice (coder_choose)
__device__ __forceinline__ int wave_choose(const BlockCoder &c, uint32_t block_src_len, int pending_bits, uint64_t &flen,
                                           uint64_t &dlen, int lane) {
  const uint64_t nlen = 3 + (uint64_t)(8 - ((pending_bits + 3) % 8)) + (4 + (uint64_t)block_src_len) * 8;
  flen = 3 + wave_symbols_bits(c, c.fix_lit, c.fix_dist, lane);
  uint64_t acc = 0;
  if (lane <= CODELEN_SYM_MAX) {
    const uint32_t rb = lane == 16 ? 2 : lane == 17 ? 3 : lane == 18 ? 7 : 0;
    acc = (uint64_t)c.codelen_freq[lane] * ((c.dyn_codelen[lane] & 0x1F) + rb);
  }
  dlen = 3 + 5 + 5 + 4 + 3 * (uint64_t)(c.hclen + 4) + wave_sum64(acc) + wave_symbols_bits(c, c.dyn_lit, c.dyn_dist, lane);
  if (nlen <= dlen && nlen <= flen) return 0;
  if (flen <= dlen) return 1;
  return 2;
}

// The blocks of a stream coded by a wave each (deflate_plan_kernel / deflate_scan_kernel / deflate_pack_kernel /
// deflate_seal_kernel below): what they leave for each other, one record per block at the block's slot.
constexpr uint32_t EMIT_SKIP = 0xFFu;  // kind of every block of a stream whose output does not fit
constexpr uint32_t EMIT_PART = 8192;   // symbols a pack wave takes
constexpr uint32_t EMIT_PARTS = 8;     // parts of a block at most (65534 symbols: all literals)
static_assert(EMIT_PART * EMIT_PARTS >= (uint32_t)MAX_BLOCK_SRC_LEN, "a block's symbols fit its parts");
// (split: a wave per part -- worth its per-wave set-up only while a call has few blocks; else a wave per block)
__host__ __device__ inline uint32_t emit_parts_of(bool split, uint32_t kind, uint32_t n_syms) {
  return !split || kind == 0 || n_syms == 0 ? 1u : (n_syms + EMIT_PART - 1) / EMIT_PART;
}
struct EmitPlan {
  uint64_t flen;          // plan: bits of the block as a fixed block,
  uint64_t dyn_sym_bits;  //   and of its symbols under its own dynamic code
  uint64_t dlen, dbits;   // codelen: the reference's estimate of the block as a dynamic block, and its bits
  uint64_t bit_start, bit_end;  // scan: the block's bits in the stream's output
  uint32_t kind;          // scan: 0 stored, 1 fixed, 2 dynamic, EMIT_SKIP
  uint32_t part_tail[EMIT_PARTS];  // pack: the bits behind a part's last whole byte
  uint64_t part_bits[EMIT_PARTS];  // bits: of the part's symbols under the block's code (a coded block is packed by a wave
                                   //   per EMIT_PART symbols: part 0 with the header in front, the last one with the end-of-block symbol)
  int32_t codelen_syms_len, hlit, hdist;  // plan
  int32_t hclen;          // codelen
  uint32_t n_chunks;      // plan (Adler-32): the block's chunks, first = len mod 5552 (possibly empty), then 5552 each
  uint32_t codelen_freq[19];  // plan: the block's own code-length symbols, counted
  uint32_t cum_freq[19];      // counts: ... and with those of the blocks before
  uint2 adler[13];        // plan: (S1, S2) per chunk
  uint32_t dyn_lit[288], dyn_dist[32], codelen_syms[320];  // plan
  uint32_t dyn_codelen[32];                                // codelen
};

// MODE 0: a wave codes a whole stream, block after block.  MODE 1 (plan): a wave takes ONE block as far as its
// bits do not depend on the blocks before it -- histogram, the codes of its symbols, its code-length symbols, the
// fixed and dynamic sizes, its Adler-32 chunk sums.  MODE 2 (pack): a wave writes one PART of a block where the scan
// and the parts before it put it.  MODE 3 (bits): a wave adds up the bits of a part's symbols under the block's code.
template <int MODE>
__device__ __forceinline__ void deflate_emit_wave(const uint8_t *__restrict__ src_arena,
                                                  uint8_t *__restrict__ dst_arena,
                                                  const StreamDesc *__restrict__ descs,
                                                  StreamResult *__restrict__ results,
                                                  DeflateScratch S, int crc_op, uint32_t stream, uint32_t only_block,
                                                  EmitPlan *__restrict__ plans, uint32_t part = 0, bool split = false) {
  __shared__ uint32_t lit_freq[288], dist_freq[32], codelen_freq[32];
  __shared__ uint32_t dyn_lit[288], dyn_dist[32], dyn_codelen[32], fix_lit[288], fix_dist[32];
  __shared__ uint32_t codelen_syms[320], heap[580];
  __shared__ uint32_t stage[STAGE_WORDS];
  __shared__ uint32_t coder_scratch[32];

  const int lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (S.error[0] || sd.src_len > MAX_STREAM_LEN || sd.dst_cap > MAX_STREAM_LEN) {
    if (MODE == 0 && lane == 0) { StreamResult r; r.status = S.error[0] == 2u ? ST_HIP : ST_INVALID_ARG; r.checksum = 0; r.out_len = 0; results[stream] = r; }
    return;
  }
  const uint8_t *src = src_arena + sd.src_off;
  const uint32_t dst_cap = (uint32_t)sd.dst_cap;
  const uint32_t nblk = S.n_blocks[stream];
  if (MODE != 0 && only_block >= nblk) return;
  const BlockDesc *blocks = S.blocks + S.blk_base[stream];
  const uint32_t *syms = S.syms + S.pos_base[stream];
  EmitPlan *plan = MODE != 0 ? plans + S.blk_base[stream] + only_block : nullptr;
  if (MODE >= 2) {
    const uint32_t k = plan->kind;
    if (k == EMIT_SKIP || part >= emit_parts_of(split, k, blocks[only_block].n_syms)) return;
    if (MODE == 3 && k == 0) return;
  }

  BlockCoder c;
  c.lit_freq = lit_freq; c.dist_freq = dist_freq; c.codelen_freq = codelen_freq;
  c.dyn_lit = dyn_lit; c.dyn_dist = dyn_dist; c.dyn_codelen = dyn_codelen;
  c.fix_lit = fix_lit; c.fix_dist = fix_dist; c.codelen_syms = codelen_syms; c.heap = heap;
  c.codelen_syms_len = 0; c.hlit = 0; c.hdist = 0; c.hclen = 0;

  for (int i = lane; i < 32; i += 64) { codelen_freq[i] = 0; dyn_dist[i] = 0; dyn_codelen[i] = 0; }
  for (int i = lane; i < 288; i += 64) dyn_lit[i] = 0;
  for (int i = lane; i < STAGE_WORDS; i += 64) stage[i] = 0;
  // fixed_litlen_encoder / fixed_dist_encoder zd.ml:514-527
  for (int i = lane; i < 288; i += 64) fix_lit[i] = i <= 143 ? 8u : i <= 255 ? 9u : i <= 279 ? 7u : 8u;
  if (lane < 32) fix_dist[lane] = 5;
  wave_sync();
  wave_init_with_lengths(fix_lit, 287, coder_scratch, lane);
  wave_init_with_lengths(fix_dist, 31, coder_scratch, lane);
  wave_sync();

  BitOut bo;
  bo.dst = dst_arena + sd.dst_off;
  bo.out_pos = 0;
  bo.acc = 0;
  bo.acc_bits = 0;
  uint32_t adler = 1;  // Adler_32.init
  uint32_t status = ST_OK;
#ifdef ZD_EMIT_PHASES  // timing-only build: the results carry cycle counts instead (tools/exp_emit_phases.py)
  uint64_t ph_hist = 0, ph_code = 0;
  const uint64_t ph_begin = __builtin_readcyclecounter();
#endif

  if (MODE == 2) {  // where the scan put the block, and the parts before this one its first bit; the bits before come from the seal
    uint64_t at = plan->bit_start;
    if (part != 0) {
      at += plan->kind == 2 ? plan->dbits - plan->dyn_sym_bits : 3u;  // type bits and header
      for (uint32_t i = 0; i < part; i++) at += plan->part_bits[i];
    }
    bo.out_pos = (uint32_t)(at >> 3);
    bo.acc_bits = (int)(at & 7u);
  }
  const uint32_t b_first = MODE == 0 ? 0u : only_block, b_end = MODE == 0 ? nblk : only_block + 1u;
  for (uint32_t b = b_first; b < b_end && status == ST_OK; b++) {
    const BlockDesc bd = blocks[b];
    const bool final = b + 1 == nblk;
    // deflated_block_src_crc zd.ml:1081-1086 (Adler: one update call per block)
    if (MODE == 0 && (crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC)) adler = wave_adler_update(adler, src + bd.src_start, bd.src_len, lane, crc_op == CRC_ADLER32_RFC);
    if (MODE == 1) {  // the chunk sums of that call (wave_adler_update's loop), applied in order by the scan
      uint32_t n_chunks = 0;
      if (crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC) {
        uint32_t start = 0, chunk_len = bd.src_len % ADLER_CHUNK;
        while (start < bd.src_len) {
          uint32_t S1, S2;
          wave_adler_chunk_sums(src + bd.src_start + start, chunk_len, lane, S1, S2);
          if (lane == 0) plan->adler[n_chunks] = make_uint2(S1, S2);
          n_chunks++;
          start += chunk_len;
          chunk_len = ADLER_CHUNK;
        }
      }
      if (lane == 0) plan->n_chunks = n_chunks;
    }
    int kind = 0;
    uint64_t block_bits = 0;
    if (MODE < 2) {

#ifdef ZD_EMIT_PHASES
    const uint64_t ph0 = __builtin_readcyclecounter();
#endif
    // symbol histograms (write_lit_symbol / write_backref_symbol zd.ml:1125-1136)
    for (int i = lane; i < 288; i += 64) lit_freq[i] = 0;
    if (lane < 32) dist_freq[lane] = 0;
    wave_sync();
    // length -> litlen symbol as a table (in the heap's LDS, idle until the codes are built)
    for (int len = lane; len <= MAX_MATCH_LEN; len += 64) heap[len] = len >= MIN_MATCH_LEN - 1 ? (uint32_t)length_to_sym(len) : 0u;
    wave_sync();
    // 4 tiles of symbols per turn, the next turn's requested before this one's are
    // counted.  The loads are unconditional (clamped index, validity tested at use) so
    // that the wait counters stay exact and nothing waits for the newest requests.
    {
      // (the block's numbers came by a vector load: made scalar here, the loop's bounds and bases are scalar work and a
      // symbol's address is a scalar base plus a lane part -- round 6, from the assembly: 5 vector instructions a load were 8)
      const uint32_t n_syms_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)bd.n_syms);
      const uint32_t last_sym = n_syms_s ? n_syms_s - 1u : 0u;
      const uint32_t *bsyms = syms + (uint32_t)__builtin_amdgcn_readfirstlane((int)bd.sym_start);
      auto load4 = [&](uint32_t k0, uint32_t *v) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t k = k0 + 64u * (uint32_t)u + (uint32_t)lane;
          v[u] = bsyms[k < last_sym ? k : last_sym];
        }
      };
      // (two sets of registers that swap roles by name: copied from "next" to "current" at the end of a turn,
      // the values just requested had to arrive first, and nothing was in flight while a turn was counted)
      auto count4 = [&](uint32_t k0, const uint32_t *v) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t k = k0 + 64u * (uint32_t)u + (uint32_t)lane;
          if (k < n_syms_s) {
            if ((v[u] >> 9) == 0) atomicAdd(&lit_freq[v[u]], 1u);
            else {
              atomicAdd(&lit_freq[heap[v[u] & 0x1FF]], 1u);
              atomicAdd(&dist_freq[dist_to_sym((int)(v[u] >> 9))], 1u);
            }
          }
        }
      };
      uint32_t va[4], vb[4];
      load4(0, va);
      for (uint32_t k0 = 0; k0 < n_syms_s; k0 += 512) {
        load4(k0 + 256u, vb);
        count4(k0, va);
        load4(k0 + 512u, va);
        count4(k0 + 256u, vb);  // (past the block's end: nothing is counted)
      }
    }
    wave_sync();
#ifdef ZD_EMIT_PHASES
    const uint64_t ph1 = __builtin_readcyclecounter();
#endif
    if (lane == 0) lit_freq[LITLEN_EOB] = 1;  // add_end_of_block_sym zd.ml:1088-1092
    wave_sync();
    if (MODE == 1) {
      wave_make_dynamic_syms(c, coder_scratch, lane);
      const uint64_t flen = 3 + wave_symbols_bits(c, c.fix_lit, c.fix_dist, lane);
      const uint64_t dsb = wave_symbols_bits(c, c.dyn_lit, c.dyn_dist, lane);
      for (int i = lane; i < 288; i += 64) plan->dyn_lit[i] = dyn_lit[i];
      if (lane < 32) plan->dyn_dist[lane] = dyn_dist[lane];
      for (int i = lane; i < c.codelen_syms_len; i += 64) plan->codelen_syms[i] = codelen_syms[i];
      if (lane < 19) plan->codelen_freq[lane] = codelen_freq[lane];
      if (lane == 0) {
        plan->flen = flen; plan->dyn_sym_bits = dsb;
        plan->codelen_syms_len = c.codelen_syms_len; plan->hlit = c.hlit; plan->hdist = c.hdist;
      }
      return;
    }
    wave_make_dynamic(c, coder_scratch, lane);
    uint64_t flen, dlen;
    kind = wave_choose(c, bd.src_len, bo.acc_bits, flen, dlen, lane);
    block_bits = kind == 1 ? flen : dlen;
    wave_sync();
#ifdef ZD_EMIT_PHASES
    const uint64_t ph2 = __builtin_readcyclecounter();
    ph_hist += ph1 - ph0;
    ph_code += ph2 - ph1;
#endif
    } else {  // MODE 2, 3: the plan's codes
      kind = (int)plan->kind;
      if (kind == 2) {
        for (int i = lane; i < 288; i += 64) dyn_lit[i] = plan->dyn_lit[i];
        if (lane < 32) { dyn_dist[lane] = plan->dyn_dist[lane]; dyn_codelen[lane] = plan->dyn_codelen[lane]; }
        c.codelen_syms_len = plan->codelen_syms_len; c.hlit = plan->hlit; c.hdist = plan->hdist; c.hclen = plan->hclen;
        for (int i = lane; i < c.codelen_syms_len; i += 64) codelen_syms[i] = plan->codelen_syms[i];
      }
      wave_sync();
    }

    if (kind == 0) {
      // write_non_compressed_block zd.ml:873-877
      const uint32_t hdr_bits = (uint32_t)bo.acc_bits + 3;
      const uint32_t hdr_bytes = (hdr_bits + 7) >> 3;
      const uint64_t need = (uint64_t)bo.out_pos + hdr_bytes + 4 + bd.src_len;
      if (MODE == 0 && need > dst_cap) { status = ST_DST_TOO_SMALL; break; }
      uint8_t *o = bo.dst + bo.out_pos;
      if (lane == 0) {
        const uint32_t v = bo.acc | ((final ? 1u : 0u) << bo.acc_bits);
        o[0] = (uint8_t)v;
        if (hdr_bytes == 2) o[1] = (uint8_t)(v >> 8);
        uint8_t *q = o + hdr_bytes;
        q[0] = (uint8_t)bd.src_len;
        q[1] = (uint8_t)(bd.src_len >> 8);
        q[2] = (uint8_t)(~bd.src_len);
        q[3] = (uint8_t)((~bd.src_len) >> 8);
        stage[0] = 0;
      }
      wave_copy(o + hdr_bytes + 4, src + bd.src_start, bd.src_len, lane);
      bo.out_pos = (uint32_t)need;
      bo.acc = 0;
      bo.acc_bits = 0;
      wave_sync();
      continue;
    }

    // fixed (zd.ml:912-916) or dynamic (zd.ml:918-945) block
    if (MODE == 0) {
      const uint64_t need = (uint64_t)bo.out_pos + (((uint64_t)bo.acc_bits + block_bits + 7) >> 3);
      if (need > dst_cap) { status = ST_DST_TOO_SMALL; break; }
    }
    const uint32_t *hl = kind == 1 ? fix_lit : dyn_lit;
    const uint32_t *hd = kind == 1 ? fix_dist : dyn_dist;
    const uint32_t n_syms_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)bd.n_syms);  // (scalar: see the histogram loop)
    const uint32_t *bsyms = syms + (uint32_t)__builtin_amdgcn_readfirstlane((int)bd.sym_start);
    kind = __builtin_amdgcn_readfirstlane(kind);  // (wave-uniform by construction; the compiler cannot see it)
    const uint32_t n_hdr = (uint32_t)__builtin_amdgcn_readfirstlane(kind == 2 ? dyn_header_items(c) : 0);
    const uint32_t n_items = 1 + n_hdr + n_syms_s + 1;  // type bits, header, symbols, EOB
    // What a match length turns into -- code and extra bits merged, like the
    // reference's single write_bits (zd.ml:893-899) -- depends on the length and the
    // block's code only: one table entry per length, (bits << 5) | count, and per
    // distance symbol its base and extra-bit count; both in the heap's LDS, idle again.
    uint32_t *len_item = heap, *dist_info = heap + 272;
    for (int len = MIN_MATCH_LEN - 1 + lane; len <= MAX_MATCH_LEN; len += 64) {
      const int lsym = length_to_sym(len);
      const uint32_t si = hl[lsym];
      const uint32_t count = si & 0x1F;
      uint32_t vbase, vextra;
      length_sym_value(lsym, vbase, vextra);
      len_item[len] = (((si >> 5) | (((uint32_t)len - vbase) << count)) << 5) | (count + vextra);
    }
    if (lane <= DIST_SYM_MAX) {
      uint32_t vbase, vextra;
      dist_sym_value(lane, vbase, vextra);
      dist_info[lane] = (vbase << 5) | vextra;
    }
    wave_sync();
    // symbols of the next tiles are requested before the current ones are packed
    // item idx of the block: 0 type bits, 1..n_hdr the dynamic header, then the symbols,
    // then EOB.  Symbol loads are unconditional (clamped); what an item is gets decided
    // when it is used, a turn after its load was issued.
    const uint32_t last_sym = n_syms_s ? n_syms_s - 1u : 0u;
    // MODE 2: this part's items -- the type bits and the header go with part 0, the end-of-block symbol with the last
    const uint32_t n_parts = MODE == 2 ? emit_parts_of(split, (uint32_t)kind, n_syms_s) : 1u;
    const uint32_t item_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)((MODE == 2 && part != 0) ? 1u + n_hdr + part * EMIT_PART : 0u));
    const uint32_t item_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)((MODE == 2 && part + 1 != n_parts) ? 1u + n_hdr + (part + 1u) * EMIT_PART : n_items));
    auto fetch = [&](uint32_t idx) -> uint32_t {  // (item counts are far below 2^31)
      int32_t k = (int32_t)(idx - 1u - n_hdr);
      k = k < 0 ? 0 : k;
      k = k < (int32_t)last_sym ? k : (int32_t)last_sym;
      return bsyms[(uint32_t)k];
    };
    // a symbol's bits (write_block_symbols zd.ml:879-910, symbol_bits), by table
    auto encode = [&](uint32_t sr, uint64_t &value, int &nbits) {
      const uint32_t dist = sr >> 9, len = sr & 0x1FF;
      if (dist == 0) {
        const uint32_t si = hl[len];
        value = si >> 5;
        nbits = (int)(si & 0x1F);
      } else {
        const uint32_t li = len_item[len];
        const int dsym = dist_to_sym((int)dist);
        const uint32_t si = hd[dsym], di = dist_info[dsym];
        const uint32_t n = li & 0x1F, count = si & 0x1F;
        value = (uint64_t)(li >> 5) | (((uint64_t)(si >> 5) | ((uint64_t)(dist - (di >> 5)) << count)) << n);
        nbits = (int)(n + count + (di & 0x1F));
      }
    };
    // one turn: PACK_TILES tiles of items from `base` on, their symbols in sref (requested a turn ago)
    auto pack_turn = [&](uint32_t base, const uint32_t *sref) {
      uint64_t value[PACK_TILES];
      int nbits[PACK_TILES];
#pragma unroll
      for (int u = 0; u < PACK_TILES; u++) {
        const uint32_t t_lo = base + 64u * (uint32_t)u;  // (scalar)
        const uint32_t idx = t_lo + (uint32_t)lane;
        value[u] = 0;
        nbits[u] = 0;
        if (t_lo > n_hdr && t_lo + 64u <= 1u + n_hdr + n_syms_s && t_lo + 64u <= item_hi) {  // wave-uniform: 64 symbols, nothing else -- all but a block's first and last tiles
          encode(sref[u], value[u], nbits[u]);
        } else if (idx < item_hi) {
          if (idx == 0) {
            value[u] = (final ? 1u : 0u) | ((uint32_t)kind << 1);
            nbits[u] = 3;
          } else if (idx <= n_hdr) {
            uint32_t v;
            dyn_header_item(c, (int)idx - 1, v, nbits[u]);
            value[u] = v;
          } else {
            encode(idx - 1u - n_hdr < n_syms_s ? sref[u] : (uint32_t)LITLEN_EOB, value[u], nbits[u]);
          }
        }
      }
      pack_tiles(bo, stage, value, nbits, lane);
    };
    if (MODE == 3) {  // the bits of the part's symbols (not the end-of-block symbol)
      const uint32_t s_lo = part * EMIT_PART, s_hi = n_syms_s - s_lo > EMIT_PART ? s_lo + EMIT_PART : n_syms_s;
      uint64_t acc = 0;
      for (uint32_t k = s_lo + (uint32_t)lane; k < s_hi; k += 64u) {
        const uint32_t sr = bsyms[k];
        const uint32_t dist = sr >> 9, len = sr & 0x1FF;
        if (dist == 0) acc += hl[len] & 0x1F;
        else {
          const int dsym = dist_to_sym((int)dist);
          acc += (len_item[len] & 0x1F) + (hd[dsym] & 0x1F) + (dist_info[dsym] & 0x1F);
        }
      }
      acc = wave_sum64(acc);
      if (lane == 0) plan->part_bits[part] = acc;
      return;
    }
    // (two sets of registers that swap roles by name, as in the histogram loop)
    uint32_t sref_a[PACK_TILES], sref_b[PACK_TILES];
#pragma unroll
    for (int u = 0; u < PACK_TILES; u++) sref_a[u] = fetch(item_lo + (uint32_t)(64 * u + lane));
    for (uint32_t base = item_lo; base < item_hi; base += 2u * 64u * PACK_TILES) {
#pragma unroll
      for (int u = 0; u < PACK_TILES; u++) sref_b[u] = fetch(base + 64u * (uint32_t)(PACK_TILES + u) + (uint32_t)lane);
      pack_turn(base, sref_a);
      const uint32_t base2 = base + 64u * PACK_TILES;
      if (base2 >= item_hi) break;  // uniform
#pragma unroll
      for (int u = 0; u < PACK_TILES; u++) sref_a[u] = fetch(base2 + 64u * (uint32_t)(PACK_TILES + u) + (uint32_t)lane);
      pack_turn(base2, sref_b);
    }
  }

  if (MODE == 2) {  // what the part leaves of its last byte: the seal puts it there
    if (lane == 0) plan->part_tail[part] = bo.acc;
    return;
  }
  if (status == ST_OK && bo.acc_bits > 0) {  // flush zd.ml:856-858
    if (bo.out_pos + 1 > dst_cap) status = ST_DST_TOO_SMALL;
    else {
      if (lane == 0) bo.dst[bo.out_pos] = (uint8_t)bo.acc;
      bo.out_pos += 1;
    }
  }
  if (lane == 0) {
    StreamResult r;
    r.status = status;
    r.out_len = status == ST_OK ? bo.out_pos : 0;
    r.checksum = ((crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC) && status == ST_OK) ? adler : 0u;  // CRC-32: checksum pass
#ifdef ZD_EMIT_PHASES
    r.checksum = (uint32_t)((__builtin_readcyclecounter() - ph_begin) >> 4);
    r.out_len = (ph_hist >> 4) | ((ph_code >> 4) << 32);
#endif
    results[stream] = r;
  }
}

__global__ __launch_bounds__(64, 4) void deflate_emit_kernel(const uint8_t *__restrict__ src_arena,
                                                          uint8_t *__restrict__ dst_arena,
                                                          const StreamDesc *__restrict__ descs,
                                                          StreamResult *__restrict__ results,
                                                          DeflateScratch S, int crc_op) {
  deflate_emit_wave<0>(src_arena, dst_arena, descs, results, S, crc_op, blockIdx.x, 0, nullptr);
}

// ---------------------------------------------------------------------------------
// A stream's blocks by a wave each.  A block's bits depend on the blocks before it in three ways only: where it
// starts (and, through the padding of a stored block, which kind is smallest), the code-length code (its counts
// are never reset, Q1) and the running Adler-32.  So: deflate_plan_kernel does per block everything else;
// deflate_counts_kernel adds the code-length counts up along each stream (a wave scan per symbol over 64 blocks a
// turn); deflate_codelen_kernel, a wave per block again, builds the 19-symbol code and the block's dynamic sizes;
// deflate_scan_kernel, a wave per stream, is what is left of the chain: the three sizes compared, bit offsets, the
// Adler-32 chunk steps, the stream's result -- a few dozen scalar operations per block; deflate_pack_kernel writes
// every block from its first bit on (whole bytes; the low bits of its first byte are the block's before, left zero)
// and deflate_seal_kernel ORs those bits in and writes the stream's last byte.
__global__ __launch_bounds__(64, 4) void deflate_plan_kernel(const uint8_t *__restrict__ src_arena,
                                                          const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                          int crc_op, uint32_t blocks_per_stream, EmitPlan *__restrict__ plans) {
  deflate_emit_wave<1>(src_arena, nullptr, descs, nullptr, S, crc_op, blockIdx.x / blocks_per_stream,
                       blockIdx.x % blocks_per_stream, plans);
}

// (grids of a wave per part of a block: slot = (stream * blocks_per_stream + block) * parts_per_block + part;
// parts_per_block is EMIT_PARTS or, for a wave per block, 1)
__global__ __launch_bounds__(64, 4) void deflate_bits_kernel(const uint8_t *__restrict__ src_arena,
                                                          const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                          int crc_op, uint32_t blocks_per_stream, EmitPlan *__restrict__ plans) {
  const uint32_t blk = blockIdx.x / EMIT_PARTS;
  deflate_emit_wave<3>(src_arena, nullptr, descs, nullptr, S, crc_op, blk / blocks_per_stream, blk % blocks_per_stream,
                       plans, blockIdx.x % EMIT_PARTS, true);
}

__global__ __launch_bounds__(64, 4) void deflate_pack_kernel(const uint8_t *__restrict__ src_arena,
                                                          uint8_t *__restrict__ dst_arena,
                                                          const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                          int crc_op, uint32_t blocks_per_stream, EmitPlan *__restrict__ plans,
                                                          uint32_t parts_per_block) {
  const uint32_t blk = blockIdx.x / parts_per_block;
  deflate_emit_wave<2>(src_arena, dst_arena, descs, nullptr, S, crc_op, blk / blocks_per_stream, blk % blocks_per_stream,
                       plans, blockIdx.x % parts_per_block, parts_per_block != 1);
}

// the counts of the code-length symbols as block b's code sees them: its own and those of every block before (Q1)
__global__ __launch_bounds__(64) void deflate_counts_kernel(const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                            EmitPlan *__restrict__ plans) {
  const uint32_t stream = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (S.error[0] || sd.src_len > MAX_STREAM_LEN || sd.dst_cap > MAX_STREAM_LEN) return;
  const uint32_t nblk = S.n_blocks[stream];
  EmitPlan *P = plans + S.blk_base[stream];
  uint32_t carry[19];
#pragma unroll
  for (int k = 0; k < 19; k++) carry[k] = 0;
  for (uint32_t c = 0; c < nblk; c += 64) {  // 64 blocks a turn, one per lane: a wave scan per symbol
    const bool mine = c + lane < nblk;
    EmitPlan *p = P + (mine ? c + lane : nblk - 1);
#pragma unroll
    for (int k = 0; k < 19; k++) {
      const uint32_t incl = wave_scan_incl(mine ? p->codelen_freq[k] : 0u) + carry[k];
      if (mine) p->cum_freq[k] = incl;
      carry[k] = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
}

// the code of the code-length symbols of ONE block from those counts, the block's size as a dynamic block and
// the reference's estimate of it (wave_choose's dlen)
__global__ __launch_bounds__(64) void deflate_codelen_kernel(const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                             uint32_t blocks_per_stream, EmitPlan *__restrict__ plans) {
  __shared__ uint32_t codelen_freq[32], dyn_codelen[32], heap[580], coder_scratch[32];
  const uint32_t stream = blockIdx.x / blocks_per_stream, b = blockIdx.x % blocks_per_stream;
  const int lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (S.error[0] || sd.src_len > MAX_STREAM_LEN || sd.dst_cap > MAX_STREAM_LEN) return;
  if (b >= S.n_blocks[stream]) return;
  EmitPlan *p = plans + S.blk_base[stream] + b;
  BlockCoder c{};
  c.codelen_freq = codelen_freq; c.dyn_codelen = dyn_codelen; c.heap = heap;
  if (lane < 32) { codelen_freq[lane] = lane < 19 ? p->cum_freq[lane] : 0u; dyn_codelen[lane] = 0; }
  wave_sync();
  wave_make_dynamic_codelen(c, coder_scratch, lane);
  wave_sync();
  // (the estimate counts the code-length symbols of every block so far -- that is the reference's, and what the
  // choice and the capacity test go by; the block itself holds its own)
  uint64_t acc = 0, own = 0;
  if (lane <= CODELEN_SYM_MAX) {
    const uint32_t rb = lane == 16 ? 2 : lane == 17 ? 3 : lane == 18 ? 7 : 0;
    acc = (uint64_t)codelen_freq[lane] * ((dyn_codelen[lane] & 0x1F) + rb);
    own = (uint64_t)p->codelen_freq[lane] * ((dyn_codelen[lane] & 0x1F) + rb);
  }
  const uint64_t dhead = 3 + 5 + 5 + 4 + 3 * (uint64_t)(c.hclen + 4) + p->dyn_sym_bits;
  const uint64_t dlen = dhead + wave_sum64(acc), dbits = dhead + wave_sum64(own);
  if (lane < 32) p->dyn_codelen[lane] = dyn_codelen[lane];
  if (lane == 0) { p->dlen = dlen; p->dbits = dbits; p->hclen = c.hclen; }
}

// what is left of the chain through a stream's blocks: the choice (a stored block's size depends on the bit
// it starts at), the bit offsets, the Adler-32 steps, the result.  64 blocks a turn: the lanes fetch their
// block's numbers together, then the turn's blocks are gone through in order on wave-uniform values.
__global__ __launch_bounds__(64) void deflate_scan_kernel(const StreamDesc *__restrict__ descs,
                                                          StreamResult *__restrict__ results, DeflateScratch S,
                                                          int crc_op, EmitPlan *__restrict__ plans) {
  const uint32_t stream = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  if (S.error[0] || sd.src_len > MAX_STREAM_LEN || sd.dst_cap > MAX_STREAM_LEN) {
    if (lane == 0) { StreamResult r; r.status = S.error[0] == 2u ? ST_HIP : ST_INVALID_ARG; r.checksum = 0; r.out_len = 0; results[stream] = r; }
    return;
  }
  const uint32_t nblk = S.n_blocks[stream];
  const BlockDesc *blocks = S.blocks + S.blk_base[stream];
  EmitPlan *P = plans + S.blk_base[stream];
  const bool want_adler = crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC;
  uint32_t adler = 1;  // Adler_32.init
  uint64_t bits = 0;
  bool fits = true;
  for (uint32_t c = 0; c < nblk; c += 64) {
    const bool mine = c + lane < nblk;
    EmitPlan *p = P + (mine ? c + lane : nblk - 1);
    const uint64_t my_flen = p->flen, my_dlen = p->dlen, my_dbits = p->dbits;
    const uint32_t my_src_len = blocks[mine ? c + lane : nblk - 1].src_len;
    uint64_t my_start = 0, my_end = 0;
    uint32_t my_kind = 0;
    const uint32_t turn = nblk - c < 64u ? nblk - c : 64u;
    for (uint32_t i = 0; i < turn; i++) {
      const uint32_t src_len = (uint32_t)__builtin_amdgcn_readlane((int)my_src_len, (int)i);
      auto lane64 = [&](uint64_t v) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)i) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)i);
      };
      const uint64_t flen = lane64(my_flen), dlen = lane64(my_dlen), dbits = lane64(my_dbits);
      if (want_adler) {  // wave_adler_update's chunk steps, with the sums the plan left
        const EmitPlan *q = P + c + i;
        uint32_t a1, a2;
        adler_unpack(adler, a1, a2);  // (one update call per block: the value is packed and unpacked between two, zd.ml:178,198)
        const uint32_t nc = q->n_chunks;
        for (uint32_t j = 0; j < nc; j++) {
          const uint2 sm = q->adler[j];
          adler_chunk_step(a1, a2, j == 0 ? src_len % ADLER_CHUNK : ADLER_CHUNK, sm.x, sm.y, crc_op == CRC_ADLER32_RFC);
        }
        adler = adler_pack(a1, a2);
      }
      // wave_choose's estimates and choice
      const uint32_t pending = (uint32_t)(bits & 7u);
      const uint64_t nlen = 3 + (uint64_t)(8 - ((pending + 3) % 8)) + (4 + (uint64_t)src_len) * 8;
      const uint32_t kind = (nlen <= dlen && nlen <= flen) ? 0u : flen <= dlen ? 1u : 2u;
      // a stored block: its type bits, zeros up to the next byte, LEN, NLEN, the bytes (nlen is the reference's
      // estimate: 8 too many when the type bits end a byte, Q3)
      const uint64_t sbits = (uint64_t)(((pending + 3u + 7u) & ~7u) - pending) + (4 + (uint64_t)src_len) * 8;
      const uint64_t block_bits = kind == 0 ? sbits : kind == 1 ? flen : dbits;
      if (((bits + (kind == 2 ? dlen : block_bits) + 7) >> 3) > sd.dst_cap) fits = false;  // deflate_emit_kernel's test per block
      if (lane == i) { my_start = bits; my_end = bits + block_bits; my_kind = kind; }
      bits += block_bits;
    }
    if (mine) { p->bit_start = my_start; p->bit_end = my_end; p->kind = my_kind; }
  }
  const uint64_t out_len = (bits + 7) >> 3;
  const uint32_t status = (!fits || out_len > sd.dst_cap) ? (uint32_t)ST_DST_TOO_SMALL : (uint32_t)ST_OK;
  if (status != ST_OK)
    for (uint32_t b = lane; b < nblk; b += 64) P[b].kind = EMIT_SKIP;
  if (lane == 0) {
    StreamResult r;
    r.status = status;
    r.out_len = status == ST_OK ? out_len : 0;
    r.checksum = (want_adler && status == ST_OK) ? adler : 0u;  // CRC-32: checksum pass
    results[stream] = r;
  }
}

__global__ __launch_bounds__(256) void deflate_seal_kernel(uint8_t *__restrict__ dst_arena,
                                                           const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                           uint32_t n_streams, uint32_t blocks_per_stream,
                                                           const EmitPlan *__restrict__ plans, uint32_t parts_per_block) {
  const bool split = parts_per_block != 1;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t part = (uint32_t)(i % parts_per_block);
  const uint64_t ib = i / parts_per_block;
  const uint32_t stream = (uint32_t)(ib / blocks_per_stream), b = (uint32_t)(ib % blocks_per_stream);
  if (stream >= n_streams || S.error[0]) return;
  const StreamDesc sd = descs[stream];
  if (sd.src_len > MAX_STREAM_LEN || sd.dst_cap > MAX_STREAM_LEN) return;
  const uint32_t nblk = S.n_blocks[stream];
  if (b >= nblk) return;
  const EmitPlan *p = plans + S.blk_base[stream] + b;
  const BlockDesc *blocks = S.blocks + S.blk_base[stream];
  if (p->kind == EMIT_SKIP) return;
  const uint32_t n_parts = emit_parts_of(split, p->kind, blocks[b].n_syms);
  if (part >= n_parts) return;
  uint8_t *dst = dst_arena + sd.dst_off;
  // A part's pack wave (deflate_emit_wave<2>) wrote the part's whole bytes and left the low bits of its first byte --
  // the bits of what lies before it -- zero; what the part leaves of its last byte is part_tail.  ONE thread puts a
  // shared byte together: the thread of the part that COMPLETES the byte ORs in the tails of what ends inside it.
  // A part that lies wholly inside one byte (the last part of a split block: a symbol or two and the end-of-block
  // symbol under short codes) completes nothing and writes nothing -- two threads ORing into one byte lost one of the
  // two updates (round 3) -- its bits travel in the next part's `before`.
  auto start_of = [&](const EmitPlan *q, uint32_t j) -> uint64_t {  // first bit of part j of q's block
    uint64_t a = q->bit_start;
    if (j != 0) {
      a += q->kind == 2 ? q->dbits - q->dyn_sym_bits : 3u;  // type bits and header
      for (uint32_t k = 0; k < j; k++) a += q->part_bits[k];
    }
    return a;
  };
  const uint64_t at = start_of(p, part);
  const uint64_t end = part + 1 != n_parts ? start_of(p, part + 1) : p->bit_end;
  const bool completes = (at >> 3) != (end >> 3);  // the part holds the last bit of its first byte
  uint32_t before = 0;
  if ((at & 7u) && (b != 0 || part != 0)) {
    // the part before me, and -- when that one lies wholly inside this byte -- the one before it (never a third: only
    // the last part of a block can be that short, and a block's first part holds at least the 10 bits of an empty block)
    const EmitPlan *q = part != 0 ? p : p - 1;
    const uint32_t qj = part != 0 ? part - 1 : emit_parts_of(split, q->kind, blocks[b - 1].n_syms) - 1;
    before = q->part_tail[qj];
    const uint64_t qs = start_of(q, qj);
    if ((qs >> 3) == (at >> 3) && (qs & 7u) && qj != 0) before |= q->part_tail[qj - 1];
    if (completes) dst[at >> 3] |= (uint8_t)before;
  }
  if (b + 1 == nblk && part + 1 == n_parts && (p->bit_end & 7u))  // flush zd.ml:856-858
    dst[p->bit_end >> 3] = (uint8_t)(p->part_tail[part] | (completes ? 0u : before));
}

// ---------------------------------------------------------------------------------
// level `None: stored blocks of up to 65534 bytes (zd.ml:1106-1116); empty input
// gives one empty final stored block.
__global__ __launch_bounds__(64) void deflate_stored_kernel(const uint8_t *__restrict__ src_arena,
                                                            uint8_t *__restrict__ dst_arena,
                                                            const StreamDesc *__restrict__ descs,
                                                            StreamResult *__restrict__ results, int crc_op) {
  const uint32_t stream = blockIdx.x;
  const int lane = threadIdx.x;
  const StreamDesc sd = descs[stream];
  StreamResult r;
  r.status = ST_OK; r.checksum = 0; r.out_len = 0;
  if (sd.src_len > MAX_STREAM_LEN || sd.dst_cap > MAX_STREAM_LEN) {
    r.status = ST_INVALID_ARG;
    if (lane == 0) results[stream] = r;
    return;
  }
  const uint32_t len = (uint32_t)sd.src_len;
  const uint64_t nblocks = (uint64_t)len / MAX_BLOCK_SRC_LEN + ((len % MAX_BLOCK_SRC_LEN) || len == 0 ? 1 : 0);
  const uint64_t need = (uint64_t)len + 5 * nblocks;
  if (need > sd.dst_cap) {
    r.status = ST_DST_TOO_SMALL;
    if (lane == 0) results[stream] = r;
    return;
  }
  const uint8_t *src = src_arena + sd.src_off;
  uint8_t *dst = dst_arena + sd.dst_off;
  uint32_t adler = 1, start = 0;
  uint64_t out = 0;
  for (;;) {
    const uint32_t n = len - start < (uint32_t)MAX_BLOCK_SRC_LEN ? len - start : (uint32_t)MAX_BLOCK_SRC_LEN;
    const bool final = start + n == len;
    if (crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC) adler = wave_adler_update(adler, src + start, n, lane, crc_op == CRC_ADLER32_RFC);
    if (lane == 0) {
      uint8_t *q = dst + out;
      q[0] = final ? 1 : 0;
      q[1] = (uint8_t)n;
      q[2] = (uint8_t)(n >> 8);
      q[3] = (uint8_t)(~n);
      q[4] = (uint8_t)((~n) >> 8);
    }
    wave_copy(dst + out + 5, src + start, n, lane);
    out += 5 + (uint64_t)n;
    if (final) break;
    start += n;
  }
  r.out_len = out;
  r.checksum = (crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC) ? adler : 0u;
  if (lane == 0) results[stream] = r;
}

// ---------------------------------------------------------------------------------
// Tests only (zipc_hip_debug_chain_links): the hash-chain links of a batch as ONE of the two kernels makes them -- by
// ordered LDS exchange (which = 0) or by the kernel that orders equal hashes itself (1) -- copied out with the streams'
// position bases, so that a test can require the two to be equal over whole batches (insert_hash, zd.ml:1150-1152: the
// exchange kernel's order rests on a property of the hardware that only a probe vouches for).  The scratch is zeroed
// first: what no kernel writes compares equal.
hipError_t debug_chain_links(zipc_hip_ctx *ctx, const uint8_t *d_src, const StreamDesc *d_descs, size_t n, size_t max_src_len,
                             size_t total_src_len, int which, uint16_t *d_links, size_t links_cap, uint64_t *d_pos_base) {
  size_t per_group, group_total;
  deflate_grouping(n, max_src_len, total_src_len, per_group, group_total);
  if (per_group != n) return hipErrorInvalidValue;  // one group (a batch of up to 8 GiB)
  DeflateScratch S = carve(ctx->deflate_scratch.p, n, total_src_len, LEVEL_DEFAULT);
  if (S.cap_positions > links_cap) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(S.prev, 0, S.cap_positions * 2, ctx->cur);
  if (e != hipSuccess) return e;
  ZD_LAUNCH(ctx, "deflate_offsets", deflate_offsets_kernel, dim3(1), dim3(1024), 0, d_descs, (uint32_t)n, S, (uint64_t)max_src_len);
  if (which == 0) {
    if (!ctx->xchg_ordered) return hipErrorNotSupported;
    ZD_LAUNCH(ctx, "lz_chain", lz_chain_xchg_kernel, dim3((unsigned)n), dim3(64 * XCHG_WAVES), 0, d_src, d_descs, S);
  } else {
    ZD_LAUNCH(ctx, "lz_chain", lz_chain_kernel, dim3((unsigned)n), dim3(CHAIN_THREADS), 0, d_src, d_descs, S);
  }
  e = hipMemcpyAsync(d_links, S.prev, S.cap_positions * 2, hipMemcpyDeviceToDevice, ctx->cur);
  if (e == hipSuccess && d_pos_base) e = hipMemcpyAsync(d_pos_base, S.pos_base, n * 8, hipMemcpyDeviceToDevice, ctx->cur);
  if (e == hipSuccess) e = hipGetLastError();
  return e;
}
size_t debug_chain_positions(size_t n, size_t total_src_len) {
  uint64_t P, Bk;
  scratch_caps(n, total_src_len, P, Bk);
  return (size_t)P;
}

// ---------------------------------------------------------------------------------
static hipError_t launch_deflate_group(zipc_hip_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst,
                                       const StreamDesc *d_descs, StreamResult *d_results, size_t n,
                                       size_t max_src_len, size_t total_src_len, int level, int crc_op);

hipError_t launch_deflate(zipc_hip_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst,
                          const StreamDesc *d_descs, StreamResult *d_results, size_t n,
                          size_t max_src_len, size_t total_src_len, int level, int crc_op) {
  if (level == LEVEL_NONE) {
    ZD_LAUNCH(ctx, "deflate_stored", deflate_stored_kernel, dim3((unsigned)n), dim3(64), 0, d_src, d_dst,
              d_descs, d_results, crc_op);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && crc_op == CRC_CRC32) {
      uint32_t *partials = (uint32_t *)ctx->crc_partials.p;
      e = crc32_segments_launch(ctx, d_src, RANGE_DEFLATE_SRC, d_descs, nullptr, n, 0, 0, max_src_len, partials);
      if (e == hipSuccess)
        e = crc32_finish_launch(ctx, RANGE_DEFLATE_SRC, d_descs, d_results, n, 0, max_src_len, partials, nullptr);
    }
    return e;
  }
  size_t per_group, group_total;
  deflate_grouping(n, max_src_len, total_src_len, per_group, group_total);
  for (size_t g0 = 0; g0 < n; g0 += per_group) {  // the descriptors carry arena offsets: a group is a slice of them
    const size_t ng = n - g0 < per_group ? n - g0 : per_group;
    const hipError_t e = launch_deflate_group(ctx, d_src, d_dst, d_descs + g0, d_results + g0, ng, max_src_len,
                                              group_total, level, crc_op);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

static hipError_t launch_deflate_group(zipc_hip_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst,
                                       const StreamDesc *d_descs, StreamResult *d_results, size_t n,
                                       size_t max_src_len, size_t total_src_len, int level, int crc_op) {
  DeflateScratch S = carve(ctx->deflate_scratch.p, n, total_src_len, level);
  int good_match, K;
  level_params(level, good_match, K);
  ZD_LAUNCH(ctx, "deflate_offsets", deflate_offsets_kernel, dim3(1), dim3(1024), 0, d_descs, (uint32_t)n, S,
            (uint64_t)max_src_len);
  // The pipeline of a slice of the group's streams: the slices share the scratch arrays (one
  // scan above gave every stream its base) and the error word, and differ in where their
  // per-stream arrays start.
  const size_t tps = max_src_len >= 4 ? (size_t)match_tiles_of(max_src_len) : 1;
  const size_t cps = max_src_len ? (max_src_len + MATCH_TILE - 1) / MATCH_TILE : 1;
  if (n * (max_src_len <= MATCHW_SMALL ? cps : tps) > 0x7FFFFFFFull) return hipErrorInvalidValue;
  // consecutive tiles of a stream per workgroup: as many as leave the grid >= 8192
  // workgroups (32 per CU: with 2048 a group of 2048 long streams had one workgroup per stream
  // and a long tail), so few long streams still spread over the chip
  // (ZIPC_HIP_MATCH_TILES_PER_GROUP, read once, overrides the rule: tuning and tests)
  const long tpg_env = tuning().match_tiles_per_group;
  const int form_env = tuning().match_form;
  size_t tpg = tpg_env > 0 ? (size_t)tpg_env : n * tps / 8192;
  tpg = tpg < 1 ? 1 : (tpg > tps ? tps : tpg);
  const size_t gps = (tps + tpg - 1) / tpg;
  // Few long streams: lz_parse by a wave per segment (lz_parse_spec_kernel) and the blocks coded by a wave each
  // (deflate_plan_kernel); many streams fill the chip with a wave each.  ZIPC_HIP_PARSE_SEGMENTS=0 never, =1
  // whenever a stream has more than one segment (tests).
  const long segs_env = tuning().parse_segments;
  const long segp_env = tuning().parse_seg;  // positions per segment (tuning)
  // segment size: the stitch's serial time per stream is segments x ~0.25 us, the parallel part's a segment's tiles x ~0.3 us
  // (one stream alone, 4-bit symbols, whole deflate, ms at 4096 / 8192 / 16384 / 32768 / 65536 positions: 1 MiB 0.91 / 1.04 /
  // 1.09 / 1.38 / 1.94, 16 MiB 2.18 / 1.89 / 1.87 / 2.10 / 2.57 -- since lz_parse_meet_kernel the stitch's turn per
  // segment is a quarter of a microsecond)
  size_t segp = max_src_len <= ((size_t)4 << 20) ? 4096 : max_src_len <= ((size_t)32 << 20) ? 8192 : 16384;
  // (many long streams: as long as the call keeps 64 Ki waves, longer segments -- fewer seams to stitch and to gather across.
  // 8192 x 1 MiB, ms a step at 4096 / 8192 / 16 384 / 32 768 positions: 175.2 / 174.3 / 172.7 / 171.9, profiles/r06_c4_parse_segments.txt)
  while (segp < 32768 && n * ((max_src_len + 2 * segp - 1) / (2 * segp)) >= 65536) segp *= 2;
  if (segp_env >= (long)PARSE_SEG_MIN && segp_env % 64 == 0 && segp_env <= (1L << 20)) segp = (size_t)segp_env;
  const size_t sps = (max_src_len + segp - 1) / segp;
  // (4096 x 1 MiB: 133 -> 124 ms; 64 KiB streams, 256 / 1024 / 2048 / 4096 / 16 384 of them:
  // 1.77 -> 0.73, 2.38 -> 1.67, 3.16 -> 2.85, 4.78 -> 5.11, 15.6 -> 17.4 ms; 8192 x 1 MiB, BASELINE's C4: the same either way
  // until round 6, then -- both parses a third shorter in instructions, the one wave per member still waiting for the LDS
  // 60 % of its cycles -- 182.1 -> 175.2 ms a step, profiles/r06_c4_parse_segments.txt)
  bool segmented = segs_env == 0 ? false : segs_env == 1 ? sps > 1
                   : (sps >= 8 && (n <= 2048 || (n <= 4096 && max_src_len >= ((size_t)512 << 10)) || (n <= 8192 && max_src_len >= ((size_t)1 << 20))));
  const size_t bps = (size_t)max_blocks_of(max_src_len);  // block slots of the longest stream
  const size_t chain_seg = n * ((max_src_len + CHAIN_SEG_MIN - 1) / CHAIN_SEG_MIN) <= 512 ? CHAIN_SEG_MIN
                           : n * ((max_src_len + 2 * CHAIN_SEG_MIN - 1) / (2 * CHAIN_SEG_MIN)) <= 1024 ? 2 * CHAIN_SEG_MIN : CHAIN_SEG_MAX;
  const size_t csegs = (max_src_len + chain_seg - 1) / chain_seg;  // lz_chain: workgroups of the longest stream
  if (segmented && (n * sps > 0x7FFFFFFFull || n * bps * EMIT_PARTS > 0x7FFFFFFFull)) segmented = false;
  ParseSegs segs{};
  EmitPlan *plans = nullptr;
  if (segmented) {
    const size_t n_slots = (size_t)(S.cap_positions / segp) + n + 1, tiles = (size_t)(S.cap_positions / 64) + 4;  // (ParseSegs::slot)
    const size_t plan_bytes = align_up((size_t)S.cap_blocks * sizeof(EmitPlan), 256);
    const size_t seg_syms = segp + PARSE_SEG_SLACK;
    const size_t bytes = plan_bytes + align_up(n_slots * seg_syms * 4, 256) + 2 * align_up(tiles * 8, 256) +
                         2 * align_up(tiles * 4, 256) + 11 * align_up(n_slots * 4, 256) + align_up(n_slots * MEET_CAP * 4, 256);
    if (ctx->ensure(ctx->parse_scratch, bytes) != hipSuccess) {
      // (8192 members of 1 MiB ask for 39 GB of segment symbols here: where the device cannot give them, the forms by a wave per
      // stream -- which need none -- take the call, as they did for this shape until round 6; only a call that ASKED for segments fails)
      (void)hipGetLastError();
      if (segs_env == 1) return hipErrorOutOfMemory;
      segmented = false;
    }
  }
  if (segmented) {
    const size_t n_slots = (size_t)(S.cap_positions / segp) + n + 1, tiles = (size_t)(S.cap_positions / 64) + 4;
    const size_t plan_bytes = align_up((size_t)S.cap_blocks * sizeof(EmitPlan), 256);
    const size_t seg_syms = segp + PARSE_SEG_SLACK;
    uint8_t *q = (uint8_t *)ctx->parse_scratch.p;
    plans = (EmitPlan *)q; q += plan_bytes;
    segs.spec_syms = (uint32_t *)q; q += align_up(n_slots * seg_syms * 4, 256);
    segs.vis = (unsigned long long *)q; q += align_up(tiles * 8, 256);
    segs.tile_sym0 = (uint32_t *)q; q += align_up(tiles * 4, 256);
    segs.seg_exit = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.seg_total = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.seg_dst = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.seg_from = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.seg_n = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.meet_f = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.meet_from = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.meet_exit = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.meet_end = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.fix_dst = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.fix_n = (uint32_t *)q; q += align_up(n_slots * 4, 256);
    segs.vis2 = (unsigned long long *)q; q += align_up(tiles * 8, 256);
    segs.sym02 = (uint32_t *)q; q += align_up(tiles * 4, 256);
    segs.meet_syms = (uint32_t *)q;
    segs.segs_per_stream = (uint32_t)sps;
    segs.seg_positions = (uint32_t)segp; segs.seg_syms = (uint32_t)seg_syms;
    const hipError_t me = hipMemsetAsync(segs.vis, 0, tiles * 8, ctx->cur);
    if (me != hipSuccess) return me;
  }
  // lz_chain: by ordered exchange where the context's probe passed (ZIPC_HIP_CHAIN=peel keeps the peel kernel: tests, A/B).
  // A wave per stream leaves most of the chip idle while there are fewer streams than CUs: a long stream is then cut into
  // segments of xseg positions, each warmed up with the 32 Ki positions before it (at most a third more work at 96 Ki).
  // A context whose first batch's check found a difference (counted in device-visible host memory, read here without waiting:
  // the next call at the latest sees it) keeps the other kernel from then on.
  if (ctx->xchg_ordered && ctx->chain_check_host && ctx->chain_check_host[0] != 0) {
    ctx->xchg_ordered = false;
    ctx->last_error = "zipc_hip: lz_chain by ordered LDS exchange disagreed with the ordering kernel on this context's first batch; "
                      "that batch's streams reported ZIPC_HIP_ERR_HIP, the context now orders equal hashes itself";
  }
  const bool xchg_chain = ctx->xchg_ordered && !tuning().chain_peel;
  // ZIPC_HIP_CHAIN_CHECK=N (default 32; 0: never): the first N streams of the context's FIRST batch -- as many of them as hold
  // 16 MiB of source -- are chained by both kernels and compared (chain_check_enqueue)
  size_t check_k = 0;
  if (xchg_chain && !ctx->chain_checked && tuning().chain_check > 0 && ctx->chain_check_host) {
    check_k = (size_t)tuning().chain_check < n ? (size_t)tuning().chain_check : n;
    const size_t fit = max_src_len ? ((size_t)16 << 20) / max_src_len : check_k;
    check_k = check_k > fit ? (fit ? fit : 1) : check_k;
    ctx->chain_checked = true;
  }
  size_t xseg = 0, xsegs = 1;
  if (xchg_chain && n < 1024 && max_src_len > ((size_t)192 << 10)) {
    xseg = (size_t)96 << 10;
    while (n * ((max_src_len + 2 * xseg - 1) / (2 * xseg)) >= 2048) xseg *= 2;  // twice the chip's CUs of waves is plenty
    xsegs = (max_src_len + xseg - 1) / xseg;
  }
  hipError_t slice_err = hipSuccess;
  auto slice = [&](size_t lo, size_t hi) {
    const size_t m = hi - lo;
    DeflateScratch Q = S;
    Q.pos_base += lo; Q.blk_base += lo; Q.n_blocks += lo; Q.snap_used += lo;
    const StreamDesc *dd = d_descs + lo;
    if (xchg_chain) {  // one wave per stream (per segment of a long one while there are few): ordered LDS exchange
      if (m * xsegs > m && m * xsegs <= 0x7FFFFFFFull)
        ZD_LAUNCH(ctx, "lz_chain", lz_chain_xchg_segments_kernel, dim3((unsigned)(m * xsegs)), dim3(64 * XCHG_WAVES), 0, d_src, dd, Q,
                  (uint32_t)xsegs, (uint32_t)xseg);
      else
        ZD_LAUNCH(ctx, "lz_chain", lz_chain_xchg_kernel, dim3((unsigned)m), dim3(64 * XCHG_WAVES), 0, d_src, dd, Q);
      if (lo == 0 && check_k) {
        const hipError_t ce = chain_check_enqueue(ctx, d_src, dd, Q, check_k < m ? check_k : m, max_src_len, ctx->chain_check_host);
        if (ce != hipSuccess) slice_err = ce;
      }
    } else if (segmented && csegs > 1 && m <= 128)  // (the run-up is a quarter more work: only while workgroups are what is missing)
      ZD_LAUNCH(ctx, "lz_chain", lz_chain_segments_kernel, dim3((unsigned)(m * csegs)), dim3(CHAIN_THREADS), 0, d_src, dd,
                Q, (uint32_t)csegs, (uint32_t)chain_seg);
    else
      ZD_LAUNCH(ctx, "lz_chain", lz_chain_kernel, dim3((unsigned)m), dim3(CHAIN_THREADS), 0, d_src, dd, Q);
    if (max_src_len <= MATCHW_SMALL)  // short streams: a whole-CU window per tile would sit mostly idle
      ZD_LAUNCH(ctx, "lz_match", lz_match_kernel, dim3((unsigned)((m * cps + 7) / 8 * 8)), dim3(MATCH_THREADS), 0,
                d_src, dd, Q, (uint32_t)m, (uint32_t)cps, K, K / 4);
    else {
      // groups a workgroup takes one behind the other: as many as leave 2048 workgroups and more, 8 at most (measured 1 / 2 / 4 / 8:
      // the benchmark's streams 4.70 / 4.60 / 4.50 / 4.50 ms, text 48.6 / 47.0 / 45.7 / 46.2, 1 MiB members of 3-bit symbols 31.9 / 32.2 / 32.7 / 30.6)
      size_t gpw = m * gps / 2048;
      gpw = gpw < 1 ? 1 : gpw > (size_t)ZD_MATCH_GROUPS_PER_WG ? (size_t)ZD_MATCH_GROUPS_PER_WG : gpw;
      // (... of groups that are short: a launch ends when its last workgroup does, and at `Best a group of text takes milliseconds --
      // 2048 streams: 8 workgroups of two groups a CU 76.8 ms, 16 of one 66.5.  Taking the groups from a counter instead of by
      // position in the grid evened that out -- 66.0 -- and cost 1 MiB members 13 % and the benchmark's streams 2-10 %; three
      // quarters of the groups in workgroups of several and the rest in workgroups of one: 73.0.  Both measured, neither kept.)
      if (K >= 1024) gpw = 1;
      const size_t wgs = (m * gps + gpw - 1) / gpw;
      ZD_LAUNCH(ctx, "lz_match", lz_match_window_kernel, dim3((unsigned)((wgs + 7) / 8 * 8)), dim3(MATCHW_THREADS),
                0, d_src, dd, Q, (uint32_t)m, (uint32_t)tps, (uint32_t)tpg, (uint32_t)gpw, K, K / 4, form_env);
    }
    if (segmented) {
      ParseSegs G = segs;
      const size_t o = lo;  // the slice's segment slots: ParseSegs::slot counts streams from the slice's first (Q.pos_base is the slice's too)
      G.spec_syms += o * G.seg_syms;
      G.seg_exit += o; G.seg_total += o; G.seg_dst += o; G.seg_from += o; G.seg_n += o;
      G.meet_syms += o * MEET_CAP;
      G.meet_f += o; G.meet_from += o; G.meet_exit += o; G.meet_end += o; G.fix_dst += o; G.fix_n += o;
      ZD_LAUNCH(ctx, "lz_parse_spec", lz_parse_spec_kernel, dim3((unsigned)(m * sps)), dim3(64), 0, d_src, dd, Q,
                good_match, G);
      ZD_LAUNCH(ctx, "lz_parse_meet", lz_parse_meet_kernel, dim3((unsigned)(m * sps)), dim3(64), 0, d_src, dd, Q,
                good_match, G);
      ZD_LAUNCH(ctx, "lz_parse_stitch", lz_parse_stitch_kernel, dim3((unsigned)m), dim3(64), 0, d_src, dd, Q,
                good_match, G);
      ZD_LAUNCH(ctx, "lz_parse_gather", lz_parse_gather_kernel, dim3((unsigned)(m * sps)), dim3(64), 0, dd, Q, G);
    } else {
      ZD_LAUNCH(ctx, "lz_parse", lz_parse_kernel, dim3((unsigned)m), dim3(64), 0, d_src, dd, Q, good_match);
    }
    if (segmented) {
      ZD_LAUNCH(ctx, "deflate_plan", deflate_plan_kernel, dim3((unsigned)(m * bps)), dim3(64), 0, d_src, dd, Q, crc_op,
                (uint32_t)bps, plans);
      ZD_LAUNCH(ctx, "deflate_counts", deflate_counts_kernel, dim3((unsigned)m), dim3(64), 0, dd, Q, plans);
      ZD_LAUNCH(ctx, "deflate_codelen", deflate_codelen_kernel, dim3((unsigned)(m * bps)), dim3(64), 0, dd, Q,
                (uint32_t)bps, plans);
      ZD_LAUNCH(ctx, "deflate_scan", deflate_scan_kernel, dim3((unsigned)m), dim3(64), 0, dd, d_results + lo, Q, crc_op,
                plans);
      // few blocks in the call: a coded block's symbols by a wave per EMIT_PART of them
      const size_t ppb = m * bps <= 2048 ? EMIT_PARTS : 1;
      if (ppb != 1)
        ZD_LAUNCH(ctx, "deflate_bits", deflate_bits_kernel, dim3((unsigned)(m * bps * EMIT_PARTS)), dim3(64), 0, d_src, dd, Q,
                  crc_op, (uint32_t)bps, plans);
      ZD_LAUNCH(ctx, "deflate_pack", deflate_pack_kernel, dim3((unsigned)(m * bps * ppb)), dim3(64), 0, d_src, d_dst,
                dd, Q, crc_op, (uint32_t)bps, plans, (uint32_t)ppb);
      ZD_LAUNCH(ctx, "deflate_seal", deflate_seal_kernel, dim3((unsigned)((m * bps * ppb + 255) / 256)), dim3(256), 0,
                d_dst, dd, Q, (uint32_t)m, (uint32_t)bps, (const EmitPlan *)plans, (uint32_t)ppb);
    } else {
      ZD_LAUNCH(ctx, "deflate_emit", deflate_emit_kernel, dim3((unsigned)m), dim3(64), 0, d_src, d_dst, dd,
                d_results + lo, Q, crc_op);
    }
    // The CRC-32 pass over the slice's source (deflated_block_src_crc, zd.ml:1081-1086: exact across blocks for
    // CRC-32) on the slice's own queue: its second half writes the checksums into the results the block coder wrote,
    // and the pass -- a plain read of the source at memory speed -- runs beside the other slice's kernels, which wait
    // for instruction issue and LDS, not for memory.
    if (crc_op == CRC_CRC32 && slice_err == hipSuccess) {
      uint32_t *partials = (uint32_t *)ctx->crc_partials.p + lo * crc32_segs(max_src_len);
      hipError_t ce = crc32_segments_launch(ctx, d_src, RANGE_DEFLATE_SRC, dd, nullptr, m, 0, 0, max_src_len, partials);
      if (ce == hipSuccess)
        ce = crc32_finish_launch(ctx, RANGE_DEFLATE_SRC, dd, d_results + lo, m, 0, max_src_len, partials, nullptr);
      if (ce != hipSuccess) slice_err = ce;
    }
  };
  // ZIPC_HIP_SLICES > 1: the group goes out in slices on queues of their own (ctx.h).  Measured on the
  // three shapes of tools/exp_wall.py and NOT the default: the kernels are each near their issue bound and
  // share the chip by workgroup, so slices changed the time by -4 .. +3 % (2 slices) or lost (more) -- until lz_chain
  // became four waves per CU (round 4): two slices are the default now, api.hip batch_slices has the numbers.
  const size_t k = batch_slices(n);
  hipError_t e = hipSuccess;
  if (k > 1) {
    e = ctx->fork(k);
    if (e != hipSuccess) return e;
    for (size_t i = 0; i < k; i++) {
      ctx->use_slice_stream(i);
      slice(n * i / k, n * (i + 1) / k);
    }
    e = ctx->join(k);
  } else {
    slice(0, n);
  }
  if (e == hipSuccess) e = slice_err;
  if (e == hipSuccess) e = hipGetLastError();
  return e;
}

}  // namespace zd

#ifdef ZD_PARSE_PHASES
extern "C" int zipc_hip_debug_parse_counts(unsigned long long *out8, int reset) {  // (the same entry point: tools/exp_wall.py PARSE_COUNTS=1)
  unsigned long long host[8] = {};
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(zd::zd_parse_phases), sizeof host) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(zd::zd_parse_phases), host, sizeof host) != hipSuccess) return 1;
  return 0;
}
#endif
#ifdef ZD_PARSE_COUNTS
extern "C" int zipc_hip_debug_parse_counts(unsigned long long *out8, int reset) {
  unsigned long long host[8] = {};
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(zd::zd_parse_counts), sizeof host) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(zd::zd_parse_counts), host, sizeof host) != hipSuccess) return 1;
  return 0;
}
#endif
#ifdef ZD_MATCH_COUNTS
extern "C" int zipc_hip_debug_match_counts(unsigned long long *out16, int reset) {
  unsigned long long host[16] = {};
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(zd::zd_match_counts), sizeof host) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(zd::zd_match_counts), host, sizeof host) != hipSuccess) return 1;
  return 0;
}
#endif
#ifdef ZD_MATCH_PHASES
extern "C" int zipc_hip_debug_match_phases(unsigned long long *out8, int reset) {
  static unsigned long long host[zd::ZD_PH_SLOTS * 8];
  if (out8) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(zd::zd_match_phases), sizeof host) != hipSuccess) return 1;
    for (int k = 0; k < 8; k++) {
      out8[k] = 0;
      for (int sl = 0; sl < zd::ZD_PH_SLOTS; sl++) out8[k] += host[sl * 8 + k];
    }
  }
  if (reset) {
    for (auto &h : host) h = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(zd::zd_match_phases), host, sizeof host) != hipSuccess) return 1;
  }
  return 0;
}
#endif
