// lz_tile.hip -- EXPERIMENT (round 4), not part of the library since round 5: match search and lazy parse of a batch of
// streams in ONE kernel, a workgroup per stream.  Exact, and 1.6 x slower than the two kernels it replaces (DESIGN.md
// section 6, round 4's record in profiles/HISTORY.md).  It shipped behind ZIPC_HIP_TILE=1 with a hand-over of the streams
// it refuses (S.punt / S.n_punt, a word per stream in the scratch; `punted_only` arguments of lz_match_window_kernel and
// lz_parse_kernel); round 5 took the file, the switch and the hand-over out of the library.  To build it again: put the
// file beside deflate.hip, add it to the Makefile, give DeflateScratch its punt / n_punt words back and launch it in front
// of lz_match (git history: the commit before this file moved).
//
// Round 1-3's pipeline ran find_backref (zd.ml:1176-1201) for every position in lz_match_window_kernel, wrote both
// answers (8 bytes per position) to memory and read them back in lz_parse_kernel, one wave per stream
// (Lz77.compress, zd.ml:1203-1244).  Here a workgroup of 16 waves walks its stream in tiles of 8 Ki positions whose
// 32 KiB window (source and chain links) sits in LDS, and the answers never leave the CU:
//
//   search   step 1, dense: lane = position.  Every position's FIRST candidate is compared in straight-line code with
//            all 64 lanes busy (link, 8 bytes, count, the candidate's own link); a position whose chain ends there --
//            about two in three on the benchmark's symbols -- has its answer.  The others are marked in a bitmap with
//            what the first step found.
//            step 2, pool: the marked positions of the whole tile are ONE pool; a wave draws a word of the bitmap at a
//            time and its run slots (deflate_lane.h match_run_step_to) go on from the second candidate.
//            The best of the first K candidates goes to LDS (4 bytes per position); the best of the first K/4 differs
//            from it for few positions only (a chain of more than K/4 candidates whose best comes late): those are
//            flagged and their second answer goes to memory (hi_exc), read back by the few parse steps that want it.
//   parse    the macro step of every position (deflate_lane.h lz_macro_position) from the answers in LDS; per block of
//            64 positions pointer doubling gives, for EVERY lane t, where a path that enters the block at t leaves it
//            (an exit table: one register per block).  With those, the path through a wave's 8 blocks from any entry
//            is 8 v_readlane, so the waves agree on their entries by iterating "my exit for the entry the wave before me
//            has now" until nothing changes (paths from different entries merge within a few symbols: two turns); then
//            every wave marks the visited positions of its blocks from its true entry, counts symbols, and -- behind a
//            scan over the waves' counts -- writes them.
//   The search runs TL_LOOK positions ahead of the parse (the lazy parse looks at the positions that follow a match);
//   a chain of more than TL_LOOK ever longer matches, a stream of less than 4 bytes: the stream is LEFT (S.punt) to the
//   kernels of the old pipeline, which run behind this one for the streams so marked.
//
// Output as lz_parse_kernel's: S.syms, S.blocks, S.n_blocks.
#include <type_traits>

#include "../../../zipc_amd/csrc/deflate_pipeline.h"
#include "../../../zipc_amd/csrc/tuning.h"

namespace zd {

constexpr uint32_t TL_THREADS = 1024, TL_WAVES = 16;
constexpr uint32_t TL_T = 8192;                  // positions parsed per tile
constexpr uint32_t TL_LOOK = 64;                 // positions searched beyond them
constexpr uint32_t TL_WPOS = TL_T / TL_WAVES;    // a wave's positions of a tile
constexpr int TL_NB = (int)(TL_WPOS / 64);       // ... in blocks of 64
constexpr uint32_t TL_RES = TL_T + TL_LOOK;      // positions searched per tile
constexpr uint32_t TL_WORDS = TL_RES / 64;       // ... in words of the bitmap
constexpr uint32_t TL_SRC_AHEAD = TL_LOOK + 288; // source bytes behind the tile the search may touch: its last position's 258, an over-read of 11
constexpr uint32_t TL_SRC_BYTES = MAX_MATCH_DIST + TL_T + TL_SRC_AHEAD;
constexpr uint32_t TL_LINKS = MAX_MATCH_DIST + TL_T + TL_LOOK;
constexpr uint32_t TL_RING = 128;                // a wave's queue of drawn positions
// A tile's new source bytes go behind the MAX_MATCH_DIST + TL_SRC_AHEAD the window already holds, its new links behind
// MAX_MATCH_DIST + TL_LOOK: in pieces of 16 bytes, these columns of the slide (lz_tile_kernel, "to the next tile")
constexpr uint32_t NEW_SRC_COL = ((MAX_MATCH_DIST + TL_SRC_AHEAD) / 16u) % (TL_T / 16u);
constexpr uint32_t NEW_LNK_COL = ((MAX_MATCH_DIST + TL_LOOK) / 8u) % (TL_T / 8u);
static_assert(TL_NB == 8 && TL_RES % 64 == 0 && TL_SRC_BYTES % 16 == 0 && TL_LINKS % 8 == 0, "tile shape");
static_assert(MAX_MATCH_DIST % TL_T == 0, "the window slides by whole tiles");
static_assert(TL_T + MAX_MATCH_LEN + TL_LOOK < MIN_BLOCK_SRC, "at most one block cut per tile");

constexpr uint32_t RES_LO = 0x1FFFFFFu;          // dist << 9 | len
constexpr uint32_t RES_HI = 1u << 31;            // final: the best of the first K/4 differs, it is in hi_exc[p]
constexpr uint32_t RES_RESTART = 1u << 30;       // step 1 -> 2: search the position from its first candidate
constexpr uint32_t RES_RESUME = 1u << 29;        // step 1 -> 2: candidate 1 is done, bits 0-24 are what it gave
constexpr uint32_t NO_CUT = 0xFFFFFFFFu;

__device__ __forceinline__ void tl_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t tl_lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t tl_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

#ifdef ZD_TILE_PHASES  // timing-only build (tools/exp_tile_phases.py): shader-clock ticks of wave 0 per phase, summed over tiles
constexpr int TL_PH = 10, TL_PH_SLOTS = 256;
__device__ unsigned long long zd_tile_phases[TL_PH_SLOTS * TL_PH];
#define TL_STAMP(k) do { if (tid == 0) { const unsigned long long _t = __builtin_readcyclecounter(); atomicAdd(&zd_tile_phases[(blockIdx.x % TL_PH_SLOTS) * TL_PH + (k)], _t - ph_t); ph_t = _t; } } while (0)
#else
#define TL_STAMP(k) do {} while (0)
#endif

// Step 2 of the search: the tile's marked positions as one pool behind *pool_next (words of the bitmap).
template <int NP>
__device__ __forceinline__ void tile_match_pool(const uint8_t *ws, const uint16_t *wp, uint32_t *res, uint32_t *hi_exc,
                                                const unsigned long long *bitmap, uint16_t *ring, uint32_t *pool_next,
                                                uint32_t n_words, uint32_t t0, uint32_t len, uint32_t K, uint32_t Kq,
                                                uint32_t lane) {
  MatchRun r[NP];
  uint32_t head = 0, tail = 0;  // of the wave's queue (wave-uniform, counted on without wrapping)
  bool empty = false;
  auto refill = [&]() {  // the marked positions of the pool's next word join the queue
    uint32_t i = 0;
    if (lane == 0) i = atomicAdd(pool_next, 1u);
    i = tl_uni(i);
    if (i >= n_words) { empty = true; return; }
    const unsigned long long mv = bitmap[i];
    const unsigned long long m = (unsigned long long)tl_uni((uint32_t)mv) | ((unsigned long long)tl_uni((uint32_t)(mv >> 32)) << 32);
    if ((m >> lane) & 1ull) ring[(tail + tl_lanes_below(m)) & (TL_RING - 1u)] = (uint16_t)(i * 64u + lane);
    tail += (uint32_t)__builtin_popcountll(m);
    tl_wave_sync();
  };
  // a run slot takes the queue's entry `slot` (take) or goes idle
  auto start = [&](MatchRun &x, bool take, uint32_t slot) {
    const uint32_t idx = take ? (uint32_t)ring[slot & (TL_RING - 1u)] : 0u;
    const uint32_t rr = take ? res[idx] : 0u;
    const uint32_t p = t0 + idx;  // (an idle slot parks on the tile's first position: staged, never stored)
    const uint32_t resume = (rr & RES_RESUME) ? 1u : 0u;
    const uint32_t d1 = wp[p];
    x.alive = take ? 1u : 0u;
    x.p = p;
    x.q = resume ? p - d1 : p;
    x.best = resume ? (rr & RES_LO) : 0u;
    x.best_len = (x.best & 0x1FFu) ? (x.best & 0x1FFu) : (uint32_t)(MIN_MATCH_LEN - 1);
    x.steps = resume;
    x.snap = (resume && Kq == 1u) ? x.best : SNAP_NONE;
    x.maxlen = len - p < (uint32_t)MAX_MATCH_LEN ? len - p : (uint32_t)MAX_MATCH_LEN;
    x.pw = load_u64_words(ws, p);
    x.dn = wp[x.q];
  };
  auto sink = [&](uint32_t p, uint32_t best, uint32_t snap) {
    const bool diff = snap != best;
    res[p - t0] = best | (diff ? RES_HI : 0u);
    if (diff) hi_exc[p] = snap;
  };
  while (tail - head <= 64u && !empty) refill();
#pragma unroll
  for (int i = 0; i < NP; i++) {
    const uint32_t avail = tail - head;
    start(r[i], lane < avail, head + lane);
    head += avail < 64u ? avail : 64u;
  }
  for (;;) {
    bool alive = false;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const bool fin = match_run_step_to<true>(r[i], ws, wp, K, Kq, sink);
      const unsigned long long fm = __builtin_amdgcn_ballot_w64(fin);
      if (fm) {  // wave-uniform
        const uint32_t taken = (uint32_t)__builtin_popcountll(fm);
        const uint32_t rank = tl_lanes_below(fm);
        while (tail - head < taken && !empty) refill();
        const uint32_t avail = tail - head;
        if (fin) start(r[i], rank < avail, head + rank);
        head += taken < avail ? taken : avail;
      }
      alive |= r[i].alive != 0;
    }
    if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
  }
}

__global__ __launch_bounds__(TL_THREADS) void lz_tile_kernel(const uint8_t *__restrict__ src_arena,
                                                             const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                             int K, int Kq, int good_match, int force_punt) {
  __shared__ __attribute__((aligned(16))) uint8_t win_src[TL_SRC_BYTES];
  __shared__ __attribute__((aligned(16))) uint16_t win_prev[TL_LINKS];
  __shared__ uint32_t res[TL_RES];
  __shared__ unsigned long long bitmap[TL_WORDS + 1];
  __shared__ uint16_t rings[TL_WAVES * TL_RING];
  __shared__ uint32_t X[TL_WAVES + 1];      // X[w]: where the path enters wave w's positions (X[16]: leaves the tile)
  __shared__ uint32_t wave_syms[TL_WAVES];  // symbols of wave w's visited positions
  __shared__ uint32_t pool_next, changed[3], punt_flag, cut_first, cut_blk_start, cut_sym_start;

  if (S.error[0]) return;
  const uint32_t stream = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = tl_uni(tid >> 6);
  const StreamDesc sd = descs[stream];
  if (sd.src_len > MAX_STREAM_LEN) {
    if (tid == 0) { S.n_blocks[stream] = 0; S.punt[stream] = 0; }
    return;
  }
  const uint32_t len = (uint32_t)sd.src_len;
  if (len < (uint32_t)MIN_MATCH_LEN || force_punt) {  // (nothing to search: the old parse writes the literals)
    if (tid == 0) { S.punt[stream] = 1; atomicAdd(S.n_punt, 1u); }
    return;
  }
  const uint8_t *s = src_arena + sd.src_off;
  const uint64_t base = S.pos_base[stream];
  const uint16_t *prev = S.prev + base;
  uint32_t *syms = S.syms + base;
  uint32_t *hi_exc = (uint32_t *)(S.match + base);  // a word per position inside the stream's 8-byte entries
  BlockDesc *blocks = S.blocks + S.blk_base[stream];
  const uint32_t max_pos = len - (uint32_t)MIN_MATCH_LEN;

  // ---- staging.  Tile at t0: source [w0, src_hi) and links [w0, link_hi), w0 = the window's first position.
  auto src_hi = [&](uint32_t t0) -> uint32_t {
    const uint64_t want = (uint64_t)t0 + TL_T + TL_SRC_AHEAD;
    return want < len ? (uint32_t)want : len;
  };
  auto link_hi = [&](uint32_t t0) -> uint32_t {  // (the scratch is padded behind len: whole groups of 8 links)
    const uint64_t want = (uint64_t)t0 + TL_T + TL_LOOK, have = ((uint64_t)len + 7u) & ~7ull;
    return want < have ? (uint32_t)want : (uint32_t)have;
  };
  // global -> LDS, source bytes [a, b) and links [c, d) (a, c multiples of 16 / 8 from w0)
  auto load_ranges = [&](uint32_t w0, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t n16 = (b - a) >> 4;
    for (uint32_t i = tid; i < n16; i += TL_THREADS) *(u32x4 *)(win_src + (a - w0) + 16u * i) = load16_unaligned(s + a + 16u * i);
    const uint32_t tail = (b - a) & 15u;
    if (tid < tail) win_src[(a - w0) + 16u * n16 + tid] = s[a + 16u * n16 + tid];
    const uint32_t n8 = (d - c) >> 3;
    for (uint32_t i = tid; i < n8; i += TL_THREADS) *(u32x4 *)(win_prev + (c - w0) + 8u * i) = *(const u32x4 *)(prev + c + 8u * i);
  };

  uint32_t t0 = 0, w0 = 0;
  load_ranges(0, 0, src_hi(0), 0, link_hi(0));
  if (tid == 0) { changed[0] = 0; changed[1] = 0; changed[2] = 0; punt_flag = 0; }
  // what the path carries from tile to tile (workgroup-uniform)
  uint32_t nsym = 0, blk_start = 0, blk_sym_start = 0, nblk = 0, entry = 0;
  __syncthreads();
#ifdef ZD_TILE_PHASES
  unsigned long long ph_t = __builtin_readcyclecounter();
#endif

  for (;;) {
    const uint8_t *ws = win_src - w0;   // indexed by stream position
    const uint16_t *wp = win_prev - w0;
    const uint32_t seg0 = t0 + w * TL_WPOS;  // my wave's first position
    // positions searched: [t0, t0 + TL_RES) up to max_pos, in words of the bitmap
    const uint32_t n_words = t0 > max_pos ? 0u : ((max_pos - t0) / 64u + 1u < TL_WORDS ? (max_pos - t0) / 64u + 1u : TL_WORDS);
    // (the tile's own words: everybody read the last tile's behind its closing barriers, and the first use of
    // these is behind the barrier below)
    if (tid == 0) { pool_next = 0; cut_first = NO_CUT; X[0] = entry; }

    // ---- search, step 1: every position's first candidate
    {
      auto first_step = [&](uint32_t chunk) {
        const uint32_t idx = chunk * 64u + lane, p = t0 + idx;
        const bool valid = p <= max_pos;
        const uint32_t d1 = valid ? (uint32_t)wp[p] : 0u;
        const uint32_t q = p - d1;
        const uint64_t x = load_u64_words(ws, q) ^ load_u64_words(ws, p);
        const uint32_t d2 = wp[q];
        const uint32_t maxlen = len - p < (uint32_t)MAX_MATCH_LEN ? len - p : (uint32_t)MAX_MATCH_LEN;  // (valid positions)
        const uint32_t l = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
        const bool hit = d1 != 0;
        const bool far = hit && (x == 0 || maxlen < 8u);  // the long compare, the stream's last bytes: the run slots' business
        const uint32_t best = (hit && l > (uint32_t)(MIN_MATCH_LEN - 1)) ? ((d1 << 9) | l) : 0u;
        const bool more = hit && d2 != 0 && d1 + d2 <= (uint32_t)MAX_MATCH_DIST;  // zd.ml:1185-1187 (1 < K; l < 8 <= maxlen)
        res[idx] = far ? RES_RESTART : more ? (best | RES_RESUME) : best;
        const unsigned long long um = __builtin_amdgcn_ballot_w64(far || more);
        if (lane == 0) bitmap[chunk] = um;
      };
      // (4 chunks side by side, twice: all 8 at once want more registers than a wave of this kernel has)
#pragma unroll 1
      for (uint32_t c0 = 0; c0 < (uint32_t)TL_NB; c0 += 4u) {
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++) first_step(w * (uint32_t)TL_NB + c0 + c);
      }
      if (w == 0) first_step(TL_WORDS - 1u);  // the look-ahead positions
    }
    lds_barrier();
    TL_STAMP(0);
    // ---- search, step 2: the marked positions
#ifndef ZD_TILE_NP
#define ZD_TILE_NP 2
#endif
    tile_match_pool<ZD_TILE_NP>(ws, wp, res, hi_exc, bitmap, rings + w * TL_RING, &pool_next, n_words, t0, len, (uint32_t)K,
                       (uint32_t)Kq, lane);
    TL_STAMP(1);
    __syncthreads();  // (hi_exc went to memory)
    TL_STAMP(2);

    // the next tile's new source bytes and links, requested now, stored behind the parse
    const bool last_tile = (uint64_t)t0 + TL_T >= len;
    const uint32_t t1 = t0 + TL_T;
    const uint32_t w1 = last_tile ? w0 : (t1 > (uint32_t)MAX_MATCH_DIST ? t1 - (uint32_t)MAX_MATCH_DIST : 0u);
    const uint32_t sa = src_hi(t0), sb = last_tile ? sa : src_hi(t1);
    const uint32_t la = link_hi(t0), lb = last_tile ? la : link_hi(t1);
    // ---- parse A: macro steps and exit tables of my 8 blocks.  The 8 blocks are worked on side by side, one level of
    // every chain of dependent LDS reads or shuffles at a time: a wave has 8 of them in flight where a block after
    // the other had one (16 waves per CU hide little).  What a block leaves for the phases behind the barriers is three
    // registers: bn = match | literals << 25; JA = J[0..4], 6 bits each (lane numbers); JB = J[5], J[6], exit - block << 12.
    uint32_t bn[TL_NB], JA[TL_NB], JB[TL_NB];
    bool my_punt = false;
    const uint32_t lane4 = lane * 4u;
    auto adv_of = [&](int b) -> uint32_t {  // my position's step in block b: 0 behind the stream's end
      const bool valid = seg0 + 64u * (uint32_t)b + lane < len;
      const uint32_t m = bn[b] & RES_LO;
      return valid ? (m ? (bn[b] >> 25) + (m & 0x1FFu) : 1u) : 0u;
    };
    auto J_of = [&](int b, int k) -> uint32_t {  // table k of block b as a ds_bpermute address
      return k < 5 ? ((JA[b] >> (6 * k)) & 63u) << 2 : ((JB[b] >> (6 * (k - 5))) & 63u) << 2;
    };
    {
      // the lazy chains of 4 blocks' positions at a time (all 8 want more registers than a wave of this kernel has)
      auto macro4 = [&](auto B0) {
        constexpr int b0 = decltype(B0)::value;
        uint32_t pend[4], lits[4];
        uint32_t ch = 0;  // bit i: my position of block b0 + i has a pending match that the next position may still beat
#pragma unroll
        for (int i = 0; i < 4; i++) {
          pend[i] = res[seg0 + 64u * (uint32_t)(b0 + i) + lane - t0] & RES_LO;  // (0 behind max_pos)
          lits[i] = 0;
          ch |= (pend[i] & 0x1FFu) != 0 ? (1u << i) : 0u;
        }
        for (uint32_t ahead = 1; __builtin_amdgcn_ballot_w64(ch != 0); ahead++) {
          if (ahead > TL_LOOK) { my_punt = true; break; }
          uint32_t rj[4];
#pragma unroll
          for (int i = 0; i < 4; i++) rj[i] = res[seg0 + 64u * (uint32_t)(b0 + i) + lane + ahead - t0];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const uint32_t j = seg0 + 64u * (uint32_t)(b0 + i) + lane + ahead;  // (a chaining lane has taken every position so far)
            const uint32_t chaining = (ch >> i) & 1u;
            const uint32_t pl = pend[i] & 0x1FFu;
            const uint32_t rem = len - j;
            const uint32_t maxlen = rem < (uint32_t)MAX_MATCH_LEN ? rem : (uint32_t)MAX_MATCH_LEN;
            uint32_t c = rj[i] & RES_LO;
            const bool want_hi = chaining != 0 && pl >= (uint32_t)good_match && (rj[i] & RES_HI) != 0;  // zd.ml:1182-1185: K/4 candidates
            if (__builtin_amdgcn_ballot_w64(want_hi)) {  // (rare)
              if (want_hi) c = __hip_atomic_load(hi_exc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t take = (chaining != 0 && j <= max_pos && pl < maxlen && (c & 0x1FFu) > pl) ? 1u : 0u;
            lits[i] += take;
            pend[i] = take ? c : pend[i];
            ch = take ? ch : (ch & ~(1u << i));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) bn[b0 + i] = (pend[i] & 0x1FFu) != 0 ? (pend[i] | (lits[i] << 25)) : 0u;
      };
      macro4(std::integral_constant<int, 0>{});
      macro4(std::integral_constant<int, 4>{});
      uint32_t Jk[TL_NB];
#pragma unroll
      for (int b = 0; b < TL_NB; b++) {
        const uint32_t j0 = lane + adv_of(b);
        Jk[b] = j0 < 64u ? j0 * 4u : lane4;  // a step that leaves the block points to itself
        JA[b] = Jk[b] >> 2;
        JB[b] = 0;
      }
#pragma unroll
      for (int k = 1; k < 7; k++) {
#pragma unroll
        for (int b = 0; b < TL_NB; b++) {
          Jk[b] = lane_value(Jk[b], Jk[b]);
          if (k < 5) JA[b] |= (Jk[b] >> 2) << (6 * k);
          else JB[b] |= (Jk[b] >> 2) << (6 * (k - 5));
        }
      }
      // 64 steps on: the path's last position inside the block, and where its step leads
#pragma unroll
      for (int b = 0; b < TL_NB; b++) JB[b] |= ((Jk[b] >> 2) + lane_value(Jk[b], adv_of(b))) << 12;
    }
    // the path through my blocks from entry e (wave-uniform): where it leaves them
    auto walk = [&](uint32_t e) -> uint32_t {
#pragma unroll
      for (int b = 0; b < TL_NB; b++) {
        const uint32_t bbase = seg0 + 64u * (uint32_t)b;
        if (e - bbase < 64u) e = bbase + ((uint32_t)__builtin_amdgcn_readlane((int)JB[b], (int)(e - bbase)) >> 12);
      }
      return e;
    };
    {
      const uint32_t e = walk(seg0);
      if (lane == 0) X[w + 1] = e;
      if (__builtin_amdgcn_ballot_w64(my_punt) && lane == 0) punt_flag = 1;
    }
    TL_STAMP(3);
    lds_barrier();
    if (punt_flag) {  // (workgroup-uniform)
      if (tid == 0) { S.punt[stream] = 1; atomicAdd(S.n_punt, 1u); }
      return;
    }
    // ---- parse B: the waves' entries
    for (uint32_t it = 0;; it++) {
      const uint32_t e_in = tl_uni(X[w]);
      const uint32_t e_out = walk(e_in);  // (an entry before my positions -- the stream ended -- passes through)
      if (lane == 0 && e_out != X[w + 1]) { X[w + 1] = e_out; changed[it % 3u] = 1; }
      lds_barrier();
      const uint32_t more = changed[it % 3u];
      if (tid == 0) changed[(it + 2u) % 3u] = 0;
      if (!more) break;
    }
    TL_STAMP(4);

    // ---- parse C1: visited positions from my true entry, their symbols counted (the blocks side by side again)
    uint32_t vis = 0, tot = 0;
    {
      uint32_t rel[TL_NB], v[TL_NB];  // rel: where the path enters block b (wave-uniform; >= 64: it does not)
      {
        uint32_t e = tl_uni(X[w]);
#pragma unroll
        for (int b = 0; b < TL_NB; b++) {
          const uint32_t bbase = seg0 + 64u * (uint32_t)b;
          rel[b] = e - bbase;
          if (rel[b] < 64u) e = bbase + ((uint32_t)__builtin_amdgcn_readlane((int)JB[b], (int)rel[b]) >> 12);
          v[b] = rel[b] < 64u ? rel[b] * 4u : 0u;  // (a block the path jumps over is searched like the others, for nothing:
        }                                          //  no branch between the shuffles of a level)
      }
      // the largest element of the path that is <= my lane: descending through the 2^k-step tables
#pragma unroll
      for (int k = 6; k >= 0; k--) {
        uint32_t y[TL_NB];
#pragma unroll
        for (int b = 0; b < TL_NB; b++) y[b] = lane_value(v[b], J_of(b, k));
#pragma unroll
        for (int b = 0; b < TL_NB; b++) v[b] = y[b] <= lane4 ? y[b] : v[b];
      }
      unsigned long long cuts = 0;  // the first visited position whose step ends past blk_start + 65534 (zd.ml:1118-1123)
#pragma unroll
      for (int b = 0; b < TL_NB; b++) {
        const uint32_t p = seg0 + 64u * (uint32_t)b + lane;
        const bool visited = rel[b] < 64u && p < len && v[b] == lane4;
        vis |= visited ? (1u << b) : 0u;
        tot += visited ? ((bn[b] & RES_LO) ? (bn[b] >> 25) + 1u : 1u) : 0u;
        const unsigned long long cb = __builtin_amdgcn_ballot_w64(visited && (p - blk_start) + adv_of(b) > (uint32_t)MAX_BLOCK_SRC_LEN);
        if (cb && cuts == 0) cuts = ((unsigned long long)(seg0 + 64u * (uint32_t)b + (uint32_t)__builtin_ctzll(cb)) << 1) | 1ull;
      }
      if (cuts && lane == 0) atomicMin(&cut_first, (uint32_t)(cuts >> 1));  // the block is cut at the tile's first such position
      tot = wave_sum(tot);
      if (lane == 0) wave_syms[w] = tot;
    }
    TL_STAMP(5);
    lds_barrier();
    TL_STAMP(6);
    // the next tile's new source bytes and links: requested now, stored behind the symbols
    u32x4 nsrc = {0, 0, 0, 0}, nlink = {0, 0, 0, 0};
    {
      // WHICH piece a thread takes: the one whose place in the window lies in the column the thread slides (below),
      // so that the piece it overwrites there is one it has moved itself
      const uint32_t n16 = (sb - sa) >> 4, n8 = (lb - la) >> 3;  // <= 512, <= 1024
      const uint32_t m16 = (tid - NEW_SRC_COL) & (TL_T / 16u - 1u), m8 = (tid - NEW_LNK_COL) & (TL_T / 8u - 1u);
      const uint32_t i16 = m16 < n16 ? m16 : (n16 ? n16 - 1u : 0u), i8 = m8 < n8 ? m8 : (n8 ? n8 - 1u : 0u);
      // (clamped, not predicated: no join behind which everything in flight is waited for)
      nsrc = load16_unaligned(n16 ? s + sa + 16u * i16 : (const uint8_t *)(prev + la));
      nlink = *(const u32x4 *)(prev + (n8 ? la + 8u * i8 : (la & ~7u) >= 8u ? (la & ~7u) - 8u : 0u));
    }
    // ---- parse C2: symbols out
    uint32_t tile_syms;
    {
      const uint32_t wv = wave_syms[lane & (TL_WAVES - 1u)];
      const uint32_t incl = wave_scan_incl(lane < TL_WAVES ? wv : 0u);
      tile_syms = (uint32_t)__builtin_amdgcn_readlane((int)incl, TL_WAVES - 1);
      uint32_t sym_at = nsym + (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)w) - tl_uni(wave_syms[w]);  // my wave's first symbol
      const uint32_t cut = cut_first;
      uint32_t lit[TL_NB];  // my positions' own bytes
#pragma unroll
      for (int b = 0; b < TL_NB; b++) lit[b] = ws[seg0 + 64u * (uint32_t)b + lane];
#pragma unroll
      for (int b = 0; b < TL_NB; b++) {
        const unsigned long long vm = __builtin_amdgcn_ballot_w64((vis >> b) & 1u);
        if (vm) {  // (wave-uniform)
          const uint32_t bbase = seg0 + 64u * (uint32_t)b;
          const uint32_t p = bbase + lane;
          const bool visited = (vis >> b) & 1u;
          const uint32_t brv = bn[b] & RES_LO;
          const uint32_t lits = brv ? bn[b] >> 25 : 0u;
          const uint32_t cnt = visited ? lits + 1u : 0u;
          const uint32_t incl_b = wave_scan_incl(cnt);
          const uint32_t first_rel = incl_b - cnt;
          uint32_t *tsyms = syms + sym_at;
          // lz_emit_position: a literal position writes its byte, a match position its deferral literals, then its match
          const bool lit_run = visited && lits != 0;
          for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(lit_run && k < lits); k++)
            if (lit_run && k < lits) tsyms[first_rel + k] = ws[p + k];
          if (visited) tsyms[first_rel + lits] = brv ? brv : lit[b];
          if (cut - bbase < 64u) {  // (wave-uniform) the block is cut at one of these positions
            const uint32_t rel = p - blk_start;
            const uint32_t first = sym_at + first_rel;
            uint32_t cutpos, symidx;
            if (brv == 0) { cutpos = p; symidx = first; }
            else if (rel + lits > (uint32_t)MAX_BLOCK_SRC_LEN) { const uint32_t i = (uint32_t)MAX_BLOCK_SRC_LEN - rel; cutpos = p + i; symidx = first + i; }
            else { cutpos = p + lits; symidx = first + lits; }
            cutpos = (uint32_t)__builtin_amdgcn_readlane((int)cutpos, (int)(cut - bbase));
            symidx = (uint32_t)__builtin_amdgcn_readlane((int)symidx, (int)(cut - bbase));
            if (lane == 0) {
              BlockDesc d;
              d.src_start = blk_start; d.src_len = cutpos - blk_start;
              d.sym_start = blk_sym_start; d.n_syms = symidx - blk_sym_start;
              blocks[nblk] = d;
              cut_blk_start = cutpos;
              cut_sym_start = symidx;
            }
          }
          sym_at += (uint32_t)__builtin_amdgcn_readlane((int)incl_b, 63);
        }
      }
    }
    TL_STAMP(7);
    // ---- to the next tile: the window slides, the new bytes and links go in
    const bool slide = !last_tile && w1 != w0;  // (by TL_T)
    lds_barrier();  // everybody is done with the tile: its window, res, X, wave_syms, cut_first
    nsym += tile_syms;
    entry = X[TL_WAVES];
    if (cut_first != NO_CUT) { nblk++; blk_start = cut_blk_start; blk_sym_start = cut_sym_start; }
    if (last_tile) break;
    if (slide) {
      // The window moves down by TL_T positions.  A thread moves the 16-byte pieces of ONE column -- pieces a slide apart
      // (links: 1024 pieces, source: 512) -- so what it overwrites is what it has read itself: no barrier between the
      // reads and the writes, and a handful of registers.
      constexpr uint32_t LNK_PIECES = TL_LINKS / 8u, LNK_STEP = TL_T / 8u;    // 5128 pieces, a slide = 1024
      constexpr uint32_t SRC_PIECES = TL_SRC_BYTES / 16u, SRC_STEP = TL_T / 16u;  // 2582 pieces, a slide = 512
      static_assert(LNK_STEP == TL_THREADS && SRC_STEP * 2u == TL_THREADS, "a column per thread");
      // (piece after piece, rolled: all of a column's pieces at once are 20 registers the kernel does not have here)
#pragma unroll 1
      for (uint32_t i = tid + LNK_STEP; i < LNK_PIECES; i += LNK_STEP) {
        const u32x4 r = *(const u32x4 *)(win_prev + 8u * i);
        *(u32x4 *)(win_prev + 8u * (i - LNK_STEP)) = r;
      }
      if (tid < SRC_STEP) {
#pragma unroll 1
        for (uint32_t i = tid + SRC_STEP; i < SRC_PIECES; i += SRC_STEP) {
          const u32x4 r = *(const u32x4 *)(win_src + 16u * i);
          *(u32x4 *)(win_src + 16u * (i - SRC_STEP)) = r;
        }
      }
    }
    {
      const uint32_t n16 = (sb - sa) >> 4, n8 = (lb - la) >> 3;
      const uint32_t m16 = (tid - NEW_SRC_COL) & (TL_T / 16u - 1u), m8 = (tid - NEW_LNK_COL) & (TL_T / 8u - 1u);
      if (tid < TL_T / 16u && m16 < n16) *(u32x4 *)(win_src + (sa - w1) + 16u * m16) = nsrc;
      const uint32_t tail = (sb - sa) & 15u;  // (the stream's last bytes: by the thread of their piece's column, like the pieces)
      if (tid < TL_T / 16u && m16 == n16)
        for (uint32_t k = 0; k < tail; k++) win_src[(sa - w1) + 16u * n16 + k] = s[sa + 16u * n16 + k];
      if (m8 < n8) *(u32x4 *)(win_prev + (la - w1) + 8u * m8) = nlink;
    }
    t0 = t1;
    w0 = w1;
    __syncthreads();
    TL_STAMP(8);
  }
  if (tid == 0) {
    BlockDesc d;  // the final block, always present (zd.ml:1216)
    d.src_start = blk_start; d.src_len = len - blk_start;
    d.sym_start = blk_sym_start; d.n_syms = nsym - blk_sym_start;
    blocks[nblk] = d;
    S.n_blocks[stream] = nblk + 1;
    S.punt[stream] = 0;
  }
}

hipError_t launch_lz_tile(zipc_hip_ctx *ctx, const uint8_t *d_src, const StreamDesc *d_descs, DeflateScratch S, size_t n,
                          int K, int good_match) {
  const int force_punt = tuning().tile_punt ? 1 : 0;  // tests: every stream is left to the kernels behind this one
  ZD_LAUNCH(ctx, "lz_tile", lz_tile_kernel, dim3((unsigned)n), dim3(TL_THREADS), 0, d_src, d_descs, S, K, K / 4, good_match,
            force_punt);
  return hipGetLastError();
}

}  // namespace zd

#ifdef ZD_TILE_PHASES
extern "C" int zipc_hip_debug_tile_phases(unsigned long long *out, int reset) {
  static unsigned long long host[zd::TL_PH_SLOTS * zd::TL_PH];
  if (out) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(zd::zd_tile_phases), sizeof host) != hipSuccess) return 1;
    for (int k = 0; k < zd::TL_PH; k++) {
      out[k] = 0;
      for (int sl = 0; sl < zd::TL_PH_SLOTS; sl++) out[k] += host[sl * zd::TL_PH + k];
    }
  }
  if (reset) {
    for (auto &h : host) h = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(zd::zd_tile_phases), host, sizeof host) != hipSuccess) return 1;
  }
  return 0;
}
#endif
