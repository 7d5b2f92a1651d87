// lz_match_queue.hip -- EXPERIMENT (round 5), not part of the library: find_backref (zd.ml:1176-1201) for every position of a
// batch of streams with the waves of a tile's workgroup stepping through CANDIDATES, 64 at a time, not through positions.
// Exact on every shape it ran on (sampled streams byte for byte against the oracle, round trips of the whole batches) and
// SLOWER than lz_match_window_kernel: C2 5.95 against 5.15 ms, 3-bit symbols 45.0 against 38.9, text 154 against 92.
// README.md beside this file has the numbers and what they say; to build it again, put the file back beside deflate.hip,
// add it to the Makefile's SRCS and call launch_lz_match_queue where deflate.hip launches lz_match_window_kernel.
//
// lz_match_window_kernel (deflate.hip) gives every lane run slots that walk a position's chain and take the next position
// when it ends: on the benchmark's symbols (1.3 candidates per position) 0.58 of the lanes of a step do something, every
// step carries the code that starts a position, and the answers leave by 8-byte stores scattered over the tile.  Here
//   * a DENSE unit takes the pool's next 64 consecutive positions, one per lane: link, 8 bytes of the position, 8 bytes
//     of its first candidate, that candidate's link -- the position-side reads are consecutive, and a position whose chain
//     ends with that candidate (two in three) leaves in one coalesced store;
//   * a position whose chain goes on leaves a two-word ENTRY in its wave's queue (a ring of 192 entries in LDS: the wave
//     pushes with a ballot and a lane count, nobody else touches it), and a QUEUE unit pops 64 entries, compares their
//     candidates and pushes those that still go on;
//   * an iteration steps TWO units whose reads are in flight together -- two queue units when 128 entries wait, a queue and a
//     dense unit from 64, two dense units below that; only when the tile's positions are all handed out do a wave's last
//     entries go through with fewer lanes.
// The number of candidates still allowed travels in the entry, so the K and K/4 tests (zd.ml:1182-1185) are compares on it:
// the best of the first K/4 is stored the moment the walk passes that candidate and goes on, and the final answer goes
// into the low word only then (queue_lane.h has the packings and queue_candidate, which tests/host_sim ran on the CPU
// against lz_match_position while the form was in the library).
//
// Window, staging and tile order are lz_match_window_kernel's: source and links of the 32 KiB before the tile, the tile
// and 272 bytes behind it in LDS, the next tile's requested into registers by each wave once it has nothing left to do.
#include "../../../zipc_amd/csrc/deflate_pipeline.h"
#include "../../../zipc_amd/csrc/tuning.h"
#include "queue_lane.h"

namespace zd {

constexpr uint32_t MQ_THREADS = 1024, MQ_WAVES = 16;
#ifndef ZD_MQ_TILE
#define ZD_MQ_TILE 13120
#endif
constexpr uint32_t MQ_TILE = ZD_MQ_TILE;  // positions per tile: what 160 KiB hold beside the queues; five tiles of a 64 KiB stream (205 dense units)
constexpr uint32_t MQ_QCAP = 192;     // entries of a wave's queue: an iteration of two units pops what it pushes at most, or starts below 64 (two
                                      // dense units) or below 128 (a queue and a dense unit) and pushes at most 128
constexpr uint32_t MQ_LINKS = MAX_MATCH_DIST + MQ_TILE;
constexpr uint32_t MQ_SRC_BYTES = MAX_MATCH_DIST + MQ_TILE + 272;  // + MAX_MATCH_LEN and the over-read of an aligned 16-byte read
constexpr uint32_t MQ_POOL_CHUNK = 256;  // positions a wave draws from the tile's pool at a time
#ifndef ZD_MQ_LOADS
#define ZD_MQ_LOADS LD_WORDS
#endif
constexpr int MQ_LD = ZD_MQ_LOADS;        // how the long compare reads 8 bytes of the window (queue_lane.h ld8)
static_assert(MQ_TILE % 64 == 0 && MQ_TILE % MQ_POOL_CHUNK % 64 == 0 && MQ_TILE <= QE_TILE_MAX && MQ_SRC_BYTES % 16 == 0 && MQ_LINKS % 8 == 0, "tile shape");
static_assert(MAX_MATCH_DIST + MQ_TILE + 272 < (1u << 16), "window-relative positions fit 16 bits");
static_assert(MQ_SRC_BYTES + 2 * MQ_LINKS + MQ_WAVES * MQ_QCAP * 8 + 64 <= 160 * 1024, "LDS of one CU");

struct MqTile {
  uint32_t t0, w0;            // first position of the tile, first staged position
  uint32_t src_end;           // one past the last staged source byte
  uint32_t n_src, n_links;    // staged in whole 16-byte units: source bytes, links
};
__device__ __forceinline__ MqTile mq_tile(uint32_t tile, uint32_t len) {
  MqTile g;
  g.t0 = tile * MQ_TILE;
  g.w0 = g.t0 > MAX_MATCH_DIST ? g.t0 - MAX_MATCH_DIST : 0;
  const uint64_t want = (uint64_t)g.t0 + MQ_TILE + 272;
  g.src_end = want < len ? (uint32_t)want : len;
  g.n_src = (g.src_end - g.w0) & ~15u;  // whole 16-byte units; the rest byte by byte
  const uint32_t link_end = (uint64_t)g.t0 + MQ_TILE < len ? g.t0 + MQ_TILE : len;
  g.n_links = (link_end - g.w0 + 7u) & ~7u;  // the scratch is padded past len
  return g;
}

// ring index below 2 * MQ_QCAP -> below MQ_QCAP (the capacity is no power of two: what the window leaves)
__device__ __forceinline__ uint32_t mq_wrap(uint32_t i) { const uint32_t j = i - MQ_QCAP; return i < j ? i : j; }

// One wave's share of a tile [pbeg, pend), any K (the form `Best ran: 12 bits of steps need the wide packing): one unit per
// iteration -- a queue unit whenever 64 entries wait, a dense unit over the chunks it draws from *pool_next otherwise --
// written on queue_lane.h's queue_candidate as it stands.  ws / wp: the window's source bytes and links, indexed by stream
// position.  out: the stream's table.
template <bool BIGK>
__device__ __forceinline__ void mq_walk(const uint8_t *ws, const uint16_t *wp, uint32_t len, uint32_t w0, uint32_t pbeg,
                                        uint32_t pend, uint32_t *pool_next, QEntry *queue, uint32_t lane, uint32_t K,
                                        uint32_t Kq, uint64_t *out) {
  uint32_t qh = 0, qn = 0;  // the ring's first entry and how many wait (wave-uniform)
  // my chunk of the pool is [next, cend); *pool_next counts the positions handed out, from 0 (relative to pbeg: it
  // overshoots the tile by a chunk per wave at the end, which must not wrap for a stream near the 4 GiB limit)
  auto fetch = [&]() -> uint32_t {
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(pool_next, MQ_POOL_CHUNK);
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    return c < pend - pbeg ? pbeg + c : pend;
  };
  uint32_t next = fetch();
  uint32_t cend = pend - next > MQ_POOL_CHUNK ? next + MQ_POOL_CHUNK : pend;
  uint64_t *tile_out = out + pbeg;  // (indexed by position - pbeg < 2^14: a 32-bit offset beside a uniform base)
  // what a step leaves behind: the first K/4's answer when the walk passes that candidate, the final answer, the entry
  auto settle = [&](bool active, bool more, uint32_t p, uint32_t dist, uint32_t steps, uint32_t blen, uint32_t bdist) {
    const uint32_t word = qe_word(blen, bdist);
    const uint32_t pt = p - pbeg;
    uint32_t *halves = (uint32_t *)tile_out;
    if (more && steps == Kq) halves[2u * pt + 1u] = word;
    if (active && !more) {
      if (steps > Kq) halves[2u * pt] = word;
      else tile_out[pt] = (uint64_t)word | ((uint64_t)word << 32);
    }
    const unsigned long long mm = __builtin_amdgcn_ballot_w64(more);
    if (mm) {  // wave-uniform
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
      if (more) queue[mq_wrap(mq_wrap(qh + qn) + rank)] = qe_pack<BIGK>(w0, pbeg, K, p, dist, steps, blen, bdist);
      qn += (uint32_t)__builtin_popcountll(mm);
    }
  };
  for (;;) {
    if (qn >= 64u || (next >= pend && qn != 0u)) {
      // ---- a queue unit: the 64 oldest entries (fewer only once the pool is empty)
      const uint32_t m = qn < 64u ? qn : 64u;
      const bool active = lane < m;
      const QEntry e = queue[mq_wrap(qh + (active ? lane : 0u))];
      qh = mq_wrap(qh + m);
      qn -= m;
      uint32_t p, dist, steps, blen, bdist;
      qe_unpack<BIGK>(e, w0, pbeg, K, p, dist, steps, blen, bdist);
      bool more = false;
      if (active) more = queue_candidate<MQ_LD>(ws, wp, len, p, ld8<MQ_LD>(ws, p), K, dist, steps, blen, bdist);
      settle(active, more, p, dist, steps, blen, bdist);
    } else if (next < pend) {
      // ---- a dense unit: the next 64 positions of my chunk, one per lane
      const uint32_t left = cend - next;
      const bool active = lane < left;
      const uint32_t p = next + (active ? lane : 0u);
      next = left > 64u ? next + 64u : cend;
      if (next == cend && cend < pend) {  // wave-uniform: the chunk is used up
        next = fetch();
        cend = pend - next > MQ_POOL_CHUNK ? next + MQ_POOL_CHUNK : pend;
      }
      uint32_t dist = wp[p], steps = 0, blen = MIN_MATCH_LEN - 1, bdist = 1;
      bool more = false;
      if (active && dist != 0u) more = queue_candidate<MQ_LD>(ws, wp, len, p, ld8<MQ_LD>(ws, p), K, dist, steps, blen, bdist);
      settle(active, more, p, dist, steps, blen, bdist);
    } else {
      break;
    }
  }
}

// ---- K <= 128 (`Fast, `Default): the same walk written for its instruction count and for two units per iteration.
// A unit's step is straight-line code on the packed entry: positions relative to the window (LDS addresses as they are), the
// best so far kept as the table word it will leave as, "candidates left" in the word's top bits so that K, K/4 and "the
// walk passed K/4" are one compare each, the stream's end folded into two compares, the ring's wrap a subtract and a
// minimum; the long compare (8 equal bytes: one candidate in 10^5 on the benchmark's symbols, one in four on text) is the
// one branch, taken by the wave when any lane wants it.
// MqUnit: 64 candidates, one per lane.  q, p: candidate and position relative to the window's first byte; b: the best so
// far as its table word, length field 3 for "none" | candidates the walk may look at AFTER this one << 25.
struct MqUnit {
  uint32_t q, p, b;
  bool pos, act;  // the lane holds a position / a candidate of it
};
enum : int { MQ_NONE = 0, MQ_QUEUE = 1, MQ_DENSE = 2 };

struct MqWalk {
  const uint8_t *wsrc;   // the window's source bytes and links (LDS), indexed by window-relative position
  const uint16_t *wlnk;
  QEntry *queue;
  uint32_t *pool_next;
  uint64_t *out_w;       // the table entry of the window's first position
  uint32_t lenrel;       // the stream's length from the window's first position
  uint32_t pbeg, pend;   // the tile's positions, window-relative
  uint32_t lane, K, Kq;
  uint32_t qh, qn;       // the ring's first entry, entries waiting (wave-uniform)
  uint32_t next, cend;   // my chunk of the pool (wave-uniform)

  __device__ __forceinline__ uint32_t fetch() {
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(pool_next, MQ_POOL_CHUNK);
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    return c < pend - pbeg ? pbeg + c : pend;
  }
  __device__ __forceinline__ void start() {
    qh = 0; qn = 0;
    next = fetch();
    cend = pend - next > MQ_POOL_CHUNK ? next + MQ_POOL_CHUNK : pend;
  }
  // a unit's candidates: the 64 oldest entries of the queue (fewer only when fewer wait) ...
  __device__ __forceinline__ void take_queue(MqUnit &u) {
    const uint32_t m = qn < 64u ? qn : 64u;
    u.act = lane < m;
    u.pos = u.act;
    const QEntry e = queue[mq_wrap(qh + (u.act ? lane : 0u))];
    qh = mq_wrap(qh + m);
    qn -= m;
    u.q = e.a & 0xFFFFu;
    u.p = e.a >> 16;
    u.b = e.b;
  }
  // ... or the first candidates of my chunk's next 64 positions
  __device__ __forceinline__ void take_dense(MqUnit &u) {
    const uint32_t left = cend - next;
    u.pos = lane < left;
    u.p = next + (u.pos ? lane : 0u);
    next = left > 64u ? next + 64u : cend;
    if (next == cend && cend < pend) {  // wave-uniform: the chunk is used up
      next = fetch();
      cend = pend - next > MQ_POOL_CHUNK ? next + MQ_POOL_CHUNK : pend;
    }
    const uint32_t d = wlnk[u.p];
    u.act = u.pos && d != 0u;
    u.q = u.p - d;  // (no candidate: the position itself, a valid address)
    u.b = 3u | ((K - 1u) << 25);
  }
  template <int KIND>
  __device__ __forceinline__ void take(MqUnit &u) {
    if (KIND == MQ_QUEUE) take_queue(u);
    else if (KIND == MQ_DENSE) take_dense(u);
    else { u.q = 0; u.p = 0; u.b = 0; u.pos = false; u.act = false; }
  }
  // the unit's reads: 8 bytes of the candidate, 8 of the position (three aligned words each), the candidate's link
  struct Words { uint32_t c0, c1, c2, w0, w1, w2, dn; };
  __device__ __forceinline__ Words read(const MqUnit &u) {
    Words r;
    const uint32_t *cq = (const uint32_t *)(wsrc + (u.q & ~3u)), *cp = (const uint32_t *)(wsrc + (u.p & ~3u));
    r.c0 = cq[0]; r.c1 = cq[1]; r.c2 = cq[2];
    r.w0 = cp[0]; r.w1 = cp[1]; r.w2 = cp[2];
    r.dn = wlnk[u.q];
    return r;
  }
  // the candidate against the position: common bytes of the first 8, cut at the stream's end.  stop: the candidate reaches
  // the longest match the position can have, nothing later can be longer (zd.ml:1194); lng: all 8 agree and there is more
  __device__ __forceinline__ uint32_t compare(const MqUnit &u, const Words &r, bool &stop, bool &lng) {
    const uint32_t xl = __builtin_amdgcn_alignbyte(r.c1, r.c0, u.q) ^ __builtin_amdgcn_alignbyte(r.w1, r.w0, u.p);
    const uint32_t xh = __builtin_amdgcn_alignbyte(r.c2, r.c1, u.q) ^ __builtin_amdgcn_alignbyte(r.w2, r.w1, u.p);
    const uint32_t fl = (uint32_t)__ffs((int)xl) - 1u, fh = (uint32_t)__ffs((int)xh) - 1u;  // first set bit, ~0 for none
    const uint32_t fh32 = (fh < 32u ? fh : 32u) + 32u;
    uint32_t l = (fl < fh32 ? fl : fh32) >> 3;
    const uint32_t rest = lenrel - u.p;
    l = l < rest ? l : rest;
    stop = l == rest;
    lng = u.act && l == 8u && !stop;
    return l;
  }
  // the long compare of the lanes that want it (queue_lane.h queue_candidate has the same lines)
  __device__ __forceinline__ void compare_long(const MqUnit &u, bool lng, uint32_t &l, bool &stop) {
    if (!lng) return;
    const uint32_t blen = u.b & 0x1FFu;
    const uint32_t rest = lenrel - u.p, maxlen = rest < (uint32_t)MAX_MATCH_LEN ? rest : (uint32_t)MAX_MATCH_LEN;
    bool cmp = true;
    if (blen >= 8u) {  // a candidate whose 8 bytes that END at offset blen differ cannot beat blen: its length is not needed
      const uint32_t toff = blen - 7u;
      cmp = ld8<MQ_LD>(wsrc, u.q + toff) == ld8<MQ_LD>(wsrc, u.p + toff);
    }
    if (cmp) {
      l = maxlen;
      for (uint32_t i = 8; i < maxlen; i += 8) {
        const uint64_t y = ld8<MQ_LD>(wsrc, u.q + i) ^ ld8<MQ_LD>(wsrc, u.p + i);
        if (y) {
          const uint32_t at = i + (uint32_t)(__builtin_ctzll(y) >> 3);
          l = at < maxlen ? at : maxlen;
          break;
        }
      }
      stop = l == maxlen;
    }
  }
  // what the step leaves behind: the better match in b, the answers of positions whose walk ends, the first K/4's answer
  // of walks that pass that candidate, the entries of walks that go on
  __device__ __forceinline__ void settle(const MqUnit &u, uint32_t dn, uint32_t l, bool stop) {
    const uint32_t dist = u.p - u.q;
    const bool better = u.act && l > (u.b & 0x1FFu);
    const uint32_t b = better ? ((u.b & 0xFE000000u) | (dist << 9) | l) : u.b;
    // the next candidate exists (its link is not 0) inside the window: dn - 1 < 32768 - dist, for dn = 0 never
    const bool more = u.act && (dn - 1u) < ((uint32_t)MAX_MATCH_DIST - dist) && !stop && b >= (1u << 25);
    const uint32_t word = (b & 0x1FFFFFFu) > 3u ? b & 0x1FFFFFFu : 0u;
    const uint32_t left = b >> 25;
    uint32_t *halves = (uint32_t *)out_w;
    if (more && left == K - Kq) halves[2u * u.p + 1u] = word;
    if (u.pos && !more) {
      if (left < K - Kq) halves[2u * u.p] = word;
      else out_w[u.p] = (uint64_t)word | ((uint64_t)word << 32);
    }
    const unsigned long long mm = __builtin_amdgcn_ballot_w64(more);
    if (mm) {  // wave-uniform
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
      if (more) {
        QEntry e;
        e.a = (u.p << 16) | (u.q - dn);
        e.b = b - (1u << 25);
        queue[mq_wrap(mq_wrap(qh + qn) + rank)] = e;
      }
      qn += (uint32_t)__builtin_popcountll(mm);
    }
  }
  // one iteration: two units (KB may be MQ_NONE), their reads in flight together
  template <int KA, int KB>
  __device__ __forceinline__ void step() {
    MqUnit a, b;
    take<KA>(a);
    take<KB>(b);
    const Words ra = read(a);
    Words rb = ra;
    if (KB != MQ_NONE) rb = read(b);
    bool stop_a, lng_a, stop_b = false, lng_b = false;
    uint32_t la = compare(a, ra, stop_a, lng_a), lb = 0;
    if (KB != MQ_NONE) lb = compare(b, rb, stop_b, lng_b);
    if (__builtin_amdgcn_ballot_w64(lng_a || lng_b)) {  // wave-uniform
      compare_long(a, lng_a, la, stop_a);
      if (KB != MQ_NONE) compare_long(b, lng_b, lb, stop_b);
    }
    settle(a, ra.dn, la, stop_a);
    if (KB != MQ_NONE) settle(b, rb.dn, lb, stop_b);
  }
  __device__ __forceinline__ void run() {
    start();
    for (;;) {
      // (wave-uniform all four, and the compiler is to know: it kept them in vector registers and made the choice below a divergent one)
      qh = (uint32_t)__builtin_amdgcn_readfirstlane((int)qh); qn = (uint32_t)__builtin_amdgcn_readfirstlane((int)qn);
      next = (uint32_t)__builtin_amdgcn_readfirstlane((int)next); cend = (uint32_t)__builtin_amdgcn_readfirstlane((int)cend);
      const bool have = next < pend;  // positions of the tile left for me
      if (qn >= 128u) step<MQ_QUEUE, MQ_QUEUE>();
      else if (qn >= 64u && have) step<MQ_QUEUE, MQ_DENSE>();
      else if (have) step<MQ_DENSE, MQ_DENSE>();
#ifdef ZD_MQ_NODRAIN  // timing only (wrong answers): what the tile costs without its last entries' walks
      else break;
#else
      else if (qn != 0u) step<MQ_QUEUE, MQ_NONE>();
      else break;
#endif
    }
  }
};

__global__ __launch_bounds__(MQ_THREADS) void lz_match_queue_kernel(const uint8_t *__restrict__ src_arena,
                                                                    const StreamDesc *__restrict__ descs, DeflateScratch S,
                                                                    uint32_t n_streams, uint32_t tiles_per_stream,
                                                                    uint32_t tiles_per_group, int K, int Kq, int lean) {
  __shared__ __attribute__((aligned(16))) uint8_t win_src[MQ_SRC_BYTES];
  __shared__ __attribute__((aligned(16))) uint16_t win_prev[MQ_LINKS];
  __shared__ __attribute__((aligned(16))) QEntry queues[MQ_WAVES * MQ_QCAP];
  __shared__ uint32_t pool_next;  // positions of the tile handed out to waves so far
  if (S.error[0]) return;
  // XCD-aware order as in lz_match_kernel: the groups of a stream re-read each other's windows, so they go to one XCD's L2
  const uint32_t nb = gridDim.x;
  const uint32_t per_xcd = (nb + 7) / 8;
  const uint32_t logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const uint32_t groups_per_stream = (tiles_per_stream + tiles_per_group - 1) / tiles_per_group;
  const uint32_t stream = logical / groups_per_stream;
  const uint32_t group = logical % groups_per_stream;
  if (stream >= n_streams) return;  // grid is padded to a multiple of 8
  const StreamDesc sd = descs[stream];
  if (sd.src_len < 4 || sd.src_len > MAX_STREAM_LEN) return;
  const uint32_t len = (uint32_t)sd.src_len;
  // tiles [tile, tile_end) of the stream; a tile exists when its first position can start a match
  const uint32_t stream_tiles = (uint32_t)(((uint64_t)len - 4) / MQ_TILE) + 1;
  uint32_t tile = group * tiles_per_group;
  if (tile >= stream_tiles) return;
  const uint32_t tile_end = tile + tiles_per_group < stream_tiles ? tile + tiles_per_group : stream_tiles;
  const uint64_t base = S.pos_base[stream];
  const uint8_t *s = src_arena + sd.src_off;
  const uint32_t tid = threadIdx.x;
  constexpr int SRC_ROUNDS = (MQ_SRC_BYTES / 16 + MQ_THREADS - 1) / MQ_THREADS;
  constexpr int LINK_ROUNDS = (MQ_LINKS / 8 + MQ_THREADS - 1) / MQ_THREADS;
  u32x4 vs[SRC_ROUNDS], vl[LINK_ROUNDS];
  // every thread issues all its 16-byte loads of a window at once (clamped indices: no branch around a load, at whose
  // join the compiler would wait for everything in flight) ...
  auto issue = [&](uint32_t w0, uint32_t n_src, uint32_t n_links) {
    const uint16_t *pv = S.prev + base + w0;
    const uint32_t last_src = n_src ? n_src - 16u : 0u;
    const uint8_t *sp = n_src ? s + w0 : (const uint8_t *)pv;  // (a stream shorter than one unit: scratch is read, nothing of it stored)
#pragma unroll
    for (int j = 0; j < SRC_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MQ_THREADS) * 16u;
      vs[j] = load16_unaligned(sp + (o < n_src ? o : last_src));
    }
#pragma unroll
    for (int j = 0; j < LINK_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MQ_THREADS) * 8u;
      vl[j] = *(const u32x4 *)(pv + (o < n_links ? o : n_links - 8u));
    }
  };
  // ... and stores them to the window later
  auto store = [&](const MqTile &g) {
#pragma unroll
    for (int j = 0; j < SRC_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MQ_THREADS) * 16u;
      if (o < g.n_src) *(u32x4 *)(win_src + o) = vs[j];
    }
#pragma unroll
    for (int j = 0; j < LINK_ROUNDS; j++) {
      const uint32_t o = (tid + (uint32_t)j * MQ_THREADS) * 8u;
      if (o < g.n_links) *(u32x4 *)(win_prev + o) = vl[j];
    }
    if (tid < ((g.src_end - g.w0) & 15u)) win_src[g.n_src + tid] = s[g.w0 + g.n_src + tid];
  };
  MqTile g = mq_tile(tile, len);
  issue(g.w0, g.n_src, g.n_links);
  store(g);
  // (a queue unit's idle lanes read an entry of their ring whatever it holds and the places it names: zeros name the window's first bytes)
  for (uint32_t i = tid; i < MQ_WAVES * MQ_QCAP; i += MQ_THREADS) { queues[i].a = 0; queues[i].b = 0; }
  if (tid == 0) pool_next = 0;
  __syncthreads();
  for (;;) {
    const bool has_next = tile + 1 < tile_end;  // uniform over the workgroup
    const MqTile gn = mq_tile(has_next ? tile + 1 : tile, len);
    const uint8_t *ws = win_src - g.w0;  // indexed by stream position
    const uint16_t *wp = win_prev - g.w0;
    // the parse reads up to PARSE_PAD entries behind the last position without a range test
    if ((uint64_t)g.t0 + MQ_TILE > (uint64_t)len - 4 && tid < PARSE_PAD) S.match[base + (len - 3) + tid] = 0;
    {
      const uint64_t tend64 = (uint64_t)g.t0 + MQ_TILE < (uint64_t)len - 3 ? (uint64_t)g.t0 + MQ_TILE : (uint64_t)len - 3;
      QEntry *q = queues + (tid >> 6) * MQ_QCAP;
      if (K > 128 || lean == 0) {
        if (K > 128) mq_walk<true>(ws, wp, len, g.w0, g.t0, (uint32_t)tend64, &pool_next, q, tid & 63u, (uint32_t)K, (uint32_t)Kq, S.match + base);
        else mq_walk<false>(ws, wp, len, g.w0, g.t0, (uint32_t)tend64, &pool_next, q, tid & 63u, (uint32_t)K, (uint32_t)Kq, S.match + base);
      } else {
        MqWalk W;
        W.wsrc = win_src; W.wlnk = win_prev; W.queue = q; W.pool_next = &pool_next;
        W.out_w = S.match + base + g.w0;
        W.lenrel = len - g.w0; W.pbeg = g.t0 - g.w0; W.pend = (uint32_t)tend64 - g.w0;
        W.lane = tid & 63u; W.K = (uint32_t)K; W.Kq = (uint32_t)Kq;
        W.run();
      }
    }
    if (!has_next) break;
    issue(gn.w0, gn.n_src, gn.n_links);  // in flight while the slower waves finish
    __syncthreads();  // every wave is done with this tile's window
    store(gn);
    if (tid == 0) pool_next = 0;
    __syncthreads();
    tile++;
    g = gn;
  }
}

// (ZIPC_HIP_MATCH_FORM=3 chose this kernel with its two-unit walk, 4 the one-unit walk at every level)
hipError_t launch_lz_match_queue(zipc_hip_ctx *ctx, const uint8_t *d_src, const StreamDesc *d_descs, DeflateScratch S,
                                 size_t n, size_t max_src_len, int K, int lean) {
  const size_t tps = (max_src_len + MQ_TILE - 1) / MQ_TILE;
  if (n * tps > 0x7FFFFFFFull) return hipErrorInvalidValue;
  // consecutive tiles of a stream per workgroup: as many as leave the grid >= 8192 workgroups (deflate.hip has the reasons)
  const long tpg_env = tuning().match_tiles_per_group;
  size_t tpg = tpg_env > 0 ? (size_t)tpg_env : n * tps / 8192;
  tpg = tpg < 1 ? 1 : (tpg > tps ? tps : tpg);
  const size_t gps = (tps + tpg - 1) / tpg;
  ZD_LAUNCH(ctx, "lz_match", lz_match_queue_kernel, dim3((unsigned)((n * gps + 7) / 8 * 8)), dim3(MQ_THREADS), 0, d_src, d_descs, S,
            (uint32_t)n, (uint32_t)tps, (uint32_t)tpg, K, K / 4, lean);
  return hipGetLastError();
}

}  // namespace zd
