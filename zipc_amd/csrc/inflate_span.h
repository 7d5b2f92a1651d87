// inflate_span.h -- the compressed block's symbols, decoded by all 64 lanes at full lane use.
//
// inflate.hip's wide turn speculates on 64 BIT OFFSETS and commits the ~8 that are real
// symbol starts: 13 % useful work, and the kernel is bound by instruction issue.  This file
// is the other decomposition of read_block_symbols (src/zipc_deflate.ml:593-616): the
// compressed bits ahead are cut into one REGION per lane (4 .. 32 granules of 256 bits), every
// lane walks its own region a symbol at a time -- 64 real symbols per wave step -- and the
// unknown region starts are settled by the fact that Huffman streams self-synchronise
// (measured on the configs' data: 50-150 bits on average, p99 under 1000):
//
//   regions  all the granules the input allows (64 x 18 at most) over the 64 lanes, cut to equal
//            numbers of SYMBOLS from a 32-step probe at 64 even places (symbols are not spread
//            evenly over the bits; any cut is a correct one).
//   phase A  lane i walks region i from its first bit (lane 0: the true position),
//            LENGTHS ONLY: bits and output bytes of each symbol, four straight-line steps and
//            then one look for a granule boundary.  At every 256-bit boundary it crosses it
//            records where the first symbol at or after the boundary starts (6 bits) and how
//            many bytes the granule before produced (10 bits): the INDEX, one u16 per granule
//            (in LDS, in the tile's place, until it is complete; then in the stream's scratch).
//            Then lane i walks on into region i+1, overwriting that region's entries with
//            its own, until it crosses a boundary at the very bit lane i+1 recorded: from
//            there on the two walks are the same walk (MERGED).  Lane 0 started on a real
//            symbol, so the chain lane 0 -> 1 -> ... makes every region's entries those of
//            the real symbol sequence, as far as every link merged inside its region.
//            Where a link did not, the span is cut there: a prefix is always exact.  A walk
//            that meets a symbol it must not decode (end of block, anything invalid) cannot
//            know whether it is on the real sequence yet: it notes the granule, skips a bit
//            and walks on; once the chain tells where the walk became real, its first such
//            note from there on is a real stop, and the span ends in front of that granule.
//   phase B  the verified granules in stream order, 64 at a time (one per lane, as many as
//            give at most 4 KiB of output): every lane decodes its granule fully -- the
//            literals go to their place in an LDS TILE of the output, a short match whose
//            source lies before the tile is requested from memory at once and lands four
//            steps later, any other match leaves its {distance, length} in the first 3 bytes
//            of its own hole and its bytes in a bitmap of unfilled bytes.
//   holes    listed (tile positions, stream order) and from there on work for any lane, 64 at
//            a time: sources wholly before the tile in one pipelined pass, the others 64 at a
//            time in stream order, a group going round -- filled when the bitmap shows every
//            source byte final -- until none of it is open.  The
//            tile leaves with coalesced 16-byte stores; literals never go to memory one by one.
//
// Codes longer than the tables' index bits -- rare per symbol, not per 64 symbols -- are found by
// bisection over per-length limits (canon_symbol, inflate_lane.h), not by the bit-by-bit walk.
//
// Everything that needs a decision of the reference -- end of block, errors, the last
// bytes of the input, the size limit -- stops the span IN FRONT of the symbol in question
// and is left to lane_one_symbol (inflate_lane.h), which owns those decisions.  A span
// that cannot run (little input left, highly compressible granules) leaves the block to
// the wide turns.  Written on wave.h's collectives, so tests/host_sim runs this very code.
#pragma once

#include "inflate_lane.h"
#include "wave.h"

namespace zd {

constexpr uint32_t SPAN_G = 256;          // granule: bits of input per index entry
constexpr uint32_t SPAN_K_MAX = 18;       // granules per lane and span at most (36 KiB of input per span)
constexpr uint32_t SPAN_PROBE_K_MIN = 8;  // spans of regions this long are cut by symbols (see the probe)
constexpr uint32_t SPAN_PROBE_BITS = 208;
constexpr int SPAN_PROBE_STEPS = 32;
constexpr int SPAN_PROBE_SMOOTH = 3;
constexpr uint32_t SPAN_K_MIN = 4;        // ... and at least: two walks can only merge at a granule boundary
constexpr uint32_t SPAN_MIN_LANES = 8;    // fewer regions than this: not worth a span
constexpr uint32_t SPAN_TILE = 4096;      // output bytes assembled in LDS at a time
constexpr uint32_t SPAN_OD_BIG = 1023;    // index value for "1023 output bytes or more": such a granule is not for a tile
constexpr uint32_t SPAN_RING = 8;         // input words a lane has waiting in LDS behind the 4 in its registers
constexpr uint32_t SPAN_TAIL_WORDS = 6;   // input words a span keeps clear of (its reads run ahead of its symbols)
constexpr uint32_t SPAN_LONG = 32;        // matches longer than this are copied by the whole wave
constexpr uint32_t SPAN_LIST_MAX = 1024;  // holes of a tile that are listed (in the idle input ring): more end the span
constexpr int SPAN_FLY = 4;               // far matches a lane has in flight: decode steps between request and arrival

// LDS of the span: byte offsets in the stream's block (inflate_lane.h has the map); the index
// lives in global scratch, 64 * SPAN_K_MAX entries per stream (2304 bytes)
constexpr uint32_t SPAN_RING_OFF = LDS_SPAN_RING_BYTE;   // u32[8 * 64], word-major
constexpr uint32_t SPAN_TILE_OFF = LDS_SPAN_TILE_BYTE;   // SPAN_TILE + 16 bytes
constexpr uint32_t SPAN_BITS_OFF = LDS_SPAN_BITS_BYTE;   // u32[128 + 2]: a bit per tile byte
constexpr uint32_t SPAN_IDX_ENTRIES = 64 * SPAN_K_MAX;
static_assert(SPAN_TILE + 16 == LDS_SPAN_TILE_BYTES && SPAN_RING * 64 * 4 <= LDS_WIDE_LIT * 4 && SPAN_IDX_ENTRIES * 2 <= SPAN_TILE, "inflate_lane.h's map");

enum : int { SPAN_NONE = 0, SPAN_AGAIN = 1, SPAN_LATER = 2, SPAN_OFF = 3, SPAN_POOR = 4 };
// d.span_fails: bits 0-3 the count that doubles the wait, SPAN_SMALL: the next span takes regions of SPAN_K_MIN granules
constexpr uint32_t SPAN_FAILS_MASK = 15u, SPAN_SMALL = 0x100u;
constexpr uint32_t SPAN_RETRY_WORDS = 8;  // input words (a granule) the wide turns take before a span is tried again
enum : uint32_t { WK_NONE = 0, WK_END = 1, WK_STOP = 2, WK_MERGED = 4, WK_NOMERGE = 5 };

// ---- one symbol, from the next 64 bits
struct SpanSym {
  uint32_t tot;     // bits
  uint32_t outlen;  // bytes: 1 (a literal), 2 (two literals), or the match length
  uint32_t val;     // literal byte, or the match distance
  uint32_t val2;    // the second literal
  bool is_lit, stop;
};
// The general form: codes longer than the tables' bits, end of block, anything invalid (their
// table entries are 0) take the canonical walk; what is not a plain literal or match stops.
ZD_HD SpanSym span_symbol_slow(uint32_t xlo, uint32_t xhi, const LaneLds &L, int lit_max_sym, int dist_max_sym) {
  SpanSym r;
  r.stop = false;
  r.tot = 0; r.outlen = 0; r.val = 0; r.val2 = 0;
  const uint32_t e = L.wide_lit((int)(xlo & ((1u << LIT_TBITS) - 1)));
  uint32_t b1, lv;
  if (e != 0) {
    r.is_lit = (int32_t)e < 0;
    b1 = (e >> 17) & 63u;
    lv = ((e >> 8) & 511u) + bit_field(xlo, e, (e >> 5) & 7u);
  } else {
    uint32_t n1;
    const int sym = canon_symbol(xlo, L, CANON_LIT, LDS_LIT_SYMS, n1);
    r.is_lit = true;
    if (sym < 0 || sym == LITLEN_EOB || sym > lit_max_sym || sym > LITLEN_SYM_MAX) {
      r.stop = true;
      return r;
    }
    if (sym < LITLEN_EOB) lv = (uint32_t)sym;
    else {
      r.is_lit = false;
      uint32_t base, extra;
      length_sym_value(sym, base, extra);
      lv = base + ((xlo >> n1) & ((1u << extra) - 1u));  // 15 + 5 bits at most
      n1 += extra;
    }
    b1 = n1;
  }
  if (r.is_lit) {
    r.tot = b1;
    r.outlen = 1;
    r.val = lv;
    return r;
  }
  const uint32_t x2 = funnel32(xhi, xlo, b1);  // b1 <= 20
  const uint32_t e2 = L.wide_dist((int)(x2 & ((1u << DIST_TBITS) - 1)));
  uint32_t dist, t2;
  if (e2 != 0) {
    dist = ((e2 >> 9) & 0xFFFFu) + bit_field(x2, e2, (e2 >> 5) & 15u);
    t2 = e2 >> 25;
  } else {
    uint32_t n2;
    const int dsym = canon_symbol(x2, L, CANON_DIST, LDS_DIST_SYMS, n2);
    if (dsym < 0 || dsym > dist_max_sym || dsym > DIST_SYM_MAX) {
      r.stop = true;
      return r;
    }
    uint32_t base, extra;
    dist_sym_value(dsym, base, extra);
    dist = base + ((x2 >> n2) & ((1u << extra) - 1u));  // 15 + 13 bits at most
    t2 = n2 + extra;
  }
  r.tot = b1 + t2;
  r.outlen = lv;
  r.val = dist;
  return r;
}
// The common case without a branch: both lookups always (a literal lane's distance lookup
// reads a valid, ignored entry, as in the wide turn).  Where an entry is 0 the general form
// takes over for that lane.  Every lane calls this (act: the lane's symbol is wanted).  Table entries are the wide turn's (inflate_lane.h) plus length symbols
// 268..285 (bit 29) and, in a literal's entry, the literal that follows it when one lookup
// resolves both: a step then takes two symbols.
template <bool FULL>
ZD_WV SpanSym span_symbol(bool act, uint32_t xlo, uint32_t xhi, const LaneLds &L, int lit_max_sym, int dist_max_sym) {
  const uint32_t e = L.wide_lit((int)(xlo & ((1u << LIT_TBITS) - 1)));
  const uint32_t b1 = (e >> 17) & 63u;
  const bool is_lit = (int32_t)e < 0;
  const uint32_t base = (e >> 8) & 511u;
  const uint32_t lv = base + bit_field(xlo, e, (e >> 5) & 7u);  // (a literal's entry has other things in these fields)
  const uint32_t x2 = funnel32(xhi, xlo, b1);
  const uint32_t e2 = L.wide_dist((int)(x2 & ((1u << DIST_TBITS) - 1)));
  SpanSym r;
  r.is_lit = is_lit;
  r.stop = false;
  r.tot = is_lit ? e & 15u : b1 + (e2 >> 25);
  r.outlen = is_lit ? 1u + ((e >> 4) & 1u) : lv;
  r.val = 0;
  r.val2 = 0;
  if (FULL) {
    r.val = is_lit ? base : ((e2 >> 9) & 0xFFFFu) + bit_field(x2, e2, (e2 >> 5) & 15u);
    r.val2 = (e >> 23) & 255u;
  }
  // (a divergent branch is skipped by the whole wave when no lane takes it: nearly always here)
  if (act && (e == 0u || (!is_lit && e2 == 0u))) r = span_symbol_slow(xlo, xhi, L, lit_max_sym, dist_max_sym);
  return r;
}

// ---- a lane's view of its input: the 4 words at its position in registers (a symbol is at most
// 48 bits), the next 8 waiting in LDS, word-major (word j of lane l at [j][l]: conflict free
// whatever word each lane is at), refilled 4 words at a time.  A refill requested at one refill
// point is written at the next, so its latency is a few symbols of the whole wave, not a wait;
// the peek of a symbol reads no memory at all.
struct SpanEnv {
  const uint8_t *src;  // the stream's input from the span's first word on
  uint32_t *ring;
  uint32_t max_word;   // words of input that may be loaded 16 bytes at a time
  int lane;
};
struct SpanReader {
  uint32_t w0, w1, w2, w3;  // input words cw .. cw + 3
  uint32_t cw;
  uint32_t wr;  // words below wr are in the registers or the ring
  wv::Quad pend;
  uint32_t has_pend;
};
// 4 words from word w on; past the loadable input the address is clamped (those words are never decoded)
ZD_WV wv::Quad span_load(const SpanEnv &E, uint32_t w) {
  const uint32_t c = w + 4u <= E.max_word ? w : E.max_word - 4u;
  return wv::load_quad(E.src + (uint64_t)c * 4u);
}
ZD_WV void span_ring_put(const SpanEnv &E, uint32_t w, const wv::Quad &q) {
  uint32_t *r = E.ring + (uint32_t)E.lane;
  r[((w + 0u) & (SPAN_RING - 1)) * 64u] = q.x;
  r[((w + 1u) & (SPAN_RING - 1)) * 64u] = q.y;
  r[((w + 2u) & (SPAN_RING - 1)) * 64u] = q.z;
  r[((w + 3u) & (SPAN_RING - 1)) * 64u] = q.w;
}
ZD_WV void span_reader_start(SpanReader &R, const SpanEnv &E, uint32_t p) {
  R.cw = p >> 5;
  const wv::Quad q0 = span_load(E, R.cw), q1 = span_load(E, R.cw + 4u), q2 = span_load(E, R.cw + 8u);
  R.w0 = q0.x; R.w1 = q0.y; R.w2 = q0.z; R.w3 = q0.w;
  span_ring_put(E, R.cw + 4u, q1);
  span_ring_put(E, R.cw + 8u, q2);
  R.wr = R.cw + 12u;
  R.has_pend = 0;
  R.pend = q0;
}
ZD_WV void span_reader_refill(SpanReader &R, const SpanEnv &E, bool active) {
  if (R.has_pend) {
    span_ring_put(E, R.wr, R.pend);
    R.wr += 4u;
    R.has_pend = 0;
  }
  if (active && R.wr - R.cw <= 8u) {  // 4 words or fewer wait in the ring
    R.pend = span_load(E, R.wr);
    R.has_pend = 1;
  }
}
// a step may cross two word boundaries
ZD_WV bool span_can_step(const SpanReader &R) { return R.cw + 6u <= R.wr; }
ZD_WV void span_peek(const SpanReader &R, uint32_t p, uint32_t &xlo, uint32_t &xhi) {
  xlo = funnel32(R.w1, R.w0, p);
  xhi = funnel32(R.w2, R.w1, p);
}
ZD_WV void span_advance(SpanReader &R, const SpanEnv &E, uint32_t p) {  // the position is now p; every lane calls this
  const uint32_t *r = E.ring + (uint32_t)E.lane;
  const uint32_t ncw = p >> 5;
  const bool s1 = ncw != R.cw;  // (nearly every step some lane crosses a word: no branch around this)
  const uint32_t nx = r[((R.cw + 4u) & (SPAN_RING - 1)) * 64u];
  R.w0 = s1 ? R.w1 : R.w0;
  R.w1 = s1 ? R.w2 : R.w1;
  R.w2 = s1 ? R.w3 : R.w2;
  R.w3 = s1 ? nx : R.w3;
  R.cw += s1 ? 1u : 0u;
  // a symbol of more than 32 bits: rare, and behind a WAVE-UNIFORM test -- behind a lane's own `if` the compiler kept both
  // versions of the words alive across it: three register moves in every step of every lane (round 6, from the assembly)
  if (wv::any(ncw != R.cw)) {
    const bool s2 = ncw != R.cw;
    const uint32_t nx2 = r[((R.cw + 4u) & (SPAN_RING - 1)) * 64u];
    R.w0 = s2 ? R.w1 : R.w0;
    R.w1 = s2 ? R.w2 : R.w1;
    R.w2 = s2 ? R.w3 : R.w2;
    R.w3 = s2 ? nx2 : R.w3;
    R.cw += s2 ? 1u : 0u;
  }
}

#ifdef SPAN_TRACE
static uint64_t span_trace_steps[8];
static uint64_t span_trace_seq;  // holes filled one after the other by the whole wave
static uint32_t span_lane_steps[64];
#endif
// ---- phase A: a lane's walk over a region, recording the index
struct SpanWalk {
  uint32_t p;         // bit position of the next symbol (from the span's first word)
  uint32_t nb;        // the next 256-bit boundary
  uint32_t k;         // granule of the region the walk is in
  uint32_t region_e;  // index entry of the region's granule 0
  uint32_t kr;        // granules in the region
  uint32_t pd, od;    // the current granule: where its first symbol starts, bytes so far
  uint32_t stops;     // own region: granules in which the walk met a stop (and skipped a bit)
  bool run;
  uint32_t kind, rk, rp;  // how the walk ended: WK_*, granule, bit position
};
ZD_WV uint16_t span_entry(uint32_t pd, uint32_t od) {
  return (uint16_t)(pd | ((od < SPAN_OD_BIG ? od : SPAN_OD_BIG) << 6));
}
ZD_WV void span_walk_end(SpanWalk &W, uint32_t kind, uint32_t k, uint32_t p) {
  W.kind = kind; W.rk = k; W.rp = p; W.run = false;
}
// The lengths of one symbol (or literal pair), no branch: both lookups always, as in span_symbol
// below.  slow: an entry was 0, the general form must look.
ZD_WV void span_lengths(uint32_t xlo, uint32_t xhi, const LaneLds &L, uint32_t &tot, uint32_t &outlen, bool &slow) {
  const uint32_t e = L.wide_lit((int)(xlo & ((1u << LIT_TBITS) - 1)));
  const uint32_t b1 = (e >> 17) & 63u;
  const bool is_lit = (int32_t)e < 0;
  const uint32_t lv = ((e >> 8) & 511u) + bit_field(xlo, e, (e >> 5) & 7u);
  const uint32_t x2 = funnel32(xhi, xlo, b1);
  const uint32_t e2 = L.wide_dist((int)(x2 & ((1u << DIST_TBITS) - 1)));
  tot = is_lit ? e & 15u : b1 + (e2 >> 25);
  outlen = is_lit ? 1u + ((e >> 4) & 1u) : lv;
  slow = e == 0u || (!is_lit && e2 == 0u);
}
// The walk, four steps at a time, by every lane.  Straight-line steps; what is rare for a lane (an
// entry of 0: long codes, stops) is a divergent branch the wave skips when no lane takes it.  A
// granule boundary is looked for once per four steps -- four symbols are at most 192 bits, so it
// is one boundary at most -- from the positions the steps left behind: some lane crosses one in
// nearly every step, and handling it there would be paid by all 64.  A stop met in the up to three
// steps a walk takes behind the granule it merges in is booked on the granule before and masked off
// with it: phase A's stop bookkeeping is NOT exact there, and entries behind such a stop may be
// accepted as verified.  What makes the result exact is phase B: it decodes every granule again from
// its real start with every check of lane_one_symbol's, and requires that each granule ends exactly
// where the next one's entry says it starts -- inside a tile and across tiles -- or refuses the tile.
template <bool STITCH>
ZD_WV void span_walk_loop(SpanWalk &W, SpanReader &R, const SpanEnv &E, const LaneLds &L, uint16_t *idx,
                          int lit_max_sym, int dist_max_sym) {
  for (;;) {
#ifdef SPAN_TRACE
    if (E.lane == 0) span_trace_steps[STITCH ? 1 : 0] += 4;
#endif
    uint32_t pq[4], oq[4];  // position and the granule's bytes after each step
#pragma unroll
    for (int u = 0; u < 4; u++) {
#ifdef SPAN_TRACE
      if (W.run && !span_can_step(R)) span_trace_steps[4]++;
      if (W.run) span_trace_steps[5]++;
      if (W.run && E.lane == 5) span_trace_steps[6]++;
      if (W.run && !STITCH) span_lane_steps[E.lane]++;
#endif
      const bool ok = W.run && span_can_step(R);
      uint32_t xlo, xhi, tot, outlen;
      bool slow;
      span_peek(R, W.p, xlo, xhi);
      span_lengths(xlo, xhi, L, tot, outlen, slow);
      if (ok && slow) {
        const SpanSym s = span_symbol_slow(xlo, xhi, L, lit_max_sym, dist_max_sym);
        tot = s.tot;
        outlen = s.outlen;
        if (s.stop) {
          tot = 0;
          outlen = 0;
          if (STITCH) {  // this walk is the real sequence: the span ends in front of this granule
            span_walk_end(W, WK_STOP, W.k, W.nb - SPAN_G + W.pd);
          } else {  // real or not is known later: note the granule, go on a bit further
            W.stops |= 1u << W.k;
            tot = 1;
          }
        }
      }
      W.p += ok ? tot : 0u;
      W.od += ok ? outlen : 0u;
      span_advance(R, E, W.p);
      pq[u] = W.p;
      oq[u] = W.od;
    }
    if (W.run && W.p >= W.nb) {  // into the next granule, with the first of the four symbols that crossed
      const uint32_t pc = pq[0] >= W.nb ? pq[0] : pq[1] >= W.nb ? pq[1] : pq[2] >= W.nb ? pq[2] : pq[3];
      const uint32_t oc = pq[0] >= W.nb ? oq[0] : pq[1] >= W.nb ? oq[1] : pq[2] >= W.nb ? oq[2] : oq[3];
      idx[W.region_e + W.k] = span_entry(W.pd, oc);  // (the symbol that crosses started in this granule)
      W.k++;
      const uint32_t npd = pc - W.nb;
      if (W.k == W.kr) span_walk_end(W, STITCH ? WK_NOMERGE : WK_END, W.k, pc);
      else if (STITCH && (idx[W.region_e + W.k] & 63u) == npd) span_walk_end(W, WK_MERGED, W.k, pc);
      // (a walk that ended here stands, reader and all, in the granule behind: where its stitch goes on)
      W.pd = npd;
      W.od -= oc;
      W.nb += SPAN_G;
    }
    span_reader_refill(R, E, W.run);
    if (!wv::any(W.run)) break;
    // lane 0 walks the real sequence from the span's first bit: when already its first granule is too
    // rich for a tile (runs of long matches: zeros, periods) the span will commit nothing: give up now
    if (!STITCH && wv::any(E.lane == 0 && W.k == 0u && W.od >= SPAN_OD_BIG)) {
      W.run = false;
      break;
    }
  }
}

// ---- phase B helpers
// the 3 bytes at tile + q (any alignment; the tile has 16 bytes of slack behind it)
ZD_WV uint32_t span_rec(const uint8_t *tile, uint32_t q) {
  const uint32_t *w = (const uint32_t *)(tile + (q & ~3u));
  return funnel32(w[1], w[0], (q & 3u) * 8u) & 0xFFFFFFu;
}
// 4 bytes at s + i (s any alignment; reads the two aligned words around them)
ZD_WV uint32_t span_rec4(const uint8_t *s, uint32_t i) {
  const uintptr_t a = (uintptr_t)(s + i);
  const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
  return funnel32(w[1], w[0], (uint32_t)(a & 3u) * 8u);
}
// A far match's bytes arrive: meta = (tile position + 1) | length << 16 (0: nothing in flight),
// 4 <= length <= 8; a = source bytes 0..3, b = source bytes length-4 .. length-1.  Two groups of
// four bytes, overlapping in the middle when length < 8: no test on the length.
ZD_WV void span_land(uint8_t *tile, uint32_t meta, uint32_t a, uint32_t b) {
  if (meta != 0u) {
    uint8_t *t = tile + ((meta & 0xFFFFu) - 1u);
    uint8_t *u = t + ((meta >> 16) - 4u);
    t[0] = (uint8_t)a; t[1] = (uint8_t)(a >> 8); t[2] = (uint8_t)(a >> 16); t[3] = (uint8_t)(a >> 24);
    u[0] = (uint8_t)b; u[1] = (uint8_t)(b >> 8); u[2] = (uint8_t)(b >> 16); u[3] = (uint8_t)(b >> 24);
  }
}
// The tile's bitmap has a bit per output byte: set while the byte belongs to a hole that is not
// filled yet.  (Literals and the far matches of the decode loop never set theirs.)
// set (SET) or clear the bits of [q, q + len)
template <bool SET>
ZD_WV void span_bits_mark(uint32_t *mbits, uint32_t q, uint32_t len) {
  uint32_t w = q >> 5, off = q & 31u;
  while (len != 0u) {
    const uint32_t n = 32u - off < len ? 32u - off : len;
    const uint32_t m = (n == 32u ? 0xFFFFFFFFu : (1u << n) - 1u) << off;
    if (SET) wv::lds_or(mbits + w, m);
    else wv::lds_and(mbits + w, ~m);
    len -= n;
    w++;
    off = 0;
  }
}
// set the bits of [q, q + len), 3 <= len: the decode loop's form, two words without a loop for
// what is shorter than 32
ZD_WV void span_bits_set(uint32_t *mbits, uint32_t q, uint32_t len) {
  if (len >= 32u) {
    span_bits_mark<true>(mbits, q, len);
    return;
  }
  const uint32_t off = q & 31u, over = off + len;
  uint32_t *w = mbits + (q >> 5);
  wv::lds_or(w, ((1u << len) - 1u) << off);
  if (over > 32u) wv::lds_or(w + 1, (1u << (over - 32u)) - 1u);
}
// clear the bits of [q, q + len), 3 <= len
ZD_WV void span_bits_clear(uint32_t *mbits, uint32_t q, uint32_t len) {
  if (len >= 32u) {
    span_bits_mark<false>(mbits, q, len);
    return;
  }
  const uint32_t off = q & 31u, over = off + len;
  uint32_t *w = mbits + (q >> 5);
  wv::lds_and(w, ~(((1u << len) - 1u) << off));
  if (over > 32u) wv::lds_and(w + 1, ~((1u << (over - 32u)) - 1u));
}
// is any bit of [a, b) set?  a < b
ZD_WV bool span_bits_any(const uint32_t *mbits, uint32_t a, uint32_t b) {
  uint32_t w = a >> 5;
  uint32_t m = mbits[w] & (0xFFFFFFFFu << (a & 31u));
  const uint32_t wl = (b - 1u) >> 5;
  while (w < wl) {
    if (m != 0u) return true;
    w++;
    m = mbits[w];
  }
  const uint32_t keep = (b & 31u) == 0u ? 0xFFFFFFFFu : (1u << (b & 31u)) - 1u;
  return (m & keep) != 0u;
}
// the first set bit in [from, to), or 0xFFFFFFFF: a lane's next open hole (it fills its holes in
// stream order, so the first unfilled byte at or after its cursor starts one)
ZD_WV uint32_t span_bits_first(const uint32_t *mbits, uint32_t from, uint32_t to) {
  if (from >= to) return 0xFFFFFFFFu;
  uint32_t w = from >> 5;
  uint32_t m = mbits[w] & (0xFFFFFFFFu << (from & 31u));
  while (m == 0u) {
    w++;
    if ((w << 5) >= to) return 0xFFFFFFFFu;
    m = mbits[w];
  }
  const uint32_t q = (w << 5) + (uint32_t)__builtin_ctz(m);
  return q < to ? q : 0xFFFFFFFFu;
}
ZD_WV uint32_t span_byte_at(const uint8_t *tile, const uint8_t *gbase, int s) {  // tile byte s, or the byte -s before it
  return s < 0 ? (uint32_t)gbase[s] : (uint32_t)tile[s];
}
// one hole by the whole wave (dp, dist, len the same in every lane): Buf.recopy zd.ml:63-75, byte i is
// the source's byte i mod dist
ZD_WV void span_fill_by_wave(uint8_t *tile, const uint8_t *gbase, uint32_t dp, uint32_t dist, uint32_t len, uint32_t ulane) {
  const int sp = (int)dp - (int)dist;
  if (dist >= len) {
    for (uint32_t i = ulane; i < len; i += 64u) tile[dp + i] = (uint8_t)span_byte_at(tile, gbase, sp + (int)i);
  } else {
    const uint32_t step = 64u % dist;
    uint32_t r = ulane % dist;
    for (uint32_t i = ulane; i < len; i += 64u) {
      tile[dp + i] = (uint8_t)span_byte_at(tile, gbase, sp + (int)r);
      r += step;
      if (r >= dist) r -= dist;
    }
  }
}
// The span.  d.phase == PH_SYMBOLS, nothing queued; returns SPAN_NONE when it did not run
// (nothing changed), else the stream position, out_pos and ring_wr are those after the
// symbols it committed: SPAN_AGAIN (more of the block may follow the same way), SPAN_LATER (it met a
// stretch it cannot take -- a granule too rich for a tile, walks that do not fall into step: runs of long
// matches, zeros, periods; the wide turns and match_run take that stretch, then a span is tried again) or SPAN_OFF
// (leave the rest of this block to the wide turns).
#ifdef ZD_INFLATE_PHASES  // timing-only build (tools/exp_inflate_phases.py): clocks per phase of the span
#define ZD_SPAN_PH(i) do { const uint64_t now_ = __builtin_readcyclecounter(); span_ph[i] += now_ - span_t; span_t = now_; } while (0)
#else
#define ZD_SPAN_PH(i) do {} while (0)
#endif
// MODE (inflate_lane.h IM_*): IM_DRY walks and checks the symbols and moves the position, touching neither `dst` nor a
// tile; IM_TOKEN stores the literals and, for every byte of a match, the output position it is a copy of in tok[]
// (inflate.hip: one stream by a wave per block; srcpos: SPAN_TILE u16 of LDS, the tile's sources while they are
// sorted out).
// bits_cap: a span walks no further than this many bits (a wave that takes a block over from a checkpoint on and only
// up to the next one: inflate.hip).  ck (IM_DRY): checkpoints of the block -- bit and output position of a tile's start,
// both known to be on the real sequence, every CK_GAP_BITS of input or so.
constexpr uint32_t CK_MAX = 31, CK_GAP_BITS = 24576;
struct SpanCk {
  uint32_t *lds;      // 2 * CK_MAX words: bit (from the block's header bit), output byte (from the block's first)
  uint32_t n, last;   // checkpoints taken, the last one's bit
  uint64_t hdr_bit;   // the block's header bit
  uint32_t out_base;  // the block's first output position
};
template <int MODE = IM_REAL>
ZD_WV int span_decode(InflateLane &d, const LaneLds &L, const uint8_t *src_stream, uint8_t *dst, uint16_t *idx, uint32_t *tok,
                      uint16_t *srcpos, uint32_t bits_cap, SpanCk *ck, int lane
#ifdef ZD_INFLATE_PHASES
                      , uint64_t *span_ph
#endif
) {
#ifdef ZD_INFLATE_PHASES
  uint64_t span_t = __builtin_readcyclecounter();
#endif
  // ---- geometry (wave-uniform: the stream's state is the same in every lane)
  const uint32_t in_word = wv::uni(d.in_word), base = wv::uni(d.boff), src_len = wv::uni(d.src_len);
  const uint32_t full_words = src_len >> 2;
  if (full_words < in_word + SPAN_TAIL_WORDS + 2u) return SPAN_NONE;
  const uint32_t max_word = full_words - in_word;
  uint32_t usable = (max_word - SPAN_TAIL_WORDS) * 32u - base;  // bits a span may walk
  if (usable > bits_cap) usable = bits_cap;
  uint32_t est = usable;
  const uint32_t prev_bits = wv::uni(d.prev_block_bits);
  if (prev_bits != 0u) {
    const uint32_t guess = prev_bits + (prev_bits >> 3) + SPAN_G;
    if (guess < est) est = guess;
  }
  uint32_t K = (est + 64u * SPAN_G - 1u) / (64u * SPAN_G);
  if (K < SPAN_K_MIN) K = SPAN_K_MIN;
  if (K > SPAN_K_MAX) K = SPAN_K_MAX;
  if (wv::uni(d.span_fails) & SPAN_SMALL) K = SPAN_K_MIN;  // behind a span whose walks did not fall into step: a cheap one (span_after)
  // the span: TG granules -- all the input allows, 64 regions of K at most -- over as many lanes as get
  // SPAN_K_MIN each; cut evenly, lane i starts at granule i * TG / n_lanes
  uint32_t TG = usable / SPAN_G;
  if (TG > 64u * K) TG = 64u * K;
  uint32_t n_lanes = TG / SPAN_K_MIN;
  if (n_lanes > 64u) n_lanes = 64u;
#ifdef SPAN_TRACE
  if (lane == 0) fprintf(stderr, "span geometry: usable %u K %u TG %u lanes %u\n", usable, K, TG, n_lanes);
#endif
  if (n_lanes < SPAN_MIN_LANES) return SPAN_NONE;

  SpanEnv E;
  E.src = src_stream + (uint64_t)in_word * 4u;
  E.ring = (uint32_t *)(L.x + SPAN_RING_OFF);
  E.max_word = max_word;
  E.lane = lane;
  uint8_t *tile = L.x + SPAN_TILE_OFF;
  uint16_t *lidx = (uint16_t *)tile;  // phase A's index: in the tile's place (SPAN_IDX_ENTRIES u16) while no tile is being made
  uint32_t *mbits = (uint32_t *)(L.x + SPAN_BITS_OFF);
  const int lit_max = (int)wv::uni((uint32_t)d.lit_max_sym), dist_max = (int)wv::uni((uint32_t)d.dist_max_sym);
  const uint32_t cap_min = wv::uni(d.cap_min), out_pos0 = wv::uni(d.out_pos), hard_cap = wv::uni(d.hard_cap);
  if (hard_cap < 8u) return SPAN_NONE;  // (phase B reads the output's first 8 bytes when it has nothing better to read)
  const uint32_t ulane = (uint32_t)lane;
  wv::fence_global();  // bytes the wide turns stored are read below as match sources

  // ---- the regions.  Symbols are not spread evenly over the bits (a stream's first kilobytes are
  // literals, matches come with the history), and phase A lasts as long as its busiest lane: a short
  // probe at 64 even places counts symbols per bit, and the regions are cut to equal symbols, not
  // equal bits.  Any cut is a correct one; the probe only decides how well the lanes are loaded.
  SpanWalk W;
  SpanReader R;
  const bool in_span = ulane < n_lanes;
  const uint32_t u0 = in_span ? ulane * TG / n_lanes : TG, u1 = in_span ? (ulane + 1u) * TG / n_lanes : TG;
  uint32_t g0 = u0, kr = u1 - u0;  // the lane's region: first granule, granules
  if (K >= SPAN_PROBE_K_MIN) {
    const uint32_t p0 = base + (in_span ? u0 : 0u) * SPAN_G;
    const uint32_t pe = p0 + SPAN_PROBE_BITS;  // (the words the reader starts with cover this: no refill)
    uint32_t p = p0, rho = 0;
    span_reader_start(R, E, p);
#pragma unroll 1
    for (int u = 0; u < SPAN_PROBE_STEPS; u++) {  // the same steps for every lane; what differs is the bits they take
      const bool act = in_span && p < pe;
      uint32_t xlo, xhi;
      span_peek(R, p, xlo, xhi);
      const SpanSym s = span_symbol<false>(act, xlo, xhi, L, lit_max, dist_max);
      p += act ? (s.stop ? 1u : s.tot) : 0u;
      rho += act ? 1u : 0u;
      span_advance(R, E, p);
    }
    rho = in_span ? rho * 4096u / (p - p0) : 0u;  // symbols per 256 bits, in 16ths
    // the density is a smooth thing, a probe's count a noisy one
    for (int round = 0; round < SPAN_PROBE_SMOOTH; round++) {
      const uint32_t a = wv::shfl(rho, ulane == 0u ? 0u : ulane - 1u);
      const uint32_t b = wv::shfl(rho, ulane + 1u < n_lanes ? ulane + 1u : n_lanes - 1u);
      rho = (a + 2u * rho + b) >> 2;
    }
    // no region more than about twice or less than half the even one
    const uint32_t avg = wv::readlane(wv::scan_incl(rho), 63u) / n_lanes;
    const uint32_t lo = (avg + 1u) >> 1, hi = avg * 2u;
    rho = !in_span ? 0u : rho < lo ? lo : rho > hi ? hi : rho;
    const uint32_t incl = wv::scan_incl(rho), before = incl - rho, total = wv::readlane(incl, 63u);
    // my region starts where the symbols before it are ulane / n_lanes of all: in probe j's stretch
    const uint32_t target = in_span ? ulane * total / n_lanes : total;
    uint32_t j = 0;
    for (uint32_t step = 32u; step != 0u; step >>= 1) {
      const uint32_t cand = j + step, c = wv::shfl(before, cand);
      if (cand < n_lanes && c <= target) j = cand;
    }
    const uint32_t bj = wv::shfl(before, j), rj = wv::shfl(rho, j), uj = wv::shfl(u0, j), kj = wv::shfl(u1 - u0, j);
    const uint32_t x = in_span ? uj + (target - bj) * kj / rj : TG;
    const uint32_t xn = wv::shfl(x, ulane + 1u);
    const uint32_t len = (ulane + 1u < n_lanes ? xn : TG) - x;
    if (!wv::any(in_span && (len < SPAN_K_MIN || len > 32u))) {  // (32: a region's stops are a mask)
      g0 = x;
      kr = len;
    }
  }
#ifdef SPAN_TRACE
  if (lane == 0) fprintf(stderr, "regions: ");
  for (int i = 0; i < 64; i++) { const uint32_t v = wv::readlane(kr, (uint32_t)i); if (lane == 0) fprintf(stderr, "%u ", v); }
  if (lane == 0) fprintf(stderr, "\n");
#endif

  ZD_SPAN_PH(2);
  // ---- phase A: every lane its own region
  // Where in its first granule a walk starts is free, and one place is better than the others when most of the block's
  // literals have ONE code length n (base64: 64 letters of 6 bits; ASCII in a fixed block: 8): symbols then start every n
  // bits from the span's first one until a symbol of another length comes by, and a walk that starts between two of
  // them never falls into step (the spans of such chunks verified a region or two; the wide turns took the rest at a
  // seventh of the span's rate per symbol).  So a walk starts on the first such bit at or behind its region's start, with n
  // the most numerous length of the literal / length code (two codes and more); what follows checks it like any start.
  // (tools/exp_inflate_fixed.py, 4096 copies each: 48 KiB of random bytes in base64 4.0 -> 2.7 ms, 64 KiB slices of
  // tests/golden/zlib_streams.json 4.4-5.5 -> 3.2-3.9, one 5.6 -> 6.2, the rest, text and symbols the same.  A chain still
  // breaks where a symbol of another length came by; trying again at once behind it -- small spans, no wait -- was
  // measured too: 6.2 -> 14 ms, a span costs its whole first pass whatever it verifies.)
  uint32_t hint_adj = 0;
  {
    const uint32_t c = ulane < 16u ? (uint32_t)L.u16(LDS_LIT_COUNTS, (int)ulane) : 0u;
    uint32_t best_len = 0, best_n = 1;
    for (uint32_t l = 1; l < 16u; l++) {
      const uint32_t n = wv::readlane(c, l);
      if (n > best_n) { best_n = n; best_len = l; }
    }
#ifndef ZD_SPAN_NO_HINT
    if (best_len != 0u && in_span) hint_adj = (best_len - (g0 * SPAN_G) % best_len) % best_len;
#endif
  }
  W.p = base + (in_span ? g0 : 0u) * SPAN_G;
  W.nb = W.p + SPAN_G;
  W.p += hint_adj;
  W.k = 0;
  W.region_e = g0;
  W.kr = kr;
  W.pd = hint_adj;
  W.od = 0;
  W.stops = 0;
  W.run = in_span;
  W.kind = WK_NONE; W.rk = 0; W.rp = 0;
  span_reader_start(R, E, W.p);
  span_walk_loop<false>(W, R, E, L, lidx, lit_max, dist_max);
  ZD_SPAN_PH(0);
  if (wv::any(lane == 0 && W.kind == WK_NONE)) {  // gave up (see the walk loop): nothing committed,
    d.ring_wr = d.in_word;                         // but the wide path's input ring was this walk's
    return SPAN_LATER;
  }
  const uint32_t m_p = W.rp, m_stops = W.stops;  // (every walk of a region ends at the region's end: WK_END)
  wv::sync();  // entries written by one lane are read by others below
  // ... and on into the next lane's, until the two walks are one
  {
    const bool stitch = ulane + 1u < n_lanes;
    const uint32_t g0n = wv::shfl(g0, ulane + 1u), krn = wv::shfl(kr, ulane + 1u);
    W.kind = WK_NONE; W.rk = 0; W.rp = 0;
    W.run = false;
    if (stitch) {
      W.region_e = g0n;
      W.kr = krn;
      W.k = 0;  // (position, reader, pd, od, nb: as the walk of the own region left them, in this granule)
      if ((lidx[W.region_e] & 63u) == W.pd) span_walk_end(W, WK_MERGED, 0, m_p);
      else W.run = true;
    }
  }
  span_walk_loop<true>(W, R, E, L, lidx, lit_max, dist_max);
  const uint32_t s_kind = W.kind, s_k = W.rk, s_p = W.rp;
  wv::sync();
  ZD_SPAN_PH(1);

  // ---- how far the chain from lane 0 holds
  uint32_t n_valid, p_end;
  bool end_stop = false;
  {
    // my own walk is the real sequence from granule mk on, if the lane before merged into it
    const uint32_t pk = wv::shfl(s_kind, ulane - 1u), pmk = wv::shfl(s_k, ulane - 1u);
    const bool real = lane == 0 || pk == WK_MERGED;
    const uint32_t mk = lane == 0 ? 0u : pmk;
    const uint32_t real_stops = real ? (m_stops >> mk) << mk : 0u;
    const bool link = in_span && real && real_stops == 0u;
    const uint64_t lm = wv::ballot(link);
    const uint32_t f = ~lm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~lm);  // first lane whose region does not end on the real sequence
    if (f >= n_lanes) {
      n_valid = TG;
      p_end = wv::readlane(m_p, n_lanes - 1u);
    } else if (wv::readlane(real ? 1u : 0u, f) != 0u) {  // its own walk met a real stop, in granule ks
      const uint32_t ks = (uint32_t)__builtin_ctz(wv::readlane(real_stops, f));
      n_valid = wv::readlane(g0, f) + ks;
      p_end = base + n_valid * SPAN_G + ((uint32_t)lidx[n_valid] & 63u);
      end_stop = true;
    } else {  // the walk of the lane before went through the whole region, or met a real stop there
      const uint32_t kind = wv::readlane(s_kind, f - 1u), kk = wv::readlane(s_k, f - 1u);
      p_end = wv::readlane(s_p, f - 1u);
      if (kind == WK_NOMERGE) n_valid = wv::readlane(g0, f) + wv::readlane(kr, f);
      else { n_valid = wv::readlane(g0, f) + kk; end_stop = true; }  // WK_STOP
    }
  }
#ifdef SPAN_TRACE
  if (lane == 0) fprintf(stderr, "span: usable %u K %u TG %u lanes %u n_valid %u p_end-base %u stop %d\n", usable, K, TG, n_lanes, n_valid, p_end - base, (int)end_stop);
#endif

  // the index leaves LDS (phase B needs the tile's place): its verified part, to the stream's scratch
  wv::sync();
  for (uint32_t i = ulane; i < n_valid; i += 64u) idx[i] = lidx[i];
  wv::fence_global();  // (read back below, by other lanes)

  // ---- phase B: the verified granules, a tile of output at a time
  uint32_t out_pos = out_pos0;
  uint32_t e = 0;
  bool cut = false;  // the span ends before n_valid: size limit reached, a granule too rich, or a tile refused
  bool limit_cut = false;  // ... the first of these
  while (e < n_valid) {
    const uint32_t ent = e + ulane;
    const bool have = ent < n_valid;
    const uint32_t iv = have ? (uint32_t)idx[ent] : 0u;
    // where the granule behind mine starts (also across the end of the tile: the next tile's first)
    const uint32_t next_epd = ent + 1u < n_valid ? (uint32_t)idx[ent + 1u] & 63u : 0u;
    const uint32_t eod = iv >> 6, epd = iv & 63u;
    const uint32_t incl = wv::scan_incl(eod);
    const bool fit = have && eod != SPAN_OD_BIG && incl <= SPAN_TILE && incl <= cap_min - out_pos;
    const uint64_t fm = wv::ballot(fit);
    const uint32_t n = ~fm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~fm);
    const uint32_t tile_start_p = base + e * SPAN_G + wv::readlane(epd, 0u);
    if (n == 0u) {  // the output limit (the symbol that crosses it is lane_one_symbol's), or a granule too rich for a tile
      p_end = tile_start_p;
      cut = true;
      limit_cut = wv::readlane(eod, 0u) != SPAN_OD_BIG;
      break;
    }
    const uint32_t tile_len = wv::readlane(incl, n - 1u);
    if (MODE == IM_DRY && ck != nullptr) {  // (the tile's start is where the tile before was checked to end, or the span's first bit)
      const uint64_t rel = (uint64_t)in_word * 32u + tile_start_p - ck->hdr_bit;
      if (ck->n < CK_MAX && rel >= (uint64_t)ck->last + CK_GAP_BITS && rel < 0xFFFFFFFFull) {
        if (ulane == 0u) { ck->lds[2u * ck->n] = (uint32_t)rel; ck->lds[2u * ck->n + 1u] = out_pos - ck->out_base; }
        ck->n++;
        ck->last = (uint32_t)rel;
      }
    }
    const bool mine = ulane < n;
    const uint32_t o0 = incl - eod, o_end = incl;
    uint32_t o = o0;
    const uint32_t p0 = base + (mine ? ent : e) * SPAN_G + (mine ? epd : 0u);
    uint32_t p = p0;
    const uint32_t pe = ent + 1u == n_valid ? p_end : base + (ent + 1u) * SPAN_G;
    mbits[ulane] = 0;
    mbits[ulane + 64u] = 0;
    if (ulane < 2u) mbits[128u + ulane] = 0;
    span_reader_start(R, E, p);
    wv::sync();
    ZD_SPAN_PH(2);
    uint32_t err = 0;
    const uint8_t *gbase = dst + out_pos;  // the tile's place in the output; gbase[-n] is final for every n >= 1
    // A match of 4 to 8 bytes whose source lies wholly before the tile (on the configs' data: most)
    // is requested from memory the moment it is decoded and lands SPAN_FLY steps later, when the
    // lane comes by the same slot again: the random window reads of all lanes overlap with the
    // decoding, and such a match needs neither a record nor a bit in the bitmap.
    uint32_t f_meta[SPAN_FLY], f_a[SPAN_FLY], f_b[SPAN_FLY];
#pragma unroll
    for (int u = 0; u < SPAN_FLY; u++) { f_meta[u] = 0; f_a[u] = 0; f_b[u] = 0; }
    // (Every step issues its two loads, lanes without such a match from the output's first bytes,
    // and the loop is left only between groups of SPAN_FLY steps: with a branch around a load the
    // compiler must assume at the landing that nothing was requested since, and wait for all.
    // A lane's state is numbers, not flags: it decodes while p < stop_p, and stop_p is 0 for a
    // lane that has no granule in the tile or met something it must not commit.)
    uint32_t stop_p = mine ? pe : 0u;
    uint32_t nh = 0;  // holes the lane leaves in the tile
    uint32_t has_far = 0;
    // (round 6, from the assembly.  Written as "for (;;) { if (no lane decodes) break; four steps }" the loop began with twelve
    // register moves -- the far matches' three words per slot, carried round the loop -- and with them a wait for ALL the
    // loads in flight: a request landed one group of steps later at the earliest, not SPAN_FLY steps later.  And the
    // reader's first words come by global loads too: the compiler's count of them is merged into the loop's, and every
    // group's first peek waited for vmcnt(0).  The test at the bottom, and the reader's words made to land here, once a tile:
    // the landings wait for exactly their own two loads (vmcnt(7), vmcnt(6)).)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(R.w0), "+v"(R.w1), "+v"(R.w2), "+v"(R.w3));  // (an operand of an asm has landed; a bare s_waitcnt the loads are moved behind)
#endif
    if (wv::any(p < stop_p)) do {
#pragma unroll
      for (int u = 0; u < SPAN_FLY; u++) {
        if (MODE == IM_REAL) span_land(tile, f_meta[u], f_a[u], f_b[u]);
        const bool act = p < stop_p;
#ifdef SPAN_TRACE
        if (lane == 0) span_trace_steps[2]++;
#endif
        uint32_t xlo, xhi;
        span_peek(R, p, xlo, xhi);
        const SpanSym s = span_symbol<true>(act, xlo, xhi, L, lit_max, dist_max);
        const uint32_t o2 = o + s.outlen;
        // what the plain decoder must see for itself: a stop, more bytes than phase A counted, a
        // distance that reaches before the output (zd.ml:614)
        const bool bad = act && (s.stop || o2 > o_end || (!s.is_lit && s.val > out_pos + o));
        const bool good = act && !bad;
        stop_p = bad ? 0u : stop_p;
        err = bad ? 1u : err;
        const bool is_match = good && !s.is_lit;
        const bool fly = MODE == IM_REAL && is_match && s.outlen - 4u <= 4u && s.val >= o2;
        if (MODE != IM_DRY && good && s.is_lit) {
          tile[o] = (uint8_t)s.val;
          if (s.outlen == 2u) tile[o + 1u] = (uint8_t)s.val2;
        }
        if (MODE == IM_REAL) {
          const uint32_t goff = fly ? out_pos + o - s.val : 0u;  // (the output has 8 bytes: checked above)
          f_a[u] = load_u32_le(dst + goff);
          f_b[u] = load_u32_le(dst + (goff + (fly ? s.outlen - 4u : 0u)));
          f_meta[u] = fly ? (o + 1u) | (s.outlen << 16) : 0u;
          nh += is_match && !fly ? 1u : 0u;
          if (is_match && !fly) {
            store_u16_le(tile + o, (uint16_t)(s.val - 1u));
            tile[o + 2u] = (uint8_t)(s.outlen - 3u);
            span_bits_set(mbits, o, s.outlen);
            if (s.val >= o2 && s.outlen <= SPAN_LONG) has_far = 1;  // (a hole the pass over far sources is for)
          }
        } else if (MODE == IM_TOKEN) {
          if (is_match) {  // (recorded like a hole: a tile that is refused below must leave nothing behind)
            store_u16_le(tile + o, (uint16_t)(s.val - 1u));
            tile[o + 2u] = (uint8_t)(s.outlen - 3u);
            span_bits_set(mbits, o, s.outlen);
          }
        }
        o = good ? o2 : o;
        p += good ? s.tot : 0u;
        span_advance(R, E, p);
      }
    } while (wv::any(p < stop_p));
    if (MODE == IM_REAL) {
#pragma unroll
      for (int u = 0; u < SPAN_FLY; u++) span_land(tile, f_meta[u], f_a[u], f_b[u]);
    }
    // both phases must have walked the same symbols
    if (mine && err == 0u) {
      if (o != o_end) err = 1;
      if (p != (ent + 1u == n_valid ? p_end : base + (ent + 1u) * SPAN_G + next_epd)) err = 1;
    }
    if (wv::any(err != 0u)) {  // leave the tile's symbols to the plain decoder: it finds what is wrong
      p_end = tile_start_p;
      cut = true;
      break;
    }
    wv::sync();
    ZD_SPAN_PH(3);
    if (MODE == IM_TOKEN && srcpos == nullptr) {  // the tile stands: where the bytes of its matches come from (every lane walks its own)
      uint32_t cursor = mine ? o0 : 0u;
      const uint32_t range_end = mine ? o_end : 0u;
      for (;;) {
        const uint32_t dp = span_bits_first(mbits, cursor, range_end);
        const bool open = dp != 0xFFFFFFFFu;
        if (!wv::any(open)) break;
        if (open) {
          const uint32_t rec = span_rec(tile, dp);
          const uint32_t dist = (rec & 0x7FFFu) + 1u, len = (rec >> 16) + 3u;
          const uint32_t at = out_pos + dp, from = at - dist;
          for (uint32_t i = 0; i < len; i++) tok[at + i] = from + i;
          cursor = dp + len;
        }
      }
    } else if (MODE == IM_TOKEN) {
      // The same for a long stream, where the chains of copies get long (a phrase of a text is a copy of a copy of
      // ... all the way to the stream's start) and every link costs the resolve rounds a pass: a source is written
      // down as what IT is a copy of, as far as that is known.  srcpos[b] = the tile position byte b of a match
      // copies + 32768 (a position before the tile is below 32768): every lane writes its own matches' (Buf.recopy
      // zd.ml:63-75: byte i of a match is byte i mod dist of its source), one pass follows the chains inside the
      // tile a step, and then the tile's bytes are swept 64 at a time -- a source before the tile is looked up in
      // tok[]: this wave wrote it a while ago with ITS source looked up the same way.  (What another wave has not
      // written yet reads as "a copy of itself", which is a valid if longer way to the same literal: nothing here
      // has to be complete, inflate_resolve_kernel follows what is left.)
      {
        uint32_t cursor = mine ? o0 : 0u;
        const uint32_t range_end = mine ? o_end : 0u;
        for (;;) {
          const uint32_t dp = span_bits_first(mbits, cursor, range_end);
          const bool open = dp != 0xFFFFFFFFu;
          if (!wv::any(open)) break;
          if (open) {
            const uint32_t rec = span_rec(tile, dp);
            const uint32_t dist = (rec & 0x7FFFu) + 1u, len = (rec >> 16) + 3u;
            const uint32_t from = dp + 32768u - dist;
            uint32_t r = 0;
            for (uint32_t i = 0; i < len; i++) {
              srcpos[dp + i] = (uint16_t)(from + r);
              r = r + 1u == dist ? 0u : r + 1u;
            }
            cursor = dp + len;
          }
        }
      }
      wv::sync();
      // (mbits: a bit per tile byte, set for the bytes of matches; lane l sweeps bytes 64 k + l: bit l & 31 of word 2 k + (l >> 5))
      const uint32_t my_bit = 1u << (ulane & 31u), my_word = ulane >> 5;
      for (uint32_t b = ulane, w = my_word; b < tile_len; b += 64u, w += 2u) {
        if (mbits[w] & my_bit) {
          const uint32_t sp = srcpos[b];
          if (sp >= 32768u && (mbits[(sp - 32768u) >> 5] >> (sp & 31u)) & 1u) srcpos[b] = srcpos[sp - 32768u];
        }
      }
      wv::sync();
      for (uint32_t b0 = 0; b0 < tile_len; b0 += 512u) {  // 8 loads in flight a lane
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
          const uint32_t b = b0 + u * 64u + ulane;
          const bool m = b < tile_len && (mbits[(b0 >> 5) + 2u * u + my_word] & my_bit) != 0u;
          const uint32_t sp = m ? (uint32_t)srcpos[b] : 32768u + b;
          v[u] = sp >= 32768u ? out_pos + (sp - 32768u) : wv::load_coherent(tok + (out_pos - (32768u - sp)));
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
          const uint32_t b = b0 + u * 64u + ulane;
          if (v[u] != out_pos + b) tok[out_pos + b] = v[u];
        }
      }
    }

    // The other holes.  They are listed first (tile positions, in stream order: every lane walks its
    // own, their places in the list from a scan of the counts the decode loop kept), so that the
    // work that follows is spread evenly over the lanes whatever granule a hole came from.
    //   Those whose source lies wholly before the tile (and that are not long) depend on nothing in
    // it: one pass over the list, the bytes a step requested from memory written by the step after.
    //   The rest 64 at a time in stream order, a group going round until none of it is open: a hole
    // is filled as soon as the bitmap shows that every byte of its source is final (its own bytes
    // count as final: an overlapping match is copied front to back).  The first open hole in stream
    // order always is, so every round makes progress; the rounds a group takes are the depth of its
    // matches' dependences, not their number.  Holes of up to 32 bytes are copied by their lanes
    // side by side, longer ones by the whole wave one after the other.
    uint16_t *list = (uint16_t *)E.ring;  // (the input ring is idle: SPAN_LIST_MAX entries)
    // (A tile of 3-byte matches has up to 1365 holes: the list takes the first SPAN_LIST_MAX in
    // stream order, and when those are filled the rest -- nothing before a hole depends on it.)
    uint32_t my_open = mine ? nh : 0u;
    for (;;) {
      const uint32_t hincl = wv::scan_incl(my_open), h_total = wv::readlane(hincl, 63u);
      if (h_total == 0u) break;
      uint32_t n_open = h_total < SPAN_LIST_MAX ? h_total : SPAN_LIST_MAX;
      {
        uint32_t cursor = mine ? o0 : 0u, at = hincl - my_open;
        const uint32_t range_end = mine ? o_end : 0u;
        for (;;) {
          const uint32_t dp = span_bits_first(mbits, cursor, range_end);
          const bool open = dp != 0xFFFFFFFFu;
          if (!wv::any(open)) break;
          if (open) {
            if (at < SPAN_LIST_MAX) list[at] = (uint16_t)dp;
            at++;
            cursor = dp + (span_rec(tile, dp) >> 16) + 3u;
          }
        }
        wv::sync();
      }
      if (out_pos + 32u <= hard_cap && wv::any(has_far != 0u)) {  // (16-byte loads of far sources may read into the tile's place)
        wv::Quad a0, a1;
        a0.x = a0.y = a0.z = a0.w = a1.x = a1.y = a1.z = a1.w = 0;
        uint32_t a_dp = 0, a_len = 0;  // what the step before requested
        for (uint32_t c0 = 0;; c0 += 64u) {
          const bool have = c0 + ulane < n_open;
          const uint32_t dp = have ? (uint32_t)list[c0 + ulane] : 0u;
          uint32_t dist = 1, len = 0;
          if (have) {
            const uint32_t rec = span_rec(tile, dp);
            dist = (rec & 0x7FFFu) + 1u;
            len = (rec >> 16) + 3u;
          }
          const bool far = have && len <= SPAN_LONG && dp + len <= dist;
          const uint32_t goff = far ? out_pos + dp - dist : 0u;  // (lanes with nothing to fetch: the output's first bytes)
          const wv::Quad q0 = wv::load_quad(dst + goff), q1 = wv::load_quad(dst + (goff + 16u));
          if (a_len != 0u) {
            uint8_t *t = tile + a_dp;
            const uint32_t w[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
              if (4u * i + 4u <= a_len) store_u32_le(t + 4u * i, w[i]);
              else if (4u * i < a_len) {
                t[4u * i] = (uint8_t)w[i];
                if (4u * i + 1u < a_len) t[4u * i + 1u] = (uint8_t)(w[i] >> 8);
                if (4u * i + 2u < a_len) t[4u * i + 2u] = (uint8_t)(w[i] >> 16);
              }
            }
            span_bits_clear(mbits, a_dp, a_len);
          }
          if (far) list[c0 + ulane] = 0xFFFFu;  // done
          a0 = q0; a1 = q1; a_dp = dp; a_len = far ? len : 0u;
          if (c0 >= n_open) break;  // (one step behind the last holes: what they requested is written)
        }
        wv::sync();
      }
      ZD_SPAN_PH(4);
      // What the far pass left: the list closes up (still stream order), then 64 holes at a time IN THAT ORDER.  Everything
      // before a group is final when its turn comes, so a hole of the group waits for holes of the same group only: the
      // group's lanes keep their holes in registers and go round -- test, copy what is ready, clear its bits -- until none is
      // open.  Nothing is listed again, no record is read twice, and a round that moves a few short holes skips the words none
      // of them has.  (Rounds over ALL open holes of the tile, which this replaces, kept what was not ready in the list for the
      // next round: text took 10 rounds a tile and looked at 15 groups of 64 in them, a third of its streams' time.  In
      // stream order a group of text is still 4.6 rounds deep -- a hole's source is mostly the text just before it -- so the
      // rounds are as many; they cost less: inflate_batch on text 4.73 -> 4.48 ms per GiB, the corpus 7.84 -> 7.5, a table
      // of records 2.37 -> 2.0, the benchmark's symbols 3.32 -> 3.30.)  A round that fills less than an eighth of what is
      // open is a chain, or a few side by side (tables of records: every field a copy of the record before's): the rest of
      // the group one after the other in stream order, each ready when its turn comes, by the whole wave.  (Only then: a
      // hole filled that way costs a third of a round -- finishing every group's last 8 / 16 / 24 open holes so: text 4.59 /
      // 4.85 / 5.13 ms.)
      {
        uint32_t n_near = 0;
        for (uint32_t c0 = 0; c0 < n_open; c0 += 64u) {
          const uint32_t dp = c0 + ulane < n_open ? (uint32_t)list[c0 + ulane] : 0xFFFFu;
          const uint64_t km = wv::ballot(dp != 0xFFFFu);
          if (dp != 0xFFFFu) list[n_near + (uint32_t)__builtin_popcountll(km & ((1ull << ulane) - 1ull))] = (uint16_t)dp;
          n_near += (uint32_t)__builtin_popcountll(km);
        }
        wv::sync();
        for (uint32_t c0 = 0; c0 < n_near; c0 += 64u) {
#ifdef SPAN_TRACE
          if (lane == 0) span_trace_steps[3]++;
#endif
          const bool have = c0 + ulane < n_near;
          const uint32_t dp = have ? (uint32_t)list[c0 + ulane] : 0u;
          uint32_t dist = 1, len = 0;
          if (have) {
            const uint32_t rec = span_rec(tile, dp);
            dist = (rec & 0x7FFFu) + 1u;
            len = (rec >> 16) + 3u;
          }
          const int sp = (int)dp - (int)dist;
          // the part of the source that lies in the tile and before the hole (its own bytes count as final: an
          // overlapping match is copied front to back)
          const int a = sp > 0 ? sp : 0;
          const int b = sp + (int)len < (int)dp ? sp + (int)len : (int)dp;
          bool open = have;
          for (;;) {
#ifdef SPAN_TRACE
            if (lane == 0) span_trace_steps[7]++;
#endif
            const bool ready = open && (b <= a || !span_bits_any(mbits, (uint32_t)a, (uint32_t)b));
            const bool go = ready && len <= SPAN_LONG;
            // a period of 1, 2 or 3 bytes: twelve bytes of it in registers
            uint32_t pw0 = 0, pw1 = 0, pw2 = 0;
            const bool pat = go && dist < 4u;
            if (pat) {
              const uint32_t b0 = span_byte_at(tile, gbase, sp), b1 = span_byte_at(tile, gbase, sp + (dist > 1u ? 1 : 0)),
                             b2 = span_byte_at(tile, gbase, sp + (dist > 2u ? 2 : 0));
              if (dist == 3u) {
                pw0 = b0 | b1 << 8 | b2 << 16 | b0 << 24;
                pw1 = b1 | b2 << 8 | b0 << 16 | b1 << 24;
                pw2 = b2 | b0 << 8 | b1 << 16 | b2 << 24;
              } else {
                pw0 = pw1 = pw2 = (b0 | b1 << 8) * 0x00010001u;  // (dist 1: b1 is b0)
              }
            }
            wv::sync();  // records and bits are read before anybody writes bytes over them
            // three ways to move the bytes: four at a time inside the tile (distance >= 4), the period's
            // words, byte by byte (what reaches before the tile from close to its start: rare)
            const bool words = go && !pat && sp >= 0;
            const bool any_w2 = wv::any((words || pat) && len >= 12u), any_w4 = wv::any((words || pat) && len >= 20u);  // (a word 2, a word 4 to move)
            if (words || pat) {
              uint8_t *t = tile + dp;
              const uint8_t *f = tile + (sp >= 0 ? sp : 0);
#pragma unroll
              for (uint32_t i = 0; i < 8u; i++) {
                // (the later rounds of a group move a few short holes: where no lane has a word 2, or a word 4, nobody pays for the rest)
                if ((i == 2u && !any_w2) || (i == 4u && !any_w4)) break;
                if (4u * i + 4u <= len) {
                  const uint32_t pv = i % 3u == 0u ? pw0 : i % 3u == 1u ? pw1 : pw2;
                  store_u32_le(t + 4u * i, pat ? pv : load_u32_le(f + 4u * i));
                }
              }
              if ((len & 3u) != 0u) {
                const uint32_t k = len >> 2, pv = k % 3u == 0u ? pw0 : k % 3u == 1u ? pw1 : pw2;
                if (!pat && len >= 4u) store_u32_le(t + (len - 4u), load_u32_le(f + (len - 4u)));  // (over bytes just written)
                else {
                  const uint32_t v = pat ? pv : load_u32_le(f + 4u * k);  // len 3: (the tile has 16 bytes behind it)
                  t[4u * k] = (uint8_t)v;
                  if ((len & 3u) >= 2u) t[4u * k + 1u] = (uint8_t)(v >> 8);
                  if ((len & 3u) == 3u) t[4u * k + 2u] = (uint8_t)(v >> 16);
                }
              }
            }
            if (wv::any(go && !words && !pat)) {
              for (uint32_t i = 0;; i++) {
                const bool g = go && !words && !pat && i < len;
                if (!wv::any(g)) break;
                if (g) tile[dp + i] = (uint8_t)span_byte_at(tile, gbase, sp + (int)i);
              }
            }
            // the long ones that are ready: Buf.recopy zd.ml:63-75, byte i is the source's byte i mod dist
            for (uint64_t lm = wv::ballot(ready && len > SPAN_LONG); lm != 0ull; lm &= lm - 1ull) {
              const uint32_t l = (uint32_t)__builtin_ctzll(lm);
              const uint32_t ldp = wv::readlane(dp, l), llen = wv::readlane(len, l), ldist = wv::readlane(dist, l);
              span_fill_by_wave(tile, gbase, ldp, ldist, llen, ulane);
            }
            if (ready) span_bits_clear(mbits, dp, len);
            open = open && !ready;
            const uint64_t om = wv::ballot(open), rm = wv::ballot(ready);  // (and every lane's bytes and bits are written)
            if (om == 0ull) break;
            if ((uint32_t)__builtin_popcountll(rm) * 8u < (uint32_t)__builtin_popcountll(om)) {
              // (open holes in neighbouring lanes that follow each other at one distance -- a period cut into short matches --
              // are one periodic copy)
              const uint32_t pdp = wv::shfl(dp, ulane - 1u), plen = wv::shfl(len, ulane - 1u), pdist = wv::shfl(dist, ulane - 1u);
              const bool joins = ulane != 0u && open && ((om >> (ulane - 1u)) & 1ull) != 0ull && dp == pdp + plen && dist == pdist;
              const uint64_t jm = wv::ballot(joins);
              const uint32_t lincl = wv::scan_incl(open ? len : 0u);
              for (uint64_t lm = om; lm != 0ull;) {
                const uint32_t l = (uint32_t)__builtin_ctzll(lm);
                const uint64_t after = l == 63u ? 0ull : ~(jm >> (l + 1u));  // the first lane behind l that does not join
                const uint32_t run = 1u + (after == 0ull ? 63u - l : (uint32_t)__builtin_ctzll(after));
                const uint32_t last = l + run - 1u > 63u ? 63u : l + run - 1u;
                const uint32_t ldp = wv::readlane(dp, l), ldist = wv::readlane(dist, l);
                const uint32_t total = wv::readlane(lincl, last) - wv::readlane(lincl, l) + wv::readlane(len, l);
#ifdef SPAN_TRACE
                if (lane == 0) span_trace_seq++;
#endif
                span_fill_by_wave(tile, gbase, ldp, ldist, total, ulane);
                if (ulane == 0u) span_bits_mark<false>(mbits, ldp, total);
                wv::sync();
                lm &= last == 63u ? 0ull : ~((2ull << last) - 1ull);
              }
              break;
            }
          }
        }
      }
      if (h_total <= SPAN_LIST_MAX) break;
      {  // what is still open of my own holes
        uint32_t cursor = mine ? o0 : 0u, cnt = 0;
        const uint32_t range_end = mine ? o_end : 0u;
        for (;;) {
          const uint32_t dp = span_bits_first(mbits, cursor, range_end);
          const bool open = dp != 0xFFFFFFFFu;
          if (!wv::any(open)) break;
          if (open) {
            cnt++;
            cursor = dp + (span_rec(tile, dp) >> 16) + 3u;
          }
        }
        my_open = cnt;
      }
    }
    ZD_SPAN_PH(5);
    // the tile leaves (IM_TOKEN: the bytes of its matches are whatever the tile held; tok[] says what they are)
    for (uint32_t i = ulane * 16u; MODE != IM_DRY && i < tile_len; i += 1024u) {
      if (i + 16u <= tile_len) {
        const uint32_t *t = (const uint32_t *)(tile + i);
        wv::Quad q;
        q.x = t[0]; q.y = t[1]; q.z = t[2]; q.w = t[3];
        wv::store_quad(dst + out_pos + i, q);
      } else {
        for (uint32_t j = i; j < tile_len; j++) dst[out_pos + j] = tile[j];
      }
    }
    wv::fence_global();
    ZD_SPAN_PH(6);
#ifdef ZD_INFLATE_PHASES
    span_ph[7] += 1;
#endif
    out_pos += tile_len;
    e += n;
  }

  // ---- the stream goes on behind the committed symbols
#ifdef SPAN_TRACE
  if (lane == 0) { for (int i = 0; i < 64; i++) fprintf(stderr, "%u ", span_lane_steps[i]); fprintf(stderr, "\n"); }
  if (lane == 0) fprintf(stderr, "hole groups %llu rounds %llu one-by-one %llu so far; ", (unsigned long long)span_trace_steps[3], (unsigned long long)span_trace_steps[7], (unsigned long long)span_trace_seq);
  if (lane == 0) fprintf(stderr, "steps: A %llu stitch %llu B %llu; lane-steps starved %llu running %llu lane5 %llu\n", (unsigned long long)span_trace_steps[0], (unsigned long long)span_trace_steps[1], (unsigned long long)span_trace_steps[2], (unsigned long long)span_trace_steps[4], (unsigned long long)span_trace_steps[5], (unsigned long long)span_trace_steps[6]);
#endif
  const bool progress = p_end != base || out_pos != out_pos0;
  d.out_pos = out_pos;
  d.in_word = in_word + (p_end >> 5);
  d.boff = p_end & 31u;
  d.ring_wr = d.in_word;  // the wide path's input ring holds nothing of use now
  // (a real stop is within a granule of the new position: the wide turns go there)
  // (a chain that held for less than a quarter of the span: walks that do not fall into step -- a bit
  // stream with a period does that -- make every further span of the block as poor: leave it)
  const bool poor = n_valid * 4u < TG && !end_stop;
  // the block ends here (a real stop), or the output's limit is near: the rest is the plain decoder's.  A rich
  // granule, a refused tile, a chain that broke early: such stretches are short (binaries: a run of zeros
  // between code and tables); leaving the whole block to the wide turns for one of them made ELF files
  // inflate 10 x slower than text.
  // (a span that was cut in front of such a stretch has NOT come to the stop its chain found, however near the
  // block's end the chain got: a kilobyte said twice in a row in the middle of a block of text left the 60 KiB
  // behind it to the wide turns, 4.7 ms of a stream that takes 1.4)
  if ((end_stop && !cut) || limit_cut) return SPAN_OFF;
  if (poor) return SPAN_POOR;
  return !progress || cut ? SPAN_LATER : SPAN_AGAIN;
}

// What the stream's wave does with a span's verdict (the same lines for the kernel and for tests/host_sim).  A span that
// was cut (SPAN_LATER: a granule too rich for a tile, a refused tile) leaves the next granule at least to the wide turns,
// and the wait doubles while such spans commit nothing (runs of runs), up to 1 Ki words.  A span whose chain of walks
// broke within a quarter of its regions (SPAN_POOR) is evidence about the CODE, not about a stretch: walks from arbitrary
// bits fall into step because code lengths differ, and where nearly all symbols of a block have one length -- base64,
// hex dumps, uniform bytes: corpus chunks of tests/golden/zlib_streams.json, 6-bit codes for 64 letters -- a walk that
// starts between two symbols stays between them.  Every such span costs its whole first pass over 64 regions (about
// 1 500 wave steps at 18 granules a region) and commits a region or two, which the wide turns decode in a tenth of
// that: 34 spans and 4.1 ms for one such 64 KiB stream where a stream of the benchmark's symbols takes 0.9.  So after a
// poor span the wait quadruples (to 4 Ki words: the rest of most blocks) and the next span is a small one, regions of
// SPAN_K_MIN granules, which costs a quarter to find out whether the walks meet again; a span that does well ends both.
ZD_HD void span_after(InflateLane &d, int sr, bool progressed) {
  d.span_off = sr == SPAN_OFF;
  const uint32_t cnt = d.span_fails & SPAN_FAILS_MASK;
  if (sr == SPAN_POOR) {
    const uint32_t c = cnt + 2u < 9u ? cnt + 2u : 9u;
    d.span_fails = c | SPAN_SMALL;
    d.span_retry_word = d.in_word + (SPAN_RETRY_WORDS << c);
  } else if (sr == SPAN_LATER) {
    const uint32_t c = progressed ? 0u : (cnt < 7u ? cnt + 1u : 7u);
    d.span_fails = c | (progressed ? 0u : d.span_fails & SPAN_SMALL);
    d.span_retry_word = d.in_word + (SPAN_RETRY_WORDS << c);
  } else {
    d.span_fails = 0;
  }
}

}  // namespace zd
