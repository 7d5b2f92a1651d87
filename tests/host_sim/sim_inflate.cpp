// Host simulation of the lane-serial inflate decoder (zipc_amd/csrc/inflate_lane.h).
// TEST TOOLING ONLY: compiles the __host__ __device__ lane code with g++ and
// drives one lane (L = 1), doing the wave-cooperative services (stored-block
// copy, per-block Adler-32) serially.  Lets the decoder logic be checked
// against the oracle on the CPU-only build box; the product never loads this.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "../../zipc_amd/csrc/inflate_lane.h"
#include "../../zipc_amd/csrc/inflate_span.h"
#include "../../zipc_amd/csrc/inflate_find.h"

using namespace zd;

static uint32_t adler_update_serial(uint32_t a, const uint8_t *p, uint32_t n) {
  uint32_t s1, s2;
  adler_unpack(a, s1, s2);
  uint32_t start = 0, block_len = n % ADLER_CHUNK;
  while (start < n) {
    uint32_t S1 = 0, S2 = 0;
    for (uint32_t i = 0; i < block_len; i++) { S1 += p[start + i]; S2 += (block_len - i) * p[start + i]; }
    adler_chunk_step(s1, s2, block_len, S1, S2);
    start += block_len;
    block_len = ADLER_CHUNK;
  }
  return adler_pack(s1, s2);
}

// ring refill as inflate.hip does it (one stream): words past the end are zero
static void refill(InflateLane &d, const LaneLds &L, const uint8_t *src) {
  if (d.phase == PH_DONE) return;
  const uint32_t lim = d.in_word + (uint32_t)RING_WORDS;
  uint32_t end = d.ring_wr + 64u < lim ? d.ring_wr + 64u : lim;
  for (uint32_t w = d.ring_wr; w < end; w++) {
    uint32_t v = 0;
    for (uint32_t b = 0; b < 4 && (uint64_t)w * 4 + b < d.src_len; b++) v |= (uint32_t)src[(uint64_t)w * 4 + b] << (8 * b);
    L.ring_put(w, v);
  }
  d.ring_wr = end;
}

// Host model of inflate.hip's wide turn: the 64 lanes are emulated with arrays,
// every step is the kernel's step.  Returns true when the chain stopped inside
// the window (the caller then decodes one symbol the plain way).
// counters for tools/sim_turn_stats.py: turns, committed symbols, matches, bits, and why turns ended
extern "C" { uint64_t sim_stats[16]; }
static bool wide_turn_model(InflateLane &d, const LaneLds &L, uint8_t *dst) {
  WideSym sp[64];
  bool ok[64];
  uint32_t end[64];
  uint32_t J[6][64];
  for (int s = 0; s < 64; s++) {
    const uint32_t p = d.boff + (uint32_t)s;
    const int slot = (int)(d.in_word & (uint32_t)(RING_WORDS - 1)) + (int)(p >> 5);
    const uint32_t w0 = L.slot(slot), w1 = L.slot(slot + 1), w2 = L.slot(slot + 2);
    sp[s] = wide_decode(funnel32(w1, w0, p), funnel32(w2, w1, p), L);
    ok[s] = (sp[s].is_lit() || sp[s].is_match()) && (int)sp[s].tot() <= (int)d.bits_left() - s && s != 63;
    end[s] = (uint32_t)s + sp[s].tot();
    J[0][s] = ok[s] ? (end[s] < 63u ? end[s] : 63u) : (uint32_t)s;
  }
  const int levels = d.levels;
  for (int k = 1; k < levels; k++)
    for (int t = 0; t < 64; t++) J[k][t] = J[k - 1][J[k - 1][t]];
  bool visited[64];
  for (int t = 0; t < 64; t++) {
    uint32_t v = 0;
    for (int k = levels - 1; k >= 0; k--) { const uint32_t y = J[k][v]; if (y <= (uint32_t)t) v = y; }
    visited[t] = v == (uint32_t)t;
  }
  uint32_t outoff[64], incl[64], mrank[64], run = 0, mr = 0;
  int first_match = -1;
  for (int t = 0; t < 64; t++) {
    outoff[t] = run;
    mrank[t] = mr;
    if (visited[t] && ok[t]) {
      run += sp[t].outlen();
      if (sp[t].is_match()) { if (first_match < 0) first_match = t; mr++; }
    }
    incl[t] = run;
  }
  const uint32_t INF = 0xFFFFFFFFu;
  const uint32_t room = d.cap_min - d.out_pos;
  const uint32_t qfree = (uint32_t)QUEUE_ENTRIES - d.q_count;
  const uint32_t first_match_dst = first_match >= 0 ? d.out_pos + outoff[first_match] : INF;
  const uint32_t qbase = d.hole_min < first_match_dst ? d.hole_min : first_match_dst;
  int c = -1;
  for (int t = 0; t < 64 && c < 0; t++) {
    if (!visited[t]) continue;
    if (!ok[t]) { c = t; break; }
    bool late = outoff[t] + sp[t].outlen() > room;
    if (sp[t].is_match()) {
      const uint32_t dstp = d.out_pos + outoff[t];
      const uint32_t h0 = d.hole_min, h1 = mrank[t] ? first_match_dst : INF;
      const uint32_t hole = h0 < h1 ? h0 : h1;
      late = late || sp[t].dist > dstp || mrank[t] >= qfree || dstp - qbase > QUEUE_REL_MAX ||
             dstp - sp[t].dist + sp[t].length > hole;
    }
    if (late) c = t;
  }
  if (c < 0) c = 63;  // no stop on the path: it ran into the sink (whose own hop the levels need not cover)
  sim_stats[0]++;
  if (c == 63) sim_stats[4]++;            // ran to the sink
  else if (!ok[c]) sim_stats[5]++;        // a stop entry / end of input
  else sim_stats[6]++;                    // late: room, queue, hazard
  uint32_t n_match = 0;
  int last = -1;
  for (int t = 0; t < c; t++) {
    if (!(visited[t] && ok[t])) continue;
    if (!sp[t].is_match()) dst[d.out_pos + outoff[t]] = (uint8_t)sp[t].lit;
    else {
      L.queue((int)(d.q_count + mrank[t])) = queue_pack(d.out_pos + outoff[t] - qbase, sp[t].dist, sp[t].length);
      n_match++;
    }
    last = t;
    sim_stats[1]++;
  }
  sim_stats[2] += n_match;
  uint32_t consumed = (uint32_t)c;
  if (last >= 0) {
    d.out_pos += incl[last];
    if (c == 63) consumed = end[last];
  }
  if (n_match && first_match_dst < d.hole_min) d.hole_min = first_match_dst;
  d.q_count += n_match;
  d.advance(consumed);
  sim_stats[3] += consumed;
  return c < 63;
}

// inflate.hip's span step on the emulated wave (wave_emu.h): every lane runs span_decode on its
// own copy of the (wave-uniform) state; the copies must agree afterwards.
struct SpanCall {
  InflateLane d[64];
  int ret[64];
  const LaneLds *L;
  const uint8_t *src;
  uint8_t *dst;
  // the token form (inflate.hip IM_TOKEN): tok[] instead of copies; srcpos: the tile's sources (following), or null
  uint32_t *tok;
  uint16_t *srcpos;
  uint32_t bits_cap;
};
static void span_lane(int lane, void *arg) {
  SpanCall &c = *(SpanCall *)arg;
  static uint16_t idx[SPAN_IDX_ENTRIES];  // the kernel's per-stream slot of global scratch
  if (c.tok) c.ret[lane] = span_decode<IM_TOKEN>(c.d[lane], *c.L, c.src, c.dst, idx, c.tok, c.srcpos, c.bits_cap, nullptr, lane);
  else c.ret[lane] = span_decode(c.d[lane], *c.L, c.src, c.dst, idx, nullptr, nullptr, 0xFFFFFFFFu, nullptr, lane);
}
extern "C" { uint64_t sim_span_stats[8]; }  // spans run, symbols' bits committed, output bytes committed, per return code
static int span_model(InflateLane &d, const LaneLds &L, const uint8_t *src, uint8_t *dst, bool descending,
                      uint32_t *tok = nullptr, uint16_t *srcpos = nullptr, uint32_t bits_cap = 0xFFFFFFFFu) {
  static wv::Emu emu;
  static SpanCall c;
  emu.descending = descending;
  for (int i = 0; i < 64; i++) c.d[i] = d;
  c.L = &L; c.src = src; c.dst = dst;
  c.tok = tok; c.srcpos = srcpos; c.bits_cap = bits_cap;
  emu.run(span_lane, &c);
  for (int i = 1; i < 64; i++) {
    if (c.ret[i] != c.ret[0] || memcmp(&c.d[i], &c.d[0], sizeof(InflateLane)) != 0) {
      fprintf(stderr, "span_model: lane %d disagrees with lane 0\n", i);
      abort();
    }
  }
  if (c.ret[0] != SPAN_NONE) {
    sim_span_stats[0]++;
    sim_span_stats[1] += (uint64_t)(c.d[0].in_word - d.in_word) * 32u + c.d[0].boff - d.boff;
    sim_span_stats[2] += c.d[0].out_pos - d.out_pos;
    sim_span_stats[3 + c.ret[0]]++;
  }
  d = c.d[0];
  return c.ret[0];
}

// where the blocks of the last sim_inflate started (bit, BTYPE): the finder below must find the dynamic ones
static std::vector<uint64_t> sim_block_bits;
static std::vector<int> sim_block_types;
extern "C" uint64_t sim_inflate_block_starts(uint64_t *bits, int *types, uint64_t cap) {
  for (uint64_t i = 0; i < sim_block_bits.size() && i < cap; i++) { bits[i] = sim_block_bits[i]; types[i] = sim_block_types[i]; }
  return sim_block_bits.size();
}
// inflate_find.h over every bit offset of a stream: the candidates, as inflate.hip's two kernels list them
static uint32_t sim_find_max_syms = 0xFFFFFFFFu;
extern "C" void sim_find_set_max_syms(uint32_t n) { sim_find_max_syms = n; }
extern "C" uint64_t sim_find_candidates(const uint8_t *src, uint64_t src_len, uint64_t *cand, uint64_t cap, uint64_t *n_first) {
  uint64_t n = 0, nf = 0;
  const uint64_t total_bits = src_len * 8u;
  if (cap) cand[n++] = 0;
  for (uint64_t p = 1; p < total_bits; p++) {
    // (the kernel looks only at the offsets find_header_mask32 lets through: it must let through whatever passes)
    const bool in_mask = (find_header_mask32(find_bits(src, src_len, p & ~31ull)) >> (p & 31u)) & 1u;
    const bool passes = find_header_test(find_bits(src, src_len, p), find_bits(src, src_len, p + 64u), total_bits - p);
    if (passes && !in_mask) { fprintf(stderr, "find_header_mask32 misses offset %llu\n", (unsigned long long)p); abort(); }
    if (!passes) continue;
    nf++;
    uint8_t tbl[128];
    if (!find_lengths_test(src, src_len, p, tbl, 1u, sim_find_max_syms)) continue;
    if (n < cap) cand[n] = p;
    n++;
  }
  *n_first = nf;
  return n;
}

extern "C" int sim_inflate(const uint8_t *src, uint64_t src_len, uint8_t *dst, uint64_t dst_cap,
                           int has_limit, uint64_t limit, int crc_op, uint64_t *out_len,
                           uint32_t *checksum, int budget) {
  static __attribute__((aligned(16))) uint8_t block[LDS_BYTES_PER_LANE];
  LaneLds L;
  L.at(block);
  StreamDesc s;
  memset(&s, 0, sizeof s);
  s.src_off = 0; s.src_len = src_len; s.dst_off = 0; s.dst_cap = dst_cap;
  s.limit = limit; s.flags = has_limit ? STREAM_HAS_LIMIT : 0;
  Arenas A;
  A.src = src; A.dst = dst;
  InflateLane d;
  lane_init(d, s);
  const bool crc_adler = crc_op == CRC_ADLER32;
  const bool wide = getenv("SIM_INFLATE_WIDE") != nullptr;
  const char *span_env = getenv("SIM_INFLATE_SPAN");  // "a": lanes resumed in ascending order, "d": descending
  const bool span = span_env != nullptr;
  const bool span_desc = span && span_env[0] == 'd';
  refill(d, L, src);
  sim_block_bits.clear();
  sim_block_types.clear();
  for (;;) {
    // decode phase: the kernel's round
    for (int turn = 0; turn < budget; turn++) {
      if (d.phase == PH_HEADER || d.phase == PH_HDR_LENGTHS || d.phase == PH_HDR_CODELEN) {
        const bool at_header = d.phase == PH_HEADER;
        const uint64_t hbit = (uint64_t)d.in_word * 32u + d.boff;
        if (!lane_header_step(d, L, src)) break;
        if (at_header) {
          sim_block_bits.push_back(hbit);
          sim_block_types.push_back((int)((find_bits(src, src_len, hbit) >> 1) & 3u));
        }
        if (d.phase == PH_TABLES) lane_finish_tables(d, L);
        if (d.phase == PH_SYMBOLS && !d.fixed_lazy) {
          uint32_t shortest = 15;
          for (int lane = 0; lane < 64; lane++) {
            const uint32_t m = build_wide_tables(d, L, lane);
            if (m < shortest) shortest = m;
          }
          d.levels = levels_for(shortest);
        }
      } else if (d.phase == PH_TABLES) {  // as inflate.hip: a fixed block beyond its table-free symbols
        lane_finish_tables(d, L);
        if (d.phase == PH_SYMBOLS) {
          uint32_t shortest = 15;
          for (int lane = 0; lane < 64; lane++) {
            const uint32_t m = build_wide_tables(d, L, lane);
            if (m < shortest) shortest = m;
          }
          d.levels = levels_for(shortest);
        }
      } else if (d.phase == PH_SYMBOLS && d.fixed_lazy) {
        const int rr = lane_one_symbol_fixed(d, L, A, true);
        if (rr == SYM_EOB) { d.fixed_lazy = 0; lane_end_of_block(d, crc_adler); }
        else if (rr == SYM_STOP) {
          if (d.phase == PH_REQ_MATCH && d.q_count == 0) {  // as inflate.hip: copied on the spot
            lane_copy_match(dst, d.out_pos, d.req_dist, d.req_len, d.hard_cap);
            lane_after_match(d);
          } else break;
        }
        if (d.phase == PH_SYMBOLS && d.fixed_lazy && --d.fixed_lazy == 0) d.phase = PH_TABLES;
      } else if (d.phase == PH_SYMBOLS) {
        if (span && !d.span_off && d.in_word >= d.span_retry_word) {  // as inflate.hip
          if (d.q_count) break;
          const uint32_t out_before = d.out_pos;
          const int sr = span_model(d, L, src, dst, span_desc);
          if (sr != SPAN_NONE) {
            span_after(d, sr, d.out_pos != out_before);
            break;
          }
          d.span_off = 1;
        }
        bool stopped = true;
        if (wide) {
          if (!d.input_ready(TURN_WORDS)) break;
          stopped = wide_turn_model(d, L, dst);
        }
        if (stopped) {
          const int rr = lane_one_symbol(d, L, A, true);
          if (rr == SYM_EOB) lane_end_of_block(d, crc_adler);
          else if (rr == SYM_STOP) break;
        }
      } else {
        break;
      }
    }
    // services, in the kernel's order
    for (uint32_t k = 0; k < d.q_count; k++) {
      DeferredCopy c;
      deferred_load(c, dst, d.hole_min, L.queue((int)k));
      deferred_store(c, dst);
    }
    d.q_count = 0;
    d.hole_min = 0xFFFFFFFFu;
    if (d.phase == PH_REQ_MATCH) {
      lane_copy_match(dst, d.out_pos, d.req_dist, d.req_len, d.hard_cap);
      lane_after_match(d);
    }
    if (d.phase == PH_REQ_COPY) {
      memcpy(dst + d.out_pos, src + d.req_src, d.req_len);
      lane_after_copy(d, crc_adler);
    }
    if (d.phase == PH_REQ_ADLER) {
      d.adler = adler_update_serial(d.adler, dst + d.blk_out_start, d.out_pos - d.blk_out_start);
      lane_after_adler(d);
    }
    if (d.phase == PH_DONE) break;
    refill(d, L, src);
  }
  *out_len = d.status == ST_OK ? d.out_pos : 0;
  *checksum = crc_adler ? d.adler : 0;
  return (int)d.status;
}

// The token form of a stream's decode, as inflate.hip's IM_TOKEN wave makes it of a block: the span decoder stores
// literals and writes down, for every byte of a match, the output position it copies (tok[]; with `follow`, what
// THAT position copies, as far as it is written: inflate_span.h); matches decoded one by one are written down here the
// way wave_match does.  Then the copies are resolved -- in stream order a byte's source is final when its turn comes
// -- and the result must be the stream's plain decode.  cut_every: spans are cut short like those of a wave that
// leaves a block at a checkpoint (bits_cap), every so many bits.
extern "C" int sim_inflate_token(const uint8_t *src, uint64_t src_len, uint8_t *dst, uint64_t dst_cap, int follow, int descending,
                                 uint32_t cut_every, uint64_t *out_len) {
  static __attribute__((aligned(16))) uint8_t block[LDS_BYTES_PER_LANE];
  static uint16_t srcpos[SPAN_TILE];
  LaneLds L;
  L.at(block);
  StreamDesc s;
  memset(&s, 0, sizeof s);
  s.src_len = src_len; s.dst_cap = dst_cap; s.limit = dst_cap; s.flags = STREAM_HAS_LIMIT;
  Arenas A;
  A.src = src; A.dst = dst;
  InflateLane d;
  lane_init(d, s);
  std::vector<uint32_t> tok(dst_cap + 1);
  for (uint64_t i = 0; i <= dst_cap; i++) tok[i] = (uint32_t)i;
  auto token_match = [&](uint32_t pos, uint32_t dist, uint32_t len) {
    for (uint32_t i = 0; i < len; i++) tok[pos + i] = follow ? tok[pos - dist + (dist < len ? i % dist : i)] : pos - dist + i;
  };
  refill(d, L, src);
  for (;;) {
    for (int turn = 0; turn < 512; turn++) {
      if (d.phase == PH_HEADER || d.phase == PH_HDR_LENGTHS || d.phase == PH_HDR_CODELEN) {
        if (!lane_header_step(d, L, src)) break;
        if (d.phase == PH_TABLES) lane_finish_tables(d, L);
        if (d.phase == PH_SYMBOLS && !d.fixed_lazy)
          for (int lane = 0; lane < 64; lane++) build_wide_tables(d, L, lane);
      } else if (d.phase == PH_TABLES) {
        lane_finish_tables(d, L);
        if (d.phase == PH_SYMBOLS)
          for (int lane = 0; lane < 64; lane++) build_wide_tables(d, L, lane);
      } else if (d.phase == PH_SYMBOLS && d.fixed_lazy) {
        const int rr = lane_one_symbol_fixed(d, L, A, true, false);
        if (rr == SYM_EOB) { d.fixed_lazy = 0; lane_end_of_block(d, false); }
        else if (rr == SYM_STOP) {
          if (d.phase != PH_REQ_MATCH) break;
          token_match(d.out_pos, d.req_dist, d.req_len);
          lane_after_match(d);
        }
        if (d.phase == PH_SYMBOLS && d.fixed_lazy && --d.fixed_lazy == 0) d.phase = PH_TABLES;
      } else if (d.phase == PH_SYMBOLS) {
        if (!d.span_off && d.in_word >= d.span_retry_word) {
          const uint32_t out_before = d.out_pos;
          const int sr = span_model(d, L, src, dst, descending != 0, tok.data(), follow ? srcpos : nullptr, cut_every ? cut_every : 0xFFFFFFFFu);
          if (sr != SPAN_NONE) {
            span_after(d, sr, d.out_pos != out_before);
            break;
          }
          if (!cut_every) d.span_off = 1;  // (a span cut short is tried again behind a few plain symbols)
          else d.span_retry_word = d.in_word + 2u;
        }
        const int rr = lane_one_symbol(d, L, A, true, false);
        if (rr == SYM_EOB) lane_end_of_block(d, false);
        else if (rr == SYM_STOP) {
          if (d.phase != PH_REQ_MATCH) break;
          token_match(d.out_pos, d.req_dist, d.req_len);
          lane_after_match(d);
        }
      } else {
        break;
      }
    }
    if (d.phase == PH_REQ_COPY) {
      memcpy(dst + d.out_pos, src + d.req_src, d.req_len);
      lane_after_copy(d, false);
    }
    if (d.phase == PH_DONE) break;
    refill(d, L, src);
  }
  if (d.status == ST_OK)
    for (uint32_t i = 0; i < d.out_pos; i++) {
      if (tok[i] > i) return -1;  // (a source lies before its copy)
      if (tok[i] != i) dst[i] = dst[tok[i]];
    }
  *out_len = d.status == ST_OK ? d.out_pos : 0;
  return (int)d.status;
}

// CRC-32 combination rule used by the checksum kernels
extern "C" uint32_t sim_crc_advance(uint32_t state, uint32_t raw, uint64_t nbytes) {
  return crc_state_advance(state, raw, gf2_xpow8n(nbytes));
}
extern "C" int sim_sym_values_match_tables(void) {
  for (int sym = 257; sym <= 285; sym++) {
    uint32_t b, e;
    length_sym_value(sym, b, e);
    if (((b << 4) | e) != k_length_value_of_sym[sym - 257]) return sym;
  }
  for (int sym = 0; sym <= 29; sym++) {
    uint32_t b, e;
    dist_sym_value(sym, b, e);
    if (((b << 4) | e) != k_dist_value_of_sym[sym]) return 1000 + sym;
  }
  return 0;
}
extern "C" int sim_length_to_sym(int len) { return length_to_sym(len); }
extern "C" int sim_dist_to_sym(int dist) { return dist_to_sym(dist); }
