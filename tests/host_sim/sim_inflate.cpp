// Host simulation of the lane-serial inflate decoder (zipc_amd/csrc/inflate_lane.h).
// TEST TOOLING ONLY: compiles the __host__ __device__ lane code with g++ and
// drives one lane (L = 1), doing the wave-cooperative services (stored-block
// copy, per-block Adler-32) serially.  Lets the decoder logic be checked
// against the oracle on the CPU-only build box; the product never loads this.
#include <string.h>
#include <stdlib.h>
#include "../../zipc_amd/csrc/inflate_lane.h"

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

// ring refill as inflate.hip's service_refill does it (one stream)
static void refill(InflateLane &d, const LaneLds &L, const uint8_t *src) {
  if (d.phase == PH_DONE) return;
  uint32_t total = d.total_words();
  uint32_t lim = d.in_word + (uint32_t)RING_WORDS;
  if (lim > total) lim = total;
  uint32_t end = d.ring_wr + 64u < lim ? d.ring_wr + 64u : lim;
  for (uint32_t w = d.ring_wr; w < end; w++) {
    uint32_t v = 0;
    for (uint32_t b = 0; b < 4 && w * 4 + b < d.src_len; b++) v |= (uint32_t)src[w * 4 + b] << (8 * b);
    L.ring(w) = v;
  }
  if (end > d.ring_wr) d.ring_wr = end;
}

extern "C" int sim_inflate(const uint8_t *src, uint64_t src_len, uint8_t *dst, uint64_t dst_cap,
                           int has_limit, uint64_t limit, int crc_op, uint64_t *out_len,
                           uint32_t *checksum, int budget) {
  static uint16_t w[LDS_U16_PER_LANE];
  static uint32_t r[LDS_U32_PER_LANE];
  LaneLds L;
  L.w = w; L.r = r; L.lane = 0; L.log2L = 0;
  StreamDesc s;
  memset(&s, 0, sizeof s);
  s.src_off = 0; s.src_len = src_len; s.dst_off = 0; s.dst_cap = dst_cap;
  s.limit = limit; s.flags = has_limit ? STREAM_HAS_LIMIT : 0;
  Arenas A;
  A.src = src; A.dst = dst;
  InflateLane d;
  lane_init(d, s);
  const bool crc_adler = crc_op == CRC_ADLER32;
  refill(d, L, src);
  for (;;) {
    SoloGroup grp;
    if (d.phase <= PH_SYMBOLS) lane_step(d, L, A, budget, crc_adler, grp);
    // services, in the kernel's order
    for (uint32_t k = 0; k < d.q_count; k++) {
      DeferredCopy c;
      deferred_load(c, dst, L.queue((int)k, 0), L.queue((int)k, 1));
      deferred_store(c, dst);
    }
    d.q_count = 0;
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
