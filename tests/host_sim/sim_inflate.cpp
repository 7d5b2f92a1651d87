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

extern "C" int sim_inflate(const uint8_t *src, uint64_t src_len, uint8_t *dst, uint64_t dst_cap,
                           int has_limit, uint64_t limit, int crc_op, uint64_t *out_len,
                           uint32_t *checksum, int budget) {
  static uint16_t w[LDS_U16_PER_LANE];
  static uint8_t b[LDS_LENGTHS_BYTES];
  LaneLds L;
  L.w = w; L.b = b; L.lane = 0; L.log2L = 0;
  StreamDesc s;
  memset(&s, 0, sizeof s);
  s.src_off = 0; s.src_len = src_len; s.dst_off = 0; s.dst_cap = dst_cap;
  s.limit = limit; s.flags = has_limit ? STREAM_HAS_LIMIT : 0;
  Arenas A;
  A.src = src; A.dst = dst;
  InflateLane d;
  lane_init(d, s);
  const bool crc_adler = crc_op == CRC_ADLER32;
  while (d.phase != PH_DONE) {
    if (d.phase == PH_HEADER || d.phase == PH_SYMBOLS) lane_step(d, L, A, budget, crc_adler);
    if (d.phase == PH_REQ_COPY) {
      memcpy(dst + d.out_pos, src + d.req_src, d.req_len);
      lane_after_copy(d, crc_adler);
    }
    if (d.phase == PH_REQ_ADLER) {
      d.adler = adler_update_serial(d.adler, dst + d.blk_out_start, d.out_pos - d.blk_out_start);
      lane_after_adler(d);
    }
  }
  *out_len = d.status == ST_OK ? d.out_pos : 0;
  *checksum = crc_adler ? d.adler : 0;
  return (int)d.status;
}

// CRC-32 combination rule used by the checksum kernels
extern "C" uint32_t sim_crc_advance(uint32_t state, uint32_t raw, uint64_t nbytes) {
  return crc_state_advance(state, raw, gf2_xpow8n(nbytes));
}
extern "C" int sim_length_to_sym(int len) { return length_to_sym(len); }
extern "C" int sim_dist_to_sym(int dist) { return dist_to_sym(dist); }
