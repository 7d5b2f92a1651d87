// inflate_lane.h -- the serial state machine of one deflate stream.
//
// This is the per-stream half of the batch inflate kernel (inflate.hip).  A
// wavefront carries 4 independent streams; each stream is run by a GROUP of 16
// lanes that all hold the same copy of the state below and execute the same
// code.  What the 16 lanes do differently is one thing: lane s of the group
// looks up the litlen table at bit offset s of the current window, so that a
// run of literals is decoded by following the chain of code lengths through
// the 16 speculative lookups (grp.entry) instead of one dependent LDS read per
// literal.  Side effects (LDS writes, global stores) are done by lane 0 of the
// group ("writer").  tests/host_sim drives the same code with a one-lane group
// whose speculative lookups are plain table reads.
//
// Everything the stream touches while it decodes symbols lives in LDS -- its
// decode tables, a 64-word ring of its compressed input, a queue of deferred
// match copies -- so the symbol loop issues NO global loads: the only
// vector-memory traffic it generates are the literal stores (8 literals per
// 8-byte store).  Whatever needs a global load is parked and done by the whole
// wave at the next service point (inflate.hip), in lockstep, so that one memory
// latency is paid per round instead of one per match:
//   * short non-overlapping matches (len <= 16 <= dist) are queued as holes
//     {dst_pos, dist, len} and filled later -- legal because a later symbol can
//     only observe those bytes through another match, and a match whose source
//     reaches into an unfilled hole parks until the holes are filled;
//   * long or overlapping matches, stored blocks, per-block Adler-32 and the
//     refill of the input ring are requests served by all 64 lanes.
//
// The decoder is table driven (9-bit litlen / 7-bit dist lookup, canonical walk
// for longer codes) instead of the reference's bit-at-a-time walk, but accepts
// and rejects exactly the same streams, in the same order of checks:
//   read_bits / read_symbol        src/zipc_deflate.ml:564-591
//   read_block_symbols             src/zipc_deflate.ml:593-616
//   read_fixed_block               src/zipc_deflate.ml:618-621
//   read_dynamic_block             src/zipc_deflate.ml:623-669
//   read_uncompressed_block        src/zipc_deflate.ml:671-680
//   inflate_and_crc (block loop)   src/zipc_deflate.ml:692-709
//   Huffman.init_decoder           src/zipc_deflate.ml:355-391
#pragma once

#include "zd_common.h"

namespace zd {

constexpr int LIT_TBITS = 9;   // primary litlen lookup bits
constexpr int DIST_TBITS = 7;  // primary dist lookup bits (also holds the <=7-bit codelen code)

// Per-stream LDS image, interleaved over the L streams that share a wave so that
// equal indices of neighbouring streams sit in neighbouring banks: element i of
// stream l lives at [(offset + i) * L + l].
constexpr int LDS_LIT_TBL = 0;                          // 512 x u16: (sym << 4) | len
constexpr int LDS_DIST_TBL = LDS_LIT_TBL + 512;         // 128 x u16
constexpr int LDS_LIT_SYMS = LDS_DIST_TBL + 128;        // 288 x u16 symbols sorted by code
constexpr int LDS_DIST_SYMS = LDS_LIT_SYMS + 288;       // 32 x u16
constexpr int LDS_LIT_COUNTS = LDS_DIST_SYMS + 32;      // 16 x u16
constexpr int LDS_DIST_COUNTS = LDS_LIT_COUNTS + 16;    // 16 x u16
constexpr int LDS_U16_PER_LANE = LDS_DIST_COUNTS + 16;  // 992
// While a dynamic header is read the litlen table is not built yet: its region
// holds the distribution-sort cursors [0,16) and the 316 code lengths [16,336).
constexpr int LDS_LENGTHS = LDS_LIT_TBL + 16;
constexpr int RING_WORDS = 64;       // input ring, 32-bit words
constexpr int QUEUE_ENTRIES = 32;    // deferred copies, 2 words each
constexpr int LDS_U32_PER_LANE = RING_WORDS + 2 * QUEUE_ENTRIES;  // 128
constexpr int LDS_BYTES_PER_LANE = LDS_U16_PER_LANE * 2 + LDS_U32_PER_LANE * 4;  // 2496
constexpr uint32_t DEFER_MAX_LEN = 16;
constexpr int SPEC_WINDOW = 16;  // bit offsets looked up speculatively (= lanes per group)

struct LaneLds {
  uint16_t *w;  // u16 regions of this wave's block
  uint32_t *r;  // u32 regions (ring, queue) of this wave's block
  int lane;     // my stream's slot in the wave, < (1 << log2L)
  int log2L;    // log2(streams per wave)
  ZD_HD uint16_t &u16(int off, int i) const { return w[((off + i) << log2L) + lane]; }
  ZD_HD uint32_t &ring(uint32_t word) const { return r[((word & (RING_WORDS - 1)) << log2L) + lane]; }
  ZD_HD uint32_t &queue(int k, int half) const { return r[((RING_WORDS + 2 * k + half) << log2L) + lane]; }
};

enum : int {
  PH_HEADER = 0,       // at a block header
  PH_HDR_LENGTHS = 1,  // inside a dynamic header, reading the code lengths
  PH_SYMBOLS = 2,      // inside a compressed block
  PH_REQ_COPY = 3,     // stored block validated: waiting for the cooperative copy
  PH_REQ_MATCH = 4,    // a match that cannot be deferred: waiting for the lockstep copy
  PH_REQ_ADLER = 5,    // block finished: waiting for the cooperative Adler-32 update
  PH_DONE = 6
};

// Arena base pointers stay kernel arguments (so every access is a global_*
// instruction); the lane only keeps its offsets into them.
struct Arenas {
  const uint8_t *__restrict__ src;
  uint8_t *__restrict__ dst;
};

ZD_HD void store_u32_le(uint8_t *p, uint32_t v) {
  typedef uint32_t __attribute__((aligned(1), may_alias)) u32u;
  *(u32u *)p = v;
}

struct InflateLane {
  uint64_t src_off, dst_off;
  uint64_t bits;       // bit buffer, LSB first (src_bits zd.ml:536)
  uint32_t src_len;
  uint32_t in_word;    // next input word to pull from the ring
  uint32_t ring_wr;    // one past the last word staged in the ring
  uint32_t skip;       // bytes to drop from the next pulled word (after a stored block)
  int32_t nbits;       // REAL bits in `bits` (src_bits_len zd.ml:537)
  uint32_t out_pos;
  uint32_t cap_min;    // min(limit, dst_cap): fast overflow test
  uint32_t limit;      // ?decompressed_size, or 0xFFFFFFFF
  uint32_t hard_cap;   // dst_cap: bytes we may physically write
  uint32_t status;
  int32_t phase;
  int32_t final_block;
  int32_t lit_max_sym, dist_max_sym;
  uint32_t blk_out_start;  // first output byte of the current block
  uint32_t req_src, req_len, req_dist;
  uint32_t q_count, hole_min;  // deferred copies: count, lowest unfilled output position
  int32_t hdr_num, hdr_hlit, hdr_hdist, hdr_cl_max;  // dynamic header in progress
  uint32_t adler;      // running Adler_32 value (zd.ml:542) when crc_op = Adler

  ZD_HD void fail(uint32_t st) { status = st; phase = PH_DONE; }

  ZD_HD uint32_t total_words() const { return (src_len + 3u) >> 2; }
  // enough staged input for `words` more pulls, or nothing left to stage
  ZD_HD bool input_ready(uint32_t words) const {
    return ring_wr - in_word >= words || ring_wr >= total_words();
  }

  // read_bits' refill (zd.ml:570-575), a word at a time out of the LDS ring.
  // nbits only ever counts REAL bits (bytes past the end are masked to zero and
  // not counted), so "count > nbits" with nothing left to pull is the
  // reference's "src_pos > src_max" exhaustion test.
  ZD_HD void pull(const LaneLds &L) {
    if (nbits <= 32 && in_word < ring_wr) {
      uint32_t w = L.ring(in_word);
      if (skip == 0 && in_word < (src_len >> 2)) {  // a whole word, the usual case
        bits |= (uint64_t)w << nbits;
        nbits += 32;
      } else {
        const uint32_t base = in_word * 4u;
        int32_t valid = (int32_t)(src_len - base < 4u ? src_len - base : 4u);
        if (skip) {
          w = skip >= 4 ? 0u : w >> (8 * skip);
          valid -= (int32_t)skip;
          if (valid < 0) valid = 0;
          skip = 0;
        }
        bits |= (uint64_t)w << nbits;
        nbits += 8 * valid;
      }
      in_word++;
    }
  }
  ZD_HD bool take(int n, uint32_t &v) {  // false = input exhausted
    if (n > nbits) return false;
    v = (uint32_t)(bits & ((1ull << n) - 1));
    bits >>= n;
    nbits -= n;
    return true;
  }
  ZD_HD bool read_bits(const LaneLds &L, int n, uint32_t &v) {
    pull(L);
    return take(n, v);
  }

  // overflow of the output: the reference's fixed Buf fails with "Expected
  // decompression size exceeded" (zd.ml:27-29); running out of the caller's
  // dst_cap with no limit given is the boundary's DST_TOO_SMALL.
  ZD_HD void overflow(uint64_t need) { fail(need > limit ? ST_SIZE_EXCEEDED : ST_DST_TOO_SMALL); }
};

// Huffman.init_decoder zd.ml:355-391 on lengths[start .. start+n) held in the
// lane's lengths scratch; fills counts/syms regions.  scratch_off: 16 free u16
// slots for the reference's `offs` cursors.  Returns false when the reference
// raises "Corrupted data stream".
ZD_HD bool init_decoder(const LaneLds &L, int counts_off, int syms_off, int scratch_off,
                        int start, int n, int &max_sym) {
#pragma unroll 1
  for (int i = 0; i < 16; i++) L.u16(counts_off, i) = 0;
  max_sym = -1;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    int len = L.u16(LDS_LENGTHS, start + i);
    if (len != 0) { max_sym = i; L.u16(counts_off, len) += 1; }
  }
  int available = 1, num_codes = 0;
#pragma unroll 1
  for (int i = 0; i < 16; i++) {
    int used = L.u16(counts_off, i);
    if (used > available) return false;  // over-subscribed zd.ml:371
    available = 2 * (available - used);
    L.u16(scratch_off, i) = (uint16_t)num_codes;
    num_codes += used;
  }
  if ((num_codes > 1 && available > 0) || (num_codes == 1 && L.u16(counts_off, 1) != 1))
    return false;  // zd.ml:377-378
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    int len = L.u16(LDS_LENGTHS, start + i);
    if (len != 0) {
      int off = L.u16(scratch_off, len);
      L.u16(syms_off, off) = (uint16_t)i;
      L.u16(scratch_off, len) = (uint16_t)(off + 1);
    }
  }
  if (num_codes == 1) {  // zd.ml:389-390: phantom code 1 -> too-large symbol
    L.u16(counts_off, 1) = 2;
    L.u16(syms_off, 1) = (uint16_t)(max_sym + 1);
  }
  return true;
}

// Primary lookup table from the canonical (counts, syms) pair: every code of
// length <= tbits is replicated over its don't-care bits; entries left 0 send
// the decoder to the canonical walk.
ZD_HD void build_table(const LaneLds &L, int tbl_off, int tbits, int counts_off, int syms_off) {
  const int size = 1 << tbits;
  int covered = 0;
#pragma unroll 1
  for (int len = 1; len <= tbits; len++) covered += (int)L.u16(counts_off, len) << (tbits - len);
  if (covered != size) {
#pragma unroll 1
    for (int i = 0; i < size; i++) L.u16(tbl_off, i) = 0;
  }
  uint32_t code = 0;
  int idx = 0;
#pragma unroll 1
  for (int len = 1; len <= tbits; len++) {
    int cnt = L.u16(counts_off, len);
#pragma unroll 1
    for (int k = 0; k < cnt; k++) {
      uint16_t e = (uint16_t)((L.u16(syms_off, idx) << 4) | len);
      idx++;
#pragma unroll 1
      for (uint32_t j = bitrev(code, len); j < (uint32_t)size; j += 1u << len) L.u16(tbl_off, j) = e;
      code++;
    }
    code <<= 1;
  }
}

// read_symbol zd.ml:584-591.  Fast path: one lookup.  Slow path: the
// reference's canonical walk over counts/symbols, one code bit at a time.
// Returns -1 for "Corrupted data stream" (input exhausted, or the walk leaves
// the 15-bit range: the reference's counts.(16) Invalid_argument).
ZD_HD int read_symbol(InflateLane &d, const LaneLds &L, int tbl_off, int tbits, int counts_off,
                      int syms_off) {
  uint32_t e = L.u16(tbl_off, (int)(d.bits & ((1u << tbits) - 1)));
  int len = e & 15;
  if (len != 0) {
    if (len > d.nbits) return -1;
    d.bits >>= len;
    d.nbits -= len;
    return (int)(e >> 4);
  }
  int base = 0, offs = 0;
#pragma unroll 1
  for (len = 1; len <= 15; len++) {
    if (len > d.nbits) return -1;
    offs = 2 * offs + (int)((d.bits >> (len - 1)) & 1);
    int count = L.u16(counts_off, len);
    if (offs < count) {
      d.bits >>= len;
      d.nbits -= len;
      return L.u16(syms_off, base + offs);
    }
    base += count;
    offs -= count;
  }
  return -1;
}

// fixed_litlen_decoder / fixed_dist_decoder zd.ml:334-349
ZD_HD void setup_fixed(InflateLane &d, const LaneLds &L) {
#pragma unroll 1
  for (int i = 0; i < 16; i++) { L.u16(LDS_LIT_COUNTS, i) = 0; L.u16(LDS_DIST_COUNTS, i) = 0; }
  L.u16(LDS_LIT_COUNTS, 7) = 24;
  L.u16(LDS_LIT_COUNTS, 8) = 152;
  L.u16(LDS_LIT_COUNTS, 9) = 112;
#pragma unroll 1
  for (int i = 0; i <= 23; i++) L.u16(LDS_LIT_SYMS, i) = (uint16_t)(256 + i);
#pragma unroll 1
  for (int i = 24; i <= 167; i++) L.u16(LDS_LIT_SYMS, i) = (uint16_t)(i - 24);
#pragma unroll 1
  for (int i = 168; i <= 175; i++) L.u16(LDS_LIT_SYMS, i) = (uint16_t)(112 + i);
#pragma unroll 1
  for (int i = 176; i <= 287; i++) L.u16(LDS_LIT_SYMS, i) = (uint16_t)(i - 32);
  d.lit_max_sym = LITLEN_SYM_MAX;  // 286 and 287 are unused
  L.u16(LDS_DIST_COUNTS, 5) = 32;
#pragma unroll 1
  for (int i = 0; i <= 31; i++) L.u16(LDS_DIST_SYMS, i) = (uint16_t)i;
  d.dist_max_sym = DIST_SYM_MAX;  // 30 and 31 are unused
  build_table(L, LDS_LIT_TBL, LIT_TBITS, LDS_LIT_COUNTS, LDS_LIT_SYMS);
  build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
}

// read_dynamic_codes zd.ml:638-643 + read_codelen_code zd.ml:624-636: the fixed
// part of a dynamic header (at most 71 bits: the caller made 3 words ready).
ZD_HD bool setup_dynamic_begin(InflateLane &d, const LaneLds &L) {
  uint32_t v;
  if (!d.read_bits(L, 5, v)) return false;
  const int hlit = 257 + (int)v;
  if (!d.read_bits(L, 5, v)) return false;
  const int hdist = 1 + (int)v;
  if (hlit > 286 || hdist > 30) return false;  // zd.ml:641
  if (!d.read_bits(L, 4, v)) return false;
  const int hclen = 4 + (int)v;
#pragma unroll 1
  for (int i = 0; i < 19; i++) L.u16(LDS_LENGTHS, i) = 0;
#pragma unroll 1
  for (int i = 0; i < hclen; i++) {
    if (!d.read_bits(L, 3, v)) return false;
    L.u16(LDS_LENGTHS, k_codelen_order[i]) = (uint16_t)v;
  }
  // the code-length code lives in the dist regions while the header is read
  int cl_max_sym;
  if (!init_decoder(L, LDS_DIST_COUNTS, LDS_DIST_SYMS, LDS_DIST_TBL, 0, 19, cl_max_sym)) return false;
  if (cl_max_sym == -1) return false;  // zd.ml:635
  build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
  d.hdr_num = 0;
  d.hdr_hlit = hlit;
  d.hdr_hdist = hdist;
  d.hdr_cl_max = cl_max_sym;
  return true;
}

// zd.ml:644-667: the code lengths.  Returns 1 when the header is complete, 0 when
// the lane must wait for input, -1 for "Corrupted data stream".
ZD_HD int setup_dynamic_lengths(InflateLane &d, const LaneLds &L) {
  const int total = d.hdr_hlit + d.hdr_hdist;
  int num = d.hdr_num;
  uint32_t v;
#pragma unroll 1
  while (num < total) {
    if (!d.input_ready(1)) { d.hdr_num = num; return 0; }  // <= 14 bits per turn
    d.pull(L);
    int sym = read_symbol(d, L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
    if (sym < 0 || sym > d.hdr_cl_max) return -1;  // zd.ml:649
    int repeat;
    switch (sym) {
    case 16:
      if (num == 0) return -1;  // zd.ml:653
      if (!d.take(2, v)) return -1;
      repeat = 3 + (int)v;
      sym = L.u16(LDS_LENGTHS, num - 1);
      break;
    case 17:
      if (!d.take(3, v)) return -1;
      repeat = 3 + (int)v;
      sym = 0;
      break;
    case 18:
      if (!d.take(7, v)) return -1;
      repeat = 11 + (int)v;
      sym = 0;
      break;
    default: repeat = 1; break;
    }
    if (repeat > total - num) return -1;  // zd.ml:659 (may span litlen/dist)
#pragma unroll 1
    while (repeat > 0) { repeat--; L.u16(LDS_LENGTHS, num) = (uint16_t)sym; num++; }
  }
  d.hdr_num = num;
  if (L.u16(LDS_LENGTHS, 256) == 0) return -1;  // zd.ml:662
  if (!init_decoder(L, LDS_LIT_COUNTS, LDS_LIT_SYMS, LDS_LIT_TBL, 0, d.hdr_hlit, d.lit_max_sym)) return -1;
  if (!init_decoder(L, LDS_DIST_COUNTS, LDS_DIST_SYMS, LDS_DIST_TBL, d.hdr_hlit, d.hdr_hdist, d.dist_max_sym))
    return -1;
  build_table(L, LDS_LIT_TBL, LIT_TBITS, LDS_LIT_COUNTS, LDS_LIT_SYMS);
  build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
  return 1;
}

// Buf.recopy zd.ml:63-75 into global memory.  Far matches move 8 bytes at a
// time and may write up to 7 bytes past the match (inside dst_cap); later
// output overwrites them in program order.
ZD_HD void lane_copy_match(uint8_t *dst, uint32_t pos, uint32_t dist, uint32_t len, uint32_t hard_cap) {
  uint8_t *o = dst + pos;
  const uint8_t *s = o - dist;
  if (dist >= 8 && (uint64_t)pos + len + 8 <= hard_cap) {
#pragma unroll 1
    for (uint32_t i = 0; i < len; i += 8) store_u64_le(o + i, load_u64_le(s + i));
  } else {
#pragma unroll 1
    for (uint32_t i = 0; i < len; i++) o[i] = s[i];
  }
}

// One block header (inflate_loop zd.ml:694-701) up to the point where symbols
// can be decoded, or a stored block can be copied.  Returns false when the lane
// must wait for input (nothing consumed).
ZD_HD bool lane_block_header(InflateLane &d, const LaneLds &L, const uint8_t *__restrict__ sa) {
  if (!d.input_ready(3)) return false;  // 3 + 14 + 57 header bits at most
  uint32_t v;
  const uint8_t *src = sa + d.src_off;
  if (!d.read_bits(L, 1, v)) { d.fail(ST_CORRUPTED); return true; }
  d.final_block = (int)v;
  if (!d.read_bits(L, 2, v)) { d.fail(ST_CORRUPTED); return true; }
  d.blk_out_start = d.out_pos;
  switch (v) {
  case 0: {  // read_uncompressed_block zd.ml:671-680
    // the reference buffers < 8 bits, so its src_pos is the byte after the
    // last one touched: ceil(consumed_bits / 8)
    const uint64_t pulled = (uint64_t)d.in_word * 4u < d.src_len ? (uint64_t)d.in_word * 4u : d.src_len;
    const uint64_t consumed_bits = pulled * 8u - (uint64_t)d.nbits;
    uint32_t pos = (uint32_t)((consumed_bits + 7u) >> 3);
    if (d.src_len - pos < 4) { d.fail(ST_CORRUPTED); return true; }
    const uint32_t length = src[pos] | ((uint32_t)src[pos + 1] << 8);
    const uint32_t inv = src[pos + 2] | ((uint32_t)src[pos + 3] << 8);
    if (length != ((~inv) & 0xFFFFu)) { d.fail(ST_CORRUPTED); return true; }
    pos += 4;
    if (d.src_len - pos < length) { d.fail(ST_CORRUPTED); return true; }
    if ((uint64_t)d.out_pos + length > d.cap_min) { d.overflow((uint64_t)d.out_pos + length); return true; }
    d.req_src = pos;
    d.req_len = length;
    // input resumes at the byte after the block: word-aligned pull + byte skip
    const uint32_t next = pos + length;
    d.in_word = next >> 2;
    d.skip = next & 3u;
    if (d.ring_wr < d.in_word) d.ring_wr = d.in_word;  // ring holds nothing useful
    d.bits = 0;
    d.nbits = 0;
    d.phase = PH_REQ_COPY;
    return true;
  }
  case 1:
    setup_fixed(d, L);
    d.phase = PH_SYMBOLS;
    return true;
  case 2:
    if (!setup_dynamic_begin(d, L)) { d.fail(ST_CORRUPTED); return true; }
    d.phase = PH_HDR_LENGTHS;
    return true;
  default: d.fail(ST_CORRUPTED); return true;  // zd.ml:701
  }
}

enum : int { SYM_BUDGET = 0, SYM_EOB = 1, SYM_STOP = 2 };

// read_block_symbols zd.ml:593-616, at most `budget` turns.  One turn:
//   1. every lane of the group has looked the litlen table up at its own bit
//      offset (grp.lookup); the literal chain is followed through those 16
//      entries with register shuffles: V collects the offsets where a literal
//      starts, o ends on the first offset that is not a (short-coded) literal;
//   2. the lanes named by V store their literals themselves, side by side;
//   3. unless the chain ran off the window, the symbol at offset o is decoded:
//      straight from the shuffled entry when it has a short code (end of block
//      or a length), else through read_symbol's canonical walk.
// SYM_STOP: failed, parked on a request, or waiting for input.
template <typename Group>
ZD_HD int lane_symbols(InflateLane &d, const LaneLds &L, const Arenas &A, int &budget, Group &grp) {
  uint8_t *dst = A.dst + d.dst_off;
  const bool writer = grp.writer();
#pragma unroll 1
  while (budget > 0) {
    budget--;
    if (!d.input_ready(2)) return SYM_STOP;  // a turn pulls at most twice
    d.pull(L);
    grp.lookup(L, d.bits);
    uint32_t V = 0, e = 0;
    int o = 0, len = 0;
    uint32_t n = 0;
    const uint32_t room = d.cap_min - d.out_pos;
    bool go;
#pragma unroll 1
    do {
      e = grp.entry(L, d.bits, o);
      len = (int)(e & 15);
      go = (len != 0) & (e < (256u << 4)) & (o + len <= d.nbits) & (n < room);
      if (go) {
        V |= 1u << o;
        n++;
        o += len;
      }
    } while (go & (o < SPEC_WINDOW));
    grp.store_literals(L, d.bits, V, dst + d.out_pos);
    d.out_pos += n;
    d.bits >>= o;
    d.nbits -= o;
    if (o >= SPEC_WINDOW) continue;

    int sym;
    if (len != 0 && e >= (256u << 4)) {
      // short-coded non-literal: the entry is already here
      if (len > d.nbits) { d.fail(ST_CORRUPTED); return SYM_STOP; }
      d.bits >>= len;
      d.nbits -= len;
      sym = (int)(e >> 4);
    } else {
      // a literal that does not fit, a long code, or exhausted input: the plain way
      d.pull(L);
      sym = read_symbol(d, L, LDS_LIT_TBL, LIT_TBITS, LDS_LIT_COUNTS, LDS_LIT_SYMS);
      if (sym < 0) { d.fail(ST_CORRUPTED); return SYM_STOP; }
      if (sym < LITLEN_EOB) {
        if (d.out_pos >= d.cap_min) { d.overflow((uint64_t)d.out_pos + 1); return SYM_STOP; }
        if (writer) dst[d.out_pos] = (uint8_t)sym;
        d.out_pos++;
        continue;
      }
    }
    if (sym == LITLEN_EOB) return SYM_EOB;
    if (sym > d.lit_max_sym || sym > LITLEN_SYM_MAX) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    uint32_t vbase, vextra, v = 0;
    length_sym_value(sym, vbase, vextra);
    if (vextra != 0 && !d.take((int)vextra, v)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    const uint32_t length = vbase + v;
    d.pull(L);
    int dsym = read_symbol(d, L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
    if (dsym < 0 || dsym > d.dist_max_sym || dsym > DIST_SYM_MAX) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    dist_sym_value(dsym, vbase, vextra);
    v = 0;
    if (vextra != 0 && !d.take((int)vextra, v)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    const uint32_t dist = vbase + v;
    if (dist > d.out_pos) { d.fail(ST_CORRUPTED); return SYM_STOP; }  // zd.ml:614
    if ((uint64_t)d.out_pos + length > d.cap_min) { d.overflow((uint64_t)d.out_pos + length); return SYM_STOP; }
    // Buf.recopy zd.ml:615 -- queued, or handed to the wave
    const uint32_t src_pos = d.out_pos - dist;
    const bool hazard = d.q_count != 0 && src_pos + length > d.hole_min;
    if (length <= DEFER_MAX_LEN && dist >= length && !hazard && d.q_count < (uint32_t)QUEUE_ENTRIES) {
      if (d.q_count == 0) d.hole_min = d.out_pos;
      if (writer) {
        L.queue((int)d.q_count, 0) = d.out_pos;
        L.queue((int)d.q_count, 1) = dist | (length << 16);
      }
      d.q_count++;
      d.out_pos += length;
    } else {
      d.req_dist = dist;
      d.req_len = length;
      d.phase = PH_REQ_MATCH;
      return SYM_STOP;
    }
  }
  return SYM_BUDGET;
}

// One header action by the group's writer lane (the other lanes take over its
// state afterwards, grp.sync): false = must wait for input.
ZD_HD bool lane_header_step(InflateLane &d, const LaneLds &L, const uint8_t *__restrict__ sa) {
  if (d.phase == PH_HEADER) return lane_block_header(d, L, sa);
  const int r = setup_dynamic_lengths(d, L);
  if (r == 0) return false;
  if (r < 0) { d.fail(ST_CORRUPTED); return true; }
  d.phase = PH_SYMBOLS;
  return true;
}

// Runs the stream until it finishes, fails, parks, or has spent `budget` turns.
template <typename Group>
ZD_HD void lane_step(InflateLane &d, const LaneLds &L, const Arenas &A, int budget, bool crc_adler,
                     Group &grp) {
#pragma unroll 1
  while (budget > 0) {
    if (d.phase == PH_HEADER || d.phase == PH_HDR_LENGTHS) {
      bool ok = true;
      if (grp.writer()) ok = lane_header_step(d, L, A.src);
      ok = grp.sync(d, ok);
      if (!ok) break;  // waits for input
      budget -= 2;
    } else if (d.phase == PH_SYMBOLS) {
      const int r = lane_symbols(d, L, A, budget, grp);
      if (r != SYM_EOB) break;
      // inflated_block_crc zd.ml:682-690, then the loop test zd.ml:704
      if (crc_adler) { d.phase = PH_REQ_ADLER; break; }
      d.phase = d.final_block ? PH_DONE : PH_HEADER;
      budget -= 2;
    } else {
      break;
    }
  }
}

// the one-lane group of tests/host_sim: every "speculative" lookup is a table read
struct SoloGroup {
  ZD_HD bool writer() const { return true; }
  ZD_HD bool sync(InflateLane &, bool ok) const { return ok; }
  ZD_HD void lookup(const LaneLds &, uint64_t) {}
  ZD_HD uint32_t entry(const LaneLds &L, uint64_t bits, int o) const {
    return L.u16(LDS_LIT_TBL, (int)((bits >> o) & ((1u << LIT_TBITS) - 1)));
  }
  ZD_HD void store_literals(const LaneLds &L, uint64_t bits, uint32_t V, uint8_t *out) const {
    for (int o = 0, k = 0; o < SPEC_WINDOW; o++)
      if ((V >> o) & 1) out[k++] = (uint8_t)(entry(L, bits, o) >> 4);
  }
};

// after the cooperative copy of a stored block
ZD_HD void lane_after_copy(InflateLane &d, bool crc_adler) {
  d.out_pos += d.req_len;
  if (crc_adler) d.phase = PH_REQ_ADLER;
  else d.phase = d.final_block ? PH_DONE : PH_HEADER;
}
// after the lockstep copy of a parked match
ZD_HD void lane_after_match(InflateLane &d) {
  d.out_pos += d.req_len;
  d.phase = PH_SYMBOLS;
}
// after the cooperative Adler-32 update of the block's output
ZD_HD void lane_after_adler(InflateLane &d) {
  d.blk_out_start = d.out_pos;
  d.phase = d.final_block ? PH_DONE : PH_HEADER;
}

// One deferred copy: 3 <= len <= 16, dist >= len.  Two possibly overlapping
// 8-byte (or 4-byte) moves cover exactly [0, len), so neighbours are untouched;
// len = 3 (never produced by the reference's encoder, but legal) loads 4 bytes
// -- the 4th lies at or before the hole's first byte -- and stores 2 + 1.
struct DeferredCopy {
  uint64_t a, b;
  uint32_t dst_pos, len;
};
ZD_HD void deferred_load(DeferredCopy &c, const uint8_t *dst, uint32_t dst_pos, uint32_t dist_len) {
  const uint32_t dist = dist_len & 0xFFFFu, len = dist_len >> 16;
  const uint8_t *s = dst + dst_pos - dist;
  c.dst_pos = dst_pos;
  c.len = len;
  if (len >= 8) { c.a = load_u64_le(s); c.b = load_u64_le(s + len - 8); }
  else if (len >= 4) { c.a = load_u32_le(s); c.b = load_u32_le(s + len - 4); }
  else { c.a = load_u32_le(s); c.b = 0; }
}
ZD_HD void deferred_store(const DeferredCopy &c, uint8_t *dst) {
  uint8_t *o = dst + c.dst_pos;
  if (c.len >= 8) { store_u64_le(o, c.a); store_u64_le(o + c.len - 8, c.b); }
  else if (c.len >= 4) { store_u32_le(o, (uint32_t)c.a); store_u32_le(o + c.len - 4, (uint32_t)c.b); }
  else { o[0] = (uint8_t)c.a; o[1] = (uint8_t)(c.a >> 8); o[2] = (uint8_t)(c.a >> 16); }
}

ZD_HD void lane_init(InflateLane &d, const StreamDesc &s) {
  d.src_off = s.src_off;
  d.dst_off = s.dst_off;
  d.bits = 0;
  d.nbits = 0;
  d.in_word = 0;
  d.ring_wr = 0;
  d.skip = 0;
  d.out_pos = 0;
  d.status = ST_OK;
  d.phase = PH_HEADER;
  d.final_block = 0;
  d.lit_max_sym = d.dist_max_sym = -1;
  d.blk_out_start = 0;
  d.req_src = d.req_len = d.req_dist = 0;
  d.q_count = 0;
  d.hole_min = 0;
  d.hdr_num = d.hdr_hlit = d.hdr_hdist = d.hdr_cl_max = 0;
  d.adler = 1;  // Adler_32.init zd.ml:173
  if (s.src_len > 0xFFFFFFF0ull || s.dst_cap > 0xFFFFFFF0ull) {
    d.src_len = 0; d.hard_cap = 0; d.limit = 0; d.cap_min = 0;
    d.fail(ST_INVALID_ARG);
    return;
  }
  d.src_len = (uint32_t)s.src_len;
  d.hard_cap = (uint32_t)s.dst_cap;
  bool has_limit = (s.flags & STREAM_HAS_LIMIT) != 0;
  d.limit = has_limit ? (s.limit > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s.limit) : 0xFFFFFFFFu;
  d.cap_min = d.limit < d.hard_cap ? d.limit : d.hard_cap;
}

}  // namespace zd
