// inflate_lane.h -- one deflate stream decoded by ONE lane.
//
// This is the lane-serial half of the batch inflate kernel (inflate.hip): a
// wavefront carries up to 16 independent streams, one per lane, each running the
// decoder below against its own tables in LDS; the wave's other half -- bulk
// copies of stored blocks and per-block Adler-32 -- is served cooperatively by
// all 64 lanes (inflate.hip).  The decoder is table driven (9-bit litlen /
// 7-bit dist lookup, canonical walk for longer codes) instead of the
// reference's bit-at-a-time walk, but it accepts and rejects exactly the same
// streams, in the same order of checks:
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

// Per-lane LDS image, interleaved over the L lanes that share a wave's block so
// that equal indices of neighbouring lanes sit in neighbouring banks:
// element i of lane l lives at [(offset + i) * L + l].
constexpr int LDS_LIT_TBL = 0;                          // 512 x u16: (sym << 4) | len
constexpr int LDS_DIST_TBL = LDS_LIT_TBL + 512;         // 128 x u16
constexpr int LDS_LIT_SYMS = LDS_DIST_TBL + 128;        // 288 x u16 symbols sorted by code
constexpr int LDS_DIST_SYMS = LDS_LIT_SYMS + 288;       // 32 x u16
constexpr int LDS_LIT_COUNTS = LDS_DIST_SYMS + 32;      // 16 x u16
constexpr int LDS_DIST_COUNTS = LDS_LIT_COUNTS + 16;    // 16 x u16
constexpr int LDS_U16_PER_LANE = LDS_DIST_COUNTS + 16;  // 992
constexpr int LDS_LENGTHS_BYTES = 320;                  // code lengths scratch (u8)
constexpr int LDS_BYTES_PER_LANE = LDS_U16_PER_LANE * 2 + LDS_LENGTHS_BYTES;  // 2304

struct LaneLds {
  uint16_t *w;  // u16 regions of this wave's block
  uint8_t *b;   // u8 lengths region of this wave's block
  int lane;     // my slot, < (1 << log2L)
  int log2L;
  ZD_HD uint16_t &u16(int off, int i) const { return w[((off + i) << log2L) + lane]; }
  ZD_HD uint8_t &len8(int i) const { return b[(i << log2L) + lane]; }
};

enum : int {
  PH_HEADER = 0,     // at a block header
  PH_SYMBOLS = 1,    // inside a compressed block
  PH_REQ_COPY = 2,   // stored block validated: waiting for the cooperative copy
  PH_REQ_ADLER = 3,  // block finished: waiting for the cooperative Adler-32 update
  PH_DONE = 4
};

// Arena base pointers stay kernel arguments (so every access is a global_*
// instruction); the lane only keeps its offsets into them.
struct Arenas {
  const uint8_t *__restrict__ src;
  uint8_t *__restrict__ dst;
};

struct InflateLane {
  uint64_t src_off, dst_off;
  uint64_t bits;
  uint32_t src_len, in_pos;
  int32_t nbits;
  uint32_t out_pos;
  uint32_t cap_min;    // min(limit, dst_cap): fast overflow test
  uint32_t limit;      // ?decompressed_size, or 0xFFFFFFFF
  uint32_t hard_cap;   // dst_cap: bytes we may physically write
  uint32_t status;
  int32_t phase;
  int32_t final_block;
  int32_t lit_max_sym, dist_max_sym;
  uint32_t blk_out_start;  // first output byte of the current block
  uint32_t req_src, req_len;
  uint32_t adler;          // running Adler_32 value (zd.ml:542) when crc_op = Adler

  ZD_HD void fail(uint32_t st) { status = st; phase = PH_DONE; }

  // read_bits' refill (zd.ml:570-575) done a word at a time: nbits only ever
  // counts REAL bits, so "count > nbits" after a refill is the reference's
  // "src_pos > src_max" exhaustion test.
  ZD_HD void refill(const uint8_t *__restrict__ sa) {
    if (nbits <= 32) {
      const uint8_t *src = sa + src_off;
      if (in_pos + 4 <= src_len) {
        bits |= (uint64_t)load_u32_le(src + in_pos) << nbits;
        nbits += 32;
        in_pos += 4;
      } else {
#pragma unroll 1
        while (in_pos < src_len && nbits <= 56) {
          bits |= (uint64_t)src[in_pos++] << nbits;
          nbits += 8;
        }
      }
    }
  }
  ZD_HD bool take(int n, uint32_t &v) {  // false = input exhausted
    if (n > nbits) return false;
    v = (uint32_t)(bits & ((1ull << n) - 1));
    bits >>= n;
    nbits -= n;
    return true;
  }
  ZD_HD bool read_bits(const uint8_t *__restrict__ sa, int n, uint32_t &v) {
    refill(sa);
    return take(n, v);
  }

  // overflow of the output: the reference's fixed Buf fails with "Expected
  // decompression size exceeded" (zd.ml:27-29); running out of the caller's
  // dst_cap with no limit given is the boundary's DST_TOO_SMALL.
  ZD_HD void overflow(uint64_t need) { fail(need > limit ? ST_SIZE_EXCEEDED : ST_DST_TOO_SMALL); }
};

// Huffman.init_decoder zd.ml:355-391 on lengths[start .. start+n) held in the
// lane's u8 scratch; fills counts/syms regions.  Returns false when the
// reference raises "Corrupted data stream".
ZD_HD bool init_decoder(const LaneLds &L, int counts_off, int syms_off, int scratch_off,
                        int start, int n, int &max_sym) {
  // scratch_off: 16 free u16 slots (the lookup-table region that is built next)
  // holding the reference's `offs` distribution-sort cursors
#pragma unroll 1
  for (int i = 0; i < 16; i++) L.u16(counts_off, i) = 0;
  max_sym = -1;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    int len = L.len8(start + i);
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
    int len = L.len8(start + i);
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

// read_dynamic_codes zd.ml:638-667 (+ read_codelen_code zd.ml:624-636)
ZD_HD bool setup_dynamic(InflateLane &d, const LaneLds &L, const uint8_t *__restrict__ sa) {
  uint32_t v;
  if (!d.read_bits(sa, 5, v)) return false;
  int hlit = 257 + (int)v;
  if (!d.read_bits(sa, 5, v)) return false;
  int hdist = 1 + (int)v;
  if (hlit > 286 || hdist > 30) return false;  // zd.ml:641
  if (!d.read_bits(sa, 4, v)) return false;
  int hclen = 4 + (int)v;
#pragma unroll 1
  for (int i = 0; i < 19; i++) L.len8(i) = 0;
#pragma unroll 1
  for (int i = 0; i < hclen; i++) {
    if (!d.read_bits(sa, 3, v)) return false;
    L.len8(k_codelen_order[i]) = (uint8_t)v;
  }
  // the code-length code lives in the dist regions while the header is read
  int cl_max_sym;
  if (!init_decoder(L, LDS_DIST_COUNTS, LDS_DIST_SYMS, LDS_DIST_TBL, 0, 19, cl_max_sym)) return false;
  if (cl_max_sym == -1) return false;  // zd.ml:635
  build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
  int num = 0;
  const int total = hlit + hdist;
#pragma unroll 1
  while (num < total) {
    d.refill(sa);
    int sym = read_symbol(d, L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
    if (sym < 0 || sym > cl_max_sym) return false;  // zd.ml:649
    int repeat;
    switch (sym) {
    case 16:
      if (num == 0) return false;  // zd.ml:653
      if (!d.read_bits(sa, 2, v)) return false;
      repeat = 3 + (int)v;
      sym = L.len8(num - 1);
      break;
    case 17:
      if (!d.read_bits(sa, 3, v)) return false;
      repeat = 3 + (int)v;
      sym = 0;
      break;
    case 18:
      if (!d.read_bits(sa, 7, v)) return false;
      repeat = 11 + (int)v;
      sym = 0;
      break;
    default: repeat = 1; break;
    }
    if (repeat > total - num) return false;  // zd.ml:659 (may span litlen/dist)
#pragma unroll 1
    while (repeat > 0) { repeat--; L.len8(num) = (uint8_t)sym; num++; }
  }
  if (L.len8(256) == 0) return false;  // zd.ml:662
  if (!init_decoder(L, LDS_LIT_COUNTS, LDS_LIT_SYMS, LDS_LIT_TBL, 0, hlit, d.lit_max_sym)) return false;
  if (!init_decoder(L, LDS_DIST_COUNTS, LDS_DIST_SYMS, LDS_DIST_TBL, hlit, hdist, d.dist_max_sym)) return false;
  build_table(L, LDS_LIT_TBL, LIT_TBITS, LDS_LIT_COUNTS, LDS_LIT_SYMS);
  build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
  return true;
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
// can be decoded, or a stored block can be copied.
ZD_HD void lane_block_header(InflateLane &d, const LaneLds &L, const uint8_t *__restrict__ sa) {
  uint32_t v;
  const uint8_t *src = sa + d.src_off;
  if (!d.read_bits(sa, 1, v)) return d.fail(ST_CORRUPTED);
  d.final_block = (int)v;
  if (!d.read_bits(sa, 2, v)) return d.fail(ST_CORRUPTED);
  d.blk_out_start = d.out_pos;
  switch (v) {
  case 0: {  // read_uncompressed_block zd.ml:671-680
    // the reference buffers < 8 bits, so its src_pos is the byte after the
    // last one touched: ceil(consumed_bits / 8)
    uint32_t consumed_bits = d.in_pos * 8u - (uint32_t)d.nbits;
    uint32_t pos = (consumed_bits + 7u) >> 3;
    if (d.src_len - pos < 4) return d.fail(ST_CORRUPTED);
    uint32_t length = src[pos] | ((uint32_t)src[pos + 1] << 8);
    uint32_t inv = src[pos + 2] | ((uint32_t)src[pos + 3] << 8);
    if (length != ((~inv) & 0xFFFFu)) return d.fail(ST_CORRUPTED);
    pos += 4;
    if (d.src_len - pos < length) return d.fail(ST_CORRUPTED);
    if ((uint64_t)d.out_pos + length > d.cap_min) return d.overflow((uint64_t)d.out_pos + length);
    d.req_src = pos;
    d.req_len = length;
    d.in_pos = pos + length;
    d.bits = 0;
    d.nbits = 0;
    d.phase = PH_REQ_COPY;
    return;
  }
  case 1:
    setup_fixed(d, L);
    d.phase = PH_SYMBOLS;
    return;
  case 2:
    if (!setup_dynamic(d, L, sa)) return d.fail(ST_CORRUPTED);
    d.phase = PH_SYMBOLS;
    return;
  default: return d.fail(ST_CORRUPTED);  // zd.ml:701
  }
}

// read_block_symbols zd.ml:593-616, at most `budget` symbols.  Returns true
// when the end-of-block symbol was read.
ZD_HD bool lane_symbols(InflateLane &d, const LaneLds &L, const Arenas &A, int budget) {
  uint8_t *dst = A.dst + d.dst_off;
#pragma unroll 1
  for (int n = 0; n < budget; n++) {
    d.refill(A.src);
    int sym = read_symbol(d, L, LDS_LIT_TBL, LIT_TBITS, LDS_LIT_COUNTS, LDS_LIT_SYMS);
    if (sym < 0) { d.fail(ST_CORRUPTED); return false; }
    if (sym < LITLEN_EOB) {
      if (d.out_pos >= d.cap_min) { d.overflow((uint64_t)d.out_pos + 1); return false; }
      dst[d.out_pos++] = (uint8_t)sym;
      continue;
    }
    if (sym == LITLEN_EOB) return true;
    if (sym > d.lit_max_sym || sym > LITLEN_SYM_MAX) { d.fail(ST_CORRUPTED); return false; }
    uint32_t lv = k_length_value_of_sym[sym - LITLEN_FIRST_LEN], v = 0;
    if ((lv & 0xF) != 0 && !d.take((int)(lv & 0xF), v)) { d.fail(ST_CORRUPTED); return false; }
    uint32_t length = (lv >> 4) + v;
    d.refill(A.src);
    int dsym = read_symbol(d, L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
    if (dsym < 0 || dsym > d.dist_max_sym || dsym > DIST_SYM_MAX) { d.fail(ST_CORRUPTED); return false; }
    uint32_t dv = k_dist_value_of_sym[dsym];
    v = 0;
    if ((dv & 0xF) != 0 && !d.take((int)(dv & 0xF), v)) { d.fail(ST_CORRUPTED); return false; }
    uint32_t dist = (dv >> 4) + v;
    if (dist > d.out_pos) { d.fail(ST_CORRUPTED); return false; }  // zd.ml:614
    if ((uint64_t)d.out_pos + length > d.cap_min) { d.overflow((uint64_t)d.out_pos + length); return false; }
    lane_copy_match(dst, d.out_pos, dist, length, d.hard_cap);
    d.out_pos += length;
  }
  return false;
}

// Runs the lane until it finishes, fails, needs a cooperative service, or has
// spent `budget` symbols.  crc_adler selects the per-block Adler-32 request.
ZD_HD void lane_step(InflateLane &d, const LaneLds &L, const Arenas &A, int budget, bool crc_adler) {
#pragma unroll 1
  while (budget > 0) {
    if (d.phase == PH_HEADER) {
      lane_block_header(d, L, A.src);
      budget -= 8;
    } else if (d.phase == PH_SYMBOLS) {
      bool eob = lane_symbols(d, L, A, budget);
      if (!eob) return;  // failed or budget spent
      // inflated_block_crc zd.ml:682-690, then the loop test zd.ml:704
      if (crc_adler) { d.phase = PH_REQ_ADLER; return; }
      d.phase = d.final_block ? PH_DONE : PH_HEADER;
      budget -= 8;
    } else {
      return;
    }
  }
}

// after the cooperative copy of a stored block
ZD_HD void lane_after_copy(InflateLane &d, bool crc_adler) {
  d.out_pos += d.req_len;
  if (crc_adler) d.phase = PH_REQ_ADLER;
  else d.phase = d.final_block ? PH_DONE : PH_HEADER;
}
// after the cooperative Adler-32 update of the block's output
ZD_HD void lane_after_adler(InflateLane &d) {
  d.blk_out_start = d.out_pos;
  d.phase = d.final_block ? PH_DONE : PH_HEADER;
}

ZD_HD void lane_init(InflateLane &d, const StreamDesc &s) {
  d.src_off = s.src_off;
  d.dst_off = s.dst_off;
  d.bits = 0;
  d.nbits = 0;
  d.in_pos = 0;
  d.out_pos = 0;
  d.status = ST_OK;
  d.phase = PH_HEADER;
  d.final_block = 0;
  d.lit_max_sym = d.dist_max_sym = -1;
  d.blk_out_start = 0;
  d.req_src = d.req_len = 0;
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
