// inflate_lane.h -- the state machine of one deflate stream.
//
// This is the per-stream half of the batch inflate kernel (inflate.hip), where
// ONE WAVEFRONT runs ONE STREAM: all 64 lanes hold the same copy of the state
// below and execute the same code (the kernel keeps it in scalar registers).
// What the lanes do differently is the speculative decode of inflate.hip's "wide
// turn": lane s decodes the symbol that would start s bits after the stream's
// position (wide_decode), the chain of real symbol starts is then found by
// pointer doubling, and every symbol on it is committed in parallel.  Whatever
// the wide turn cannot commit -- long codes, end of block, matches that must not
// be deferred, the tail of the input -- is decoded one symbol at a time by
// lane_one_symbol, which is also all that tests/host_sim's plain model needs.
//
// Everything the stream touches while it decodes symbols lives in LDS -- its
// decode tables, a 64-word ring of its compressed input, a queue of deferred
// match copies -- so decoding issues NO global loads: the only vector-memory
// traffic it generates are the literal stores.  The stream position is a word
// index into the input plus a bit offset; there is no shifted bit buffer to
// maintain, every reader takes its bits straight from the ring.  Whatever needs
// a global load is parked and done at the next service point (inflate.hip) by
// all 64 lanes:
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

// Per-stream LDS image (10072 bytes: 16 streams per CU).  Byte offsets from the block's start:
//      0  the span decoder's input ring, 8 words x 64 lanes (inflate_span.h); between spans its
//         first 560 bytes are the wide turn's input ring (64 + 4 words) and deferred-copy queue
//   2048  wide litlen table, 512 x u32 (wide_lit_entry)
//   4096  wide dist table, 128 x u32 (wide_dist_entry)
//   4608  the span decoder's output tile, 4096 + 16 bytes; while a block's header is read and its
//         tables are built, the tile's last 1280 bytes are the primary u16 lookup tables (and the
//         code-length scratch): read_symbol's table form is used for the code-length code only
//   8720  symbols sorted by code and counts per length of both codes (u16): what the canonical walk
//         (read_symbol_walk) needs
//   9424  the tile's bitmap, 130 words
//   9944  limits and bases per code length of both codes, 64 x u16: the canonical decode without a
//         walk (canon_symbol) for the span decoder's long codes
// The u16 regions are addressed as LaneLds::w[off + i] with w = block + 7440:
constexpr int LDS_LIT_TBL = 0;                          // 512 x u16: (sym << 4) | len   (transient, in the tile)
constexpr int LDS_DIST_TBL = LDS_LIT_TBL + 512;         // 128 x u16                     (transient, in the tile)
constexpr int LDS_LIT_SYMS = LDS_DIST_TBL + 128;        // 288 x u16 symbols sorted by code
constexpr int LDS_DIST_SYMS = LDS_LIT_SYMS + 288;       // 32 x u16
constexpr int LDS_LIT_COUNTS = LDS_DIST_SYMS + 32;      // 16 x u16
constexpr int LDS_DIST_COUNTS = LDS_LIT_COUNTS + 16;    // 16 x u16
constexpr int LDS_U16_PER_LANE = LDS_DIST_COUNTS + 16;  // 992
// While a dynamic header is read the litlen table is not built yet: its region
// holds the distribution-sort cursors [0,16) and the 316 code lengths [16,336).
constexpr int LDS_LENGTHS = LDS_LIT_TBL + 16;
// the u32 regions, addressed as LaneLds::r[off + i] with r = block:
constexpr int RING_WORDS = 64;     // input ring, 32-bit words, slot = word index & 63
constexpr int RING_MIRROR = 4;     // slots 64..67 repeat slots 0..3: readers take 3 consecutive words
constexpr int QUEUE_ENTRIES = 64;  // deferred copies, one packed word each (queue_pack)
constexpr int LDS_RING = 0;
constexpr int LDS_QUEUE = LDS_RING + RING_WORDS + RING_MIRROR;  // 64 entries + the spare slot (+3)
constexpr int LDS_WIDE_LIT = 512;                            // 512 x u32, see wide_lit_entry
constexpr int LDS_WIDE_DIST = LDS_WIDE_LIT + 512;            // 128 x u32, see wide_dist_entry
static_assert(LDS_QUEUE + QUEUE_ENTRIES + 4 <= LDS_WIDE_LIT, "ring and queue fit in front of the tables");
constexpr int LDS_SPAN_RING_BYTE = 0;
constexpr int LDS_SPAN_TILE_BYTE = (LDS_WIDE_DIST + 128) * 4;          // 4608
constexpr int LDS_SPAN_TILE_BYTES = 4096 + 16;
constexpr int LDS_W_BYTE = LDS_SPAN_TILE_BYTE + LDS_SPAN_TILE_BYTES - (LDS_LIT_SYMS * 2);  // 7440
constexpr int LDS_SPAN_BITS_BYTE = LDS_W_BYTE + LDS_U16_PER_LANE * 2;  // 9424
constexpr int LDS_CANON_BYTE = LDS_SPAN_BITS_BYTE + (4096 / 32 + 2) * 4;  // 9944: u16[64], see canon_symbol
constexpr int CANON_LIT = 0, CANON_DIST = 32;  // each: 16 limits, 16 bases
constexpr int LDS_BYTES_PER_LANE = LDS_CANON_BYTE + 64 * 2;  // 10072
static_assert(LDS_SPAN_TILE_BYTE % 16 == 0 && LDS_W_BYTE % 2 == 0 && LDS_SPAN_BITS_BYTE % 4 == 0, "alignment");
constexpr uint32_t DEFER_MAX_LEN = 16;
constexpr uint32_t DEFER_MIN_DIST = 16;  // nearer matches are not queued (they mostly depend on the copy just before them)
// A queued copy is one word: its destination relative to the first queued copy's
// (InflateLane::hole_min), distance and length.
constexpr uint32_t QUEUE_REL_MAX = 8191;
ZD_HD uint32_t queue_pack(uint32_t dst_rel, uint32_t dist, uint32_t len) {  // dist <= 32768, 3 <= len <= 16
  return (dist - 1u) | ((len - 3u) << 15) | (dst_rel << 19);
}
constexpr int SPEC_WINDOW = 64;    // bit offsets decoded speculatively (= lanes of the wave)
constexpr int SPEC_SYM_BITS = 48;  // longest symbol: 15 + 5 + 15 + 13 bits
constexpr int SPAN_SYM_BITS_MAX = SPEC_SYM_BITS;
constexpr int TURN_WORDS = 5;      // words a wide turn may touch: (31 + 63) / 32 + 3
constexpr int HEADER_WORDS = 6;    // words a block header step may touch: (31 + 3 + 71) / 32 + 3

struct LaneLds {
  uint16_t *w;  // u16 regions of this stream's block
  uint32_t *r;  // u32 regions
  uint8_t *x;   // the block itself (the span decoder's regions are byte offsets, inflate_span.h), 16-byte aligned
  ZD_HD void at(uint8_t *block) {
    x = block;
    r = (uint32_t *)block;
    w = (uint16_t *)(block + LDS_W_BYTE);
  }
  ZD_HD uint16_t &u16(int off, int i) const { return w[off + i]; }
  ZD_HD uint32_t &slot(int s) const { return r[LDS_RING + s]; }
  ZD_HD uint32_t &queue(int k) const { return r[LDS_QUEUE + k]; }
  ZD_HD uint32_t &wide_lit(int i) const { return r[LDS_WIDE_LIT + i]; }
  ZD_HD uint32_t &wide_dist(int i) const { return r[LDS_WIDE_DIST + i]; }
  ZD_HD uint16_t &canon(int i) const { return ((uint16_t *)(x + LDS_CANON_BYTE))[i]; }
  // stage input word `word` (and its mirror)
  ZD_HD void ring_put(uint32_t word, uint32_t v) const {
    const int s = (int)(word & (uint32_t)(RING_WORDS - 1));
    r[LDS_RING + s] = v;
    if (s < RING_MIRROR) r[LDS_RING + RING_WORDS + s] = v;
  }
};

enum : int {
  PH_HEADER = 0,       // at a block header
  PH_HDR_LENGTHS = 1,  // inside a dynamic header, reading the code lengths
  PH_SYMBOLS = 2,      // inside a compressed block
  PH_REQ_COPY = 3,     // stored block validated: waiting for the cooperative copy
  PH_REQ_MATCH = 4,    // a match that cannot be deferred: waiting for the lockstep copy
  PH_REQ_ADLER = 5,    // block finished: waiting for the cooperative Adler-32 update
  PH_DONE = 6,
  PH_TABLES = 7,       // code lengths known (hdr_fixed, or the lengths scratch): decode tables to be built
  PH_HDR_CODELEN = 8   // a dynamic header's counts are read: the code-length code's lengths and tables next
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
  uint32_t src_len;
  uint32_t in_word;    // stream position: input word ...
  uint32_t boff;       // ... and bit in it, < 32 (src_pos / src_bits of zd.ml:534-537)
  uint32_t ring_wr;    // one past the last word staged in the ring
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
  uint32_t q_count, hole_min;  // deferred copies: count, lowest unfilled output position (0xFFFFFFFF: none)
  int32_t hdr_num, hdr_hlit, hdr_hdist, hdr_cl_max, hdr_hclen;  // dynamic header in progress
  int32_t hdr_fixed;   // PH_TABLES: 1 = the fixed codes
  uint32_t adler;      // running Adler_32 value (zd.ml:542) when crc_op = Adler
  int32_t levels;      // doubling levels the block's wide turns need (levels_for)
  // the span decoder (inflate_span.h)
  uint32_t blk_in_word, blk_boff;  // where the current block's symbols start
  uint32_t prev_block_bits;        // bits of the previous compressed block's symbols (0: none yet): sizes the regions
  int32_t span_off;                // the rest of this block is left to the wide turns
  uint32_t span_retry_word;        // no span before the input has reached this word (after one that met a stretch it cannot take)
  uint32_t span_fails;             // such spans in a row that committed nothing (the wait doubles with each) | SPAN_SMALL (inflate_span.h span_after)
  int32_t fixed_lazy;              // > 0: a fixed block whose tables are not built yet; symbols left before they are

  ZD_HD void fail(uint32_t st) { status = st; phase = PH_DONE; }

  ZD_HD uint32_t total_words() const { return (src_len + 3u) >> 2; }
  // `words` words from the position on are staged.  The refill stages zero
  // words past the end of the input, so this always comes true.
  ZD_HD bool input_ready(uint32_t words) const { return ring_wr - in_word >= words; }

  // REAL bits from the position to the end of the input, clamped to 2^24.
  // "count > bits_left" is the reference's "src_pos > src_max" exhaustion test
  // of read_bits (zd.ml:570-575): bytes past the end read as zero here and are
  // never counted.
  ZD_HD uint32_t bits_left() const {
    const uint32_t total = total_words();
    if (in_word >= total) return 0;
    uint32_t wl = total - in_word;
    if (wl > (1u << 19)) wl = 1u << 19;
    const uint32_t pad = (total * 4u - src_len) * 8u;  // bits of the last word past the end
    const uint32_t have = wl * 32u - pad;
    return have > boff ? have - boff : 0;
  }
  // the next 64 bits (3 words staged)
  ZD_HD uint64_t peek64(const LaneLds &L) const {
    const int s = (int)(in_word & (uint32_t)(RING_WORDS - 1));
    const uint32_t w0 = L.slot(s), w1 = L.slot(s + 1), w2 = L.slot(s + 2);
    return ((uint64_t)funnel32(w2, w1, boff) << 32) | funnel32(w1, w0, boff);
  }
  ZD_HD void advance(uint32_t n) {
    const uint32_t p = boff + n;
    in_word += p >> 5;
    boff = p & 31u;
  }
  // read_bits zd.ml:564-582, n <= 16; false = input exhausted
  ZD_HD bool read_bits(const LaneLds &L, int n, uint32_t &v) {
    if ((uint32_t)n > bits_left()) return false;
    v = (uint32_t)peek64(L) & ((1u << n) - 1u);
    advance((uint32_t)n);
    return true;
  }

  // overflow of the output: the reference's fixed Buf fails with "Expected
  // decompression size exceeded" (zd.ml:27-29); running out of the caller's
  // dst_cap with no limit given is the boundary's DST_TOO_SMALL.
  ZD_HD void overflow(uint64_t need) { fail(need > limit ? ST_SIZE_EXCEEDED : ST_DST_TOO_SMALL); }
};

// A symbol being decoded out of a 64-bit peek: `used` of `avail` real bits.
struct BitCursor {
  uint64_t x;
  int used, avail;
  ZD_HD bool take(int n, uint32_t &v) {  // false = input exhausted
    if (used + n > avail) return false;
    v = (uint32_t)(x >> used) & ((1u << n) - 1u);
    used += n;
    return true;
  }
};
ZD_HD BitCursor cursor_at(const InflateLane &d, const LaneLds &L) {
  BitCursor c;
  c.x = d.peek64(L);
  c.used = 0;
  const uint32_t left = d.bits_left();
  c.avail = left > 64u ? 64 : (int)left;
  return c;
}

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
ZD_HD int read_symbol(BitCursor &c, const LaneLds &L, int tbl_off, int tbits, int counts_off,
                      int syms_off) {
  const uint32_t e = L.u16(tbl_off, (int)((uint32_t)(c.x >> c.used) & ((1u << tbits) - 1)));
  int len = e & 15;
  if (len != 0) {
    if (c.used + len > c.avail) return -1;
    c.used += len;
    return (int)(e >> 4);
  }
  int base = 0, offs = 0;
#pragma unroll 1
  for (len = 1; len <= 15; len++) {
    if (c.used + len > c.avail) return -1;
    offs = 2 * offs + (int)((c.x >> (c.used + len - 1)) & 1);
    int count = L.u16(counts_off, len);
    if (offs < count) {
      c.used += len;
      return L.u16(syms_off, base + offs);
    }
    base += count;
    offs -= count;
  }
  return -1;
}

// read_symbol by the canonical walk alone: what decodes the symbols of a block outside the
// wide tables (the primary u16 tables share their LDS with the span decoder's output tile and
// are only valid while a block's tables are being built).
ZD_HD int read_symbol_walk(BitCursor &c, const LaneLds &L, int counts_off, int syms_off) {
  int base = 0, offs = 0;
#pragma unroll 1
  for (int len = 1; len <= 15; len++) {
    if (c.used + len > c.avail) return -1;
    offs = 2 * offs + (int)((c.x >> (c.used + len - 1)) & 1);
    int count = L.u16(counts_off, len);
    if (offs < count) {
      c.used += len;
      return L.u16(syms_off, base + offs);
    }
    base += count;
    offs -= count;
  }
  return -1;
}

// read_symbol without the walk, for input that does not end within the code (x: the next 15 bits
// at least).  A canonical code read first bit first is a number that grows with the code's length:
// limit[l] = the first 15-bit number no code of length <= l starts with (build_canon, from the
// decoder's counts), so the length is found by bisection and the symbol by its rank among the
// codes of that length.  -1: no code (the reference's "Corrupted data stream").
ZD_HD int canon_symbol(uint32_t x, const LaneLds &L, int canon_off, int syms_off, uint32_t &len) {
  const uint32_t c = bitrev(x & 0x7FFFu, 15);
  uint32_t lo = 1, hi = 16;
#pragma unroll
  for (int step = 0; step < 4; step++) {
    const uint32_t mid = (lo + hi) >> 1;
    if (c < (uint32_t)L.canon(canon_off + (int)mid)) hi = mid;
    else lo = mid + 1u;
  }
  len = lo;
  if (lo > 15u) return -1;
  const uint32_t rank = (c - (uint32_t)L.canon(canon_off + (int)lo - 1)) >> (15u - lo);
  return L.u16(syms_off, (int)((uint32_t)L.canon(canon_off + 16 + (int)lo) + rank));
}
// lanes 0..15 (litlen) and 16..31 (distance): limit and base of one length each
ZD_HD void build_canon(const LaneLds &L, int lane) {
  if (lane >= 32) return;
  const int j = lane & 15, counts_off = lane < 16 ? LDS_LIT_COUNTS : LDS_DIST_COUNTS, to = lane < 16 ? CANON_LIT : CANON_DIST;
  uint32_t lim = 0, base = 0;
#pragma unroll 1
  for (int i = 1; i <= j; i++) {
    const uint32_t n = L.u16(counts_off, i);
    lim += n << (15 - i);
    if (i < j) base += n;
  }
  L.canon(to + j) = (uint16_t)(lim < 32768u ? lim : 32768u);
  L.canon(to + 16 + j) = (uint16_t)base;
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
// part of a dynamic header (at most 71 bits: the caller made HEADER_WORDS ready).
ZD_HD bool setup_dynamic_counts(InflateLane &d, const LaneLds &L) {
  uint32_t v;
  if (!d.read_bits(L, 5, v)) return false;
  const int hlit = 257 + (int)v;
  if (!d.read_bits(L, 5, v)) return false;
  const int hdist = 1 + (int)v;
  if (hlit > 286 || hdist > 30) return false;  // zd.ml:641
  if (!d.read_bits(L, 4, v)) return false;
  d.hdr_hlit = hlit;
  d.hdr_hdist = hdist;
  d.hdr_hclen = 4 + (int)v;
  return true;
}
// read_codelen_code zd.ml:624-636, one lane's form (the kernel has a wave-parallel one with the
// same results, wave_tables in inflate.hip)
ZD_HD bool setup_codelen_code(InflateLane &d, const LaneLds &L) {
  uint32_t v;
#pragma unroll 1
  for (int i = 0; i < 19; i++) L.u16(LDS_LENGTHS, i) = 0;
#pragma unroll 1
  for (int i = 0; i < d.hdr_hclen; i++) {
    if (!d.read_bits(L, 3, v)) return false;
    L.u16(LDS_LENGTHS, k_codelen_order[i]) = (uint16_t)v;
  }
  // the code-length code lives in the dist regions while the header is read
  int cl_max_sym;
  if (!init_decoder(L, LDS_DIST_COUNTS, LDS_DIST_SYMS, LDS_DIST_TBL, 0, 19, cl_max_sym)) return false;
  if (cl_max_sym == -1) return false;  // zd.ml:635
  build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
  d.hdr_num = 0;
  d.hdr_cl_max = cl_max_sym;
  return true;
}

// zd.ml:644-667: the code lengths.  Returns 1 when the header is complete, 0 when
// the lane must wait for input, -1 for "Corrupted data stream".
// (One lane reads these, symbol after symbol, while 63 wait: the loop keeps what it can in
// registers -- the bits left in the input, the last length -- and takes a symbol and its extra
// bits out of one 32-bit peek: a code-length code has at most 7 bits, which the primary table
// resolves or no code does, and every way to fail is the same "Corrupted data stream".
// Reading the code-length code's own lengths and building its table by the whole wave was
// measured too: 20 K clocks less per header, and 27 spilled VGPRs in the symbol loops for the
// code it adds to the kernel -- 4.30 ms against 4.15 on C2.)
ZD_HD int setup_dynamic_lengths(InflateLane &d, const LaneLds &L) {
  const int total = d.hdr_hlit + d.hdr_hdist;
  int num = d.hdr_num;
  uint32_t left = d.bits_left();  // (clamped far above what a header can take)
  uint32_t prev = num > 0 ? L.u16(LDS_LENGTHS, num - 1) : 0u;
#pragma unroll 1
  while (num < total) {
    if (!d.input_ready(3)) { d.hdr_num = num; return 0; }  // <= 14 bits per turn
    const int s = (int)(d.in_word & (uint32_t)(RING_WORDS - 1));
    const uint32_t x = funnel32(L.slot(s + 1), L.slot(s), d.boff);
    const uint32_t e = L.u16(LDS_DIST_TBL, (int)(x & ((1u << DIST_TBITS) - 1)));
    const uint32_t len = e & 15u, sym = e >> 4;
    if (len == 0u || (int)sym > d.hdr_cl_max) return -1;  // no such code (read_symbol's walk ends the same way); zd.ml:649
    const uint32_t extra = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
    const uint32_t used = len + extra;
    if (used > left) return -1;  // the input ends inside the code or its extra bits
    const uint32_t v = (x >> len) & ((1u << extra) - 1u);
    uint32_t repeat = 1, fill = sym;
    if (sym >= 16u) {
      if (sym == 16u) {
        if (num == 0) return -1;  // zd.ml:653
        repeat = 3u + v;
        fill = prev;
      } else {
        repeat = (sym == 17u ? 3u : 11u) + v;
        fill = 0;
      }
    }
    d.advance(used);
    left -= used;
    if (repeat > (uint32_t)(total - num)) return -1;  // zd.ml:659 (may span litlen/dist)
#pragma unroll 1
    while (repeat > 0u) { repeat--; L.u16(LDS_LENGTHS, num) = (uint16_t)fill; num++; }
    prev = fill;
  }
  d.hdr_num = num;
  if (L.u16(LDS_LENGTHS, 256) == 0) return -1;  // zd.ml:662
  return 1;  // the two decoders (zd.ml:663-666) are built in phase PH_TABLES
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

ZD_HD void lane_begin_symbols(InflateLane &d);
constexpr int FIXED_LAZY_SYMBOLS = 48;  // symbols of a fixed block decoded without tables

// One block header (inflate_loop zd.ml:694-701) up to the point where symbols
// can be decoded, or a stored block can be copied.  Returns false when the lane
// must wait for input (nothing consumed).
ZD_HD bool lane_block_header(InflateLane &d, const LaneLds &L, const uint8_t *__restrict__ sa) {
  if (!d.input_ready(HEADER_WORDS)) return false;  // 3 + 14 + 57 header bits at most
  uint32_t v;
  const uint8_t *src = sa + d.src_off;
  if (!d.read_bits(L, 1, v)) { d.fail(ST_CORRUPTED); return true; }
  d.final_block = (int)v;
  if (!d.read_bits(L, 2, v)) { d.fail(ST_CORRUPTED); return true; }
  d.blk_out_start = d.out_pos;
  switch (v) {
  case 0: {  // read_uncompressed_block zd.ml:671-680
    // the reference buffers < 8 bits, so its src_pos is the byte after the
    // last one touched
    uint32_t pos = d.in_word * 4u + ((d.boff + 7u) >> 3);
    if (pos > d.src_len || d.src_len - pos < 4) { d.fail(ST_CORRUPTED); return true; }
    const uint32_t length = src[pos] | ((uint32_t)src[pos + 1] << 8);
    const uint32_t inv = src[pos + 2] | ((uint32_t)src[pos + 3] << 8);
    if (length != ((~inv) & 0xFFFFu)) { d.fail(ST_CORRUPTED); return true; }
    pos += 4;
    if (d.src_len - pos < length) { d.fail(ST_CORRUPTED); return true; }
    if ((uint64_t)d.out_pos + length > d.cap_min) { d.overflow((uint64_t)d.out_pos + length); return true; }
    d.req_src = pos;
    d.req_len = length;
    // input resumes at the byte after the block
    const uint32_t next = pos + length;
    d.in_word = next >> 2;
    d.boff = (next & 3u) * 8u;
    if (d.ring_wr < d.in_word) d.ring_wr = d.in_word;  // ring holds nothing useful
    d.phase = PH_REQ_COPY;
    return true;
  }
  case 1:
    // A fixed block's first symbols are decoded from the code's arithmetic (lane_one_symbol_fixed):
    // the reference's encoder ends every stream of a few KiB or less, and every 64 KiB stream whose
    // last block holds a couple of bytes, with such a block, and 1152 table entries would be built
    // for a handful of symbols.  A block that goes on gets its tables then (PH_TABLES).
    d.hdr_fixed = 1;
    lane_begin_symbols(d);
    d.lit_max_sym = LITLEN_SYM_MAX;  // 286 and 287 are unused (zd.ml:342)
    d.dist_max_sym = DIST_SYM_MAX;   // 30 and 31 are unused (zd.ml:349)
    d.fixed_lazy = FIXED_LAZY_SYMBOLS;
    return true;
  case 2:
    if (!setup_dynamic_counts(d, L)) { d.fail(ST_CORRUPTED); return true; }
    d.phase = PH_HDR_CODELEN;
    return true;
  default: d.fail(ST_CORRUPTED); return true;  // zd.ml:701
  }
}

enum : int { SYM_OK = 0, SYM_EOB = 1, SYM_STOP = 2 };
// What a wave does with the bytes (inflate.hip): IM_REAL the stream's output; IM_DRY nothing -- one block's symbols
// are walked for its end and its size; IM_TOKEN literals are stored and a match leaves, per byte, the position it
// copies (one stream's blocks side by side: the copies are resolved once all of them are known).
enum : int { IM_REAL = 0, IM_DRY = 1, IM_TOKEN = 2 };

// A decoded match (its bits in c): the reference's checks, then Buf.recopy zd.ml:615 -- queued, or
// handed to the wave
ZD_HD int lane_match_commit(InflateLane &d, const LaneLds &L, bool writer, const BitCursor &c, uint32_t length, uint32_t dist,
                            bool may_queue = true) {
  if (dist > d.out_pos) { d.fail(ST_CORRUPTED); return SYM_STOP; }  // zd.ml:614
  if ((uint64_t)d.out_pos + length > d.cap_min) { d.overflow((uint64_t)d.out_pos + length); return SYM_STOP; }
  d.advance((uint32_t)c.used);
  // Buf.recopy zd.ml:615 -- queued, or handed to the wave
  const uint32_t src_pos = d.out_pos - dist;
  const bool hazard = src_pos + length > d.hole_min;
  const uint32_t qbase = d.hole_min < d.out_pos ? d.hole_min : d.out_pos;  // hole_min once this one is queued
  const bool in_reach = d.out_pos - qbase <= QUEUE_REL_MAX;
  if (may_queue && length <= DEFER_MAX_LEN && dist >= length && dist >= DEFER_MIN_DIST && !hazard && in_reach && d.q_count < (uint32_t)QUEUE_ENTRIES) {
    d.hole_min = qbase;
    if (writer) L.queue((int)d.q_count) = queue_pack(d.out_pos - d.hole_min, dist, length);
    d.q_count++;
    d.out_pos += length;
    return SYM_OK;
  }
  d.req_dist = dist;
  d.req_len = length;
  d.phase = PH_REQ_MATCH;
  return SYM_STOP;
}

// Exactly one symbol of read_block_symbols (zd.ml:593-616), decoded the plain
// way.  SYM_STOP: failed, parked on a request, or waiting for input.
ZD_HD int lane_one_symbol(InflateLane &d, const LaneLds &L, const Arenas &A, bool writer, bool may_queue = true) {
  uint8_t *dst = A.dst + d.dst_off;
  if (!d.input_ready(3)) return SYM_STOP;
  BitCursor c = cursor_at(d, L);
  uint32_t length, dist;
  // A match whose two codes the wide tables resolve (the entries hold only symbols of the block's
  // alphabets, every check on them made when the tables were built) with the input not about to
  // end: no walk.  Runs of long matches -- zeros, periods -- come through here one by one.
  const uint32_t xlo = (uint32_t)c.x;
  const uint32_t we = L.wide_lit((int)(xlo & ((1u << LIT_TBITS) - 1)));
  const uint32_t wb1 = (we >> 17) & 63u;
  const uint32_t wx2 = (uint32_t)(c.x >> wb1);
  const uint32_t we2 = L.wide_dist((int)(wx2 & ((1u << DIST_TBITS) - 1)));
  if ((int32_t)we >= 0x20000000 && we2 != 0u && c.avail >= SPAN_SYM_BITS_MAX) {  // a length symbol's entry (a literal's has bit 31)
    length = ((we >> 8) & 511u) + bit_field(xlo, we, (we >> 5) & 7u);
    dist = ((we2 >> 9) & 0xFFFFu) + bit_field(wx2, we2, (we2 >> 5) & 15u);
    c.used = (int)(wb1 + (we2 >> 25));
  } else {
    int sym = read_symbol_walk(c, L, LDS_LIT_COUNTS, LDS_LIT_SYMS);
    if (sym < 0) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    if (sym < LITLEN_EOB) {
      if (d.out_pos >= d.cap_min) { d.overflow((uint64_t)d.out_pos + 1); return SYM_STOP; }
      if (writer) dst[d.out_pos] = (uint8_t)sym;
      d.out_pos++;
      d.advance((uint32_t)c.used);
      return SYM_OK;
    }
    if (sym == LITLEN_EOB) { d.advance((uint32_t)c.used); return SYM_EOB; }
    if (sym > d.lit_max_sym || sym > LITLEN_SYM_MAX) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    uint32_t vbase, vextra, v = 0;
    length_sym_value(sym, vbase, vextra);
    if (vextra != 0 && !c.take((int)vextra, v)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    length = vbase + v;
    int dsym = read_symbol_walk(c, L, LDS_DIST_COUNTS, LDS_DIST_SYMS);
    if (dsym < 0 || dsym > d.dist_max_sym || dsym > DIST_SYM_MAX) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    dist_sym_value(dsym, vbase, vextra);
    v = 0;
    if (vextra != 0 && !c.take((int)vextra, v)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
    dist = vbase + v;
  }
  return lane_match_commit(d, L, writer, c, length, dist, may_queue);
}

// read_symbol on the FIXED codes (fixed_litlen_decoder / fixed_dist_decoder zd.ml:334-349) without
// tables: 7-bit codes 0000000..0010111 are 256..279, 8-bit 00110000..10111111 are 0..143,
// 11000000..11000111 are 280..287, 9-bit 110010000..111111111 are 144..255; codes come most
// significant bit first.  -1: the input ends inside the code.
ZD_HD int fixed_litlen_symbol(BitCursor &c) {
  const uint32_t x = (uint32_t)(c.x >> c.used);
  const int have = c.avail - c.used;
  if (have < 7) return -1;
  const uint32_t r7 = bitrev(x & 127u, 7);
  if (r7 <= 23u) { c.used += 7; return (int)(256u + r7); }
  if (have < 8) return -1;
  const uint32_t r8 = bitrev(x & 255u, 8);
  if (r8 >= 0x30u && r8 <= 0xBFu) { c.used += 8; return (int)(r8 - 0x30u); }
  if (r8 >= 0xC0u && r8 <= 0xC7u) { c.used += 8; return (int)(280u + r8 - 0xC0u); }
  if (have < 9) return -1;
  c.used += 9;
  return (int)(144u + bitrev(x & 511u, 9) - 0x190u);
}
// lane_one_symbol for a fixed block whose tables are not built
ZD_HD int lane_one_symbol_fixed(InflateLane &d, const LaneLds &L, const Arenas &A, bool writer, bool may_queue = true) {
  uint8_t *dst = A.dst + d.dst_off;
  if (!d.input_ready(3)) return SYM_STOP;
  BitCursor c = cursor_at(d, L);
  const int sym = fixed_litlen_symbol(c);
  if (sym < 0) { d.fail(ST_CORRUPTED); return SYM_STOP; }
  if (sym < LITLEN_EOB) {
    if (d.out_pos >= d.cap_min) { d.overflow((uint64_t)d.out_pos + 1); return SYM_STOP; }
    if (writer) dst[d.out_pos] = (uint8_t)sym;
    d.out_pos++;
    d.advance((uint32_t)c.used);
    return SYM_OK;
  }
  if (sym == LITLEN_EOB) { d.advance((uint32_t)c.used); return SYM_EOB; }
  if (sym > LITLEN_SYM_MAX) { d.fail(ST_CORRUPTED); return SYM_STOP; }
  uint32_t vbase, vextra, v = 0;
  length_sym_value(sym, vbase, vextra);
  if (vextra != 0 && !c.take((int)vextra, v)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
  const uint32_t length = vbase + v;
  uint32_t five;
  if (!c.take(5, five)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
  const int dsym = (int)bitrev(five, 5);
  if (dsym > DIST_SYM_MAX) { d.fail(ST_CORRUPTED); return SYM_STOP; }
  dist_sym_value(dsym, vbase, vextra);
  v = 0;
  if (vextra != 0 && !c.take((int)vextra, v)) { d.fail(ST_CORRUPTED); return SYM_STOP; }
  return lane_match_commit(d, L, writer, c, length, vbase + v, may_queue);
}

// One header action: false = must wait for input.
ZD_HD bool lane_header_step(InflateLane &d, const LaneLds &L, const uint8_t *__restrict__ sa) {
  if (d.phase == PH_HEADER) return lane_block_header(d, L, sa);
  if (d.phase == PH_HDR_CODELEN) {  // (the host model's way; its bits were made ready with the header's: HEADER_WORDS)
    if (!setup_codelen_code(d, L)) d.fail(ST_CORRUPTED);
    else d.phase = PH_HDR_LENGTHS;
    return true;
  }
  const int r = setup_dynamic_lengths(d, L);
  if (r == 0) return false;
  if (r < 0) { d.fail(ST_CORRUPTED); return true; }
  d.hdr_fixed = 0;
  d.phase = PH_TABLES;
  return true;
}

// the block's symbols start at the current position
ZD_HD void lane_begin_symbols(InflateLane &d) {
  d.phase = PH_SYMBOLS;
  d.blk_in_word = d.in_word;
  d.blk_boff = d.boff;
  d.span_off = 0;
  d.span_retry_word = 0;
  d.span_fails = 0;
  d.fixed_lazy = 0;
}

// Phase PH_TABLES, serial form (the kernel has a wave-parallel one with the same
// results, inflate.hip): the block's two decoders and their primary tables.
ZD_HD void lane_finish_tables(InflateLane &d, const LaneLds &L) {
  if (d.hdr_fixed) setup_fixed(d, L);
  else {
    if (!init_decoder(L, LDS_LIT_COUNTS, LDS_LIT_SYMS, LDS_LIT_TBL, 0, d.hdr_hlit, d.lit_max_sym) ||
        !init_decoder(L, LDS_DIST_COUNTS, LDS_DIST_SYMS, LDS_DIST_TBL, d.hdr_hlit, d.hdr_hdist, d.dist_max_sym)) {
      d.fail(ST_CORRUPTED);
      return;
    }
    build_table(L, LDS_LIT_TBL, LIT_TBITS, LDS_LIT_COUNTS, LDS_LIT_SYMS);
    build_table(L, LDS_DIST_TBL, DIST_TBITS, LDS_DIST_COUNTS, LDS_DIST_SYMS);
  }
  lane_begin_symbols(d);
}

// end of block: inflated_block_crc zd.ml:682-690, then the loop test zd.ml:704
ZD_HD void lane_end_of_block(InflateLane &d, bool crc_adler) {
  const uint32_t words = d.in_word - d.blk_in_word;
  d.prev_block_bits = words >= (1u << 17) ? (1u << 22) : words * 32u + d.boff - d.blk_boff;
  if (crc_adler) d.phase = PH_REQ_ADLER;
  else d.phase = d.final_block ? PH_DONE : PH_HEADER;
}

// ---- the wide turn's tables and per-lane decode
//
// The wide tables restate the primary tables with everything a speculative
// lane needs in one word, and with every check that does not depend on the
// stream position already made: an entry is 0 ("stop") unless the wide turn may
// commit the symbol -- a literal, or a length symbol 257..267 (length <= 16 =
// DEFER_MAX_LEN) in the block's alphabet; long codes, end of block, invalid and
// longer length symbols stop the chain and are left to lane_one_symbol, which
// owns every error and edge decision.
//   litlen: [4:0] code bits (bit 4 clear: the word is its own bit-field offset operand)
//           [7:5] extra bits  [16:8] literal byte / base length  [22:17] code + extra bits
//           [29] length symbol 268..285 (not for the wide turn)  [30] length symbol 257..267  [31] literal
//           a LITERAL's entry also tells the span decoder (inflate_span.h) about the literal that
//           follows it when both codes lie inside the table's 9 index bits (wide_lit_pair):
//           [3:0] bits of the one or two literals  [4] two  [30:23] the second literal -- fields the
//           wide turn does not look at in a literal's entry ([22:17] stays the first literal's bits)
//   dist:   [4:0] code bits  [8:5] extra bits  [24:9] base distance  [29:25] code + extra bits
//           (an invalid code is 0: it decodes to distance 0, which no length is <= to)
ZD_HD uint32_t wide_lit_entry(uint32_t e16, int lit_max_sym) {
  const uint32_t len = e16 & 15u, sym = e16 >> 4;
  if (len == 0) return 0;
  if (sym < (uint32_t)LITLEN_EOB) return len | (sym << 8) | (len << 17) | (1u << 31);
  if (sym == (uint32_t)LITLEN_EOB || (int)sym > lit_max_sym || sym > (uint32_t)LITLEN_SYM_MAX) return 0;
  uint32_t base, extra;
  length_sym_value((int)sym, base, extra);
  // 268..285 (length > DEFER_MAX_LEN): bit 29 instead of bit 30 -- a stop for the wide turn
  // like the 0 it used to be, a length symbol for the span decoder (inflate_span.h)
  return len | (extra << 5) | (base << 8) | ((len + extra) << 17) | (sym > 267u ? 1u << 29 : 1u << 30);
}
ZD_HD uint32_t wide_dist_entry(uint32_t e16, int dist_max_sym) {
  const uint32_t len = e16 & 15u, sym = e16 >> 4;
  if (len == 0 || (int)sym > dist_max_sym || sym > (uint32_t)DIST_SYM_MAX) return 0;
  uint32_t base, extra;
  dist_sym_value((int)sym, base, extra);
  return len | (extra << 5) | (base << 9) | ((len + extra) << 25);
}
// A literal's entry at table index i, with the literal after it: the index bits above the first
// code are the start of the next code, and when the primary table resolves them to a literal
// whose code ends inside the index, one lookup decodes both.
ZD_HD uint32_t wide_lit_pair(uint32_t e, int i, const LaneLds &L) {
  const uint32_t len1 = (e >> 17) & 63u;
  uint32_t bits = len1, two = 0, lit2 = 0;
  if (len1 < (uint32_t)LIT_TBITS) {
    const uint32_t e2 = L.u16(LDS_LIT_TBL, i >> len1);  // the unknown bits above read as 0
    const uint32_t len2 = e2 & 15u, sym2 = e2 >> 4;
    if (len2 != 0 && len1 + len2 <= (uint32_t)LIT_TBITS && sym2 < (uint32_t)LITLEN_EOB) {
      bits = len1 + len2;
      two = 1;
      lit2 = sym2;
    }
  }
  return (e & ~0x1Fu) | bits | (two << 4) | (lit2 << 23);
}
// lane `lane` of 64 restates its share of both tables; returns the bits of the
// shortest symbol it saw that a wide turn may commit
ZD_HD uint32_t build_wide_tables(const InflateLane &d, const LaneLds &L, int lane) {
  uint32_t shortest = 15;
#pragma unroll 1
  for (int i = lane; i < (1 << LIT_TBITS); i += 64) {
    uint32_t e = wide_lit_entry(L.u16(LDS_LIT_TBL, i), d.lit_max_sym);
    const uint32_t code_bits = e & 15u;
    if ((int32_t)e < 0) e = wide_lit_pair(e, i, L);
    L.wide_lit(i) = e;
    if (e != 0 && code_bits < shortest) shortest = code_bits;
  }
#pragma unroll 1
  for (int i = lane; i < (1 << DIST_TBITS); i += 64) L.wide_dist(i) = wide_dist_entry(L.u16(LDS_DIST_TBL, i), d.dist_max_sym);
  build_canon(L, lane);
  return shortest;
}
// The offsets 0..62 hold at most 62 / shortest + 1 symbol starts; the wide turn's
// descending search over J[0 .. levels) reaches 2^levels - 1 hops from offset 0.
// (The hop from the last start into the sink, lane 63, is not counted: a path
// without a stop ends there by construction.)
ZD_HD int levels_for(uint32_t shortest) { return shortest >= 4 ? 4 : shortest >= 2 ? 5 : 6; }

// What lane s of the wide turn finds s bits after the position: xlo/xhi = the
// next 64 bits from there.  Branch free: the distance lookup of a literal lane
// reads a valid (ignored) entry.
struct WideSym {
  uint32_t e;       // litlen entry: 0 stop, bit 31 literal, bit 30 length symbol
  uint32_t lit;     // literal byte (literal)
  uint32_t length;  // match length (length symbol)
  uint32_t dist;    // match distance (length symbol; 0 when the distance code is not committable)
  uint32_t b1, t2;  // bits of the litlen part, of the distance part
  // the predicates the turn forms out of these (as wave masks in the kernel)
  ZD_HD bool is_lit() const { return (int32_t)e < 0; }
  ZD_HD bool is_len() const { return (int32_t)e >= 0x40000000; }
  ZD_HD bool is_match() const { return is_len() && dist >= length && dist >= DEFER_MIN_DIST; }  // may be deferred
  ZD_HD uint32_t tot() const { return b1 + (is_lit() ? 0u : t2); }
  ZD_HD uint32_t outlen() const { return is_lit() ? 1u : length; }
};
ZD_HD WideSym wide_decode(uint32_t xlo, uint32_t xhi, const LaneLds &L) {
  WideSym r;
  const uint32_t e = L.wide_lit((int)(xlo & ((1u << LIT_TBITS) - 1)));
  r.e = e;
  r.b1 = (e >> 17) & 63u;
  r.lit = (e >> 8) & 511u;
  r.length = r.lit + bit_field(xlo, e, (e >> 5) & 7u);  // offset operand: the entry's low 5 bits = code bits
  const uint32_t x2 = funnel32(xhi, xlo, r.b1);  // b1 <= 14
  const uint32_t e2 = L.wide_dist((int)(x2 & ((1u << DIST_TBITS) - 1)));
  r.dist = ((e2 >> 9) & 0xFFFFu) + bit_field(x2, e2, (e2 >> 5) & 15u);
  r.t2 = e2 >> 25;
  return r;
}

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
ZD_HD void deferred_load(DeferredCopy &c, const uint8_t *dst, uint32_t base, uint32_t packed) {
  const uint32_t dist = (packed & 0x7FFFu) + 1u, len = ((packed >> 15) & 15u) + 3u, dst_pos = base + (packed >> 19);
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
  d.in_word = 0;
  d.boff = 0;
  d.ring_wr = 0;
  d.out_pos = 0;
  d.status = ST_OK;
  d.phase = PH_HEADER;
  d.final_block = 0;
  d.lit_max_sym = d.dist_max_sym = -1;
  d.blk_out_start = 0;
  d.req_src = d.req_len = d.req_dist = 0;
  d.q_count = 0;
  d.hole_min = 0xFFFFFFFFu;
  d.hdr_num = d.hdr_hlit = d.hdr_hdist = d.hdr_cl_max = 0;
  d.hdr_fixed = 0;
  d.adler = 1;  // Adler_32.init zd.ml:173
  d.levels = 6;
  d.blk_in_word = d.blk_boff = 0;
  d.prev_block_bits = 0;
  d.span_off = 0;
  d.span_retry_word = 0;
  d.span_fails = 0;
  d.fixed_lazy = 0;
  if (s.src_len > MAX_STREAM_LEN || s.dst_cap > MAX_STREAM_LEN) {
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
