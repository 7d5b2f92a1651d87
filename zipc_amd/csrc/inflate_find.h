// inflate_find.h -- is this bit offset the header of a dynamic block?  (inflate.hip: one stream by a wave per block.)
//
// The two tests of the search for block starts, in the reference's terms (zd.ml:638-669 read_code_lengths,
// zd.ml:355-391 Huffman.init_decoder): a real header passes both; of random bits about one offset in 2^30 does.
// They only FILTER: every candidate is decoded by the stream's own wave code afterwards, with every check of the
// reference, and a block whose header is not found is walked by the chain itself.  ZD_HD: tests/host_sim runs these
// on the CPU against streams whose block starts are known.
#pragma once

#include "zd_common.h"

namespace zd {

// the 64 bits from bit `pos` on (zero behind the input)
ZD_HD uint64_t find_bits(const uint8_t *__restrict__ s, uint64_t len, uint64_t pos) {
  const uint64_t b = pos >> 3;
  uint64_t lo = 0, hi = 0;
  if (b + 9u <= len) { lo = load_u64_le(s + b); hi = s[b + 8u]; }
  else {
    for (uint32_t i = 0; i < 8u; i++) if (b + i < len) lo |= (uint64_t)s[b + i] << (8u * i);
  }
  const uint32_t sh = (uint32_t)pos & 7u;
  return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

// The three cheap conditions of find_header_test for 32 offsets at once: bit `off` of the result is set when, `off`
// bits into a (the 64 bits of input from some bit on), the block type is "dynamic" (bits off+1, off+2 = 0, 1) and
// neither 5-bit count is 30 or 31 (their upper four bits not all set): a few word operations instead of 32 tests --
// and what a thread then looks at closely is the 7 of its 32 offsets that pass, not all of them.
ZD_HD uint32_t find_header_mask32(uint64_t a) {
  const uint64_t dyn = ~(a >> 1) & (a >> 2);
  const uint64_t lit_big = (a >> 4) & (a >> 5) & (a >> 6) & (a >> 7);
  const uint64_t dist_big = (a >> 9) & (a >> 10) & (a >> 11) & (a >> 12);
  return (uint32_t)(dyn & ~lit_big & ~dist_big);
}

// x: the 64 bits from the offset on, x_hi: the bits from 64 on (at least 10 of them), left: bits of input from
// the offset on.  The block type, the three counts, and a code-length code that init_decoder accepts.
ZD_HD bool find_header_test(uint64_t x, uint64_t x_hi, uint64_t left) {
  const uint32_t h = (uint32_t)x;
  if (((h >> 1) & 3u) != 2u) return false;                             // zd.ml:697-701: a dynamic block
  if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) return false;  // zd.ml:641
  const uint32_t hclen = ((h >> 13) & 15u) + 4u;
  if (17u + 3u * hclen > left) return false;
  const uint64_t ls = (x >> 17) | (x_hi << 47);  // the up to 57 bits of the code-length code's lengths
  uint32_t kraft = 0, n = 0;
  for (uint32_t i = 0; i < hclen; i++) {
    const uint32_t l = (uint32_t)(ls >> (3u * i)) & 7u;
    if (l) { kraft += 128u >> l; n++; }
  }
  // zd.ml:371-378: not over-subscribed, complete -- or a single code of one bit
  return kraft == 128u || (n == 1u && kraft == 64u);
}

// The code lengths behind a header that passed find_header_test at bit `start`: read with the code-length code,
// they must make a literal/length code that is complete (or a single one-bit code) and has the end-of-block symbol
// (zd.ml:662), and a distance code that is complete, a single one-bit code, or empty.
// max_syms: code-length symbols read at most; a header that has passed so many is let through untested beyond (the
// dry run reads it again anyway).
// tbl: 128 bytes of the caller's (entry i at tbl[i * stride]: the kernel gives every thread a column of LDS) for the
// code-length code's look-up table -- a symbol is then one read, not a walk of the code bit by bit.
ZD_HD bool find_lengths_test(const uint8_t *__restrict__ s, uint64_t len, uint64_t start, uint8_t *tbl, uint32_t stride,
                             uint32_t max_syms = 0xFFFFFFFFu) {
  const uint64_t total_bits = len * 8u;
  uint64_t pos = start;
  uint64_t x = find_bits(s, len, pos);
  const uint32_t hlit = (((uint32_t)x >> 3) & 31u) + 257u, hdist = (((uint32_t)x >> 8) & 31u) + 1u;
  const uint32_t hclen = (((uint32_t)x >> 13) & 15u) + 4u;
  pos += 17u;
  x = find_bits(s, len, pos);
  // the code-length code: lengths by symbol (3 bits each), counts by length (5 bits each), symbols by (length, symbol)
  uint64_t len_of = 0, cnt = 0;
  for (uint32_t k = 0; k < hclen; k++) {
    const uint64_t l = (x >> (3u * k)) & 7u;
    len_of |= l << (3u * k_codelen_order[k]);
    if (l) cnt += 1ull << (5u * (uint32_t)l);
  }
  pos += 3u * hclen;
  // the table: entry = symbol | length << 5 at every 7-bit index whose low `length` bits are the symbol's code as it
  // comes in the stream (read_symbol zd.ml:584-591 takes a code most significant bit first: the index is the code
  // reversed); codes are handed out in the order (length, symbol).  0: no code (an incomplete code's gap).
  for (uint32_t i = 0; i < 128u; i++) tbl[i * stride] = 0;
  {
    uint32_t code = 0;
    for (uint32_t l = 1; l <= 7u; l++) {
      for (uint32_t sym = 0; sym < 19u; sym++)
        if (((len_of >> (3u * sym)) & 7u) == l) {
          uint32_t r = 0;
          for (uint32_t b = 0; b < l; b++) r |= ((code >> b) & 1u) << (l - 1u - b);
          for (uint32_t i = r; i < 128u; i += 1u << l) tbl[i * stride] = (uint8_t)(sym | (l << 5));
          code++;
        }
      code <<= 1;
    }
  }
  (void)cnt;
  const uint32_t total = hlit + hdist;
  uint32_t num = 0, prev = 0;
  uint32_t kraft_lit = 0, n_lit = 0, kraft_dist = 0, n_dist = 0;
  bool has_eob = false;
  uint32_t have = 0;  // bits of x not used yet (a symbol is 14 at most)
  uint32_t n_syms = 0;
  while (num < total) {
    if (n_syms++ >= max_syms) return true;
    if (have < 14u) { x = find_bits(s, len, pos); have = 64; }
    const uint32_t ent = tbl[((uint32_t)x & 127u) * stride];
    if (ent == 0u) return false;
    const uint32_t sym = ent & 31u;
    uint32_t used = ent >> 5;
    uint32_t len, rep;
    if (sym < 16u) { len = sym; rep = 1; }
    else if (sym == 16u) {
      if (num == 0) return false;  // zd.ml:653
      len = prev; rep = 3u + ((uint32_t)(x >> used) & 3u); used += 2u;
    } else if (sym == 17u) { len = 0; rep = 3u + ((uint32_t)(x >> used) & 7u); used += 3u; }
    else { len = 0; rep = 11u + ((uint32_t)(x >> used) & 127u); used += 7u; }
    pos += used;
    x >>= used;
    have -= used;
    if (pos > total_bits || rep > total - num) return false;  // zd.ml:659
    if (len) {
      const uint32_t in_lit = num >= hlit ? 0u : (hlit - num < rep ? hlit - num : rep);
      kraft_lit += in_lit * (32768u >> len);
      n_lit += in_lit;
      kraft_dist += (rep - in_lit) * (32768u >> len);
      n_dist += rep - in_lit;
      if (num <= 256u && 256u < num + rep) has_eob = true;
      if (kraft_lit > 32768u || kraft_dist > 32768u) return false;  // over-subscribed zd.ml:371
    }
    prev = len;
    num += rep;
  }
  if (!has_eob) return false;  // zd.ml:662
  if (!(kraft_lit == 32768u || (n_lit == 1u && kraft_lit == 16384u))) return false;
  if (!(n_dist == 0u || kraft_dist == 32768u || (n_dist == 1u && kraft_dist == 16384u))) return false;
  return true;
}

}  // namespace zd
