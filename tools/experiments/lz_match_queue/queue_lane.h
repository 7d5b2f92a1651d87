// queue_lane.h -- the lane-serial pieces of the queue form of lz_match (experiment, not part of the library): entry packings,
// the candidate step and its host model.  Was a section of zipc_amd/csrc/deflate_lane.h while the form was built and measured.
#pragma once
#include "../../../zipc_amd/csrc/deflate_lane.h"

namespace zd {

template <bool WORDS>
ZD_HD uint32_t link_at(const uint16_t *prev, uint32_t q) { return prev[q]; }

// ---------------------------------------------------------------------------
// The chain walk of lz_match_queue_kernel (lz_match_queue.hip, beside this file), third form: candidates, not positions, are what a
// wave steps through.  A DENSE step takes 64 consecutive positions, one per lane, and compares every position's first
// candidate (link and source of the positions read side by side: no bank conflicts); a position whose chain goes on
// leaves an ENTRY in the wave's queue in LDS -- position, distance of the next candidate, candidates so far, best so
// far: two words -- and a QUEUE step pops 64 entries, compares their candidates and pushes those that go on.  Every
// lane of every step works on a candidate (the run slots of the first two forms idled while their lanes' neighbours
// walked: 0.58 of the lanes busy on the benchmark's symbols), the K and K/4 tests are a compare on the entry, and the
// answers of two positions in three leave in the dense step's coalesced store.
// The same pieces run on the CPU in tests/host_sim (lz_match_queue_serial) against lz_match_position.
//
// Two packings.  K <= 128 (`Fast, `Default): positions relative to the tile's first staged position w0 (the LDS
// addresses of the step's reads, < 2^16) and the candidates LEFT instead of those looked at, so that the step's tests are
// compares on the packed word:
//   a = (next candidate - w0) | (p - w0) << 16        b = word | left << 25   (word: dist << 9 | len of the best so far, 0: none;
//                                                                               left = K - steps, 1..127)
// K = 4096 (`Best): 12 bits of steps do not fit beside a 25-bit word:
//   a = (p - t0) | (dist - 1) << 14 | (steps & 7) << 29     b = (bdist - 1) | (blen - 3) << 15 | (steps >> 3) << 23
struct QEntry { uint32_t a, b; };
constexpr uint32_t QE_TILE_MAX = 1u << 14;  // positions of a tile at most (BIGK's 14 bits; the other packing: window + tile < 2^16)
// the table word of a best match: dist << 9 | len, 0 for none (the reference's backref, zd.ml:766-771)
ZD_HD uint32_t qe_word(uint32_t blen, uint32_t bdist) { return blen > 3u ? (bdist << 9) | blen : 0u; }
template <bool BIGK>
ZD_HD QEntry qe_pack(uint32_t w0, uint32_t t0, uint32_t K, uint32_t p, uint32_t dist, uint32_t steps, uint32_t blen, uint32_t bdist) {
  QEntry e;
  if (BIGK) {
    e.a = (p - t0) | ((dist - 1u) << 14) | ((steps & 7u) << 29);
    e.b = ((bdist - 1u) & 0x7FFFu) | ((blen - 3u) << 15) | ((steps >> 3) << 23);
  } else {
    e.a = (p - dist - w0) | ((p - w0) << 16);
    e.b = qe_word(blen, bdist) | ((K - steps) << 25);
  }
  return e;
}
template <bool BIGK>
ZD_HD void qe_unpack(QEntry e, uint32_t w0, uint32_t t0, uint32_t K, uint32_t &p, uint32_t &dist, uint32_t &steps, uint32_t &blen,
                     uint32_t &bdist) {
  if (BIGK) {
    p = t0 + (e.a & 0x3FFFu);
    dist = ((e.a >> 14) & 0x7FFFu) + 1u;
    steps = (e.a >> 29) | ((e.b >> 23) << 3);
    blen = ((e.b >> 15) & 0xFFu) + 3u;
    bdist = (e.b & 0x7FFFu) + 1u;
  } else {
    p = w0 + (e.a >> 16);
    dist = (e.a >> 16) - (e.a & 0xFFFFu);
    steps = K - (e.b >> 25);
    const uint32_t l = e.b & 0x1FFu;
    blen = l > 3u ? l : 3u;
    bdist = (e.b >> 9) & 0xFFFFu;  // (0 with no match: never used then)
  }
}

// 8 bytes at s + pos from two ALIGNED 8-byte words (s 8-byte aligned, up to 15 bytes past pos are touched): in LDS a
// random 8-byte read costs what a random 4-byte read does, so 16 bytes come for the price of 8 of load_u64_words' 12
ZD_HD uint64_t load_u64_pair(const uint8_t *s, uint32_t pos) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 *w = (const u32x2 *)(s + (pos & ~7u));
  const u32x2 w0 = w[0], w1 = w[1];
  const bool up = (pos & 4u) != 0;
  const uint32_t e0 = up ? w0.y : w0.x, e1 = up ? w1.x : w0.y, e2 = up ? w1.y : w1.x;
  const uint32_t lo = __builtin_amdgcn_alignbyte(e1, e0, pos), hi = __builtin_amdgcn_alignbyte(e2, e1, pos);
  return ((uint64_t)hi << 32) | lo;
#else
  return load_u64_le(s + pos);
#endif
}

// how the step reads 8 bytes of the window: as they lie (global memory, the host model), from three aligned 4-byte words
// (load_u64_words: 3 vector instructions, 12 bytes of LDS) or from two aligned 8-byte words (load_u64_pair: 8, 16 bytes at
// two thirds of the LDS clocks)
enum : int { LD_PLAIN = 0, LD_WORDS = 1, LD_PAIR = 2 };
template <int LD>
ZD_HD uint64_t ld8(const uint8_t *s, uint32_t pos) {
  return LD == LD_PAIR ? load_u64_pair(s, pos) : LD == LD_WORDS ? load_u64_words(s, pos) : load_u64_le(s + pos);
}

// One candidate of one position: the step both kinds of wave step are made of.
//   in:  p, the candidate's distance dist (valid: 1..32768, within the chain's first K), steps candidates looked at
//        before it, the best so far (blen 3: none), pw = s[p .. p+8)
//   out: the same after the candidate; returns true when the walk goes on (dist is then the NEXT candidate's)
// zd.ml:1176-1201: longest, then nearest; the walk ends with the chain, at K candidates, beyond the window, or when a
// candidate reaches maxlen (nothing later can be longer, zd.ml:1194).
template <int LD>
ZD_HD bool queue_candidate(const uint8_t *s, const uint16_t *prev, uint32_t len, uint32_t p, uint64_t pw, uint32_t K,
                           uint32_t &dist, uint32_t &steps, uint32_t &blen, uint32_t &bdist) {
  const uint32_t q = p - dist;
  const uint64_t cw = ld8<LD>(s, q);
  const uint32_t dn = link_at<LD != LD_PLAIN>(prev, q);
  const uint32_t rest = len - p;
  const uint32_t maxlen = rest < (uint32_t)MAX_MATCH_LEN ? rest : (uint32_t)MAX_MATCH_LEN;
  const uint64_t x = cw ^ pw;
  uint32_t l = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
  l = l < maxlen ? l : maxlen;  // (maxlen < 8: the last positions of the stream; bytes behind it do not count)
  if (x == 0 && maxlen > 8u) {
    // the first 8 bytes agree.  Before the long compare: a candidate that cannot beat blen anyway -- the 8 bytes that END
    // at offset blen differ -- is at most blen long (on text 9 in 10 of those that get here); its length is not needed
    bool compare = true;
    if (blen >= 8u) {  // blen < maxlen here (else the walk had ended): these bytes lie inside both strings
      const uint32_t toff = blen - 7u;
      compare = ld8<LD>(s, q + toff) == ld8<LD>(s, p + toff);
    }
    if (compare) {
      l = maxlen;
      for (uint32_t i = 8; i < maxlen; i += 8) {  // (reads up to 15 bytes past p + 256: the window holds them; bytes past maxlen are cut off below)
        const uint64_t y = ld8<LD>(s, q + i) ^ ld8<LD>(s, p + i);
        if (y) {
          const uint32_t at = i + (uint32_t)(__builtin_ctzll(y) >> 3);
          l = at < maxlen ? at : maxlen;
          break;
        }
      }
    }
  }
  steps += 1u;
  if (l > blen) { blen = l; bdist = dist; }
  dist += dn;
  return dn != 0 && dist <= (uint32_t)MAX_MATCH_DIST && l != maxlen && steps != K;
}

// Host model of the queue form over positions [pbeg, pend) of one tile that starts at t0 and whose window starts at w0
// (test tooling; the device runs the same queue_candidate and packings from lz_match_queue_kernel): dense steps over the
// positions, entries through a plain FIFO in their packed form, the two answers written the way the kernel writes
// them -- the best of the first K/4 when the walk passes that candidate and goes on, the final answer into both halves
// otherwise, into the low half only behind such a store.
template <int LD, bool BIGK>
ZD_HD void lz_match_queue_serial(const uint8_t *s, uint32_t len, uint32_t w0, uint32_t t0, uint32_t pbeg, uint32_t pend,
                                 const uint16_t *prev, int K, int Kq, uint64_t *out, QEntry *fifo, uint32_t cap) {
  uint32_t head = 0, tail = 0;  // fifo: a ring of cap > pend - pbeg entries (a position has at most one entry)
  auto settle = [&](uint32_t p, bool more, uint32_t dist, uint32_t steps, uint32_t blen, uint32_t bdist) {
    const uint32_t word = qe_word(blen, bdist);
    if (more && steps == (uint32_t)Kq) out[p] = (out[p] & 0xFFFFFFFFull) | ((uint64_t)word << 32);
    if (more) { fifo[tail] = qe_pack<BIGK>(w0, t0, (uint32_t)K, p, dist, steps, blen, bdist); tail = tail + 1 == cap ? 0 : tail + 1; }
    else if (steps > (uint32_t)Kq) out[p] = (out[p] & ~0xFFFFFFFFull) | word;
    else out[p] = (uint64_t)word | ((uint64_t)word << 32);
  };
  for (uint32_t p = pbeg; p < pend; p++) {
    uint32_t dist = link_at<LD != LD_PLAIN>(prev, p), steps = 0, blen = MIN_MATCH_LEN - 1, bdist = 1;
    bool more = false;
    if (dist != 0 && K > 0) more = queue_candidate<LD>(s, prev, len, p, ld8<LD>(s, p), (uint32_t)K, dist, steps, blen, bdist);
    settle(p, more, dist, steps, blen, bdist);
  }
  while (head != tail) {
    uint32_t p, dist, steps, blen, bdist;
    qe_unpack<BIGK>(fifo[head], w0, t0, (uint32_t)K, p, dist, steps, blen, bdist);
    head = head + 1 == cap ? 0 : head + 1;
    const bool more = queue_candidate<LD>(s, prev, len, p, ld8<LD>(s, p), (uint32_t)K, dist, steps, blen, bdist);
    settle(p, more, dist, steps, blen, bdist);
  }
}


}  // namespace zd
