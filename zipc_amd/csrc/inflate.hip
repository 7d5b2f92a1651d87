// inflate.hip -- batch inflate kernel for gfx950 (CDNA4, wave64).
//
// Decomposition: ONE WAVEFRONT PER STREAM (one 64-thread workgroup each).  A
// deflate stream is a serial chain -- the bit position of symbol k+1 depends on
// symbol k -- so the wave shortens the chain instead of running many of them:
//
//   span        a compressed block's symbols: a region of the bits ahead per lane, region
//               starts settled by letting the walks self-synchronise, output assembled in
//               LDS tiles (inflate_span.h).  Nearly all of a stream's time; what follows is
//               what a span cannot take: headers, the end of every block, the last bytes of
//               the input, runs of long matches, stored blocks.
//   wide turn   lane s decodes the whole symbol (literal, or length + distance
//               with their extra bits) that WOULD start s bits after the
//               stream's position: 64 speculative decodes, each three words of
//               the input ring and two table lookups, branch free
//               (wide_decode, inflate_lane.h).  The offsets where symbols
//               really start form a path 0 -> 0 + bits(0) -> ...; it is found
//               without walking it: J_k[s] = offset reached from s after 2^k
//               symbols (5 rounds of ds_bpermute, offsets kept as bpermute
//               addresses so a round is one instruction), then every lane
//               finds by a descending search whether it lies on the path.
//               Lane 63 is the path's sink.  An exclusive wave scan (DPP) of
//               the output lengths over the lanes on the path gives every
//               symbol its output position: literal lanes store their byte,
//               match lanes append {dst, dist, len} to the deferred-copy queue
//               in LDS.  About 8 symbols are committed per turn on 4-bit data.
//   plain step  whatever a wide turn must not commit (long codes, end of block,
//               matches that cannot be deferred, hazards, the last bytes of the
//               input) stops the path; lane_one_symbol (inflate_lane.h) decodes
//               that one symbol with all the reference's checks.
//   service     everything that needs global loads, by all 64 lanes: fill the
//               queued match copies (up to 64 in flight), copy long/overlapping
//               matches, copy stored blocks 16 B per lane
//               (read_uncompressed_block zd.ml:671-680), per-block Adler-32 with
//               the reference's 5552-byte chunking (inflated_block_crc
//               zd.ml:682-690), refill the input ring with one coalesced load.
//
// The stream state is wave-uniform; it is pinned to scalar registers with
// readfirstlane so that the control flow around the turns is scalar branches,
// not exec-mask arithmetic.  LDS: 10072 B per stream (inflate_lane.h has the map): 16 streams per CU.
//
// ONE long stream handed over alone gets a wave per BLOCK instead (further down: "ONE stream by a wave per BLOCK"):
// the same wave code in two more modes -- IM_DRY walks a block for its end and size, IM_TOKEN stores its literals
// and writes down what its matches copy -- around a search for block headers, a chain of the blocks and pointer
// jumping over the copies.
#include "inflate_lane.h"
#include "inflate_span.h"
#include "inflate_find.h"
#include "kernels.h"
#include "wave_ops.h"

namespace zd {

static_assert(LDS_BYTES_PER_LANE == INFLATE_LDS_BYTES_PER_LANE, "kernels.h");
static_assert(SPEC_WINDOW == 64 && QUEUE_ENTRIES <= 64, "one lane per offset / per queue entry");
constexpr int ROUND_TURNS = 24;  // wide turns between two service points

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni(uint64_t v) {
  return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

// every field of the (wave-uniform) state back into scalar registers
__device__ __forceinline__ void uniformize(InflateLane &d) {
#define ZD_U(x) d.x = uni(d.x)
  ZD_U(src_off); ZD_U(dst_off); ZD_U(src_len); ZD_U(in_word); ZD_U(boff); ZD_U(ring_wr);
  ZD_U(out_pos); ZD_U(cap_min); ZD_U(limit); ZD_U(hard_cap); ZD_U(status);
  ZD_U(phase); ZD_U(final_block); ZD_U(lit_max_sym); ZD_U(dist_max_sym); ZD_U(blk_out_start);
  ZD_U(req_src); ZD_U(req_len); ZD_U(req_dist); ZD_U(q_count); ZD_U(hole_min); ZD_U(hdr_num);
  ZD_U(hdr_hlit); ZD_U(hdr_hdist); ZD_U(hdr_cl_max); ZD_U(hdr_hclen); ZD_U(hdr_fixed); ZD_U(adler); ZD_U(levels);
  ZD_U(blk_in_word); ZD_U(blk_boff); ZD_U(prev_block_bits); ZD_U(span_off); ZD_U(span_retry_word); ZD_U(span_fails); ZD_U(fixed_lazy);
#undef ZD_U
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t u = (uint32_t)__shfl_xor((int)v, o, 64);
    v = u < v ? u : v;
  }
  return uni(v);
}

// a - b if a > b, else 0 (v_sub_u32 ... clamp)
__device__ __forceinline__ uint32_t over(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }

// lanes where p holds, as a mask -- and back
__device__ __forceinline__ unsigned long long wave_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }


// ---- phase PH_TABLES by the whole wave (serial form: lane_finish_tables)
//
// Huffman.init_decoder zd.ml:355-391 on lengths[start .. start + n): the counts
// per length are an LDS histogram, the verdicts depend on the counts alone, and
// the symbols sorted by (length, symbol) get their slot from a ballot per length
// present in each group of 64 symbols.  Same arrays as init_decoder.
__device__ __forceinline__ bool wave_init_decoder(const LaneLds &L, int counts_off, int syms_off, int scratch_off,
                                                  int start, int n, int32_t &max_sym, int lane) {
  if (lane < 16) L.u16(counts_off, lane) = 0;
  int my_max = -1;
  for (int i = lane; i < n; i += 64) {
    const int len = L.u16(LDS_LENGTHS, start + i);
    if (len != 0) {
      my_max = i;
      // counts are u16: add into the containing word (no carry: counts <= 288)
      const int at = counts_off + len;
      atomicAdd((uint32_t *)(L.w + (at & ~1)), (at & 1) ? 0x10000u : 1u);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int u = __shfl_xor(my_max, o, 64);
    my_max = u > my_max ? u : my_max;
  }
  max_sym = uni(my_max);
  int available = 1, num_codes = 0;
  bool ok = true;
#pragma unroll 1
  for (int i = 0; i < 16; i++) {  // every lane, same values
    const int used = L.u16(counts_off, i);
    if (used > available) ok = false;  // over-subscribed zd.ml:371
    available = 2 * (available - used);
    if (lane == 0) L.u16(scratch_off, i) = (uint16_t)num_codes;
    num_codes += used;
  }
  if (!uni((uint32_t)ok)) return false;
  if ((num_codes > 1 && available > 0) || (num_codes == 1 && L.u16(counts_off, 1) != 1)) return false;  // zd.ml:377-378
#pragma unroll 1
  for (int c = 0; c < n; c += 64) {
    const int i = c + lane;
    const int len = i < n ? (int)L.u16(LDS_LENGTHS, start + i) : 0;
    unsigned long long todo = wave_mask(len != 0);
#pragma unroll 1
    while (todo) {
      const int l = __builtin_amdgcn_readlane(len, __ffsll((long long)todo) - 1);
      const unsigned long long m = wave_mask(len == l);
      const uint32_t first = L.u16(scratch_off, l);
      if (len == l)
        L.u16(syms_off, (int)(first + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)))) = (uint16_t)i;
      if (lane == 0) L.u16(scratch_off, l) = (uint16_t)(first + (uint32_t)__popcll(m));
      todo &= ~m;
    }
  }
  if (num_codes == 1 && lane == 0) {  // zd.ml:389-390: phantom code 1 -> too-large symbol
    L.u16(counts_off, 1) = 2;
    L.u16(syms_off, 1) = (uint16_t)(max_sym + 1);
  }
  return true;
}

// build_table: every entry walks the canonical code for its own index -- the
// reference's read_symbol (zd.ml:584-591) on the entry's bits
__device__ __forceinline__ void wave_build_table(const LaneLds &L, int tbl_off, int tbits, int counts_off, int syms_off,
                                                 int lane) {
#pragma unroll 1
  for (int e = lane; e < (1 << tbits); e += 64) {
    int base = 0, offs = 0;
    uint32_t ent = 0;
#pragma unroll 1
    for (int len = 1; len <= tbits; len++) {
      offs = 2 * offs + ((e >> (len - 1)) & 1);
      const int count = L.u16(counts_off, len);
      if (offs < count) { ent = ((uint32_t)L.u16(syms_off, base + offs) << 4) | (uint32_t)len; break; }
      base += count;
      offs -= count;
    }
    L.u16(tbl_off, e) = (uint16_t)ent;
  }
}

// The decoders and primary tables of a block's header, by the wave, ONE instance of the code for all
// three of a dynamic header's codes (inlined per use it cost the symbol loops their registers):
// phase PH_HDR_CODELEN -- the code-length code (read_codelen_code zd.ml:624-636: its up to 19 three-bit
// lengths are read by 19 lanes at once; it lives in the dist regions while the header is read) --
// and phase PH_TABLES -- the litlen and the distance code, from the lengths or the fixed ones
// (zd.ml:334-349).  One lane doing the code-length code's part was most of a dynamic header's time.
__device__ __forceinline__ void wave_tables(InflateLane &d, const LaneLds &L, int lane) {
  const bool cl = d.phase == PH_HDR_CODELEN;
  if (cl) {
    const uint32_t need = 3u * (uint32_t)d.hdr_hclen;
    if (need > d.bits_left()) { d.fail(ST_CORRUPTED); return; }
    if (lane < 19) L.u16(LDS_LENGTHS, lane) = 0;
    if (lane < d.hdr_hclen) {
      const uint32_t p = d.boff + 3u * (uint32_t)lane, w = d.in_word + (p >> 5);
      const uint32_t w0 = L.slot((int)(w & (RING_WORDS - 1))), w1 = L.slot((int)((w + 1u) & (RING_WORDS - 1)));
      L.u16(LDS_LENGTHS, k_codelen_order[lane]) = (uint16_t)(funnel32(w1, w0, p & 31u) & 7u);
    }
    d.advance(need);
  } else if (d.hdr_fixed) {
    if (lane < 16) {
      L.u16(LDS_LIT_COUNTS, lane) = lane == 7 ? 24 : lane == 8 ? 152 : lane == 9 ? 112 : 0;
      L.u16(LDS_DIST_COUNTS, lane) = lane == 5 ? 32 : 0;
    }
    for (int i = lane; i < 288; i += 64)
      L.u16(LDS_LIT_SYMS, i) = (uint16_t)(i < 24 ? 256 + i : i < 168 ? i - 24 : i < 176 ? 112 + i : i - 32);
    if (lane < 32) L.u16(LDS_DIST_SYMS, lane) = (uint16_t)lane;
    d.lit_max_sym = LITLEN_SYM_MAX;  // 286 and 287 are unused
    d.dist_max_sym = DIST_SYM_MAX;   // 30 and 31 are unused
  }
  // jobs: 0 the code-length code, 1 litlen, 2 distance; all decoders first (the lengths lie in the
  // litlen table's place), then the tables
  const int first = cl ? 0 : 1, last = cl ? 0 : 2;
  if (cl || !d.hdr_fixed) {
#pragma unroll 1
    for (int job = first; job <= last; job++) {
      const bool lit = job == 1;
      int32_t max_sym;
      if (!wave_init_decoder(L, lit ? LDS_LIT_COUNTS : LDS_DIST_COUNTS, lit ? LDS_LIT_SYMS : LDS_DIST_SYMS,
                             lit ? LDS_LIT_TBL : LDS_DIST_TBL, job == 2 ? d.hdr_hlit : 0,
                             job == 0 ? 19 : lit ? d.hdr_hlit : d.hdr_hdist, max_sym, lane) ||
          (job == 0 && max_sym == -1)) {  // zd.ml:635
        d.fail(ST_CORRUPTED);
        return;
      }
      if (job == 0) d.hdr_cl_max = max_sym;
      else if (lit) d.lit_max_sym = max_sym;
      else d.dist_max_sym = max_sym;
    }
  }
#pragma unroll 1
  for (int job = first; job <= last; job++) {
    const bool lit = job == 1;
    wave_build_table(L, lit ? LDS_LIT_TBL : LDS_DIST_TBL, lit ? LIT_TBITS : DIST_TBITS, lit ? LDS_LIT_COUNTS : LDS_DIST_COUNTS,
                     lit ? LDS_LIT_SYMS : LDS_DIST_SYMS, lane);
  }
  if (cl) {
    d.hdr_num = 0;
    d.phase = PH_HDR_LENGTHS;
  } else {
    lane_begin_symbols(d);
  }
}

// Phase PH_HDR_LENGTHS by the wave: the code lengths of a dynamic header (read_code_lengths zd.ml:638-669;
// setup_dynamic_lengths, inflate_lane.h, is the serial form -- one lane, a symbol per turn of three dependent LDS
// reads: 6.6 % of a stream of the benchmark's symbols, 8.8 % on text).  Like the wide turn: lane i decodes the
// code-length symbol that would start i bits ahead, the symbols that do start are the chain from lane 0 (pointer
// doubling), and then everything the serial loop carries from symbol to symbol is a scan over the chain -- where a
// symbol's lengths go (sum of the repeats before it), how many bits were used before it, and the length a
// "repeat the previous" symbol repeats (the nearest earlier symbol that has a length of its own).  Every check of
// the serial loop is made for every symbol that is needed (those that start below the header's count), in the
// same terms; any of them failing is the same ST_CORRUPTED.  Returns 0: waits for input, -1: corrupt, 1: done.
__device__ __forceinline__ int wave_dynamic_lengths(InflateLane &d, const LaneLds &L, int lane) {
  const uint32_t total = (uint32_t)(d.hdr_hlit + d.hdr_hdist);
  const uint32_t lane4 = (uint32_t)lane * 4u;
  for (;;) {
    uint32_t num = (uint32_t)d.hdr_num;
    if (num >= total) break;
    if (!d.input_ready(4)) return 0;  // 64 + 14 bits from a position inside a word
    const uint32_t left = d.bits_left();
    const uint32_t prev = num > 0 ? (uint32_t)L.u16(LDS_LENGTHS, (int)num - 1) : 0u;
    const uint32_t p = d.boff + (uint32_t)lane, w = d.in_word + (p >> 5);
    const uint32_t w0 = L.slot((int)(w & (RING_WORDS - 1))), w1 = L.slot((int)((w + 1u) & (RING_WORDS - 1)));
    const uint32_t x = funnel32(w1, w0, p & 31u);
    const uint32_t e = L.u16(LDS_DIST_TBL, (int)(x & ((1u << DIST_TBITS) - 1)));
    const uint32_t len = e & 15u, sym = e >> 4;
    const bool no_code = len == 0u || (int)sym > d.hdr_cl_max;  // zd.ml:649
    const uint32_t extra = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
    const uint32_t used = no_code ? 64u : len + extra;
    const uint32_t v = (x >> len) & ((1u << extra) - 1u);
    const uint32_t repeat = sym < 16u ? 1u : sym == 16u ? 3u + v : (sym == 17u ? 3u : 11u) + v;
    // the chain of symbol starts from lane 0 (a symbol that ends behind the 64 bits, or is no code, ends it)
    uint32_t J[7];
    J[0] = (uint32_t)lane + used < 64u ? ((uint32_t)lane + used) * 4u : lane4;
#pragma unroll
    for (int k = 1; k < 7; k++) J[k] = lane_value(J[k - 1], J[k - 1]);
    uint32_t at = 0;
#pragma unroll
    for (int k = 6; k >= 0; k--) {
      const uint32_t y = lane_value(at, J[k]);
      if (y <= lane4) at = y;
    }
    const bool on = at == lane4;
    // where my lengths go, what was used before me
    const uint32_t rep = on && !no_code ? repeat : 0u, rincl = wave_scan_incl(rep);
    const uint32_t start = num + rincl - rep;
    const uint32_t bits = on && !no_code ? used : 0u, bincl = wave_scan_incl(bits);
    const bool needed = on && start < total;  // the serial loop would get to this symbol
    const bool err = needed && (no_code || used > left - (bincl - bits) ||  // (no wrap: an earlier symbol that used more than was left has failed)
                                (sym == 16u && start == 0u) ||                // zd.ml:653
                                repeat > total - start);                      // zd.ml:659 (may span litlen / dist)
    // (a symbol behind a failing one sees a `left` that may have wrapped: it does not matter, the call fails)
    if (__builtin_amdgcn_ballot_w64(err)) return -1;
    // the length a symbol fills in: its own, 0, or that of the nearest earlier symbol with one of its own
    const unsigned long long has_own = __builtin_amdgcn_ballot_w64(needed && sym != 16u);
    const uint32_t own = sym < 16u ? sym : 0u;
    const unsigned long long before = has_own & ((1ull << lane) - 1ull);
    const uint32_t from_lane = before ? 63u - (uint32_t)__builtin_clzll(before) : 0u;
    const uint32_t theirs = lane_value(from_lane * 4u, own);
    const uint32_t fill = sym != 16u ? own : before ? theirs : prev;
    if (needed)
      for (uint32_t r = 0; r < repeat; r++) L.u16(LDS_LENGTHS, (int)(start + r)) = (uint16_t)fill;
    // the bits of the needed symbols, the lengths they made
    const unsigned long long nm = __builtin_amdgcn_ballot_w64(needed);
    const int last = 63 - __builtin_clzll(nm);  // (lane 0 is always needed here: num < total)
    const uint32_t used_all = (uint32_t)__builtin_amdgcn_readlane((int)bincl, last);
    const uint32_t made = (uint32_t)__builtin_amdgcn_readlane((int)rincl, last);
    d.advance(used_all);
    d.hdr_num = (int32_t)(num + made);
    wv::sync();  // the lengths written are read by other lanes (the next turn's `prev`, the tables)
  }
  if (L.u16(LDS_LENGTHS, 256) == 0) return -1;  // zd.ml:662
  return 1;  // the two decoders (zd.ml:663-666) are built in phase PH_TABLES
}

// A match that could not be queued (overlapping, long, or reading a hole) copied by the
// whole wave.  Buf.recopy's bytewise overlapped copy (zd.ml:63-75) produces the periodic
// extension of the last dist bytes, so output byte i is src[i mod dist]: no byte depends
// on another new one and all of them move at once.  (Copied by the writer lane alone, a
// byte at a time through memory when dist < 8, a 258-byte match took 28 us: a 1 MiB run
// of one byte inflated in 114 ms.)
// `pattern`: 80 bytes of LDS nobody uses while symbols are decoded one by one (the span's tile).
constexpr uint32_t MATCH_RUN_DIST = 64;  // periods below this are laid out in LDS once, longer ones copied from where they are
__device__ __forceinline__ void wave_copy_match(uint8_t *dst, uint32_t pos, uint32_t dist, uint32_t len, int lane,
                                                uint8_t *pattern) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the source bytes were stored by other lanes of this wave
  uint8_t *o = dst + pos;
  const uint8_t *s = o - dist;
  const uint32_t l = (uint32_t)lane;
  if (dist < MATCH_RUN_DIST && dist < len) {
    // a short period, any length (match_run: up to 65 matches of 258): the period laid out once,
    // 80 bytes of it, then every lane stores 16-byte pieces of it, each from where its piece
    // starts in the period
    pattern[l] = s[l % dist];
    if (l < 16u) pattern[64u + l] = s[(64u + l) % dist];
    __builtin_amdgcn_wave_barrier();
    const uint32_t step = 1024u % dist;
    uint32_t off = (16u * l) % dist;
    for (uint32_t i = 16u * l; i < len; i += 1024u) {
      wv::Quad q;
      q.x = load_u32_le(pattern + off);
      q.y = load_u32_le(pattern + off + 4u);
      q.z = load_u32_le(pattern + off + 8u);
      q.w = load_u32_le(pattern + off + 12u);
      if (i + 16u <= len) wv::store_quad(o + i, q);
      else
        for (uint32_t j = 0; i + j < len; j++) o[i + j] = pattern[off + j];
      off += step;
      if (off >= dist) off -= dist;
    }
    __builtin_amdgcn_wave_barrier();  // (the next pattern is not written before this one is read)
    return;
  }
  // A longer period, or no repetition at all (dist >= len): every byte is one of the `dist` bytes before
  // the match, byte i the (i mod dist)-th.  The copy goes repetition by repetition in 16-byte pieces
  // that start at multiples of 16 inside the period -- the last piece of a period moved back to end
  // with it, over bytes its neighbour writes too, with the same values -- so no piece wraps, every
  // load reads bytes that were there before the match, and nothing is read or written outside
  // [pos - dist, pos + len).
  const uint32_t P = dist < len ? dist : len;  // bytes of one repetition
  if (P < 16u) {                               // (dist >= 64 here: a short match, once)
    if (l < len) o[l] = s[l];
    return;
  }
  const uint32_t cpp = (P + 15u) >> 4;  // pieces per repetition
  uint32_t r = l / cpp, c = l - r * cpp;
  const uint32_t dr = 64u / cpp, dc = 64u - dr * cpp;
  for (;;) {
    const uint32_t j = c * 16u + 16u <= P ? c * 16u : P - 16u;
    const uint32_t i = r * dist + j;
    if (wave_mask(i < len) == 0ull) break;  // (i grows with the lane: lane 0 is the last to leave)
    if (i < len) {
      const wv::Quad q = wv::load_quad(s + j);
      if (i + 16u <= len) wv::store_quad(o + i, q);
      else {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (uint32_t k = 0; k < 16u; k++)
          if (i + k < len) o[i + k] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
      }
    }
    c += dc;
    r += dr;
    if (c >= cpp) { c -= cpp; r++; }
  }
}

// The match lane_one_symbol has just handed to the wave (d.req_len bytes from d.req_dist back, its
// symbol was `bits` long): how often is the next symbol the same match again?  Lane k decodes what
// starts k symbols of that length from here; as far as they all are that match, the copies are one
// periodic copy (zeros, a repeated word or record: 258 bytes per symbol, a symbol at a time they cost
// a turn and a memory round trip each).  Moves the position over them and returns the bytes of
// the whole run; nothing that needs a check of its own is passed over: the symbols are matches
// of a distance already accepted, inside the input, their bytes within the output's limit.
__device__ __forceinline__ uint32_t match_run(InflateLane &d, const LaneLds &L, int lane, uint32_t bits) {
  const uint32_t k = (uint32_t)lane;
  const uint32_t p = d.boff + k * bits;
  const uint32_t wi = p >> 5;
  const bool staged = wi + 3u <= d.ring_wr - d.in_word;   // the three words a decode reads are in the ring
  const bool inside = (k + 1u) * bits <= d.bits_left();   // ... and the symbol is real input
  const uint32_t w = d.in_word + (staged ? wi : 0u);
  const uint32_t w0 = L.slot((int)(w & (RING_WORDS - 1))), w1 = L.slot((int)((w + 1u) & (RING_WORDS - 1))),
                 w2 = L.slot((int)((w + 2u) & (RING_WORDS - 1)));
  const WideSym sp = wide_decode(funnel32(w1, w0, p), funnel32(w2, w1, p), L);
  const bool same = staged && inside && (int32_t)sp.e >= 0x20000000 && sp.length == d.req_len && sp.dist == d.req_dist &&
                    sp.b1 + sp.t2 == bits;
  const unsigned long long m = wave_mask(same);
  uint32_t reps = ~m == 0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
  const uint32_t room = (d.cap_min - d.out_pos) / d.req_len;  // (the first one fits: lane_match_commit)
  if (reps + 1u > room) reps = room - 1u;
  d.advance(reps * bits);
  return d.req_len * (1u + reps);
}

// One wide turn.  Returns the lane the path was cut at: below 63 it stopped inside
// the window and the symbol at the new position is for lane_one_symbol.
//
// LEVELS: the path has at most 63 / (shortest code of the block) symbols, so
// 2^LEVELS - 1 hops of doubling are enough (InflateLane::levels).
// PLENTY: the input does not end within the turn's reach, no lane can run out of bits.
template <int LEVELS, bool PLENTY, int MODE>
__device__ __forceinline__ uint32_t wide_turn(InflateLane &d, const LaneLds &L, uint8_t *__restrict__ dst, uint32_t *__restrict__ tok, int lane) {
  typedef unsigned long long mask_t;  // one bit per lane; the predicates of the turn are kept as masks
  // the symbol that would start at my offset
  const uint32_t p = d.boff + (uint32_t)lane;
  const int slot = (int)(d.in_word & (uint32_t)(RING_WORDS - 1)) + (int)(p >> 5);
  const uint32_t w0 = L.slot(slot), w1 = L.slot(slot + 1), w2 = L.slot(slot + 2);
  const WideSym sp = wide_decode(funnel32(w1, w0, p), funnel32(w2, w1, p), L);
  const bool is_lit = (int32_t)sp.e < 0;
  const mask_t lit_m = wave_mask(is_lit);
  // (a match from very close by mostly copies the match before it: queued, each would cost the queue's
  // round trip through memory; left to lane_one_symbol they go to the wave at once, runs of them as one copy)
  const mask_t match_m = wave_mask((int32_t)sp.e >= 0x40000000) & wave_mask(sp.dist >= sp.length && sp.dist >= DEFER_MIN_DIST);
  const uint32_t tot = sp.b1 + (is_lit ? 0u : sp.t2);
  const uint32_t outlen = is_lit ? 1u : sp.length;
  // Lane 63 is the path's sink (the next turn starts there): its J[0] below is itself
  // whatever it decodes, and it never commits because the cut is at 63 at the latest.
  mask_t ok_m = lit_m | match_m;
  if (!PLENTY) ok_m &= wave_mask((int)tot <= (int)d.bits_left() - lane);
  const uint32_t lane4 = (uint32_t)lane * 4u;
  const uint32_t end = (uint32_t)lane + tot;
  // J[k]: offset reached after 2^k symbols, as a bpermute address; a lane that
  // stops the path points to itself
  uint32_t J[LEVELS];
  J[0] = lane_in(ok_m) ? (end < 63u ? end : 63u) * 4u : lane4;
#pragma unroll
  for (int k = 1; k < LEVELS; k++) J[k] = lane_value(J[k - 1], J[k - 1]);
  // am I on the path from offset 0?  largest path element <= lane, descending
  uint32_t v = 0;
#pragma unroll
  for (int k = LEVELS - 1; k >= 0; k--) {
    const uint32_t y = lane_value(v, J[k]);
    if (y <= lane4) v = y;
  }
  const mask_t visited_m = wave_mask(v == lane4);
  const mask_t commit0_m = visited_m & ok_m;
  const mask_t match0_m = commit0_m & match_m;
  // output offsets: exclusive scan of the produced bytes over the path
  const uint32_t mine = lane_in(commit0_m) ? outlen : 0u;
  const uint32_t incl = wave_scan_incl(mine);
  const uint32_t outoff = incl - mine;
  const uint32_t mrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(match0_m >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)match0_m, 0u));
  const uint32_t INF = 0xFFFFFFFFu;
  // (scalar work is the scarce resource of this kernel: one scalar instruction
  // issues per clock and CU, against four vector ones -- so what follows prefers
  // vector arithmetic and selects to mask algebra and branches)
  // the turn's first match (lane 63 when there is none: the sink commits nothing)
  const uint32_t fm_lane = (uint32_t)__builtin_ctzll(match0_m | (1ull << 63));
  const uint32_t first_match_dst = d.out_pos + (uint32_t)__builtin_amdgcn_readlane((int)outoff, (int)fm_lane);
  // symbols that would overflow, reach before the start, overfill the queue or
  // read an unfilled hole end the turn in front of them; the holes are the
  // queued copies (from hole_min on) and, behind the turn's first match, that
  // match.  over(a, b) > 0 exactly when a > b; the conditions are OR-ed as numbers
  // and tested once.
  const uint32_t room = d.cap_min - d.out_pos;
  const uint32_t qfree = (uint32_t)QUEUE_ENTRIES - d.q_count;
  const uint32_t dstp = d.out_pos + outoff;
  const uint32_t src_end = dstp - sp.dist + sp.length;
  const uint32_t h0 = d.hole_min;
  const uint32_t h1 = h0 < first_match_dst ? h0 : first_match_dst;  // also what queued destinations are relative to
  const uint32_t hole = mrank == 0 ? h0 : h1;  // the turn's first match only sees the queued copies
  // (IM_DRY / IM_TOKEN copy nothing: no queue, no holes)
  const uint32_t bad_match = MODE != IM_REAL ? over(sp.dist, dstp)
                                             : over(sp.dist, dstp) | over(mrank + 1u, qfree) | over(dstp - h1, QUEUE_REL_MAX) |
                                                   over(src_end, hole);
  const uint32_t bad = over(outoff + outlen, room) | (lane_in(match0_m) ? bad_match : 0u);
  const mask_t late_m = commit0_m & wave_mask(bad != 0u);
  // the path ends in a stop, or it runs into the sink (bit 63: the sink's own hop
  // need not be covered by LEVELS)
  const mask_t cut_m = (visited_m & ~ok_m) | late_m | (1ull << 63);
  const uint32_t c = (uint32_t)__builtin_ctzll(cut_m);  // a lane on the path, or the sink
  const mask_t commit_m = commit0_m & ((1ull << c) - 1ull);
  const mask_t commit_match_m = commit_m & match_m;
  if (MODE != IM_DRY && lane_in(commit_m & lit_m)) dst[dstp] = (uint8_t)sp.lit;
  // every lane writes a queue word: the ones without a committed match into the spare slot
  if (MODE == IM_REAL)
    L.queue(lane_in(commit_match_m) ? (int)(d.q_count + mrank) : QUEUE_ENTRIES) = queue_pack(dstp - h1, sp.dist, sp.length);
  if (MODE == IM_TOKEN && lane_in(commit_match_m)) {  // (length <= 16 <= distance)
    const uint32_t from = dstp - sp.dist;
    uint32_t v[DEFER_MAX_LEN];
#pragma unroll
    for (uint32_t i = 0; i < DEFER_MAX_LEN; i++) v[i] = wv::load_coherent(tok + (from + (i < sp.length ? i : 0u)));
#pragma unroll
    for (uint32_t i = 0; i < DEFER_MAX_LEN; i++)
      if (i < sp.length) tok[dstp + i] = v[i];
  }
  // bytes produced: what precedes the cut lane.  Bits used: up to the cut lane, or
  // past it when the last symbol runs over the sink.
  d.out_pos += (uint32_t)__builtin_amdgcn_readlane((int)outoff, (int)c);
  const uint32_t end_c = lane_in(commit_m) ? end : 0u;
  const uint32_t end_last = (uint32_t)__builtin_amdgcn_readlane((int)end_c, (63 - __clzll((long long)commit_m)) & 63);
  const uint32_t consumed = c > end_last ? c : end_last;
  if (MODE == IM_REAL) {
    const uint32_t fm_kept = fm_lane < c ? first_match_dst : INF;  // the first match, if it was committed
    d.hole_min = d.hole_min < fm_kept ? d.hole_min : fm_kept;
    d.q_count += (uint32_t)__popcll(commit_match_m);
  }
  d.advance(consumed);
  return c;
}

// Wide turns until one stops, the round's turns are used up or the staged input
// runs low.  Only the position and the output/queue counters change in here; the
// rest of the state stays put in its scalar registers.
template <int LEVELS, int MODE>
__device__ __forceinline__ bool wide_turns(InflateLane &d, const LaneLds &L, uint8_t *__restrict__ dst, uint32_t *__restrict__ tok, int lane,
                                           int &turn) {
  // turns with input to spare (PLENTY) while the staged words reach, then the careful form
  const uint32_t total = d.total_words();
  const uint32_t plenty_below = total > 8u ? total - 8u : 0u;  // 8 words > 31 + 63 + 48 bits + the peek
  const uint32_t staged_below = d.ring_wr >= (uint32_t)TURN_WORDS ? d.ring_wr - (uint32_t)TURN_WORDS + 1u : 0u;
  const uint32_t fast_below = plenty_below < staged_below ? plenty_below : staged_below;
  for (;;) {
    if (turn >= ROUND_TURNS) return false;
    if (d.in_word >= fast_below) break;
    turn++;
    if (wide_turn<LEVELS, true, MODE>(d, L, dst, tok, lane) < 63u) return true;
  }
  for (;;) {
    if (turn >= ROUND_TURNS) return false;
    if (!d.input_ready(TURN_WORDS)) return false;
    turn++;
    if (wide_turn<LEVELS, false, MODE>(d, L, dst, tok, lane) < 63u) return true;
  }
}

// A turn for the stretches a wide turn is worst at: literals that all have ONE code length n (base64: 64 letters of 6 bits --
// a wide turn's 64 bit offsets hold ten of them).  Lane k looks at the bits n k behind the position; while every lane before
// it found a literal of n bits, what it finds is the k-th symbol from here.  The literals in front of the first lane that
// found something else are stored side by side and the position moves behind them; that lane's symbol is the next turn's.
// (n <= LIT_TBITS: the table's entry is the whole code.  ready: words of input staged from the position's word on.)
#ifndef ZD_STRIDE_MIN
#define ZD_STRIDE_MIN 4
#endif
#ifndef ZD_STRIDE_WAIT
#define ZD_STRIDE_WAIT 8
#endif
template <int MODE>
__device__ __forceinline__ uint32_t strided_turn(InflateLane &d, const LaneLds &L, uint8_t *__restrict__ dst, int lane, uint32_t n, uint32_t ready) {
  const uint32_t p = d.boff + n * (uint32_t)lane;
  const int s = (int)((d.in_word + (p >> 5)) & (uint32_t)(RING_WORDS - 1));
  const uint32_t w0 = L.slot(s), w1 = L.slot(s + 1);
  const uint32_t e = L.wide_lit((int)(funnel32(w1, w0, p) & ((1u << LIT_TBITS) - 1)));
  const bool ok = (p >> 5) + 2u <= ready && n * ((uint32_t)lane + 1u) <= d.bits_left() && (int32_t)e < 0 && ((e >> 17) & 63u) == n;
  const unsigned long long m = wave_mask(ok);
  uint32_t c = ~m == 0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
  const uint32_t room = d.cap_min - d.out_pos;
  c = c < room ? c : room;
  if (MODE != IM_DRY && (uint32_t)lane < c) dst[d.out_pos + (uint32_t)lane] = (uint8_t)(e >> 8);
  d.out_pos += c;
  d.advance(c * n);
  return c;
}

// Are the n bits from bit a on the n bits from bit b on (b < a, all inside the input)?  32 bits a lane.
__device__ __forceinline__ bool same_bits(const uint8_t *__restrict__ src, uint32_t src_len, uint64_t a, uint64_t b, uint32_t n, int lane) {
  bool differ = false;
  for (uint32_t o = (uint32_t)lane * 32u; o < n; o += 2048u) {
    const uint64_t pa = a + o, pb = b + o;
    if ((pa >> 5) * 4u + 8u > src_len) { differ = true; break; }  // (two whole words are read: not at the input's very end)
    const uint32_t wa = funnel32(load_u32_le(src + (pa >> 5) * 4u + 4u), load_u32_le(src + (pa >> 5) * 4u), (uint32_t)pa & 31u);
    const uint32_t wb = funnel32(load_u32_le(src + (pb >> 5) * 4u + 4u), load_u32_le(src + (pb >> 5) * 4u), (uint32_t)pb & 31u);
    const uint32_t left = n - o;
    differ |= ((wa ^ wb) & (left >= 32u ? 0xFFFFFFFFu : (1u << left) - 1u)) != 0u;
  }
  return __builtin_amdgcn_ballot_w64(differ) == 0ull;
}

// A match handed to the wave, by MODE: its bytes (wave_copy_match), nothing, or where its bytes come from
template <int MODE>
__device__ __forceinline__ void wave_match(uint8_t *dst, uint32_t *__restrict__ tok, uint32_t pos, uint32_t dist, uint32_t len, int lane,
                                           uint8_t *pattern) {
  if (MODE == IM_REAL) wave_copy_match(dst, pos, dist, len, lane, pattern);
  else if (MODE == IM_TOKEN) {  // (what the source bytes are copies of, where that is known already: inflate_span.h)
    if (dist < len && 64u % dist == 0u) {  // (a run whose period divides the wave: a lane's bytes all copy one source)
      const uint32_t v = wv::load_coherent(tok + (pos - dist + (uint32_t)lane % dist));
      for (uint32_t i = (uint32_t)lane; i < len; i += 64u) tok[pos + i] = v;
    } else {
      for (uint32_t i = (uint32_t)lane; i < len; i += 64u)
        tok[pos + i] = wv::load_coherent(tok + (pos - dist + (dist < len ? i % dist : i)));
    }
  }
}

// A block of ONE stream decoded by a wave of its own (IM_DRY, IM_TOKEN): where it starts in the stream's input and
// in its output.  IM_DRY knows neither the output position nor what lies before it: it counts from BLOCK_DRY_BASE,
// where every distance is allowed, and reports the block's size and end; IM_TOKEN is given the real position and so
// makes the reference's check of a distance against it (zd.ml:614).
constexpr uint32_t BLOCK_DRY_BASE = 32768;

// The stream's wave: MODE IM_REAL is the whole stream (inflate_batch_kernel); the other two stop at the end of the
// block they were started on.
// (returns, in every lane, what the block modes found of their block)
// MULTI (with IM_DRY): an explorer (inflate_explore_kernel) -- the wave goes on from block to block and lists every
// block it walked from header to end in X.recs, until one of them started at or behind X.stop_bit; with
// X.inside_fixed it starts on what is taken for a symbol of a FIXED block (whose header lies somewhere before).
struct Explore {
  BlockRec *recs;
  uint32_t *n_recs;
  uint32_t cap, max_recs, inside_fixed;
  uint32_t *fate;  // (LDS, 3 words, or null) what became of an explorer: blocks listed, and the bit of the header it stood behind at the end
  uint64_t stop_bit;
  // every IM_DRY wave: where the checkpoints of the block being walked are gathered (2 * CK_MAX words of LDS) and,
  // for an explorer, where the listed blocks' go
  uint32_t *ck_lds;
  BlockCk *cks;
  // IM_TOKEN: the wave takes the block over at a checkpoint (resume: the block's tables are built from its header as
  // ever, then the position jumps) and/or leaves it at the next (until_bit: it stops at the first convenient
  // point at or behind that bit -- what it decodes beyond it, the next wave decodes too, to the same values)
  uint32_t resume, resume_out;
  uint64_t resume_bit, until_bit;
  // what the block's symbols are likely to take in bits (0: no idea): sizes the span decoder's regions like the
  // block before does for a stream's wave (prev_block_bits)
  uint32_t est_bits;
};
static_assert(CK_MAX == BLOCK_CK_MAX, "kernels.h");
constexpr uint64_t NO_BIT = ~0ull;
__device__ __forceinline__ Explore no_explore() {
  Explore X;
  X.recs = nullptr; X.n_recs = nullptr; X.cap = 0; X.max_recs = 0; X.inside_fixed = 0; X.stop_bit = 0; X.fate = nullptr;
  X.ck_lds = nullptr; X.cks = nullptr; X.resume = 0; X.resume_out = 0; X.resume_bit = 0; X.until_bit = NO_BIT;
  X.est_bits = 0;
  return X;
}
template <int MODE, bool MULTI = false, bool REUSE = false>
__device__ __forceinline__ BlockEnd inflate_wave(uint8_t *lds_raw, const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                                 const StreamDesc &sd, const BlockStart at, StreamResult *__restrict__ result,
                                                 uint16_t *__restrict__ span_idx, uint32_t *__restrict__ tok, int crc_op,
                                                 const Explore X = no_explore(), uint16_t *srcpos = nullptr) {
  const int lane = threadIdx.x;
  const bool crc_adler = MODE == IM_REAL && (crc_op == CRC_ADLER32 || crc_op == CRC_ADLER32_RFC);
  const bool adler_rfc = crc_op == CRC_ADLER32_RFC;
  const bool writer = lane == 0 && MODE != IM_DRY;  // (the lane that stores a literal decoded alone)

  LaneLds L;
  L.at(lds_raw);  // (the input ring first: its reads encode their offsets)

  Arenas A;
  A.src = src_arena;
  A.dst = dst_arena;
  InflateLane d;
  lane_init(d, sd);
  if (MODE != IM_REAL && d.status == ST_OK) {
    d.in_word = (uint32_t)(at.bit >> 5);
    d.boff = (uint32_t)at.bit & 31u;
    d.ring_wr = d.in_word;
    d.out_pos = MODE == IM_DRY ? BLOCK_DRY_BASE : at.out_pos;
    d.blk_out_start = d.out_pos;
    if (MODE == IM_DRY) d.hard_cap = d.limit = d.cap_min = (uint32_t)MAX_STREAM_LEN;
    d.prev_block_bits = X.est_bits;
    if (MULTI && X.inside_fixed) {  // as behind a fixed block's header (lane_block_header), tables to be built
      d.hdr_fixed = 1;
      d.phase = PH_TABLES;
    }
  }
  // the explorer's current block: its header's bit (NO_BIT: it was not seen), blocks listed so far
  uint64_t blk_hdr_bit = MULTI && X.inside_fixed ? NO_BIT : at.bit;
  uint32_t n_listed = 0;
  SpanCk ck;
  ck.lds = X.ck_lds; ck.n = 0; ck.last = 0; ck.hdr_bit = at.bit; ck.out_base = BLOCK_DRY_BASE;
  bool resume_pending = MODE == IM_TOKEN && X.resume != 0u;
  uint32_t lit_stride = 0;  // the one code length of most of the block's literals, or 0 (strided_turn)
  bool left_early = false;  // IM_TOKEN: the wave stopped at until_bit
  bool fixed_tables = false;  // MULTI: the tables in LDS are the fixed codes'
  // REUSE (inflate_batch_few_kernel: calls of a few streams): the last dynamic header read -- where it stood and how
  // long it was (0: none whose tables still stand).  A block behind a short block with the very same header bits (a
  // run: 1 MiB of zeros is 16 such blocks of 254 matches, their 15 us of code lengths and tables each half the
  // stream's time) goes on with the tables there are.  Not in the kernel of the large batches: three words of state
  // more in its main loop cost it 1.5 %.
  uint64_t hdr_at = 0, prev_hdr_at = 0;
  uint32_t prev_hdr_bits = 0;
  // a block has ended (the phase says what the stream's wave would do next): the block modes stop here, or go on
  auto block_done = [&]() {
    if (MODE == IM_REAL || d.status != ST_OK) return;
    if (!MULTI) { d.phase = PH_DONE; return; }
    const uint64_t end_bit = (uint64_t)d.in_word * 32u + d.boff;
    if (blk_hdr_bit != NO_BIT) {
      if (lane == 0) {
        const uint32_t k = atomicAdd(X.n_recs, 1u);
        if (k < X.cap) {
          BlockRec r;
          r.bit = blk_hdr_bit;
          r.e.status = ST_OK; r.e.final_block = (uint32_t)d.final_block; r.e.end_bit = end_bit;
          r.e.out_len = d.out_pos - BLOCK_DRY_BASE; r.e.pad = ck.n;
          X.recs[k] = r;
          if (X.cks) {
            X.cks[k].n = ck.n;
            for (uint32_t i = 0; i < 2u * ck.n; i++) X.cks[k].e[i] = ck.lds[i];
          }
        }
      }
      n_listed++;
    }
    ck.n = 0; ck.last = 0; ck.hdr_bit = end_bit;
    if (d.final_block || (blk_hdr_bit != NO_BIT && blk_hdr_bit >= X.stop_bit) || n_listed >= X.max_recs) { d.phase = PH_DONE; return; }
    blk_hdr_bit = end_bit;
    d.out_pos = BLOCK_DRY_BASE;
    d.blk_out_start = BLOCK_DRY_BASE;
  };
  uint8_t *dst = dst_arena + d.dst_off;
  const uint8_t *src = src_arena + d.src_off;

#ifdef ZD_INFLATE_PHASES  // timing-only build: the results carry cycle counts (tools/exp_inflate_phases.py)
  uint64_t ph_hdr = 0, ph_wide = 0, ph_plain = 0, ph_t;
  uint64_t span_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t ph_begin = __builtin_readcyclecounter();
#define ZD_PH(acc) do { const uint64_t now_ = __builtin_readcyclecounter(); acc += now_ - ph_t; ph_t = now_; } while (0)
#define ZD_PH_START() ph_t = __builtin_readcyclecounter()
#else
#define ZD_PH(acc) do {} while (0)
#define ZD_PH_START() do {} while (0)
#endif
  for (;;) {
    // ---- refill the input ring: lane t stages word ring_wr + t (zero past the end)
    if (d.phase != PH_DONE) {
      const uint32_t lim = d.in_word + (uint32_t)RING_WORDS;
      const uint32_t my = d.ring_wr + (uint32_t)lane;
      if (my < lim) {
        const uint64_t at = (uint64_t)my * 4u;
        const uint8_t *p = src + at;
        uint32_t wv = 0;
        if (at + 4u <= d.src_len) wv = load_u32_le(p);
        else
          for (uint32_t b = 0; at + b < d.src_len; b++) wv |= (uint32_t)p[b] << (8 * b);
        L.ring_put(my, wv);
      }
      uint32_t nw = d.ring_wr + 64u;
      if (nw > lim) nw = lim;
      d.ring_wr = nw;
    }
    uniformize(d);

    // ---- decode
    int plain_run = 0, stride_wait = 0;
    for (int turn = 0; turn < ROUND_TURNS;) {
      if (d.phase == PH_HDR_LENGTHS) {
        turn++;
        ZD_PH_START();
        const int r = wave_dynamic_lengths(d, L, lane);  // (wave-uniform: every lane keeps the same state)
        if (r == 0) break;  // waits for input
        if (r < 0) d.fail(ST_CORRUPTED);
        else { d.hdr_fixed = 0; d.phase = PH_TABLES; }
        ZD_PH(ph_hdr);
      } else if (d.phase == PH_HEADER) {
        turn++;
        ZD_PH_START();
        if (REUSE) {
          hdr_at = (uint64_t)d.in_word * 32u + d.boff;
          if (prev_hdr_bits != 0u && d.prev_block_bits < 16384u && prev_hdr_bits <= d.bits_left() &&
              same_bits(src, d.src_len, hdr_at, prev_hdr_at, prev_hdr_bits, lane)) {  // (the first bit, "final", is one of them)
            d.advance(prev_hdr_bits);
            d.blk_out_start = d.out_pos;
            lane_begin_symbols(d);  // (final_block 0, the codes, their tables, levels: the block before's)
            prev_hdr_at = hdr_at;
            ZD_PH(ph_hdr);
            // The header's bits -- up to 8191 -- were compared in memory, not read through the ring: the position may
            // now stand behind what is staged (input_ready's unsigned difference would wrap to "plenty" and the next
            // turn decode stale slots).  The ring starts over at the position, as behind a checkpoint.
            if (d.in_word + (uint32_t)TURN_WORDS > d.ring_wr) { d.ring_wr = d.in_word; break; }
            continue;
          }
          prev_hdr_bits = 0;
        }
        lit_stride = 0;  // (a block whose tables are not built -- a fixed block's table-free start -- has no stride of the block before's)
        bool ok = true;
        if (lane == 0) ok = lane_header_step(d, L, src_arena);
        uniformize(d);  // lane 0 is the first active lane: everybody takes its state
        if (!uni((uint32_t)ok)) break;  // waits for input
        ZD_PH(ph_hdr);
      } else if (d.phase == PH_TABLES || d.phase == PH_HDR_CODELEN) {  // (PH_TABLES: a dynamic header's lengths are read, or a fixed block went on beyond its table-free symbols)
        ZD_PH_START();
        if (MULTI) fixed_tables = d.phase == PH_TABLES && d.hdr_fixed != 0;  // (a dynamic header's codes go where the fixed ones stood)
        wave_tables(d, L, lane);
#ifdef ZD_HDR_SPLIT  // (experiment: the tables booked as "wide turns", the wide tables as "services+rest")
        ZD_PH(ph_wide);
#endif
        if (d.phase == PH_SYMBOLS) {
          const uint32_t shortest = build_wide_tables(d, L, lane);
          d.levels = levels_for(wave_min(shortest));
          {  // the length that 32 and more of the block's literal / length codes have (strided_turn), 0: none
            const uint32_t cnt = lane < 16 ? (uint32_t)L.u16(LDS_LIT_COUNTS, lane) : 0u;
            uint32_t bl = 0, bn = 31;
            for (int l = 1; l <= LIT_TBITS; l++) {
              const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l);
              if (nl > bn) { bn = nl; bl = (uint32_t)l; }
            }
            lit_stride = bl;
          }
          if (REUSE && !d.hdr_fixed && !d.final_block) {  // (a dynamic header's tables stand: its bits are hdr_at .. here)
            prev_hdr_at = hdr_at;
            const uint64_t len = (uint64_t)d.in_word * 32u + d.boff - hdr_at;
            prev_hdr_bits = len < 8192u ? (uint32_t)len : 0u;
          }
          if (resume_pending) {  // the tables stand: on from the checkpoint
            resume_pending = false;
            d.in_word = (uint32_t)(X.resume_bit >> 5);
            d.boff = (uint32_t)X.resume_bit & 31u;
            d.ring_wr = d.in_word;
            d.out_pos = X.resume_out;
            break;  // (the input ring starts over)
          }
        }
#ifndef ZD_HDR_SPLIT
        ZD_PH(ph_hdr);
#endif
      } else if (MULTI && d.phase == PH_SYMBOLS && d.fixed_lazy && fixed_tables) {
        // (an explorer walks fixed block after fixed block: the tables of the one before are this one's -- nothing
        // but a dynamic header's wave_tables writes where they stand -- instead of 48 symbols decoded one by one and
        // 15 us of building them again)
        d.fixed_lazy = 0;
      } else if (d.phase == PH_SYMBOLS && d.fixed_lazy && resume_pending) {  // (a fixed block taken over at a checkpoint: its tables now)
        d.fixed_lazy = 0;
        d.phase = PH_TABLES;
      } else if (MODE == IM_TOKEN && d.phase == PH_SYMBOLS && X.until_bit != NO_BIT && d.q_count == 0 &&
                 (uint64_t)d.in_word * 32u + d.boff >= X.until_bit) {  // the next wave's part
        d.phase = PH_DONE;
        left_early = true;
      } else if (d.phase == PH_SYMBOLS && d.fixed_lazy) {  // a fixed block's first symbols, straight from the code
        turn++;
        ZD_PH_START();
        const int r = lane_one_symbol_fixed(d, L, A, writer, MODE == IM_REAL);
        uniformize(d);
        const int ru = uni(r);
        ZD_PH(ph_plain);
        if (ru == SYM_EOB) { d.fixed_lazy = 0; lane_end_of_block(d, crc_adler); block_done(); }
        else if (ru == SYM_STOP) {
          if (d.phase == PH_REQ_MATCH && d.q_count == 0) {
            wave_match<MODE>(dst, tok, d.out_pos, d.req_dist, d.req_len, lane, L.x + LDS_SPAN_TILE_BYTE);
            lane_after_match(d);
          } else break;
        }
        if (d.phase == PH_SYMBOLS && d.fixed_lazy && --d.fixed_lazy == 0) d.phase = PH_TABLES;
      } else if (d.phase == PH_SYMBOLS) {
        if (!d.span_off && d.in_word >= d.span_retry_word) {  // the block's symbols by regions, all lanes busy (inflate_span.h)
          if (d.q_count) break;  // queued copies first: the span reads its match sources from memory
          const uint32_t out_before = d.out_pos;
          SpanCk *ckp = MODE == IM_DRY && ck.lds != nullptr && blk_hdr_bit != NO_BIT ? &ck : nullptr;
          uint32_t bits_cap = 0xFFFFFFFFu;
          if (MODE == IM_TOKEN && X.until_bit != NO_BIT) {
            const uint64_t pos = (uint64_t)d.in_word * 32u + d.boff;
            bits_cap = pos >= X.until_bit ? 0u : (uint32_t)(X.until_bit - pos) + 4096u;
          }
          ZD_PH_START();
#ifdef ZD_INFLATE_PHASES
          const int sr = span_decode<MODE>(d, L, src, dst, span_idx, tok, srcpos, bits_cap, ckp, lane, span_ph);
#else
          const int sr = span_decode<MODE>(d, L, src, dst, span_idx, tok, srcpos, bits_cap, ckp, lane);
#endif
          if (sr != SPAN_NONE) {
            uniformize(d);
            span_after(d, sr, d.out_pos != out_before);  // (inflate_span.h: when the next span is tried, and how large)
            ZD_PH(ph_plain);  // (timing build: the span's clocks are booked as "plain")
            break;  // the input ring starts over at the new position
          }
          d.span_off = 1;  // too little input left for a span: the wide turns take the rest
        }
        if (!d.input_ready(TURN_WORDS)) break;
        ZD_PH_START();
        bool stopped = true;
        // (where the wide turns commit nothing -- runs of long matches: zeros, periods -- they are
        // skipped for a few symbols after one that stopped at its very first symbol)
        const bool try_stride = plain_run == 0 && lit_stride != 0u && stride_wait == 0 && d.ring_wr - d.in_word >= 8u;
        const uint32_t stride_got = try_stride ? strided_turn<MODE>(d, L, dst, lane, lit_stride, d.ring_wr - d.in_word) : 0u;
        if (plain_run > 0) { plain_run--; turn++; }
        else if (stride_got >= 16u) {
          turn++;  // (a turn that went well: the next is one of the same)
          stopped = false;
        } else {
          // (a strided turn that got little: wide turns for a while -- the stretch is not of one length)
          if (try_stride) stride_wait = stride_got < (uint32_t)ZD_STRIDE_MIN ? ZD_STRIDE_WAIT : 0;  // (a strided turn costs a quarter of a wide one: four symbols pay for it)
          else if (stride_wait > 0) stride_wait--;
          const uint32_t before = d.out_pos;
          if (d.levels == 4) stopped = wide_turns<4, MODE>(d, L, dst, tok, lane, turn);
          else if (d.levels == 5) stopped = wide_turns<5, MODE>(d, L, dst, tok, lane, turn);
          else stopped = wide_turns<6, MODE>(d, L, dst, tok, lane, turn);
          if (stopped && d.out_pos == before) plain_run = 6;
        }
        ZD_PH(ph_wide);
        if (stopped) {
          const uint32_t pos_before = d.in_word * 32u + d.boff;
          const int r = lane_one_symbol(d, L, A, writer, MODE == IM_REAL);
          uniformize(d);
          const int ru = uni(r);
          ZD_PH(ph_plain);
          if (ru == SYM_EOB) { lane_end_of_block(d, crc_adler); block_done(); }
          else if (ru == SYM_STOP) {
            // a match that cannot be queued (long, or overlapping its own output) and nothing queued
            // before it: the wave copies it here and now instead of going round through the services
            if (d.phase == PH_REQ_MATCH && d.q_count == 0) {
              d.req_len = match_run(d, L, lane, d.in_word * 32u + d.boff - pos_before);
              wave_match<MODE>(dst, tok, d.out_pos, d.req_dist, d.req_len, lane, L.x + LDS_SPAN_TILE_BYTE);
              lane_after_match(d);
              continue;
            }
            break;
          }
        }
      } else {
        break;
      }
    }

    // ---- services
    if (d.q_count) {  // fill the queued copies: entry `lane`, all loads before the stores
      DeferredCopy c0;
      c0.len = 0;
      if ((uint32_t)lane < d.q_count) deferred_load(c0, dst, d.hole_min, L.queue(lane));
      if (c0.len) deferred_store(c0, dst);
      d.q_count = 0;
      d.hole_min = 0xFFFFFFFFu;
    }
    if (d.phase == PH_REQ_MATCH) {
      wave_match<MODE>(dst, tok, d.out_pos, d.req_dist, d.req_len, lane, L.x + LDS_SPAN_TILE_BYTE);
      lane_after_match(d);
    } else if (d.phase == PH_REQ_COPY) {
      if (MODE != IM_DRY) wave_copy(dst + d.out_pos, src + d.req_src, d.req_len, lane);
      lane_after_copy(d, crc_adler);
      block_done();
    }
    if (d.phase == PH_REQ_ADLER) {
      // the block's bytes were stored by other lanes of this wave: make them visible
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      d.adler = wave_adler_update(d.adler, dst + d.blk_out_start, d.out_pos - d.blk_out_start, lane, adler_rfc);
      lane_after_adler(d);
    }
    if (d.phase == PH_DONE && d.q_count == 0) break;
  }

  if (MULTI && X.fate != nullptr && lane == 0) {
    X.fate[0] = n_listed;
    X.fate[1] = (uint32_t)blk_hdr_bit;
    X.fate[2] = (uint32_t)(blk_hdr_bit >> 32);
  }
  BlockEnd e;
  e.status = d.status;
  e.final_block = (uint32_t)d.final_block;
  e.end_bit = (uint64_t)d.in_word * 32u + d.boff;
  e.out_len = d.out_pos - (MODE == IM_DRY ? BLOCK_DRY_BASE : at.out_pos);
  e.pad = MODE == IM_DRY ? ck.n : left_early ? 1u : 0u;
  if (MODE != IM_REAL) return e;
  if (writer) {
    StreamResult r;
    r.status = d.status;
    r.out_len = d.status == ST_OK ? d.out_pos : 0;
    // CRC-32 is filled in by the checksum pass over the produced bytes
    // (chaining per block is exact for CRC-32); Adler-32 is final here.
    r.checksum = (crc_adler && d.status == ST_OK) ? d.adler : 0u;
#ifdef ZD_INFLATE_PHASES
    r.status = (uint32_t)((__builtin_readcyclecounter() - ph_begin) >> 6);
    r.checksum = (uint32_t)(ph_hdr >> 6);
    r.out_len = (ph_wide >> 6) | ((ph_plain >> 6) << 32);
    for (int i = 0; i < 8; i++) ((uint64_t *)dst)[i] = span_ph[i];  // over the stream's first output bytes (dst slots are 256-byte aligned)
#endif
    *result = r;
  }
  return e;
}

// A descriptor's flags before its wave starts.  STREAM_DONE is the library's own mark, in its own copy of a call's
// descriptors (api.hip: the stream went by blocks, its result stands) -- such launches say so in crc_op (CRC_OP_MARKED).
// In a caller's descriptors that bit, like every bit but STREAM_HAS_LIMIT, is an invalid argument and is reported as one
// (it used to skip the stream silently and leave whatever its result slot held).
__device__ __forceinline__ bool inflate_skips_stream(uint32_t flags, int &crc_op, StreamResult *result) {
  const bool marked = (crc_op & CRC_OP_MARKED) != 0;
  crc_op &= ~CRC_OP_MARKED;
  const uint32_t other = flags & ~STREAM_HAS_LIMIT;
  if (other == 0) return false;
  if (other == STREAM_DONE && marked) return true;
  if ((threadIdx.x & 63u) == 0) {
    StreamResult r;
    r.status = ST_INVALID_ARG; r.checksum = 0; r.out_len = 0;
    *result = r;
  }
  return true;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void inflate_batch_kernel(const uint8_t *__restrict__ src_arena,
                                                           uint8_t *__restrict__ dst_arena,
                                                           const StreamDesc *__restrict__ descs,
                                                           StreamResult *__restrict__ results,
                                                           uint32_t n_streams, uint16_t *__restrict__ span_scratch, int crc_op) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_BYTES_PER_LANE];
  const uint32_t stream = blockIdx.x;
  if (stream >= n_streams) return;
  if (inflate_skips_stream(descs[stream].flags, crc_op, results + stream)) return;
  BlockStart at;
  at.bit = 0; at.out_pos = 0; at.chunk0 = 0;
  inflate_wave<IM_REAL>(lds_raw, src_arena, dst_arena, descs[stream], at, results + stream,
                        span_scratch + (size_t)stream * SPAN_IDX_ENTRIES, nullptr, crc_op);
}
// the same for calls of a few streams (api.hip: up to 256), where a wave has more to itself than its share of a full
// GPU and a stream of runs -- blocks of a few hundred matches with one and the same header -- is not lost in a batch:
// REUSE (inflate_wave)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void inflate_batch_few_kernel(const uint8_t *__restrict__ src_arena,
                                                           uint8_t *__restrict__ dst_arena,
                                                           const StreamDesc *__restrict__ descs,
                                                           StreamResult *__restrict__ results,
                                                           uint32_t n_streams, uint16_t *__restrict__ span_scratch, int crc_op) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_BYTES_PER_LANE];
  const uint32_t stream = blockIdx.x;
  if (stream >= n_streams) return;
  if (inflate_skips_stream(descs[stream].flags, crc_op, results + stream)) return;
  BlockStart at;
  at.bit = 0; at.out_pos = 0; at.chunk0 = 0;
  inflate_wave<IM_REAL, false, true>(lds_raw, src_arena, dst_arena, descs[stream], at, results + stream,
                                     span_scratch + (size_t)stream * SPAN_IDX_ENTRIES, nullptr, crc_op);
}


// ---------------------------------------------------------------------------------
// ONE stream by a wave per BLOCK -- the shape of the reference's own functions (inflate zd.ml:711-724 takes one
// stream), where the kernel above has one wave to give.  A block's symbols can be walked without anything that lies
// before it, and where its bytes come from can be written down without knowing them:
//
//   find    every bit offset of the input is tried as the header of a dynamic block (zd.ml:638-669: the block type,
//           the three counts in range, a code-length code that is complete, then the code lengths read with it: a
//           literal/length code that is complete and has an end-of-block symbol, a distance code that is complete,
//           single or empty).  A block's real header passes; of random bits about one offset in 2^30 does.
//           (inflate_find_headers_kernel: a thread per input byte, 8 offsets; inflate_find_lengths_kernel: a thread
//           per offset that got as far as the code-length code.)  Bit 0 is a block's start whatever its type.
//   dry     a wave per candidate runs the stream's wave (IM_DRY) over that one block: every check but the distance
//           against the output position, nothing stored: the block's end bit, size, status.
//   chain   from bit 0 on, a block must end where a candidate starts, until a final block: the blocks' output
//           positions.  Anything else -- a fixed or stored block behind the first, an error, output beyond the
//           limit -- and the stream is left to inflate_batch_kernel, which owns every message of the reference.
//   token   a wave per block of the chain runs the stream's wave again (IM_TOKEN) at the block's real output
//           position: literals (and stored bytes) are stored, every byte of a match gets, in tok[], the position
//           it is a copy of (a position that holds a literal is a copy of itself).
//   resolve tok[i] <- tok[tok[i]] until every byte points at a literal (pointer jumping: rounds ~ log of the longest
//           chain of copies, whatever blocks it crosses), then out[i] = out[tok[i]].

// (a thread per 4 bytes = 32 offsets; what passes is gathered per workgroup -- 1 KiB of input, about 8 offsets -- so the
// list's counter sees one atomic per workgroup, not one per offset: 110 000 on one address took a millisecond)
constexpr uint32_t FIND_WG_LIST = 512;
__global__ __launch_bounds__(256) void inflate_find_headers_kernel(const uint8_t *__restrict__ src_arena,
                                                                  const StreamDesc *__restrict__ descs,
                                                                  const BlocksJob *__restrict__ jobs) {
  __shared__ uint32_t l_n, l_base, l_list[FIND_WG_LIST];
  const BlocksJob J = jobs[blockIdx.y];  // (a call's streams side by side: the grid's second dimension)
  const StreamDesc sd = descs[J.stream];
  uint32_t *first = J.first;
  const uint32_t first_cap = J.first_cap;
  FindCounts *counts = J.counts;
  if ((uint64_t)blockIdx.x * 1024u >= sd.src_len) return;  // (the grid is the longest stream's)
  const uint8_t *s = src_arena + sd.src_off;
  const uint64_t t = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;  // first byte
  if (threadIdx.x == 0) l_n = 0;
  __syncthreads();
  if (t < sd.src_len) {
    const uint64_t total_bits = sd.src_len * 8u;
    const uint64_t a = find_bits(s, sd.src_len, t * 8u), b = find_bits(s, sd.src_len, t * 8u + 64u);
#pragma unroll 1
    for (uint32_t m = find_header_mask32(a); m != 0u; m &= m - 1u) {
      const uint32_t off = (uint32_t)__builtin_ctz(m);
      const uint64_t bit = t * 8u + off;
      if (bit == 0 || bit >= total_bits) continue;  // (bit 0 is a candidate anyway)
      if (!find_header_test(off ? (a >> off) | (b << (64u - off)) : a, b >> off, total_bits - bit)) continue;
      const uint32_t k = atomicAdd(&l_n, 1u);
      if (k < FIND_WG_LIST) l_list[k] = (uint32_t)bit;
    }
  }
  __syncthreads();
  const uint32_t n = l_n < FIND_WG_LIST ? l_n : FIND_WG_LIST;
  if (n == 0) return;
  if (threadIdx.x == 0) l_base = atomicAdd(&counts->n_first, n);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += 256u)
    if (l_base + k < first_cap) first[l_base + k] = l_list[k];
}

// the code lengths behind a header that passed: a thread per offset
__global__ __launch_bounds__(64) void inflate_find_lengths_kernel(const uint8_t *__restrict__ src_arena,
                                                                 const StreamDesc *__restrict__ descs,
                                                                 const BlocksJob *__restrict__ jobs) {
  __shared__ uint8_t find_tbl[128 * 64];  // (a column per thread: entry e of thread t at e * 64 + t)
  const BlocksJob J = jobs[blockIdx.y];
  const StreamDesc sd = descs[J.stream];
  const uint32_t *first = J.first;
  const uint32_t first_cap = J.first_cap, cand_cap = J.cand_cap;
  uint32_t *cand = J.cand;
  FindCounts *counts = J.counts;
  const uint8_t *s = src_arena + sd.src_off;
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  uint32_t n_first = counts->n_first;
  if (n_first > first_cap) n_first = first_cap;
  if (i == 0) {  // bit 0
    const uint32_t at = atomicAdd(&counts->n_cand, 1u);
    if (at < cand_cap) cand[at] = 0;
  }
  if (i >= n_first) return;
  const uint32_t start = first[i];
  if (!find_lengths_test(s, sd.src_len, start, find_tbl + threadIdx.x, 64u)) return;
  const uint32_t at = atomicAdd(&counts->n_cand, 1u);
  if (at < cand_cap) cand[at] = start;
}

// A dry run per candidate: recs[b] = the candidate's bit and what became of its block
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void inflate_blocks_dry_kernel(
    const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
    const BlocksJob *__restrict__ jobs) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_BYTES_PER_LANE];
  __shared__ uint32_t ck_lds[2 * CK_MAX];
  const BlocksJob J = jobs[blockIdx.y];
  const uint32_t *cand = J.cand;
  BlockRec *recs = J.recs;
  BlockCk *cks = J.cks;
  const uint32_t cand_cap = J.cand_cap;
  uint16_t *span_scratch = J.span;
  FindCounts *counts = J.counts;
  const uint32_t b = blockIdx.x;
  const uint32_t n = counts->n_cand <= cand_cap ? counts->n_cand : 0u;  // (more candidates than the list holds: the call gives up)
  if (b >= n) return;
  BlockStart at;
  at.bit = cand[b]; at.out_pos = 0; at.chunk0 = 0;
  Explore X = no_explore();
  X.ck_lds = ck_lds;
  if (n <= 8192u) {  // the block probably ends where the next candidate starts (the candidates are few: a pass over them)
    uint32_t next = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < n; i += 64u) {
      const uint32_t c = cand[i];
      if (c > (uint32_t)at.bit && c < next) next = c;
    }
    next = wave_min(next);
    if (next != 0xFFFFFFFFu && next - (uint32_t)at.bit < (1u << 22)) X.est_bits = next - (uint32_t)at.bit;
  }
  const BlockEnd e = inflate_wave<IM_DRY>(lds_raw, src_arena, dst_arena, descs[J.stream], at, nullptr,
                                          span_scratch + (size_t)b * SPAN_IDX_ENTRIES, nullptr, CRC_NOP, X);
  if (threadIdx.x == 0) {
    BlockRec r;
    r.bit = at.bit; r.e = e;
    recs[b] = r;
    cks[b].n = e.pad;
    for (uint32_t i = 0; i < 2u * e.pad; i++) cks[b].e[i] = ck_lds[i];
    if (b == 0) counts->n_recs = n;
  }
}

// Blocks whose headers cannot be looked for: FIXED ones (the reference's encoder, whose estimate of a dynamic
// block's size grows from block to block, ends up coding nearly every block of a long stream that way)
// and stored ones.  From the first bit the chain could not go on from, an explorer every `stride` bits: the first
// starts on that block's header; the others take their bit for the start of a symbol of a fixed block -- they all
// have the one code -- and as two walks of one Huffman stream fall into step within a few symbols, what such a wave
// walks is soon the real sequence: the end of its block, the next header, and block after block from there, each
// listed with its bit, end and size until one started behind the next explorer's first bit.  What an explorer lists
// before its walk was real are blocks that no chain leads to.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void inflate_explore_kernel(
    const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
    const BlocksJob *__restrict__ jobs, uint32_t stride_bits) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_BYTES_PER_LANE];
  __shared__ uint32_t ck_lds[2 * CK_MAX];
  __shared__ uint32_t fate[3];
  const BlocksJob J = jobs[blockIdx.y];
  FindCounts *counts = J.counts;
  const FindCounts *from = counts;
  BlockRec *recs = J.recs;
  BlockCk *cks = J.cks;
  const uint32_t rec_cap = J.rec_cap, n = J.n, n_explorers = J.n_blocks;
  uint16_t *span_scratch = J.span;
  const uint32_t b = blockIdx.x;
  if (b >= n) return;
  BlockStart at;
  at.out_pos = 0; at.chunk0 = 0;
  Explore X = no_explore();
  X.recs = recs; X.n_recs = &counts->n_recs; X.cap = rec_cap; X.max_recs = 16;
  if (b < n_explorers) {
    at.bit = from->miss_bit + (uint64_t)b * stride_bits;
    X.inside_fixed = b != 0;
    X.stop_bit = at.bit + stride_bits;
  } else {
    // a FOLLOWER: the block behind a listed (dynamic) block that lies behind the chain's stop.  An explorer that
    // starts inside a dynamic block finds nothing, so the fixed block behind one was listed by nobody and the chain
    // walked it itself, one after the other (the reference's encoder mixes the two kinds: 5 or 6 such walks of
    // 0.35 ms in 64 MiB).  Where that block starts is known exactly -- the listed block's end -- and one wave each
    // lists it.
    const BlockRec r = recs[b - n_explorers];  // (listed before this launch: the explorers append behind them)
    if (r.e.status != ST_OK || r.e.final_block || r.e.end_bit <= from->miss_bit) return;
    at.bit = r.e.end_bit;
    X.inside_fixed = 0;
    X.stop_bit = at.bit;  // (the first block it lists starts there: one block)
  }
  X.ck_lds = ck_lds; X.cks = cks;
  X.fate = fate;
  // An explorer that meets what reads as an end-of-block code before its walk has fallen into step (seven zero bits:
  // one symbol in 128, and falling into step takes a few dozen) goes on to read a header that is none, and three of
  // four such headers end it there.  Of the three or four explorers that start inside a block of 30 KiB, all ended that
  // way for one block in 130, which the chain's one wave then walked itself (0.35 ms each, one after the other).
  // Such an explorer starts again behind the false end, on what is once more taken for a symbol: a few times.
  for (int attempt = 0; attempt < 4; attempt++) {
    if (at.bit + 64u > descs[J.stream].src_len * 8u) return;
    const BlockEnd e = inflate_wave<IM_DRY, true>(lds_raw, src_arena, dst_arena, descs[J.stream], at, nullptr,
                                                  span_scratch + (size_t)b * SPAN_IDX_ENTRIES, nullptr, CRC_NOP, X);
    __syncthreads();
    const uint32_t listed = fate[0];
    uint64_t stood = (uint64_t)fate[1] | ((uint64_t)fate[2] << 32);
    __syncthreads();
    if (stood == NO_BIT) stood = e.end_bit + 1u;  // (it met a code that is none before any end of block: on, a bit further)
    if (e.status == ST_OK || listed != 0u || !X.inside_fixed || stood == NO_BIT || stood <= at.bit || stood >= X.stop_bit) return;
    at.bit = stood;
  }
}

// the listed blocks in stream order: each finds its rank (they are few; the same block may be listed more than once)
__global__ __launch_bounds__(256) void inflate_sort_blocks_kernel(const BlocksJob *__restrict__ jobs) {
  const BlocksJob J = jobs[blockIdx.y];
  const BlockRec *recs = J.recs;
  const FindCounts *counts = J.counts;
  const uint32_t rec_cap = J.rec_cap;
  BlockRec *sorted = J.sorted;
  uint32_t *sorted_src = J.sorted_src;
  const uint32_t n = counts->n_recs < rec_cap ? counts->n_recs : rec_cap;
  // (the others' bits 256 at a time through LDS: a load per comparison from memory was a round trip each, 0.5-0.8 ms
  // of a stream whose explorers listed a few thousand blocks)
  __shared__ uint64_t others[256];
  if (blockIdx.x * 256u >= n) return;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint64_t mine = i < n ? recs[i].bit : 0ull;
  uint32_t r = 0;
  for (uint32_t base = 0; base < n; base += 256u) {
    const uint32_t j = base + threadIdx.x;
    others[threadIdx.x] = j < n ? recs[j].bit : ~0ull;
    __syncthreads();
    const uint32_t m = n - base < 256u ? n - base : 256u;
    for (uint32_t t = 0; t < m; t++) {
      const uint64_t b = others[t];
      r += b < mine || (b == mine && base + t < i) ? 1u : 0u;
    }
    __syncthreads();
  }
  if (i >= n) return;
  sorted[r] = recs[i];
  sorted_src[r] = i;
}

// The chain of blocks from bit 0, by one wave.  chain[k]: block k's header bit and output position; chain_end[k]:
// what the dry run said of it (the token run must agree).  A block that ends where no listed block starts:
// walk == 0 -- the chain stops and says where (miss_bit: the explorers start there); walk != 0 -- this wave makes
// that block's dry run here and now, and goes on.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void inflate_chain_kernel(
    const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
    const BlocksJob *__restrict__ jobs, int walk) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_BYTES_PER_LANE];
  __shared__ uint32_t ck_lds[2 * CK_MAX];
  const BlocksJob J = jobs[blockIdx.y];
  const BlockRec *sorted = J.sorted;
  const uint32_t *sorted_src = J.sorted_src;
  const uint32_t rec_cap = J.rec_cap, chain_cap = J.chain_cap;
  BlockStart *chain = J.chain;
  BlockEnd *chain_end = J.chain_end;
  ChainIv *chain_iv = J.chain_iv;
  BlockCk *cks = J.cks;
  uint16_t *span_scratch = J.span;
  FindCounts *counts = J.counts;
  const StreamDesc sd = descs[J.stream];
  const uint32_t n = counts->n_recs < rec_cap ? counts->n_recs : rec_cap;
  const uint64_t room = (sd.flags & STREAM_HAS_LIMIT) && sd.limit < sd.dst_cap ? sd.limit : sd.dst_cap;
  uint64_t out = 0, bit = 0, miss = NO_BIT;
  uint32_t k = 0, walked = 0, chunks = 0, intervals = 0;
  // the listed blocks, 64 at a time in the lanes' registers (a step of the chain is then a ballot and a few
  // readlanes, not three loads one after the other): lane l holds block wj + l
  uint32_t wj = 0;
  uint64_t wbit = NO_BIT;
  BlockEnd we;
  uint32_t wsrc = 0;
  auto window = [&](uint32_t base) {
    wj = base;
    const uint32_t i = base + threadIdx.x;
    wbit = NO_BIT;
    we.status = 0; we.final_block = 0; we.end_bit = 0; we.out_len = 0; we.pad = 0;
    wsrc = 0;
    if (i < n) { wbit = sorted[i].bit; we = sorted[i].e; wsrc = sorted_src[i]; }
  };
  window(0);
  bool ok = n != 0 && sorted[0].bit == 0;
  while (ok) {
    unsigned long long hit;
    for (;;) {  // the window that can hold `bit`
      hit = __builtin_amdgcn_ballot_w64(wbit == bit);
      if (hit != 0ull || wj + 64u >= n || __builtin_amdgcn_ballot_w64(wbit != NO_BIT && wbit > bit) != 0ull) break;
      window(wj + 64u);
    }
    if (hit != 0ull) {
      // A RUN of the window's blocks at once: behind the block that starts at `bit`, every lane whose block starts
      // where the lane before it ends (a stream's listed blocks are mostly its real ones, in order).  Their output
      // positions, intervals and Adler chunks are scans over the lanes, their records one store per lane -- a step
      // per block by lane 0 was 0.43 us: 0.39 ms of a 64 MiB stream's 5.2.  Whatever a run stops in front of (a
      // block that does not follow, an error, the limit, the chain's capacity) the one-block step below decides.
      const uint32_t l0 = (uint32_t)__builtin_ctzll(hit), lane = threadIdx.x;
      // (a block listed more than once -- by a candidate's dry run, an explorer or two, a follower -- stands at
      // neighbouring lanes: the first listing counts, as in the one-block step, and the others are passed over)
      const uint64_t bit_before = ((uint64_t)(uint32_t)__shfl_up((int)(wbit >> 32), 1) << 32) | (uint32_t)__shfl_up((int)(uint32_t)wbit, 1);
      const bool again = lane > l0 && wbit == bit_before;
      int first_of = again ? -1 : (int)lane;  // the lane of the nearest listing at or before mine that counts
      uint32_t counted = lane >= l0 && !again ? 1u : 0u;  // ... and how many count from l0 up to mine
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int fu = __shfl_up(first_of, d);
        const uint32_t cu = (uint32_t)__shfl_up((int)counted, d);
        if ((int)lane >= d) { first_of = first_of > fu ? first_of : fu; counted += cu; }
      }
      const int before = __shfl_up(first_of, 1);  // the listing that counts before mine
      const uint64_t prev_end = ((uint64_t)(uint32_t)__shfl((int)(we.end_bit >> 32), before < 0 ? 0 : before) << 32) |
                                (uint32_t)__shfl((int)(uint32_t)we.end_bit, before < 0 ? 0 : before);
      const bool follows = lane == l0 || wbit == prev_end;
      const bool sound = wbit != NO_BIT && we.status == ST_OK && we.end_bit > wbit;
      const bool counts_here = lane >= l0 && !again;
      const uint64_t len64 = counts_here && sound ? (uint64_t)we.out_len : 0ull;
      uint64_t incl = len64;
      uint32_t iv_incl = counts_here ? 1u + we.pad : 0u, ch_incl = counts_here && we.out_len ? 1u + we.out_len / ADLER_CHUNK : 0u;
      const uint32_t iv_own = iv_incl, ch_own = ch_incl;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = ((uint64_t)(uint32_t)__shfl_up((int)(incl >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)incl, d);
        const uint32_t iu = (uint32_t)__shfl_up((int)iv_incl, d), cu = (uint32_t)__shfl_up((int)ch_incl, d);
        if ((int)lane >= d) { incl += up; iv_incl += iu; ch_incl += cu; }
      }
      const bool fits = out + incl <= room && out + incl <= MAX_STREAM_LEN && (uint64_t)k + (counted - 1u) < chain_cap;
      const unsigned long long from_l0 = ~0ull << l0;
      const unsigned long long bad = __builtin_amdgcn_ballot_w64(counts_here && !(follows && sound && fits)) & from_l0;
      const unsigned long long fin = __builtin_amdgcn_ballot_w64(counts_here && we.final_block != 0) & from_l0;
      uint32_t last = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;  // the run: lanes [l0, last)
      if (fin && (uint32_t)__builtin_ctzll(fin) < last) last = (uint32_t)__builtin_ctzll(fin) + 1u;  // (a final block ends it, and the chain)
      const uint32_t run_blocks = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(counts_here && lane < last));
      if (run_blocks >= 2u) {
        if (counts_here && lane < last) {
          const uint32_t r = k + (counted - 1u);
          BlockStart b;
          b.bit = wbit; b.out_pos = (uint32_t)(out + incl - len64); b.chunk0 = chunks + ch_incl - ch_own;
          chain[r] = b;
          chain_end[r] = we;
          ChainIv iv;
          iv.first = intervals + iv_incl - iv_own; iv.ck = wsrc;
          chain_iv[r] = iv;
        }
        int ll = (int)last - 1;  // the run's last listing that counts
        ll = __builtin_amdgcn_readlane(first_of, ll);
        const uint64_t tot = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(incl >> 32), ll) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)incl, ll);
        out += tot;
        intervals += (uint32_t)__builtin_amdgcn_readlane((int)iv_incl, ll);
        chunks += (uint32_t)__builtin_amdgcn_readlane((int)ch_incl, ll);
        k += run_blocks;
        if (__builtin_amdgcn_readlane((int)we.final_block, ll) != 0) break;
        bit = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(we.end_bit >> 32), ll) << 32) |
              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)we.end_bit, ll);
        continue;
      }
    }
    BlockEnd e;
    uint32_t ck_at;  // the block's checkpoints: the listed block's, or (a block walked here) slot rec_cap + k
    if (hit != 0ull) {
      const int l = __builtin_ctzll(hit);
      e.status = (uint32_t)__builtin_amdgcn_readlane((int)we.status, l);
      e.final_block = (uint32_t)__builtin_amdgcn_readlane((int)we.final_block, l);
      e.end_bit = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(we.end_bit >> 32), l) << 32) |
                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)we.end_bit, l);
      e.out_len = (uint32_t)__builtin_amdgcn_readlane((int)we.out_len, l);
      e.pad = (uint32_t)__builtin_amdgcn_readlane((int)we.pad, l);
      ck_at = (uint32_t)__builtin_amdgcn_readlane((int)wsrc, l);
    }
    else if (!walk) { ok = false; miss = bit; break; }
    else {
      if (k >= chain_cap) { ok = false; break; }
      BlockStart at;
      at.bit = bit; at.out_pos = 0; at.chunk0 = 0;
      walked++;
      Explore X = no_explore();
      X.ck_lds = ck_lds;
      e = inflate_wave<IM_DRY>(lds_raw, src_arena, dst_arena, sd, at, nullptr, span_scratch, nullptr, CRC_NOP, X);
      ck_at = rec_cap + k;
      if (threadIdx.x == 0) {
        cks[ck_at].n = e.pad;
        for (uint32_t i = 0; i < 2u * e.pad; i++) cks[ck_at].e[i] = ck_lds[i];
      }
    }
    if (e.status != ST_OK || out + e.out_len > room || out + e.out_len > MAX_STREAM_LEN || k >= chain_cap) { ok = false; break; }
    if (threadIdx.x == 0) {
      BlockStart b;
      b.bit = bit; b.out_pos = (uint32_t)out; b.chunk0 = chunks;
      chain[k] = b;
      chain_end[k] = e;
      ChainIv iv;
      iv.first = intervals; iv.ck = ck_at;
      chain_iv[k] = iv;
    }
    intervals += 1u + e.pad;
    k++;
    out += e.out_len;
    if (e.out_len) chunks += 1u + e.out_len / ADLER_CHUNK;  // (Adler_32.string_update zd.ml:175-198: a first chunk of len mod 5552, maybe empty)
    if (e.final_block) break;
    if (e.end_bit <= bit) { ok = false; break; }  // (cannot be: a block has a header)
    bit = e.end_bit;
  }
  if (threadIdx.x == 0) {
    counts->chain_ok = ok ? 1u : 0u;
    counts->n_blocks = k;
    counts->out_len = out;
    counts->n_walked = walked;
    counts->miss_bit = miss;
    counts->n_chunks = chunks;
    counts->n_intervals = intervals;
  }
}

__global__ __launch_bounds__(256) void inflate_tok_init_kernel(const BlocksJob *__restrict__ jobs) {
  const BlocksJob J = jobs[blockIdx.y];
  uint32_t *tok = J.tok;
  const uint32_t n = J.out_len;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) tok[i] = i;
}

// The token run: a wave per INTERVAL of the chain's blocks -- a block from its header to its first checkpoint, from
// checkpoint to checkpoint, from the last one to the block's end (the dry run took one every 6 KiB of input or so, at
// starts of the span decoder's tiles: bit and output position on the real sequence).  A wave that starts at a
// checkpoint builds the block's tables from the header first.  The last wave of a block must end where and with as
// many bytes as the dry run did; the others leave at or behind the next checkpoint (what they decode behind it the
// next wave decodes too: the same literals, and for a match byte a source on the same chain of copies).
// (two waves per SIMD: its LDS -- the stream's 10 KiB and 8 KiB of source positions -- lets a CU hold eight workgroups)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void inflate_blocks_token_kernel(
    const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
    const BlocksJob *__restrict__ jobs) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[LDS_BYTES_PER_LANE];
  __shared__ uint16_t srcpos[SPAN_TILE];
  const BlocksJob J = jobs[blockIdx.y];
  const BlockStart *chain = J.chain;
  const BlockEnd *chain_end = J.chain_end;
  const ChainIv *chain_iv = J.chain_iv;
  const BlockCk *cks = J.cks;
  const uint32_t n_blocks = J.n_blocks, n = J.n;
  uint16_t *span_scratch = J.span;
  uint32_t *tok = J.tok;
  FindCounts *counts = J.counts;
  const int follow = J.follow;
  const uint32_t w = blockIdx.x;
  if (w >= n) return;
  // (follow: a long stream, where a wave writes down what its sources are copies of -- inflate_span.h -- and that
  // works best when ONE wave takes a block from start to end: a wave per block, n = n_blocks)
  uint32_t lo = follow ? w : 0u, hi = follow ? w : n_blocks - 1u;  // the last block whose first interval is at or before w
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (chain_iv[mid].first <= w) lo = mid;
    else hi = mid - 1u;
  }
  const ChainIv iv = chain_iv[lo];
  const BlockStart blk = chain[lo];
  const BlockCk *ck = cks + iv.ck;
  const uint32_t j = follow ? 0u : w - iv.first, n_ck = follow ? 0u : ck->n;
  Explore X = no_explore();
  if (j > 0u) {
    X.resume = 1;
    X.resume_bit = blk.bit + ck->e[2u * (j - 1u)];
    X.resume_out = blk.out_pos + ck->e[2u * (j - 1u) + 1u];
  }
  if (j < n_ck) X.until_bit = blk.bit + ck->e[2u * j];
  const BlockEnd got = inflate_wave<IM_TOKEN>(lds_raw, src_arena, dst_arena, descs[J.stream], blk, nullptr,
                                              span_scratch + (size_t)w * SPAN_IDX_ENTRIES, tok, CRC_NOP, X, follow ? srcpos : nullptr);
  if (threadIdx.x == 0) {
    const BlockEnd want = chain_end[lo];
    // (a wave that was to leave at a checkpoint close to the block's end may have reached that end instead)
    const bool at_end = got.pad == 0u && got.end_bit == want.end_bit && got.out_len == want.out_len;
    const bool good = got.status == ST_OK && (at_end || (j < n_ck && got.pad == 1u && got.end_bit >= X.until_bit));
    if (!good) atomicAdd(&counts->token_bad, 1u);
  }
}

// Pointer jumping, up to `hops` hops a thread and round (api.hip: 256): a pointer only ever moves to an earlier byte of the
// same chain of copies, so reading one that another thread has already moved is as good.  Round 0 looks at every
// byte; a thread that did not arrive at a literal lists its byte (more[round] counts them), and the rounds behind
// look at the listed bytes only (all rounds are launched; one whose list is empty returns at once).

__device__ __forceinline__ void resolve_one(uint32_t *__restrict__ tok, uint32_t i, bool have, uint32_t *__restrict__ list_out,
                                            uint32_t *__restrict__ count_out, int hops) {
  bool open = false;
  if (have) {
    uint32_t j = tok[i];
    if (j != i) {
      uint32_t j2 = tok[j];
      for (int h = 1; h < hops && j2 != j; h++) { j = j2; j2 = tok[j]; }
      open = j2 != j;
      tok[i] = j2;
    }
  }
  const unsigned long long m = __builtin_amdgcn_ballot_w64(open);
  if (m != 0ull) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(count_out, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    if (open) list_out[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
  }
}
__global__ __launch_bounds__(256) void inflate_resolve_kernel(const BlocksJob *__restrict__ jobs, int round, int hops) {
  const BlocksJob J = jobs[blockIdx.y];
  uint32_t *tok = J.tok;
  const uint32_t n = J.out_len;
  FindCounts *counts = J.counts;
  // the two lists of bytes still open, behind tok[]: a round reads the one the round before wrote
  const uint32_t *list_in = tok + (size_t)n * (1u + (uint32_t)((round + 1) & 1));
  uint32_t *list_out = tok + (size_t)n * (1u + (uint32_t)(round & 1));
  if (round == 0) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x * 256u >= n) return;  // (the grid is the longest stream's)
    resolve_one(tok, i, i < n, list_out, &counts->more[0], hops);
    return;
  }
  const uint32_t n_in = counts->more[round - 1];
  for (uint32_t t0 = blockIdx.x * 256u; t0 < n_in; t0 += gridDim.x * 256u) {  // (wave-uniform trip count)
    const uint32_t t = t0 + threadIdx.x;
    resolve_one(tok, t < n_in ? list_in[t] : 0u, t < n_in, list_out, &counts->more[round], hops);
  }
}
// Adler-32 of a stream that went by blocks: the reference updates it block by block (inflated_block_crc zd.ml:682-690),
// every block's bytes in chunks of their own grid (first len mod 5552, then 5552 each), and its signed remainder makes
// the value depend on that grid.  A wave per chunk: its two sums (sums[3 c .. 3 c + 2] = S1, S2, length); then one
// wave folds the chunks in stream order, 64 loaded at a time, with the reference's step.
__global__ __launch_bounds__(64) void inflate_adler_chunks_kernel(const uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
                                                                 const BlocksJob *__restrict__ jobs) {
  const BlocksJob J = jobs[blockIdx.y];
  const BlockStart *chain = J.chain;
  const BlockEnd *chain_end = J.chain_end;
  const uint32_t n_blocks = J.n_blocks, n_chunks = J.counts->n_chunks;
  uint32_t *sums = J.sums;
  const uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  uint32_t lo = 0, hi = n_blocks - 1u;  // the last block whose chunks start at or before c (and that has chunks)
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (chain[mid].chunk0 <= c) lo = mid;
    else hi = mid - 1u;
  }
  // (blocks without output have no chunks and share their chunk0 with the block behind: step back to the one that has)
  while (lo > 0u && chain_end[lo].out_len == 0u) lo--;
  const BlockStart b = chain[lo];
  const uint32_t n = chain_end[lo].out_len, j = c - b.chunk0, first = n % ADLER_CHUNK;
  const uint32_t start = j == 0u ? 0u : first + (j - 1u) * ADLER_CHUNK, len = j == 0u ? first : ADLER_CHUNK;
  uint32_t S1, S2;
  wave_adler_chunk_sums(dst_arena + descs[J.stream].dst_off + b.out_pos + start, len, (int)threadIdx.x, S1, S2);
  // (bit 31 of the length: the block's last chunk -- the reference packs the value into one word between two blocks
  // and unpacks it again, zd.ml:178,198, which is not the identity once a signed remainder has gone negative)
  if (threadIdx.x == 0) { sums[3u * c] = S1; sums[3u * c + 1u] = S2; sums[3u * c + 2u] = len | (j == n / ADLER_CHUNK ? 0x80000000u : 0u); }
}
__global__ __launch_bounds__(64) void inflate_adler_fold_kernel(const BlocksJob *__restrict__ jobs, int rfc,
                                                               StreamResult *__restrict__ results) {
  const BlocksJob J = jobs[blockIdx.y];
  const uint32_t *sums = J.sums;
  const uint32_t n_chunks = J.counts->n_chunks;
  StreamResult *result = results + J.stream;
  uint32_t s1, s2;
  adler_unpack(1u, s1, s2);  // Adler_32.init zd.ml:173
  for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64u) {
    const uint32_t c = c0 + threadIdx.x;
    const uint32_t S1 = c < n_chunks ? sums[3u * c] : 0u, S2 = c < n_chunks ? sums[3u * c + 1u] : 0u, len = c < n_chunks ? sums[3u * c + 2u] : 0u;
    const uint32_t m = n_chunks - c0 < 64u ? n_chunks - c0 : 64u;
    for (uint32_t l = 0; l < m; l++) {
      const uint32_t ln = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)l);
      adler_chunk_step(s1, s2, ln & 0x7FFFFFFFu, (uint32_t)__builtin_amdgcn_readlane((int)S1, (int)l),
                       (uint32_t)__builtin_amdgcn_readlane((int)S2, (int)l), rfc != 0);
      if (ln >> 31) adler_unpack(adler_pack(s1, s2), s1, s2);
    }
  }
  if (threadIdx.x == 0) result->checksum = adler_pack(s1, s2);
}
// the stream's result, written where the caller reads it (no copy from the host, no wait for one)
__global__ void inflate_blocks_result_kernel(const BlocksJob *__restrict__ jobs, StreamResult *__restrict__ results, uint32_t n_jobs) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  StreamResult r;
  r.status = ST_OK; r.checksum = 0; r.out_len = jobs[j].out_len;
  results[jobs[j].stream] = r;
}
// every byte of a match takes the literal its chain of sources ends in (tok[i] after the resolve rounds; a literal's is i).
// (Four bytes a thread -- one load of their four words, one word stored -- takes the same 0.21-0.23 ms per 64 MiB of text: the
// kernel is its scattered byte reads.)
__global__ __launch_bounds__(256) void inflate_gather_kernel(uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
                                                            const BlocksJob *__restrict__ jobs) {
  const BlocksJob J = jobs[blockIdx.y];
  const uint32_t *tok = J.tok;
  const uint32_t n = J.out_len;
  uint8_t *o = dst_arena + descs[J.stream].dst_off;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = tok[i];
  if (j != i) o[i] = o[j];
}


// ---------------------------------------------------------------------------------
// A stream longer than MAX_STREAM_LEN (positions are 32-bit in the kernel above) whose
// blocks are STORED ones of one length -- what the reference's encoder makes of incompressible
// data, BASELINE's "8 GiB pre-compressed high-entropy stream": 131 073 blocks of 65 534 bytes
// (read_uncompressed_block zd.ml:671-680, the loop zd.ml:692-709).  Walking 131 073 headers one
// after the other is a chain of dependent loads; instead every block's header is looked for
// where it has to be if all blocks before it have the first block's length (offset j * (5 + LEN))
// and all candidates are checked at once: the chain holds as far as every candidate is a stored
// header (BTYPE 00 on a byte boundary, LEN = ~NLEN) of that length.  The blocks of that prefix
// are copied with 64-bit offsets, one workgroup each; whatever follows -- a shorter last block,
// a block of another kind -- is an ordinary stream for the kernel above (api.hip).
__global__ void stored_chain_probe_kernel(const uint8_t *__restrict__ src_arena, const StreamDesc *__restrict__ descs,
                                          StoredChain *__restrict__ st) {
  const StreamDesc sd = descs[0];
  const uint8_t *s = src_arena + sd.src_off;
  StoredChain c;
  c.len0 = 0; c.first_bad = 0; c.final_at = 0xFFFFFFFFu; c.candidates = 0;
  if (sd.src_len >= 5 && (s[0] & 6u) == 0u) {
    const uint32_t len = s[1] | ((uint32_t)s[2] << 8), nlen = s[3] | ((uint32_t)s[4] << 8);
    if (len == ((~nlen) & 0xFFFFu) && len != 0u) {
      c.len0 = len;
      const uint64_t stride = 5ull + len;
      const uint64_t j = sd.src_len / stride;
      c.candidates = j > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)j;
      c.first_bad = c.candidates;
    }
  }
  *st = c;
}
__global__ __launch_bounds__(256) void stored_chain_scan_kernel(const uint8_t *__restrict__ src_arena,
                                                               const StreamDesc *__restrict__ descs,
                                                               StoredChain *__restrict__ st) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  const uint32_t len0 = st->len0;
  if (j >= st->candidates) return;
  const uint64_t stride = 5ull + len0;
  const uint8_t *h = src_arena + descs[0].src_off + (uint64_t)j * stride;
  const uint32_t len = h[1] | ((uint32_t)h[2] << 8), nlen = h[3] | ((uint32_t)h[4] << 8);
  const bool ok = (h[0] & 6u) == 0u && len == len0 && len == ((~nlen) & 0xFFFFu);
  if (!ok) atomicMin(&st->first_bad, j);
  else if (h[0] & 1u) atomicMin(&st->final_at, j);
}
// block j of the chain: LEN bytes from src + j * (5 + LEN) + 5 to dst + j * LEN
__global__ __launch_bounds__(256) void stored_chain_copy_kernel(const uint8_t *__restrict__ src_arena,
                                                               uint8_t *__restrict__ dst_arena,
                                                               const StreamDesc *__restrict__ descs, uint32_t len0) {
  const StreamDesc sd = descs[0];
  const uint8_t *s = src_arena + sd.src_off + (uint64_t)blockIdx.x * (5ull + len0) + 5u;
  uint8_t *o = dst_arena + sd.dst_off + (uint64_t)blockIdx.x * len0;
  const uint32_t body = len0 & ~15u;
  for (uint32_t i = threadIdx.x * 16u; i < body; i += 256u * 16u) store16_unaligned(o + i, load16_unaligned(s + i));
  if (threadIdx.x < (len0 & 15u)) o[body + threadIdx.x] = s[body + threadIdx.x];
}

// The same for stored blocks of any lengths (a stream some other encoder made, or the reference's own
// behind a run of shorter blocks).  A block's length says where the next header is, so the headers are a
// chain of dependent loads; but runs of equal blocks are the rule, so the wave guesses that the next 64
// blocks have the length of the last one seen and checks 64 headers per step: all that hold in a row are
// listed, the first that does not (another length: it becomes the next guess) ends the step.
__global__ __launch_bounds__(64) void stored_walk_kernel(const uint8_t *__restrict__ src_arena,
                                                         const StreamDesc *__restrict__ descs,
                                                         StoredWalk *__restrict__ walk, StoredBlock *__restrict__ list,
                                                         uint32_t list_cap) {
  const StreamDesc sd = descs[0];
  const uint8_t *s = src_arena + sd.src_off;
  const uint32_t lane = threadIdx.x;
  uint64_t p = walk->src_pos, o = walk->dst_pos, room = walk->room;  // wave-uniform
  uint32_t n = 0, stop = WALK_MORE;
  uint32_t guess = 0xFFFFFFFFu;  // no guess: only lane 0's header counts
  while (n + 64u <= list_cap) {
    // lane k: the header that starts k blocks of the guessed length further on
    const uint64_t q = p + (uint64_t)lane * (5ull + (guess == 0xFFFFFFFFu ? 0u : guess));
    const bool here = guess != 0xFFFFFFFFu || lane == 0;
    uint32_t len = 0, kind = 0;  // kind: 0 a stored header that holds, 1 another kind of block, 2 damaged / cut short
    bool final = false;
    if (here) {
      if (q + 1u > sd.src_len) kind = 2;  // (no byte left for the block's 3 header bits: the reference's read fails)
      else if ((s[q] & 6u) != 0u) kind = 1;
      else if (q + 5u > sd.src_len) kind = 2;
      else {
        len = s[q + 1] | ((uint32_t)s[q + 2] << 8);
        const uint32_t nlen = s[q + 3] | ((uint32_t)s[q + 4] << 8);
        if (len != ((~nlen) & 0xFFFFu) || q + 5u + len > sd.src_len) kind = 2;
        final = (s[q] & 1u) != 0u;
      }
    }
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
    const uint32_t kind0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)kind);
    if (kind0 != 0u) { stop = kind0 == 1u ? WALK_OTHER : WALK_CORRUPT; break; }
    if (guess != len0) {  // the guess was wrong for lane 0 already (or there was none): only lane 0 stands
      guess = len0;
      if (lane != 0) kind = 3;
    }
    // the lanes that hold in a row: good header, the guessed length, room for their bytes, no final block before them
    const bool good = here && kind == 0u && len == guess && (uint64_t)(lane + 1u) * guess <= room;
    const unsigned long long gm = __builtin_amdgcn_ballot_w64(good);
    const unsigned long long fm = __builtin_amdgcn_ballot_w64(good && final);
    uint32_t m = ~gm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~gm);
    if (fm) { const uint32_t f = (uint32_t)__builtin_ctzll(fm); if (f < m) { m = f + 1u; stop = WALK_FINAL; } }
    if (m == 0u) { stop = WALK_ROOM; break; }  // lane 0 holds but does not fit
    if (lane < m) {
      StoredBlock b;
      b.src = q + 5u; b.dst = o + (uint64_t)lane * guess; b.len = guess; b.pad = 0;
      list[n + lane] = b;
    }
    n += m;
    p += (uint64_t)m * (5ull + guess);
    o += (uint64_t)m * guess;
    room -= (uint64_t)m * guess;
    if (stop == WALK_FINAL) break;
    // (the guess stays: a step that listed one block because its guess was fresh or wrong has just LEARNED the length
    // the next 64 headers are tried with -- round 3 dropped it again here and walked one header per step -- and a guess
    // that no longer holds costs the one-block step in which lane 0 corrects it)
  }
  if (lane == 0) { walk->src_pos = p; walk->dst_pos = o; walk->room = room; walk->n_blocks = n; walk->stop = stop; }
}
__global__ __launch_bounds__(256) void stored_list_copy_kernel(const uint8_t *__restrict__ src_arena,
                                                              uint8_t *__restrict__ dst_arena,
                                                              const StreamDesc *__restrict__ descs,
                                                              const StoredBlock *__restrict__ list, uint32_t n_blocks) {
  const StreamDesc sd = descs[0];
  for (uint32_t j = blockIdx.x; j < n_blocks; j += gridDim.x) {
    const StoredBlock b = list[j];
    const uint8_t *s = src_arena + sd.src_off + b.src;
    uint8_t *o = dst_arena + sd.dst_off + b.dst;
    const uint32_t body = b.len & ~15u;
    for (uint32_t i = threadIdx.x * 16u; i < body; i += 256u * 16u) store16_unaligned(o + i, load16_unaligned(s + i));
    if (threadIdx.x < (b.len & 15u)) o[body + threadIdx.x] = s[body + threadIdx.x];
  }
}

}  // namespace zd
