// Host simulation of the lane-serial deflate pieces (zipc_amd/csrc/deflate_lane.h).
// TEST TOOLING ONLY: builds the hash-chain links serially (the GPU builds them
// with lz_chain_kernel), then runs the SAME match / parse / block-coder / bit
// item functions the kernels run, packing bits with a plain serial writer.  The
// result must equal the oracle's deflate byte for byte.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../zipc_amd/csrc/deflate_lane.h"
#include "serial_walk.h"

using namespace zd;

namespace {
struct BitWriter {
  std::vector<uint8_t> out;
  uint64_t acc = 0;
  int nbits = 0;
  void put(uint64_t v, int n) {
    acc |= v << nbits;
    nbits += n;
    while (nbits >= 8) { out.push_back((uint8_t)acc); acc >>= 8; nbits -= 8; }
  }
  void flush() { if (nbits > 0) { out.push_back((uint8_t)acc); acc = 0; nbits = 0; } }
};

uint32_t adler_update_serial(uint32_t a, const uint8_t *p, uint32_t n) {
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
}  // namespace


// Host model of lz_parse_kernel (zipc_amd/csrc/deflate.hip): one wave per stream,
// tiles of 64 positions (one per lane); inside a tile the visited positions are
// found by pointer doubling + top-down marking instead of a serial walk.
static uint32_t parse_tiles_model(const uint8_t *s, uint32_t len, const uint32_t *brefs, const uint32_t *steps,
                                  uint32_t *syms, BlockDesc *blocks) {
  const int T = 64, X = 576;
  uint32_t entry = 0, nsym = 0, blk_start = 0, blk_sym_start = 0, nblk = 0;
  uint32_t B = 0;
  while (B < len) {
    if (entry >= B + T) { B = entry & ~63u; continue; }
    uint32_t J[7][64], cnt[64], adv[64];  // 2^k hops, k = 0..6: the exit can be 64 hops away
    bool valid[64];
    for (int t = 0; t < T; t++) {
      uint32_t p = B + t;
      valid[t] = p < len;
      uint32_t br = valid[t] ? brefs[p] : 0, st = valid[t] ? steps[p] : 0;
      adv[t] = valid[t] ? (br ? macro_advance(st) : 1u) : 0u;
      cnt[t] = valid[t] ? (br ? macro_lits(st) + 1u : 1u) : 0u;
      J[0][t] = valid[t] ? t + adv[t] : t;
    }
    for (int k = 1; k < 7; k++)
      for (int t = 0; t < T; t++) { uint32_t x = J[k - 1][t]; J[k][t] = x < 64 ? J[k - 1][x] : x; }
    bool M[64];
    uint32_t vv[64];
    for (int t = 0; t < T; t++) {  // descending search, as the kernel's lanes do
      uint32_t v = entry - B;
      for (int k = 6; k >= 0; k--) { uint32_t y = J[k][v]; if (y <= (uint32_t)t) v = y; }
      vv[t] = v;
      M[t] = v == (uint32_t)t;
    }
    uint32_t exit_abs = B + J[0][vv[63]];
    if (exit_abs > len) exit_abs = len;
    // scan + emission
    uint32_t first[64], run = nsym;
    for (int t = 0; t < T; t++) { first[t] = run; if (valid[t] && M[t]) run += cnt[t]; }
    for (int t = 0; t < T; t++) {
      if (!(valid[t] && M[t])) continue;
      MacroStep m;
      m.bref = brefs[B + t];
      m.step = steps[B + t];
      lz_emit_position(s, B + t, m, syms, first[t]);
    }
    // block cut (write_block_symbol zd.ml:1118-1123)
    const uint64_t limit = (uint64_t)blk_start + MAX_BLOCK_SRC_LEN;
    for (int t = 0; t < T; t++) {
      if (!(valid[t] && M[t])) continue;
      const uint32_t p = B + t;
      if ((uint64_t)p + adv[t] <= limit) continue;
      uint32_t cutpos, symidx;
      const uint32_t br = brefs[p], lits = br ? macro_lits(steps[p]) : 0;
      if (br == 0) { cutpos = p; symidx = first[t]; }
      else if ((uint64_t)p + lits > limit) { uint32_t i = (uint32_t)(limit - p); cutpos = p + i; symidx = first[t] + i; }
      else { cutpos = p + lits; symidx = first[t] + lits; }
      BlockDesc b;
      b.src_start = blk_start; b.src_len = cutpos - blk_start;
      b.sym_start = blk_sym_start; b.n_syms = symidx - blk_sym_start;
      blocks[nblk++] = b;
      blk_start = cutpos;
      blk_sym_start = symidx;
      break;
    }
    nsym = run;
    entry = exit_abs;
    B += T;
  }
  BlockDesc b;
  b.src_start = blk_start; b.src_len = len - blk_start;
  b.sym_start = blk_sym_start; b.n_syms = nsym - blk_sym_start;
  blocks[nblk++] = b;
  return nblk;
}

// Host model of lz_parse_spec_kernel / lz_parse_stitch_kernel / lz_parse_gather_kernel (deflate.hip): every
// segment of SEG positions parsed from its first position, then segment by segment from the true entry until the
// path shares a position OF THE SAME TILE with the segment's own (what the kernel's tile test sees), runs of
// equal long steps taken 64 at a time, the block cut at the step that holds source byte blk_start + 65534.
static uint32_t parse_segments_model(const uint8_t *s, uint32_t len, const uint32_t *brefs, const uint32_t *steps,
                                     uint32_t SEG, uint32_t *syms, BlockDesc *blocks) {
  auto adv = [&](uint32_t p) { return brefs[p] ? macro_advance(steps[p]) : 1u; };
  auto step_of = [&](uint32_t p) { MacroStep m; m.bref = brefs[p]; m.step = steps[p]; return m; };
  const uint32_t nseg = (len + SEG - 1) / SEG, ntiles = len / 64 + 2;
  std::vector<uint64_t> vis(ntiles, 0), vis2(ntiles, 0xA5A5A5A5A5A5A5A5ull);  // (vis2: stale marks wherever nothing wrote)
  std::vector<uint32_t> tsym0(ntiles, 0), tsym2(ntiles, 0xDEADu), sexit(nseg + 1, 0), stotal(nseg + 1, 0);
  std::vector<std::vector<uint32_t>> spec(nseg + 1);
  for (uint32_t k = 0; k < nseg; k++) {
    const uint32_t lim = (k + 1) * SEG < len ? (k + 1) * SEG : len;
    uint32_t p = k * SEG, n = 0, cur = 0xFFFFFFFFu;
    spec[k].assign(SEG + 600, 0);
    while (p < lim) {
      if ((p >> 6) != cur) { cur = p >> 6; tsym0[cur] = n; }
      vis[p >> 6] |= 1ull << (p & 63);
      lz_emit_position(s, p, step_of(p), spec[k].data(), n);
      n += macro_sym_count(step_of(p));
      p += adv(p);
    }
    sexit[k] = p < len ? p : len;
    stotal[k] = n;
  }
  uint32_t off = 0, exit_prev = 0, blk_start = 0, blk_sym_start = 0, nblk = 0;
  for (uint32_t k = 0; k < nseg; k++) {
    const uint32_t seg_start = k * SEG;
    uint32_t seg_end = (k + 1) * SEG < len ? (k + 1) * SEG : len;
    const uint32_t e = exit_prev;
    uint32_t f = 0, from = 0, exit_k = sexit[k], again_end = seg_start;
    if (k != 0 && e >= seg_end) { from = stotal[k]; exit_k = e; again_end = seg_end; }
    else if (k != 0) {
      uint32_t B = e & ~63u, entry = e, rest = 0;
      // a path that leaves the segment without having met goes on into the next one (parse_again's `leave`); true: the end
      auto leave = [&](uint32_t Bx) -> bool {
        while (Bx >= seg_end) {
          if (k + 1 >= nseg || seg_end >= len) { again_end = Bx; from = stotal[k]; exit_k = entry; return true; }
          k++;
          seg_end = (k + 1) * SEG < len ? (k + 1) * SEG : len;
        }
        return false;
      };
      const uint32_t ext_end = len;
      for (;;) {
        const uint64_t own = vis[B >> 6];
        uint64_t vm = 0;
        const uint32_t f0 = f;
        uint32_t p = entry;
        while (p < B + 64 && p < len) {
          vm |= 1ull << (p & 63);
          lz_emit_position(s, p, step_of(p), syms + off, f);
          f += macro_sym_count(step_of(p));
          p += adv(p);
        }
        const uint32_t node_pos = B + (uint32_t)__builtin_ctzll(vm | (1ull << 63));
        vis2[B >> 6] = vm; tsym2[B >> 6] = f0;
        entry = p < len ? p : len;
        const uint32_t Be = entry & ~63u, Bn = Be > B + 64 ? Be : B + 64;
        const bool met = (vm & own) != 0;
        for (uint32_t tz = B + 64; tz < Bn && tz < ext_end; tz += 64) vis2[tz >> 6] = 0;
        if (met) {
          again_end = Bn;
          from = Bn < seg_end ? tsym0[Bn >> 6] : stotal[k];
          exit_k = sexit[k];
          break;
        }
        if (leave(Bn)) break;
        const uint32_t stride = entry - node_pos;
        if (rest != 0) rest--;
        else if (__builtin_popcountll(vm) == 1 && stride >= 64) {
          uint32_t lead = 0;
          while (lead < 64 && (uint64_t)lead * stride < (uint64_t)(ext_end - entry) && adv(entry + lead * stride) == stride) lead++;
          if (lead == 0) rest = 8;
          else {
            for (uint32_t i = 0; i < lead; i++) {
              const uint32_t q = entry + i * stride;
              lz_emit_position(s, q, step_of(q), syms + off, f);
              vis2[q >> 6] = 1ull << (q & 63);
              tsym2[q >> 6] = f;
              for (uint32_t tz = (q & ~63u) + 64; tz < ((q + stride) & ~63u) && tz < ext_end; tz += 64) vis2[tz >> 6] = 0;
              f += macro_sym_count(step_of(q));
            }
            entry += lead * stride;
            if (entry > len) entry = len;
            const uint32_t Bs = entry & ~63u;
            if (leave(Bs)) break;
            B = Bs;
            continue;
          }
        }
        B = Bn;
      }
    }
    const uint32_t n_own = stotal[k] - from;
    for (uint32_t i = 0; i < n_own; i++) syms[off + f + i] = spec[k][from + i];  // the gather
    while (exit_k - blk_start > (uint32_t)MAX_BLOCK_SRC_LEN) {
      const uint32_t T = blk_start + MAX_BLOCK_SRC_LEN;
      const uint32_t Ts = T < seg_end ? T : seg_end - 1;
      uint32_t Bt = Ts & ~63u;
      auto again_tile = [&](uint32_t b) { return k != 0 && b >= (e & ~63u) && b < again_end; };
      auto marks = [&](uint32_t b) { return again_tile(b) ? vis2[b >> 6] : vis[b >> 6]; };
      uint64_t w = marks(Bt) & (~0ull >> (63 - (Ts & 63)));
      while (w == 0) { Bt -= 64; w = marks(Bt); }
      const uint32_t c = 63 - (uint32_t)__builtin_clzll(w);
      uint32_t first = again_tile(Bt) ? off + tsym2[Bt >> 6] : off + f - from + tsym0[Bt >> 6];
      for (uint32_t t = 0; t < c; t++)
        if ((marks(Bt) >> t) & 1) first += macro_sym_count(step_of(Bt + t));
      const uint32_t p = Bt + c, rel = p - blk_start;
      const uint32_t br = brefs[p], lits = br ? macro_lits(steps[p]) : 0;
      uint32_t cutpos, symidx;
      if (br == 0) { cutpos = p; symidx = first; }
      else if (rel + lits > (uint32_t)MAX_BLOCK_SRC_LEN) { const uint32_t i = MAX_BLOCK_SRC_LEN - rel; cutpos = p + i; symidx = first + i; }
      else { cutpos = p + lits; symidx = first + lits; }
      BlockDesc b;
      b.src_start = blk_start; b.src_len = cutpos - blk_start;
      b.sym_start = blk_sym_start; b.n_syms = symidx - blk_sym_start;
      blocks[nblk++] = b;
      blk_start = cutpos;
      blk_sym_start = symidx;
    }
    off += f + n_own;
    exit_prev = exit_k;
  }
  BlockDesc b;
  b.src_start = blk_start; b.src_len = len - blk_start;
  b.sym_start = blk_sym_start; b.n_syms = off - blk_sym_start;
  blocks[nblk++] = b;
  return nblk;
}

// kinds[] receives the block kinds (0 stored, 1 fixed, 2 dynamic), up to max_kinds
extern "C" int sim_deflate(const uint8_t *src, uint32_t len, int level, uint8_t *dst, uint64_t dst_cap,
                           uint64_t *out_len, uint32_t *adler_out, int *kinds, int max_kinds,
                           int *n_kinds) {
  BitWriter w;
  uint32_t adler = 1;
  int nk = 0;
  if (level == LEVEL_NONE) {  // write_all_non_compressed zd.ml:1106-1116
    uint32_t start = 0;
    for (;;) {
      uint32_t n = len - start < (uint32_t)MAX_BLOCK_SRC_LEN ? len - start : (uint32_t)MAX_BLOCK_SRC_LEN;
      bool final = start + n == len;
      adler = adler_update_serial(adler, src + start, n);
      w.put(final ? 1 : 0, 3);
      w.flush();
      w.put(n & 0xFFFF, 16);
      w.put((~n) & 0xFFFF, 16);
      for (uint32_t i = 0; i < n; i++) w.put(src[start + i], 8);
      if (nk < max_kinds) kinds[nk] = 0;
      nk++;
      if (final) break;
      start += n;
    }
  } else {
    int good_match, K;
    level_params(level, good_match, K);
    std::vector<uint16_t> prev(len + 8, 0);
    std::vector<uint64_t> match(len + 8, 0);
    if (len >= 4) {
      std::vector<int64_t> head(32768, -1);
      for (uint32_t p = 0; p + 4 <= len; p++) {
        uint32_t h = hash4(load_u32_le(src + p));
        int64_t q = head[h];
        prev[p] = (q >= 0 && p - q <= 32768) ? (uint16_t)(p - q) : 0;
        head[h] = p;
      }
      // the kernel's 4-positions-per-lane form, positions 256 apart
      for (uint32_t b0 = 0; b0 + 4 <= len; b0 += 1024) {
        for (uint32_t t = 0; t < 256; t++) {
          uint32_t pp[4];
          bool act[4];
          uint64_t out[4];
          for (int i = 0; i < 4; i++) { pp[i] = b0 + t + 256 * i; act[i] = pp[i] + 4 <= len; if (!act[i]) pp[i] = 0; }
          lz_match_positions<4>(src, len, pp, act, prev.data(), K, K / 4, out);
          for (int i = 0; i < 4; i++) {
            if (!act[i]) continue;
            match[pp[i]] = out[i];
            if ((pp[i] & 63) == 0 && out[i] != lz_match_position(src, len, pp[i], prev.data(), K, K / 4)) return 99;
          }
        }
      }
    }
    // the window kernel's form: a lane runs every 64th position of 1 Ki positions
    if (len >= 4) {
      std::vector<uint64_t> match2(len + 8, 0), match3(len + 8, 0);
      std::vector<uint8_t> padded(len + 16, 0xA5);  // the word-read form over-reads
      memcpy(padded.data(), src, len);
      for (uint32_t wbeg = 0; wbeg + 4 <= len; wbeg += 1024) {
        const uint32_t wend = wbeg + 1024 < len - 3 ? wbeg + 1024 : len - 3;
        for (uint32_t lane = 0; lane < 64; lane++) {
          lz_match_runs<2, true>(padded.data(), len, wbeg + lane, 64, wend, prev.data(), K, K / 4, match2.data());
          lz_match_runs<1, false>(src, len, wbeg + lane, 64, wend, prev.data(), K, K / 4, match3.data());
        }
      }
      for (uint32_t p = 0; p + 4 <= len; p++)
        if (match2[p] != match[p] || match3[p] != match[p]) return 98;
      // the walk the window kernel runs now: byte at best_len first, cheap steps, compares together
      std::vector<uint64_t> match4(len + 8, 0), match5(len + 8, 0);
      for (uint32_t wbeg = 0; wbeg + 4 <= len; wbeg += 1024) {
        const uint32_t wend = wbeg + 1024 < len - 3 ? wbeg + 1024 : len - 3;
        for (uint32_t lane = 0; lane < 64; lane++) {
          lz_match_scan_serial<true>(padded.data(), len, wbeg + lane, 64, wend, prev.data(), K, K / 4, match4.data());
          lz_match_scan_serial<false>(src, len, wbeg + lane, 64, wend, prev.data(), K, K / 4, match5.data());
        }
      }
      for (uint32_t p = 0; p + 4 <= len; p++)
        if (match4[p] != match[p] || match5[p] != match[p]) return 97;
    }
    // macro step of every position (+ literal runs), then the walk in small
    // resumable slices like the kernel's ring, then the symbol emission
    std::vector<uint32_t> brefs(len + 8, 0), steps(len + 8, 0);
    auto get = [&](uint32_t j) { return match[j]; };
    for (uint32_t p = 0; p < len; p++) {
      MacroStep m = lz_macro_position(p, len, good_match, get);
      brefs[p] = m.bref;
      steps[p] = m.step;
    }
    for (uint32_t p = 0; p < len; p++) {  // literal runs, capped and cut at 256-position tiles like the kernel
      if (brefs[p]) continue;
      uint32_t run = 1;
      const uint32_t tile_end = (p / 256 + 1) * 256 + 32;
      while (run < MAX_LIT_RUN && p + run < len && p + run < tile_end && brefs[p + run] == 0) run++;
      steps[p] = macro_literal_run(run);
    }
    std::vector<uint64_t> bitmap((len + 63) / 64 + 1, ~0ull);
    std::vector<uint32_t> tile_sym(len / WALK_TILE + 2, 0xFFFFFFFFu);
    std::vector<uint32_t> syms(len + 8);
    std::vector<BlockDesc> blocks(len / 65277 + 4);
    WalkState ws;
    lz_walk_init(ws);
    auto gets = [&](uint32_t j) { return steps[j]; };
    uint32_t stop = 0;
    while (ws.p < len) {
      stop = ws.p + 64;  // ring window
      lz_walk_advance(ws, len, stop, 48, gets, bitmap.data(), tile_sym.data(), blocks.data());
    }
    uint32_t nblk = lz_walk_finish(ws, len, bitmap.data(), tile_sym.data(), blocks.data());
    bool use_tiles = getenv("SIM_PARSE_TILES") != nullptr;
    if (use_tiles) nblk = parse_tiles_model(src, len, brefs.data(), steps.data(), syms.data(), blocks.data());
    if (const char *seg = getenv("SIM_PARSE_SEGMENTS")) {  // positions per segment
      if (len >= 4) {
        use_tiles = true;  // (the symbols are in place)
        nblk = parse_segments_model(src, len, brefs.data(), steps.data(), (uint32_t)atoi(seg), syms.data(), blocks.data());
      }
    }
    for (uint32_t t = 0; !use_tiles && t * WALK_TILE < len; t++) {
      uint32_t idx = tile_sym[t];
      for (uint32_t p = t * WALK_TILE; p < len && p < (t + 1) * WALK_TILE; p++) {
        if (!((bitmap[p >> 6] >> (p & 63)) & 1)) continue;
        MacroStep m;
        m.bref = brefs[p];
        m.step = steps[p];
        lz_emit_position(src, p, m, syms.data(), idx);
        idx += macro_sym_count(m);
      }
    }
    uint32_t lit_freq[288], dist_freq[32], codelen_freq[19] = {0}, dyn_lit[288], dyn_dist[32],
        dyn_codelen[32], fix_lit[288], fix_dist[32], codelen_syms[320], heap[577];
    memset(dyn_lit, 0, sizeof dyn_lit); memset(dyn_dist, 0, sizeof dyn_dist);
    memset(dyn_codelen, 0, sizeof dyn_codelen); memset(heap, 0, sizeof heap);
    BlockCoder c;
    c.lit_freq = lit_freq; c.dist_freq = dist_freq; c.codelen_freq = codelen_freq;
    c.dyn_lit = dyn_lit; c.dyn_dist = dyn_dist; c.dyn_codelen = dyn_codelen;
    c.fix_lit = fix_lit; c.fix_dist = fix_dist; c.codelen_syms = codelen_syms; c.heap = heap;
    huff_fixed_encoders(fix_lit, fix_dist);
    // SIM_EMIT_BLOCKS: the blocks as deflate_plan / _counts / _codelen / _scan_kernel code them -- each block by
    // itself as far as its bits do not depend on the blocks before, then the counts of the code-length symbols
    // added up, the 19-symbol code, the choice and the offsets along the stream
    struct Plan {
      uint32_t dyn_lit[288], dyn_dist[32], dyn_codelen[32], codelen_syms[320], own_freq[19], cum_freq[19];
      int codelen_syms_len, hlit, hdist, hclen, kind;
      uint64_t flen, dsb, dlen, dbits, bit_start, bit_end;
    };
    const bool by_blocks = getenv("SIM_EMIT_BLOCKS") != nullptr;
    std::vector<Plan> plans(by_blocks ? nblk : 0);
    if (by_blocks) {
      for (uint32_t b = 0; b < nblk; b++) {  // plan
        const BlockDesc &bd = blocks[b];
        Plan &P = plans[b];
        memset(lit_freq, 0, sizeof lit_freq);
        memset(dist_freq, 0, sizeof dist_freq);
        memset(codelen_freq, 0, sizeof codelen_freq);
        for (uint32_t k = 0; k < bd.n_syms; k++) {
          uint32_t s = syms[bd.sym_start + k];
          if ((s >> 9) == 0) lit_freq[s]++;
          else { lit_freq[length_to_sym(s & 0x1FF)]++; dist_freq[dist_to_sym(s >> 9)]++; }
        }
        lit_freq[LITLEN_EOB] = 1;
        coder_make_dynamic_syms(c);
        P.flen = 3 + coder_symbols_bits(c, fix_lit, fix_dist);
        P.dsb = coder_symbols_bits(c, dyn_lit, dyn_dist);
        memcpy(P.dyn_lit, dyn_lit, sizeof dyn_lit); memcpy(P.dyn_dist, dyn_dist, sizeof dyn_dist);
        memcpy(P.codelen_syms, codelen_syms, sizeof codelen_syms); memcpy(P.own_freq, codelen_freq, sizeof P.own_freq);
        P.codelen_syms_len = c.codelen_syms_len; P.hlit = c.hlit; P.hdist = c.hdist;
      }
      uint32_t run[19] = {0};
      for (uint32_t b = 0; b < nblk; b++)  // counts
        for (int k = 0; k < 19; k++) { run[k] += plans[b].own_freq[k]; plans[b].cum_freq[k] = run[k]; }
      for (uint32_t b = 0; b < nblk; b++) {  // codelen
        Plan &P = plans[b];
        memcpy(codelen_freq, P.cum_freq, sizeof P.cum_freq);
        coder_make_dynamic_codelen(c);
        uint64_t head = 3 + 5 + 5 + 4 + 3 * (uint64_t)(c.hclen + 4) + P.dsb, acc = 0, own = 0;
        for (int sym = 0; sym <= CODELEN_SYM_MAX; sym++) {
          const uint32_t rb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
          acc += (uint64_t)P.cum_freq[sym] * ((dyn_codelen[sym] & 0x1F) + rb);
          own += (uint64_t)P.own_freq[sym] * ((dyn_codelen[sym] & 0x1F) + rb);
        }
        P.dlen = head + acc; P.dbits = head + own; P.hclen = c.hclen;
        memcpy(P.dyn_codelen, dyn_codelen, sizeof dyn_codelen);
      }
      uint64_t bits = 0;
      for (uint32_t b = 0; b < nblk; b++) {  // scan
        Plan &P = plans[b];
        const uint32_t pending = (uint32_t)(bits & 7), src_len = blocks[b].src_len;
        const uint64_t nlen = 3 + (uint64_t)(8 - ((pending + 3) % 8)) + (4 + (uint64_t)src_len) * 8;
        P.kind = (nlen <= P.dlen && nlen <= P.flen) ? 0 : P.flen <= P.dlen ? 1 : 2;
        const uint64_t sbits = (uint64_t)(((pending + 3u + 7u) & ~7u) - pending) + (4 + (uint64_t)src_len) * 8;
        P.bit_start = bits;
        bits += P.kind == 0 ? sbits : P.kind == 1 ? P.flen : P.dbits;
        P.bit_end = bits;
      }
    }
    for (uint32_t b = 0; b < nblk; b++) {
      const BlockDesc &bd = blocks[b];
      const bool final = b + 1 == nblk;
      adler = adler_update_serial(adler, src + bd.src_start, bd.src_len);
      int kind;
      if (by_blocks) {
        const Plan &P = plans[b];
        if (P.bit_start != (uint64_t)w.out.size() * 8 + (uint64_t)w.nbits) return 96;  // the scan's offsets are the real ones
        kind = P.kind;
        memcpy(dyn_lit, P.dyn_lit, sizeof dyn_lit); memcpy(dyn_dist, P.dyn_dist, sizeof dyn_dist);
        memcpy(dyn_codelen, P.dyn_codelen, sizeof dyn_codelen); memcpy(codelen_syms, P.codelen_syms, sizeof codelen_syms);
        c.codelen_syms_len = P.codelen_syms_len; c.hlit = P.hlit; c.hdist = P.hdist; c.hclen = P.hclen;
      } else {
      memset(lit_freq, 0, sizeof lit_freq);
      memset(dist_freq, 0, sizeof dist_freq);
      for (uint32_t k = 0; k < bd.n_syms; k++) {
        uint32_t s = syms[bd.sym_start + k];
        if ((s >> 9) == 0) lit_freq[s]++;
        else { lit_freq[length_to_sym(s & 0x1FF)]++; dist_freq[dist_to_sym(s >> 9)]++; }
      }
      lit_freq[LITLEN_EOB] = 1;
      coder_make_dynamic(c);
      uint64_t flen, dlen;
      kind = coder_choose(c, bd.src_len, w.nbits, flen, dlen);
      }
      if (nk < max_kinds) kinds[nk] = kind;
      nk++;
      if (kind == 0) {
        w.put(final ? 1 : 0, 3);
        w.flush();
        w.put(bd.src_len & 0xFFFF, 16);
        w.put((~bd.src_len) & 0xFFFF, 16);
        for (uint32_t i = 0; i < bd.src_len; i++) w.put(src[bd.src_start + i], 8);
      } else {
        const uint32_t *hl = kind == 1 ? fix_lit : dyn_lit, *hd = kind == 1 ? fix_dist : dyn_dist;
        w.put((final ? 1 : 0) | (kind << 1), 3);
        if (kind == 2) {
          int items = dyn_header_items(c);
          for (int i = 0; i < items; i++) { uint32_t v; int n; dyn_header_item(c, i, v, n); w.put(v, n); }
        }
        for (uint32_t k = 0; k <= bd.n_syms; k++) {
          uint32_t s = k < bd.n_syms ? syms[bd.sym_start + k] : (uint32_t)LITLEN_EOB;
          uint64_t v; int n;
          symbol_bits(s, hl, hd, v, n);
          w.put(v, n);
        }
      }
      if (by_blocks && plans[b].bit_end != (uint64_t)w.out.size() * 8 + (uint64_t)w.nbits) return 95;
    }
    w.flush();
  }
  *out_len = w.out.size();
  *adler_out = adler;
  *n_kinds = nk;
  if (w.out.size() > dst_cap) return 16;
  memcpy(dst, w.out.data(), w.out.size());
  return 0;
}

// Huffman.lengths_of_freqs two ways: the reference's heap (huff_lengths_of_freqs) and the two
// queues the kernel uses (huff_lengths_of_freqs_tq).  Returns 0 when the lengths are equal.
extern "C" int sim_huff_lengths(const uint32_t *freqs, int max_sym, int max_code_len, uint32_t *out_heap,
                                uint32_t *out_tq) {
  std::vector<uint32_t> heap(2 * (max_sym + 1) + 8, 0), scratch(4 * (max_sym + 1) + 8, 0);
  huff_lengths_of_freqs(heap.data(), out_heap, freqs, max_sym, max_code_len);
  huff_lengths_of_freqs_tq(scratch.data(), out_tq, freqs, max_sym, max_code_len);
  for (int i = 0; i <= max_sym; i++)
    if (out_heap[i] != out_tq[i]) return 1 + i;
  return 0;
}
