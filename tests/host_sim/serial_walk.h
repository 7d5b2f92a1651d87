// serial_walk.h -- TEST TOOLING ONLY.  The lazy parse as a plain serial walk over the
// macro steps of deflate_lane.h (one step after the other from position 0): the
// second opinion tests/host_sim/sim_deflate.cpp holds against the kernel's tile-wise
// pointer-doubling parse.  Not part of the product.
#pragma once

#include "../../zipc_amd/csrc/deflate_lane.h"

namespace zd {

constexpr uint32_t WALK_TILE = 256;  // positions per symbol-emission tile
constexpr uint32_t MAX_LIT_RUN = 32;  // literal positions one walk step may cover

// A position with no match is a literal; `run` consecutive such positions
// (1 <= run <= MAX_LIT_RUN, all without a match) are covered by one walk step.
ZD_HD uint32_t macro_literal_run(uint32_t run) { return run | (run << 16); }

// One lane per stream, resumable: the walk consumes steps[p] through get(p) for
// p < stop (what the caller has staged) and at most `budget` steps per call.
// Outputs: visited bitmap (1 bit per position, every word of the stream written),
// tile_sym[t] = index of the first symbol emitted at or after position
// t * WALK_TILE, the block list (greedy cut at 65534 source bytes,
// zd.ml:1119-1120) and the symbol count.
struct WalkState {
  uint64_t acc;   // bits of bitmap word `word` gathered so far
  uint32_t p, nsym, nblk, blk_src_start, blk_sym_start, word, tile;
};
ZD_HD void lz_walk_init(WalkState &w) {
  w.acc = 0;
  w.p = w.nsym = w.nblk = w.blk_src_start = w.blk_sym_start = w.word = w.tile = 0;
}
ZD_HD void walk_cut(WalkState &w, BlockDesc *blocks, uint32_t at) {
  BlockDesc b;
  b.src_start = w.blk_src_start; b.src_len = at - w.blk_src_start;
  b.sym_start = w.blk_sym_start; b.n_syms = w.nsym - w.blk_sym_start;
  blocks[w.nblk++] = b;
  w.blk_src_start = at;
  w.blk_sym_start = w.nsym;
}
// mark positions [a, b) visited
ZD_HD void walk_mark(WalkState &w, uint64_t *bitmap, uint32_t a, uint32_t b) {
  while (a < b) {
    const uint32_t wd = a >> 6;
    if (wd != w.word) {
      bitmap[w.word] = w.acc;
      for (uint32_t k = w.word + 1; k < wd; k++) bitmap[k] = 0;
      w.acc = 0;
      w.word = wd;
    }
    const uint32_t lo = a & 63;
    const uint32_t end = ((wd + 1) << 6) < b ? 64u : b - (wd << 6);
    const uint64_t m = (end >= 64 ? ~0ull : ((1ull << end) - 1)) & ~((1ull << lo) - 1);
    w.acc |= m;
    a = (wd << 6) + end;
  }
}
template <typename GetStep>
ZD_HD void lz_walk_advance(WalkState &w, uint32_t len, uint32_t stop, int budget, GetStep get,
                           uint64_t *bitmap, uint32_t *tile_sym, BlockDesc *blocks) {
  const uint32_t n_tiles = (len + WALK_TILE - 1) / WALK_TILE;
  const uint32_t lim = stop < len ? stop : len;
  while (w.p < lim && budget-- > 0) {
    const uint32_t p = w.p;
    while (w.tile < n_tiles && w.tile * WALK_TILE <= p) tile_sym[w.tile++] = w.nsym;
    const uint32_t st = get(p);
    uint32_t lits = macro_lits(st);
    const uint32_t adv = macro_advance(st);
    const uint32_t mlen = adv - lits;  // 0: a run of `lits` literal positions
    if (mlen == 0) {
      walk_mark(w, bitmap, p, p + lits);
      // tiles that start inside the run: one symbol per position before them
      while (w.tile < n_tiles && w.tile * WALK_TILE < p + lits) {
        tile_sym[w.tile] = w.nsym + (w.tile * WALK_TILE - p);
        w.tile++;
      }
    } else {
      walk_mark(w, bitmap, p, p + 1);
    }
    uint32_t q = p;
    // write_block_symbol zd.ml:1118-1123: a cut can only fall where the block
    // already holds 65534 source bytes
    while (lits) {
      const uint32_t room = (uint32_t)MAX_BLOCK_SRC_LEN - (q - w.blk_src_start);
      const uint32_t take = lits < room ? lits : room;
      w.nsym += take;
      q += take;
      lits -= take;
      if (lits) walk_cut(w, blocks, q);
    }
    if (mlen) {
      if ((q - w.blk_src_start) + mlen > (uint32_t)MAX_BLOCK_SRC_LEN) walk_cut(w, blocks, q);
      w.nsym += 1;
      q += mlen;
    }
    w.p = q;
  }
}
// after the last position: flush the bitmap and tile table, close the final
// block (always present, zd.ml:1216).  Returns the number of blocks.
ZD_HD uint32_t lz_walk_finish(WalkState &w, uint32_t len, uint64_t *bitmap, uint32_t *tile_sym,
                              BlockDesc *blocks) {
  const uint32_t n_words = (len + 63) >> 6, n_tiles = (len + WALK_TILE - 1) / WALK_TILE;
  while (w.tile < n_tiles) tile_sym[w.tile++] = w.nsym;
  if (n_words) {
    bitmap[w.word] = w.acc;
    for (uint32_t k = w.word + 1; k < n_words; k++) bitmap[k] = 0;
  }
  walk_cut(w, blocks, len);
  return w.nblk;
}


}  // namespace zd
