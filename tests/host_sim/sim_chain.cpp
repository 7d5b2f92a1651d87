// Host model of lz_chain_kernel's round algorithm (zipc_amd/csrc/deflate.hip).
// TEST TOOLING ONLY.  The kernel relies on "of several lanes storing to one LDS
// address, exactly one store lands"; this model replays the same steps for 1024
// virtual threads with the landing store chosen at random, and its links must
// equal the reference's serial insert_hash chain for any choice.
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "../../zipc_amd/csrc/zd_common.h"

using namespace zd;

// B_first, own_lo, own_hi: lz_chain_segments_kernel's form -- rounds from B_first on with an empty table, the links of
// [own_lo, own_hi] stored, nothing inserted behind own_hi (the whole stream: 0, 0, len - 4)
static void chain_rounds(const uint8_t *s, uint32_t len, uint16_t *prev, uint32_t seed, int *max_turns,
                         uint32_t B_first, uint32_t own_lo, uint32_t own_hi);
extern "C" void sim_chain(const uint8_t *s, uint32_t len, uint16_t *prev, uint32_t seed, int *max_turns) {
  *max_turns = 0;
  if (len < 4) return;
  chain_rounds(s, len, prev, seed, max_turns, 0, 0, len - 4);
}
// the stream's links made segment by segment of seg_positions (a multiple of 16384), each from 32768 positions before
extern "C" void sim_chain_segments(const uint8_t *s, uint32_t len, uint16_t *prev, uint32_t seed, uint32_t seg_positions) {
  if (len < 4) return;
  int mt = 0;
  for (uint32_t lo = 0; lo <= len - 4; lo += seg_positions) {
    const uint32_t hi = len - 4 - lo >= seg_positions ? lo + seg_positions - 1 : len - 4;
    chain_rounds(s, len, prev, seed + lo, &mt, lo > 32768 ? lo - 32768 : 0, lo, hi);
  }
}
static void chain_rounds(const uint8_t *s, uint32_t len, uint16_t *prev, uint32_t seed, int *max_turns,
                         uint32_t B_first, uint32_t own_lo, uint32_t own_hi) {
  const uint32_t T = 1024, SWEEP_PERIOD = 16384, SWEEP_MARK = 20000;  // CHAIN_ROUND positions per round
  const int PLAIN_TURNS = 4;
  const int NEAR = 8;
  std::vector<uint16_t> head(32768);
  std::vector<uint16_t> hs(T + 2 * NEAR, 0xFFFF);
  const uint32_t max_pos = own_hi;
  (void)len;
  srand(seed);
  for (uint32_t B = B_first; B <= max_pos; B += T) {
    if ((B % SWEEP_PERIOD) == 0) {
      const uint16_t mark = (uint16_t)(B + SWEEP_MARK);
      for (uint32_t i = 0; i < 32768; i++) {
        bool keep = false;
        if (B != B_first) { uint32_t d = (B - head[i]) & 0xFFFFu; keep = d >= 1 && d <= 32768; }
        if (!keep) head[i] = mark;
      }
    }
    std::vector<uint32_t> h(T), e_old(T), near_pred(T, 0);
    std::vector<char> active(T), reader(T), writer(T), pending(T), notmax(T, 0);
    std::vector<int> pred_local(T, -1);
    for (uint32_t t = 0; t < T; t++) {
      uint32_t p = B + t;
      active[t] = p <= max_pos;
      h[t] = active[t] ? hash4(load_u32_le(s + p)) : 0xFFFFu;
      hs[NEAR + t] = (uint16_t)h[t];
    }
    for (uint32_t t = 0; t < T; t++) e_old[t] = active[t] ? head[h[t]] : 0;
    for (uint32_t t = 0; t < T; t++) { reader[t] = active[t]; writer[t] = active[t]; pending[t] = active[t]; }
    int turns = 0;
    for (;;) {
      if (turns == PLAIN_TURNS) {
        for (uint32_t t = 0; t < T; t++) {
          bool has_succ = false;
          if (active[t]) {
            for (int k = NEAR; k >= 1; k--) if (hs[NEAR + t - k] == h[t]) near_pred[t] = k;
            for (int k = 1; k <= NEAR; k++) has_succ |= hs[NEAR + t + k] == h[t];
          }
          reader[t] = active[t] && near_pred[t] == 0;
          if (has_succ) { writer[t] = 0; pending[t] = 0; }
        }
      }
      turns++;
      // all pending threads store; a random one per address lands: apply in random order
      std::vector<uint32_t> order;
      for (uint32_t t = 0; t < T; t++) if (pending[t]) order.push_back(t);
      for (size_t i = order.size(); i > 1; i--) { size_t j = rand() % i; std::swap(order[i - 1], order[j]); }
      for (uint32_t t : order) head[h[t]] = (uint16_t)(B + t);
      bool any = false;
      for (uint32_t t = 0; t < T; t++) {
        if (reader[t] || writer[t]) {
          uint32_t r_local = ((uint32_t)head[h[t]] - B) & 0xFFFFu;
          if (r_local == t) pending[t] = 0;
          else if (r_local < t) { if (reader[t] && (int)r_local > pred_local[t]) pred_local[t] = (int)r_local; }
          else notmax[t] = 1;
        }
      }
      for (uint32_t t = 0; t < T; t++) any |= pending[t];
      if (!any) break;
    }
    if (turns > *max_turns) *max_turns = turns;
    for (uint32_t t = 0; t < T; t++) if (writer[t] && !notmax[t]) head[h[t]] = (uint16_t)(B + t);
    for (uint32_t t = 0; t < T; t++) {
      if (!active[t]) continue;
      uint32_t p = B + t, d;
      if (near_pred[t]) d = near_pred[t];
      else if (pred_local[t] >= 0) d = t - (uint32_t)pred_local[t];
      else { d = (p - e_old[t]) & 0xFFFFu; if (d > 32768) d = 0; }
      if (p >= own_lo) prev[p] = (uint16_t)d;
    }
  }
}

// the reference's serial chain (insert_hash zd.ml:1150-1152) as distances
extern "C" void sim_chain_serial(const uint8_t *s, uint32_t len, uint16_t *prev) {
  if (len < 4) return;
  std::vector<int64_t> head(32768, -1);
  for (uint32_t p = 0; p + 4 <= len; p++) {
    uint32_t h = hash4(load_u32_le(s + p));
    int64_t q = head[h];
    prev[p] = (q >= 0 && p - q <= 32768) ? (uint16_t)(p - q) : 0;
    head[h] = p;
  }
}
