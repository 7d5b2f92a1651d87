// deflate_lane.h -- the lane-serial pieces of the deflate pipeline (deflate.hip).
//
// The reference's encoder (src/zipc_deflate.ml:742-1277) is one sequential loop.
// It is split here along what is and is not order dependent:
//   * the hash chains do not depend on parse decisions, so the chain links
//     (lz_chain_kernel) and, per position, the best match over the first K and
//     first K/4 chain candidates (lz_match_position) are computed for ALL
//     positions in parallel;
//   * the lazy parse is then: a "macro step" per position (lz_macro_position,
//     parallel), the chain of visited positions over the steps (lz_parse_kernel:
//     pointer doubling per tile) and a parallel symbol emission;
//   * per block, code construction and the stored/fixed/dynamic choice
//     (BlockCoder) are serial per stream because codelen_sym_freqs (Q1) and the
//     pending bit count (Q3) carry from block to block.
// Every function cites the reference lines it reproduces; all of it is checked
// byte for byte against the oracle through tests/host_sim.
#pragma once

#include "zd_common.h"

namespace zd {

struct BlockDesc {
  uint32_t src_start, src_len;  // block_src_start / block_src_len (zd.ml:788-789)
  uint32_t sym_start, n_syms;   // slice of the stream's symbol array (no EOB stored)
};

// level_params zd.ml:754-764: (good_match, max_chain_len); the other two are never read
ZD_HD void level_params(int level, int &good_match, int &max_chain) {
  switch (level) {
  case LEVEL_FAST: good_match = 4; max_chain = 4; break;
  case LEVEL_DEFAULT: good_match = 8; max_chain = 128; break;
  case LEVEL_BEST: good_match = 32; max_chain = 4096; break;
  default: good_match = 0; max_chain = 0; break;
  }
}

// Common prefix of s[q..] and s[p..], capped at maxlen (match_bwd/match_fwd,
// zd.ml:1154-1174, without the early-out order: the result is the same length).
// (from: bytes already known to agree, a multiple of 8 not above maxlen)
ZD_HD uint32_t common_prefix(const uint8_t *s, uint32_t q, uint32_t p, uint32_t maxlen, uint32_t from = 0) {
  uint32_t i = from;
  while (i + 8 <= maxlen) {
    uint64_t x = load_u64_le(s + q + i) ^ load_u64_le(s + p + i);
    if (x) return i + (uint32_t)(__builtin_ctzll(x) >> 3);
    i += 8;
  }
  while (i < maxlen && s[q + i] == s[p + i]) i++;
  return i;
}

// ... in the LDS window (WORDS: s 4-byte aligned, may be over-read by 11 bytes): aligned words and a funnel shift, like
// every other read of the window -- a misaligned 8-byte LDS read stalls the pipe for ~55 clocks (SQ_LDS_IDX_ACTIVE per
// instruction on 1 MiB of zeros, where every position's one candidate is compared over 258 bytes)
// (S: what the window's bytes are read through -- a pointer indexed by stream position, or deflate.hip's RingSrc, which
// maps a position to its place in a ring; P likewise for the links.  The walks below are written on s[x], s + x,
// load_u64_words(s, x) and prev[x] alone.)
template <bool WORDS, typename S>
ZD_HD uint32_t common_prefix_t(S s, uint32_t q, uint32_t p, uint32_t maxlen, uint32_t from) {
  if constexpr (!WORDS) return common_prefix(s, q, p, maxlen, from);
  uint32_t i = from;
  // (the first 16 bytes as they come: on text most compares end there, and three aligned reads and two funnel shifts
  // per 8 bytes cost it more than the odd stall -- lz_match 95.5 -> 101.3 ms with words from the first byte on)
#ifndef ZD_PREFIX_PLAIN
#define ZD_PREFIX_PLAIN 2
#endif
  for (int k = 0; k < ZD_PREFIX_PLAIN && i + 8 <= maxlen; k++) {
    const uint64_t x = load_u64_le(s + q + i) ^ load_u64_le(s + p + i);
    if (x) return i + (uint32_t)(__builtin_ctzll(x) >> 3);
    i += 8;
  }
  while (i + 8 <= maxlen) {
    const uint64_t x = load_u64_words(s, q + i) ^ load_u64_words(s, p + i);
    if (x) return i + (uint32_t)(__builtin_ctzll(x) >> 3);
    i += 8;
  }
  while (i < maxlen && s[q + i] == s[p + i]) i++;
  return i;
}

// (A chain link out of the LDS window is read as it is, 16 bits.  ds_read_u16 at a random address of 2 mod 4 measures 41
// clocks per wave instruction where ds_read_b32 measures 6 -- tools/probes/lds_costs.hip -- and half of a walk's links sit at
// such addresses; but the walks wait for vector instruction issue, not for the LDS: reading the aligned word that holds
// the link and taking the half out of it cost lz_match 6 % on the benchmark's symbols, 4 % on text and 11 % on 3-bit
// symbols, round 5.)
// find_backref zd.ml:1176-1201 as a pure function of the position: walks the
// hash chain of p (prev[] holds the distance to the previous position with the
// same hash, 0 = none within 32768) and returns, packed like the reference's
// backref (dist << 9 | len), the best match among the first K candidates (low
// word) and among the first Kq = K/4 candidates (high word): longest common
// prefix, nearest candidate on ties, only if longer than min_match_len - 1 = 3.
// The reference's running threshold starts at the pending match length instead
// of 3; lz_macro_position applies that comparison afterwards, which selects the
// same candidate (the maximum does not depend on the threshold).
ZD_HD uint64_t lz_match_position(const uint8_t *s, uint32_t len, uint32_t p, const uint16_t *prev,
                                 int K, int Kq) {
  const uint32_t maxlen = len - p < (uint32_t)MAX_MATCH_LEN ? len - p : (uint32_t)MAX_MATCH_LEN;
  uint32_t best_len = MIN_MATCH_LEN - 1, best = 0, snap = 0;
  bool snapped = false;
  uint32_t q = p;
  int steps = 0;
  if (best_len < maxlen) {  // zd.ml:1181
    for (;;) {
      uint32_t d = prev[q];
      if (d == 0 || steps == K) break;
      q -= d;
      if (p - q > (uint32_t)MAX_MATCH_DIST) break;  // zd.ml:1187
      steps++;
      uint32_t l = common_prefix(s, q, p, maxlen);
      if (l > best_len) {
        best_len = l;
        best = ((p - q) << 9) | l;
      }
      if (steps == Kq) { snap = best; snapped = true; }
      if (l == maxlen) break;  // zd.ml:1194: nothing later can be longer
    }
  }
  if (!snapped) snap = best;  // chain ended before K/4 candidates
  if (Kq == 0) snap = 0;
  return (uint64_t)best | ((uint64_t)snap << 32);
}

// lz_match_position for NP positions at once, their chain walks interleaved so
// that the NP link loads, then the NP source loads, are in flight together (the
// walk is a chain of dependent loads: one lane alone keeps a single load in
// flight).  Same result as NP calls of lz_match_position.
template <int NP>
ZD_HD void lz_match_positions(const uint8_t *s, uint32_t len, const uint32_t *p, const bool *act,
                              const uint16_t *prev, int K, int Kq, uint64_t *out) {
  uint32_t q[NP], best_len[NP], best[NP], snap[NP], maxlen[NP];
  int steps[NP];
  bool snapped[NP], done[NP];
  uint64_t pw[NP];
  bool any = false;
#pragma unroll
  for (int i = 0; i < NP; i++) {
    q[i] = p[i];
    best_len[i] = MIN_MATCH_LEN - 1;
    best[i] = 0;
    snap[i] = 0;
    steps[i] = 0;
    snapped[i] = false;
    maxlen[i] = 0;
    pw[i] = 0;
    done[i] = true;
    if (act[i]) {
      maxlen[i] = len - p[i] < (uint32_t)MAX_MATCH_LEN ? len - p[i] : (uint32_t)MAX_MATCH_LEN;
      done[i] = !(best_len[i] < maxlen[i]);  // zd.ml:1181
      if (maxlen[i] >= 8) pw[i] = load_u64_le(s + p[i]);
    }
    any |= !done[i];
  }
  while (any) {
    uint32_t d[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) d[i] = done[i] ? 0u : (uint32_t)prev[q[i]];
    uint64_t a[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
      a[i] = 0;
      if (!done[i]) {
        if (d[i] == 0 || steps[i] == K) done[i] = true;
        else {
          q[i] -= d[i];
          if (p[i] - q[i] > (uint32_t)MAX_MATCH_DIST) done[i] = true;  // zd.ml:1187
          else if (maxlen[i] >= 8) a[i] = load_u64_le(s + q[i]);
        }
      }
    }
    any = false;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      if (!done[i]) {
        steps[i]++;
        uint32_t l;
        if (maxlen[i] >= 8) {
          const uint64_t x = a[i] ^ pw[i];
          l = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : common_prefix(s, q[i], p[i], maxlen[i]);
        } else {
          l = common_prefix(s, q[i], p[i], maxlen[i]);
        }
        if (l > best_len[i]) {
          best_len[i] = l;
          best[i] = ((p[i] - q[i]) << 9) | l;
        }
        if (steps[i] == Kq) { snap[i] = best[i]; snapped[i] = true; }
        if (l == maxlen[i]) done[i] = true;  // zd.ml:1194
      }
      any |= !done[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NP; i++) {
    if (!snapped[i]) snap[i] = best[i];
    if (Kq == 0) snap[i] = 0;
    out[i] = (uint64_t)best[i] | ((uint64_t)snap[i] << 32);
  }
}

// lz_match_position for runs of positions -- a run is pbeg, pbeg + stride, ...
// < pend, one position after the other -- as ONE loop: an iteration is either a
// chain step of a run's current position or its last step plus the start of the
// run's next position.  The lanes of a wave work on different positions at
// their own pace, so a wave iterates max-over-lanes of the SUM of the chain
// lengths of a lane's positions -- not the sum over positions of the max over
// lanes that walking position groups in lockstep costs (chain lengths vary a lot
// between neighbouring positions).  NP independent runs per lane are stepped in
// the same iteration: a step is a chain of dependent reads, NP of them overlap.
// The common path of a step is branch free; long matches (first 8 bytes equal)
// and the last 7 positions of the stream take common_prefix.
// WORDS: read candidates with load_u64_words (s is 4-byte aligned and may be
// over-read by 11 bytes) -- the LDS window.  out is indexed by position.  Same
// results as lz_match_position.
// What "no link" reads as through a links accessor P: 0 in the tables of lz_chain; 0xFFFF in an LDS window, whose kernel
// stages them that way (a candidate that far back is out of range whatever the position: the second form of the walk
// has no test for it).
template <typename P> struct LinkNone { static constexpr uint32_t value = 0u; };
struct WinLinks {
  const uint16_t *b;
  ZD_HD uint32_t operator[](uint32_t i) const { return b[i]; }
};
template <> struct LinkNone<WinLinks> { static constexpr uint32_t value = 0xFFFFu; };
struct MatchRun {
  uint32_t p, q, best_len, best, maxlen, steps;
  uint32_t snap;   // best after Kq candidates, SNAP_NONE before
  uint32_t dn;     // prev[q], read ahead: whether the chain goes on is known before the next step
  uint32_t alive;  // 0 / 1 (an integer, like every flag of the step: see match_run_step)
  uint64_t pw;
};
constexpr uint32_t SNAP_NONE = 0xFFFFFFFFu;  // no snapshot yet (a real one has length bits <= 258)
template <bool WORDS, typename S, typename P>
ZD_HD void match_run_start(MatchRun &r, S s, uint32_t len, uint32_t p, uint32_t pend, P prev) {
  r.alive = p < pend ? 1u : 0u;
  r.p = r.alive ? p : (pend ? pend - 1u : 0u);  // a finished run parks on a valid position
  r.q = r.p;
  r.best_len = MIN_MATCH_LEN - 1;
  r.best = 0; r.snap = SNAP_NONE; r.steps = 0;
  r.maxlen = len - r.p < (uint32_t)MAX_MATCH_LEN ? len - r.p : (uint32_t)MAX_MATCH_LEN;
  r.pw = 0;
  if (WORDS) r.pw = load_u64_words(s, r.p);
  else if (r.maxlen >= 8) r.pw = load_u64_le(s + r.p);
  r.dn = prev[r.p];
}
// One iteration of a run: a chain step of its current position.  Returns true when
// the position is finished (its result has been written to out[p]); the caller
// starts the run's next position.  The step that compares a candidate also reads
// that candidate's link, so a position ends in the step of its LAST candidate
// (max(1, chain length) steps per position, not chain length + 1).
// Flags are integers and updates are selects (round 3): the workgroup's 16 waves share the CU's one
// scalar issue per clock -- this kernel's tighter bound -- and a loop-carried bool costs three scalar
// instructions per update, a divergent `if` four to eight.  A run without a position goes through the
// step like any other (it parks on a valid position and never walks); only the long compare and
// the store of a finished position stay behind branches.
// (sink(p, best, snap): where a finished position's two answers go)
template <bool WORDS, typename Sink, typename S, typename P>
ZD_HD bool match_run_step_to(MatchRun &r, S s, P prev, uint32_t K, uint32_t Kq, Sink sink) {
  const uint32_t qn = r.q - r.dn;
  const bool walk = r.alive != 0 && r.dn != LinkNone<P>::value && r.steps != K && r.best_len < r.maxlen && r.p - qn <= (uint32_t)MAX_MATCH_DIST;  // zd.ml:1181,1187
  const uint32_t qc = walk ? qn : r.p;
  uint64_t x = 0;
  if (WORDS) x = load_u64_words(s, qc) ^ r.pw;
  else if (r.maxlen >= 8) x = load_u64_le(s + qc) ^ r.pw;
  const uint32_t d2 = prev[qc];
  uint32_t l = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
  if (walk && (x == 0 || r.maxlen < 8)) {
    // The first 8 bytes agree: the long compare -- unless the candidate cannot beat best_len
    // anyway, which takes agreement in the 8 bytes that END at best_len as well.  On text a
    // quarter of the candidates get here (8.5 per position on the reference's own documents)
    // and 91 % of them fail this check; their exact length (8 .. best_len) is not needed.
    bool compare = true;
    if (WORDS && x == 0 && r.best_len >= 8u) {
      const uint32_t toff = r.best_len - 7u;  // best_len < maxlen: these bytes lie inside both strings
      compare = load_u64_words(s, qc + toff) == load_u64_words(s, r.p + toff);
    }
    if (compare) l = common_prefix_t<WORDS>(s, qc, r.p, r.maxlen, r.maxlen >= 8u ? 8u : 0u);  // (maxlen >= 8 here means x == 0: the first 8 agree)
  }
  const uint32_t steps = r.steps + (walk ? 1u : 0u);
  const bool better = walk && l > r.best_len;
  const uint32_t best = better ? (((r.p - qc) << 9) | l) : r.best;
  r.best_len = better ? l : r.best_len;
  r.best = best;
  r.snap = (walk && steps == Kq) ? best : r.snap;
  r.q = qc;
  r.steps = steps;
  r.dn = d2;
  // would the next step walk?  (l < maxlen implies best_len < maxlen; zd.ml:1194:
  // after l == maxlen nothing later can be longer)
  const bool more = walk && l != r.maxlen && d2 != LinkNone<P>::value && steps != K && r.p - qc + d2 <= (uint32_t)MAX_MATCH_DIST;
  const bool fin = r.alive != 0 && !more;
  if (fin) {
    const uint32_t snap = Kq == 0 ? 0u : (r.snap != SNAP_NONE ? r.snap : best);
    sink(r.p, best, snap);
  }
  return fin;
}
template <bool WORDS, typename S, typename P>
ZD_HD bool match_run_step(MatchRun &r, S s, P prev, uint32_t K, uint32_t Kq, uint64_t *out) {
  return match_run_step_to<WORDS>(r, s, prev, K, Kq, [out](uint32_t p, uint32_t best, uint32_t snap) {
    out[p] = (uint64_t)best | ((uint64_t)snap << 32);
  });
}
// A lane's positions are first, first + step, first + 2 step, ... < pend.  Its NP run
// slots draw from ONE cursor over them: a slot that finishes a position takes the
// lane's next one, whichever slot that is.  (With the positions split into NP fixed
// runs a lane needed max over its runs of the run's steps and the early slot idled;
// now it needs about the sum / NP.  No cross-lane traffic -- handing positions out
// across the wave was measured and lost on C2)  Results are
// stored per position, so the order in which a lane takes them does not matter.
// Returns the number of iterations of its loop (what the window kernel's waves go by to
// pick their schedule for the next tile).
template <int NP, bool WORDS>
ZD_HD uint32_t lz_match_runs(const uint8_t *s, uint32_t len, uint32_t first, uint32_t step, uint32_t pend,
                             const uint16_t *prev, int K, int Kq, uint64_t *out) {
  MatchRun r[NP];
  uint32_t iters = 0;
  uint32_t cursor = first;
#pragma unroll
  for (int i = 0; i < NP; i++) {
    match_run_start<WORDS>(r[i], s, len, cursor, pend, prev);
    cursor = cursor < pend ? cursor + step : cursor;  // stays put once past the end (no wrap)
  }
  // On the GPU the loop is left by the whole wave at once (the exit test is
  // wave-uniform): that keeps it ONE loop whose iterations mix positions, instead
  // of a loop per position that the lanes would have to leave together.
  for (;;) {
    bool alive = false;
    iters++;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      if (match_run_step<WORDS>(r[i], s, prev, (uint32_t)K, (uint32_t)Kq, out)) {
        match_run_start<WORDS>(r[i], s, len, cursor, pend, prev);
        cursor = cursor < pend ? cursor + step : cursor;
      }
      alive |= r[i].alive;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
#else
    if (!alive) break;
#endif
  }
  return iters;
}

// ---------------------------------------------------------------------------
// The chain walk of the window kernel (lz_match_window_kernel), second form.  A candidate can only
// become the best match if it is LONGER than the best so far, which takes agreement in the bytes
// at offsets best_len - 1 and best_len -- so those two are looked at first (two single bytes of the
// LDS window, any alignment; on text one byte lets about a quarter of the candidates through, two
// a sixteenth) and a candidate that differs there is passed over without reading its 8 bytes:
// its common prefix is <= best_len, it changes neither the best match nor -- being shorter than
// maxlen -- the end of the walk (zd.ml:1190-1194), and it still counts as one of the K
// candidates.  On 3-bit symbols 7 of 8 candidates go that way, on text about as many.  What makes
// this pay on a wave is the shape of the loop: lanes do up to SCAN_ROUNDS cheap steps (link, two bytes,
// compare) until they stand on a candidate that passes -- or their chain ends -- and only then
// all lanes with such a candidate go through the full compare TOGETHER.  In the first form every
// step of every lane paid for the long-compare branches, because some lane of the 64 always
// took them.
//
// Round 5 counted what the waves of this form do on text (tools/exp_wall.py MATCH_COUNTS=1): 41 % of their vector
// instructions were cheap steps (49 a round of two run slots, 74 % of the slots walking), 30 % compares (15 % of the slots
// in one had a hit) and 23 % the handing out of positions (a tenth of a slot's lanes finishing at a time) -- and all of it
// waits for vector issue.  Since then:
//   * a cheap step is 8 vector instructions under the walking lanes' mask (it was 24 of selects): the run keeps the NEXT
//     candidate t (the step's one subtraction: t - link doubles as the range test, see lim), "no link" reads as 0xFFFF
//     out of the LDS window (LinkNone: the candidate then lies out of range, no test of its own), and the two limits on the
//     count of candidates -- K, and K/4 for the second answer -- are ONE compare with klim: a run that reaches K/4 stops
//     like one at its chain's end, notes its second answer where stopped runs are looked at anyway, and walks on;
//   * the compares of a lane's run slots are one piece of code (a lane with hits in both slots takes them one after the
//     other, which is rare);
//   * finished positions wait until SCAN_HANDOUT lanes of the wave have one (or nothing else is left to do), and one handout
//     serves a lane's first finished slot, whichever it is.
// A run is in ONE of: walking (t is a candidate in range and steps < klim), hit (it stands on q = t + dn, which passed
// the byte test), fin (done, waits for its store), none of them with live set (stopped: scan_run_settle says which
// of the others it becomes) or dead (!live: no position).
// Positions are COORDINATES below 2^31 (the window kernel's are the LDS addresses of its window's bytes): the range test is signed.
struct ScanRun {
  uint32_t p;      // the position
  uint32_t t;      // the next candidate
  uint32_t dn;     // the link of the candidate last looked at, q = t + dn
  uint32_t blm1;   // best_len - 1
  uint32_t pb;     // s[p + best_len - 1] | s[p + best_len] << 16
  uint32_t steps;  // candidates looked at
  uint32_t klim;   // ... at which the run stops next: K/4 (the second answer is noted there), then K
  uint32_t maxlen, best;
  uint32_t snap;   // best after K/4 candidates, SNAP_NONE before
  int32_t lim;     // p - MAX_MATCH_DIST: candidates below are out of range (zd.ml:1187)
  uint64_t pw;     // s[p .. p+8)
};
struct ScanFlags { bool walking, hit, fin, live; };  // (the device keeps them as four lane masks per run slot)
template <typename P>
ZD_HD bool scan_far(const ScanRun &r) {  // is there no candidate t?  zd.ml:1185-1187
  if constexpr (LinkNone<P>::value == 0xFFFFu) return (int32_t)r.t < r.lim;  // (no link: t = q - 0xFFFF < p - 32768)
  else return r.dn == LinkNone<P>::value || (int32_t)r.t < r.lim;
}
template <bool WORDS, typename S, typename P>
ZD_HD ScanFlags scan_run_start(ScanRun &r, S s, uint32_t len, uint32_t p, uint32_t pend, P prev, uint32_t K, uint32_t Kq) {
  const bool alive = p < pend;
  r.p = alive ? p : pend - 1u;  // a run without a position parks on a valid one (pend >= 1)
  r.blm1 = MIN_MATCH_LEN - 2;
  r.best = 0; r.snap = SNAP_NONE; r.steps = 0;
  r.klim = Kq ? Kq : K;
  r.maxlen = len - r.p < (uint32_t)MAX_MATCH_LEN ? len - r.p : (uint32_t)MAX_MATCH_LEN;
  r.lim = (int32_t)r.p - (int32_t)MAX_MATCH_DIST;
  r.pw = 0;
  if (WORDS) r.pw = load_u64_words(s, r.p);
  else if (r.maxlen >= 8) r.pw = load_u64_le(s + r.p);
  else for (uint32_t i = 0; i < r.maxlen; i++) r.pw |= (uint64_t)s[r.p + i] << (8 * i);
  r.pb = ((uint32_t)(r.pw >> 16) & 0xFFu) | (((uint32_t)(r.pw >> 24) & 0xFFu) << 16);  // s[p + 2], s[p + 3]
  r.dn = prev[r.p];
  r.t = r.p - r.dn;
  // zd.ml:1181: no search when even the shortest match does not fit
  const bool walk = alive && r.blm1 + 1u < r.maxlen && K != 0 && !scan_far<P>(r);
  return ScanFlags{walk, false, alive && !walk, alive};
}
// One cheap step of a walking run: its candidate's two bytes at best_len - 1 and best_len, its link, the next candidate.
// (the device's is scan_rounds_lds below)
template <typename S, typename P>
ZD_HD void scan_run_step(ScanRun &r, ScanFlags &f, S s, P prev) {
  const uint32_t h = (uint32_t)s[r.t + r.blm1] | ((uint32_t)s[r.t + r.blm1 + 1u] << 16);  // 2 <= best_len - 1, best_len < maxlen: inside both strings
  r.dn = prev[r.t];
  r.steps++;
  r.t -= r.dn;
  f.hit = h == r.pb;
  f.walking = !(f.hit | scan_far<P>(r) | (r.steps == r.klim));
}
// the full compare of the candidate q = t + dn of the run at p: its common prefix if that is above best_len, else
// something that is not
template <bool WORDS, typename S>
ZD_HD uint32_t scan_hit_length(S s, uint32_t q, uint32_t p, uint64_t pw, uint32_t best_len, uint32_t maxlen) {
  if (maxlen < 8) return common_prefix_t<WORDS>(s, q, p, maxlen, 0u);
  const uint64_t x = (WORDS ? load_u64_words(s, q) : load_u64_le(s + q)) ^ pw;
  if (x) return (uint32_t)(__builtin_ctzll(x) >> 3);
  // the first 8 bytes agree: before the long compare, the 8 bytes that END at best_len
  // (they include the bytes already tested); what fails here is at most best_len long
  if (best_len >= 8u) {
    const uint32_t toff = best_len - 7u;
    const bool same = WORDS ? load_u64_words(s, q + toff) == load_u64_words(s, p + toff)
                            : load_u64_le(s + q + toff) == load_u64_le(s + p + toff);
    if (!same) return 8u;
  }
  return common_prefix_t<WORDS>(s, q, p, maxlen, 8u);
}
// ... and what it does to the run; true: the match is as long as a match can be, the run is done (zd.ml:1194)
template <typename S>
ZD_HD bool scan_run_hit_done(ScanRun &r, S s, uint32_t q, uint32_t l) {
  if (l > r.blm1 + 1u) {
    r.blm1 = l - 1u;
    r.best = ((r.p - q) << 9) | l;
    if (l < r.maxlen) r.pb = (uint32_t)s[r.p + l - 1u] | ((uint32_t)s[r.p + l] << 16);
  }
  return l == r.maxlen;
}
// a run that neither walks nor stands on a hit (full: its last compare was scan_run_hit_done's true): true, it is done;
// false, it walks on
template <typename P>
ZD_HD bool scan_run_settle(ScanRun &r, bool full, uint32_t K) {
  bool fin = full;
  if (r.steps == r.klim) {
    if (r.klim == K) fin = true;
    else { r.snap = r.best; r.klim = K; }  // the first K/4 candidates' answer
  }
  return fin | scan_far<P>(r);
}
ZD_HD uint64_t scan_run_result(const ScanRun &r, uint32_t Kq) {
  const uint32_t snap = Kq == 0 ? 0u : (r.snap != SNAP_NONE ? r.snap : r.best);
  return (uint64_t)r.best | ((uint64_t)snap << 32);
}
#ifndef ZD_SCAN_ROUNDS
#define ZD_SCAN_ROUNDS 8
#endif
#ifndef ZD_SCAN_ROUNDS_DENSE
#define ZD_SCAN_ROUNDS_DENSE 4
#endif
#ifndef ZD_SCAN_DENSE_HITS
#define ZD_SCAN_DENSE_HITS 32
#endif
#ifndef ZD_SCAN_MIN_WALKERS
#define ZD_SCAN_MIN_WALKERS 16
#endif
#ifndef ZD_SCAN_HANDOUT
#define ZD_SCAN_HANDOUT 24
#endif
constexpr int SCAN_ROUNDS = ZD_SCAN_ROUNDS;  // cheap steps between two compares, at most ...
constexpr uint32_t SCAN_DENSE_HITS = ZD_SCAN_DENSE_HITS;  // ... ZD_SCAN_ROUNDS_DENSE when the compare before had this many runs in it (of 128): measured,
                                                          // text (20 a compare) is 6 % faster with 8 rounds than with 4, 3-bit symbols (56) 5 % slower
constexpr int SCAN_MIN_WALKERS = ZD_SCAN_MIN_WALKERS;  // ... and only while this many lanes of the wave still walk
constexpr uint32_t SCAN_HANDOUT = ZD_SCAN_HANDOUT;  // finished positions wait for this many lanes with one

// The same walk for one lane's positions first, first + step, ... < pend, serially (the host
// model's form; the device runs lz_match_scan_pool below on the same pieces).
template <bool WORDS>
ZD_HD void lz_match_scan_serial(const uint8_t *s, uint32_t len, uint32_t first, uint32_t step, uint32_t pend,
                                const uint16_t *prev, int K, int Kq, uint64_t *out) {
  typedef const uint16_t *P;
  for (uint32_t p = first; p < pend; p += step) {
    ScanRun r;
    ScanFlags f = scan_run_start<WORDS>(r, s, len, p, pend, prev, (uint32_t)K, (uint32_t)Kq);
    while (!f.fin) {
      for (int i = 0; i < SCAN_ROUNDS && f.walking; i++) scan_run_step(r, f, s, prev);
      bool full = false;
      if (f.hit) {
        const uint32_t q = r.t + r.dn;
        full = scan_run_hit_done(r, s, q, scan_hit_length<WORDS>(s, q, r.p, r.pw, r.blm1 + 1u, r.maxlen));
        f.hit = false;
      }
      if (!f.walking) { f.fin = scan_run_settle<P>(r, full, (uint32_t)K); f.walking = !f.fin; }
    }
    out[p] = scan_run_result(r, (uint32_t)Kq);
    if (pend - p <= step) break;  // (no wrap near 2^32)
  }
}

#if defined(__HIPCC__)  // device code of the HIP build only (the host models walk one lane at a time)
// The window kernel's schedule: ONE pool of a tile's positions [*, pend) for the whole workgroup,
// behind a counter in LDS.  A wave fetches chunks of POOL_CHUNK positions from it, and inside its
// chunk a run slot that finishes a position takes the next unassigned one, whichever lane it
// belongs to (rank among the slots finishing in the same iteration: ballot + mbcnt).  With
// fixed positions per lane (lz_match_runs above) a wave needs the max over its lanes of a lane's
// total steps and a tile the slowest of its 16 waves; the pool needs about the mean.  The handout
// costs a dozen instructions per iteration; a wave-level pool (each wave its own 1 Ki positions)
// paid for that only on long chains (-15 % on text, +8 % on 3-bit symbols), the tile-wide one pays
// everywhere: -6.5 % on the benchmark's symbols, -2 % on 3-bit symbols, -36 % on text.  Results
// are stored per position, so who walks which position does not matter.  Returns the wave's
// iteration count.
#ifndef ZD_POOL_TAPER
#define ZD_POOL_TAPER 4096
#endif
constexpr uint32_t POOL_TAPER = ZD_POOL_TAPER;
#ifndef ZD_POOL_TAPER_Q
#define ZD_POOL_TAPER_Q 2
#endif
constexpr uint32_t POOL_CHUNK = 256;  // >= 64 * NP: a fresh chunk serves any one handout.  Measured, same box, 128 / 256 / 512:
                                      // C2 5.58 / 5.44-5.48 / 5.83 ms, real text 145.5 / 150.1 / 167.5 ms
// first form of the walk (match_run_step: every candidate's 8 bytes are read): the faster one
// where chains are a candidate or two long.  Returns the wave's iterations: x 64 NP / positions = steps per position / lane use.
// (sink(p, best, first): where a finished position's two answers go -- the kernel's tables)
#ifdef ZD_MATCH_COUNTS  // counting-only build (tools/exp_wall.py MATCH_COUNTS=1): what the waves of the pools below do, summed
// [0] waves  [1] outer iterations  [2] rounds of cheap steps (first form: iterations)  [3] run slots walking in them (of 64 NP a round)
// [4] compare phases run  [5] run slots with a hit in them  [6] handouts run (per slot)  [7] run slots finishing in them
// [8] first form: run slots alive over its iterations
static __device__ unsigned long long zd_match_counts[16];
struct MatchCounts {
  uint32_t c[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
  __device__ void flush(uint32_t lane) {
    if (lane == 0)
      for (int i = 0; i < 9; i++) atomicAdd(&zd_match_counts[i], (unsigned long long)c[i]);
  }
};
#define ZD_COUNT(i, v) (mc.c[i] += (uint32_t)(v))
#else
#define ZD_COUNT(i, v) ((void)0)
#endif
// Where a wave's chunks come from.  take(oldest): wave-uniform, the first position of a fresh chunk of POOL_CHUNK positions, or pend:
// the pool is empty.  oldest (pools with WANTS_OLDEST): the lowest position one of the wave's run slots still walks, ~0 for none.
// TilePool: a counter in LDS over the positions [pbeg, pend) of one tile (it counts from 0: it overshoots the tile by a chunk per
// wave at the end, which must not wrap for a stream near the 4 GiB limit).
// (size: what the chunk taken last holds.  A pool with a taper hands its tile's last positions out in chunks of half the size: a
// tile ends when its last wave has walked its last chunk, and the waves that found the pool empty wait for it at the barrier -- the
// gap between a tile's mean wave and its slowest is 7 to 12 % of the tile, tools/exp_match_phases.py.  The second form's pool has
// one, of POOL_TAPER positions (and chunks of a quarter for the last half of those: 1 MiB members of 3-bit symbols 31.0 -> 30.5 ms per
// 2 GiB, text the same): text 46.0 -> 44.0 ms per GiB (2048 and 8192: the same); the first form's has none: its handouts
// run out of their chunk more often with the small ones, the benchmark's streams 4.5 -> 4.8 ms.)
struct TilePool {
  static constexpr bool WANTS_OLDEST = false;
  uint32_t *pool_next;
  uint32_t pbeg, pend, lane;
  uint32_t taper;  // the tile's last `taper` positions in chunks of half the size (0: none)
  uint32_t size = POOL_CHUNK;
  __device__ __forceinline__ uint32_t take(uint32_t) {
    uint32_t c = 0;
    if (lane == 0) {
      if (taper) {
        const uint32_t seen = __hip_atomic_load(pool_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t left = seen > pend - pbeg ? 0u : pend - pbeg - seen;  // (what another wave takes in between: a chunk more or less)
        const uint32_t small = left <= taper / ZD_POOL_TAPER_Q ? 2u : left <= taper ? 1u : 0u;
        c = atomicAdd(pool_next, POOL_CHUNK >> small) | (small << 30);
      } else {
        c = atomicAdd(pool_next, POOL_CHUNK);
      }
    }
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    size = POOL_CHUNK >> (c >> 30);
    c &= 0x3FFFFFFFu;
    return c < pend - pbeg ? pbeg + c : pend;
  }
};
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64); v = w < v ? w : v; }
  return v;
}

// match_run_step_to for the device alone (round 6): the same step, with every condition a lane mask made of its compares' own
// ballots -- which is what the compiler makes of the shared form's bools too, EXCEPT where a bool is asked for as a mask or an
// integer again: `ballot(fin)` went through v_cndmask 0 / 1 + v_cmp_ne, `steps + (walk ? 1 : 0)` through a select and an add.
// Here the finished runs come back as a mask and the steps are counted by an add-with-carry of the walk mask: three vector
// instructions fewer per run slot and iteration, no scalar instruction more (the mask form that moved MORE to the scalar
// unit lost: tools/experiments/lz_match_walk_masks).  Returns the lanes whose position is finished.
template <typename Sink, typename S, typename P>
__device__ __forceinline__ unsigned long long match_run_step_masks(MatchRun &r, S s, P prev, uint32_t K, uint32_t Kq, Sink sink) {
  auto ballot = [](bool b) { return (unsigned long long)__builtin_amdgcn_ballot_w64(b); };
  auto mine = [](unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); };
  const uint32_t qn = r.q - r.dn;
  const unsigned long long alive_m = ballot(r.alive != 0);
  const unsigned long long walk_m = alive_m & ballot(r.dn != LinkNone<P>::value) & ballot(r.steps != K) & ballot(r.best_len < r.maxlen) &
                                    ballot(r.p - qn <= (uint32_t)MAX_MATCH_DIST);  // zd.ml:1181,1187
  const bool walk = mine(walk_m);
  const uint32_t qc = walk ? qn : r.p;
  const uint64_t x = load_u64_words(s, qc) ^ r.pw;
  const uint32_t d2 = prev[qc];
  uint32_t l = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
  if (walk && (x == 0 || r.maxlen < 8)) {  // (as in match_run_step_to)
    bool compare = true;
    if (x == 0 && r.best_len >= 8u) {
      const uint32_t toff = r.best_len - 7u;
      compare = load_u64_words(s, qc + toff) == load_u64_words(s, r.p + toff);
    }
    if (compare) l = common_prefix_t<true>(s, qc, r.p, r.maxlen, r.maxlen >= 8u ? 8u : 0u);
  }
  unsigned long long carry_out;
  uint32_t steps;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(steps), "=s"(carry_out) : "v"(r.steps), "s"(walk_m));  // steps + walk
  const bool better = mine(walk_m & ballot(l > r.best_len));
  const uint32_t best = better ? (((r.p - qc) << 9) | l) : r.best;
  r.best_len = better ? l : r.best_len;
  r.best = best;
  r.snap = mine(walk_m & ballot(steps == Kq)) ? best : r.snap;
  r.q = qc;
  r.steps = steps;
  r.dn = d2;
  const unsigned long long more_m = walk_m & ballot(l != r.maxlen) & ballot(d2 != LinkNone<P>::value) & ballot(steps != K) &
                                    ballot(r.p - qc + d2 <= (uint32_t)MAX_MATCH_DIST);
  const unsigned long long fin_m = alive_m & ~more_m;
  if (mine(fin_m)) {
    const uint32_t snap = Kq == 0 ? 0u : (r.snap != SNAP_NONE ? r.snap : best);
    sink(r.p, best, snap);
  }
  return fin_m;
}

template <int NP, typename Sink, typename S, typename P, typename Pool>
__device__ __forceinline__ uint32_t lz_match_runs_pool(S s, uint32_t len, Pool &pool, uint32_t pend, uint32_t lane,
                                                        P prev, int K, int Kq, Sink sink) {
  static_assert(64u * NP <= POOL_CHUNK, "chunk");
  MatchRun r[NP];
  uint32_t iters = 0;
#ifdef ZD_MATCH_COUNTS
  MatchCounts mc;
#endif
  auto fetch = [&](bool first) -> uint32_t {
    uint32_t oldest = 0xFFFFFFFFu;
    if (Pool::WANTS_OLDEST && !first) {
#pragma unroll
      for (int i = 0; i < NP; i++) oldest = r[i].alive && r[i].p < oldest ? r[i].p : oldest;
      oldest = wave_min_u32(oldest);
    }
    return pool.take(oldest);
  };
  uint32_t next = fetch(true);  // my chunk is [next, cend)
  uint32_t cend = pend - next > pool.size ? next + pool.size : pend;
  bool empty = next >= pend;
  // (positions are formed as "start + offset if offset < what is left, else the limit": a stream
  // may end within a chunk of 2^32 and a sum must not wrap into a position that looks valid)
#pragma unroll
  for (int i = 0; i < NP; i++) {
    const uint32_t off = lane + 64u * (uint32_t)i;
    match_run_start<true>(r[i], s, len, off < cend - next ? next + off : cend, cend, prev);
  }
  next = cend - next > 64u * NP ? next + 64u * NP : cend;
  for (;;) {
    bool alive = false;
    iters++;
    ZD_COUNT(1, 1); ZD_COUNT(2, 1);
#pragma unroll
    for (int i = 0; i < NP; i++) {
      ZD_COUNT(8, __builtin_popcountll(__builtin_amdgcn_ballot_w64(r[i].alive != 0)));
#ifdef ZD_MATCH_FIRST_SHARED
      const bool fin = match_run_step_to<true>(r[i], s, prev, (uint32_t)K, (uint32_t)Kq, sink);
      const unsigned long long fm = __builtin_amdgcn_ballot_w64(fin);
#else
      const unsigned long long fm = match_run_step_masks(r[i], s, prev, (uint32_t)K, (uint32_t)Kq, sink);
      const bool fin = __builtin_amdgcn_inverse_ballot_w64(fm);
#endif
      if (fm) {  // wave-uniform
        ZD_COUNT(6, 1); ZD_COUNT(7, __builtin_popcountll(fm));
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
        const uint32_t taken = (uint32_t)__builtin_popcountll(fm);
        const uint32_t rem = cend - next;
        uint32_t np = rank < rem ? next + rank : cend, lim = cend;
        if (taken > rem && !empty) {  // wave-uniform: the chunk runs out within this handout
          const uint32_t c = fetch(false);
          const uint32_t ce = pend - c > pool.size ? c + pool.size : pend;
          empty = c >= pend;
          if (rank >= rem) { np = rank - rem < ce - c ? c + (rank - rem) : ce; lim = ce; }
          next = ce - c > taken - rem ? c + (taken - rem) : ce;
          cend = ce;
        } else {
          next = rem > taken ? next + taken : cend;
        }
#ifdef ZD_MATCH_FIRST_SHARED
        if (fin) match_run_start<true>(r[i], s, len, np < lim ? np : lim, lim, prev);
#else
        if (fin) {  // match_run_start with the position's test made once (the shared form clamps np to lim and then compares again)
          MatchRun &n = r[i];
          const uint32_t park = lim ? lim - 1u : 0u;  // a finished run parks on a valid position
          n.alive = np < lim ? 1u : 0u;
          n.p = np < park ? np : park;
          n.q = n.p;
          n.best_len = MIN_MATCH_LEN - 1;
          n.best = 0; n.snap = SNAP_NONE; n.steps = 0;
          n.maxlen = len - n.p < (uint32_t)MAX_MATCH_LEN ? len - n.p : (uint32_t)MAX_MATCH_LEN;
          n.pw = load_u64_words(s, n.p);
          n.dn = prev[n.p];
        }
#endif
      }
      alive |= r[i].alive;
    }
    if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
  }
#ifdef ZD_MATCH_COUNTS
  mc.flush(lane);
#endif
  return iters;
}

// second form (scan_run_*): the faster one on long chains.  Returns the wave's rounds of cheap steps.
// Up to SCAN_ROUNDS rounds of cheap steps of a lane's two run slots in the LDS window (scan_run_step by hand: the compiler
// kept the runs' flags as integers in vector registers and turned them into masks and back every round; then the scalar
// instructions around the masks were the kernel's bound: 25 a round, now 12).  The runs' coordinates are LDS addresses of
// the window's bytes; cs: address of the window's links - 2 x address of its bytes.  W: the slots' walking runs (lane
// masks).  A round: each slot's three reads are issued under the slot's mask before either is waited for; then, slot by
// slot, the three tests that end a walk -- the bytes are the position's (a hit), the next candidate is out of range,
// K/4 or K candidates are done -- are v_cmpx, each of which takes its lanes out of exec, and what is left of exec is the
// slot's new mask.  WHICH test stopped a run is asked after the rounds, of the stopped runs only (H: those that stand on a
// hit; h, the two bytes read last, stays with the run).  The loop ends after SCAN_ROUNDS rounds or when fewer than SCAN_MIN_WALKERS lanes walk on.
// The link is read with ds_read_u16_d16: the d16 reads of this chip ZERO the half of the register they do not load
// (tools/probes/d16_loads.hip; SRAM ECC -- which is why the compiler never emits them) and take 7 ticks a wave at any even
// address, where ds_read_u16 takes 41 at 2 mod 4 (tools/probes/lds_costs.hip); the bytes with ds_read_u8, 4.7 each.
// (A slot without a walking lane issues its reads under an empty mask; they count in lgkmcnt like any other -- the waits
// below rely on it, the compiler's own code does, and tools/probes/exec0_lgkm.hip looked: profiles/r05_exec0_lgkm.txt.)
// Returns the rounds done.
struct ScanSlotMasks { unsigned long long W, H, F, L; };  // walking, hit, fin, live
#define ZD_SCAN_PROBE(i)                                       \
  "s_and_b64 exec, %[sv], %[W" #i "]\n\t"                      \
  "v_add_u32 %[a], %[t" #i "], %[bl" #i "]\n\t"                \
  "v_lshl_add_u32 %[la], %[t" #i "], 1, %[cs]\n\t"             \
  "ds_read_u8 %[h" #i "], %[a]\n\t"                            \
  "ds_read_u8 %[g" #i "], %[a] offset:1\n\t"                   \
  "ds_read_u16_d16 %[dn" #i "], %[la]\n\t"
#define ZD_SCAN_TAKE(i, BEHIND)                                \
  "s_and_b64 exec, %[sv], %[W" #i "]\n\t"                      \
  "v_add_u32 %[st" #i "], 1, %[st" #i "]\n\t"                  \
  "s_waitcnt lgkmcnt(" BEHIND ")\n\t"                          \
  "v_lshl_or_b32 %[h" #i "], %[g" #i "], 16, %[h" #i "]\n\t"   \
  "v_sub_u32 %[t" #i "], %[t" #i "], %[dn" #i "]\n\t"          \
  "v_cmpx_ne_u32 %[h" #i "], %[pb" #i "]\n\t"                  \
  "v_cmpx_ge_i32 %[t" #i "], %[lim" #i "]\n\t"                 \
  "v_cmpx_ne_u32 %[st" #i "], %[kl" #i "]\n\t"                 \
  "s_mov_b64 %[W" #i "], exec\n\t"
#define ZD_SCAN_HITS(i)                                        \
  "s_andn2_b64 exec, %[ws" #i "], %[W" #i "]\n\t"              \
  "v_cmp_eq_u32 vcc, %[h" #i "], %[pb" #i "]\n\t"              \
  "s_or_b64 %[H" #i "], %[H" #i "], vcc\n\t"
#define ZD_SCAN_CHECK(EXIT)                                    \
  "s_bcnt1_i32_b64 %[n], %[any]\n\t"                           \
  "s_cmp_lt_u32 %[n], %[minw]\n\t"                             \
  "s_cbranch_scc1 " EXIT "\n\t"
#define ZD_SCAN_ROUND2(EXIT) ZD_SCAN_PROBE(0) ZD_SCAN_PROBE(1) ZD_SCAN_TAKE(0, "3") ZD_SCAN_TAKE(1, "0") \
  "s_or_b64 %[any], %[W0], %[W1]\n\t" ZD_SCAN_CHECK(EXIT)
#define ZD_SCAN_ROUND3(EXIT) ZD_SCAN_PROBE(0) ZD_SCAN_PROBE(1) ZD_SCAN_PROBE(2) ZD_SCAN_TAKE(0, "6") ZD_SCAN_TAKE(1, "3") ZD_SCAN_TAKE(2, "0") \
  "s_or_b64 %[any], %[W0], %[W1]\n\ts_or_b64 %[any], %[any], %[W2]\n\t" ZD_SCAN_CHECK(EXIT)
#define ZD_SCAN_ROUND4(EXIT) ZD_SCAN_PROBE(0) ZD_SCAN_PROBE(1) ZD_SCAN_PROBE(2) ZD_SCAN_PROBE(3) \
  ZD_SCAN_TAKE(0, "9") ZD_SCAN_TAKE(1, "6") ZD_SCAN_TAKE(2, "3") ZD_SCAN_TAKE(3, "0") \
  "s_or_b64 %[any], %[W0], %[W1]\n\ts_or_b64 %[any], %[any], %[W2]\n\ts_or_b64 %[any], %[any], %[W3]\n\t" ZD_SCAN_CHECK(EXIT)
#define ZD_SCAN_EXIT_ASM(L, N) L ":\n\ts_mov_b32 %[n], " N "\n\ts_branch 99f\n"
#define ZD_STR_(x) #x
#define ZD_STR(x) ZD_STR_(x)
// (SCAN_ROUNDS rounds at most, ZD_SCAN_ROUNDS_DENSE -- the exit behind the fourth -- when dense is set)
#define ZD_SCAN_BODY(ROUND, SAVE, HITS)                                                                                 \
  "s_mov_b64 %[sv], exec\n\t" SAVE                                                                                      \
  ROUND("11f") ROUND("12f") ROUND("13f") ROUND("14f")                                                                   \
  "s_cmp_lg_u32 %[dense], 0\n\t"                                                                                        \
  "s_cbranch_scc1 14f\n\t"                                                                                              \
  ROUND("15f") ROUND("16f") ROUND("17f") ROUND("18f")                                                                   \
  "s_mov_b32 %[n], 8\n\t"                                                                                               \
  "s_branch 99f\n"                                                                                                      \
  ZD_SCAN_EXIT_ASM("11", "1") ZD_SCAN_EXIT_ASM("12", "2") ZD_SCAN_EXIT_ASM("13", "3") ZD_SCAN_EXIT_ASM("14", "4")       \
  ZD_SCAN_EXIT_ASM("15", "5") ZD_SCAN_EXIT_ASM("16", "6") ZD_SCAN_EXIT_ASM("17", "7") ZD_SCAN_EXIT_ASM("18", "8")       \
  "99:\n\t" /* of the runs that stopped, those whose last candidate passed the test */                                 \
  HITS "s_mov_b64 exec, %[sv]"
#define ZD_SCAN_OUT(i) [g##i] "=&v"(g[i]), [ws##i] "=&s"(ws[i]), [H##i] "+s"(m[i].H), [h##i] "+v"(h[i]), [W##i] "+s"(m[i].W), \
                       [t##i] "+v"(r[i].t), [st##i] "+v"(r[i].steps), [dn##i] "+v"(r[i].dn)
#define ZD_SCAN_IN(i) [bl##i] "v"(r[i].blm1), [pb##i] "v"(r[i].pb), [lim##i] "v"(r[i].lim), [kl##i] "v"(r[i].klim)
#define ZD_SCAN_SHARED_OUT [sv] "=&s"(sv), [any] "=&s"(any), [n] "=&s"(n), [a] "=&v"(a), [la] "=&v"(la)
#define ZD_SCAN_SHARED_IN [cs] "s"(cs), [minw] "s"((uint32_t)SCAN_MIN_WALKERS), [dense] "s"(dense)
template <int NP>
__device__ __forceinline__ uint32_t scan_rounds_lds(ScanRun (&r)[NP], ScanSlotMasks (&m)[NP], uint32_t (&h)[NP], uint32_t cs, uint32_t dense) {
  static_assert(SCAN_ROUNDS == 8 && ZD_SCAN_ROUNDS_DENSE == 4, "the loop below is unrolled by hand");
  static_assert(NP >= 2 && NP <= 4, "written out for two to four run slots");
  unsigned long long sv, any, ws[NP];
  uint32_t a, la, g[NP], n;
  if constexpr (NP == 2) {
    asm volatile(ZD_SCAN_BODY(ZD_SCAN_ROUND2, "s_mov_b64 %[ws0], %[W0]\n\ts_mov_b64 %[ws1], %[W1]\n\t", ZD_SCAN_HITS(0) ZD_SCAN_HITS(1))
                 : ZD_SCAN_SHARED_OUT, ZD_SCAN_OUT(0), ZD_SCAN_OUT(1)
                 : ZD_SCAN_IN(0), ZD_SCAN_IN(1), ZD_SCAN_SHARED_IN
                 : "vcc", "scc", "memory");
  } else if constexpr (NP == 3) {
    asm volatile(ZD_SCAN_BODY(ZD_SCAN_ROUND3, "s_mov_b64 %[ws0], %[W0]\n\ts_mov_b64 %[ws1], %[W1]\n\ts_mov_b64 %[ws2], %[W2]\n\t",
                              ZD_SCAN_HITS(0) ZD_SCAN_HITS(1) ZD_SCAN_HITS(2))
                 : ZD_SCAN_SHARED_OUT, ZD_SCAN_OUT(0), ZD_SCAN_OUT(1), ZD_SCAN_OUT(2)
                 : ZD_SCAN_IN(0), ZD_SCAN_IN(1), ZD_SCAN_IN(2), ZD_SCAN_SHARED_IN
                 : "vcc", "scc", "memory");
  } else {
    asm volatile(ZD_SCAN_BODY(ZD_SCAN_ROUND4,
                              "s_mov_b64 %[ws0], %[W0]\n\ts_mov_b64 %[ws1], %[W1]\n\ts_mov_b64 %[ws2], %[W2]\n\ts_mov_b64 %[ws3], %[W3]\n\t",
                              ZD_SCAN_HITS(0) ZD_SCAN_HITS(1) ZD_SCAN_HITS(2) ZD_SCAN_HITS(3))
                 : ZD_SCAN_SHARED_OUT, ZD_SCAN_OUT(0), ZD_SCAN_OUT(1), ZD_SCAN_OUT(2), ZD_SCAN_OUT(3)
                 : ZD_SCAN_IN(0), ZD_SCAN_IN(1), ZD_SCAN_IN(2), ZD_SCAN_IN(3), ZD_SCAN_SHARED_IN
                 : "vcc", "scc", "memory");
  }
  return n;
}
#undef ZD_SCAN_PROBE
#undef ZD_SCAN_TAKE
#undef ZD_SCAN_HITS
#undef ZD_SCAN_CHECK
#undef ZD_SCAN_ROUND2
#undef ZD_SCAN_ROUND3
#undef ZD_SCAN_ROUND4
#undef ZD_SCAN_EXIT_ASM
#undef ZD_SCAN_BODY
#undef ZD_SCAN_OUT
#undef ZD_SCAN_IN
#undef ZD_SCAN_SHARED_OUT
#undef ZD_SCAN_SHARED_IN
#undef ZD_STR
#undef ZD_STR_

// s, prev: the window's bytes and links indexed by coordinate (for the LDS window: s + c is the byte at LDS address c),
// cs: see scan_rounds_lds.
template <int NP, typename Sink, typename S, typename P, typename Pool>
__device__ __forceinline__ uint32_t lz_match_scan_pool(S s, uint32_t len, Pool &pool, uint32_t pend, uint32_t lane,
                                                        P prev, uint32_t cs, int K, int Kq, Sink sink) {
  static_assert(64u * NP <= POOL_CHUNK, "chunk");
  ScanRun r[NP];
  ScanSlotMasks m[NP];
  uint32_t h[NP];      // the two bytes a run's last step read (scan_rounds_lds)
  uint32_t hits = 0;   // runs in the last compare
  uint32_t iters = 0;  // rounds of cheap steps (what is returned: the same measure as the first form's iterations)
#ifdef ZD_MATCH_COUNTS
  MatchCounts mc;
#endif
  auto ballot = [](bool b) { return (unsigned long long)__builtin_amdgcn_ballot_w64(b); };
  auto mine = [](unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); };  // my lane's bit of a mask
  auto any_of = [&](auto pick) { unsigned long long o = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) o |= pick(m[i]);
    return o; };
  auto fetch = [&](bool first) -> uint32_t {
    uint32_t oldest = 0xFFFFFFFFu;
    if (Pool::WANTS_OLDEST && !first) {
#pragma unroll
      for (int i = 0; i < NP; i++) oldest = mine(m[i].L) && r[i].p < oldest ? r[i].p : oldest;
      oldest = wave_min_u32(oldest);
    }
    return pool.take(oldest);
  };
  uint32_t next = fetch(true);  // my chunk is [next, cend)
  uint32_t cend = pend - next > pool.size ? next + pool.size : pend;
  bool empty = next >= pend;
  // (positions are formed as "start + offset if offset < what is left, else the limit": a stream
  // may end within a chunk of 2^32 and a sum must not wrap into a position that looks valid)
#pragma unroll
  for (int i = 0; i < NP; i++) {
    const uint32_t off = lane + 64u * (uint32_t)i;
    const ScanFlags f = scan_run_start<true>(r[i], s, len, off < cend - next ? next + off : cend, cend, prev, (uint32_t)K, (uint32_t)Kq);
    m[i].W = ballot(f.walking); m[i].H = 0; m[i].F = ballot(f.fin); m[i].L = ballot(f.live);
    h[i] = 0;
  }
  next = cend - next > 64u * NP ? next + 64u * NP : cend;
  for (;;) {
    ZD_COUNT(1, 1);
    // cheap steps: every walking run goes from candidate to candidate until one passes the byte test -- or its walk ends
    unsigned long long full[NP];  // runs whose compare found a match as long as a match can be
#pragma unroll
    for (int i = 0; i < NP; i++) full[i] = 0;
    if (any_of([](const ScanSlotMasks &x) { return x.W; })) {
      const uint32_t n = scan_rounds_lds<NP>(r, m, h, cs, (uint32_t)__builtin_amdgcn_readfirstlane((int)(hits >= SCAN_DENSE_HITS ? 1u : 0u)));
      iters += n;
      ZD_COUNT(2, n);
    }
    // the compares of the runs that stand on a candidate which passed, together: a lane's first such slot.  Straight-line for the
    // candidate that differs within its first 8 bytes (on text: nearly all of them); the others behind one branch.
    // (a lane with hits in two slots takes the second next time: a second turn of this code at once, for 1, 8 or 16 such
    // lanes and more, measured 1-4 % slower on text, 3-bit symbols and the corpus)
    const unsigned long long hall = any_of([](const ScanSlotMasks &x) { return x.H; });
    if (hall) {
      unsigned long long hm[NP], seen = 0;  // the runs compared now
      hits = 0;
#pragma unroll
      for (int i = 0; i < NP; i++) { hits += (uint32_t)__builtin_popcountll(m[i].H); hm[i] = m[i].H & ~seen; seen |= m[i].H; }
      ZD_COUNT(4, 1); ZD_COUNT(5, hits);
      // (every lane goes through it -- a run's q = t + dn and p are positions of the window whatever the run's state -- and
      // the runs compared take the results: masks stay masks, where a value set under a lane's condition went through a
      // vector register and back)
      uint32_t p = r[NP - 1].p, q = r[NP - 1].t + r[NP - 1].dn, bl = r[NP - 1].blm1, ml = r[NP - 1].maxlen;
      uint64_t pw = r[NP - 1].pw;
#pragma unroll
      for (int i = NP - 2; i >= 0; i--) {
        const bool w = mine(hm[i]);
        p = w ? r[i].p : p; q = w ? r[i].t + r[i].dn : q; bl = w ? r[i].blm1 : bl; ml = w ? r[i].maxlen : ml; pw = w ? r[i].pw : pw;
      }
      bl += 1u;
      const uint64_t x = load_u64_words(s, q) ^ pw;
      uint32_t l = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
      l = l < ml ? l : ml;  // (the last 7 positions of a stream: what lies behind its end does not count)
      const unsigned long long more = ballot(x == 0 && ml > 8u) & hall;  // the first 8 bytes agree and there are others
      if (more) {
        if (mine(more)) {
          // before the long compare, the 8 bytes that END at best_len (they include the bytes already tested); what fails
          // here is at most best_len long
          bool same = true;
          if (bl >= 8u) same = load_u64_words(s, q + bl - 7u) == load_u64_words(s, p + bl - 7u);
          if (same) l = common_prefix_t<true>(s, q, p, ml, 8u);
        }
      }
      const bool better = l > bl;
      const uint32_t nb = better ? l - 1u : bl - 1u;
      const uint32_t npb = (uint32_t)s[p + nb] | ((uint32_t)s[p + nb + 1u] << 16);  // (of a run that is done: never looked at)
      const uint32_t nbest = ((p - q) << 9) | l;
      const unsigned long long fm = ballot(l == ml);  // zd.ml:1194: after l == maxlen nothing later can be longer
      const unsigned long long bm = ballot(better);
#pragma unroll
      for (int i = 0; i < NP; i++) {
        const bool w = mine(hm[i]);
        r[i].blm1 = w ? nb : r[i].blm1; r[i].pb = w ? npb : r[i].pb; r[i].best = mine(hm[i] & bm) ? nbest : r[i].best;
        full[i] = fm & hm[i];
        m[i].H &= ~hm[i];
      }
    }
    // stopped runs: done, or on with the walk
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const unsigned long long st = m[i].L & ~(m[i].W | m[i].H | m[i].F);
      if (st) {  // (scan_run_settle on masks)
        const unsigned long long at = ballot(r[i].steps == r[i].klim) & st, last = ballot(r[i].klim == (uint32_t)K);
        const bool note = mine(at & ~last);  // the first K/4 candidates' answer; on with K
        r[i].snap = note ? r[i].best : r[i].snap;
        r[i].klim = note ? (uint32_t)K : r[i].klim;
        const unsigned long long fm = ((at & last) | full[i] | ballot(scan_far<P>(r[i]))) & st;
        m[i].F |= fm;
        m[i].W |= st & ~fm;
      }
    }
    const bool busy = any_of([](const ScanSlotMasks &x) { return x.W | x.H; }) != 0;
    // finished positions are stored and their slots take the pool's next positions
    {  // one handout for the lane's slots: a lane's first finished slot (another one of the same lane waits for the next)
      unsigned long long fk[NP], fu = 0;
#pragma unroll
      for (int i = 0; i < NP; i++) { fk[i] = m[i].F & ~fu; fu |= m[i].F; }
      const uint32_t taken = (uint32_t)__builtin_popcountll(fu);
      if (taken >= SCAN_HANDOUT || (fu && !busy)) {  // wave-uniform
        ZD_COUNT(6, 1); ZD_COUNT(7, taken);
        const bool fin = mine(fu);
        uint32_t fp = r[NP - 1].p, fbest = r[NP - 1].best, fsnap = r[NP - 1].snap;
#pragma unroll
        for (int i = NP - 2; i >= 0; i--) { const bool w = mine(fk[i]); fp = w ? r[i].p : fp; fbest = w ? r[i].best : fbest; fsnap = w ? r[i].snap : fsnap; }
        if (fin) sink(fp, fbest, Kq == 0 ? 0u : (fsnap != SNAP_NONE ? fsnap : fbest));
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fu, 0u));
        const uint32_t rem = cend - next;
        uint32_t np = rank < rem ? next + rank : cend, lim = cend;
        if (taken > rem && !empty) {  // wave-uniform: the chunk runs out within this handout
          const uint32_t c = fetch(false);
          const uint32_t ce = pend - c > pool.size ? c + pool.size : pend;
          empty = c >= pend;
          if (rank >= rem) { np = rank - rem < ce - c ? c + (rank - rem) : ce; lim = ce; }
          next = ce - c > taken - rem ? c + (taken - rem) : ce;
          cend = ce;
        } else {
          next = rem > taken ? next + taken : cend;
        }
        ScanRun nr;
        scan_run_start<true>(nr, s, len, np < lim ? np : lim, lim, prev, (uint32_t)K, (uint32_t)Kq);
        const unsigned long long alive = ballot(np < lim) & fu;
        const unsigned long long walk = K != 0 ? ballot(nr.blm1 + 1u < nr.maxlen && !scan_far<P>(nr)) & alive : 0;
#pragma unroll
        for (int i = 0; i < NP; i++) {
          if (mine(fk[i])) r[i] = nr;
          m[i].W |= walk & fk[i];
          m[i].F = (m[i].F & ~fk[i]) | (alive & ~walk & fk[i]);
          m[i].L = (m[i].L & ~fk[i]) | (alive & fk[i]);
        }
      }
    }
    if (any_of([](const ScanSlotMasks &x) { return x.L; }) == 0) break;
  }
#ifdef ZD_MATCH_COUNTS
  mc.flush(lane);
#endif
  return iters;
}
#endif

// ---------------------------------------------------------------------------
// The lazy parse (Lz77.compress zd.ml:1203-1244) in three parallel-friendly
// pieces.  Between two positions where the reference has NO pending match the
// parse is a pure function of the position it starts from:
//   * no match at p: literal s[p], continue at p+1;
//   * a match at p: it is deferred; while the next position offers a strictly
//     longer one (found with K candidates, or K/4 once the pending length reaches
//     good_match, zd.ml:1182-1185) a literal is emitted and the longer match
//     becomes pending; then the pending match is emitted and the parse continues
//     right after it.
// lz_macro_position computes that "macro step" for EVERY position (one lane
// each); lz_parse_kernel follows the steps from position 0 tile by tile, marking
// the visited positions, counting symbols and cutting blocks (write_block_symbol
// zd.ml:1118-1123), and writes the symbols of every visited position
// (lz_emit_position), all positions of a tile in parallel.  (tests/host_sim has the
// plain serial walk over the same steps as a second opinion.)

struct MacroStep {
  uint32_t bref;  // the match that ends the step (dist << 9 | len), 0: a lone literal
  uint32_t step;  // advance (bits 0-15) | literals emitted before the match (bits 16-31)
};
ZD_HD uint32_t macro_advance(uint32_t step) { return step & 0xFFFFu; }
ZD_HD uint32_t macro_lits(uint32_t step) { return step >> 16; }

// get(j) returns lz_match_position(j): best-of-K in the low word, best-of-K/4 in
// the high word.
template <typename GetMatch>
ZD_HD MacroStep lz_macro_position(uint32_t p, uint32_t len, int good_match, GetMatch get) {
  MacroStep r;
  r.bref = 0;
  r.step = 1u | (1u << 16);  // one literal
  if (len < (uint32_t)MIN_MATCH_LEN || p > len - MIN_MATCH_LEN) return r;  // i > max_pos: literals
  const uint32_t max_pos = len - MIN_MATCH_LEN;
  uint32_t pend = (uint32_t)get(p);  // no pending match: threshold 3 < good_match, K candidates
  if ((pend & 0x1FF) == 0) return r;
  uint32_t n_lits = 0;
  uint32_t j = p + 1;
  while (j <= max_pos) {
    const uint32_t pl = pend & 0x1FF;
    const uint32_t rem = len - j;
    const uint32_t maxlen = rem < (uint32_t)MAX_MATCH_LEN ? rem : (uint32_t)MAX_MATCH_LEN;
    uint32_t b = 0;
    if (pl < maxlen) {
      const uint64_t m = get(j);
      const uint32_t c = pl >= (uint32_t)good_match ? (uint32_t)(m >> 32) : (uint32_t)m;
      if ((c & 0x1FF) > pl) b = c;
    }
    if (b == 0) break;  // previous match wins (zd.ml:1224)
    n_lits++;           // literal s[j-1], defer the longer match (zd.ml:1236-1240)
    pend = b;
    j++;
  }
  r.bref = pend;
  r.step = (n_lits + (pend & 0x1FF)) | (n_lits << 16);
  return r;
}

// Symbols of one visited position written at syms[first..]: a literal position
// writes its byte; a match position its deferral literals, then its match.
ZD_HD uint32_t macro_sym_count(MacroStep m) { return m.bref ? macro_lits(m.step) + 1u : 1u; }
ZD_HD void lz_emit_position(const uint8_t *s, uint32_t p, MacroStep m, uint32_t *syms, uint32_t first) {
  if (m.bref == 0) { syms[first] = s[p]; return; }
  const uint32_t lits = macro_lits(m.step);
  for (uint32_t k = 0; k < lits; k++) syms[first + k] = s[p + k];
  syms[first + lits] = m.bref;
}

// ---------------------------------------------------------------------------
// Huffman encoder construction (zd.ml:393-528) on u32 arrays (LDS on device).

// heapdown zd.ml:408-417
ZD_HD void huff_heapdown(uint32_t *h, int max, int i) {
  for (;;) {
    int l = 2 * i, r = l + 1;
    if (l > max) return;
    int k = (r > max) ? l : (h[l] < h[r] ? l : r);
    if (h[i] > h[k]) { uint32_t v = h[i]; h[i] = h[k]; h[k] = v; i = k; }
    else return;
  }
}

// Huffman.lengths_of_freqs zd.ml:404-473.  Node = (freq << 10) | link; with
// leaf freqs capped at 65535 and at most 286 leaves the packed value stays
// below 2^32.  heap: 577 words.  Writes plain code lengths to e[0..max_sym].
ZD_HD void huff_lengths_of_freqs(uint32_t *heap, uint32_t *e, const uint32_t *freqs, int max_sym,
                                 int max_code_len) {
  uint32_t freq_cap = 65535;
  for (;;) {
    int max = 0;
    for (int sym = 0; sym <= max_sym; sym++) {
      uint32_t f = freqs[sym];
      if (f == 0) continue;
      if (f > freq_cap) f = freq_cap;
      max++;
      heap[max] = (f << 10) | (uint32_t)(max_sym + 1 + max);
    }
    for (int i = max / 2; i >= 1; i--) huff_heapdown(heap, max, i);
    if (max < 2) {  // trivial_codeword_lengths zd.ml:462-466
      for (int sym = 0; sym <= max_sym; sym++) e[sym] = freqs[sym] == 0 ? 0u : 1u;
      return;
    }
    for (int m = max; m > 1; m--) {  // make_huffman_tree zd.ml:432-445
      const int new_max = m - 1;
      const uint32_t p = heap[1];
      heap[1] = heap[m];
      huff_heapdown(heap, new_max, 1);
      const uint32_t q = heap[1];
      const uint32_t f = (p >> 10) + (q >> 10);
      heap[1] = (f << 10) | (uint32_t)m;
      heap[p & 0x3FF] = (uint32_t)m;
      heap[q & 0x3FF] = (uint32_t)m;
      huff_heapdown(heap, new_max, 1);
    }
    bool overflow = false;  // code_lengths_of_tree zd.ml:446-461
    int rank = 0;
    for (int sym = 0; sym <= max_sym; sym++) {
      if (freqs[sym] == 0) { e[sym] = 0; continue; }
      rank++;
      uint32_t p = heap[max_sym + 1 + rank];
      int l = 1;
      while (p != 2) { l++; p = heap[p]; }
      if (l > max_code_len) { overflow = true; break; }
      e[sym] = (uint32_t)l;
    }
    if (!overflow) return;
    freq_cap >>= 1;  // flatten and retry zd.ml:470-473
  }
}

// The tree of Huffman.lengths_of_freqs (make_huffman_tree zd.ml:432-445) without the heap.
// The reference pops the two smallest nodes of a binary heap whose keys (freq << 10) | link are
// all distinct, so WHICH nodes meet is a function of the key order alone:
//   * a leaf's link is max_sym + 1 + rank (rank: 1-based among the used symbols, in symbol
//     order), a merged node's link is the heap size m at its creation (n, n-1, .., 2): at equal
//     frequency merged nodes come before leaves, later merged nodes before earlier ones, lower
//     symbols before higher ones;
//   * the frequencies of merged nodes never decrease in creation order (both popped nodes are
//     at least as large as the larger one popped before, all frequencies are >= 1), and a node
//     created while a group of equal merged nodes is being popped is strictly larger than them.
// So the leaves sorted by (freq, rank) form one queue that is read front to back, and the
// merged nodes, in creation order, a second one whose front GROUP of equal frequencies is
// popped last-created first: n - 1 steps of O(1) instead of 2 (n - 1) sift-downs of depth
// log n, each level a dependent LDS round trip on the device.  Same tree, same lengths
// (tests/test_host_sim.py compares the two on ties of every kind and on the flatten-and-retry path).
//   q[0, n): in: the leaf keys (freq << 10) | rank, ascending (n >= 2, rank 1..n); the merged nodes'
//            frequencies overwrite them in creation order (node c, link n - c, at q[c]): when node c
//            is created at least c + 2 leaves have been taken -- 2 (c + 1) pops, at most c of them
//            merged nodes -- so the slot is free.
//   par[0, 2n + 2): out, par[m] = link of the parent of the merged node with link m (2 < m <= n),
//                   par[n + rank] = link of the parent of leaf `rank`.  The root has link 2.
template <typename Par>
ZD_HD void huff_tree_two_queues(uint32_t *q, int n, Par *par) {
  int li = 0;                   // next leaf
  int cnt = 0;                  // merged nodes created
  int lo = 0, top = -1, hi = -1;  // front group of the merged queue: nodes [lo, hi], not yet popped [lo, top]
  for (int t = 0; t + 1 < n; t++) {
    const uint32_t m = (uint32_t)(n - t);
    uint32_t fsum = 0;
    for (int pop = 0; pop < 2; pop++) {
      if (top < lo && hi + 1 < cnt) {  // the group is used up: the next one starts behind it
        lo = hi + 1;
        hi = lo;
        while (hi + 1 < cnt && q[hi + 1] == q[lo]) hi++;
        top = hi;
      }
      const bool have_node = top >= lo, have_leaf = li < n;
      const uint32_t lkey = have_leaf ? q[li] : 0u;
      const uint32_t lf = lkey >> 10;
      if (have_node && (!have_leaf || q[top] <= lf)) {  // equal frequencies: the merged node's link is smaller
        fsum += q[top];
        par[n - top] = (Par)m;
        top--;
      } else {
        fsum += lf;
        par[n + (lkey & 0x3FF)] = (Par)m;
        li++;
      }
    }
    q[cnt] = fsum;
    if (top >= lo && top == hi && hi == cnt - 1 && q[lo] == fsum) { hi = cnt; top = cnt; }  // joins the untouched front group
    cnt++;
  }
}

// Huffman.lengths_of_freqs zd.ml:404-473 on the two queues (the device form: wave_lengths_of_freqs
// in deflate.hip sorts and climbs with all lanes and runs the merge on one).  scratch: 2 n + 2 words.
ZD_HD void huff_lengths_of_freqs_tq(uint32_t *scratch, uint32_t *e, const uint32_t *freqs, int max_sym,
                                    int max_code_len) {
  uint32_t freq_cap = 65535;
  for (;;) {
    int n = 0;
    uint32_t *lk = scratch;
    for (int sym = 0; sym <= max_sym; sym++) {
      uint32_t f = freqs[sym];
      if (f == 0) continue;
      if (f > freq_cap) f = freq_cap;
      n++;
      lk[n - 1] = (f << 10) | (uint32_t)n;
    }
    if (n < 2) {  // trivial_codeword_lengths zd.ml:462-466
      for (int sym = 0; sym <= max_sym; sym++) e[sym] = freqs[sym] == 0 ? 0u : 1u;
      return;
    }
    for (int i = 1; i < n; i++) {  // insertion sort: keys are distinct
      const uint32_t v = lk[i];
      int j = i;
      while (j > 0 && lk[j - 1] > v) { lk[j] = lk[j - 1]; j--; }
      lk[j] = v;
    }
    uint16_t *par = (uint16_t *)(scratch + n);
    huff_tree_two_queues(lk, n, par);
    bool overflow = false;
    int rank = 0;
    for (int sym = 0; sym <= max_sym; sym++) {
      if (freqs[sym] == 0) { e[sym] = 0; continue; }
      rank++;
      uint32_t p = par[n + rank];
      int l = 1;
      while (p != 2) { l++; p = par[p]; }
      if (l > max_code_len) { overflow = true; break; }
      e[sym] = (uint32_t)l;
    }
    if (!overflow) return;
    freq_cap >>= 1;  // flatten and retry zd.ml:470-473
  }
}

// Huffman.init_with_lengths zd.ml:477-506: canonical codes, stored bit-reversed,
// packed (code << 5) | len
ZD_HD void huff_init_with_lengths(uint32_t *e, int max_sym) {
  uint32_t count[16], code[16];
  for (int i = 0; i < 16; i++) { count[i] = 0; code[i] = 0; }
  for (int sym = 0; sym <= max_sym; sym++) count[e[sym] & 0x1F]++;
  count[0] = 0;
  for (int l = 1; l <= 15; l++) code[l] = (code[l - 1] + count[l - 1]) << 1;
  for (int sym = 0; sym <= max_sym; sym++) {
    const uint32_t l = e[sym] & 0x1F;
    if (l != 0) {
      const uint32_t c = code[l];
      e[sym] = (bitrev(c, (int)l) << 5) | l;
      code[l] = c + 1;
    }
  }
}

// fixed_litlen_encoder / fixed_dist_encoder zd.ml:514-527
ZD_HD void huff_fixed_encoders(uint32_t *lit /*288*/, uint32_t *dist /*32*/) {
  for (int i = 0; i <= 143; i++) lit[i] = 8;
  for (int i = 144; i <= 255; i++) lit[i] = 9;
  for (int i = 256; i <= 279; i++) lit[i] = 7;
  for (int i = 280; i <= 287; i++) lit[i] = 8;
  huff_init_with_lengths(lit, 287);
  for (int i = 0; i <= 31; i++) dist[i] = 5;
  huff_init_with_lengths(dist, 31);
}

// Per-stream block coder state: what the reference keeps in its encoder record
// across blocks (zd.ml:778-815).  All arrays live in LDS on the device.
struct BlockCoder {
  uint32_t *lit_freq;      // 286 (+2 pad)   litlen_sym_freqs
  uint32_t *dist_freq;     // 30 (+2 pad)    dist_sym_freqs
  uint32_t *codelen_freq;  // 19             codelen_sym_freqs: NEVER reset (Q1, zd.ml:849-854)
  uint32_t *dyn_lit;       // 288            dyn_litlen
  uint32_t *dyn_dist;      // 32             dyn_dist
  uint32_t *dyn_codelen;   // 32 (19 used)   dyn_codelen
  uint32_t *fix_lit;       // 288
  uint32_t *fix_dist;      // 32
  uint32_t *codelen_syms;  // 316 + 4        codelen_syms (sym | repeat_bits << 8)
  uint32_t *heap;          // 577
  int codelen_syms_len, hlit, hdist, hclen;
};

// make_dynamic_huffman zd.ml:953-957 + make_dynamic_huffman_encoding zd.ml:959-1043
// (in two halves, like the kernels' wave_make_dynamic_syms / wave_make_dynamic_codelen: the second one depends on
// the blocks before through c.codelen_freq)
ZD_HD void coder_make_dynamic_syms(BlockCoder &c) {
  huff_lengths_of_freqs(c.heap, c.dyn_lit, c.lit_freq, LITLEN_SYM_MAX, 15);
  huff_init_with_lengths(c.dyn_lit, LITLEN_SYM_MAX);
  huff_lengths_of_freqs(c.heap, c.dyn_dist, c.dist_freq, DIST_SYM_MAX, 15);
  huff_init_with_lengths(c.dyn_dist, DIST_SYM_MAX);
  // gather_dynamic_huffman_code_lengths zd.ml:963-988
  int litlen_count = LITLEN_SYM_MAX;
  while (litlen_count >= 0 && (c.dyn_lit[litlen_count] & 0x1F) == 0) litlen_count--;
  litlen_count++;
  int dist_count = DIST_SYM_MAX;
  while (dist_count >= 0 && (c.dyn_dist[dist_count] & 0x1F) == 0) dist_count--;
  dist_count++;
  if (dist_count == 0) {  // HDIST 0 means 1: symbol 0 gets length 1, code 0 (zd.ml:974-979)
    c.dyn_dist[0] = 1;
    dist_count = 1;
  }
  c.hlit = litlen_count - 257;
  c.hdist = dist_count - 1;
  uint32_t *l = c.codelen_syms;
  for (int i = 0; i < litlen_count; i++) l[i] = c.dyn_lit[i] & 0x1F;
  for (int i = 0; i < dist_count; i++) l[litlen_count + i] = c.dyn_dist[i] & 0x1F;
  // compute_codelen_syms zd.ml:989-1030 (in place: the encoding never expands)
  const int len_max = litlen_count + dist_count - 1;
  int k = 0, i = 0;
  while (i <= len_max) {
    if (l[i] == 0) {
      const int mx = len_max < i + 138 - 1 ? len_max : i + 138 - 1;
      int j = i + 1;
      while (j <= mx && l[j] == 0) j++;
      const int zcount = j - i;
      if (zcount < 3) { l[k] = 0; c.codelen_freq[0]++; i = i + 1; }
      else if (zcount <= 10) { l[k] = ((uint32_t)(zcount - 3) << 8) | 17; c.codelen_freq[17]++; i = j; }
      else { l[k] = ((uint32_t)(zcount - 11) << 8) | 18; c.codelen_freq[18]++; i = j; }
      k++;
    } else {
      const uint32_t sym = l[i];
      const int mx = len_max < i + 6 ? len_max : i + 6;
      int j = i + 1;
      while (j <= mx && l[j] == sym) j++;
      const int scount = j - i;
      l[k] = sym;
      c.codelen_freq[sym]++;
      if (scount <= 3) { k++; i = i + 1; }
      else {
        l[k + 1] = ((uint32_t)(scount - 3 - 1) << 8) | 16;
        c.codelen_freq[16]++;
        k += 2;
        i = j;
      }
    }
  }
  c.codelen_syms_len = k;
}
ZD_HD void coder_make_dynamic_codelen(BlockCoder &c) {
  huff_lengths_of_freqs(c.heap, c.dyn_codelen, c.codelen_freq, CODELEN_SYM_MAX, 7);
  huff_init_with_lengths(c.dyn_codelen, CODELEN_SYM_MAX);
  int o = CODELEN_SYM_MAX;  // codelen_length_count zd.ml:1032-1036
  while (o > 0 && (c.dyn_codelen[k_codelen_order[o]] & 0x1F) == 0) o--;
  c.hclen = (o + 1) - 4;
}
ZD_HD void coder_make_dynamic(BlockCoder &c) {
  coder_make_dynamic_syms(c);
  coder_make_dynamic_codelen(c);
}

ZD_HD int length_extra_bits(int sym) {
  if (sym < LITLEN_FIRST_LEN) return 0;
  uint32_t b, e;
  length_sym_value(sym, b, e);
  return (int)e;
}
ZD_HD int dist_extra_bits(int sym) {
  uint32_t b, e;
  dist_sym_value(sym, b, e);
  return (int)e;
}

// bit_length_of_block_symbols zd.ml:1049-1064
ZD_HD uint64_t coder_symbols_bits(const BlockCoder &c, const uint32_t *hlit, const uint32_t *hdist) {
  uint64_t acc = 0;
  for (int sym = 0; sym <= LITLEN_SYM_MAX; sym++)
    acc += (uint64_t)c.lit_freq[sym] * ((hlit[sym] & 0x1F) + (uint32_t)length_extra_bits(sym));
  for (int sym = 0; sym <= DIST_SYM_MAX; sym++)
    acc += (uint64_t)c.dist_freq[sym] * ((hdist[sym] & 0x1F) + (uint32_t)dist_extra_bits(sym));
  return acc;
}

// write_block's three estimates and its choice (zd.ml:1045-1047,1066-1079,1099-1104).
// pending_bits = dst_bits_len.  Returns 0 stored / 1 fixed / 2 dynamic.
ZD_HD int coder_choose(const BlockCoder &c, uint32_t block_src_len, int pending_bits, uint64_t &flen,
                       uint64_t &dlen) {
  const uint64_t nlen = 3 + (uint64_t)(8 - ((pending_bits + 3) % 8)) + (4 + (uint64_t)block_src_len) * 8;
  flen = 3 + coder_symbols_bits(c, c.fix_lit, c.fix_dist);
  uint64_t acc = 3 + 5 + 5 + 4 + 3 * (uint64_t)(c.hclen + 4);
  for (int sym = 0; sym <= CODELEN_SYM_MAX; sym++) {
    const uint32_t rb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
    acc += (uint64_t)c.codelen_freq[sym] * ((c.dyn_codelen[sym] & 0x1F) + rb);
  }
  dlen = acc + coder_symbols_bits(c, c.dyn_lit, c.dyn_dist);
  if (nlen <= dlen && nlen <= flen) return 0;
  if (flen <= dlen) return 1;
  return 2;
}

// One block symbol as bits (write_block_symbols zd.ml:879-910): value holds the
// code, then the extra bits, LSB first; at most 48 bits.
ZD_HD void symbol_bits(uint32_t bref, const uint32_t *hlit, const uint32_t *hdist, uint64_t &value,
                       int &nbits) {
  const uint32_t dist = bref >> 9, len = bref & 0x1FF;
  if (dist == 0) {
    const uint32_t si = hlit[len];
    value = si >> 5;
    nbits = (int)(si & 0x1F);
    return;
  }
  const int lsym = length_to_sym((int)len);
  uint32_t si = hlit[lsym];
  int count = (int)(si & 0x1F);
  uint32_t vbase, vextra;
  length_sym_value(lsym, vbase, vextra);
  uint64_t v = (uint64_t)(si >> 5) | ((uint64_t)(len - vbase) << count);
  int n = count + (int)vextra;
  const int dsym = dist_to_sym((int)dist);
  si = hdist[dsym];
  count = (int)(si & 0x1F);
  dist_sym_value(dsym, vbase, vextra);
  v |= ((uint64_t)(si >> 5) | ((uint64_t)(dist - vbase) << count)) << n;
  n += count + (int)vextra;
  value = v;
  nbits = n;
}

// The header of a dynamic block after the 3 type bits (write_dynamic_codes,
// zd.ml:919-941) as a sequence of (value, nbits) items: item index -> bits.
// Items: 0 hlit(5) 1 hdist(5) 2 hclen(4), then hclen+4 lengths of 3 bits, then
// the codelen symbols.  Returns the number of items.
ZD_HD int dyn_header_items(const BlockCoder &c) { return 3 + (c.hclen + 4) + c.codelen_syms_len; }
ZD_HD void dyn_header_item(const BlockCoder &c, int idx, uint32_t &value, int &nbits) {
  if (idx == 0) { value = (uint32_t)c.hlit; nbits = 5; return; }
  if (idx == 1) { value = (uint32_t)c.hdist; nbits = 5; return; }
  if (idx == 2) { value = (uint32_t)c.hclen; nbits = 4; return; }
  idx -= 3;
  if (idx < c.hclen + 4) {
    value = c.dyn_codelen[k_codelen_order[idx]] & 0x1F;
    nbits = 3;
    return;
  }
  idx -= c.hclen + 4;
  const uint32_t symref = c.codelen_syms[idx];
  const uint32_t sym = symref & 0xFF;
  const uint32_t si = c.dyn_codelen[sym];
  const int count = (int)(si & 0x1F);
  if (sym <= 15) { value = si >> 5; nbits = count; return; }
  const int rb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
  value = (si >> 5) | ((symref >> 8) << count);
  nbits = count + rb;
}

}  // namespace zd
