// walk_masks.h -- the first form of lz_match's walk with lane masks for its flags (round 6: built, exact, slower, shelved).
// This is the text that stood in zipc_amd/csrc/deflate_lane.h between lz_match_runs_pool and the second form (see README.md);
// the library does not build it.

// The first form again, written for the device alone (round 6; lz_match_runs_pool above is the same walk on the shared step of
// the host models and stays for A/B builds, -DZD_MATCH_FIRST_SHARED).  What the shared step costs the device, read off its
// assembly (~76 vector instructions per run slot and iteration, of which this form keeps ~55):
//  * its flags are integers (alive) and bools the compiler turns into 0 / 1 and back around every ballot: here the alive and
//    the walking runs of a slot are lane masks, and "will it walk on?" -- which the shared step works out at the end of a step
//    AND, as "does it walk?", at the start of the next -- is worked out once and carried;
//  * the 64-bit count of trailing zeros, its equality test and select: a min3 over the two halves' counts;
//  * steps counted by a select of 0 / 1 and an add: an add-with-carry of the mask;
//  * the link is read by ds_read_u16 (41 ticks of the LDS pipe at 2 mod 4, profiles/r05_lds_costs.txt): ds_read_u16_d16, 7 at
//    any even address, issued WITH the candidate's bytes and waited for once -- and the reads of all the lane's slots go out
//    together, so an iteration is one round trip to the LDS for the steps and one for the handouts, not one per slot;
//  * one handout for all the slots that finish in an iteration.
// Coordinates are LDS byte addresses of the window (as in the second form): s + c is the byte at address c, a link is the
// u16 at cs + 2 c.  The compiler does not see the reads (asm): each asm waits for its own.
struct WalkSlotMasks { unsigned long long A, W; };  // alive; walks in the next step
template <int NP, typename Sink, typename S, typename Pool>
__device__ __forceinline__ uint32_t lz_match_walk_pool(S s, uint32_t len, Pool &pool, uint32_t pend, uint32_t lane,
                                                        uint32_t cs, int K, int Kq, Sink sink) {
  static_assert(NP == 2, "the reads of two slots are issued by hand");
  static_assert(64u * NP <= POOL_CHUNK, "chunk");
  struct Run { uint32_t p, q, blen, best, maxlen, steps, snap, dn, pw_lo, pw_hi; };
  Run r[NP];
  WalkSlotMasks m[NP];
  uint32_t iters = 0;
#ifdef ZD_MATCH_COUNTS
  MatchCounts mc;
#endif
  auto ballot = [](bool b) { return (unsigned long long)__builtin_amdgcn_ballot_w64(b); };
  auto mine = [](unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); };
  // 8 bytes and the link at coordinate c of both slots: three aligned words each (a misaligned 8-byte read stalls the pipe)
  struct Read { uint64_t w01; uint32_t w2, d; };
  auto read2 = [&](uint32_t c0, uint32_t c1, Read &x0, Read &x1) {
    const uint32_t a0 = c0 & ~3u, a1 = c1 & ~3u, l0 = cs + 2u * c0, l1 = cs + 2u * c1;
    asm volatile("ds_read2_b32 %0, %6 offset1:1\n\t"
                 "ds_read_b32 %1, %6 offset:8\n\t"
                 "ds_read_u16_d16 %2, %7\n\t"
                 "ds_read2_b32 %3, %8 offset1:1\n\t"
                 "ds_read_b32 %4, %8 offset:8\n\t"
                 "ds_read_u16_d16 %5, %9\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x0.w01), "=&v"(x0.w2), "=&v"(x0.d), "=&v"(x1.w01), "=&v"(x1.w2), "=&v"(x1.d)
                 : "v"(a0), "v"(l0), "v"(a1), "v"(l1));  // (no "memory": with it the compiler waits for the iteration's global stores first)
  };
  auto bytes_lo = [](const Read &x, uint32_t c) { return funnel32((uint32_t)(x.w01 >> 32), (uint32_t)x.w01, c * 8u); };
  auto bytes_hi = [](const Read &x, uint32_t c) { return funnel32(x.w2, (uint32_t)(x.w01 >> 32), c * 8u); };
  // a run on position np (behind lim: none -- the slot parks on a valid position and is not alive); its reads come from `x`
  auto start = [&](Run &n, uint32_t np, uint32_t lim) {
    n.p = np < lim ? np : (lim ? lim - 1u : 0u);
    n.q = n.p; n.blen = MIN_MATCH_LEN - 1; n.best = 0; n.snap = SNAP_NONE; n.steps = 0;
    n.maxlen = len - n.p < (uint32_t)MAX_MATCH_LEN ? len - n.p : (uint32_t)MAX_MATCH_LEN;
  };
  auto started = [&](Run &n, const Read &x, unsigned long long alive) -> unsigned long long {  // -> the runs that walk (zd.ml:1181,1187)
    n.pw_lo = bytes_lo(x, n.p); n.pw_hi = bytes_hi(x, n.p); n.dn = x.d;
    return K != 0 ? alive & ballot(n.dn != 0xFFFFu) & ballot(n.maxlen > (uint32_t)(MIN_MATCH_LEN - 1)) & ballot(n.dn <= (uint32_t)MAX_MATCH_DIST) : 0ull;
  };
  auto fetch = [&]() -> uint32_t { return pool.take(0xFFFFFFFFu); };
  static_assert(!Pool::WANTS_OLDEST, "the tile's pool");
  // (every load of the wave has landed by now -- the window's went into the LDS a barrier ago -- but the compiler's count of them
  // is carried round the tile loop, and where it thinks a register of this loop may still be a load's destination it waits
  // for vmcnt(0): for the iteration's own global STORES.  Told here, once a tile, it has nothing to wait for inside.)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  uint32_t next = fetch();  // my chunk is [next, cend)
  uint32_t cend = pend - next > pool.size ? next + pool.size : pend;
  bool empty = next >= pend;
  {
    Read x[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const uint32_t off = lane + 64u * (uint32_t)i;
      const uint32_t np = off < cend - next ? next + off : cend;
      m[i].A = ballot(np < cend);
      start(r[i], np, cend);
    }
    read2(r[0].p, r[1].p, x[0], x[1]);
#pragma unroll
    for (int i = 0; i < NP; i++) m[i].W = started(r[i], x[i], m[i].A);
  }
  next = cend - next > 64u * NP ? next + 64u * NP : cend;
  for (;;) {
    iters++;
    ZD_COUNT(1, 1); ZD_COUNT(2, 1);
    uint32_t qc[NP];
    Read x[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) qc[i] = mine(m[i].W) ? r[i].q - r[i].dn : r[i].p;
    read2(qc[0], qc[1], x[0], x[1]);
    unsigned long long fin[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
      Run &n = r[i];
      const unsigned long long W = m[i].W;
      ZD_COUNT(8, __builtin_popcountll(m[i].A));
      const uint32_t lo = bytes_lo(x[i], qc[i]) ^ n.pw_lo, hi = bytes_hi(x[i], qc[i]) ^ n.pw_hi;
      // bytes in common, 0 .. 8: v_ffbl_b32 of 0 is ~0, so an equal half drops out of the minimum
      uint32_t fl, fh;
      asm("v_ffbl_b32 %0, %1" : "=v"(fl) : "v"(lo));
      asm("v_ffbl_b32 %0, %1" : "=v"(fh) : "v"(hi));
      fl >>= 3; fh = (fh >> 3) + 4u;
      uint32_t l = fl < fh ? fl : fh;
      l = l < 8u ? l : 8u;
      const unsigned long long more8 = W & ballot(l == 8u) & ballot(n.maxlen > 8u);  // the first 8 bytes agree and there are others
      l = l < n.maxlen ? l : n.maxlen;  // (the last 7 positions of a stream: what lies behind its end does not count)
      if (more8) {
        if (mine(more8)) {
          // before the long compare, the 8 bytes that END at best_len (they include the bytes already tested); what fails
          // here is at most best_len long
          bool same = true;
          if (n.blen >= 8u) same = load_u64_words(s, qc[i] + n.blen - 7u) == load_u64_words(s, n.p + n.blen - 7u);
          if (same) l = common_prefix_t<true>(s, qc[i], n.p, n.maxlen, 8u);
        }
      }
      unsigned long long carry_out;
      asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(n.steps), "=s"(carry_out) : "v"(n.steps), "s"(W));  // steps += walk
      const bool better = mine(W & ballot(l > n.blen));
      const uint32_t dist = n.p - qc[i];
      n.best = better ? ((dist << 9) | l) : n.best;
      n.blen = better ? l : n.blen;
      if (Kq != 0) n.snap = mine(W & ballot(n.steps == (uint32_t)Kq)) ? n.best : n.snap;
      n.q = qc[i];
      n.dn = x[i].d;
      // does the next step walk?  (l < maxlen implies best_len < maxlen; zd.ml:1194: after l == maxlen nothing later can be longer)
      const unsigned long long more = W & ballot(l != n.maxlen) & ballot(n.dn != 0xFFFFu) & ballot(n.steps != (uint32_t)K) &
                                      ballot(dist + n.dn <= (uint32_t)MAX_MATCH_DIST);
      fin[i] = m[i].A & ~more;
      m[i].W = more;
    }
    // finished positions are stored, and their slots take the pool's next positions: one handout for all of them, slot 0's
    // lanes first
    unsigned long long fu = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) fu |= fin[i];
    if (fu) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < NP; i++) {
        if (Kq == 0) { if (mine(fin[i])) sink(r[i].p, r[i].best, 0u); }  // (wave-uniform: as a select the compiler builds the flag's lane mask in vector registers every iteration)
        else if (mine(fin[i])) sink(r[i].p, r[i].best, r[i].snap != SNAP_NONE ? r[i].snap : r[i].best);
      }
      uint32_t taken = 0, rank[NP];
#pragma unroll
      for (int i = 0; i < NP; i++) {
        rank[i] = taken + __builtin_amdgcn_mbcnt_hi((uint32_t)(fin[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fin[i], 0u));
        taken += (uint32_t)__builtin_popcountll(fin[i]);
      }
      ZD_COUNT(6, 1); ZD_COUNT(7, taken);
      const uint32_t rem = cend - next;
      uint32_t c = 0, ce = 0;
      const bool refill = taken > rem && !empty;  // wave-uniform: the chunk runs out within this handout
      if (refill) {
        c = fetch();
        ce = pend - c > pool.size ? c + pool.size : pend;
        empty = c >= pend;
      }
      Run nr[NP];
      unsigned long long alive[NP];
#pragma unroll
      for (int i = 0; i < NP; i++) {
        uint32_t np = rank[i] < rem ? next + rank[i] : cend, lim = cend;
        if (refill && rank[i] >= rem) { np = rank[i] - rem < ce - c ? c + (rank[i] - rem) : ce; lim = ce; }
        alive[i] = fin[i] & ballot(np < lim);
        start(nr[i], np, lim);
      }
      if (refill) { next = ce - c > taken - rem ? c + (taken - rem) : ce; cend = ce; }
      else next = rem > taken ? next + taken : cend;
      Read y[NP];
      read2(nr[0].p, nr[1].p, y[0], y[1]);
#pragma unroll
      for (int i = 0; i < NP; i++) {
        const unsigned long long w = started(nr[i], y[i], alive[i]);
        if (mine(fin[i])) r[i] = nr[i];
        m[i].A = (m[i].A & ~fin[i]) | alive[i];
        m[i].W = (m[i].W & ~fin[i]) | w;
      }
    }
    unsigned long long any = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) any |= m[i].A;
    if (any == 0) break;
  }
#ifdef ZD_MATCH_COUNTS
  mc.flush(lane);
#endif
  return iters;
}

