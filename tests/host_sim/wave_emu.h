// wave_emu.h -- TEST TOOLING ONLY: the wave64 collectives of zipc_amd/csrc/wave.h on the CPU.
//
// The 64 lanes of one wave run as 64 coroutines (ucontext).  A lane runs until it calls a
// collective, deposits its operand and parks; when all 64 have parked at the SAME collective
// the scheduler publishes the operands and resumes them.  Lanes are resumed in ascending or
// descending order (Emu::descending): code that lets one lane read LDS another lane wrote
// without a wv::sync() between the two behaves differently in the two orders, so the tests
// run both.  A lane that returns while others wait, or lanes waiting at different
// collectives, abort: the span code must keep its collectives under wave-uniform control flow.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#define ZD_WV inline

namespace zd {
namespace wv {

struct Emu {
  static const int N = 64;
  static const size_t STACK = 256 * 1024;
  ucontext_t sched;
  ucontext_t ctx[N];
  char *stack[N];
  bool finished[N];
  int waiting_op[N];
  uint64_t dep[N], snap[N];
  int cur;
  bool descending;
  void (*fn)(int lane, void *arg);
  void *arg;
  Emu() : descending(false) {
    for (int i = 0; i < N; i++) stack[i] = (char *)malloc(STACK);
  }
  ~Emu() {
    for (int i = 0; i < N; i++) free(stack[i]);
  }
  static Emu *&current() {
    static Emu *e = nullptr;
    return e;
  }
  static void trampoline() {
    Emu &E = *current();
    const int me = E.cur;
    E.fn(me, E.arg);
    E.finished[me] = true;
    swapcontext(&E.ctx[me], &E.sched);
  }
  // run fn(lane, arg) on all 64 lanes to completion
  void run(void (*f)(int, void *), void *a) {
    Emu *outer = current();
    current() = this;
    fn = f;
    arg = a;
    for (int i = 0; i < N; i++) {
      finished[i] = false;
      waiting_op[i] = 0;
      getcontext(&ctx[i]);
      ctx[i].uc_stack.ss_sp = stack[i];
      ctx[i].uc_stack.ss_size = STACK;
      ctx[i].uc_link = &sched;
      makecontext(&ctx[i], (void (*)())trampoline, 0);
    }
    for (;;) {
      for (int s = 0; s < N; s++) {
        const int i = descending ? N - 1 - s : s;
        if (finished[i]) continue;
        cur = i;
        waiting_op[i] = 0;
        swapcontext(&sched, &ctx[i]);
      }
      int nfin = 0, op = 0;
      for (int i = 0; i < N; i++) {
        if (finished[i]) { nfin++; continue; }
        if (op == 0) op = waiting_op[i];
        if (waiting_op[i] != op) {
          fprintf(stderr, "wave_emu: lanes wait at different collectives (%d vs %d at lane %d)\n", op, waiting_op[i], i);
          abort();
        }
      }
      if (nfin == N) break;
      if (nfin != 0) {
        fprintf(stderr, "wave_emu: %d lanes returned while the others wait at collective %d\n", nfin, op);
        abort();
      }
      memcpy(snap, dep, sizeof snap);
    }
    current() = outer;
  }
};

enum { OP_BALLOT = 1, OP_VALUE = 2, OP_SYNC = 3, OP_SCAN = 4 };

inline int emu_lane() { return Emu::current()->cur; }
inline void rendezvous(int op, uint64_t v) {
  Emu &E = *Emu::current();
  const int me = E.cur;
  E.dep[me] = v;
  E.waiting_op[me] = op;
  swapcontext(&E.ctx[me], &E.sched);
  E.cur = me;  // (the scheduler set it before resuming; kept for clarity)
}

inline uint64_t ballot(bool p) {
  rendezvous(OP_BALLOT, p ? 1 : 0);
  const Emu &E = *Emu::current();
  uint64_t m = 0;
  for (int i = 0; i < Emu::N; i++)
    if (E.snap[i]) m |= 1ull << i;
  return m;
}
inline bool any(bool p) { return ballot(p) != 0; }
inline uint32_t readlane(uint32_t v, uint32_t l) {
  rendezvous(OP_VALUE, v);
  return (uint32_t)Emu::current()->snap[l & 63u];
}
inline uint32_t shfl(uint32_t v, uint32_t src) {
  rendezvous(OP_VALUE, v);
  return (uint32_t)Emu::current()->snap[src & 63u];
}
inline uint32_t scan_incl(uint32_t v) {
  Emu &E = *Emu::current();
  const int me = E.cur;
  rendezvous(OP_SCAN, v);
  uint32_t s = 0;
  for (int i = 0; i <= me; i++) s += (uint32_t)E.snap[i];
  return s;
}
inline uint32_t uni(uint32_t v) { return readlane(v, 0); }
inline void sync() { rendezvous(OP_SYNC, 0); }
inline void fence_global() { rendezvous(OP_SYNC, 0); }
inline void lds_or(uint32_t *p, uint32_t v) { *p |= v; }
inline uint32_t load_coherent(const uint32_t *p) { return *p; }
inline void lds_and(uint32_t *p, uint32_t v) { *p &= v; }

struct Quad { uint32_t x, y, z, w; };
inline Quad load_quad(const uint8_t *p) {
  Quad q;
  memcpy(&q, p, 16);
  return q;
}
inline void store_quad(uint8_t *p, const Quad &q) { memcpy(p, &q, 16); }

}  // namespace wv
}  // namespace zd
