// wave.h -- the wave64 collectives the span decoder (inflate_span.h) is written in.
//
// Under hipcc these are the gfx950 instructions (ballot, readlane, ds_bpermute, the DPP
// scan of wave_ops.h).  Under g++ the same names come from tests/host_sim/wave_emu.h, a
// TEST-ONLY emulation that runs the 64 lanes of one wave as 64 coroutines which meet at
// every collective -- so the very code the kernel runs (inflate_span.h) is also checked
// against the oracle in the GPU-less build container.  The product never loads that.
//
// Rules the span code keeps so that both forms mean the same:
//   * collectives are only called under wave-uniform control flow (all 64 lanes arrive);
//   * where one lane reads LDS another lane wrote, a wv::sync() stands between the two
//     (free on the GPU: a wave executes in lockstep and its LDS operations stay in order).
#pragma once

#include "zd_common.h"

#if defined(__HIPCC__)
#include "wave_ops.h"

#define ZD_WV __device__ __forceinline__

namespace zd {
namespace wv {

__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// value of lane l, l wave-uniform
__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
// value of lane src, src per lane
__device__ __forceinline__ uint32_t shfl(uint32_t v, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}
__device__ __forceinline__ uint32_t scan_incl(uint32_t v) { return wave_scan_incl(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// (the fence emits no instruction: it tells the compiler that LDS accesses do not move across this point)
__device__ __forceinline__ void sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// global stores of this wave before, its global loads after
__device__ __forceinline__ void fence_global() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }
// a word other waves (or this one, a while ago) may have stored: not from this CU's vector cache
__device__ __forceinline__ uint32_t load_coherent(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_and(uint32_t *p, uint32_t v) { atomicAnd(p, v); }

struct Quad { uint32_t x, y, z, w; };
__device__ __forceinline__ Quad load_quad(const uint8_t *p) {  // 16 bytes, any alignment
  const u32x4 v = load16_unaligned(p);
  Quad q;
  q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
  return q;
}
__device__ __forceinline__ void store_quad(uint8_t *p, const Quad &q) {
  u32x4 v;
  v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
  store16_unaligned(p, v);
}

}  // namespace wv
}  // namespace zd

#else
#include "wave_emu.h"  // tests/host_sim (test tooling): the same names on coroutines
#endif
