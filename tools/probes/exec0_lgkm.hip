// Do LDS instructions issued with EXEC = 0 count in lgkmcnt like any other?  lz_match's hand-written rounds
// (deflate_lane.h scan_rounds_lds) wait for a run slot's three reads with s_waitcnt lgkmcnt(3) while the next slot's three are
// behind them -- also when that slot has no walking lane and its reads were issued under an empty mask.  If such reads were
// not counted, the wait would pass at once and the registers be read before the data is there.  Here: a read under the
// full mask, three reads under EXEC = 0, s_waitcnt lgkmcnt(3), and the first read's register is looked at at once.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/exec0_lgkm.hip -o /tmp/exec0_lgkm && /tmp/exec0_lgkm
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void probe(unsigned *bad) {
  __shared__ unsigned tbl[16384];
  for (unsigned i = threadIdx.x; i < 16384u; i += 1024u) tbl[i] = i * 2654435761u;
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  const unsigned base = (unsigned)(unsigned long long)(lds_u32 *)tbl;
  unsigned x = threadIdx.x * 40503u + blockIdx.x * 9973u, wrong = 0;
  for (int it = 0; it < 4096; it++) {
    x = x * 1664525u + 1013904223u;
    const unsigned i = (x >> 9) & 16383u;
    unsigned v = 0xDEADBEEFu, j0 = 0, j1 = 0, j2 = 0;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "ds_read_b32 %[v], %[a]\n\t"
                 "s_mov_b64 exec, 0\n\t"
                 "ds_read_b32 %[j0], %[a]\n\t"
                 "ds_read_b32 %[j1], %[a] offset:4\n\t"
                 "ds_read_b32 %[j2], %[a] offset:8\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt lgkmcnt(3)\n\t"
                 "v_mov_b32 %[v], %[v]\n\t"  // (a use right behind the wait)
                 "s_waitcnt lgkmcnt(0)"
                 : [sv] "=&s"(sv), [v] "+v"(v), [j0] "+v"(j0), [j1] "+v"(j1), [j2] "+v"(j2)
                 : [a] "v"(base + i * 4u)
                 : "memory");
    wrong += v != i * 2654435761u ? 1u : 0u;
  }
  if (wrong) atomicAdd(bad, wrong);
}
int main() {
  unsigned *d, h = 0;
  hipMalloc(&d, 4); hipMemcpy(d, &h, 4, hipMemcpyHostToDevice);
  probe<<<256, 1024>>>(d);
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("reads under EXEC = 0 behind a real one, s_waitcnt lgkmcnt(3), 256 x 1024 threads x 4096 rounds: %u values read before they were there\n", h);
  return h != 0;
}
