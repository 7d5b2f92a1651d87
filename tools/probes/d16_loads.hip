// What the d16 forms of the LDS reads leave in the OTHER half of their register on this chip.  (With SRAM ECC on, LLVM's
// d16PreservesUnusedBits() is false and the compiler never emits them; lz_match's second form issues them by hand.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/d16_loads.hip -o /tmp/d16_loads && /tmp/d16_loads
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
  __shared__ unsigned char b[256];
  for (int i = threadIdx.x; i < 256; i += 64) b[i] = (unsigned char)(i + 1);
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned a = (unsigned)(unsigned long long)(lds_u8 *)b + threadIdx.x;
  unsigned lo = 0xAAAABBBBu, hi = 0xAAAABBBBu, both = 0xAAAABBBBu, w = 0xAAAABBBBu, whi = 0xAAAABBBBu;
  asm volatile("ds_read_u8_d16 %0, %5\n\t"
               "ds_read_u8_d16_hi %1, %5 offset:1\n\t"
               "ds_read_u8_d16 %2, %5\n\t"
               "ds_read_u8_d16_hi %2, %5 offset:1\n\t"
               "ds_read_u16_d16 %3, %5 offset:2\n\t"
               "ds_read_u16_d16_hi %4, %5 offset:2\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "+v"(lo), "+v"(hi), "+v"(both), "+v"(w), "+v"(whi)
               : "v"(a)
               : "memory");
  out[threadIdx.x * 5 + 0] = lo; out[threadIdx.x * 5 + 1] = hi; out[threadIdx.x * 5 + 2] = both;
  out[threadIdx.x * 5 + 3] = w; out[threadIdx.x * 5 + 4] = whi;
}
int main() {
  unsigned *d, h[320];
  hipMalloc(&d, sizeof h);
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int t = 0; t < 4; t++)
    printf("lane %d (bytes %02x %02x %02x %02x, registers held aaaabbbb): u8_d16 %08x  u8_d16_hi %08x  u8_d16 then _hi %08x  u16_d16 %08x  u16_d16_hi %08x\n",
           t, t + 1, t + 2, t + 3, t + 4, h[t * 5], h[t * 5 + 1], h[t * 5 + 2], h[t * 5 + 3], h[t * 5 + 4]);
  return 0;
}
