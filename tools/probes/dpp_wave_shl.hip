// Probe (gfx950): v_mov_b32_dpp wave_shl:1 -- lane i takes lane i + 1's value, lane 63 keeps the `old` operand.
// (What parse_tile_macro, deflate.hip, shifts the next positions' match entries with.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned v = threadIdx.x * 3u + 1u;
  out[threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp((int)0xDEADu, (int)v, 0x130, 0xf, 0xf, false);
}
int main() {
  unsigned *d, h[64];
  hipMalloc(&d, 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 63; i++) bad += h[i] != (unsigned)(i + 1) * 3u + 1u;
  bad += h[63] != 0xDEADu;
  printf("DPP_WAVE_SHL1 %s (lane0 %u lane62 %u lane63 %x)\n", bad ? "UNEXPECTED" : "lane i <- lane i+1, lane 63 <- old", h[0], h[62], h[63]);
  return bad != 0;
}
