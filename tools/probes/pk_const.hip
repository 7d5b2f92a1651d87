// Does x - 1, then a saturating + 1, on packed 16-bit halves map 0 -> 0xFFFF and leave the rest (deflate.hip: links_for_window)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const uint32_t *in, uint32_t *out) {
  const u16x2 x = __builtin_bit_cast(u16x2, in[threadIdx.x]);
  const u16x2 one = {1, 1};
  out[threadIdx.x] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat((u16x2)(x - one), one));
}
int main() {
  uint32_t h[8] = {0x00000000u, 0x00010000u, 0x00000001u, 0x80000005u, 0xFFFF0000u, 0x0000FFFFu, 0x12340000u, 0x00001234u}, o[8], *di, *dout;
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof h);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  probe<<<1, 8>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; i++) printf("%08x -> %08x\n", h[i], o[i]);
  return 0;
}
