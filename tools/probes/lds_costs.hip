// Probe (gfx950): what do the LDS accesses of a hash-chain walk cost a CU?  One 1024-thread workgroup per CU (16 waves,
// the shape of lz_match_window_kernel) issues one kind of access back to back at random or consecutive addresses inside
// a 128 KiB window; prints LDS clocks per wave-instruction per CU (wall clocks of the workgroup x 1 / instructions its
// 16 waves issued).  Build: hipcc --offload-arch=gfx950 -O2 lds_costs.hip -o lds_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int WIN = 128 * 1024;
constexpr int ITERS = 512, UNROLL = 8;

enum Mode {
  R32_RANDOM, R32_SEQ, U16_EVEN, U16_ANY, U8_ANY, R64_ALIGNED8, R64_ALIGNED4, R64_ANY, R2X32, R96_ALIGNED4, R128_ALIGNED16,
  R128_SEQ, W64_SEQ, W128_SEQ, W32_SEQ, R64_SEQ, BPERM, MAX_RANDOM, W64_RANDOM_HALF, U16_AL4, U16_OFF2, U16D16_OFF2, U16D16HI_OFF2, R2X64, W16_OFF2, W16_AL4, W8_ANY, U16_SEQ, U16D16_ANY, U8D16_ANY, U16D16_EVEN, U16D16_AL4, R32_ANY, N_MODES
};
static const char *names[N_MODES] = {
  "ds_read_b32 random", "ds_read_b32 consecutive", "ds_read_u16 random even", "ds_read_u16 random any byte", "ds_read_u8 random",
  "ds_read_b64 random 8-aligned", "ds_read_b64 random 4-aligned", "ds_read_b64 random any byte", "ds_read2_b32 random 4-aligned (0,1)",
  "ds_read_b96 random 4-aligned", "ds_read_b128 random 16-aligned", "ds_read_b128 consecutive", "ds_write_b64 consecutive",
  "ds_write_b128 consecutive", "ds_write_b32 consecutive", "ds_read_b64 consecutive", "ds_bpermute_b32 random lanes", "ds_max_u32 random",
  "ds_write_b64 consecutive, a third of the lanes", "ds_read_u16 random 4-aligned", "ds_read_u16 random at 4k+2", "ds_read_u16_d16 random at 4k+2",
  "ds_read_u16_d16_hi random at 4k+2", "ds_read2_b64 random 8-aligned (0,1)", "ds_write_b16 random at 4k+2", "ds_write_b16 random 4-aligned", "ds_write_b8 random",
  "ds_read_u16 consecutive (2 B per lane)", "ds_read_u16_d16 random any byte", "ds_read_u8_d16 random", "ds_read_u16_d16 random even",
  "ds_read_u16_d16 random 4-aligned", "ds_read_b32 random any byte"};

template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned long long *clocks, unsigned *sink, unsigned seed) {
  __shared__ __attribute__((aligned(16))) unsigned char win[WIN];
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  for (unsigned i = tid * 16u; i < (unsigned)WIN; i += 1024u * 16u) *(uint4 *)(win + i) = make_uint4(i, i * 3u, i * 5u, i * 7u);
  __syncthreads();
  unsigned a[UNROLL];
  unsigned x = (tid * 2654435761u) ^ seed ^ (blockIdx.x * 40503u);
  for (int k = 0; k < UNROLL; k++) {
    x = x * 1664525u + 1013904223u;
    unsigned r = (x >> 8) % (unsigned)(WIN - 64);
    unsigned wave_base = (tid >> 6) * 8192u;  // consecutive forms: a wave's own 8 KiB
    switch (MODE) {
      case R32_RANDOM: case R2X32: case R96_ALIGNED4: case R64_ALIGNED4: case MAX_RANDOM: case U16_AL4: case W16_AL4: a[k] = r & ~3u; break;
      case U16_OFF2: case U16D16_OFF2: case U16D16HI_OFF2: case W16_OFF2: a[k] = (r & ~3u) + 2u; break;
      case R2X64: a[k] = r & ~7u; break;
      case W8_ANY: a[k] = r; break;
      case U16_SEQ: a[k] = wave_base + lane * 2u + (unsigned)k * 128u; break;
      case R32_SEQ: case W32_SEQ: a[k] = wave_base + lane * 4u + (unsigned)k * 256u; break;
      case U16_EVEN: a[k] = r & ~1u; break;
      case U16_ANY: case U8_ANY: case R64_ANY: case U16D16_ANY: case U8D16_ANY: case R32_ANY: a[k] = r; break;
      case U16D16_EVEN: a[k] = r & ~1u; break;
      case U16D16_AL4: a[k] = r & ~3u; break;
      case R64_ALIGNED8: a[k] = r & ~7u; break;
      case R128_ALIGNED16: a[k] = r & ~15u; break;
      case R128_SEQ: case W128_SEQ: a[k] = wave_base + lane * 16u + (unsigned)(k & 3) * 1024u; break;
      case W64_SEQ: case R64_SEQ: case W64_RANDOM_HALF: a[k] = wave_base + lane * 8u + (unsigned)k * 512u; break;
      case BPERM: a[k] = (r & 63u) * 4u; break;
      default: a[k] = 0; break;
    }
  }
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0;
      switch (MODE) {
        case R32_RANDOM: case R32_SEQ: asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(a[k])); break;
        case U16_EVEN: case U16_ANY: case U16_AL4: case U16_OFF2: case U16_SEQ: asm volatile("ds_read_u16 %0, %1" : "=v"(v0) : "v"(a[k])); break;
        case U16D16_OFF2: case U16D16_ANY: case U16D16_EVEN: case U16D16_AL4: asm volatile("ds_read_u16_d16 %0, %1" : "+v"(v0) : "v"(a[k])); break;
        case U8D16_ANY: asm volatile("ds_read_u8_d16 %0, %1" : "+v"(v0) : "v"(a[k])); break;
        case R32_ANY: asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(a[k])); break;
        case U16D16HI_OFF2: asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(v0) : "v"(a[k])); break;
        case R2X64: {
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          u4 v;
          asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(a[k]));
          asm volatile("s_waitcnt lgkmcnt(7)");
          v0 = v.x ^ v.y ^ v.z ^ v.w;
          break;
        }
        case W16_OFF2: case W16_AL4: asm volatile("ds_write_b16 %0, %1" :: "v"(a[k]), "v"(acc) : "memory"); break;
        case W8_ANY: asm volatile("ds_write_b8 %0, %1" :: "v"(a[k]), "v"(acc) : "memory"); break;
        case U8_ANY: asm volatile("ds_read_u8 %0, %1" : "=v"(v0) : "v"(a[k])); break;
        case R64_ALIGNED8: case R64_ALIGNED4: case R64_ANY: case R64_SEQ: {
          unsigned long long v;
          asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a[k]));
          asm volatile("s_waitcnt lgkmcnt(7)");
          v0 = (unsigned)v ^ (unsigned)(v >> 32);
          break;
        }
        case R2X32: {
          unsigned long long v;
          asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(a[k]));
          asm volatile("s_waitcnt lgkmcnt(7)");
          v0 = (unsigned)v ^ (unsigned)(v >> 32);
          break;
        }
        case R96_ALIGNED4: {
          typedef unsigned u3 __attribute__((ext_vector_type(3)));
          u3 v;
          asm volatile("ds_read_b96 %0, %1" : "=v"(v) : "v"(a[k]));
          asm volatile("s_waitcnt lgkmcnt(7)");
          v0 = v.x ^ v.y ^ v.z;
          break;
        }
        case R128_ALIGNED16: case R128_SEQ: {
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          u4 v;
          asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a[k]));
          asm volatile("s_waitcnt lgkmcnt(7)");
          v0 = v.x ^ v.y ^ v.z ^ v.w;
          break;
        }
        case W32_SEQ: asm volatile("ds_write_b32 %0, %1" :: "v"(a[k]), "v"(acc) : "memory"); break;
        case W64_SEQ: { unsigned long long v = acc; asm volatile("ds_write_b64 %0, %1" :: "v"(a[k]), "v"(v) : "memory"); break; }
        case W64_RANDOM_HALF: { unsigned long long v = acc; if ((lane % 3u) == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(a[k]), "v"(v) : "memory"); break; }
        case W128_SEQ: {
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          u4 v = {acc, acc, acc, acc};
          asm volatile("ds_write_b128 %0, %1" :: "v"(a[k]), "v"(v) : "memory");
          break;
        }
        case BPERM: asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v0) : "v"(a[k]), "v"(acc)); break;
        case MAX_RANDOM: asm volatile("ds_max_u32 %0, %1" :: "v"(a[k]), "v"(acc) : "memory"); break;
        default: break;
      }
      (void)v1; (void)v2; (void)v3;
      asm volatile("s_waitcnt lgkmcnt(7)");
      acc += v0;
      // next address of this slot: a fixed odd stride through the window (random forms), the same place (consecutive forms)
      if (MODE == R32_RANDOM || MODE == R2X32 || MODE == R96_ALIGNED4 || MODE == R64_ALIGNED4 || MODE == MAX_RANDOM || MODE == U16_AL4 || MODE == W16_AL4 || MODE == U16D16_AL4 ||
          MODE == U16_OFF2 || MODE == U16D16_OFF2 || MODE == U16D16HI_OFF2 || MODE == W16_OFF2) { a[k] += 4u * 7919u; if (a[k] >= (unsigned)(WIN - 64)) a[k] -= (unsigned)(WIN - 64) & ~3u; }
      if (MODE == U16_EVEN || MODE == U16D16_EVEN) { a[k] += 2u * 15013u; if (a[k] >= (unsigned)(WIN - 64)) a[k] -= (unsigned)(WIN - 64) & ~1u; }
      if (MODE == U16_ANY || MODE == U8_ANY || MODE == R64_ANY || MODE == W8_ANY || MODE == U16D16_ANY || MODE == U8D16_ANY || MODE == R32_ANY) { a[k] += 30011u; if (a[k] >= (unsigned)(WIN - 64)) a[k] -= (unsigned)(WIN - 64); }
      if (MODE == R64_ALIGNED8 || MODE == R2X64) { a[k] += 8u * 3761u; if (a[k] >= (unsigned)(WIN - 64)) a[k] -= (unsigned)(WIN - 64) & ~7u; }
      if (MODE == R128_ALIGNED16) { a[k] += 16u * 1877u; if (a[k] >= (unsigned)(WIN - 64)) a[k] -= (unsigned)(WIN - 64) & ~15u; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (tid == 0) clocks[blockIdx.x] = t2 - t0;
  if (acc == 0x12345u) sink[0] = acc + (unsigned)(t1 - t0);
}

template <int MODE>
static void run(unsigned long long *d_clk, unsigned *d_sink, int n_wg) {
  std::vector<unsigned long long> h(n_wg);
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(probe<MODE>, dim3(n_wg), dim3(1024), 0, 0, d_clk, d_sink, 12345u + rep);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_clk, n_wg * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto c : h) s += (double)c;
    s /= n_wg;
    if (s < best) best = s;
  }
  // shader clocks (s_memtime counts at 100 MHz on gfx950?  readcyclecounter = s_memtime): report both raw and per instruction
  const double insts = 16.0 * ITERS * UNROLL;
  printf("%-48s %10.0f ticks/WG  %8.3f ticks per wave-instruction per CU\n", names[MODE], best, best / insts);
}

int main() {
  unsigned long long *d_clk; unsigned *d_sink;
  const int n_wg = 256;
  hipMalloc(&d_clk, n_wg * 8); hipMalloc(&d_sink, 4);
  // calibrate the tick: a kernel of known duration
  run<R32_SEQ>(d_clk, d_sink, n_wg);
  run<R32_RANDOM>(d_clk, d_sink, n_wg);
  run<U16_EVEN>(d_clk, d_sink, n_wg);
  run<U16_ANY>(d_clk, d_sink, n_wg);
  run<U8_ANY>(d_clk, d_sink, n_wg);
  run<R64_SEQ>(d_clk, d_sink, n_wg);
  run<R64_ALIGNED8>(d_clk, d_sink, n_wg);
  run<R64_ALIGNED4>(d_clk, d_sink, n_wg);
  run<R64_ANY>(d_clk, d_sink, n_wg);
  run<R2X32>(d_clk, d_sink, n_wg);
  run<R96_ALIGNED4>(d_clk, d_sink, n_wg);
  run<R128_SEQ>(d_clk, d_sink, n_wg);
  run<R128_ALIGNED16>(d_clk, d_sink, n_wg);
  run<W32_SEQ>(d_clk, d_sink, n_wg);
  run<W64_SEQ>(d_clk, d_sink, n_wg);
  run<W64_RANDOM_HALF>(d_clk, d_sink, n_wg);
  run<W128_SEQ>(d_clk, d_sink, n_wg);
  run<BPERM>(d_clk, d_sink, n_wg);
  run<MAX_RANDOM>(d_clk, d_sink, n_wg);
  run<U16_AL4>(d_clk, d_sink, n_wg);
  run<U16_OFF2>(d_clk, d_sink, n_wg);
  run<U16D16_OFF2>(d_clk, d_sink, n_wg);
  run<U16D16HI_OFF2>(d_clk, d_sink, n_wg);
  run<U16_SEQ>(d_clk, d_sink, n_wg);
  run<U16D16_ANY>(d_clk, d_sink, n_wg);
  run<U8D16_ANY>(d_clk, d_sink, n_wg);
  run<U16D16_EVEN>(d_clk, d_sink, n_wg);
  run<U16D16_AL4>(d_clk, d_sink, n_wg);
  run<R32_ANY>(d_clk, d_sink, n_wg);
  run<R2X64>(d_clk, d_sink, n_wg);
  run<W16_OFF2>(d_clk, d_sink, n_wg);
  run<W16_AL4>(d_clk, d_sink, n_wg);
  run<W8_ANY>(d_clk, d_sink, n_wg);
  // wall-clock calibration of the tick
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<R32_RANDOM>, dim3(n_wg), dim3(1024), 0, 0, d_clk, d_sink, 999u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(n_wg);
  hipMemcpy(h.data(), d_clk, n_wg * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto c : h) s += (double)c; s /= n_wg;
  printf("calibration: kernel %.3f ms, %0.f ticks per WG -> %.1f MHz tick (one WG per CU, all at once)\n", ms, s, s / (ms * 1e3));
  return 0;
}
