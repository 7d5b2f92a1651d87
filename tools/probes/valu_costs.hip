// Probe (gfx950): how many clocks does a SIMD need per wave64 vector instruction of the kinds the deflate kernels are made of,
// with 1, 2 and 4 waves per SIMD issuing independent instructions?  One workgroup per CU; prints clocks per instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_costs.hip -o valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 2048;
enum Op { ADD, XOR, AND_OR, CNDMASK, CMP_VCC, CMP_SGPR, ALIGNBYTE, LSHL_OR, MIN, FFBL, MBCNT, FMA, LSHRREV_B64, MUL_LO, READLANE, MOV_DPP, ADD3, SALU, MIX_VS, CND_SGPR, CND_FRESH, CND_OTHER, BFI, AND, OR, LSHL, LSHR, SUB, BFE, MAX, PERM, CMP_CND_PAIR, CND_VCC_INIT, ADDC, SUBREV, OR3, N_OPS };
static const char *names[N_OPS] = {"v_add_u32", "v_xor_b32", "v_and_or_b32", "v_cndmask_b32 (vcc)", "v_cmp_lt_u32 -> vcc", "v_cmp_lt_u32 -> sgpr pair", "v_alignbyte_b32",
  "v_lshl_or_b32", "v_min_u32", "v_ffbl_b32", "v_mbcnt_lo_u32_b32", "v_fma_f32", "v_lshrrev_b64", "v_mul_lo_u32", "v_readlane_b32", "v_mov_b32 dpp row_shr:1", "v_add3_u32",
  "s_add_u32 (scalar)", "v_add_u32 + s_add_u32 alternating (per pair)", "v_cndmask_b32 e64 (sgpr pair mask)", "v_cndmask_b32 dst != src (vcc)",
  "v_cndmask_b32 src1 = other reg, e64", "v_bfi_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_bfe_u32", "v_max_u32", "v_perm_b32",
  "v_cmp + v_cndmask pair (per pair)", "v_cndmask_b32 (vcc set to exec before the loop)", "v_addc_co_u32", "v_subrev_u32", "v_or3_b32"};

template <int OP>
__global__ __launch_bounds__(1024) void probe(unsigned long long *clocks, unsigned *sink) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 21, a7 = a0 ^ 55;
  unsigned s0 = blockIdx.x;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
  unsigned long long w0 = a0, w1 = a1;
  if (OP == CND_VCC_INIT) asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
  if (OP == CND_SGPR || OP == CND_OTHER) asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#define R8(x) x(a0) x(a1) x(a2) x(a3) x(a4) x(a5) x(a6) x(a7)
    switch (OP) {
#define X(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a0));
      case ADD: R8(X) break;
#undef X
#define X(r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case XOR: R8(X) break;
#undef X
#define X(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a1), "v"(a2));
      case AND_OR: R8(X) break;
#undef X
#define X(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(a1) : );
      case CNDMASK: R8(X) break;
#undef X
#define X(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(r), "v"(a1) : "vcc");
      case CMP_VCC: R8(X) break;
#undef X
#define X(r) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" :: "v"(r), "v"(a1) : "s20", "s21");
      case CMP_SGPR: R8(X) break;
#undef X
#define X(r) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a1), "v"(a2));
      case ALIGNBYTE: R8(X) break;
#undef X
#define X(r) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(a1));
      case LSHL_OR: R8(X) break;
#undef X
#define X(r) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case MIN: R8(X) break;
#undef X
#define X(r) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r));
      case FFBL: R8(X) break;
#undef X
#define X(r) asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case MBCNT: R8(X) break;
#undef X
      case FMA:
        asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2\n"
                     "v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        break;
      case LSHRREV_B64:
        asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n"
                     "v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1" : "+v"(w0), "+v"(w1));
        break;
#define X(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case MUL_LO: R8(X) break;
#undef X
#define X(r) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(r) : "s20");
      case READLANE: R8(X) break;
#undef X
#define X(r) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));
      case MOV_DPP: R8(X) break;
#undef X
#define X(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(a1), "v"(a2));
      case ADD3: R8(X) break;
#undef X
      case SALU:
        asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc");
        break;
      case MIX_VS:
        asm volatile("v_add_u32 %1, %1, %2\n s_add_u32 %0, %0, 1\n v_add_u32 %2, %2, %1\n s_add_u32 %0, %0, 1\n v_add_u32 %3, %3, %2\n s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, %1\n s_add_u32 %0, %0, 1\n"
                     "v_add_u32 %1, %1, %2\n s_add_u32 %0, %0, 1\n v_add_u32 %2, %2, %1\n s_add_u32 %0, %0, 1\n v_add_u32 %3, %3, %2\n s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, %1\n s_add_u32 %0, %0, 1"
                     : "+s"(s0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "scc");
        break;
#define X(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r) : "v"(a1));
      case CND_SGPR: R8(X) break;
#undef X
      case CND_FRESH: case CND_VCC_INIT:
        asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
                     "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(a1));
        break;
#define X(r) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "+v"(r) : "v"(a1), "v"(a2));
      case CND_OTHER: R8(X) break;
#undef X
#define X(r) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(a1), "v"(a2));
      case BFI: R8(X) break;
#undef X
#define X(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case AND: R8(X) break;
#undef X
#define X(r) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case OR: R8(X) break;
#undef X
#define X(r) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r));
      case LSHL: R8(X) break;
#undef X
#define X(r) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r));
      case LSHR: R8(X) break;
#undef X
#define X(r) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case SUB: R8(X) break;
#undef X
#define X(r) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r));
      case BFE: R8(X) break;
#undef X
#define X(r) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case MAX: R8(X) break;
#undef X
#define X(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a1), "v"(a2));
      case PERM: R8(X) break;
#undef X
      case CMP_CND_PAIR:
        asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                     "v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc"
                     : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(a1) : "vcc");
        break;
#define X(r) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r) : "v"(a1) : "vcc");
      case ADDC: R8(X) break;
#undef X
#define X(r) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(r) : "v"(a1));
      case SUBREV: R8(X) break;
#undef X
#define X(r) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a1), "v"(a2));
      case OR3: R8(X) break;
#undef X
      default: break;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&clocks[blockIdx.x], t1 - t0);
  unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ (unsigned)f0 ^ (unsigned)f1 ^ (unsigned)f2 ^ (unsigned)f3 ^ (unsigned)w0 ^ (unsigned)w1;
  if (r == 0x12345u) sink[0] = r;
}

template <int OP>
static void run(unsigned long long *d_clk, unsigned *d_sink) {
  const int n_wg = 256;
  std::vector<unsigned long long> h(n_wg);
  printf("%-48s", names[OP]);
  for (int threads : {256, 512, 1024}) {
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipMemset(d_clk, 0, n_wg * 8);
      hipLaunchKernelGGL(probe<OP>, dim3(n_wg), dim3(threads), 0, 0, d_clk, d_sink);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d_clk, n_wg * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (auto c : h) s += (double)c;
      s /= n_wg;
      if (s < best) best = s;
    }
    const int waves_per_simd = threads / 256;
    const double per = best / (double(ITERS) * (OP == CMP_CND_PAIR ? 4 : 8) * waves_per_simd);
    printf("  %d wave/SIMD: %6.2f", waves_per_simd, per);
  }
  printf("   clocks per instruction per SIMD\n");
}

int main() {
  unsigned long long *d_clk; unsigned *d_sink;
  (void)hipMalloc(&d_clk, 256 * 8); (void)hipMalloc(&d_sink, 4);
  run<ADD>(d_clk, d_sink); run<XOR>(d_clk, d_sink); run<AND_OR>(d_clk, d_sink); run<CNDMASK>(d_clk, d_sink); run<CMP_VCC>(d_clk, d_sink);
  run<CMP_SGPR>(d_clk, d_sink); run<ALIGNBYTE>(d_clk, d_sink); run<LSHL_OR>(d_clk, d_sink); run<MIN>(d_clk, d_sink); run<FFBL>(d_clk, d_sink);
  run<MBCNT>(d_clk, d_sink); run<FMA>(d_clk, d_sink); run<LSHRREV_B64>(d_clk, d_sink); run<MUL_LO>(d_clk, d_sink); run<READLANE>(d_clk, d_sink);
  run<MOV_DPP>(d_clk, d_sink); run<ADD3>(d_clk, d_sink); run<SALU>(d_clk, d_sink); run<MIX_VS>(d_clk, d_sink);
  run<CND_SGPR>(d_clk, d_sink); run<CND_FRESH>(d_clk, d_sink); run<CND_VCC_INIT>(d_clk, d_sink); run<CND_OTHER>(d_clk, d_sink); run<CMP_CND_PAIR>(d_clk, d_sink);
  run<BFI>(d_clk, d_sink); run<AND>(d_clk, d_sink); run<OR>(d_clk, d_sink); run<LSHL>(d_clk, d_sink); run<LSHR>(d_clk, d_sink); run<SUB>(d_clk, d_sink);
  run<SUBREV>(d_clk, d_sink); run<BFE>(d_clk, d_sink); run<MAX>(d_clk, d_sink); run<PERM>(d_clk, d_sink); run<ADDC>(d_clk, d_sink); run<OR3>(d_clk, d_sink);
  return 0;
}
