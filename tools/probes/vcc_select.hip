// Probe (gfx950): what does a select that reads VCC cost in CONTEXT?  profiles/r05_valu_costs.txt has `v_cndmask_b32 ..., vcc`
// (VOP2, implicit VCC) at 19.2 clocks per wave64 instruction whatever the waves per SIMD, in loops that never WRITE vcc, while a
// v_cmp + v_cndmask pair costs 7.8.  The compiled kernels (lz_parse, inflate_batch, deflate_pack) hold hundreds of such selects,
// most of them not directly behind the compare that wrote VCC.  This probe times short groups as the compiler emits them:
//   v_cmp -> vcc, k unrelated vector instructions, m selects on vcc          (k = 0,1,2,4; m = 1,2,4)
//   s_mov / s_and vcc (a scalar instruction wrote the mask), m selects
//   the same selects in the e64 encoding (vcc named as an SGPR pair) and on s[20:21]
//   v_addc_co / v_subb_co chains, s_nop between compare and select
// and prints clocks of the SIMD per GROUP and, with the group's other instructions priced from the one-instruction rows of
// valu_costs (v_cmp 4.37, v_add 2.5, s_* ~2), what is left per select.
// Build: hipcc --offload-arch=gfx950 -O2 vcc_select.hip -o vcc_select
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 2048;

#define CMP "v_cmp_lt_u32 vcc, %0, %4\n"
#define CMP2 "v_cmp_lt_u32 vcc, %1, %4\n"
#define CMPS "v_cmp_lt_u32 s[20:21], %0, %4\n"
#define F1 "v_add_u32 %6, %6, %4\n"
#define F2 F1 "v_add_u32 %7, %7, %5\n"
#define F4 F2 "v_add_u32 %6, %6, %5\n v_add_u32 %7, %7, %4\n"
#define S1 "v_cndmask_b32_e32 %0, %0, %5, vcc\n"
#define S2 S1 "v_cndmask_b32_e32 %1, %1, %5, vcc\n"
#define S4 S2 "v_cndmask_b32_e32 %2, %2, %5, vcc\n v_cndmask_b32_e32 %3, %3, %5, vcc\n"
#define T1 "v_cndmask_b32_e64 %0, %0, %5, vcc\n"
#define T2 T1 "v_cndmask_b32_e64 %1, %1, %5, vcc\n"
#define T4 T2 "v_cndmask_b32_e64 %2, %2, %5, vcc\n v_cndmask_b32_e64 %3, %3, %5, vcc\n"
#define U1 "v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n"
#define U2 U1 "v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n"
#define U4 U2 "v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cndmask_b32_e64 %3, %3, %5, s[20:21]\n"

struct Variant { const char *name; double others; int selects; };
// `others`: the clocks of the group's instructions that are not selects, at 4 waves per SIMD (valu_costs' rows)
#define VARIANTS(X) \
  X(0, "select e32 alone (vcc never written in the loop) x4", S4, 0.0, 4) \
  X(1, "select e64 vcc alone x4", T4, 0.0, 4) \
  X(2, "select e64 s[20:21] alone x4", U4, 0.0, 4) \
  X(3, "v_cmp vcc; 1 select e32", CMP S1, 4.37, 1) \
  X(4, "v_cmp vcc; 2 selects e32", CMP S2, 4.37, 2) \
  X(5, "v_cmp vcc; 4 selects e32", CMP S4, 4.37, 4) \
  X(6, "v_cmp vcc; 1 v_add; 1 select e32", CMP F1 S1, 4.37 + 2.5, 1) \
  X(7, "v_cmp vcc; 2 v_add; 1 select e32", CMP F2 S1, 4.37 + 5.0, 1) \
  X(8, "v_cmp vcc; 4 v_add; 1 select e32", CMP F4 S1, 4.37 + 10.0, 1) \
  X(9, "v_cmp vcc; 4 v_add; 4 selects e32", CMP F4 S4, 4.37 + 10.0, 4) \
  X(10, "v_cmp vcc; 1 select e64 vcc", CMP T1, 4.37, 1) \
  X(11, "v_cmp vcc; 4 selects e64 vcc", CMP T4, 4.37, 4) \
  X(12, "v_cmp vcc; 4 v_add; 4 selects e64 vcc", CMP F4 T4, 4.37 + 10.0, 4) \
  X(13, "v_cmp s[20:21]; 1 select e64 s[20:21]", CMPS U1, 4.37, 1) \
  X(14, "v_cmp s[20:21]; 4 selects e64 s[20:21]", CMPS U4, 4.37, 4) \
  X(15, "v_cmp s[20:21]; 4 v_add; 4 selects e64 s[20:21]", CMPS F4 U4, 4.37 + 10.0, 4) \
  X(16, "s_mov vcc, s[22:23]; 1 select e32", "s_mov_b64 vcc, s[22:23]\n" S1, 2.0, 1) \
  X(17, "s_mov vcc, s[22:23]; 4 selects e32", "s_mov_b64 vcc, s[22:23]\n" S4, 2.0, 4) \
  X(18, "s_mov vcc, s[22:23]; 4 selects e64 vcc", "s_mov_b64 vcc, s[22:23]\n" T4, 2.0, 4) \
  X(19, "s_mov s[20:21], s[22:23]; 4 selects e64 s[20:21]", "s_mov_b64 s[20:21], s[22:23]\n" U4, 2.0, 4) \
  X(20, "v_cmp vcc; s_and vcc, vcc, s[22:23]; 1 select e32", CMP "s_and_b64 vcc, vcc, s[22:23]\n" S1, 4.37 + 2.0, 1) \
  X(21, "v_cmp vcc; s_and vcc, vcc, s[22:23]; 4 selects e32", CMP "s_and_b64 vcc, vcc, s[22:23]\n" S4, 4.37 + 2.0, 4) \
  X(22, "v_cmp vcc; s_nop 0; 1 select e32", CMP "s_nop 0\n" S1, 4.37, 1) \
  X(23, "v_cmp vcc; s_nop 3; 1 select e32", CMP "s_nop 3\n" S1, 4.37, 1) \
  X(24, "v_cmp vcc; select; v_cmp vcc; select (two pairs)", CMP S1 CMP2 "v_cndmask_b32_e32 %1, %1, %5, vcc\n", 8.74, 2) \
  X(25, "v_cmp vcc; v_addc_co x1 (vcc in and out)", CMP "v_addc_co_u32 %0, vcc, %0, %5, vcc\n", 4.37, 1) \
  X(26, "v_cmp vcc; v_addc_co x4 chain", CMP "v_addc_co_u32 %0, vcc, %0, %5, vcc\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_addc_co_u32 %2, vcc, %2, %5, vcc\n v_addc_co_u32 %3, vcc, %3, %5, vcc\n", 4.37, 4) \
  X(27, "v_cmp vcc; v_subb_co x4 chain", CMP "v_subb_co_u32 %0, vcc, %0, %5, vcc\n v_subb_co_u32 %1, vcc, %1, %5, vcc\n v_subb_co_u32 %2, vcc, %2, %5, vcc\n v_subb_co_u32 %3, vcc, %3, %5, vcc\n", 4.37, 4) \
  X(28, "v_add_co vcc; v_addc_co (64-bit add)", "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n", 4.5, 1) \
  X(29, "v_cmp vcc; s_and_saveexec s[20:21], vcc; v_add x2; s_mov exec (a branchless if)", CMP "s_and_saveexec_b64 s[20:21], vcc\n" F2 "s_mov_b64 exec, s[20:21]\n", 4.37 + 5.0, 1) \
  X(30, "v_cmpx (exec masked) ; v_mov x1; s_mov exec, s[22:23] restored", "s_mov_b64 s[20:21], exec\n v_cmpx_lt_u32 %0, %4\n v_mov_b32 %1, %5\n s_mov_b64 exec, s[20:21]\n", 4.37, 1) \
  X(31, "select e32 x4, vcc written ONCE per 2048 iterations by v_cmp before the loop", S4, 0.0, 4)


#define N1 "s_nop 1\n"
#define A1 "v_add_u32 %6, %6, %4\n"
#define A4 A1 A1 A1 A1
#define A8 A4 A4
#define A16 A8 A8
#define A32 A16 A16
#define SB "v_cndmask_b32_e32 %1, %1, %5, vcc\n"
#define VARIANTS2(X) \
  X(32, "v_cmp vcc; s_nop 1; select e32 (as the compiler emits it)", CMP N1 S1, 4.37, 1) \
  X(33, "v_cmp vcc; s_nop 1; select; select", CMP N1 S1 SB, 4.37, 2) \
  X(34, "v_cmp vcc; s_nop 1; select; v_add; select", CMP N1 S1 A1 SB, 4.37 + 2.5, 2) \
  X(35, "v_cmp vcc; s_nop 1; select; 4 v_add; select", CMP N1 S1 A4 SB, 4.37 + 10.0, 2) \
  X(36, "v_cmp vcc; s_nop 1; select; 8 v_add; select", CMP N1 S1 A8 SB, 4.37 + 20.0, 2) \
  X(37, "v_cmp vcc; s_nop 1; select; 16 v_add; select", CMP N1 S1 A16 SB, 4.37 + 40.0, 2) \
  X(38, "v_cmp vcc; 8 v_add; select", CMP A8 S1, 4.37 + 20.0, 1) \
  X(39, "v_cmp vcc; 16 v_add; select", CMP A16 S1, 4.37 + 40.0, 1) \
  X(40, "v_cmp vcc; 32 v_add; select", CMP A32 S1, 4.37 + 80.0, 1) \
  X(41, "v_cmp vcc; 32 v_add; 4 selects", CMP A32 S4, 4.37 + 80.0, 4) \
  X(42, "v_cmp vcc; s_nop 1; select; s_and vcc, vcc, s[22:23]; select", CMP N1 S1 "s_and_b64 vcc, vcc, s[22:23]\n" SB, 4.37 + 2.0, 2) \
  X(43, "v_cmp s[20:21]; s_and vcc, s[20:21], s[22:23]; select (a scalar instruction wrote vcc)", CMPS "s_and_b64 vcc, s[20:21], s[22:23]\n" S1, 4.37 + 2.0, 1) \
  X(44, "v_cmp s[20:21]; s_and vcc, ...; s_nop 1; select", CMPS "s_and_b64 vcc, s[20:21], s[22:23]\n" N1 S1, 4.37 + 2.0, 1) \
  X(45, "v_cmp s[20:21]; s_and vcc, ...; 4 v_add; select", CMPS "s_and_b64 vcc, s[20:21], s[22:23]\n" A4 S1, 4.37 + 2.0 + 10.0, 1) \
  X(46, "v_cmp s[20:21]; s_and vcc, ...; select e64 vcc", CMPS "s_and_b64 vcc, s[20:21], s[22:23]\n" T1, 4.37 + 2.0, 1) \
  X(47, "v_cmp vcc; v_cmp s[20:21]; s_nop 1; select e32 vcc", CMP CMPS N1 S1, 8.74, 1) \
  X(48, "4 v_add; select  (vcc never written in the loop)", A4 S1, 10.0, 1) \
  X(49, "16 v_add; select (vcc never written in the loop)", A16 S1, 40.0, 1) \
  X(50, "v_cmp vcc; s_nop 1; select; ds_bpermute; s_waitcnt; select", CMP N1 S1 "ds_bpermute_b32 %2, %3, %2\n s_waitcnt lgkmcnt(0)\n" SB, 4.37 + 7.0, 2) \
  X(51, "v_cmp vcc; s_nop 1; select; v_mov dpp; select", CMP N1 S1 "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" SB, 4.37 + 4.25, 2) \
  X(52, "v_cmp vcc; s_nop 1; select; s_add (scalar, not vcc); select", CMP N1 S1 "s_add_u32 s24, s24, 1\n" SB, 4.37 + 2.0, 2) \
  X(53, "v_cmp vcc; s_nop 1; select; v_cmp s[20:21] (another mask); select vcc", CMP N1 S1 CMPS SB, 8.74, 2) \
  X(54, "v_cmp vcc; s_nop 1; select e64 vcc; select e64 vcc", CMP N1 T1 "v_cndmask_b32_e64 %1, %1, %5, vcc\n", 4.37, 2) \
  X(55, "v_cmp vcc; s_nop 1; select e32; select e64 vcc", CMP N1 S1 "v_cndmask_b32_e64 %1, %1, %5, vcc\n", 4.37, 2) \
  X(56, "v_cmp vcc; s_nop 1; select e64 vcc; select e32", CMP N1 T1 SB, 4.37, 2)

template <int V>
__global__ __launch_bounds__(1024) void probe(unsigned long long *clocks, unsigned *sink) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b0 = a0 + 11, b1 = a0 ^ 13, c0 = a0 ^ 21, c1 = a0 ^ 55;
  asm volatile("s_mov_b64 s[22:23], 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "s20", "s21", "s22", "s23");
  if (V == 31) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(b0) : "vcc");
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
    switch (V) {
#define X(id, name, body, others, nsel) \
      case id: asm volatile(body body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(c0), "v"(c1) : "vcc", "s20", "s21", "s24", "scc"); break;
      VARIANTS(X)
      VARIANTS2(X)
#undef X
      default: break;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&clocks[blockIdx.x], t1 - t0);
  unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ c0 ^ c1;
  if (r == 0x12345u) sink[0] = r;
}

template <int V>
static void run(const char *name, double others, int nsel, unsigned long long *d_clk, unsigned *d_sink) {
  const int n_wg = 256;
  std::vector<unsigned long long> h(n_wg);
  printf("%-84s", name);
  double at4 = 0;
  for (int threads : {256, 512, 1024}) {
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipMemset(d_clk, 0, n_wg * 8);
      hipLaunchKernelGGL(probe<V>, dim3(n_wg), dim3(threads), 0, 0, d_clk, d_sink);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d_clk, n_wg * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (auto c : h) s += (double)c;
      s /= n_wg;
      if (s < best) best = s;
    }
    const int waves_per_simd = threads / 256;
    const double per_group = best / (double(ITERS) * 2 * waves_per_simd);   // the body is issued twice an iteration
    printf("  %dw: %6.2f", waves_per_simd, per_group);
    at4 = per_group;
  }
  printf("   per select at 4w: %6.2f\n", (at4 - others) / nsel);
}

int main() {
  unsigned long long *d_clk; unsigned *d_sink;
  (void)hipMalloc(&d_clk, 256 * 8); (void)hipMalloc(&d_sink, 4);
  printf("clocks of the SIMD per group (1, 2, 4 waves per SIMD), and per select with the group's other instructions taken off\n");
#define X(id, name, body, others, nsel) run<id>(name, others, nsel, d_clk, d_sink);
  VARIANTS(X)
  printf("-- with the wait states the compiler puts behind a compare (s_nop 1), and how far a compare's grant reaches\n");
  VARIANTS2(X)
#undef X
  return 0;
}
