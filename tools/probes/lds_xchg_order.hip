// Probe (gfx950): in what order does ONE ds_wrxchg_rtn_b32 wave instruction serve lanes that hit the same LDS address?
// If it is ascending lane order, a hash-head table can take 64 inserts per instruction and hand every lane the
// previous head or the nearest lower lane with its hash -- exactly insert_hash's order (zd.ml:1150-1152).
// Build: hipcc --offload-arch=gfx950 -O2 lds_xchg_order.hip -o lds_xchg_order ; prints mismatches per pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned *__restrict__ bucket, unsigned *__restrict__ got, int rounds, unsigned *__restrict__ mask_words) {
  __shared__ unsigned head[4096];
  const unsigned lane = threadIdx.x;
  for (unsigned i = lane; i < 4096; i += 64) head[i] = 0xFFFFFFFFu;
  __syncthreads();
  for (int r = 0; r < rounds; r++) {
    const unsigned b = bucket[r * 64 + lane];
    const bool active = ((mask_words[r * 2 + (lane >> 5)] >> (lane & 31)) & 1u) != 0;
    unsigned old = 0xEEEEEEEEu;
    if (active) old = __hip_atomic_exchange(&head[b & 4095u], (unsigned)(r * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    got[r * 64 + lane] = old;
  }
}

int main() {
  const int rounds = 4096;
  std::vector<unsigned> bucket(rounds * 64), mask(rounds * 2), got(rounds * 64), want(rounds * 64);
  srand(12345);
  for (int r = 0; r < rounds; r++) {
    const int kind = r % 8;
    for (int l = 0; l < 64; l++) {
      unsigned b;
      switch (kind) {
        case 0: b = 7; break;                               // all lanes one address
        case 1: b = l & 1; break;                           // two addresses interleaved
        case 2: b = rand() % 8; break;                      // few addresses
        case 3: b = rand() % 64; break;
        case 4: b = rand() % 4096; break;                   // mostly distinct
        case 5: b = (l / 4) * 33; break;                    // groups of neighbours, different banks
        case 6: b = (rand() % 4) * 32; break;               // same bank, different addresses
        default: b = (63 - l) & 15; break;
      }
      bucket[r * 64 + l] = b;
    }
    mask[r * 2] = (r % 3 == 0) ? 0xFFFFFFFFu : (unsigned)rand() * 65537u;
    mask[r * 2 + 1] = (r % 3 == 0) ? 0xFFFFFFFFu : (unsigned)rand() * 65537u;
  }
  // the reference's order: lanes ascending
  std::vector<unsigned> head(4096, 0xFFFFFFFFu);
  for (int r = 0; r < rounds; r++)
    for (int l = 0; l < 64; l++) {
      const bool active = (mask[r * 2 + (l >> 5)] >> (l & 31)) & 1u;
      if (!active) { want[r * 64 + l] = 0xEEEEEEEEu; continue; }
      unsigned &h = head[bucket[r * 64 + l] & 4095u];
      want[r * 64 + l] = h;
      h = r * 64 + l;
    }
  unsigned *d_b, *d_g, *d_m;
  hipMalloc(&d_b, bucket.size() * 4); hipMalloc(&d_g, got.size() * 4); hipMalloc(&d_m, mask.size() * 4);
  hipMemcpy(d_b, bucket.data(), bucket.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_m, mask.data(), mask.size() * 4, hipMemcpyHostToDevice);
  long bad_total = 0;
  for (int rep = 0; rep < 20; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_b, d_g, rounds, d_m);
    hipMemcpy(got.data(), d_g, got.size() * 4, hipMemcpyDeviceToHost);
    long bad[8] = {0};
    for (int r = 0; r < rounds; r++)
      for (int l = 0; l < 64; l++)
        if (got[r * 64 + l] != want[r * 64 + l]) bad[r % 8]++;
    long s = 0;
    for (int k = 0; k < 8; k++) s += bad[k];
    bad_total += s;
    if (rep == 0 || s) printf("rep %d mismatches by pattern: %ld %ld %ld %ld %ld %ld %ld %ld\n", rep, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7]);
  }
  printf("LDS_XCHG_ORDER %s (%ld mismatches over 20 launches x %d rounds)\n", bad_total ? "NOT ascending-lane" : "ascending-lane", bad_total, rounds);
  return 0;
}
