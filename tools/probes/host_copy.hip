// How fast do bytes cross the bus when a KERNEL moves them (stores into / loads from pinned host memory) beside the
// copy engines (hipMemcpyAsync), what do both directions at once cost each other, and what does a kernel that works on
// device memory lose meanwhile?  Behind api.hip many_streams: results come back by a kernel that writes the pinned
// buffer.  What this prints on the pool's MI355X boxes (profiles/r05_host_copy.txt):
//  * alone, every way of moving 256 MiB takes 4.7-4.9 ms (55-57 GB/s): engine or kernel, in or out, 64 to 4096 workgroups;
//  * an engine copy out beside an engine copy in: 13.4 / 14.1 ms -- unless ANY wave is resident (a kernel spinning on
//    registers, or one wave that sleeps: 4.8 / 5.5 ms).  (A wave kept resident for the length of a call is not a way
//    out: a kernel that waits for the host holds up whatever other queue of the process shares its hardware queue.)
//  * a kernel's stores out beside an engine copy in: 5.3 / 6.3 ms; a kernel's loads in beside an engine copy out: 4.7 / 9.0;
//  * kernels that copy device memory take 1.5 ms alone, 1.55 beside engine copies, 2.9 beside 8 workgroups storing to the
//    host, 5.9 beside 64 of them, 5.7 beside a kernel loading from the host.
//  * the kind of pinned memory (default, coherent, non-coherent) changes none of it.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/host_copy.hip -o /tmp/host_copy && /tmp/host_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(u32x4 *dst, const u32x4 *src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) if (i + k * 256 < n16) v[k] = src[i + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) if (i + k * 256 < n16) {
      if (NT) __builtin_nontemporal_store(v[k], dst + i + k * 256); else dst[i + k * 256] = v[k];
    }
  }
}
// one wave that mostly sleeps
__global__ void doze_kernel(unsigned long long clocks, unsigned *out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < clocks) __builtin_amdgcn_s_sleep(127);
  if (clocks == 1) *out = 1;
}
__global__ void spin_kernel(unsigned long long clocks, unsigned *out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned x = threadIdx.x;
  while (__builtin_readcyclecounter() - t0 < clocks) x = x * 1664525u + 1013904223u;
  if (x == 0x12345u) *out = x;
}
int main() {
  const size_t bytes = (size_t)256 << 20, n16 = bytes / 16;
  void *dev, *dev2, *pin; unsigned *flag;
  CK(hipMalloc(&dev, bytes)); CK(hipMalloc(&dev2, bytes)); CK(hipMalloc((void **)&flag, 4));
  CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
  memset(pin, 1, bytes); CK(hipMemset(dev, 2, bytes));
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto report = [&](const char *what, float ms) { printf("%-72s %7.3f ms  %6.1f GB/s\n", what, ms, bytes / ms / 1e6); };
#define TIMED(what, stmt) do { float best = 1e9; for (int r = 0; r < 5; r++) { CK(hipEventRecord(a, s)); stmt; CK(hipEventRecord(b, s)); \
    CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; } report(what, best); } while (0)
  TIMED("hipMemcpyAsync device -> pinned host", CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s)));
  TIMED("hipMemcpyAsync pinned host -> device", CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)));
  TIMED("hipMemcpyAsync device -> device", CK(hipMemcpyAsync(dev2, dev, bytes, hipMemcpyDeviceToDevice, s)));
  char name[128];
#define KROW(NT, U, wgs, D, S, dir) do { snprintf(name, sizeof name, "kernel %s, %d workgroups, %d x 16 B a thread%s", dir, wgs, U, NT ? ", nontemporal stores" : ""); \
    TIMED(name, hipLaunchKernelGGL((copy_kernel<NT, U>), dim3(wgs), dim3(256), 0, s, (u32x4 *)(D), (const u32x4 *)(S), n16)); } while (0)
  for (int wgs : {64, 256, 1024, 4096}) {
    KROW(false, 1, wgs, pin, dev, "device -> host");
    KROW(true, 1, wgs, pin, dev, "device -> host");
    KROW(false, 4, wgs, pin, dev, "device -> host");
    KROW(true, 4, wgs, pin, dev, "device -> host");
  }
  for (int wgs : {256, 1024, 4096}) {
    KROW(false, 1, wgs, dev, pin, "host -> device");
    KROW(false, 4, wgs, dev, pin, "host -> device");
  }
  // the copy engine after the device has been idle, and with a kernel spinning on another queue
  for (int idle_ms : {0, 5, 20, 100}) {
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < idle_ms) {}
    CK(hipEventRecord(a, s));
    for (int k = 0; k < 4; k++) CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s));
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    snprintf(name, sizeof name, "4 x hipMemcpyAsync device -> host after %d ms of idle (per copy)", idle_ms);
    report(name, ms / 4);
  }
  for (int rep = 0; rep < 2; rep++) {
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, s2, 200000000ull, flag);  // ~100 ms
    CK(hipEventRecord(a, s));
    for (int k = 0; k < 4; k++) CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s));
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    report("4 x hipMemcpyAsync device -> host beside a spinning kernel (per copy)", ms / 4);
    CK(hipDeviceSynchronize());
  }
  // both directions at once, each side's own time; and what a kernel that works on device memory (or on registers only)
  // loses while bytes cross the bus beside it
  hipStream_t s3; CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  hipEvent_t a2, b2, a3, b3; CK(hipEventCreate(&a2)); CK(hipEventCreate(&b2)); CK(hipEventCreate(&a3)); CK(hipEventCreate(&b3));
  void *dev3, *dev4; CK(hipMalloc(&dev3, bytes)); CK(hipMalloc(&dev4, bytes));
  auto work = [&](int kind) {  // on s3: 0 nothing, 1 registers only (~5 ms), 2 device memory (16 x 256 MiB copied)
    if (kind == 1) hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(256), 0, s3, 10000000ull, flag);
    if (kind == 3) hipLaunchKernelGGL(doze_kernel, dim3(1), dim3(64), 0, s3, 40000000ull, flag);  // ~20 ms
    if (kind == 4) hipLaunchKernelGGL(doze_kernel, dim3(256), dim3(64), 0, s3, 40000000ull, flag);
    if (kind == 2) for (int k = 0; k < 16; k++) hipLaunchKernelGGL((copy_kernel<false, 4>), dim3(2048), dim3(256), 0, s3, (u32x4 *)dev4, (const u32x4 *)dev3, n16);
  };
  const char *work_name[5] = {"", " + a kernel on registers", " + kernels on device memory", " + one wave that dozes", " + 256 waves that doze"};
  for (int kind = 0; kind < 5; kind++)
    for (int in_mode = 0; in_mode < 3; in_mode++)      // 0 nothing comes in, 1 engine, 2 kernel (64 workgroups)
      for (int out_mode = 0; out_mode < 4; out_mode++) {  // 0 nothing goes out, 1 engine, 2 kernel (64 workgroups), 3 kernel (8)
        if (kind == 0 && (in_mode == 0 || out_mode == 0)) continue;
        if (kind >= 3 && !(in_mode == 1 && out_mode == 1)) continue;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, s)); CK(hipEventRecord(a2, s2)); CK(hipEventRecord(a3, s3));
        for (int k = 0; k < 2; k++) {
          if (in_mode == 1) CK(hipMemcpyAsync(dev2, pin, bytes / 2, hipMemcpyHostToDevice, s2));
          if (in_mode == 2) hipLaunchKernelGGL((copy_kernel<false, 4>), dim3(64), dim3(256), 0, s2, (u32x4 *)dev2, (const u32x4 *)pin, n16 / 2);
          if (out_mode == 1) CK(hipMemcpyAsync((char *)pin + bytes / 2, dev, bytes / 2, hipMemcpyDeviceToHost, s));
          if (out_mode >= 2) hipLaunchKernelGGL((copy_kernel<false, 4>), dim3(out_mode == 2 ? 64 : 8), dim3(256), 0, s, (u32x4 *)((char *)pin + bytes / 2), (const u32x4 *)dev, n16 / 2);
        }
        work(kind);
        CK(hipEventRecord(b, s)); CK(hipEventRecord(b2, s2)); CK(hipEventRecord(b3, s3));
        CK(hipDeviceSynchronize());
        float ms_out, ms_in, ms_w;
        CK(hipEventElapsedTime(&ms_out, a, b)); CK(hipEventElapsedTime(&ms_in, a2, b2)); CK(hipEventElapsedTime(&ms_w, a3, b3));
        const char *im[3] = {"-", "engine", "kernel"}, *om[4] = {"-", "engine", "kernel 64", "kernel 8"};
        printf("in: %-7s out: %-10s%-28s 256 MiB in %6.2f ms, 256 MiB out %6.2f ms, the work %6.2f ms\n", im[in_mode], om[out_mode],
               work_name[kind], in_mode ? ms_in : 0.f, out_mode ? ms_out : 0.f, kind ? ms_w : 0.f);
      }
  // the kind of pinned memory: does a kernel's way out disturb less when the host buffer may be cached by the device?
  {
    const unsigned flags[3] = {hipHostMallocDefault, hipHostMallocCoherent, hipHostMallocNonCoherent};
    const char *fname[3] = {"hipHostMallocDefault", "hipHostMallocCoherent", "hipHostMallocNonCoherent"};
    for (int f = 0; f < 3; f++) {
      void *p2;
      CK(hipHostMalloc(&p2, bytes, flags[f]));
      memset(p2, 3, bytes);
      for (int wgs : {8, 64}) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, s)); CK(hipEventRecord(a2, s2)); CK(hipEventRecord(a3, s3));
        CK(hipMemcpyAsync(dev2, pin, bytes, hipMemcpyHostToDevice, s2));
        hipLaunchKernelGGL((copy_kernel<false, 4>), dim3(wgs), dim3(256), 0, s, (u32x4 *)p2, (const u32x4 *)dev, n16);
        work(2);
        CK(hipEventRecord(b, s)); CK(hipEventRecord(b2, s2)); CK(hipEventRecord(b3, s3));
        CK(hipDeviceSynchronize());
        float ms_out, ms_in, ms_w;
        CK(hipEventElapsedTime(&ms_out, a, b)); CK(hipEventElapsedTime(&ms_in, a2, b2)); CK(hipEventElapsedTime(&ms_w, a3, b3));
        printf("%-26s engine in %6.2f ms, kernel (%2d workgroups) out %6.2f ms, kernels on device memory %6.2f ms\n", fname[f], ms_in, wgs,
               ms_out, ms_w);
      }
      CK(hipHostFree(p2));
    }
  }
  return 0;
}
