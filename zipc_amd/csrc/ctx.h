// ctx.h -- the context behind zipc_hip_ctx (host side, internal).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "kernels.h"

struct zipc_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool profiling = false;
  bool adler_rfc1950 = false;  // zipc_hip_set_adler_rfc1950: the zlib forms and checksum_device use RFC 1950's Adler-32
  std::string last_error;
  zd::CrcConsts crc_consts;

  // ---- per-kernel timing (HIP events on `stream`)
  struct Pending { int name_idx; hipEvent_t start, stop; };
  struct Acc { std::string name; uint64_t launches = 0; double total_ms = 0; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  std::vector<Acc> acc;

  // ---- device scratch, grown on demand, reused across calls
  struct Buf {
    void *p = nullptr;
    size_t cap = 0;
  };
  Buf io_src, io_dst, io_desc, io_res, io_small;  // staging of the host forms
  Buf pin_src, pin_dst, pin_res;                  // pinned host memory of the many-stream host forms
  hipStream_t copy_in = nullptr, copy_out = nullptr;  // their H2D / D2H streams (made on first use)
  Buf crc_partials, adler_sums;                   // checksum kernels
  Buf crc_nib;                                    // nibble tables of the CRC merge constants (kernels.h)
  Buf deflate_scratch;                            // deflate pipeline (deflate.hip)
  Buf inflate_scratch;                            // inflate: the span decoder's index, 2304 bytes per stream

  int name_index(const char *name);
  hipEvent_t get_event();
  void begin(const char *name, hipEvent_t &start);
  void end(const char *name, hipEvent_t start);
  hipError_t ensure(Buf &b, size_t bytes);
  hipError_t ensure_pinned(Buf &b, size_t bytes);
  hipError_t collect_times();
};

// Launch `kernel<<<grid, block, lds, ctx->stream>>>(args...)`, bracketed by HIP
// events on the same stream when profiling is on.
#define ZD_LAUNCH(ctx, name, kernel, grid, block, lds, ...)                       \
  do {                                                                            \
    hipEvent_t _zd_start = nullptr;                                               \
    if ((ctx)->profiling) (ctx)->begin(name, _zd_start);                          \
    hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);     \
    if ((ctx)->profiling) (ctx)->end(name, _zd_start);                            \
  } while (0)

namespace zd {
// deflate.hip: bytes of scratch the pipeline needs, and the pipeline itself
size_t deflate_scratch_bytes(size_t n_streams, size_t max_src_len, size_t total_src_len, int level);
hipError_t launch_deflate(zipc_hip_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst,
                          const StreamDesc *d_descs, StreamResult *d_results, size_t n_streams,
                          size_t max_src_len, size_t total_src_len, int level, int crc_op);
}  // namespace zd
