// ctx.h -- the context behind zipc_hip_ctx (host side, internal).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "kernels.h"

struct zipc_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // The stream ZD_LAUNCH enqueues on: `stream`, or one of `side` while a batch call has its
  // slices in flight (fork()/join() below).
  hipStream_t cur = nullptr;
  // Side streams of the batch forms (ZIPC_HIP_SLICES > 1, not the default: deflate.hip has the
  // measurement).  A large batch is cut into slices of consecutive streams and every slice's kernels
  // go to a queue of their own.  Work is still ordered behind everything enqueued on `stream` before
  // the call (fork) and everything enqueued on `stream` after the call is ordered behind the slices (join).
  std::vector<hipStream_t> side;
  std::vector<hipEvent_t> side_done;
  hipEvent_t fork_ev = nullptr;
  bool profiling = false;
  bool xchg_ordered = false;   // deflate.hip xchg_order_probe: one LDS exchange serves lanes on one address in ascending lane order
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
  Buf io_pack_off;                                // ... where each output of a sub-batch begins once they lie end to end (api.hip pack_offsets_kernel)
  hipStream_t copy_in = nullptr, copy_out = nullptr;  // their H2D / D2H queues (made on first use)
  Buf crc_partials, adler_sums;                   // checksum kernels
  Buf crc_nib;                                    // nibble tables of the CRC merge constants (kernels.h)
  Buf deflate_scratch;                            // deflate pipeline (deflate.hip)
  Buf parse_scratch;                              // lz_parse by segments: their symbols, paths and verdicts (deflate.hip ParseSegs)
  Buf inflate_scratch;                            // inflate: the span decoder's index, 2304 bytes per stream
  Buf stored_list;                                // inflate of one stream beyond 4 GiB: the stored blocks a walk listed
  Buf blocks_scratch, tok_scratch;                // inflate of streams by a wave per block: candidates, chains; a word per output byte
  Buf descs_marked;                               // ... a call's descriptors with those streams marked as done, for the one waves of the rest
  Buf chain_check_links;                          // lz_chain's run-time check: the links the exchange kernel made of a batch's first streams
  unsigned long long *chain_check_host = nullptr; // ... [0] differences, [1] positions compared in the context's first batch; [2], [3]: the same for the create-time probe (device-visible host memory)
  bool chain_checked = false;                     // ... that first batch has been seen
  uint32_t last_inflate_blocks = 0;               // blocks of the last one-stream inflate that went that way (0: it did not)

  int name_index(const char *name);
  hipEvent_t get_event();
  void begin(const char *name, hipEvent_t &start);
  void end(const char *name, hipEvent_t start);
  hipError_t ensure(Buf &b, size_t bytes);
  hipError_t ensure_pinned(Buf &b, size_t bytes);
  hipError_t collect_times();
  // side streams [0, k) wait for what `stream` holds now / `stream` waits for side streams [0, k)
  // (join also points `cur` back at `stream`; every fork is followed by a join on every path)
  hipError_t fork(size_t k);
  hipError_t join(size_t k);
  // slice i of a batch call goes to: `stream` for i = 0, side[i] above
  void use_slice_stream(size_t i) { cur = i == 0 ? stream : side[i]; }
};

// Launch `kernel<<<grid, block, lds, ctx->stream>>>(args...)`, bracketed by HIP
// events on the same stream when profiling is on.
#define ZD_LAUNCH(ctx, name, kernel, grid, block, lds, ...)                       \
  do {                                                                            \
    hipEvent_t _zd_start = nullptr;                                               \
    if ((ctx)->profiling) (ctx)->begin(name, _zd_start);                          \
    hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->cur, __VA_ARGS__);        \
    if ((ctx)->profiling) (ctx)->end(name, _zd_start);                            \
  } while (0)

namespace zd {
// api.hip: into how many slices of consecutive streams a batch of n is cut (1: no side streams).
// ZIPC_HIP_SLICES overrides the default; a slice holds at least 2048 streams.
size_t batch_slices(size_t n_streams);
// api.hip: the two halves of the CRC-32 pass over n_ranges ranges of up to max_len bytes, on ctx->cur.
// partials: n_ranges * crc32_segs(max_len) words.
size_t crc32_segs(size_t max_len);
hipError_t crc32_segments_launch(zipc_hip_ctx *ctx, const uint8_t *base, int mode, const StreamDesc *d_descs,
                                 const StreamResult *d_results, size_t n_ranges, uint64_t single_off,
                                 uint64_t single_len, size_t max_len, uint32_t *partials);
hipError_t crc32_finish_launch(zipc_hip_ctx *ctx, int mode, const StreamDesc *d_descs, StreamResult *d_results,
                               size_t n_ranges, uint64_t single_len, size_t max_len, const uint32_t *partials,
                               uint32_t *d_single_out);
// deflate.hip: the probe behind ctx->xchg_ordered
bool xchg_order_probe(zipc_hip_ctx *ctx);
// deflate.hip: bytes of scratch the pipeline needs, and the pipeline itself
size_t deflate_scratch_bytes(size_t n_streams, size_t max_src_len, size_t total_src_len, int level);
// deflate.hip (tests): the links one of the two chain kernels makes of a batch, and how many link slots that takes
hipError_t debug_chain_links(zipc_hip_ctx *ctx, const uint8_t *d_src, const StreamDesc *d_descs, size_t n, size_t max_src_len,
                             size_t total_src_len, int which, uint16_t *d_links, size_t links_cap, uint64_t *d_pos_base);
size_t debug_chain_positions(size_t n, size_t total_src_len);
hipError_t launch_deflate(zipc_hip_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst,
                          const StreamDesc *d_descs, StreamResult *d_results, size_t n_streams,
                          size_t max_src_len, size_t total_src_len, int level, int crc_op);
}  // namespace zd
