// api.hip -- the C ABI of include/zipc_hip.h: context, scratch, kernel launches.
//
// Host forms stage one stream through device scratch and run the same kernels as
// the batch forms (a batch of one).  Nothing here computes on the CPU: with no
// usable device the calls fail with ZIPC_HIP_ERR_NO_DEVICE / ZIPC_HIP_ERR_HIP.
#include "../../include/zipc_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "ctx.h"
#include "host_pipeline.h"
#include "tuning.h"

using namespace zd;

static_assert(sizeof(zipc_hip_stream_desc) == sizeof(StreamDesc), "desc layout");
static_assert(sizeof(zipc_hip_stream_result) == sizeof(StreamResult), "result layout");

#define HIP_TRY(ctx, expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);           \
      return ZIPC_HIP_ERR_HIP;                                                         \
    }                                                                                  \
  } while (0)

// ---- context internals -------------------------------------------------------

int zipc_hip_ctx::name_index(const char *name) {
  for (size_t i = 0; i < acc.size(); i++)
    if (acc[i].name == name) return (int)i;
  Acc a;
  a.name = name;
  acc.push_back(a);
  return (int)acc.size() - 1;
}

hipEvent_t zipc_hip_ctx::get_event() {
  if (!event_pool.empty()) {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void zipc_hip_ctx::begin(const char *, hipEvent_t &start) {
  start = get_event();
  (void)hipEventRecord(start, cur);
}

hipError_t zipc_hip_ctx::fork(size_t k) {
  while (side.size() < k) {
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    hipError_t r = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (r != hipSuccess) return r;
    r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (r != hipSuccess) { (void)hipStreamDestroy(s); return r; }
    side.push_back(s);
    side_done.push_back(e);
  }
  if (!fork_ev) {
    const hipError_t r = hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming);
    if (r != hipSuccess) return r;
  }
  hipError_t r = hipEventRecord(fork_ev, stream);
  for (size_t i = 0; i < k && r == hipSuccess; i++) r = hipStreamWaitEvent(side[i], fork_ev, 0);
  return r;
}

hipError_t zipc_hip_ctx::join(size_t k) {
  hipError_t r = hipSuccess;
  for (size_t i = 0; i < k && r == hipSuccess; i++) {
    r = hipEventRecord(side_done[i], side[i]);
    if (r == hipSuccess) r = hipStreamWaitEvent(stream, side_done[i], 0);
  }
  cur = stream;
  return r;
}

void zipc_hip_ctx::end(const char *name, hipEvent_t start) {
  Pending p;
  p.name_idx = name_index(name);
  p.start = start;
  p.stop = get_event();
  (void)hipEventRecord(p.stop, cur);
  pending.push_back(p);
}

hipError_t zipc_hip_ctx::ensure_pinned(Buf &b, size_t bytes) {
  if (bytes <= b.cap && b.p) return hipSuccess;
  if (b.p) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    (void)hipHostFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes < 4096 ? 4096 : bytes;
  hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
  if (e != hipSuccess) { b.p = nullptr; return e; }
  b.cap = want;
  return hipSuccess;
}

hipError_t zipc_hip_ctx::ensure(Buf &b, size_t bytes) {
  if (bytes <= b.cap && b.p) return hipSuccess;
  if (b.p) {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes < 256 ? 256 : bytes;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { b.p = nullptr; return e; }
  b.cap = want;
  return hipSuccess;
}

hipError_t zipc_hip_ctx::collect_times() {
  if (pending.empty()) return hipSuccess;
  hipError_t e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  for (auto &p : pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      acc[p.name_idx].launches++;
      acc[p.name_idx].total_ms += ms;
    }
    event_pool.push_back(p.start);
    event_pool.push_back(p.stop);
  }
  pending.clear();
  return hipSuccess;
}

static void free_buf(zipc_hip_ctx::Buf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// ---- context API -------------------------------------------------------------

// host-side loop over streams [lo, hi) of a batch on a few threads (memcpy bound).
// ZIPC_HIP_HOST_THREADS overrides the count (default: 8 or the core count, if lower).
static size_t host_threads() {
  static const size_t nt = [] {
    long v = zd::tuning().host_threads;
    if (v < 1) {
      const unsigned hw = std::thread::hardware_concurrency();
      v = hw >= 8 ? 8 : (hw ? hw : 1);
    }
    return (size_t)(v > 64 ? 64 : v);
  }();
  return nt;
}
// ZIPC_HIP_HOST_CHUNKS: sub-batches a many-stream call is cut into; each goes through gather, copy in, kernels, the
// way back and scatter on its own, so those overlap (1 = one after the other).  Default: by the bytes staged (many_streams).
static size_t host_chunks(uint64_t staged_bytes) {
  long v = zd::tuning().host_chunks;
  if (v < 1) v = staged_bytes >= ((uint64_t)1 << 30) ? 6 : 4;
  return (size_t)(v > 64 ? 64 : v);
}
// The threads behind the host memcpys of the many-stream forms, the copies that go around the cache and the pipeline of a
// call's sub-batches live in host_pipeline.h (no HIP in it: tests/host_sim compiles the same code under the thread and
// address sanitizers with host threads standing in for the device).  The pools are shared by the process's contexts,
// made on first use, and their threads are joined when the last context is destroyed.
static zd_host::Pools &host_pools() {
  static zd_host::Pools *const p = new zd_host::Pools;  // (the object outlives every context; its threads do not)
  return *p;
}
// events of one call, destroyed on every exit path
struct EventSet {
  std::vector<hipEvent_t> ev;
  ~EventSet() { for (auto e : ev) (void)hipEventDestroy(e); }
  hipError_t make(size_t k, bool timed = false) {
    for (size_t i = 0; i < k; i++) {
      hipEvent_t e;
      hipError_t r = hipEventCreateWithFlags(&e, timed ? hipEventDefault : hipEventDisableTiming);
      if (r != hipSuccess) return r;
      ev.push_back(e);
    }
    return hipSuccess;
  }
};

namespace zd {
const Tuning &tuning() {
  static const Tuning t = [] {
    auto num = [](const char *name, long dflt) { const char *e = getenv(name); return e ? atol(e) : dflt; };
    auto is = [](const char *name, const char *v) { const char *e = getenv(name); return e && !strcmp(e, v); };
    Tuning x;
    x.chain_peel = is("ZIPC_HIP_CHAIN", "peel");
    x.chain_check = num("ZIPC_HIP_CHAIN_CHECK", 32);
    x.parse_segments = num("ZIPC_HIP_PARSE_SEGMENTS", -1);
    x.parse_seg = num("ZIPC_HIP_PARSE_SEG", 0);
    x.match_tiles_per_group = num("ZIPC_HIP_MATCH_TILES_PER_GROUP", 0);
    const long form = num("ZIPC_HIP_MATCH_FORM", 0);
    x.match_form = form == 1 || form == 2 ? (int)form : 0;
    const long long group = getenv("ZIPC_HIP_DEFLATE_GROUP_BYTES") ? atoll(getenv("ZIPC_HIP_DEFLATE_GROUP_BYTES")) : 0;
    x.deflate_group_bytes = group > 0 ? (size_t)group : (size_t)8 << 30;
    x.slices = num("ZIPC_HIP_SLICES", 0);
    x.slice_min = num("ZIPC_HIP_SLICE_MIN", 0);
    x.inflate_blocks = num("ZIPC_HIP_INFLATE_BLOCKS", 1) != 0;
    x.inflate_follow = (int)num("ZIPC_HIP_INFLATE_FOLLOW", -1);
    x.explore_stride = (uint64_t)num("ZIPC_HIP_EXPLORE_STRIDE", 16384);
    if (x.explore_stride < 1024) x.explore_stride = 1024;  // (a divisor: never 0 or negative, whatever the environment says)
    x.resolve_hops0 = (int)num("ZIPC_HIP_RESOLVE_HOPS0", 256);
    x.resolve_hops1 = (int)num("ZIPC_HIP_RESOLVE_HOPS1", 256);
    x.checksum_fused = num("ZIPC_HIP_CHECKSUM_FUSED", 1) != 0;
    x.host_threads = num("ZIPC_HIP_HOST_THREADS", 0);
    x.host_chunks = num("ZIPC_HIP_HOST_CHUNKS", 0);
    x.host_chunk_min = num("ZIPC_HIP_HOST_CHUNK_MIN", 1024);
    if (x.host_chunk_min < 1) x.host_chunk_min = 1;
    x.host_pack = num("ZIPC_HIP_HOST_PACK", 1) != 0;
    x.host_pack_wgs = num("ZIPC_HIP_HOST_PACK_WGS", 6);
    if (x.host_pack_wgs < 1) x.host_pack_wgs = 1;
    x.host_h2d_mib = num("ZIPC_HIP_HOST_H2D_MIB", 16);
    x.host_timing = num("ZIPC_HIP_HOST_TIMING", 0) != 0;
    return x;
  }();
  return t;
}
static long g_slices_override = 0;  // zipc_hip_debug_set_slices: measurements that want every kernel alone on the device
size_t batch_slices(size_t n_streams) {
  // Two slices by default (each of at least 2048 streams): since lz_chain is four waves per CU (round 4) the second
  // slice's chain links are made beside the first one's parse and blocks -- C2 deflate 13.25 -> 12.67 ms, the step
  // 16.97 -> 16.34; text the same either way; 3 / 4 / 6 slices lose 7 / 3 / 8 % (one box, tools/exp_wall.py).
  const long env = g_slices_override > 0 ? g_slices_override : tuning().slices, env_min = tuning().slice_min;
  size_t k = env > 0 ? (size_t)env : 2;
  if (k > 8) k = 8;
  const size_t least = env_min > 0 ? (size_t)env_min : 2048;
  while (k > 1 && n_streams / k < least) k--;
  return k;
}
size_t crc32_segs(size_t max_len) {
  const size_t segs = (max_len + CRC_SEG_BYTES - 1) / CRC_SEG_BYTES;
  return segs ? segs : 1;
}
hipError_t crc32_segments_launch(zipc_hip_ctx *ctx, const uint8_t *base, int mode, const StreamDesc *d_descs,
                                 const StreamResult *d_results, size_t n_ranges, uint64_t single_off,
                                 uint64_t single_len, size_t max_len, uint32_t *partials) {
  const size_t segs = crc32_segs(max_len);
  if (n_ranges * segs > 0x7FFFFFFFull) return hipErrorInvalidValue;
  ZD_LAUNCH(ctx, "crc32_segments", crc32_segments_kernel, dim3((unsigned)(n_ranges * segs)), dim3(256), 0,
            base, mode, d_descs, d_results, single_off, single_len, (uint32_t)segs, (const uint32_t *)ctx->crc_nib.p,
            partials);
  return hipGetLastError();
}
hipError_t crc32_finish_launch(zipc_hip_ctx *ctx, int mode, const StreamDesc *d_descs, StreamResult *d_results,
                               size_t n_ranges, uint64_t single_len, size_t max_len, const uint32_t *partials,
                               uint32_t *d_single_out) {
  const size_t segs = crc32_segs(max_len);
  if (mode != RANGE_SINGLE && segs <= 16)  // a batch of short streams: one per thread
    ZD_LAUNCH(ctx, "crc32_finish", crc32_finish_streams_kernel, dim3((unsigned)((n_ranges + 255) / 256)), dim3(256), 0,
              mode, d_descs, d_results, (uint32_t)n_ranges, (uint32_t)segs, ctx->crc_consts, partials);
  else
    ZD_LAUNCH(ctx, "crc32_finish", crc32_finish_kernel, dim3((unsigned)n_ranges), dim3(segs > 4096 ? 1024 : 256), 0, mode,
              d_descs, d_results, single_len, (uint32_t)segs, ctx->crc_consts, (const uint32_t *)ctx->crc_nib.p,
              partials, d_single_out);
  return hipGetLastError();
}
}  // namespace zd

// the whole CRC-32 pass on ctx->cur; partials: the context's buffer from word `partials_at` on
static int crc32_pass(zipc_hip_ctx *ctx, const uint8_t *base, int mode, const StreamDesc *d_descs,
                      StreamResult *d_results, size_t n_ranges, uint64_t single_off,
                      uint64_t single_len, size_t max_len, uint32_t *d_single_out, size_t partials_at = 0,
                      bool ensured = false) {
  const size_t segs = crc32_segs(max_len);
  if (n_ranges * segs > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
  if (!ensured) HIP_TRY(ctx, ctx->ensure(ctx->crc_partials, (partials_at + n_ranges * segs) * sizeof(uint32_t)));
  uint32_t *partials = (uint32_t *)ctx->crc_partials.p + partials_at;
  HIP_TRY(ctx, crc32_segments_launch(ctx, base, mode, d_descs, (const StreamResult *)d_results, n_ranges, single_off,
                                     single_len, max_len, partials));
  HIP_TRY(ctx, crc32_finish_launch(ctx, mode, d_descs, d_results, n_ranges, single_len, max_len, partials, d_single_out));
  return ZIPC_HIP_OK;
}

extern "C" {

int zipc_hip_abi_version(void) { return ZIPC_HIP_ABI_VERSION; }

int zipc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *zipc_hip_strerror(int status) {
  switch (status) {
  case ZIPC_HIP_OK: return "";
  case ZIPC_HIP_ERR_CORRUPTED: return "Corrupted data stream";
  case ZIPC_HIP_ERR_SIZE_EXCEEDED: return "Expected decompression size exceeded";
  case ZIPC_HIP_ERR_ZLIB_METHOD: return "Unknown compression method (%d)";
  case ZIPC_HIP_ERR_ZLIB_WINDOW: return "Window size too large";
  case ZIPC_HIP_ERR_ZLIB_DICT: return "Preset dictionary unsupported";
  case ZIPC_HIP_ERR_CHECKSUM: return "Checksum mismatch, expected %lx found %lx)";
  case ZIPC_HIP_ERR_DST_TOO_SMALL: return "destination buffer too small";
  case ZIPC_HIP_ERR_HIP: return "HIP runtime error";
  case ZIPC_HIP_ERR_INVALID_ARG: return "invalid argument";
  case ZIPC_HIP_ERR_NO_DEVICE: return "no usable HIP device";
  case ZIPC_HIP_ERR_NOMEM: return "out of memory";
  default: return "unknown status";
  }
}

int zipc_hip_create(zipc_hip_ctx **out, int device) {
  if (!out) return ZIPC_HIP_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ZIPC_HIP_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return ZIPC_HIP_ERR_INVALID_ARG;
  if (hipSetDevice(device) != hipSuccess) return ZIPC_HIP_ERR_HIP;
  zipc_hip_ctx *ctx = new (std::nothrow) zipc_hip_ctx();
  if (!ctx) return ZIPC_HIP_ERR_NOMEM;
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return ZIPC_HIP_ERR_HIP;
  }
  ctx->cur = ctx->stream;
  host_pools().acquire();  // (released by zipc_hip_destroy: the last context to go joins the staging threads)
  // CRC merge constants (zd_common.h), computed with the same GF(2) routines the
  // kernels use
  uint32_t x = gf2_xpow8n(CRC_PIECE_BYTES);
  for (int k = 0; k < 8; k++) { ctx->crc_consts.xpiece[k] = x; x = gf2_mul(x, x); }
  ctx->crc_consts.xseg = gf2_xpow8n(CRC_SEG_BYTES);
  x = 0x00800000u;  // x^8
  for (int k = 0; k < 48; k++) { ctx->crc_consts.xbyte[k] = x; x = gf2_mul(x, x); }
  {
    std::vector<uint32_t> nib((size_t)CRC_NIB_CONSTS * GF2_NIB_WORDS);
    for (int k = 0; k < 8; k++) gf2_nib_table(ctx->crc_consts.xpiece[k], nib.data() + (size_t)k * GF2_NIB_WORDS);
    gf2_nib_table(ctx->crc_consts.xseg, nib.data() + (size_t)CRC_NIB_XSEG * GF2_NIB_WORDS);
    if (ctx->ensure(ctx->crc_nib, nib.size() * sizeof(uint32_t)) != hipSuccess ||
        hipMemcpy(ctx->crc_nib.p, nib.data(), nib.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
      zipc_hip_destroy(ctx);
      return ZIPC_HIP_ERR_HIP;
    }
  }
  ctx->xchg_ordered = zd::xchg_order_probe(ctx);
  *out = ctx;
  return ZIPC_HIP_OK;
}

void zipc_hip_destroy(zipc_hip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &p : ctx->pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  free_buf(ctx->io_src); free_buf(ctx->io_dst); free_buf(ctx->io_desc); free_buf(ctx->io_res);
  free_buf(ctx->io_pack_off);
  free_buf(ctx->io_small); free_buf(ctx->crc_partials); free_buf(ctx->crc_nib); free_buf(ctx->adler_sums);
  free_buf(ctx->deflate_scratch); free_buf(ctx->parse_scratch);
  free_buf(ctx->inflate_scratch);
  free_buf(ctx->blocks_scratch);
  free_buf(ctx->tok_scratch);
  free_buf(ctx->descs_marked);
  free_buf(ctx->stored_list);
  free_buf(ctx->chain_check_links);
  if (ctx->chain_check_host) (void)hipHostFree(ctx->chain_check_host);
  if (ctx->pin_src.p) (void)hipHostFree(ctx->pin_src.p);
  if (ctx->pin_dst.p) (void)hipHostFree(ctx->pin_dst.p);
  if (ctx->pin_res.p) (void)hipHostFree(ctx->pin_res.p);
  if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
  if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
  for (auto s : ctx->side) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  for (auto e : ctx->side_done) (void)hipEventDestroy(e);
  if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  host_pools().release();
}

void *zipc_hip_stream(zipc_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int zipc_hip_synchronize(zipc_hip_ctx *ctx) {
  if (!ctx) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ZIPC_HIP_OK;
}

const char *zipc_hip_last_error(zipc_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
unsigned zipc_hip_last_inflate_blocks(zipc_hip_ctx *ctx) { return ctx ? ctx->last_inflate_blocks : 0u; }
int zipc_hip_lds_exchange_ordered(zipc_hip_ctx *ctx) { return ctx && ctx->xchg_ordered ? 1 : 0; }
int zipc_hip_chain_check(zipc_hip_ctx *ctx, unsigned long long *compared, unsigned long long *differences) {
  if (!ctx || !compared || !differences) return ZIPC_HIP_ERR_INVALID_ARG;
  *compared = *differences = 0;
  if (!ctx->chain_check_host) return ZIPC_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *compared = ctx->chain_check_host[1] + ctx->chain_check_host[3];
  *differences = ctx->chain_check_host[0] + ctx->chain_check_host[2];
  return ZIPC_HIP_OK;
}
void zipc_hip_debug_set_slices(long k) { zd::g_slices_override = k; }

int zipc_hip_set_adler_rfc1950(zipc_hip_ctx *ctx, int enabled) {
  if (!ctx) return ZIPC_HIP_ERR_INVALID_ARG;
  ctx->adler_rfc1950 = enabled != 0;
  return ZIPC_HIP_OK;
}

int zipc_hip_set_profiling(zipc_hip_ctx *ctx, int enabled) {
  if (!ctx) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, ctx->collect_times());
  ctx->profiling = enabled != 0;
  return ZIPC_HIP_OK;
}

int zipc_hip_reset_kernel_times(zipc_hip_ctx *ctx) {
  if (!ctx) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, ctx->collect_times());
  for (auto &a : ctx->acc) { a.launches = 0; a.total_ms = 0; }
  return ZIPC_HIP_OK;
}

int zipc_hip_kernel_times(zipc_hip_ctx *ctx, zipc_hip_kernel_time *out, size_t cap, size_t *n) {
  if (!ctx || !n) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, ctx->collect_times());
  size_t k = 0;
  for (auto &a : ctx->acc) {
    if (a.launches == 0) continue;
    if (out && k < cap) {
      memset(&out[k], 0, sizeof out[k]);
      snprintf(out[k].name, sizeof out[k].name, "%s", a.name.c_str());
      out[k].launches = a.launches;
      out[k].total_ms = a.total_ms;
    }
    k++;
  }
  *n = k;
  return ZIPC_HIP_OK;
}

size_t zipc_hip_deflate_bound(size_t len) {
  // all-stored worst case: 5 header bytes per <= 65534 source bytes, +1 per block
  // because the reference's stored-block estimate can be 8 bits high
  // (src/zipc_deflate.ml:1045-1047), so a compressed block may beat it by < 1 byte
  size_t blocks = len / 65534 + 1;
  return len + 6 * blocks + 8;
}
size_t zipc_hip_zlib_bound(size_t len) { return zipc_hip_deflate_bound(len) + 6; }

// ---- batch forms ---------------------------------------------------------------

// One stream beyond ZIPC_HIP_MAX_STREAM_LEN: the stored blocks it has to start with (inflate.hip) are found
// and copied with 64-bit offsets -- a chain of equal blocks all at once, blocks of other lengths by a walk over
// their headers -- and what follows, if anything, goes through the batch kernel as a stream of its own; the
// results are put together.  A damaged or cut-short stored header, and a block that does not fit the limit,
// get the reference's messages.  This path copies a few words to the host between its steps: it SYNCHRONISES
// the context's stream, unlike the rest of zipc_hip_inflate_batch.
static int inflate_huge_stream(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena,
                               const zipc_hip_stream_desc *d_descs, zipc_hip_stream_result *d_results, int crc_op) {
  if (crc_op == ZIPC_HIP_CRC_ADLER32 || crc_op == ZIPC_HIP_CRC_ADLER32_RFC1950) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, ctx->ensure(ctx->io_small, 256));
  StoredChain *d_st = (StoredChain *)ctx->io_small.p;
  StreamDesc sd;
  HIP_TRY(ctx, hipMemcpyAsync(&sd, d_descs, sizeof sd, hipMemcpyDeviceToHost, ctx->stream));
  ZD_LAUNCH(ctx, "stored_chain_probe", stored_chain_probe_kernel, dim3(1), dim3(1), 0, (const uint8_t *)d_src_arena,
            (const StreamDesc *)d_descs, d_st);
  StoredChain st;
  HIP_TRY(ctx, hipMemcpyAsync(&st, d_st, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  StreamResult res;
  res.status = ZIPC_HIP_ERR_INVALID_ARG; res.checksum = 0; res.out_len = 0;
  uint64_t blocks = 0;
  bool done = false;
  if (st.len0 != 0 && st.candidates != 0) {
    ZD_LAUNCH(ctx, "stored_chain_scan", stored_chain_scan_kernel, dim3((unsigned)(((uint64_t)st.candidates + 255) / 256)), dim3(256), 0,
              (const uint8_t *)d_src_arena, (const StreamDesc *)d_descs, d_st);
    HIP_TRY(ctx, hipMemcpyAsync(&st, d_st, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    blocks = st.first_bad;
    if (st.final_at < blocks) { blocks = (uint64_t)st.final_at + 1; done = true; }  // the final block is one of the chain
    // blocks that would overrun the destination are left to the walk below: it reports the reference's error
    {
      const uint64_t lim = (sd.flags & STREAM_HAS_LIMIT) ? sd.limit : ~0ull;
      const uint64_t room = lim < sd.dst_cap ? lim : sd.dst_cap;
      if (blocks * st.len0 > room) { blocks = room / st.len0; done = false; }
    }
    if (blocks)
      ZD_LAUNCH(ctx, "stored_chain_copy", stored_chain_copy_kernel, dim3((unsigned)blocks), dim3(256), 0,
                (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, (const StreamDesc *)d_descs, st.len0);
  }
  uint64_t used_src = blocks * (5ull + st.len0), made = blocks * (uint64_t)st.len0;
  const uint64_t limit = (sd.flags & STREAM_HAS_LIMIT) ? sd.limit : ~0ull;
  const uint64_t room_all = limit < sd.dst_cap ? limit : sd.dst_cap;
  bool settled = done;  // the result is known without the batch kernel
  if (done) {
    res.status = ZIPC_HIP_OK;
    res.out_len = made;
  }
  if (!settled) {
    // stored blocks of other lengths: walked header by header (64 at a time while the length stays), listed and
    // copied, a list at a time
    constexpr uint32_t LIST_CAP = 1u << 20;
    HIP_TRY(ctx, ctx->ensure(ctx->stored_list, (size_t)LIST_CAP * sizeof(StoredBlock)));
    StoredWalk *d_walk = (StoredWalk *)((uint8_t *)ctx->io_small.p + 192);
    static_assert(192 + sizeof(StoredWalk) <= 256, "io_small layout");
    for (;;) {
      StoredWalk w;
      w.src_pos = used_src; w.dst_pos = made; w.room = room_all - made; w.n_blocks = 0; w.stop = WALK_MORE;
      HIP_TRY(ctx, hipMemcpyAsync(d_walk, &w, sizeof w, hipMemcpyHostToDevice, ctx->stream));
      ZD_LAUNCH(ctx, "stored_walk", stored_walk_kernel, dim3(1), dim3(64), 0, (const uint8_t *)d_src_arena,
                (const StreamDesc *)d_descs, d_walk, (StoredBlock *)ctx->stored_list.p, LIST_CAP);
      HIP_TRY(ctx, hipMemcpyAsync(&w, d_walk, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (w.n_blocks)
        ZD_LAUNCH(ctx, "stored_list_copy", stored_list_copy_kernel, dim3(w.n_blocks < 65536u ? w.n_blocks : 65536u), dim3(256), 0,
                  (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, (const StreamDesc *)d_descs,
                  (const StoredBlock *)ctx->stored_list.p, w.n_blocks);
      used_src = w.src_pos;
      made = w.dst_pos;
      if (w.stop == WALK_MORE && w.n_blocks) continue;  // the list was full
      if (w.stop == WALK_FINAL) { res.status = ZIPC_HIP_OK; res.out_len = made; settled = true; }
      else if (w.stop == WALK_CORRUPT) { res.status = ZIPC_HIP_ERR_CORRUPTED; settled = true; }  // zd.ml:672-677
      else if (w.stop == WALK_ROOM) {  // Buf.add_string past the fixed size (zd.ml:29), or the caller's buffer is full
        res.status = (sd.flags & STREAM_HAS_LIMIT) && limit <= sd.dst_cap ? ZIPC_HIP_ERR_SIZE_EXCEEDED : ZIPC_HIP_ERR_DST_TOO_SMALL;
        settled = true;
      }
      break;  // WALK_OTHER: a block of another kind follows
    }
  }
  if (!settled) {
    StreamDesc rest = sd;
    rest.src_off += used_src; rest.src_len -= used_src;
    rest.dst_off += made; rest.dst_cap -= made;
    if (rest.flags & STREAM_HAS_LIMIT) rest.limit -= made;
    if (rest.src_len <= MAX_STREAM_LEN) {
      if (rest.dst_cap > MAX_STREAM_LEN) rest.dst_cap = MAX_STREAM_LEN;  // (the rest is an ordinary stream: it may produce up to that much)
      // the remainder's descriptor and result live in io_small, behind the StoredChain: the host forms hand
      // THEIR descriptors in io_desc / io_res to this function (d_descs, d_results), which must stay as they are
      // for the CRC pass below
      zipc_hip_stream_desc *d_rest = (zipc_hip_stream_desc *)((uint8_t *)ctx->io_small.p + 64);
      zipc_hip_stream_result *d_rest_res = (zipc_hip_stream_result *)((uint8_t *)ctx->io_small.p + 128);
      static_assert(sizeof(StoredChain) <= 64 && sizeof(StreamDesc) <= 64 && 128 + sizeof(StreamResult) <= 192, "io_small layout");
      HIP_TRY(ctx, hipMemcpyAsync(d_rest, &rest, sizeof rest, hipMemcpyHostToDevice, ctx->stream));
      const int stb = zipc_hip_inflate_batch(ctx, d_src_arena, d_dst_arena, d_rest, d_rest_res, 1, (size_t)rest.dst_cap,
                                             ZIPC_HIP_CRC_NOP);
      if (stb) return stb;
      HIP_TRY(ctx, hipMemcpyAsync(&res, d_rest_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (res.status == ZIPC_HIP_OK) res.out_len += made;
      else res.out_len = 0;
    }  // else: compressed blocks begin too early for the rest to be one ordinary stream (32-bit positions): INVALID_ARG stands
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_results, &res, sizeof res, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (crc_op == ZIPC_HIP_CRC_CRC32 && res.status == ZIPC_HIP_OK)
    return crc32_pass(ctx, (const uint8_t *)d_dst_arena, RANGE_INFLATE_OUT, (const StreamDesc *)d_descs,
                      (StreamResult *)d_results, 1, 0, 0, (size_t)res.out_len, nullptr);
  return ZIPC_HIP_OK;
}

// Streams of at least BLOCKS_MIN_SRC bytes by a wave per block (inflate.hip: find, dry, explore, chain, token, resolve;
// Adler-32 block by block as the reference updates it; CRC-32 is the caller's pass over the output).  The streams of
// a call go through every step side by side -- the kernels' grids have them as their second dimension -- and the
// host reads the counts of all of them back at once between the steps (three or four times a group, not per stream).
// handled[i]: stream i went that way (its result is in d_results); else it is left to inflate_batch_kernel -- a stream
// that is not a chain of dynamic blocks behind its first block, anything the dry run or the chain did not like: the
// one-wave kernel owns the reference's messages.  It SYNCHRONISES the context's stream.  ZIPC_HIP_INFLATE_BLOCKS=0
// turns it off.
constexpr size_t BLOCKS_MIN_SRC = 40u << 10, BLOCKS_MAX_SRC = 0x1FFFFFFFull;  // (bit offsets are 32-bit words here)
constexpr uint32_t BLOCKS_CAND_CAP = 65536, BLOCKS_REC_CAP = 262144;
// (a call whose descriptors are worth reading back: its longest stream alone is 4 ms of one wave.  Round 4 began with
// 1 MiB here and 96 KiB of input above: 64 x 512 KiB of text 8.5 -> 4.0 ms, 64 x 256 KiB 4.3 -> 2.6, one stream of
// 256 KiB 3.8 -> 1.3, of 128 KiB 2.0 -> 1.2; the block path's own floor is a good millisecond)
constexpr size_t BLOCKS_BATCH_MIN_DST = 256u << 10, BLOCKS_MAX_STREAMS = 1u << 20;
// tok[] and the two lists: 12 bytes of scratch per output byte.  Streams share a group while their capacities fit
// this much of it (a stream that needs more has a group to itself, and its scratch goes back afterwards)
constexpr size_t BLOCKS_TOK_BUDGET = (size_t)1 << 30;

static int inflate_blocks_group(zipc_hip_ctx *ctx, const uint8_t *src, uint8_t *dst, const StreamDesc *dd, StreamResult *d_results,
                                const StreamDesc *sds, const uint32_t *streams, size_t nj, int crc_op, uint8_t *handled) {
  const uint64_t EXPLORE_STRIDE = zd::tuning().explore_stride;  // bytes of input between two explorers
  const bool adler = crc_op == ZIPC_HIP_CRC_ADLER32 || crc_op == ZIPC_HIP_CRC_ADLER32_RFC1950;
  std::vector<BlocksJob> jobs(nj);
  std::vector<uint32_t> max_explorers(nj);
  // scratch: counts of every stream | the launches' job lists | per stream: first | cand | recs | sorted | sorted_src |
  // chain | chain_end | chain_iv | cks (listed blocks, then blocks the chain walked)
  size_t off = 0;
  auto carve = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_counts = carve(nj * sizeof(FindCounts)), o_jobs = carve(nj * sizeof(BlocksJob));
  struct Lists { size_t first, cand, recs, sorted, sorted_src, chain, chain_end, chain_iv, cks; };
  std::vector<Lists> at(nj);
  for (size_t j = 0; j < nj; j++) {
    const StreamDesc &sd = sds[streams[j]];
    BlocksJob &J = jobs[j];
    memset(&J, 0, sizeof J);
    J.stream = streams[j];
    J.first_cap = (uint32_t)(sd.src_len / 8 + 4096);
    J.cand_cap = (uint32_t)(sd.src_len / 512 + 64);
    if (J.cand_cap > BLOCKS_CAND_CAP) J.cand_cap = BLOCKS_CAND_CAP;
    // explorers (blocks without a findable header): one every EXPLORE_STRIDE bytes at most, 4 blocks listed each on average
    max_explorers[j] = (uint32_t)(sd.src_len / EXPLORE_STRIDE + 1);
    uint64_t rec_cap64 = 2ull * J.cand_cap + 4ull * max_explorers[j];  // (candidates, the blocks behind them, the explorers')
    if (rec_cap64 > BLOCKS_REC_CAP) rec_cap64 = BLOCKS_REC_CAP;
    J.rec_cap = J.chain_cap = (uint32_t)rec_cap64;
    Lists &L = at[j];
    L.first = carve((size_t)J.first_cap * 4); L.cand = carve((size_t)J.cand_cap * 4);
    L.recs = carve((size_t)J.rec_cap * sizeof(BlockRec)); L.sorted = carve((size_t)J.rec_cap * sizeof(BlockRec));
    L.sorted_src = carve((size_t)J.rec_cap * 4);
    L.chain = carve((size_t)J.chain_cap * sizeof(BlockStart)); L.chain_end = carve((size_t)J.chain_cap * sizeof(BlockEnd));
    L.chain_iv = carve((size_t)J.chain_cap * sizeof(ChainIv)); L.cks = carve(((size_t)J.rec_cap + J.chain_cap) * sizeof(BlockCk));
  }
  if (ctx->ensure(ctx->blocks_scratch, off) != hipSuccess) {
    (void)hipGetLastError();  // (no room for the lists: the streams' one waves need none)
    return ZIPC_HIP_OK;
  }
  uint8_t *base = (uint8_t *)ctx->blocks_scratch.p;
  FindCounts *d_counts = (FindCounts *)(base + o_counts);
  BlocksJob *d_jobs = (BlocksJob *)(base + o_jobs);
  for (size_t j = 0; j < nj; j++) {
    BlocksJob &J = jobs[j];
    const Lists &L = at[j];
    J.counts = d_counts + j;
    J.first = (uint32_t *)(base + L.first); J.cand = (uint32_t *)(base + L.cand);
    J.recs = (BlockRec *)(base + L.recs); J.sorted = (BlockRec *)(base + L.sorted);
    J.sorted_src = (uint32_t *)(base + L.sorted_src);
    J.chain = (BlockStart *)(base + L.chain); J.chain_end = (BlockEnd *)(base + L.chain_end);
    J.chain_iv = (ChainIv *)(base + L.chain_iv); J.cks = (BlockCk *)(base + L.cks);
  }
  std::vector<FindCounts> fc(nj);
  auto read_counts = [&]() -> hipError_t {
    const hipError_t e = hipMemcpyAsync(fc.data(), d_counts, nj * sizeof(FindCounts), hipMemcpyDeviceToHost, ctx->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
  };
  // a launch's streams: the jobs still on their way, as the kernels index them by blockIdx.y (the lists handed to
  // the copies stay until the group is through)
  std::deque<std::vector<BlocksJob>> handed;
  auto hand = [&](const std::vector<uint32_t> &which) -> hipError_t {
    handed.emplace_back();
    std::vector<BlocksJob> &v = handed.back();
    for (uint32_t j : which) v.push_back(jobs[j]);
    return hipMemcpyAsync(d_jobs, v.data(), v.size() * sizeof(BlocksJob), hipMemcpyHostToDevice, ctx->stream);
  };
  // the span decoder's index, a slot per wave of a launch: every stream's waves behind those of the one before
  auto span_slots = [&](const std::vector<uint32_t> &which) -> hipError_t {
    size_t waves = 0;
    for (uint32_t j : which) waves += jobs[j].n;
    const hipError_t e = ctx->ensure(ctx->inflate_scratch, waves * INFLATE_SCRATCH_PER_STREAM);
    if (e != hipSuccess) return e;
    waves = 0;
    for (uint32_t j : which) {
      jobs[j].span = (uint16_t *)ctx->inflate_scratch.p + waves * (INFLATE_SCRATCH_PER_STREAM / 2);
      waves += jobs[j].n;
    }
    return hipSuccess;
  };
  auto widest = [&](const std::vector<uint32_t> &which, auto need) {
    unsigned w = 1;
    for (uint32_t j : which) { const unsigned x = (unsigned)need(jobs[j], j); if (x > w) w = x; }
    return w;
  };
  std::vector<uint32_t> alive(nj), keep;
  for (size_t j = 0; j < nj; j++) alive[j] = (uint32_t)j;
  const unsigned ny = (unsigned)nj;

  HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, nj * sizeof(FindCounts), ctx->stream));
  HIP_TRY(ctx, hand(alive));
  ZD_LAUNCH(ctx, "inflate_find_headers", inflate_find_headers_kernel,
            dim3(widest(alive, [&](const BlocksJob &J, uint32_t) { return (sds[J.stream].src_len + 1023) / 1024; }), ny), dim3(256), 0, src, dd,
            (const BlocksJob *)d_jobs);
  ZD_LAUNCH(ctx, "inflate_find_lengths", inflate_find_lengths_kernel,
            dim3(widest(alive, [](const BlocksJob &J, uint32_t) { return (J.first_cap + 63u) / 64u; }), ny), dim3(64), 0, src, dd,
            (const BlocksJob *)d_jobs);
  // (how many candidates there are: the host asks when the lists are long or many -- a wave each is launched -- and lets
  // the kernels read it themselves for one stream of a few MiB: a round trip less)
  if (nj > 1 || jobs[0].cand_cap > 8192u) {
    HIP_TRY(ctx, read_counts());
    keep.clear();
    for (uint32_t j : alive)
      if (fc[j].n_cand != 0 && fc[j].n_cand <= jobs[j].cand_cap) { jobs[j].n = fc[j].n_cand; keep.push_back(j); }
    alive.swap(keep);
    if (alive.empty()) return ZIPC_HIP_OK;
  } else {
    jobs[0].n = jobs[0].cand_cap;
  }
  if (span_slots(alive) != hipSuccess) { (void)hipGetLastError(); return ZIPC_HIP_OK; }  // (no room for the waves' span index: the streams' one waves need 2304 bytes each)
  HIP_TRY(ctx, hand(alive));
  unsigned na = (unsigned)alive.size();
  const unsigned dry_waves = widest(alive, [](const BlocksJob &J, uint32_t) { return J.n; });
  ZD_LAUNCH(ctx, "inflate_blocks_dry", inflate_blocks_dry_kernel, dim3(dry_waves, na), dim3(64), 0, src, dst, dd, (const BlocksJob *)d_jobs);
  ZD_LAUNCH(ctx, "inflate_sort_blocks", inflate_sort_blocks_kernel, dim3((dry_waves + 255u) / 256u, na), dim3(256), 0, (const BlocksJob *)d_jobs);
  ZD_LAUNCH(ctx, "inflate_chain", inflate_chain_kernel, dim3(1, na), dim3(64), 0, src, dst, dd, (const BlocksJob *)d_jobs, 0);
  HIP_TRY(ctx, read_counts());
  keep.clear();
  std::vector<uint32_t> lost;  // streams whose chain came to a block nobody listed
  for (uint32_t j : alive) {
    if (fc[j].n_cand == 0 || fc[j].n_cand > jobs[j].cand_cap) continue;
    if (!fc[j].chain_ok && fc[j].miss_bit != ~0ull) lost.push_back(j);
    keep.push_back(j);
  }
  alive.swap(keep);
  if (!lost.empty()) {
    // explorers from there on, then the chain again (which now walks what is still missing itself)
    for (uint32_t j : lost) {
      const uint64_t bits_left = sds[jobs[j].stream].src_len * 8u - fc[j].miss_bit;
      uint64_t ne = (bits_left + EXPLORE_STRIDE * 8u - 1) / (EXPLORE_STRIDE * 8u);
      if (ne > max_explorers[j]) ne = max_explorers[j];
      // (behind the explorers a wave per block listed so far: the block that follows it, inflate.hip)
      const uint32_t listed = fc[j].n_recs < jobs[j].rec_cap ? fc[j].n_recs : jobs[j].rec_cap;
      jobs[j].n_blocks = (uint32_t)ne;
      jobs[j].n = (uint32_t)ne + listed;
    }
    if (span_slots(lost) != hipSuccess) { (void)hipGetLastError(); return ZIPC_HIP_OK; }
    HIP_TRY(ctx, hand(lost));
    const unsigned nl = (unsigned)lost.size();
    ZD_LAUNCH(ctx, "inflate_explore", inflate_explore_kernel, dim3(widest(lost, [](const BlocksJob &J, uint32_t) { return J.n; }), nl), dim3(64), 0,
              src, dst, dd, (const BlocksJob *)d_jobs, (uint32_t)(EXPLORE_STRIDE * 8u));
    ZD_LAUNCH(ctx, "inflate_sort_blocks", inflate_sort_blocks_kernel,
              dim3(widest(lost, [](const BlocksJob &J, uint32_t) { return (J.rec_cap + 255u) / 256u; }), nl), dim3(256), 0, (const BlocksJob *)d_jobs);
    ZD_LAUNCH(ctx, "inflate_chain", inflate_chain_kernel, dim3(1, nl), dim3(64), 0, src, dst, dd, (const BlocksJob *)d_jobs, 1);
    HIP_TRY(ctx, read_counts());
  }
  // (sources written down as what they are copies of -- inflate_span.h -- cost the token run 0.2-0.4 ms a block and a
  // wave per block instead of one per interval, and save the resolve rounds of a long stream more: with 256 hops a
  // round, 64 MiB of text 6.3-10.8 -> 5.7-6.5 ms, 16 MiB 2.7-4.4 <- 3.3-4.2)
  const int follow_env = zd::tuning().inflate_follow;
  // (what counts is the output of the whole call, whose bytes the rounds look at side by side: 64 x 1 MiB of text
  // 5.8 -> 5.0 ms, 8 x 8 MiB 5.3 -> 4.7; 16 x 1 MiB 2.5 <- 3.2, one MiB 1.4 <- 2.4)
  size_t call_out = 0;
  for (uint32_t j : alive)
    if (fc[j].chain_ok && fc[j].n_blocks >= 2) call_out += fc[j].out_len;
  keep.clear();
  size_t tok_words = 0;
  for (uint32_t j : alive) {
    if (!fc[j].chain_ok || fc[j].n_blocks < 2 || fc[j].out_len == 0) continue;  // (one block: nothing to gain)
    BlocksJob &J = jobs[j];
    J.out_len = (uint32_t)fc[j].out_len;
    J.n_blocks = fc[j].n_blocks;
    // (... and nothing on data with few matches: 16 MiB of records that deflate to 0.85, resolve 0.13 ms either way)
    J.follow = follow_env >= 0 ? follow_env : call_out >= ((size_t)32 << 20) && (uint64_t)J.out_len * 2u >= sds[J.stream].src_len * 3u;
    // the token run: a wave per interval of a block (its checkpoints), or -- follow -- a wave per block
    J.n = J.follow ? J.n_blocks : fc[j].n_intervals;
    tok_words += ((size_t)J.out_len * 3 + 63) & ~(size_t)63;
    keep.push_back(j);
  }
  alive.swap(keep);
  if (alive.empty()) return ZIPC_HIP_OK;
  if (ctx->ensure(ctx->tok_scratch, tok_words * 4) != hipSuccess) {  // tok[], and two lists of bytes still to resolve
    (void)hipGetLastError();  // (no room for a word per byte and the lists: the streams' one waves need none)
    return ZIPC_HIP_OK;
  }
  tok_words = 0;
  for (uint32_t j : alive) {
    jobs[j].tok = (uint32_t *)ctx->tok_scratch.p + tok_words;
    tok_words += ((size_t)jobs[j].out_len * 3 + 63) & ~(size_t)63;
  }
  if (span_slots(alive) != hipSuccess) { (void)hipGetLastError(); return ZIPC_HIP_OK; }  // (no room for the waves' span index: the streams' one waves need 2304 bytes each)
  HIP_TRY(ctx, hand(alive));
  na = (unsigned)alive.size();
  const unsigned out_grid = widest(alive, [](const BlocksJob &J, uint32_t) { return (J.out_len + 255u) / 256u; });
  ZD_LAUNCH(ctx, "inflate_tok_init", inflate_tok_init_kernel, dim3(out_grid, na), dim3(256), 0, (const BlocksJob *)d_jobs);
  ZD_LAUNCH(ctx, "inflate_blocks_token", inflate_blocks_token_kernel, dim3(widest(alive, [](const BlocksJob &J, uint32_t) { return J.n; }), na),
            dim3(64), 0, src, dst, dd, (const BlocksJob *)d_jobs);
  // (hops a thread follows in a round: 8 left most bytes of a text for the next round -- 16 MiB: three rounds over nearly
  // everything, 3.4 ms; 64 and more let nearly every byte arrive in the first: 0.28 ms)
  const int hops0 = zd::tuning().resolve_hops0, hops1 = zd::tuning().resolve_hops1;
  const int rounds = hops0 >= 16 && hops1 >= 16 ? 6 : RESOLVE_ROUNDS;  // (16^6 links: more than a stream has bytes)
  for (int r = 0; r < rounds; r++)
    ZD_LAUNCH(ctx, "inflate_resolve", inflate_resolve_kernel, dim3(r == 0 || out_grid < 2048u ? out_grid : 2048u, na), dim3(256), 0,
              (const BlocksJob *)d_jobs, r, r == 0 ? hops0 : hops1);
  ZD_LAUNCH(ctx, "inflate_gather", inflate_gather_kernel, dim3(out_grid, na), dim3(256), 0, dst, dd, (const BlocksJob *)d_jobs);
  HIP_TRY(ctx, read_counts());
  // (12 bytes of scratch per output byte: what a long stream took goes back -- a context lives as long as its thread,
  // and a 1 GiB member would pin 12 GiB per device; the stream is idle here, the gather has been waited for)
  if (ctx->tok_scratch.cap > BLOCKS_TOK_BUDGET) free_buf(ctx->tok_scratch);
  keep.clear();
  size_t n_chunks = 0;
  for (uint32_t j : alive) {
    if (fc[j].token_bad != 0 || fc[j].more[rounds - 1] != 0) continue;  // (the one-wave kernel writes the output again)
    n_chunks += fc[j].n_chunks;
    keep.push_back(j);
  }
  alive.swap(keep);
  if (alive.empty()) return ZIPC_HIP_OK;
  na = (unsigned)alive.size();
  if (adler) {  // block by block, every block's bytes in chunks of their own (inflate.hip)
    if (ctx->ensure(ctx->adler_sums, (n_chunks + 1) * 12) != hipSuccess) { (void)hipGetLastError(); return ZIPC_HIP_OK; }  // (the one waves write output and checksum again)
    n_chunks = 0;
    for (uint32_t j : alive) { jobs[j].sums = (uint32_t *)ctx->adler_sums.p + n_chunks * 3; n_chunks += fc[j].n_chunks; }
  }
  HIP_TRY(ctx, hand(alive));
  ZD_LAUNCH(ctx, "inflate_blocks_result", inflate_blocks_result_kernel, dim3((na + 63u) / 64u), dim3(64), 0, (const BlocksJob *)d_jobs, d_results, na);
  if (adler) {
    if (n_chunks)
      ZD_LAUNCH(ctx, "inflate_adler_chunks", inflate_adler_chunks_kernel, dim3(widest(alive, [&](const BlocksJob &, uint32_t j) { return fc[j].n_chunks; }), na),
                dim3(64), 0, (const uint8_t *)dst, dd, (const BlocksJob *)d_jobs);
    ZD_LAUNCH(ctx, "inflate_adler_fold", inflate_adler_fold_kernel, dim3(1, na), dim3(64), 0, (const BlocksJob *)d_jobs,
              crc_op == ZIPC_HIP_CRC_ADLER32_RFC1950 ? 1 : 0, d_results);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the lists handed to the copies go with this frame)
  for (uint32_t j : alive) { handled[jobs[j].stream] = 1; ctx->last_inflate_blocks += jobs[j].n_blocks; }
  return ZIPC_HIP_OK;
}

// which of a call's streams go by blocks, group by group; *n_handled: how many did
// h_descs: the caller's own host copy of the descriptors (the many-stream host forms have one), or null: they are read
// back from the device, which waits for everything the stream holds -- the host forms feed sub-batch g + 1 while g's
// kernels run, and a read-back per sub-batch (before the model below had even said whether any stream goes by blocks:
// for an archive of equal members none does) put the host behind every sub-batch's copies and kernels.
static int inflate_by_blocks(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena, const zipc_hip_stream_desc *d_descs,
                             zipc_hip_stream_result *d_results, size_t n_streams, int crc_op, std::vector<StreamDesc> &sds,
                             std::vector<uint8_t> &handled, size_t *n_handled, const StreamDesc *h_descs) {
  *n_handled = 0;
  handled.assign(n_streams, 0);
  if (!zd::tuning().inflate_blocks) return ZIPC_HIP_OK;
  if (h_descs) {
    sds.assign(h_descs, h_descs + n_streams);
  } else {
    sds.resize(n_streams);
    HIP_TRY(ctx, hipMemcpyAsync(sds.data(), d_descs, n_streams * sizeof(StreamDesc), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<uint32_t> group;
  size_t group_cap = 0;
  auto run = [&]() -> int {
    if (group.empty()) return ZIPC_HIP_OK;
    const int st = inflate_blocks_group(ctx, (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, (const StreamDesc *)d_descs,
                                        (StreamResult *)d_results, sds.data(), group.data(), group.size(), crc_op, handled.data());
    group.clear();
    group_cap = 0;
    return st;
  };
  std::vector<uint32_t> fit;  // the streams the block path takes at all
  for (size_t i = 0; i < n_streams; i++) {
    const StreamDesc &sd = sds[i];
    if (sd.src_len < BLOCKS_MIN_SRC || sd.src_len > BLOCKS_MAX_SRC || sd.dst_cap < 8 || sd.dst_cap > MAX_STREAM_LEN) continue;
    // Runs (zeros, short periods: output beyond 64 x the input) are not for this path: a word of tok[] per byte of a
    // run costs more than the run (16 MiB of zeros as zlib codes them, 4 blocks: token run 10-11 ms, the one wave
    // 3.4-6.9), and where the reference's encoder has coded them with the fixed code, the explorers' walks never fall
    // into step with a bit stream that has a period (64 MiB: the chain walks nearly every block itself, 65 ms).
    if (sd.dst_cap / 64 > sd.src_len) continue;
    fit.push_back((uint32_t)i);
  }
  // Which of them go by blocks: the one waves of a call run side by side, and a call of thousands of streams fills
  // the device with them -- its time is the longest stream's, about 15 ms per MiB of output -- while the block path
  // takes the streams' bytes one after the other, about 0.09 ms per MiB and 1 ms for a group's launches and
  // read-backs (64 x 1 MiB: 5.0 ms against 17; 4096 x 1 MiB: 370 ms against 16).  So the k longest streams go by
  // blocks, with the k that makes the sum of both parts smallest: all of a few long streams, the few long members
  // among an archive's many short ones, none of thousands of equal ones.
  {
    constexpr double WAVE_MS_PER_MIB = 15.0, BLOCKS_MS_PER_MIB = 0.09, BLOCKS_MS_FIXED = 1.0, MIB = 1048576.0;
    std::sort(fit.begin(), fit.end(), [&](uint32_t x, uint32_t y) { return sds[x].dst_cap != sds[y].dst_cap ? sds[x].dst_cap > sds[y].dst_cap : x < y; });
    uint64_t longest_other = 0;  // (of the streams the block path does not take)
    {
      std::vector<uint8_t> in_fit(n_streams, 0);
      for (uint32_t i : fit) in_fit[i] = 1;
      for (size_t i = 0; i < n_streams; i++)
        if (!in_fit[i] && sds[i].dst_cap > longest_other) longest_other = sds[i].dst_cap;
    }
    size_t best_k = 0;
    double best_ms = 0, taken_mib = 0;
    for (size_t k = 0; k <= fit.size(); k++) {
      const uint64_t longest_left = k < fit.size() ? (sds[fit[k]].dst_cap > longest_other ? sds[fit[k]].dst_cap : longest_other) : longest_other;
      const bool any_left = k < n_streams;
      const double ms = (k ? BLOCKS_MS_FIXED + taken_mib * BLOCKS_MS_PER_MIB : 0.0) + (any_left ? (double)longest_left / MIB * WAVE_MS_PER_MIB : 0.0);
      if (k == 0 || ms < best_ms) { best_ms = ms; best_k = k; }
      if (k < fit.size()) taken_mib += (double)sds[fit[k]].dst_cap / MIB;
    }
    fit.resize(best_k);
    std::sort(fit.begin(), fit.end());
  }
  for (uint32_t i : fit) {
    // (what a stream may produce: its capacity; the group's share of tok[] is sized by what the chains then say)
    const size_t may = (size_t)sds[i].dst_cap * 12;
    if (!group.empty() && group_cap + may > BLOCKS_TOK_BUDGET) { const int st = run(); if (st) return st; }
    group.push_back(i);
    group_cap += may;
  }
  const int st = run();
  if (st) return st;
  for (size_t i = 0; i < n_streams; i++) *n_handled += handled[i];
  return ZIPC_HIP_OK;
}

static int inflate_batch_one_wave(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena, const zipc_hip_stream_desc *d_descs,
                                  zipc_hip_stream_result *d_results, size_t n_streams, size_t max_dst_cap, int crc_op, bool marked = false);
static int inflate_batch_impl(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena, const zipc_hip_stream_desc *d_descs,
                              zipc_hip_stream_result *d_results, size_t n_streams, size_t max_dst_cap, int crc_op, const StreamDesc *h_descs,
                              bool first_of_call);

int zipc_hip_inflate_batch(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena,
                           const zipc_hip_stream_desc *d_descs, zipc_hip_stream_result *d_results,
                           size_t n_streams, size_t max_dst_cap, int crc_op) {
  return inflate_batch_impl(ctx, d_src_arena, d_dst_arena, d_descs, d_results, n_streams, max_dst_cap, crc_op, nullptr, true);
}

// first_of_call: zipc_hip_last_inflate_blocks counts from zero (the host forms' sub-batches add up)
static int inflate_batch_impl(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena, const zipc_hip_stream_desc *d_descs,
                              zipc_hip_stream_result *d_results, size_t n_streams, size_t max_dst_cap, int crc_op, const StreamDesc *h_descs,
                              bool first_of_call) {
  if (!ctx || !d_descs || !d_results) return ZIPC_HIP_ERR_INVALID_ARG;
  if (crc_op < 0 || crc_op > 3 || n_streams > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
  if (first_of_call) ctx->last_inflate_blocks = 0;  // (also for a call that never reaches the block path: "0 when the stream's one wave decoded it")
  if (n_streams == 0) return ZIPC_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_streams == 1 && max_dst_cap > MAX_STREAM_LEN)
    return inflate_huge_stream(ctx, d_src_arena, d_dst_arena, d_descs, d_results, crc_op);
  // one long stream, or a call of long streams (an archive's big members): by blocks, side by side -- their one
  // waves take 10-17 ms per MiB of the longest
  if (max_dst_cap <= MAX_STREAM_LEN && n_streams <= BLOCKS_MAX_STREAMS &&
      max_dst_cap >= (n_streams == 1 ? BLOCKS_MIN_SRC : BLOCKS_BATCH_MIN_DST)) {
    std::vector<StreamDesc> sds;
    std::vector<uint8_t> handled;
    size_t n_handled = 0;
    const int by = inflate_by_blocks(ctx, d_src_arena, d_dst_arena, d_descs, d_results, n_streams, crc_op, sds, handled, &n_handled, h_descs);
    if (by != ZIPC_HIP_OK) return by;
    if (n_handled == n_streams) {
      if (crc_op != ZIPC_HIP_CRC_CRC32) return ZIPC_HIP_OK;
      return crc32_pass(ctx, (const uint8_t *)d_dst_arena, RANGE_INFLATE_OUT, (const StreamDesc *)d_descs, (StreamResult *)d_results,
                        n_streams, 0, 0, max_dst_cap, nullptr);
    }
    if (n_handled) {
      // the others by their one waves, over a copy of the descriptors that says which streams are through (the CRC
      // pass behind the kernel takes every stream's output as it finds it in d_results)
      for (size_t i = 0; i < n_streams; i++)
        if (handled[i]) sds[i].flags |= STREAM_DONE;
      HIP_TRY(ctx, ctx->ensure(ctx->descs_marked, n_streams * sizeof(StreamDesc)));
      HIP_TRY(ctx, hipMemcpyAsync(ctx->descs_marked.p, sds.data(), n_streams * sizeof(StreamDesc), hipMemcpyHostToDevice, ctx->stream));
      const int st = inflate_batch_one_wave(ctx, d_src_arena, d_dst_arena, (const zipc_hip_stream_desc *)ctx->descs_marked.p, d_results,
                                            n_streams, max_dst_cap, crc_op, true);
      const hipError_t e = hipStreamSynchronize(ctx->stream);  // (sds goes with this frame)
      if (st == ZIPC_HIP_OK) HIP_TRY(ctx, e);
      return st;
    }
  }
  return inflate_batch_one_wave(ctx, d_src_arena, d_dst_arena, d_descs, d_results, n_streams, max_dst_cap, crc_op);
}

// the batch kernel: one wave per stream
static int inflate_batch_one_wave(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena, const zipc_hip_stream_desc *d_descs,
                                  zipc_hip_stream_result *d_results, size_t n_streams, size_t max_dst_cap, int crc_op, bool marked) {
  const int k_crc_op = crc_op | (marked ? CRC_OP_MARKED : 0);  // what the kernels are told (inflate.hip inflate_skips_stream)
  // one wave per stream (ZIPC_HIP_SLICES > 1: in slices on queues of their own, the CRC pass of one slice
  // beside the inflate kernel of the next; measured, not the default: deflate.hip)
  HIP_TRY(ctx, ctx->ensure(ctx->inflate_scratch, n_streams * INFLATE_SCRATCH_PER_STREAM));
  const size_t segs = crc32_segs(max_dst_cap);
  if (crc_op == ZIPC_HIP_CRC_CRC32) {
    if (n_streams * segs > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
    HIP_TRY(ctx, ctx->ensure(ctx->crc_partials, n_streams * segs * sizeof(uint32_t)));
  }
  const size_t k = crc_op == ZIPC_HIP_CRC_CRC32 ? batch_slices(n_streams) : 1;
  if (k > 1) HIP_TRY(ctx, ctx->fork(k));
  int st = ZIPC_HIP_OK;
  for (size_t i = 0; i < k && st == ZIPC_HIP_OK; i++) {
    const size_t lo = n_streams * i / k, hi = n_streams * (i + 1) / k;
    if (k > 1) ctx->use_slice_stream(i);
    const StreamDesc *dd = (const StreamDesc *)d_descs + lo;
    StreamResult *dr = (StreamResult *)d_results + lo;
    if (n_streams <= 256)  // (a few streams: the form that shares the tables of blocks with one and the same header, inflate.hip)
      ZD_LAUNCH(ctx, "inflate_batch", inflate_batch_few_kernel, dim3((unsigned)(hi - lo)), dim3(64), 0,
                (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, dd, dr, (uint32_t)(hi - lo),
                (uint16_t *)ctx->inflate_scratch.p + lo * (INFLATE_SCRATCH_PER_STREAM / 2), k_crc_op);
    else
      ZD_LAUNCH(ctx, "inflate_batch", inflate_batch_kernel, dim3((unsigned)(hi - lo)), dim3(64), 0,
                (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, dd, dr, (uint32_t)(hi - lo),
                (uint16_t *)ctx->inflate_scratch.p + lo * (INFLATE_SCRATCH_PER_STREAM / 2), k_crc_op);
    if (hipGetLastError() != hipSuccess) { ctx->last_error = "inflate_batch launch failed"; st = ZIPC_HIP_ERR_HIP; break; }
    if (crc_op == ZIPC_HIP_CRC_CRC32)
      st = crc32_pass(ctx, (const uint8_t *)d_dst_arena, RANGE_INFLATE_OUT, dd, dr, hi - lo, 0, 0, max_dst_cap, nullptr,
                      lo * segs, true);
  }
  if (k > 1) {
    const hipError_t e = ctx->join(k);
    if (st == ZIPC_HIP_OK) HIP_TRY(ctx, e);
  }
  return st;
}

int zipc_hip_deflate_batch(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena,
                           const zipc_hip_stream_desc *d_descs, zipc_hip_stream_result *d_results,
                           size_t n_streams, size_t max_src_len, size_t total_src_len, int level,
                           int crc_op) {
  if (!ctx || !d_descs || !d_results) return ZIPC_HIP_ERR_INVALID_ARG;
  if (crc_op < 0 || crc_op > 3 || level < 0 || level > 3 || n_streams > 0x7FFFFFFFull)
    return ZIPC_HIP_ERR_INVALID_ARG;
  if (max_src_len > MAX_STREAM_LEN) return ZIPC_HIP_ERR_INVALID_ARG;
  if (n_streams == 0) return ZIPC_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, ctx->ensure(ctx->deflate_scratch,
                           deflate_scratch_bytes(n_streams, max_src_len, total_src_len, level)));
  if (crc_op == ZIPC_HIP_CRC_CRC32) {  // (launch_deflate runs the pass, group by group)
    if (n_streams * crc32_segs(max_src_len) > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
    HIP_TRY(ctx, ctx->ensure(ctx->crc_partials, n_streams * crc32_segs(max_src_len) * sizeof(uint32_t)));
  }
  HIP_TRY(ctx, launch_deflate(ctx, (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena,
                              (const StreamDesc *)d_descs, (StreamResult *)d_results, n_streams,
                              max_src_len, total_src_len, level, crc_op));
  return ZIPC_HIP_OK;
}

size_t zipc_hip_debug_chain_positions(size_t n_streams, size_t total_src_len) { return zd::debug_chain_positions(n_streams, total_src_len); }
int zipc_hip_debug_chain_links(zipc_hip_ctx *ctx, const void *d_src_arena, const zipc_hip_stream_desc *d_descs, size_t n_streams,
                               size_t max_src_len, size_t total_src_len, int which, void *d_links, size_t links_cap, void *d_pos_base) {
  if (!ctx || !d_descs || !d_links || which < 0 || which > 1 || n_streams == 0 || n_streams > 0x7FFFFFFFull || max_src_len > MAX_STREAM_LEN)
    return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, ctx->ensure(ctx->deflate_scratch, deflate_scratch_bytes(n_streams, max_src_len, total_src_len, ZIPC_HIP_LEVEL_DEFAULT)));
  HIP_TRY(ctx, zd::debug_chain_links(ctx, (const uint8_t *)d_src_arena, (const StreamDesc *)d_descs, n_streams, max_src_len, total_src_len,
                                     which, (uint16_t *)d_links, links_cap, (uint64_t *)d_pos_base));
  return ZIPC_HIP_OK;
}

int zipc_hip_reserve(zipc_hip_ctx *ctx, size_t n_streams, size_t max_src_len, size_t total_src_len) {
  if (!ctx) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, ctx->ensure(ctx->deflate_scratch,
                           deflate_scratch_bytes(n_streams, max_src_len, total_src_len, ZIPC_HIP_LEVEL_BEST)));
  size_t segs = (max_src_len + CRC_SEG_BYTES - 1) / CRC_SEG_BYTES + 1;
  HIP_TRY(ctx, ctx->ensure(ctx->crc_partials, n_streams * segs * sizeof(uint32_t)));
  return ZIPC_HIP_OK;
}

int zipc_hip_checksum_device(zipc_hip_ctx *ctx, const void *d_buf, size_t len, int want_crc32,
                             int want_adler32, uint32_t *d_out) {
  if (!ctx || !d_out || (!d_buf && len)) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // Both checksums: ONE pass over the bytes (crc32_adler_segments_kernel leaves the CRC partials and the
  // Adler chunk sums), then the two finishes -- on two queues for a large buffer.  ZIPC_HIP_CHECKSUM_FUSED=0
  // keeps the two passes of rounds 1-3 (CRC on a side queue from 64 MiB on) for comparison and for the tests.
  const bool fused_ok = zd::tuning().checksum_fused;
  const bool c3_two_queues = true;
  const bool fused = want_crc32 && want_adler32 && len > 0 && fused_ok;
  const bool side = want_crc32 && want_adler32 && len >= (64u << 20) && c3_two_queues;
  if (side || fused)  // (before any fork: growing a buffer synchronises)
    HIP_TRY(ctx, ctx->ensure(ctx->crc_partials, crc32_segs(len) * sizeof(uint32_t)));
  if (side && !fused) {
    HIP_TRY(ctx, ctx->fork(1));
    ctx->cur = ctx->side[0];
  }
  if (want_crc32 && !fused) {
    int st = crc32_pass(ctx, (const uint8_t *)d_buf, RANGE_SINGLE, nullptr, nullptr, 1, 0, len, len, d_out);
    if (side) ctx->cur = ctx->stream;
    if (st != ZIPC_HIP_OK) { if (side) (void)ctx->join(1); return st; }
  }
  struct Joiner {  // the side queue is joined on every way out of the Adler half
    zipc_hip_ctx *c; bool on;
    ~Joiner() { if (on) { c->cur = c->stream; (void)c->join(1); } }
  } joiner{ctx, side && !fused};
  if (want_adler32) {
    const uint64_t n_chunks = len ? len / ADLER_CHUNK + 1 : 0;
    // chunk sums, then the ambiguous-chunk list and the per-run arrays of the chain kernels
    const size_t sums_bytes = ((size_t)(n_chunks + 1) * sizeof(uint2) + 255) / 256 * 256;
    AdlerRuns R;
    R.n_runs = 1024;  // one thread per run; at least ~8 chunks per run
    while (R.n_runs < ADLER_MAX_RUNS && (uint64_t)R.n_runs * 8 < n_chunks) R.n_runs *= 2;
    const size_t run_bytes = (size_t)R.n_runs * sizeof(uint32_t);
    HIP_TRY(ctx, ctx->ensure(ctx->adler_sums, sums_bytes + ADLER_AMB_CAP * 16 + 5 * run_bytes + 256));
    uint2 *sums = (uint2 *)ctx->adler_sums.p;
    uint8_t *q = (uint8_t *)ctx->adler_sums.p + sums_bytes;
    uint32_t *amb = (uint32_t *)q; q += ADLER_AMB_CAP * 16;
    R.sum = (uint32_t *)q; q += run_bytes;
    R.s1_before = (uint32_t *)q; q += run_bytes;
    R.s1_after = (uint32_t *)q; q += run_bytes;
    R.last_hi = (uint32_t *)q; q += run_bytes;
    R.res_before = (uint32_t *)q; q += run_bytes;
    R.amb_count = (uint32_t *)q;
    if (fused) {
      const size_t segs = crc32_segs(len);
      if (segs > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
      uint32_t *partials = (uint32_t *)ctx->crc_partials.p;
      HIP_TRY(ctx, hipMemsetAsync(sums, 0, (size_t)n_chunks * sizeof(uint2), ctx->stream));
      ZD_LAUNCH(ctx, "crc32_adler_segments", crc32_adler_segments_kernel, dim3((unsigned)segs), dim3(256), 0,
                (const uint8_t *)d_buf, (uint64_t)len, (uint32_t)segs, (const uint32_t *)ctx->crc_nib.p, partials,
                sums, n_chunks);
      HIP_TRY(ctx, hipGetLastError());
      if (side) {  // the CRC's finish beside the Adler chain
        HIP_TRY(ctx, ctx->fork(1));
        ctx->cur = ctx->side[0];
        joiner.on = true;
      }
      HIP_TRY(ctx, crc32_finish_launch(ctx, RANGE_SINGLE, nullptr, nullptr, 1, len, len, partials, d_out));
      ctx->cur = ctx->stream;
    } else if (n_chunks) {
      if ((n_chunks + 3) / 4 > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
      ZD_LAUNCH(ctx, "adler_chunks", adler_chunks_kernel, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0,
                (const uint8_t *)d_buf, (uint64_t)len, n_chunks, sums);
    }
    const uint64_t per = n_chunks ? (n_chunks + R.n_runs - 1) / R.n_runs : 1;
    HIP_TRY(ctx, hipMemsetAsync(R.amb_count, 0, sizeof(uint32_t), ctx->stream));
    ZD_LAUNCH(ctx, "adler_runs_s1", adler_runs_s1_kernel, dim3(R.n_runs / 256), dim3(256), 0, (const uint2 *)sums,
              n_chunks, per, R);
    ZD_LAUNCH(ctx, "adler_scan_runs", adler_scan_runs_kernel, dim3(1), dim3(1024), 0, (const uint32_t *)R.sum,
              R.s1_before, R.n_runs, 1u);
    ZD_LAUNCH(ctx, "adler_runs_a", adler_runs_a_kernel, dim3(R.n_runs / 256), dim3(256), 0, (const uint2 *)sums,
              (uint64_t)len, n_chunks, per, R, amb, ADLER_AMB_CAP);
    ZD_LAUNCH(ctx, "adler_scan_runs", adler_scan_runs_kernel, dim3(1), dim3(1024), 0, (const uint32_t *)R.sum,
              R.res_before, R.n_runs, 0u);
    if (ctx->adler_rfc1950) {  // RFC 1950's arithmetic: the chunk sums combine without the reference's sign cases
      ZD_LAUNCH(ctx, "adler_rfc_finish", adler_rfc_finish_kernel, dim3(1), dim3(1024), 0, (const uint2 *)sums,
                (uint64_t)len, n_chunks, d_out + 1);
      HIP_TRY(ctx, hipGetLastError());
      return ZIPC_HIP_OK;
    }
    ZD_LAUNCH(ctx, "adler_replay", adler_replay_kernel, dim3(1), dim3(1024), 0, (const uint2 *)sums, (uint64_t)len,
              n_chunks, per, R, amb, ADLER_AMB_CAP, d_out + 1);
    HIP_TRY(ctx, hipGetLastError());
  }
  return ZIPC_HIP_OK;
}

// ---- host forms ------------------------------------------------------------------

static int stage_in(zipc_hip_ctx *ctx, const void *src, size_t len) {
  HIP_TRY(ctx, ctx->ensure(ctx->io_src, len + 64));
  if (len) HIP_TRY(ctx, hipMemcpyAsync(ctx->io_src.p, src, len, hipMemcpyHostToDevice, ctx->stream));
  return ZIPC_HIP_OK;
}

static int checksum_host(zipc_hip_ctx *ctx, const void *src, size_t len, int want_crc, uint32_t *out) {
  if (!ctx || !out || (!src && len)) return ZIPC_HIP_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int st = stage_in(ctx, src, len);
  if (st) return st;
  HIP_TRY(ctx, ctx->ensure(ctx->io_small, 64));
  uint32_t *d_out = (uint32_t *)ctx->io_small.p;
  st = zipc_hip_checksum_device(ctx, ctx->io_src.p, len, want_crc, !want_crc, d_out);
  if (st) return st;
  uint32_t h[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *out = want_crc ? h[0] : h[1];
  return ZIPC_HIP_OK;
}

int zipc_hip_crc32(zipc_hip_ctx *ctx, const void *src, size_t len, uint32_t *crc) {
  return checksum_host(ctx, src, len, 1, crc);
}
int zipc_hip_adler32(zipc_hip_ctx *ctx, const void *src, size_t len, uint32_t *adler) {
  return checksum_host(ctx, src, len, 0, adler);
}

// one stream through the batch kernels; is_inflate selects the direction
static int one_stream(zipc_hip_ctx *ctx, bool is_inflate, const void *src, size_t len, int has_limit,
                      size_t limit, int level, int crc_op, void *dst, size_t dst_cap, size_t *out_len,
                      uint32_t *checksum) {
  if (!ctx || (!src && len) || (!dst && dst_cap) || !out_len) return ZIPC_HIP_ERR_INVALID_ARG;
  *out_len = 0;
  if (checksum) *checksum = 0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int st = stage_in(ctx, src, len);
  if (st) return st;
  HIP_TRY(ctx, ctx->ensure(ctx->io_dst, dst_cap + 64));
  HIP_TRY(ctx, ctx->ensure(ctx->io_desc, sizeof(StreamDesc)));
  HIP_TRY(ctx, ctx->ensure(ctx->io_res, sizeof(StreamResult)));
  StreamDesc d;
  memset(&d, 0, sizeof d);
  d.src_off = 0; d.src_len = len; d.dst_off = 0; d.dst_cap = dst_cap;
  d.limit = limit; d.flags = has_limit ? STREAM_HAS_LIMIT : 0;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->io_desc.p, &d, sizeof d, hipMemcpyHostToDevice, ctx->stream));
  if (is_inflate)
    st = zipc_hip_inflate_batch(ctx, ctx->io_src.p, ctx->io_dst.p, (zipc_hip_stream_desc *)ctx->io_desc.p,
                                (zipc_hip_stream_result *)ctx->io_res.p, 1, dst_cap, crc_op);
  else
    st = zipc_hip_deflate_batch(ctx, ctx->io_src.p, ctx->io_dst.p, (zipc_hip_stream_desc *)ctx->io_desc.p,
                                (zipc_hip_stream_result *)ctx->io_res.p, 1, len, len, level, crc_op);
  if (st) return st;
  StreamResult r;
  HIP_TRY(ctx, hipMemcpyAsync(&r, ctx->io_res.p, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (r.status != ST_OK) return (int)r.status;
  if (r.out_len > dst_cap) return ZIPC_HIP_ERR_DST_TOO_SMALL;
  if (r.out_len) {
    HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->io_dst.p, r.out_len, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  *out_len = r.out_len;
  if (checksum) *checksum = r.checksum;
  return ZIPC_HIP_OK;
}

int zipc_hip_inflate(zipc_hip_ctx *ctx, const void *src, size_t len, int has_limit, size_t limit,
                     int crc_op, void *dst, size_t dst_cap, size_t *out_len, uint32_t *checksum) {
  return one_stream(ctx, true, src, len, has_limit, limit, 0, crc_op, dst, dst_cap, out_len, checksum);
}

int zipc_hip_deflate(zipc_hip_ctx *ctx, const void *src, size_t len, int level, int crc_op, void *dst,
                     size_t dst_cap, size_t *out_len, uint32_t *checksum) {
  if (level < 0 || level > 3) return ZIPC_HIP_ERR_INVALID_ARG;
  return one_stream(ctx, false, src, len, 0, 0, level, crc_op, dst, dst_cap, out_len, checksum);
}

// ---- the many-stream forms' way back: a sub-batch's outputs end to end, written by a kernel ----------------
// What a sub-batch made goes into the pinned host buffer by a KERNEL's stores, one output behind the other on 16-byte
// boundaries, not by the copy engine:
//  * how many bytes that is is known on the device when the kernels are through -- deflate's destination slots are as
//    large as the caller's capacities (the bound: more than the source), what is in them is half of that or less; an
//    engine copy's size would have to come from the host, which would have to wait for the results first;
//  * on this pool an engine copy out beside an engine copy in runs at a third of the bus whenever no kernel happens to
//    be running (tools/probes/host_copy.hip, profiles/r05_host_copy.txt: 256 MiB each way 13.4 / 14.1 ms, 4.8 / 5.5 with
//    a kernel spinning beside them; a kernel's stores beside an engine copy in: 5.3 / 6.3): the calls took 8 or 15 ms,
//    30 or 55, from one process to the next.
// The price: stores that wait for the bus hold up the memory path they share with everybody else (the same probe: a
// kernel that copies device memory takes 2.9 ms instead of 1.5 beside 8 such workgroups, 5.9 beside 64), so the kernel
// is as few workgroups as fill the bus.  The host makes the same sums from the results (many_streams below).

// (the host makes the same sums: zd_host::packed_size, host_pipeline.h)
__device__ static inline uint64_t packed_size(uint32_t status, uint64_t out_len, uint64_t dst_cap) {
  return status == ST_OK && out_len <= dst_cap ? (out_len + 15) / 16 * 16 : 0;
}

// off[i] = base + the packed sizes of streams [0, i), i = 0 .. n (one workgroup)
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const StreamDesc *descs, const StreamResult *res, uint32_t n,
                                                            uint64_t base, uint64_t *off) {
  __shared__ uint64_t part[1024];
  const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  uint64_t sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += packed_size(res[i].status, res[i].out_len, descs[i].dst_cap);
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d *= 2) {
    const uint64_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint64_t at = base + part[threadIdx.x] - sum;
  for (uint32_t i = lo; i < hi; i++) {
    off[i] = at;
    at += packed_size(res[i].status, res[i].out_len, descs[i].dst_cap);
  }
  if (threadIdx.x == 1023) off[n] = base + part[1023];
}

// Workgroup w of G moves the w-th part of the packed bytes (parts of whole 4 KiB): the stream its part begins in is
// found by bisection of off[], the next ones follow; every thread moves 16 bytes at a time, four loads in flight (slots
// begin on 256-byte boundaries).
__global__ __launch_bounds__(256) void pack_copy_kernel(const uint8_t *dst_arena, uint8_t *pack_arena, const StreamDesc *descs,
                                                        const uint64_t *off, uint32_t n, uint64_t base) {
  const uint64_t total_end = off[n];
  const uint64_t per = ((total_end - base + gridDim.x - 1) / gridDim.x + 4095) / 4096 * 4096;
  uint64_t pos = base + blockIdx.x * per;
  if (pos >= total_end) return;
  const uint64_t end = total_end - pos < per ? total_end : pos + per;
  uint32_t a = 0, b = n;  // the last stream that begins at or before pos
  while (b - a > 1) {
    const uint32_t m = a + (b - a) / 2;
    if (off[m] <= pos) a = m; else b = m;
  }
  for (uint32_t s = a; s < n && pos < end; s++) {
    const uint64_t s_beg = off[s], s_end = off[s + 1] < end ? off[s + 1] : end;
    if (s_end <= pos) continue;  // (a stream with nothing to hand over)
    const uint4 *from = (const uint4 *)(dst_arena + descs[s].dst_off + (pos - s_beg));
    uint4 *to = (uint4 *)(pack_arena + pos);
    const uint64_t n16 = (s_end - pos) / 16;
    uint64_t i = threadIdx.x;
    for (; i + 768 < n16; i += 1024) {
      const uint4 v0 = from[i], v1 = from[i + 256], v2 = from[i + 512], v3 = from[i + 768];
      to[i] = v0; to[i + 256] = v1; to[i + 512] = v2; to[i + 768] = v3;
    }
    for (; i < n16; i += 256) to[i] = from[i];
    pos = s_end;
  }
}

// n host-resident streams through the batch kernels: arenas are the context's
// staging buffers, streams packed at 256-byte aligned offsets
static int many_streams(zipc_hip_ctx *ctx, bool is_inflate, size_t n, const void *const *src, const size_t *src_len,
                        const size_t *limit, int level, int crc_op, void *const *dst, const size_t *dst_cap,
                        zipc_hip_stream_result *results, bool want_bytes = true) {
  if (!ctx || (n && (!src || !src_len || (!dst && want_bytes) || !dst_cap || !results))) return ZIPC_HIP_ERR_INVALID_ARG;
  if (crc_op < 0 || crc_op > 3 || level < 0 || level > 3 || n > 0x7FFFFFFFull) return ZIPC_HIP_ERR_INVALID_ARG;
  if (n == 0) return ZIPC_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const bool timing = zd::tuning().host_timing;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  std::vector<StreamDesc> descs(n);
  uint64_t so = 0, dof = 0;
  size_t max_src = 0, max_cap = 0;
  for (size_t i = 0; i < n; i++) {
    if ((!src[i] && src_len[i]) || (want_bytes && !dst[i] && dst_cap[i])) return ZIPC_HIP_ERR_INVALID_ARG;
    StreamDesc &d = descs[i];
    memset(&d, 0, sizeof d);
    d.src_off = so; d.src_len = src_len[i]; d.dst_off = dof; d.dst_cap = dst_cap[i];
    if (limit) { d.limit = limit[i]; d.flags = STREAM_HAS_LIMIT; }
    so += (src_len[i] + 255) / 256 * 256 + 256;
    dof += (dst_cap[i] + 255) / 256 * 256 + 256;
    max_src = src_len[i] > max_src ? src_len[i] : max_src;
    max_cap = dst_cap[i] > max_cap ? dst_cap[i] : max_cap;
  }
  if (!is_inflate && max_src > MAX_STREAM_LEN) return ZIPC_HIP_ERR_INVALID_ARG;  // (inflate reports it per stream)
  // The batch is cut into K sub-batches, and sub-batch g goes through
  //   gather (host threads, into pinned memory) -> copy in (the engine, queue copy_in, in runs of 16 MiB as they are
  //   gathered) -> kernels (the context's queue) -> the way back (the kernel above, queue copy_out) -> scatter (host threads)
  // on its own, so the bus and the kernels of one sub-batch run under the host memcpys of the others; PCIe is full
  // duplex and the kernels do not touch it.  This thread gathers and feeds the device; a second one (`taker` below)
  // waits for what comes back and scatters it with threads of its own, so the first sub-batch's results are in the
  // caller's buffers while the last one's sources are still being gathered.  Thousands of small pageable copies -- the
  // first version of this function -- cost far more than the kernels.
  // K (ZIPC_HIP_HOST_CHUNKS): 4, or 6 from a GiB of staging on; fewer when sub-batches would get too small to fill the
  // chip (under 1024 streams AND under 64 MiB of sources).  The first and the last sub-batch are half as large as the others: the first is what the bus and the kernels
  // wait for before they have anything to do, the last what the caller waits for when everything else is through.
  // (profiles/r05_host_forms_sweep.txt: 4096 x 64 KiB: 3 / 4 / 5 / 6 sub-batches deflate 9.9 / 9.8 / 9.4 / 10.0 ms,
  // inflate 8.3 / 8.9 / 8.9 / 9.5; 16 384 x 64 KiB: 34.0 / 31.2 / 29.8 / 30.2 and 29.4 / 27.6 / 26.6 / 25.7.)
  size_t K = host_chunks(so + dof);
  {  // a sub-batch holds host_chunk_min streams, or as many source bytes as that many streams of 64 KiB (long members)
    const uint64_t least = (uint64_t)zd::tuning().host_chunk_min;
    while (K > 1 && n / K < least && so / K < least * 65536) K--;
  }
  std::vector<size_t> cut(K + 1, n);
  cut[0] = 0;
  const bool taper = K >= 3;
  const size_t shares = taper ? 2 * K - 2 : K;
  for (size_t g = 1, i = 0; g < K; g++) {
    const size_t before = taper ? 2 * g - 1 : g;  // shares of sub-batches [0, g)
    while (i < n && descs[i].src_off < so / shares * before) i++;
    cut[g] = i;
  }
  size_t n_max = 0, total_max = 0;
  for (size_t g = 0; g < K; g++) {
    size_t t = 0;
    for (size_t i = cut[g]; i < cut[g + 1]; i++) t += src_len[i];
    n_max = cut[g + 1] - cut[g] > n_max ? cut[g + 1] - cut[g] : n_max;
    total_max = t > total_max ? t : total_max;
  }
  const bool packed = zd::tuning().host_pack && want_bytes;  // (false: whole destination slots by the copy engine)
  // everything is allocated before the first sub-batch is under way (growing a buffer
  // synchronises the stream)
  HIP_TRY(ctx, ctx->ensure(ctx->io_src, so + 64));
  HIP_TRY(ctx, ctx->ensure(ctx->io_dst, dof + 64));
  HIP_TRY(ctx, ctx->ensure(ctx->io_desc, n * sizeof(StreamDesc)));
  HIP_TRY(ctx, ctx->ensure(ctx->io_res, n * sizeof(StreamResult)));
  if (packed) HIP_TRY(ctx, ctx->ensure(ctx->io_pack_off, (n + K + 1) * sizeof(uint64_t)));
  HIP_TRY(ctx, ctx->ensure_pinned(ctx->pin_src, so + 64));
  if (want_bytes) HIP_TRY(ctx, ctx->ensure_pinned(ctx->pin_dst, dof + 64));
  HIP_TRY(ctx, ctx->ensure_pinned(ctx->pin_res, n * sizeof(StreamResult)));
  if (!is_inflate) {
    const int st = zipc_hip_reserve(ctx, n_max, max_src, total_max);
    if (st) return st;
  } else {
    HIP_TRY(ctx, ctx->ensure(ctx->inflate_scratch, n_max * INFLATE_SCRATCH_PER_STREAM));
  }
  if (crc_op == ZIPC_HIP_CRC_CRC32) {
    const size_t longest = is_inflate ? max_cap : max_src;
    size_t segs = (longest + CRC_SEG_BYTES - 1) / CRC_SEG_BYTES;
    HIP_TRY(ctx, ctx->ensure(ctx->crc_partials, n_max * (segs ? segs : 1) * sizeof(uint32_t)));
  }
  if (!ctx->copy_in) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
  if (!ctx->copy_out) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
  EventSet ev_in, ev_k, ev_out;
  HIP_TRY(ctx, ev_in.make(K, timing));
  HIP_TRY(ctx, ev_k.make(K, timing));
  HIP_TRY(ctx, ev_out.make(K, timing));
  EventSet ev_t;  // timing: the call's begin on the device, a sub-batch's first copy in, its kernels' begin
  if (timing) HIP_TRY(ctx, ev_t.make(1 + 2 * K, true));
  // earlier work of this context (the previous call's kernels read io_src / io_desc; a call that
  // failed half way may have left copies on the two copy streams) first
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_in));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_out));
  const double ms_setup = since(t_begin);
  static_assert(sizeof(StreamResult) == sizeof(zipc_hip_stream_result), "result layout");
  auto dst_end = [&](size_t i) { return i < n ? descs[i].dst_off : dof; };

  // ---- the device's part of the pipeline (host_pipeline.h Device): copies, kernels and events on three queues.  From
  // begin() on, work is in flight that reads `descs` and the pinned buffers and records into the event sets above:
  // many_pipeline returns only when its second thread is through, and after a failure all three queues are waited for
  // below before anything is freed or the next call reuses the buffers.
  struct Dev {
    zipc_hip_ctx *ctx;
    bool is_inflate, timing, packed, want_bytes, first_batch = true;
    size_t n, max_src, max_cap;
    int level, crc_op;
    const size_t *src_len;
    const std::vector<StreamDesc> &descs;
    EventSet &ev_in, &ev_k, &ev_out, &ev_t;
    decltype(dst_end) &dst_end_of;
    std::string error;
#define PIPE_TRY(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      error = std::string(#expr) + ": " + hipGetErrorString(_e);                         \
      return ZIPC_HIP_ERR_HIP;                                                           \
    }                                                                                    \
  } while (0)
    int begin() {
      if (timing) PIPE_TRY(hipEventRecord(ev_t.ev[0], ctx->copy_in));
      PIPE_TRY(hipMemcpyAsync(ctx->io_desc.p, descs.data(), n * sizeof(StreamDesc), hipMemcpyHostToDevice, ctx->copy_in));
      return ZIPC_HIP_OK;
    }
    int send(size_t g, bool first, uint64_t from, uint64_t to) {
      if (timing && first) PIPE_TRY(hipEventRecord(ev_t.ev[1 + 2 * g], ctx->copy_in));
      PIPE_TRY(hipMemcpyAsync((uint8_t *)ctx->io_src.p + from, (const uint8_t *)ctx->pin_src.p + from, to - from,
                              hipMemcpyHostToDevice, ctx->copy_in));
      return ZIPC_HIP_OK;
    }
    int sent(size_t g) {
      PIPE_TRY(hipEventRecord(ev_in.ev[g], ctx->copy_in));
      return ZIPC_HIP_OK;
    }
    int launch(size_t g, size_t lo, size_t hi) {
      PIPE_TRY(hipStreamWaitEvent(ctx->stream, ev_in.ev[g], 0));
      if (timing) PIPE_TRY(hipEventRecord(ev_t.ev[2 + 2 * g], ctx->stream));
      zipc_hip_stream_desc *dd = (zipc_hip_stream_desc *)ctx->io_desc.p + lo;
      zipc_hip_stream_result *dr = (zipc_hip_stream_result *)ctx->io_res.p + lo;
      size_t total_g = 0;
      for (size_t i = lo; i < hi; i++) total_g += src_len[i];
      int st;
      if (is_inflate)  // (with the descriptors it has on the host: no read-back, nothing waited for unless a stream goes by blocks)
        st = inflate_batch_impl(ctx, ctx->io_src.p, ctx->io_dst.p, dd, dr, hi - lo, max_cap, crc_op, descs.data() + lo, first_batch);
      else
        st = zipc_hip_deflate_batch(ctx, ctx->io_src.p, ctx->io_dst.p, dd, dr, hi - lo, max_src, total_g, level, crc_op);
      first_batch = false;
      if (st) { error = ctx->last_error; return st; }
      PIPE_TRY(hipMemcpyAsync((StreamResult *)ctx->pin_res.p + lo, dr, (hi - lo) * sizeof(StreamResult),
                              hipMemcpyDeviceToHost, ctx->stream));
      const uint64_t c = dst_end_of(lo), e = dst_end_of(hi);
      uint64_t *off = packed ? (uint64_t *)ctx->io_pack_off.p + lo + g : nullptr;
      if (packed)
        ZD_LAUNCH(ctx, "pack_offsets", pack_offsets_kernel, dim3(1), dim3(1024), 0, (const StreamDesc *)dd,
                  (const StreamResult *)dr, (uint32_t)(hi - lo), c, off);
      PIPE_TRY(hipGetLastError());
      PIPE_TRY(hipEventRecord(ev_k.ev[g], ctx->stream));
      if (!want_bytes) {  // results only: they are on their way behind the kernels, nothing else comes back
        PIPE_TRY(hipEventRecord(ev_out.ev[g], ctx->stream));
        return ZIPC_HIP_OK;
      }
      PIPE_TRY(hipStreamWaitEvent(ctx->copy_out, ev_k.ev[g], 0));
      if (packed) {  // its stores ARE the copy back, of as many bytes as the device knows it made, beside the next sub-batch's kernels
        hipLaunchKernelGGL(pack_copy_kernel, dim3((unsigned)zd::tuning().host_pack_wgs), dim3(256), 0, ctx->copy_out,
                           (const uint8_t *)ctx->io_dst.p, (uint8_t *)ctx->pin_dst.p, (const StreamDesc *)dd,
                           (const uint64_t *)off, (uint32_t)(hi - lo), c);
        PIPE_TRY(hipGetLastError());
      } else {
        PIPE_TRY(hipMemcpyAsync((uint8_t *)ctx->pin_dst.p + c, (const uint8_t *)ctx->io_dst.p + c, e - c,
                                hipMemcpyDeviceToHost, ctx->copy_out));
      }
      PIPE_TRY(hipEventRecord(ev_out.ev[g], ctx->copy_out));
      return ZIPC_HIP_OK;
    }
    int wait_back(size_t g) {  // (the taker's thread)
      PIPE_TRY(hipSetDevice(ctx->device));
      PIPE_TRY(hipEventSynchronize(ev_out.ev[g]));  // (behind ev_k[g]: the results have landed too)
      return ZIPC_HIP_OK;
    }
#undef PIPE_TRY
  } dev{ctx, is_inflate, timing, packed, want_bytes, true, n, max_src, max_cap, level, crc_op, src_len, descs, ev_in, ev_k, ev_out, ev_t, dst_end, {}};

  zd_host::ManyJob<StreamDesc> job;
  job.n = n; job.src = src; job.src_len = src_len; job.dst = dst; job.dst_cap = dst_cap; job.results = results;
  job.descs = descs.data(); job.src_arena_end = so; job.dst_arena_end = dof;
  job.cut = cut; job.n_max = n_max; job.packed = packed; job.want_bytes = want_bytes;
  job.ahead = is_inflate && max_cap >= BLOCKS_BATCH_MIN_DST;
  job.h2d_bytes = zd::tuning().host_h2d_mib > 0 ? (uint64_t)zd::tuning().host_h2d_mib << 20 : 0;
  job.pin_src = (uint8_t *)ctx->pin_src.p; job.pin_dst = want_bytes ? (const uint8_t *)ctx->pin_dst.p : nullptr;
  job.pin_res = (const zipc_hip_stream_result *)ctx->pin_res.p;
  job.threads = host_threads();
  zd_host::ManyTimes times;
  std::string why;
  const int pst = zd_host::many_pipeline(job, dev, host_pools(), why, timing ? &times : nullptr);
  if (pst) {  // a batch call refused its arguments or a HIP call failed: the call fails as a whole
    (void)hipStreamSynchronize(ctx->copy_in);  // (sub-batches scattered before that stay where they are, with their results;
    (void)hipStreamSynchronize(ctx->stream);   //  every other entry of results[] carries the call's status and no bytes)
    (void)hipStreamSynchronize(ctx->copy_out);
    ctx->last_error = why;
    return pst;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_in));
  if (timing) {  // where each sub-batch was when: host clock from the call's begin, device clock from the first copy's begin
    fprintf(stderr, "zipc_hip %s_many n=%zu src_arena=%llu dst_arena=%llu ms: setup %.2f feed %.2f (of it gather %.2f) "
                    "scatter %.2f whole %.2f (threads %zu sub-batches %zu)\n",
            is_inflate ? "inflate" : "deflate", n, (unsigned long long)so, (unsigned long long)dof, ms_setup, times.ms_feed,
            times.ms_gather, times.ms_scatter, since(t_begin), host_threads(), K);
    for (size_t g = 0; g < K; g++) {
      if (cut[g] == cut[g + 1]) continue;
      float h0 = 0, h1 = 0, k0 = 0, k1 = 0, o1 = 0;
      (void)hipEventElapsedTime(&h0, ev_t.ev[0], ev_t.ev[1 + 2 * g]);
      (void)hipEventElapsedTime(&h1, ev_t.ev[0], ev_in.ev[g]);
      (void)hipEventElapsedTime(&k0, ev_t.ev[0], ev_t.ev[2 + 2 * g]);
      (void)hipEventElapsedTime(&k1, ev_t.ev[0], ev_k.ev[g]);
      (void)hipEventElapsedTime(&o1, ev_t.ev[0], ev_out.ev[g]);
      fprintf(stderr, "  sub-batch %zu (%zu streams): host gathered at %.2f, scatter %.2f - %.2f | device copy in %.2f - %.2f, "
                      "kernels %.2f - %.2f, back by %.2f\n",
              g, cut[g + 1] - cut[g], times.gathered[g], times.scatter_begin[g], times.scatter_end[g], h0, h1, k0, k1, o1);
    }
  }
  return ZIPC_HIP_OK;
}

// (host vectors sized by n: whatever they throw -- bad_alloc when memory runs out, length_error, system_error from a mutex
// or a thread -- stays on this side of the C boundary: the call fails as out of memory, says so in zipc_hip_last_error, and
// every entry of results[] is defined.  Only the setup before the pipeline can throw: many_pipeline itself does not.)
static int many_threw(zipc_hip_ctx *ctx, size_t n, zipc_hip_stream_result *results) {
  try { if (ctx) ctx->last_error = "zipc_hip: out of memory (or no thread) on the host while setting up a many-stream call"; } catch (...) {}
  if (results) for (size_t i = 0; i < n; i++) results[i] = zipc_hip_stream_result{ZIPC_HIP_ERR_NOMEM, 0, 0};
  return ZIPC_HIP_ERR_NOMEM;
}
int zipc_hip_deflate_many(zipc_hip_ctx *ctx, size_t n, const void *const *src, const size_t *src_len, int level,
                          int crc_op, void *const *dst, const size_t *dst_cap, zipc_hip_stream_result *results) {
  try { return many_streams(ctx, false, n, src, src_len, nullptr, level, crc_op, dst, dst_cap, results); }
  catch (...) { return many_threw(ctx, n, results); }
}
int zipc_hip_inflate_many(zipc_hip_ctx *ctx, size_t n, const void *const *src, const size_t *src_len,
                          const size_t *limit, int crc_op, void *const *dst, const size_t *dst_cap,
                          zipc_hip_stream_result *results) {
  try { return many_streams(ctx, true, n, src, src_len, limit, 0, crc_op, dst, dst_cap, results); }
  catch (...) { return many_threw(ctx, n, results); }
}
int zipc_hip_inflate_many_check(zipc_hip_ctx *ctx, size_t n, const void *const *src, const size_t *src_len,
                                const size_t *limit, int crc_op, const size_t *dst_cap, zipc_hip_stream_result *results) {
  try { return many_streams(ctx, true, n, src, src_len, limit, 0, crc_op, nullptr, dst_cap, results, false); }
  catch (...) { return many_threw(ctx, n, results); }
}

// zlib_decompress src/zipc_deflate.ml:720-740 (start = 0): header checks on the
// host (6 bytes of parsing), body through the inflate kernel with Adler-32
int zipc_hip_zlib_decompress(zipc_hip_ctx *ctx, const void *src, size_t len, int has_limit,
                             size_t limit, void *dst, size_t dst_cap, size_t *out_len,
                             uint32_t *adler, uint32_t *expect, uint32_t *found) {
  if (!ctx || (!src && len) || !out_len) return ZIPC_HIP_ERR_INVALID_ARG;
  *out_len = 0;
  const uint8_t *s = (const uint8_t *)src;
  if (len < 6) return ZIPC_HIP_ERR_CORRUPTED;
  const int cmf = s[0], flg = s[1];
  if ((256 * cmf + flg) % 31 != 0) return ZIPC_HIP_ERR_CORRUPTED;
  if ((cmf & 0x0F) != 8) return ZIPC_HIP_ERR_ZLIB_METHOD;
  if ((cmf >> 4) > 7) return ZIPC_HIP_ERR_ZLIB_WINDOW;
  if ((flg & 0x20) != 0) return ZIPC_HIP_ERR_ZLIB_DICT;
  const uint32_t e = ((uint32_t)s[len - 4] << 24) | ((uint32_t)s[len - 3] << 16) |
                     ((uint32_t)s[len - 2] << 8) | (uint32_t)s[len - 1];
  uint32_t f = 0;
  // the reference hands inflate the range [2, len-2) (src/zipc_deflate.ml:732)
  int st = zipc_hip_inflate(ctx, s + 2, len - 4, has_limit, limit,
                            ctx->adler_rfc1950 ? ZIPC_HIP_CRC_ADLER32_RFC1950 : ZIPC_HIP_CRC_ADLER32, dst, dst_cap,
                            out_len, &f);
  if (st) return st;
  if (expect) *expect = e;
  if (found) *found = f;
  if (e != f) { *out_len = 0; return ZIPC_HIP_ERR_CHECKSUM; }
  if (adler) *adler = f;
  return ZIPC_HIP_OK;
}

// zlib_compress src/zipc_deflate.ml:1262-1277 (start = 0)
int zipc_hip_zlib_compress(zipc_hip_ctx *ctx, const void *src, size_t len, int level, void *dst,
                           size_t dst_cap, size_t *out_len, uint32_t *adler) {
  if (!ctx || !dst || !out_len || level < 0 || level > 3) return ZIPC_HIP_ERR_INVALID_ARG;
  *out_len = 0;
  if (dst_cap < 6) return ZIPC_HIP_ERR_DST_TOO_SMALL;
  uint8_t *o = (uint8_t *)dst;
  const int cmf = (7 << 4) | 8;
  const int header = (cmf << 8) | (level << 6);
  const int flg = (header + 31 - (header % 31)) & 0xFF;
  o[0] = (uint8_t)cmf;
  o[1] = (uint8_t)flg;
  size_t body = 0;
  uint32_t a = 0;
  int st = zipc_hip_deflate(ctx, src, len, level,
                            ctx->adler_rfc1950 ? ZIPC_HIP_CRC_ADLER32_RFC1950 : ZIPC_HIP_CRC_ADLER32, o + 2, dst_cap - 6,
                            &body, &a);
  if (st) return st;
  o[2 + body] = (uint8_t)(a >> 24);
  o[3 + body] = (uint8_t)(a >> 16);
  o[4 + body] = (uint8_t)(a >> 8);
  o[5 + body] = (uint8_t)a;
  *out_len = body + 6;
  if (adler) *adler = a;
  return ZIPC_HIP_OK;
}

}  // extern "C"
