/* zipc_hip.h -- C ABI of the MI355X-native Zipc_deflate hot path.
 *
 * This is the drop-in boundary for the reference's src/zipc_deflate.ml: every
 * entry point below replaces one value of the reference's module signature
 * src/zipc_deflate.mli (cited per function).  The reference has no FFI of its
 * own -- the seam is that OCaml signature -- so an OCaml shim
 * (bindings/ocaml/, shown in INTEGRATION.md) marshals strings to these calls,
 * applies ?start as a pointer offset, supplies the reference's default level
 * (`Best, zipc_deflate.ml:817) and maps status codes to the reference's error
 * strings (zipc_hip_strerror).
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Status codes, never
 * exceptions.  The library never frees or retains caller memory.
 *
 * Two families:
 *   - host forms   (zipc_hip_crc32 ... zipc_hip_zlib_decompress): host
 *     pointers in, host buffers out; one stream = a batch of one.  They copy
 *     H2D/D2H around the same kernels as the batch forms.
 *   - batch forms  (zipc_hip_*_batch, zipc_hip_checksum_device): descriptors
 *     over DEVICE-resident arenas; nothing crosses PCIe.  These are what
 *     bench.py times.
 *
 * All compute happens in HIP kernels on gfx950.  There is no CPU fallback: with
 * no usable device every call returns ZIPC_HIP_ERR_NO_DEVICE / _HIP.
 */
#ifndef ZIPC_HIP_H
#define ZIPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZIPC_HIP_ABI_VERSION 1

/* Longest single stream, and largest destination capacity, in bytes: 4 GiB - 64 KiB (positions are
 * 32-bit inside the kernels).  The reference's own limit is OCaml's string length.
 * One exception, for inflate: a batch of ONE stream that begins with stored blocks -- what the
 * reference's encoder makes of incompressible data -- may be of any size (zipc_hip_inflate_batch with
 * max_dst_cap above this limit; checksum none or CRC-32): stored blocks of any lengths are copied with
 * 64-bit offsets (a run of equal blocks all at once, others by a walk over their headers), with the
 * reference's messages for a damaged header and for a block beyond ?decompressed_size; the first block
 * of another kind and what follows it has to fit the limit as a stream of its own. */
#define ZIPC_HIP_MAX_STREAM_LEN 0xFFFF0000ull

/* ---- status codes --------------------------------------------------------
 * 1..6 are the reference's Failure messages (zipc_deflate.ml:233,29,728-730,104) */
enum {
  ZIPC_HIP_OK = 0,
  ZIPC_HIP_ERR_CORRUPTED = 1,     /* "Corrupted data stream" */
  ZIPC_HIP_ERR_SIZE_EXCEEDED = 2, /* "Expected decompression size exceeded" */
  ZIPC_HIP_ERR_ZLIB_METHOD = 3,   /* "Unknown compression method (%d)" */
  ZIPC_HIP_ERR_ZLIB_WINDOW = 4,   /* "Window size too large" */
  ZIPC_HIP_ERR_ZLIB_DICT = 5,     /* "Preset dictionary unsupported" */
  ZIPC_HIP_ERR_CHECKSUM = 6,      /* "Checksum mismatch, expected %lx found %lx)" */
  /* boundary-only conditions (no reference counterpart) */
  ZIPC_HIP_ERR_DST_TOO_SMALL = 16, /* dst_cap too small and no decompressed_size
                                      limit was given: retry with a larger dst */
  ZIPC_HIP_ERR_HIP = 17,           /* a HIP runtime call failed */
  ZIPC_HIP_ERR_INVALID_ARG = 18,
  ZIPC_HIP_ERR_NO_DEVICE = 19,
  ZIPC_HIP_ERR_NOMEM = 20
};

/* crc_op (zipc_deflate.ml:210): which checksum the reference updates once per deflate block inside
 * inflate (over the output) / deflate (over the input).  Here Adler-32 is computed inside the codec
 * kernels block by block (its value depends on that chunking, see below); CRC-32, whose value does
 * not, is a separate pass of the checksum kernels over the same bytes once the codec kernel is done
 * (one more read of them; the result is the same word for word). */
enum {
  ZIPC_HIP_CRC_NOP = 0,
  ZIPC_HIP_CRC_CRC32 = 1,
  ZIPC_HIP_CRC_ADLER32 = 2,         /* Zipc_deflate.Adler_32 as the reference computes it (below) */
  ZIPC_HIP_CRC_ADLER32_RFC1950 = 3  /* RFC 1950's Adler-32: what zlib and everybody else computes */
};
/* The reference's Adler-32 takes a SIGNED 32-bit remainder after every 5552-byte chunk
 * (src/zipc_deflate.ml:95,196) and restarts its chunking at every deflate block (:688,1084): on
 * data whose running sum crosses 2^31 -- bytes above 0x7F in bulk: 6 KB of random bytes do -- its
 * value differs from RFC 1950's, and depends on how the bytes were cut into calls.  A drop-in has
 * to reproduce that (ZIPC_HIP_CRC_ADLER32, the default everywhere): zlib_compress then writes
 * trailers standard zlib REJECTS for such data, and zlib_decompress reports a checksum mismatch on
 * valid RFC 1950 streams of such data, exactly as the reference does.  Where interoperation with
 * zlib matters more than equality with the reference, ask for RFC 1950's value instead: crc_op
 * ZIPC_HIP_CRC_ADLER32_RFC1950 in the inflate / deflate forms, zipc_hip_set_adler_rfc1950() for the
 * zlib forms and zipc_hip_checksum_device of a context. */

/* type level (zipc_deflate.mli:123-125) */
enum {
  ZIPC_HIP_LEVEL_NONE = 0,
  ZIPC_HIP_LEVEL_FAST = 1,
  ZIPC_HIP_LEVEL_DEFAULT = 2,
  ZIPC_HIP_LEVEL_BEST = 3
};

/* ---- context ---------------------------------------------------------------
 * Owns a HIP stream, constant tables and device scratch (grown on demand and
 * reused).  One context per host thread; contexts are independent (the
 * reference is re-entrant with no shared state). */
typedef struct zipc_hip_ctx zipc_hip_ctx;

int zipc_hip_abi_version(void);
int zipc_hip_device_count(void);
int zipc_hip_create(zipc_hip_ctx **ctx, int device);
void zipc_hip_destroy(zipc_hip_ctx *ctx);
/* the hipStream_t all of this context's work is enqueued on */
void *zipc_hip_stream(zipc_hip_ctx *ctx);
int zipc_hip_synchronize(zipc_hip_ctx *ctx);
/* text of the last HIP runtime error seen by this context ("" if none) */
const char *zipc_hip_last_error(zipc_hip_ctx *ctx);
/* how the last inflate of ONE stream ran: the number of blocks it was decoded by, a wave per block (inflate.hip),
 * or 0 when the stream's one wave decoded it (short streams, streams that are not a chain of dynamic blocks,
 * anything that reports an error).  For tests and measurements; the results are the same either way. */
unsigned zipc_hip_last_inflate_blocks(zipc_hip_ctx *ctx);
/* 1 when this context builds hash chains by ordered LDS exchange (deflate.hip lz_chain_xchg_kernel): the probe run
 * by zipc_hip_create found that one LDS exchange instruction serves the lanes that hit one address in ascending lane
 * order, which is the order insert_hash (zipc_deflate.ml:1150-1152) inserts positions in.  0: the probe failed and the
 * context keeps the kernel that orders them itself (same links either way).  For tests and measurements. */
int zipc_hip_lds_exchange_ordered(zipc_hip_ctx *ctx);
/* What the guards of that property have seen so far (round 6): positions whose links BOTH chain kernels made and were
 * compared -- the create-time probe runs lz_chain_xchg_kernel itself, whole and by segments, on a stream made of runs,
 * short periods and few-symbol alphabets; the context's first deflate batch has its first streams (ZIPC_HIP_CHAIN_CHECK,
 * default 32, at most 16 MiB) chained by both kernels under that batch's load -- and how many differed.  A difference in
 * the first batch fails that batch's streams with ZIPC_HIP_ERR_HIP and moves the context to the ordering kernel.
 * Synchronises the context's stream. */
int zipc_hip_chain_check(zipc_hip_ctx *ctx, unsigned long long *compared, unsigned long long *differences);
/* Measurements only: the number of slices the batch forms cut a call into (side queues, api.hip batch_slices), for
 * every context of the process from now on; 0 = the default again (two slices of at least 2048 streams, or
 * ZIPC_HIP_SLICES).  bench.py times a kernel alone on the device with 1.  Results are the same for every value. */
void zipc_hip_debug_set_slices(long k);
/* the reference's message for a status (format strings kept verbatim) */
const char *zipc_hip_strerror(int status);

/* ---- per-kernel timing -----------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the
 * context stream; zipc_hip_kernel_times reports, per kernel name, launches and
 * total milliseconds since the last reset.  bench.py uses this for the
 * roofline's live per-launch duration. */
typedef struct {
  char name[48];
  uint64_t launches;
  double total_ms;
} zipc_hip_kernel_time;
int zipc_hip_set_profiling(zipc_hip_ctx *ctx, int enabled);
/* enabled != 0: zipc_hip_zlib_compress / _decompress and zipc_hip_checksum_device's Adler-32 of this
 * context follow RFC 1950 (see ZIPC_HIP_CRC_ADLER32_RFC1950); 0 (the default): the reference's. */
int zipc_hip_set_adler_rfc1950(zipc_hip_ctx *ctx, int enabled);
int zipc_hip_reset_kernel_times(zipc_hip_ctx *ctx);
int zipc_hip_kernel_times(zipc_hip_ctx *ctx, zipc_hip_kernel_time *out, size_t cap,
                          size_t *n);

/* ---- host forms ------------------------------------------------------------ */

/* Crc_32.string  (zipc_deflate.mli:46; zipc_deflate.ml:161) */
int zipc_hip_crc32(zipc_hip_ctx *ctx, const void *src, size_t len, uint32_t *crc);
/* Adler_32.string (zipc_deflate.mli:72; zipc_deflate.ml:203), signed-remainder
 * behaviour of zipc_deflate.ml:95,196 included */
int zipc_hip_adler32(zipc_hip_ctx *ctx, const void *src, size_t len, uint32_t *adler);

/* inflate / inflate_and_crc_32 / inflate_and_adler_32
 * (zipc_deflate.mli:79-102; zipc_deflate.ml:692-718).
 * has_limit != 0  <=>  ?decompressed_size = limit.  dst_cap >= limit required
 * then.  With has_limit == 0 the reference grows its buffer without bound; here
 * ZIPC_HIP_ERR_DST_TOO_SMALL asks the caller to retry with a larger dst. */
int zipc_hip_inflate(zipc_hip_ctx *ctx, const void *src, size_t len, int has_limit,
                     size_t limit, int crc_op, void *dst, size_t dst_cap,
                     size_t *out_len, uint32_t *checksum);

/* zlib_decompress (zipc_deflate.mli:104-118; zipc_deflate.ml:720-740).  On
 * ZIPC_HIP_ERR_CHECKSUM, *expect and *found hold the two Adler-32 values. */
int zipc_hip_zlib_decompress(zipc_hip_ctx *ctx, const void *src, size_t len,
                             int has_limit, size_t limit, void *dst, size_t dst_cap,
                             size_t *out_len, uint32_t *adler, uint32_t *expect,
                             uint32_t *found);

/* upper bound of the deflate output for len input bytes, any level */
size_t zipc_hip_deflate_bound(size_t len);
size_t zipc_hip_zlib_bound(size_t len);

/* deflate / crc_32_and_deflate / adler_32_and_deflate
 * (zipc_deflate.mli:128-150; zipc_deflate.ml:1247-1260).  level is explicit
 * here; the shim passes ZIPC_HIP_LEVEL_BEST when ?level is absent, which is
 * what the reference does (zipc_deflate.ml:817). */
int zipc_hip_deflate(zipc_hip_ctx *ctx, const void *src, size_t len, int level,
                     int crc_op, void *dst, size_t dst_cap, size_t *out_len,
                     uint32_t *checksum);

/* zlib_compress (zipc_deflate.mli:152-162; zipc_deflate.ml:1262-1277) */
int zipc_hip_zlib_compress(zipc_hip_ctx *ctx, const void *src, size_t len, int level,
                           void *dst, size_t dst_cap, size_t *out_len,
                           uint32_t *adler);

/* Many independent streams held in HOST memory through the batch kernels in one
 * go: stream i is src[i][0 .. src_len[i]) -> dst[i] (capacity dst_cap[i]);
 * results[i] (host memory, declared below) gets its status, length and checksum.
 * This is what a caller that handles the members of an archive together uses
 * instead of n calls of the single-stream forms -- Zipc.File.deflate_of_binary_string
 * / to_binary_string over all members (src/zipc.ml:180-186,208-231).  limit may
 * be NULL (no ?decompressed_size for any stream); a stream whose output does not
 * fit reports ZIPC_HIP_ERR_DST_TOO_SMALL (or the reference's size message when a
 * limit is given) in its own result.  The call itself fails only for bad
 * arguments or HIP errors.
 * Staging: the call runs as a pipeline of a few sub-batches.  Each is gathered into
 * pinned memory the context keeps (sized to the batch; the first call of a size pays
 * for pinning it) by a few host threads the library keeps, copied to the device as
 * it is gathered, run, brought back -- the outputs end to end, by a kernel that
 * writes the pinned memory -- and scattered to the caller's buffers by a second
 * thread, so bus copies and kernels of one sub-batch run under the host memcpys of
 * the others.  A call that FAILS (bad arguments, a HIP error, no memory or thread on the host) may have filled some of
 * the caller's buffers, and results[] is defined all the same: the streams of sub-batches that had come back keep their
 * results, every other stream carries the call's status and out_len 0; zipc_hip_last_error says what happened.  No C++
 * exception crosses this boundary.  The staging threads are shared by the process's contexts and joined when the last
 * context is destroyed; a forked child makes its own (csrc/host_pipeline.h, which tests/test_sanitizers.py runs under
 * the thread and address sanitizers with host threads standing in for the device).
 * Environment, read once per process: ZIPC_HIP_HOST_THREADS (default 8 or the core
 * count), ZIPC_HIP_HOST_CHUNKS (sub-batches, default 4, 6 from a GiB of staging on;
 * fewer when a sub-batch would hold under 1024 streams), ZIPC_HIP_HOST_TIMING=1
 * (where each sub-batch was when, on stderr); csrc/tuning.h has the rest. */
struct zipc_hip_stream_result_s;
int zipc_hip_deflate_many(zipc_hip_ctx *ctx, size_t n, const void *const *src, const size_t *src_len,
                          int level, int crc_op, void *const *dst, const size_t *dst_cap,
                          struct zipc_hip_stream_result_s *results);
int zipc_hip_inflate_many(zipc_hip_ctx *ctx, size_t n, const void *const *src, const size_t *src_len,
                          const size_t *limit, int crc_op, void *const *dst, const size_t *dst_cap,
                          struct zipc_hip_stream_result_s *results);
/* The same decode with nothing brought back but the results: every stream is inflated into the context's device arena
 * (dst_cap[i] bytes of room each), its checksum taken there, and results[i] = {status, checksum, out_len}.  What a caller
 * that TESTS an archive needs -- File.to_binary_string of every member for its Ok / Error alone, the reference's
 * `zipc unzip -t` (test/zipc_tool.ml:635-660 check_archive) -- without the decompressed bytes crossing the bus. */
int zipc_hip_inflate_many_check(zipc_hip_ctx *ctx, size_t n, const void *const *src, const size_t *src_len,
                                const size_t *limit, int crc_op, const size_t *dst_cap,
                                struct zipc_hip_stream_result_s *results);

/* ---- batch forms (device-resident) ----------------------------------------- */

/* One independent stream: bytes [src_off, src_off+src_len) of the source arena
 * go to [dst_off, dst_off+dst_cap) of the destination arena.  For inflate,
 * limit is ?decompressed_size when flags & ZIPC_HIP_STREAM_HAS_LIMIT. */
typedef struct {
  uint64_t src_off;
  uint64_t src_len;
  uint64_t dst_off;
  uint64_t dst_cap;
  uint64_t limit;
  uint32_t flags;
  uint32_t reserved;
} zipc_hip_stream_desc;
#define ZIPC_HIP_STREAM_HAS_LIMIT 1u

typedef struct zipc_hip_stream_result_s {
  uint32_t status;   /* ZIPC_HIP_OK or an error code */
  uint32_t checksum; /* per crc_op, finished (0 for NOP) */
  uint64_t out_len;  /* bytes produced at dst_off */
} zipc_hip_stream_result;

/* All pointers are DEVICE pointers (descs and results too).  Work is enqueued
 * on the context stream and NOT synchronised: call zipc_hip_synchronize (or
 * synchronise the stream) before reading results.  Two exceptions, both zipc_hip_inflate_batch: ONE stream with
 * max_dst_cap above ZIPC_HIP_MAX_STREAM_LEN (a stream of stored blocks beyond 4 GiB, below), and a call whose
 * max_dst_cap is 256 KiB and more (one stream: 40 KiB of input and more): the library reads the descriptors back, and
 * the call's long streams -- all of a few, the long ones among many short ones, none of thousands of equal ones: chosen
 * by what each way costs -- are decoded by a wave per BLOCK, side by side: block starts searched for, the blocks walked
 * at once, copies resolved afterwards (a MiB in 1.3-1.9 ms instead of 10-30, 64 of them in 5 instead of 17).  That path
 * reads a few words back between its steps and so synchronises the stream itself.
 * Results, messages and limits are the same whichever way a stream is decoded (zipc_hip_last_inflate_blocks tells how
 * many blocks of the last call went that way; ZIPC_HIP_INFLATE_BLOCKS=0 in the environment keeps every stream on its
 * one wave).  Bits of flags other than ZIPC_HIP_STREAM_HAS_LIMIT must be zero: a stream whose descriptor has one set
 * reports ZIPC_HIP_ERR_INVALID_ARG and nothing is written to its destination. */
/* max_dst_cap: upper bound of dst_cap over the batch (sizes the CRC-32 pass). A stream
 * whose output is longer than that reports ZIPC_HIP_ERR_INVALID_ARG in its result when a
 * CRC-32 is asked for (its checksum would cover only a part). A descriptor with src_len or
 * dst_cap above ZIPC_HIP_MAX_STREAM_LEN reports ZIPC_HIP_ERR_INVALID_ARG in its own result. */
int zipc_hip_inflate_batch(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena,
                           const zipc_hip_stream_desc *d_descs,
                           zipc_hip_stream_result *d_results, size_t n_streams,
                           size_t max_dst_cap, int crc_op);

/* max_src_len: upper bound of src_len over the batch (sizes the kernels' grids);
 * total_src_len: sum of src_len over the batch (sizes the scratch). Both are checked on the
 * device against the descriptors: if a stream is longer than max_src_len or the sum exceeds
 * total_src_len, EVERY stream of the batch reports ZIPC_HIP_ERR_INVALID_ARG and nothing is
 * compressed. max_src_len above ZIPC_HIP_MAX_STREAM_LEN fails the call; a single descriptor
 * with src_len or dst_cap above it reports ZIPC_HIP_ERR_INVALID_ARG in its own result only. */
int zipc_hip_deflate_batch(zipc_hip_ctx *ctx, const void *d_src_arena, void *d_dst_arena,
                           const zipc_hip_stream_desc *d_descs,
                           zipc_hip_stream_result *d_results, size_t n_streams,
                           size_t max_src_len, size_t total_src_len, int level,
                           int crc_op);

/* CRC-32 and Adler-32 of one device buffer (Crc_32.string + Adler_32.string).  Both asked for: ONE pass over the bytes
 * leaves the CRC-32 partials and the Adler-32 chunk sums (len bytes of traffic; ZIPC_HIP_CHECKSUM_FUSED=0 keeps the two
 * passes of rounds 1-2), then the two short finishes.  d_out receives {crc32, adler32}.  Either selector may be 0 to
 * skip that checksum. */
int zipc_hip_checksum_device(zipc_hip_ctx *ctx, const void *d_buf, size_t len,
                             int want_crc32, int want_adler32, uint32_t *d_out);

/* Tests only: the hash-chain links (zipc_deflate.ml:1150-1152, insert_hash: per position the distance to the previous
 * position of equal hash, 0 beyond 32768) of a batch of device-resident streams as ONE of the library's two chain
 * kernels makes them -- which = 0: by ordered LDS exchange (ZIPC_HIP_ERR_HIP where the context's probe failed), 1: by
 * the kernel that orders equal hashes itself -- copied to d_links (links_cap 16-bit slots, at least
 * zipc_hip_debug_chain_positions(n_streams, total_src_len); slots no kernel writes are zero) with every stream's first
 * slot in d_pos_base (n_streams 64-bit words, may be null).  The batch must fit one pass (8 GiB of source).  The
 * suite requires the two kernels' links to be equal over the benchmark's whole batches (tests/test_gpu_limits.py). */
size_t zipc_hip_debug_chain_positions(size_t n_streams, size_t total_src_len);
int zipc_hip_debug_chain_links(zipc_hip_ctx *ctx, const void *d_src_arena, const zipc_hip_stream_desc *d_descs,
                               size_t n_streams, size_t max_src_len, size_t total_src_len, int which, void *d_links,
                               size_t links_cap, void *d_pos_base);

/* grow the context scratch up front (keeps hipMalloc out of timed regions) */
int zipc_hip_reserve(zipc_hip_ctx *ctx, size_t n_streams, size_t max_src_len,
                     size_t total_src_len);

#ifdef __cplusplus
}
#endif
#endif
