// host_pipeline.h -- the host half of the many-stream forms (zipc_hip_deflate_many / zipc_hip_inflate_many), free of HIP.
//
// Everything here runs on host threads: the pools behind the gathers and scatters, the copies that go around the cache,
// and the pipeline that moves a call's sub-batches through   gather -> copy in -> kernels -> way back -> scatter   with one
// thread feeding the device and a second one taking results back.  The device's part is five callbacks (struct Device
// below), so that the same code is compiled twice:
//   * into libzipc_hip.so (api.hip many_streams: the callbacks enqueue copies, kernels and events on three HIP queues);
//   * into tests/host_sim/pipeline_sim.cpp with g++ -fsanitize=thread / address, where three host threads with in-order
//     queues and memcpy stand in for the device (round 5's review: "host-side concurrency has no sanitizer coverage").
// Nothing in this file computes a result: bytes are moved, statuses copied.
#pragma once

#include <stdint.h>
#include <string.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zipc_hip.h"

namespace zd_host {

// The threads behind the host memcpys of the many-stream forms: made once per pool (a call starts one thread of its own,
// the taker), parked on a condition variable between jobs, handing out work in grains from one counter (a thread that
// loses its core for a while holds up one grain, not its whole share).  The caller of run() works too.
// Lifetime: the workers are JOINED by the destructor (round 5 detached them: a library that is unloaded, or a host
// that tears down at exit while a worker still parks on a destroyed condition variable, had no defined behaviour);
// pools_release() below destroys the pools when the last context goes.
// fork(): a child process holds the pool's memory but none of its threads.  The pool remembers who made it and runs a
// child's jobs on the calling thread alone (round 5: run() waited for workers that do not exist, for ever).
class HostPool {
 public:
  explicit HostPool(size_t workers) : owner_(getpid()) {
    threads_.reserve(workers);
    try {
      for (size_t t = 0; t < workers; t++) threads_.emplace_back([this] { worker(); });
    } catch (...) {
      // (the thread limit: the pool works with the threads it got -- they only ever see this fully built object,
      // because nothing is handed to them before run())
    }
  }
  ~HostPool() {
    if (getpid() == owner_) {
      {
        std::lock_guard<std::mutex> l(m_);
        stop_ = true;
        gen_++;
      }
      work_.notify_all();
      for (auto &t : threads_) t.join();
    } else {
      for (auto &t : threads_) t.detach();  // a forked child: there is nothing behind these handles to join
    }
  }
  HostPool(const HostPool &) = delete;
  HostPool &operator=(const HostPool &) = delete;
  size_t workers() const { return threads_.size(); }
  // f(i) for every i of [lo, hi); returns when all of them have run
  template <class F>
  void run(size_t lo, size_t hi, size_t grain, F f) {
    if (hi <= lo) return;
    if (threads_.empty() || hi - lo <= grain || getpid() != owner_) { for (size_t i = lo; i < hi; i++) f(i); return; }
    std::lock_guard<std::mutex> one_job(run_m_);
    {
      std::lock_guard<std::mutex> l(m_);
      fn_ = [](void *a, size_t i) { (*(F *)a)(i); };
      arg_ = &f; hi_ = hi; grain_ = grain;
      next_.store(lo, std::memory_order_relaxed);
      busy_ = threads_.size();
      gen_++;
    }
    work_.notify_all();
    take();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [&] { return busy_ == 0; });  // every worker has seen this job and left it: f and the fields are free again
  }

 private:
  void take() {
    for (;;) {
      const size_t a = next_.fetch_add(grain_, std::memory_order_relaxed);
      if (a >= hi_) return;
      const size_t b = hi_ - a < grain_ ? hi_ : a + grain_;
      for (size_t i = a; i < b; i++) fn_(arg_, i);
    }
  }
  void worker() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      work_.wait(l, [&] { return gen_ != seen; });
      seen = gen_;
      if (stop_) return;
      l.unlock();
      take();
      l.lock();
      if (--busy_ == 0) done_.notify_one();
    }
  }
  std::mutex run_m_, m_;
  std::condition_variable work_, done_;
  void (*fn_)(void *, size_t) = nullptr;
  void *arg_ = nullptr;
  size_t hi_ = 0, grain_ = 1, busy_ = 0;
  std::atomic<size_t> next_{0};
  uint64_t gen_ = 0;
  bool stop_ = false;
  const pid_t owner_;
  std::vector<std::thread> threads_;
};

// Two pools -- a call's gathers (the thread that feeds the device) and its scatters (the thread that takes results back)
// run side by side -- shared by the contexts of a process, made on first use, destroyed (threads joined) when the last
// context that acquired them is released: zipc_hip_create / zipc_hip_destroy.
class Pools {
 public:
  void acquire() { std::lock_guard<std::mutex> l(m_); users_++; }
  void release() {
    HostPool *a = nullptr, *b = nullptr;
    {
      std::lock_guard<std::mutex> l(m_);
      if (users_ == 0 || --users_ != 0) return;
      a = p_[0]; b = p_[1];
      p_[0] = p_[1] = nullptr;
    }
    delete a;  // (joins; no call is in flight: a call belongs to a context that is still alive)
    delete b;
  }
  HostPool &get(int which, size_t workers) {
    std::lock_guard<std::mutex> l(m_);
    if (p_[which] && made_by_ != getpid()) { p_[0] = p_[1] = nullptr; }  // a forked child: the parent's pools are not ours (leaked: their threads are not here to join)
    if (!p_[which]) { p_[which] = new HostPool(workers); made_by_ = getpid(); }
    return *p_[which];
  }

 private:
  std::mutex m_;
  size_t users_ = 0;
  HostPool *p_[2] = {nullptr, nullptr};
  pid_t made_by_ = 0;
};

// A copy whose destination is not read again by this core: stores that go around the cache (no line is fetched to be
// overwritten: two passes over memory instead of three, and the caches keep what they held).  The gathers and scatters
// of the many-stream forms are bound by the host's memory, beside the bus copies that read and write the same DIMMs:
// against memcpy the calls take 3-7 % less (profiles/r05_host_forms_sweep.txt).
static inline void copy_streaming(void *dst, const void *src, size_t len) {
#if defined(__x86_64__) && !defined(ZD_HOST_PLAIN_COPY)
  uint8_t *d = (uint8_t *)dst;
  const uint8_t *s = (const uint8_t *)src;
  if (len < 4096) { memcpy(d, s, len); return; }
  const size_t head = (64 - ((uintptr_t)d & 63)) & 63;
  memcpy(d, s, head);
  d += head; s += head; len -= head;
  const size_t body = len & ~(size_t)63;
  for (size_t i = 0; i < body; i += 64) {
    const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 16));
    const __m128i c = _mm_loadu_si128((const __m128i *)(s + i + 32)), e = _mm_loadu_si128((const __m128i *)(s + i + 48));
    _mm_stream_si128((__m128i *)(d + i), a);
    _mm_stream_si128((__m128i *)(d + i + 16), b);
    _mm_stream_si128((__m128i *)(d + i + 32), c);
    _mm_stream_si128((__m128i *)(d + i + 48), e);
  }
  _mm_sfence();
  memcpy(d + body, s + body, len - body);
#else
  memcpy(dst, src, len);  // (the sanitizer builds: their runtime sees memcpy, not vector stores)
#endif
}

// bytes a stream's output takes in the pinned buffer when a sub-batch's outputs lie end to end (api.hip pack_copy_kernel
// makes the same sums on the device)
static inline uint64_t packed_size(uint32_t status, uint64_t out_len, uint64_t dst_cap) {
  return status == ZIPC_HIP_OK && out_len <= dst_cap ? (out_len + 15) / 16 * 16 : 0;
}

// One call of a many-stream form, as the pipeline sees it.  Desc: anything with src_off / dst_off (where stream i's source
// and destination slots begin in the staging arenas; the pinned buffers mirror the arenas).
template <class Desc>
struct ManyJob {
  size_t n = 0;
  const void *const *src = nullptr;
  const size_t *src_len = nullptr;
  void *const *dst = nullptr;
  const size_t *dst_cap = nullptr;
  zipc_hip_stream_result *results = nullptr;  // the caller's
  const Desc *descs = nullptr;
  uint64_t src_arena_end = 0, dst_arena_end = 0;
  std::vector<size_t> cut;                    // sub-batch g holds streams [cut[g], cut[g + 1])
  size_t n_max = 0;                           // streams of the largest sub-batch
  bool want_bytes = true;                     // false: results only (status, checksum, length) -- nothing comes back, nothing is scattered, dst may be NULL
  bool packed = true;                         // a sub-batch's outputs come back end to end (from dst_off of its first stream on)
  bool ahead = false;                         // sub-batch g + 1 is gathered and sent before g's kernels are asked for
  uint64_t h2d_bytes = 0;                     // a sub-batch's sources are sent in runs of about this many bytes (0: one run)
  uint8_t *pin_src = nullptr;                 // pinned staging: sources as the arena holds them
  const uint8_t *pin_dst = nullptr;           // ... outputs as they come back
  const zipc_hip_stream_result *pin_res = nullptr;  // ... results as the device wrote them (n entries)
  size_t threads = 1;                         // host threads per pool, the caller included
};

// what the pipeline measured on the host (ms from the call's begin), for ZIPC_HIP_HOST_TIMING
struct ManyTimes {
  std::vector<double> gathered, scatter_begin, scatter_end;
  double ms_gather = 0, ms_scatter = 0, ms_feed = 0;
};

// The device's part of the pipeline.  Every callback returns a status of include/zipc_hip.h (0: fine) and, on failure,
// leaves what went wrong in `error`.  send / sent / launch are called by the feeding thread in order; wait_back by the
// taker, for sub-batches whose launch() has returned.
//   int begin();                                       once, before anything is sent (the descriptor table goes on its way)
//   int send(size_t g, bool first, uint64_t from, uint64_t to);   bytes [from, to) of pin_src -> the source arena, asynchronously
//   int sent(size_t g);                                every run of sub-batch g is under way
//   int launch(size_t g, size_t lo, size_t hi);        g's kernels and its way back into pin_dst / pin_res, asynchronously
//   int wait_back(size_t g);                           returns when g's outputs and results are in pinned memory
//   std::string error;

struct Piece { uint32_t stream; uint64_t at, len; };  // a long stream is moved in pieces of 1 MiB so that a few long members keep every thread busy too
constexpr uint64_t PIECE_BYTES = 1 << 20;

template <class LenOf>
static inline void pieces_of(size_t lo, size_t hi, LenOf len_of, std::vector<Piece> &out) {
  out.clear();
  for (size_t i = lo; i < hi; i++)
    for (uint64_t at = 0, L = len_of(i); at < L; at += PIECE_BYTES) out.push_back({(uint32_t)i, at, L - at < PIECE_BYTES ? L - at : PIECE_BYTES});
}

// Runs the call.  Returns the call's status; on failure `error` says why, the sub-batches that were scattered before the
// failure stay in the caller's buffers with their results, and EVERY OTHER entry of results[] is set to that status
// with out_len 0 (round 5 left them unwritten).  Never throws.  The caller still has to wait for the device's queues
// before it frees or reuses the staging buffers after a failure.
template <class Desc, class Device>
int many_pipeline(const ManyJob<Desc> &job, Device &dev, Pools &pools, std::string &error, ManyTimes *times = nullptr) {
  const size_t n = job.n, K = job.cut.size() - 1;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  auto src_end = [&](size_t i) { return i < n ? (uint64_t)job.descs[i].src_off : job.src_arena_end; };
  auto dst_end = [&](size_t i) { return i < n ? (uint64_t)job.descs[i].dst_off : job.dst_arena_end; };
  auto grain_of = [&](size_t count) {
    const size_t g = count / (job.threads * 8);
    return g < 1 ? (size_t)1 : (g > 16 ? (size_t)16 : g);
  };
  const size_t workers = job.threads > 0 ? job.threads - 1 : 0;

  // ---- what the two threads share: how many sub-batches have been enqueued, whether the feeding thread gave up, and
  // how many sub-batches the taker has handed to the caller
  std::mutex pm;
  std::condition_variable pcv;
  size_t fed = 0;
  bool gave_up = false;
  size_t taken = 0;  // (written by the taker, read after the join)
  std::string taker_error;
  int taker_status = ZIPC_HIP_OK, feed_status = ZIPC_HIP_OK;
  try {
    if (times) { times->gathered.assign(K, 0); times->scatter_begin.assign(K, 0); times->scatter_end.assign(K, 0); }

    // ---- the taker: sub-batch g is back -> its results as the caller gets them, where each output lies -> scatter
    auto taker = [&]() -> int {
      std::vector<Piece> pieces;
      std::vector<uint64_t> from(job.n_max);  // where a stream's output begins in pin_dst
      HostPool &pool = pools.get(1, workers);
      for (size_t g = 0; g < K; g++) {
        const size_t lo = job.cut[g], hi = job.cut[g + 1];
        {
          std::unique_lock<std::mutex> l(pm);
          pcv.wait(l, [&] { return fed > g || gave_up; });
          if (fed <= g) return ZIPC_HIP_OK;  // the feeding thread gave up before this one: its status is the call's
        }
        if (lo == hi) { taken = g + 1; continue; }
        const int st = dev.wait_back(g);
        if (st) { taker_error = dev.error; return st; }
        uint64_t at = dst_end(lo);
        for (size_t i = lo; i < hi; i++) {
          zipc_hip_stream_result r = job.pin_res[i];
          from[i - lo] = job.packed ? at : (uint64_t)job.descs[i].dst_off;
          at += packed_size(r.status, r.out_len, job.dst_cap[i]);
          if (r.status != ZIPC_HIP_OK) r.out_len = 0;
          else if (r.out_len > job.dst_cap[i]) { r.status = ZIPC_HIP_ERR_DST_TOO_SMALL; r.out_len = 0; }
          job.results[i] = r;
        }
        const auto t_sc = std::chrono::steady_clock::now();
        if (times) times->scatter_begin[g] = since(t_begin);
        if (!job.want_bytes) { taken = g + 1; continue; }
        pieces_of(lo, hi, [&](size_t i) { return job.results[i].status == ZIPC_HIP_OK ? (uint64_t)job.results[i].out_len : 0; }, pieces);
        pool.run(0, pieces.size(), grain_of(pieces.size()), [&](size_t j) {
          const Piece &p = pieces[j];
          copy_streaming((uint8_t *)job.dst[p.stream] + p.at, job.pin_dst + from[p.stream - lo] + p.at, p.len);
        });
        if (times) { times->ms_scatter += since(t_sc); times->scatter_end[g] = since(t_begin); }
        taken = g + 1;
      }
      return ZIPC_HIP_OK;
    };
    // (nothing that throws may leave this function while the taker runs: a joinable thread's destructor ends the process)
    std::thread taker_thread;
    try {
      taker_thread = std::thread([&] {
        try { taker_status = taker(); }
        catch (...) { taker_error = "zipc_hip: out of memory in the thread that takes the results back"; taker_status = ZIPC_HIP_ERR_NOMEM; }
      });
    } catch (...) {  // (no thread to be had; nothing is enqueued yet)
      error = "zipc_hip: could not start the thread that takes the results back";
      for (size_t i = 0; i < n; i++) job.results[i] = zipc_hip_stream_result{ZIPC_HIP_ERR_NOMEM, 0, 0};
      return ZIPC_HIP_ERR_NOMEM;
    }

    // ---- this thread: gather, copy in, kernels, the way back
    auto feed = [&]() -> int {
      int st = dev.begin();
      if (st) return st;
      std::vector<Piece> pieces;
      HostPool &pool = pools.get(0, workers);
      // sub-batch g's sources: gathered and sent in runs of streams of about h2d_bytes -- the bus starts on the first run
      // while the next is gathered
      auto gather_and_send = [&](size_t g) -> int {
        const size_t lo = job.cut[g], hi = job.cut[g + 1];
        for (size_t a = lo; a < hi;) {
          size_t b = a + 1;
          while (b < hi && (job.h2d_bytes == 0 || src_end(b) - src_end(a) < job.h2d_bytes)) b++;
          const auto t_g = std::chrono::steady_clock::now();
          pieces_of(a, b, [&](size_t i) { return (uint64_t)job.src_len[i]; }, pieces);
          pool.run(0, pieces.size(), grain_of(pieces.size()), [&](size_t j) {
            const Piece &p = pieces[j];
            copy_streaming(job.pin_src + job.descs[p.stream].src_off + p.at, (const uint8_t *)job.src[p.stream] + p.at, p.len);
          });
          if (times) times->ms_gather += since(t_g);
          const int s = dev.send(g, a == lo, src_end(a), src_end(b));
          if (s) return s;
          a = b;
        }
        if (times) times->gathered[g] = since(t_begin);
        return lo < hi ? dev.sent(g) : ZIPC_HIP_OK;
      };
      // Long members' inflate may go by blocks inside inflate_batch, which waits for the device on the way (api.hip
      // inflate_by_blocks): the NEXT sub-batch's sources are gathered and sent before this one's kernels are asked for, or
      // they would not leave the host before those kernels are through (256 x 1 MiB: 22.0 -> 21.4 ms: what is left is the
      // blocks' kernels, 7-8 ms a sub-batch of 128 MiB).  Everywhere else the kernels of a sub-batch are enqueued the
      // moment its sources are under way.
      if (job.ahead) { st = gather_and_send(0); if (st) return st; }
      for (size_t g = 0; g < K; g++) {
        const size_t lo = job.cut[g], hi = job.cut[g + 1];
        if (job.ahead ? g + 1 < K : true) { st = gather_and_send(job.ahead ? g + 1 : g); if (st) return st; }
        if (lo < hi) { st = dev.launch(g, lo, hi); if (st) return st; }
        {
          std::lock_guard<std::mutex> l(pm);
          fed = g + 1;
        }
        pcv.notify_all();
      }
      return ZIPC_HIP_OK;
    };
    const auto t_feed = std::chrono::steady_clock::now();
    try { feed_status = feed(); if (feed_status) error = dev.error; }
    catch (...) { error = "zipc_hip: out of memory while feeding the device"; feed_status = ZIPC_HIP_ERR_NOMEM; }
    if (times) times->ms_feed = since(t_feed);
    if (feed_status) {
      {
        std::lock_guard<std::mutex> l(pm);
        gave_up = true;
      }
      pcv.notify_all();
    }
    taker_thread.join();
  } catch (...) {  // (only the allocations before the taker's start can get here)
    error = "zipc_hip: out of memory";
    feed_status = ZIPC_HIP_ERR_NOMEM;
  }
  if (feed_status || taker_status) {  // a batch call refused its arguments or a device call failed: the call fails as a whole
    const int st = feed_status ? feed_status : taker_status;
    if (!feed_status) error = taker_error;
    // sub-batches scattered before that stay where they are, with their results; every other stream says why it has none
    for (size_t i = job.cut[taken < K ? taken : K]; i < n; i++) job.results[i] = zipc_hip_stream_result{(uint32_t)st, 0, 0};
    return st;
  }
  return ZIPC_HIP_OK;
}

}  // namespace zd_host
