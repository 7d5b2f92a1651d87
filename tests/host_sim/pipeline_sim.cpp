// pipeline_sim.cpp -- the many-stream forms' host pipeline (zipc_amd/csrc/host_pipeline.h: the staging pools, the thread
// that feeds the device, the thread that takes results back) with host threads standing in for the device.  Built by
// tests/test_sanitizers.py with -fsanitize=thread and with -fsanitize=address,undefined: the code is the product's,
// unchanged; only the five Device callbacks differ from api.hip's.
//
// The stand-in device: three in-order queues (copy in, kernels, way back), each a thread that runs closures in the order
// they were enqueued, plus events one queue records and another waits for -- the shape of the three HIP queues.  Its
// "kernel" copies a stream's source to its destination slot reversed and XORed (so a stale or misplaced byte shows),
// "compressing" to a length that depends on the source, and the way back packs a sub-batch's outputs end to end exactly as
// pack_copy_kernel does.  A run is checked byte for byte on the caller's side.
//
//   pipeline_sim [seed] [calls] [nofork]      exit 0: every call's results and bytes were right
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>

#include <deque>
#include <functional>
#include <random>

#include "../../zipc_amd/csrc/host_pipeline.h"

using namespace zd_host;

struct Desc { uint64_t src_off, src_len, dst_off, dst_cap; };

// an in-order queue: a thread that runs closures one after the other
class Queue {
 public:
  Queue() : t_([this] { loop(); }) {}
  ~Queue() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    t_.join();
  }
  void push(std::function<void()> f) {
    { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(f)); }
    cv_.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> l(m_);
    idle_.wait(l, [&] { return q_.empty() && !busy_; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      cv_.wait(l, [&] { return stop_ || !q_.empty(); });
      if (q_.empty()) return;
      auto f = std::move(q_.front());
      q_.pop_front();
      busy_ = true;
      l.unlock();
      f();
      l.lock();
      busy_ = false;
      if (q_.empty()) idle_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_, idle_;
  std::deque<std::function<void()>> q_;
  bool stop_ = false, busy_ = false;
  std::thread t_;
};

// an event: recorded on one queue (a closure that sets it), waited for by another queue's closure or by a host thread
struct Event {
  std::mutex m;
  std::condition_variable cv;
  bool set = false;
  void fire() { { std::lock_guard<std::mutex> l(m); set = true; } cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return set; }); }
};

static uint64_t out_len_of(const uint8_t *s, uint64_t len) { return len == 0 ? 0 : len - (s[0] % 7) * (len / 16); }  // "compressed" size
static void transform(uint8_t *d, const uint8_t *s, uint64_t len, uint64_t out_len) {
  for (uint64_t i = 0; i < out_len; i++) d[i] = (uint8_t)(s[len - 1 - i] ^ 0x5A);
}

struct SimDevice {
  const ManyJob<Desc> &job;
  std::vector<uint8_t> dev_src, dev_dst;            // the "device" arenas
  std::vector<zipc_hip_stream_result> dev_res;
  uint8_t *pin_dst;                                 // pinned buffers the device writes
  zipc_hip_stream_result *pin_res;
  std::vector<Event> ev_in, ev_out;
  Queue copy_in, kernels, copy_out;
  int fail_launch_at = -1, fail_wait_at = -1;       // failure injection: sub-batch index
  std::string error;

  SimDevice(const ManyJob<Desc> &j, uint8_t *pd, zipc_hip_stream_result *pr)
      : job(j), dev_src(j.src_arena_end + 64), dev_dst(j.dst_arena_end + 64), dev_res(j.n), pin_dst(pd), pin_res(pr),
        ev_in(j.cut.size() - 1), ev_out(j.cut.size() - 1) {}
  int begin() { return ZIPC_HIP_OK; }
  int send(size_t, bool, uint64_t from, uint64_t to) {
    copy_in.push([this, from, to] { memcpy(dev_src.data() + from, job.pin_src + from, to - from); });
    return ZIPC_HIP_OK;
  }
  int sent(size_t g) { copy_in.push([this, g] { ev_in[g].fire(); }); return ZIPC_HIP_OK; }
  int launch(size_t g, size_t lo, size_t hi) {
    if ((int)g == fail_launch_at) { error = "injected: launch failed"; return ZIPC_HIP_ERR_HIP; }
    auto done_k = std::make_shared<Event>();
    kernels.push([this, g, lo, hi, done_k] {
      ev_in[g].wait();
      for (size_t i = lo; i < hi; i++) {
        const Desc &d = job.descs[i];
        const uint64_t ol = out_len_of(dev_src.data() + d.src_off, d.src_len);
        zipc_hip_stream_result r{ZIPC_HIP_OK, (uint32_t)(d.src_len * 2654435761u), ol};
        if (d.src_len % 13 == 5) r = zipc_hip_stream_result{ZIPC_HIP_ERR_CORRUPTED, 0, 0};  // a stream that fails on its own
        else if (ol > d.dst_cap) r.status = ZIPC_HIP_ERR_DST_TOO_SMALL;
        else transform(dev_dst.data() + d.dst_off, dev_src.data() + d.src_off, d.src_len, ol);
        dev_res[i] = r;
      }
      memcpy(pin_res + lo, dev_res.data() + lo, (hi - lo) * sizeof(zipc_hip_stream_result));
      done_k->fire();
      if (!job.want_bytes) ev_out[g].fire();  // results only: nothing else comes back (api.hip records ev_out behind the kernels)
    });
    if (!job.want_bytes) return ZIPC_HIP_OK;
    copy_out.push([this, g, lo, hi, done_k] {
      done_k->wait();
      uint64_t at = job.descs[lo].dst_off;
      for (size_t i = lo; i < hi; i++) {  // outputs end to end, as pack_copy_kernel lays them out
        const uint64_t sz = packed_size(dev_res[i].status, dev_res[i].out_len, job.descs[i].dst_cap);
        if (job.packed) { memcpy(pin_dst + at, dev_dst.data() + job.descs[i].dst_off, sz); at += sz; }
        else memcpy(pin_dst + job.descs[i].dst_off, dev_dst.data() + job.descs[i].dst_off, job.descs[i].dst_cap);
      }
      ev_out[g].fire();
    });
    return ZIPC_HIP_OK;
  }
  int wait_back(size_t g) {
    if ((int)g == fail_wait_at) { error = "injected: wait failed"; return ZIPC_HIP_ERR_HIP; }
    ev_out[g].wait();
    return ZIPC_HIP_OK;
  }
  void drain() { copy_in.drain(); kernels.drain(); copy_out.drain(); }
};

static int one_call(std::mt19937_64 &rng, Pools &pools, int inject) {
  const size_t n = 1 + rng() % 300;
  std::vector<std::vector<uint8_t>> srcs(n), dsts(n);
  std::vector<const void *> src(n);
  std::vector<void *> dst(n);
  std::vector<size_t> src_len(n), dst_cap(n);
  std::vector<Desc> descs(n);
  uint64_t so = 0, dof = 0;
  const bool long_ones = rng() % 4 == 0;
  for (size_t i = 0; i < n; i++) {
    size_t len = rng() % 5 == 0 ? 0 : 1 + rng() % 20000;
    if (long_ones && rng() % 16 == 0) len = (1 << 20) + rng() % (3 << 20);  // moved in pieces of 1 MiB
    srcs[i].resize(len);
    {  // (eight bytes a draw: the sanitizers' builds are slow enough)
      size_t k = 0;
      for (; k + 8 <= len; k += 8) { const uint64_t v = rng(); memcpy(srcs[i].data() + k, &v, 8); }
      for (; k < len; k++) srcs[i][k] = (uint8_t)rng();
    }
    const size_t cap = rng() % 11 == 0 ? len / 3 : len + 16;  // some too small
    dsts[i].assign(cap + 1, 0xEE);
    src[i] = srcs[i].data(); dst[i] = dsts[i].data(); src_len[i] = len; dst_cap[i] = cap;
    descs[i] = Desc{so, len, dof, cap};
    so += (len + 255) / 256 * 256 + 256;
    dof += (cap + 255) / 256 * 256 + 256;
  }
  ManyJob<Desc> job;
  const size_t K = 1 + rng() % 6;
  job.cut.assign(K + 1, n);
  job.cut[0] = 0;
  for (size_t g = 1; g < K; g++) job.cut[g] = std::min(n, job.cut[g - 1] + (size_t)(rng() % (2 * n / K + 1)));
  for (size_t g = 0; g < K; g++) job.n_max = std::max(job.n_max, job.cut[g + 1] - job.cut[g]);
  std::vector<uint8_t> pin_src(so + 64), pin_dst(dof + 64);
  std::vector<zipc_hip_stream_result> pin_res(n), results(n, zipc_hip_stream_result{0xDEAD, 0xDEAD, 0xDEAD});
  job.n = n; job.src = src.data(); job.src_len = src_len.data(); job.dst = dst.data(); job.dst_cap = dst_cap.data();
  job.results = results.data(); job.descs = descs.data(); job.src_arena_end = so; job.dst_arena_end = dof;
  job.want_bytes = rng() % 5 != 0;
  job.packed = job.want_bytes && rng() % 4 != 0; job.ahead = rng() % 3 == 0; job.h2d_bytes = rng() % 3 == 0 ? 0 : 1 << (12 + rng() % 10);
  job.pin_src = pin_src.data(); job.pin_dst = pin_dst.data(); job.pin_res = pin_res.data();
  job.threads = 1 + rng() % 6;
  SimDevice dev(job, pin_dst.data(), pin_res.data());
  if (inject == 1) dev.fail_launch_at = (int)(rng() % K);
  if (inject == 2) dev.fail_wait_at = (int)(rng() % K);
  std::string why;
  ManyTimes times;
  const int st = many_pipeline(job, dev, pools, why, rng() % 2 ? &times : nullptr);
  dev.drain();  // (api.hip waits for its three queues after a failure in the same way)
  size_t defined_upto = n;
  if (inject) {
    // a sub-batch that holds no streams cannot fail: the injection may not have fired
    if (st == ZIPC_HIP_OK) inject = 0;
    else if (st != ZIPC_HIP_ERR_HIP || why.find("injected") == std::string::npos) { fprintf(stderr, "wrong failure: %d %s\n", st, why.c_str()); return 1; }
  } else if (st != ZIPC_HIP_OK) { fprintf(stderr, "call failed: %d %s\n", st, why.c_str()); return 1; }
  for (size_t i = 0; i < n; i++) {
    const zipc_hip_stream_result &r = results[i];
    if (r.status == 0xDEAD) { fprintf(stderr, "results[%zu] was never written\n", i); return 1; }
    if (inject && r.status == ZIPC_HIP_ERR_HIP) { if (r.out_len != 0) return 1; defined_upto = std::min(defined_upto, i); continue; }
    if (inject && i > defined_upto) { fprintf(stderr, "a served stream behind an unserved one\n"); return 1; }
    const uint64_t ol = out_len_of(srcs[i].data(), src_len[i]);
    uint32_t want = ZIPC_HIP_OK;
    if (src_len[i] % 13 == 5) want = ZIPC_HIP_ERR_CORRUPTED;
    else if (ol > dst_cap[i]) want = ZIPC_HIP_ERR_DST_TOO_SMALL;
    if (r.status != want) { fprintf(stderr, "stream %zu: status %u, expected %u\n", i, r.status, want); return 1; }
    if (want != ZIPC_HIP_OK) { if (r.out_len != 0) return 1; continue; }
    if (r.out_len != ol || r.checksum != (uint32_t)(src_len[i] * 2654435761u)) { fprintf(stderr, "stream %zu: result differs\n", i); return 1; }
    if (!job.want_bytes) {  // a call for the results alone touches no destination
      for (size_t k = 0; k <= dst_cap[i]; k++) if (dsts[i][k] != 0xEE) { fprintf(stderr, "stream %zu: bytes written by a results-only call\n", i); return 1; }
      continue;
    }
    std::vector<uint8_t> expect(ol);
    transform(expect.data(), srcs[i].data(), src_len[i], ol);
    if (ol && memcmp(expect.data(), dsts[i].data(), ol) != 0) { fprintf(stderr, "stream %zu: bytes differ\n", i); return 1; }
    if (dsts[i][dst_cap[i]] != 0xEE) { fprintf(stderr, "stream %zu: wrote past its capacity\n", i); return 1; }
  }
  return 0;
}

int main(int argc, char **argv) {
  const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
  const int calls = argc > 2 ? atoi(argv[2]) : 40;
  const bool with_fork = !(argc > 3 && !strcmp(argv[3], "nofork"));  // (ThreadSanitizer cannot follow threads made after a fork of a threaded process)
  std::mt19937_64 rng(seed);
  Pools pools;
  pools.acquire();
  for (int c = 0; c < calls; c++) {
    const int inject = c % 5 == 3 ? 1 : (c % 5 == 4 ? 2 : 0);
    if (one_call(rng, pools, inject)) { fprintf(stderr, "call %d (seed %llu) failed\n", c, (unsigned long long)seed); return 1; }
  }
  // two callers at once, contexts of their own threads, the same pools (run() serialises jobs per pool)
  {
    int bad[2] = {0, 0};
    std::thread a([&] { std::mt19937_64 r(seed * 3 + 1); for (int c = 0; c < calls / 4 + 1; c++) bad[0] |= one_call(r, pools, 0); });
    std::thread b([&] { std::mt19937_64 r(seed * 3 + 2); for (int c = 0; c < calls / 4 + 1; c++) bad[1] |= one_call(r, pools, 0); });
    a.join(); b.join();
    if (bad[0] || bad[1]) { fprintf(stderr, "concurrent callers failed\n"); return 1; }
  }
  // a forked child holds the pools' memory but none of their threads: its calls run on the calling thread and finish
  if (with_fork) {
    fflush(nullptr);
    const pid_t pid = fork();
    if (pid == 0) {
      std::mt19937_64 r(seed + 99);
      int bad = 0;
      for (int c = 0; c < 3; c++) bad |= one_call(r, pools, 0);
      _exit(bad ? 1 : 0);
    }
    int status = 0;
    if (waitpid(pid, &status, 0) != pid || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "the forked child failed\n"); return 1; }
  }
  // the last user's release joins the staging threads; a later acquire + call makes new ones
  pools.release();
  pools.acquire();
  if (one_call(rng, pools, 0)) return 1;
  pools.release();
  printf("pipeline_sim: %d calls ok (seed %llu)\n", calls, (unsigned long long)seed);
  return 0;
}
