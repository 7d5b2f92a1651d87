// zipc_deflate.cpp -- Zipc_deflate's functions over the C ABI (see zipc_deflate.hpp).
#include "zipc_deflate.hpp"

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>

namespace zipc_deflate {

// One context PER HOST THREAD (include/zipc_hip.h: a context owns one HIP stream and staging
// buffers that every call reuses, so it serves one thread at a time).  The reference module is
// re-entrant and a drop-in caller may well call it from several threads: each gets its own
// context on first use, on the device it chose (set_thread_device; the first of devices() otherwise),
// destroyed when the thread exits.  The many-stream forms spread a batch over ALL of devices().
namespace {
std::mutex g_devices_mu;
std::vector<int> g_devices;      // empty: not decided yet
bool g_devices_set = false;

std::vector<int> default_devices() {
  std::vector<int> d;
  if (const char *e = getenv("ZIPC_HIP_DEVICES")) {
    const char *p = e;
    while (*p) {
      char *end = nullptr;
      const long v = strtol(p, &end, 10);
      if (end == p) break;
      if (v >= 0) d.push_back((int)v);
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') break;
    }
  }
  if (d.empty()) {
    if (const char *one = getenv("ZIPC_HIP_DEVICE")) {  // a process pinned to one GPU (one process per GPU)
      const int v = atoi(one);
      if (v >= 0) return std::vector<int>(1, v);
    }
    const int n = zipc_hip_device_count();
    for (int i = 0; i < (n > 0 ? n : 1); i++) d.push_back(i);
  }
  return d;
}

struct ThreadContext {
  zipc_hip_ctx *ctx = nullptr;
  int device = -1;  // -1: the first of devices()
  int made_on = -1;
  int status = 0;
  ~ThreadContext() {
    if (ctx) zipc_hip_destroy(ctx);
  }
};
thread_local ThreadContext tc;

// One more context per entry of devices() for the many-stream forms: made on first use, kept for the
// life of the process (a context pins its staging buffers: ~100 ms the first time), one user at a time.
struct PoolContext {
  std::mutex mu;
  zipc_hip_ctx *ctx = nullptr;
  int device = -1;
};
std::mutex g_pool_mu;
std::vector<std::unique_ptr<PoolContext>> g_pool;
PoolContext &pool_context(std::size_t slot, int device) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  while (g_pool.size() <= slot) g_pool.emplace_back(new PoolContext());
  PoolContext &pc = *g_pool[slot];
  if (pc.ctx && pc.device != device) {  // the device list changed under it
    std::lock_guard<std::mutex> u(pc.mu);
    zipc_hip_destroy(pc.ctx);
    pc.ctx = nullptr;
  }
  pc.device = device;
  return pc;
}
}  // namespace

std::vector<int> devices() {
  std::lock_guard<std::mutex> g(g_devices_mu);
  if (!g_devices_set) { g_devices = default_devices(); g_devices_set = true; }
  return g_devices;
}
void set_devices(const std::vector<int> &list) {
  std::lock_guard<std::mutex> g(g_devices_mu);
  g_devices = list.empty() ? default_devices() : list;
  g_devices_set = true;
}
void set_thread_device(int device) { tc.device = device; }
int thread_device() { return tc.device >= 0 ? tc.device : devices().front(); }

zipc_hip_ctx *context() {
  const int want = thread_device();
  if (tc.ctx && tc.made_on != want) {
    zipc_hip_destroy(tc.ctx);
    tc.ctx = nullptr;
  }
  if (!tc.ctx) {
    tc.status = zipc_hip_create(&tc.ctx, want);
    tc.made_on = want;
  }
  if (!tc.ctx) throw std::runtime_error(std::string("zipc_hip_create: ") + zipc_hip_strerror(tc.status));
  return tc.ctx;
}

static std::string message(int status) { return zipc_hip_strerror(status); }
// statuses the reference cannot produce (no device, HIP failure, bad argument)
static bool is_library_failure(int status) { return status >= ZIPC_HIP_ERR_HIP; }
static void throw_if_library_failure(int status) {
  if (is_library_failure(status)) throw std::runtime_error(std::string("zipc_hip: ") + message(status));
}

std::string crc_error(uint32 expect, uint32 found) {
  char b[96];
  snprintf(b, sizeof b, "Checksum mismatch, expected %x found %x)", expect, found);
  return b;
}
static std::string hex(uint32 v) {
  char b[16];
  snprintf(b, sizeof b, "%x", v);
  return b;
}

Result<Unit> Crc_32::check(t expect, t found) {
  return equal(expect, found) ? Result<Unit>::Ok(Unit{}) : Result<Unit>::Error(crc_error(expect, found));
}
std::string Crc_32::pp(t crc) { return hex(crc); }
Crc_32::t Crc_32::string(const std::string &s, std::size_t start, std::size_t len) {
  const auto r = range(s, start, len);
  uint32 v = 0;
  const int st = zipc_hip_crc32(context(), s.data() + r.first, r.second, &v);
  if (st) throw std::runtime_error(std::string("zipc_hip_crc32: ") + message(st));
  return v;
}
Result<Unit> Adler_32::check(t expect, t found) {
  return equal(expect, found) ? Result<Unit>::Ok(Unit{}) : Result<Unit>::Error(crc_error(expect, found));
}
std::string Adler_32::pp(t crc) { return hex(crc); }
Adler_32::t Adler_32::string(const std::string &s, std::size_t start, std::size_t len) {
  const auto r = range(s, start, len);
  uint32 v = 0;
  const int st = zipc_hip_adler32(context(), s.data() + r.first, r.second, &v);
  if (st) throw std::runtime_error(std::string("zipc_hip_adler32: ") + message(st));
  return v;
}

// inflate_and_crc zipc_deflate.ml:692-709 through zipc_hip_inflate.  Without
// ?decompressed_size the reference's buffer starts at 3 * len (at least 1024,
// zipc_deflate.ml:19,552-555) and grows without bound: here the call is repeated
// with a doubled buffer while the library answers DST_TOO_SMALL.
static int inflate_raw(const char *p, std::size_t n, std::optional<std::size_t> decompressed_size, int crc_op,
                       std::string &out, uint32 &checksum) {
  std::size_t cap = decompressed_size ? *decompressed_size : (3 * n < 1024 ? 1024 : 3 * n);
  for (;;) {
    out.resize(cap);
    std::size_t out_len = 0;
    const int st = zipc_hip_inflate(context(), p, n, decompressed_size ? 1 : 0, decompressed_size ? *decompressed_size : 0,
                                    crc_op, cap ? &out[0] : nullptr, cap, &out_len, &checksum);
    if (st == ZIPC_HIP_ERR_DST_TOO_SMALL && !decompressed_size) {
      cap = cap < 1024 ? 2048 : cap * 2;
      continue;
    }
    throw_if_library_failure(st);
    out.resize(st == ZIPC_HIP_OK ? out_len : 0);
    return st;
  }
}

Result<std::string> inflate(const std::string &s, std::optional<std::size_t> decompressed_size, std::size_t start,
                            std::size_t len) {
  const auto r = range(s, start, len);
  std::string out;
  uint32 c = 0;
  const int st = inflate_raw(s.data() + r.first, r.second, decompressed_size, ZIPC_HIP_CRC_NOP, out, c);
  if (st) return Result<std::string>::Error(message(st));
  return Result<std::string>::Ok(std::move(out));
}
Result<std::pair<std::string, Crc_32::t>> inflate_and_crc_32(const std::string &s,
                                                             std::optional<std::size_t> decompressed_size,
                                                             std::size_t start, std::size_t len) {
  typedef Result<std::pair<std::string, Crc_32::t>> R;
  const auto r = range(s, start, len);
  std::string out;
  uint32 c = 0;
  const int st = inflate_raw(s.data() + r.first, r.second, decompressed_size, ZIPC_HIP_CRC_CRC32, out, c);
  if (st) return R::Error(message(st));
  return R::Ok({std::move(out), c});
}
Result<std::pair<std::string, Adler_32::t>> inflate_and_adler_32(const std::string &s,
                                                                 std::optional<std::size_t> decompressed_size,
                                                                 std::size_t start, std::size_t len) {
  typedef Result<std::pair<std::string, Adler_32::t>> R;
  const auto r = range(s, start, len);
  std::string out;
  uint32 c = 0;
  const int st = inflate_raw(s.data() + r.first, r.second, decompressed_size, ZIPC_HIP_CRC_ADLER32, out, c);
  if (st) return R::Error(message(st));
  return R::Ok({std::move(out), c});
}

ZlibResult zlib_decompress(const std::string &s, std::optional<std::size_t> decompressed_size, std::size_t start,
                           std::size_t len) {
  const auto r = range(s, start, len);
  ZlibResult z;
  std::size_t cap = decompressed_size ? *decompressed_size : (3 * r.second < 1024 ? 1024 : 3 * r.second);
  for (;;) {
    z.value.resize(cap);
    std::size_t out_len = 0;
    uint32 adler = 0, expect = 0, found = 0;
    const int st = zipc_hip_zlib_decompress(context(), s.data() + r.first, r.second, decompressed_size ? 1 : 0,
                                            decompressed_size ? *decompressed_size : 0, cap ? &z.value[0] : nullptr, cap,
                                            &out_len, &adler, &expect, &found);
    if (st == ZIPC_HIP_ERR_DST_TOO_SMALL && !decompressed_size) {
      cap = cap < 1024 ? 2048 : cap * 2;
      continue;
    }
    throw_if_library_failure(st);
    if (st == ZIPC_HIP_OK) {
      z.ok = true;
      z.value.resize(out_len);
      z.adler = adler;
      return z;
    }
    z.value.clear();
    if (st == ZIPC_HIP_ERR_CHECKSUM) {
      z.error.mismatch = std::make_pair(expect, found);
      z.error.message = crc_error(expect, found);
    } else if (st == ZIPC_HIP_ERR_ZLIB_METHOD) {  // failwithf "Unknown compression method (%d)" cm, zipc_deflate.ml:728
      char b[64];
      snprintf(b, sizeof b, "Unknown compression method (%d)", (int)((unsigned char)s[r.first] & 0x0F));
      z.error.message = b;
    } else {
      z.error.message = message(st);
    }
    return z;
  }
}

static int level_of(std::optional<level> l) { return l ? (int)*l : ZIPC_HIP_LEVEL_BEST; }  // Q2

static int deflate_raw(const char *p, std::size_t n, std::optional<level> lvl, int crc_op, std::string &out,
                       uint32 &checksum) {
  const std::size_t cap = zipc_hip_deflate_bound(n);
  out.resize(cap);
  std::size_t out_len = 0;
  const int st = zipc_hip_deflate(context(), p, n, level_of(lvl), crc_op, &out[0], cap, &out_len, &checksum);
  throw_if_library_failure(st);
  out.resize(st == ZIPC_HIP_OK ? out_len : 0);
  return st;
}

Result<std::string> deflate(const std::string &s, std::optional<level> lvl, std::size_t start, std::size_t len) {
  const auto r = range(s, start, len);
  std::string out;
  uint32 c = 0;
  const int st = deflate_raw(s.data() + r.first, r.second, lvl, ZIPC_HIP_CRC_NOP, out, c);
  if (st) return Result<std::string>::Error(message(st));
  return Result<std::string>::Ok(std::move(out));
}
Result<std::pair<Crc_32::t, std::string>> crc_32_and_deflate(const std::string &s, std::optional<level> lvl,
                                                             std::size_t start, std::size_t len) {
  typedef Result<std::pair<Crc_32::t, std::string>> R;
  const auto r = range(s, start, len);
  std::string out;
  uint32 c = 0;
  const int st = deflate_raw(s.data() + r.first, r.second, lvl, ZIPC_HIP_CRC_CRC32, out, c);
  if (st) return R::Error(message(st));
  return R::Ok({c, std::move(out)});
}
Result<std::pair<Adler_32::t, std::string>> adler_32_and_deflate(const std::string &s, std::optional<level> lvl,
                                                                 std::size_t start, std::size_t len) {
  typedef Result<std::pair<Adler_32::t, std::string>> R;
  const auto r = range(s, start, len);
  std::string out;
  uint32 c = 0;
  const int st = deflate_raw(s.data() + r.first, r.second, lvl, ZIPC_HIP_CRC_ADLER32, out, c);
  if (st) return R::Error(message(st));
  return R::Ok({c, std::move(out)});
}
Result<std::string> zlib_compress(const std::string &s, std::optional<level> lvl, std::size_t start,
                                  std::size_t len) {
  const auto r = range(s, start, len);
  const std::size_t cap = zipc_hip_zlib_bound(r.second);
  std::string out(cap, '\0');
  std::size_t out_len = 0;
  uint32 adler = 0;
  const int st = zipc_hip_zlib_compress(context(), s.data() + r.first, r.second, level_of(lvl), &out[0], cap, &out_len,
                                        &adler);
  throw_if_library_failure(st);
  if (st) return Result<std::string>::Error(message(st));
  out.resize(out_len);
  return Result<std::string>::Ok(std::move(out));
}

std::vector<std::pair<std::size_t, std::size_t>> partition_items(const std::vector<ManyItem> &items, std::size_t n_devices) {
  // contiguous ranges balanced by bytes: boundary k is the first index where the running byte count reaches
  // k / n_devices of the total (zipc_amd/shard.partition is the same rule for bench.py's ranks)
  const std::size_t n = items.size();
  if (n_devices < 1) n_devices = 1;
  std::vector<unsigned long long> csum(n + 1, 0);
  for (std::size_t i = 0; i < n; i++) csum[i + 1] = csum[i] + items[i].len;
  const unsigned __int128 total = csum[n];
  std::vector<std::size_t> bounds(1, 0);
  for (std::size_t k = 1; k < n_devices; k++) {
    std::size_t b = bounds.back();
    while (b < n && (unsigned __int128)csum[b] * n_devices < total * k) b++;
    bounds.push_back(b);
  }
  bounds.push_back(n);
  std::vector<std::pair<std::size_t, std::size_t>> out;
  for (std::size_t k = 0; k < n_devices; k++) out.push_back({bounds[k], bounds[k + 1]});
  return out;
}

// a batch smaller than this stays on one device: a second context costs a launch chain and staging of its own
static const std::size_t SHARD_MIN_BYTES = 32u << 20;

// runs call(ctx, first, last) for the ranges of the items, one range per device of devices()
template <class Call>
static void over_devices(const std::vector<ManyItem> &items, Call call) {
  const std::vector<int> devs = devices();
  unsigned long long total = 0;
  for (const auto &it : items) total += it.len;
  if (devs.size() < 2 || items.size() < 2 * devs.size() || total < SHARD_MIN_BYTES) {
    call(context(), (std::size_t)0, items.size());
    return;
  }
  const auto parts = partition_items(items, devs.size());
  std::vector<std::thread> workers;
  std::vector<std::string> failed(devs.size());
  for (std::size_t r = 0; r < devs.size(); r++) {
    if (parts[r].first == parts[r].second) continue;
    workers.emplace_back([&, r] {
      try {
        PoolContext &pc = pool_context(r, devs[r]);
        std::lock_guard<std::mutex> g(pc.mu);
        if (!pc.ctx) {
          const int st = zipc_hip_create(&pc.ctx, devs[r]);
          if (st) throw std::runtime_error(std::string("zipc_hip_create (device ") + std::to_string(devs[r]) + "): " + zipc_hip_strerror(st));
        }
        call(pc.ctx, parts[r].first, parts[r].second);
      } catch (const std::exception &e) {
        failed[r] = e.what();
        if (failed[r].empty()) failed[r] = "failed";
      }
    });
  }
  for (auto &w : workers) w.join();
  for (const auto &f : failed)
    if (!f.empty()) throw std::runtime_error(f);
}

std::vector<ManyResult> crc_32_and_deflate_many(const std::vector<ManyItem> &items, std::optional<level> lvl) {
  const std::size_t n = items.size();
  std::vector<ManyResult> out(n);
  std::vector<const void *> src(n);
  std::vector<void *> dst(n);
  std::vector<std::size_t> len(n), cap(n);
  for (std::size_t i = 0; i < n; i++) {
    src[i] = items[i].data;
    len[i] = items[i].len;
    cap[i] = zipc_hip_deflate_bound(items[i].len);
    out[i].value.resize(cap[i]);
    dst[i] = &out[i].value[0];
  }
  std::vector<zipc_hip_stream_result> res(n);
  over_devices(items, [&](zipc_hip_ctx *ctx, std::size_t lo, std::size_t hi) {
    const int st = zipc_hip_deflate_many(ctx, hi - lo, src.data() + lo, len.data() + lo, level_of(lvl), ZIPC_HIP_CRC_CRC32,
                                         dst.data() + lo, cap.data() + lo, res.data() + lo);
    if (st) throw std::runtime_error(std::string("zipc_hip_deflate_many: ") + message(st) + " (" + zipc_hip_last_error(ctx) + ")");
  });
  for (std::size_t i = 0; i < n; i++) {
    throw_if_library_failure((int)res[i].status);
    out[i].ok = res[i].status == ZIPC_HIP_OK;
    out[i].value.resize(out[i].ok ? res[i].out_len : 0);
    out[i].checksum = res[i].checksum;
    if (!out[i].ok) out[i].error = message((int)res[i].status);
  }
  return out;
}

std::vector<ManyResult> inflate_and_crc_32_many_check(const std::vector<ManyItem> &items) {
  const std::size_t n = items.size();
  std::vector<ManyResult> out(n);
  std::vector<const void *> src(n);
  std::vector<std::size_t> len(n), cap(n), limit(n);
  std::vector<ManyItem> by_output(n);
  for (std::size_t i = 0; i < n; i++) {
    if (!items[i].decompressed_size) throw std::invalid_argument("inflate_and_crc_32_many_check: decompressed_size missing");
    src[i] = items[i].data;
    len[i] = items[i].len;
    cap[i] = limit[i] = *items[i].decompressed_size;
    by_output[i].len = cap[i];
  }
  std::vector<zipc_hip_stream_result> res(n);
  over_devices(by_output, [&](zipc_hip_ctx *ctx, std::size_t lo, std::size_t hi) {
    const int st = zipc_hip_inflate_many_check(ctx, hi - lo, src.data() + lo, len.data() + lo, limit.data() + lo, ZIPC_HIP_CRC_CRC32,
                                               cap.data() + lo, res.data() + lo);
    if (st) throw std::runtime_error(std::string("zipc_hip_inflate_many_check: ") + message(st) + " (" + zipc_hip_last_error(ctx) + ")");
  });
  for (std::size_t i = 0; i < n; i++) {
    throw_if_library_failure((int)res[i].status);
    out[i].ok = res[i].status == ZIPC_HIP_OK;
    out[i].checksum = res[i].checksum;
    if (!out[i].ok) out[i].error = message((int)res[i].status);
  }
  return out;
}

std::vector<ManyResult> inflate_and_crc_32_many(const std::vector<ManyItem> &items) {
  const std::size_t n = items.size();
  std::vector<ManyResult> out(n);
  std::vector<const void *> src(n);
  std::vector<void *> dst(n);
  std::vector<std::size_t> len(n), cap(n), limit(n);
  std::vector<ManyItem> by_output(n);  // the work of inflate goes by what it produces
  for (std::size_t i = 0; i < n; i++) {
    if (!items[i].decompressed_size) throw std::invalid_argument("inflate_and_crc_32_many: decompressed_size missing");
    src[i] = items[i].data;
    len[i] = items[i].len;
    cap[i] = limit[i] = *items[i].decompressed_size;
    out[i].value.resize(cap[i]);
    dst[i] = cap[i] ? &out[i].value[0] : nullptr;
    by_output[i].len = cap[i];
  }
  std::vector<zipc_hip_stream_result> res(n);
  over_devices(by_output, [&](zipc_hip_ctx *ctx, std::size_t lo, std::size_t hi) {
    const int st = zipc_hip_inflate_many(ctx, hi - lo, src.data() + lo, len.data() + lo, limit.data() + lo, ZIPC_HIP_CRC_CRC32,
                                         dst.data() + lo, cap.data() + lo, res.data() + lo);
    if (st) throw std::runtime_error(std::string("zipc_hip_inflate_many: ") + message(st) + " (" + zipc_hip_last_error(ctx) + ")");
  });
  for (std::size_t i = 0; i < n; i++) {
    throw_if_library_failure((int)res[i].status);
    out[i].ok = res[i].status == ZIPC_HIP_OK;
    out[i].value.resize(out[i].ok ? res[i].out_len : 0);
    out[i].checksum = res[i].checksum;
    if (!out[i].ok) out[i].error = message((int)res[i].status);
  }
  return out;
}

}  // namespace zipc_deflate
