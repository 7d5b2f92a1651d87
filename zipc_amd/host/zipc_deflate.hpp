// zipc_deflate.hpp -- the reference's `Zipc_deflate` signature (src/zipc_deflate.mli)
// as a C++ interface over the C ABI of include/zipc_hip.h.
//
// This is the host side of the MI355X codec for callers written in a compiled
// language: same names, argument meaning and error behaviour as the OCaml module
// (the OCaml shim of bindings/ocaml does the same marshalling).  Every function
// runs on the GPU through libzipc_hip.so; there is no CPU implementation behind it
// -- with no device the calls fail with the library's status text.
//
//   OCaml                                  here
//   ?start ?len s                          (s, start = 0, len = npos); out of range -> std::invalid_argument
//   ?level                                 std::optional<level>; absent = `Best (zipc_deflate.ml:817)
//   ?decompressed_size                     std::optional<size_t>
//   ('a, string) result                    Result<T>: ok / value / error
//   Failure escaping as an exception       never: statuses only
#pragma once

#include <cstddef>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zipc_hip.h"

namespace zipc_deflate {

constexpr std::size_t npos = static_cast<std::size_t>(-1);

template <class T>
struct Result {
  bool ok = false;
  T value{};
  std::string error;
  static Result Ok(T v) { Result r; r.ok = true; r.value = std::move(v); return r; }
  static Result Error(std::string e) { Result r; r.error = std::move(e); return r; }
};
struct Unit {};

typedef std::uint16_t uint16;  // zipc_deflate.mli:17
typedef std::uint32_t uint32;  // zipc_deflate.mli:20

// The context the calling THREAD's calls go through (created on the thread's first use on the
// thread's device, destroyed when it exits): a zipc_hip context serves one thread at a time, and
// like the reference module these functions may be called from several threads at once.  Throws
// std::runtime_error when the library cannot create one.
zipc_hip_ctx *context();
// The device of the calling thread's context: the first of devices() unless set_thread_device() chose
// another (a context made on the earlier device is replaced).
void set_thread_device(int device);
int thread_device();
// The devices the many-stream forms below spread a batch over, one host thread and one context per
// entry: every visible device (zipc_hip_device_count()) by default, ZIPC_HIP_DEVICES="0,2,3" from the
// environment, or set_devices().  A device may be named more than once (two contexts on it).
std::vector<int> devices();
void set_devices(const std::vector<int> &list);

// (start, len) of the reference's optional arguments -> the checked range
inline std::pair<std::size_t, std::size_t> range(const std::string &s, std::size_t start, std::size_t len) {
  if (start > s.size()) throw std::invalid_argument("index out of bounds");
  if (len == npos) len = s.size() - start;
  if (len > s.size() - start) throw std::invalid_argument("index out of bounds");
  return {start, len};
}

// crc_error zipc_deflate.ml:103-104 (the unbalanced parenthesis is the reference's)
std::string crc_error(uint32 expect, uint32 found);

struct Crc_32 {  // zipc_deflate.mli:24-48
  typedef uint32 t;
  static bool equal(t a, t b) { return a == b; }
  static Result<Unit> check(t expect, t found);
  static std::string pp(t crc);  // "%lx": no zero padding (zipc_deflate.ml:111)
  static t string(const std::string &s, std::size_t start = 0, std::size_t len = npos);
};
struct Adler_32 {  // zipc_deflate.mli:50-75
  typedef uint32 t;
  static bool equal(t a, t b) { return a == b; }
  static Result<Unit> check(t expect, t found);
  static std::string pp(t crc);
  static t string(const std::string &s, std::size_t start = 0, std::size_t len = npos);
};

// zipc_deflate.mli:79-102
Result<std::string> inflate(const std::string &s, std::optional<std::size_t> decompressed_size = std::nullopt,
                            std::size_t start = 0, std::size_t len = npos);
Result<std::pair<std::string, Crc_32::t>> inflate_and_crc_32(const std::string &s,
                                                             std::optional<std::size_t> decompressed_size = std::nullopt,
                                                             std::size_t start = 0, std::size_t len = npos);
Result<std::pair<std::string, Adler_32::t>> inflate_and_adler_32(const std::string &s,
                                                                 std::optional<std::size_t> decompressed_size = std::nullopt,
                                                                 std::size_t start = 0, std::size_t len = npos);

// zlib_decompress zipc_deflate.mli:104-118: the error carries (expect, found) on a checksum mismatch
struct ZlibError {
  std::optional<std::pair<Adler_32::t, Adler_32::t>> mismatch;
  std::string message;
};
struct ZlibResult {
  bool ok = false;
  std::string value;
  Adler_32::t adler = 0;
  ZlibError error;
};
ZlibResult zlib_decompress(const std::string &s, std::optional<std::size_t> decompressed_size = std::nullopt,
                           std::size_t start = 0, std::size_t len = npos);

enum class level { None = 0, Fast = 1, Default = 2, Best = 3 };  // zipc_deflate.mli:123-125

// zipc_deflate.mli:128-162
Result<std::string> deflate(const std::string &s, std::optional<level> lvl = std::nullopt, std::size_t start = 0,
                            std::size_t len = npos);
Result<std::pair<Crc_32::t, std::string>> crc_32_and_deflate(const std::string &s,
                                                             std::optional<level> lvl = std::nullopt,
                                                             std::size_t start = 0, std::size_t len = npos);
Result<std::pair<Adler_32::t, std::string>> adler_32_and_deflate(const std::string &s,
                                                                 std::optional<level> lvl = std::nullopt,
                                                                 std::size_t start = 0, std::size_t len = npos);
Result<std::string> zlib_compress(const std::string &s, std::optional<level> lvl = std::nullopt,
                                  std::size_t start = 0, std::size_t len = npos);

// ---- many streams at once (no counterpart in the reference, which handles one
// string per call: what an archive-level caller uses, zipc.hpp)
struct ManyItem {
  const char *data = nullptr;
  std::size_t len = 0;
  std::optional<std::size_t> decompressed_size;  // inflate only
};
struct ManyResult {
  bool ok = false;
  std::string value;
  uint32 checksum = 0;
  std::string error;
};
// crc_32_and_deflate of every item through the batch kernels.  The items are independent streams (archive
// members: "trivially parallelizable", test/zipc_tool.ml:6-8), so with several devices() they are cut into
// contiguous ranges of about equal bytes, one range per device, each through its own context on a host
// thread of its own -- no exchange between the devices, results in the order of the items.
std::vector<ManyResult> crc_32_and_deflate_many(const std::vector<ManyItem> &items,
                                                std::optional<level> lvl = std::nullopt);
// the ranges [first, last) the items are cut into for n_devices (what the sharded forms use; bench.py's
// zipc_amd/shard.partition is the same rule)
std::vector<std::pair<std::size_t, std::size_t>> partition_items(const std::vector<ManyItem> &items, std::size_t n_devices);
// inflate_and_crc_32 of every item; items need decompressed_size (the members of
// an archive have it)
std::vector<ManyResult> inflate_and_crc_32_many(const std::vector<ManyItem> &items);
// ... for its verdict alone: ok / error and the CRC-32 found, value left empty -- nothing but the results comes back
// from the device (zipc_hip_inflate_many_check: what testing an archive takes)
std::vector<ManyResult> inflate_and_crc_32_many_check(const std::vector<ManyItem> &items);

}  // namespace zipc_deflate
