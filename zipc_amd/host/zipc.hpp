// zipc.hpp -- the reference's `Zipc` module (src/zipc.mli): ZIP archive model,
// member glue and container codec, as a C++ interface.  SURVEY.md section 8(f) rows 1
// and 2: what sits either side of the deflate hot path.
//
// Same names, argument meaning, defaults and error messages as the OCaml module;
// the codec calls (Zipc_deflate.crc_32_and_deflate, inflate_and_crc_32,
// Crc_32.string) go to the MI355X library through zipc_deflate.hpp.  Values are
// immutable and cheap to copy like the reference's: a File shares the bytes it
// points into (std::shared_ptr<const std::string>), an Archive is an ordered map.
//
// Beyond the reference (which handles one member per call) Archive has two batch
// operations that hand all members to the GPU at once -- the use the hot path was
// built for: add_deflated_files and extract_all.
#pragma once

#include <map>
#include <memory>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include "zipc_deflate.hpp"

namespace zipc {

using zipc_deflate::Result;
using zipc_deflate::Unit;

// type compression zipc.mli:20-28 (zipc.ml:23-35)
struct compression {
  enum kind_t { Bzip2, Deflate, Lzma, Stored, Xz, Zstd, Other } kind = Stored;
  int other = 0;  // the method number of Other
  static compression of_int(int c);
  int to_int() const;
  std::string to_string() const;  // pp_compression
  bool operator==(const compression &o) const { return to_int() == o.to_int(); }
};

namespace Fpath {  // zipc.mli:34-67
typedef std::string t;
typedef int mode;
t ensure_unix(const t &p);
t ensure_directoryness(const t &p);
t sanitize(const t &p);
std::string pp_mode(int m);
}  // namespace Fpath

namespace Ptime {  // zipc.mli:71-87 (+ the DOS conversions of zipc.ml:93-124)
typedef long long t;  // POSIX seconds
constexpr t dos_epoch = 315532800;
std::tuple<int, int, int, int, int, int> to_date_time(t ptime_s);  // (y, m, d, hh, mm, ss)
std::string pp(t ptime_s);
t of_dos_date_time(int dos_date, int dos_time);
std::pair<int, int> to_dos_date_time(t ptime_s);  // (dos_date, dos_time)
}  // namespace Ptime

// optional arguments of File.make (zipc.mli:100-121)
struct file_make_args {
  int version_made_by = (3 << 8) | 20;  // UNIX, PKZIP 2.0
  int version_needed_to_extract = 20;   // PKZIP 2.0
  int gp_flags = 0x800;                 // UTF-8 names
  std::size_t start = 0;
  std::optional<long long> compressed_size;
};

class File {  // zipc.mli:93-212
 public:
  static constexpr long long max_size = 0xFFFFFFFFll;
  static constexpr int gp_default = 0x800;
  static constexpr int version_made_by_default = (3 << 8) | 20;
  static constexpr int version_needed_to_extract_default = 20;

  typedef file_make_args make_args;
  // File.make; negative sizes -> std::invalid_argument like the reference's Invalid_argument
  static Result<File> make(compression c, std::shared_ptr<const std::string> compressed_bytes,
                           long long decompressed_size, zipc_deflate::uint32 decompressed_crc_32,
                           const make_args &a = make_args());
  static Result<File> stored_of_binary_string(const std::string &s, std::size_t start = 0,
                                              std::size_t len = zipc_deflate::npos);
  static Result<File> deflate_of_binary_string(const std::string &s,
                                               std::optional<zipc_deflate::level> level = std::nullopt,
                                               std::size_t start = 0, std::size_t len = zipc_deflate::npos);

  compression compression_() const { return compression__; }
  std::size_t start() const { return start_; }
  long long compressed_size() const { return compressed_size_; }
  const std::string &compressed_bytes() const { return *bytes_; }
  std::shared_ptr<const std::string> compressed_bytes_ptr() const { return bytes_; }
  std::string compressed_bytes_to_binary_string() const;
  long long decompressed_size() const { return decompressed_size_; }
  zipc_deflate::uint32 decompressed_crc_32() const { return crc_; }
  int version_made_by() const { return made_by_; }
  int version_needed_to_extract() const { return needed_; }
  int gp_flags() const { return gp_; }
  bool is_encrypted() const { return (gp_ & 0x1) != 0; }
  bool can_extract() const;
  Result<std::string> to_binary_string() const;
  Result<std::pair<std::string, zipc_deflate::uint32>> to_binary_string_no_crc_check() const;

 private:
  friend class Archive;
  int made_by_ = version_made_by_default, needed_ = version_needed_to_extract_default, gp_ = gp_default;
  compression compression__;
  std::size_t start_ = 0;
  long long compressed_size_ = 0, decompressed_size_ = 0;
  std::shared_ptr<const std::string> bytes_;
  zipc_deflate::uint32 crc_ = 0;
};

class Member {  // zipc.mli:220-281
 public:
  static constexpr int max = 0xFFFF;
  static constexpr int max_path_length = 0xFFFF;
  // kind: no file = Dir
  static Result<Member> make(const Fpath::t &path, std::optional<File> file_kind,
                             std::optional<Ptime::t> mtime = std::nullopt, std::optional<int> mode = std::nullopt);
  const Fpath::t &path() const { return path_; }
  bool is_dir() const { return !file_; }
  const File &file() const { return *file_; }
  int mode() const { return mode_; }
  Ptime::t mtime() const { return mtime_; }
  std::string pp(bool long_form = false) const;  // pp / pp_long

 private:
  friend class Archive;
  Fpath::t path_;
  std::optional<File> file_;
  int mode_ = 0;
  Ptime::t mtime_ = Ptime::dos_epoch;
};

class Archive {  // type t and its functions, zipc.mli:287-384
 public:
  bool is_empty() const { return members_.empty(); }
  bool mem(const Fpath::t &p) const { return members_.count(p) != 0; }
  const Member *find(const Fpath::t &p) const;
  template <class F>
  void fold(F f) const { for (const auto &kv : members_) f(kv.second); }  // increasing path order
  void add(const Member &m) { members_[m.path()] = m; }
  void remove(const Fpath::t &p) { members_.erase(p); }
  std::size_t member_count() const { return members_.size(); }
  const std::map<std::string, Member> &to_string_map() const { return members_; }

  static bool string_has_magic(const std::string &s);
  static Result<Archive> of_binary_string(std::shared_ptr<const std::string> s);
  static Result<Archive> of_binary_string(const std::string &s) { return of_binary_string(std::make_shared<const std::string>(s)); }
  std::size_t encoding_size() const;
  Result<std::string> to_binary_string(const Fpath::t &first = "mimetype") const;
  // write_bytes: encodes at b[start ..); b must hold start + encoding_size() bytes
  Result<Unit> write_bytes(std::string &b, std::size_t start = 0, const Fpath::t &first = "mimetype") const;

  // ---- all members at once on the GPU
  struct NewFile {
    Fpath::t path;
    const std::string *data;
    std::optional<Ptime::t> mtime;
    std::optional<int> mode;
  };
  // File.deflate_of_binary_string + Member.make + add for every entry, the
  // compression as ONE batch; the first error (in entry order) is returned and
  // nothing is added then.
  Result<Unit> add_deflated_files(const std::vector<NewFile> &files,
                                  std::optional<zipc_deflate::level> level = std::nullopt);
  // File.to_binary_string of every extractable Deflate/Stored file member, in path
  // order; per member the reference's result (bytes, or its error message)
  std::vector<std::pair<Fpath::t, Result<std::string>>> extract_all() const;
  // ... for its Ok / Error alone (`zipc unzip -t`, test/zipc_tool.ml:635-660): deflated members are decoded and CRC-checked on
  // the device and only their results come back; the others take the single-member path.  Same members, same order, same
  // messages as extract_all.
  std::vector<std::pair<Fpath::t, Result<Unit>>> test_all() const;

 private:
  std::map<std::string, Member> members_;
};

}  // namespace zipc
