// zipc.cpp -- see zipc.hpp.  Row references are to /root/reference/src/zipc.ml.
#include "zipc.hpp"

#include <cstdio>
#include <cstring>

namespace zipc {

using zipc_deflate::uint32;

namespace {

std::string strf(const char *fmt, long long a) {
  char b[160];
  snprintf(b, sizeof b, fmt, a);
  return b;
}
std::string strf2(const char *fmt, long long a, long long b2) {
  char b[200];
  snprintf(b, sizeof b, fmt, a, b2);
  return b;
}

// little-endian fields of a byte string
inline unsigned u8(const std::string &s, std::size_t i) { return (unsigned char)s[i]; }
inline unsigned u16(const std::string &s, std::size_t i) { return u8(s, i) | (u8(s, i + 1) << 8); }
inline uint32 u32(const std::string &s, std::size_t i) { return (uint32)u16(s, i) | ((uint32)u16(s, i + 2) << 16); }
inline void put16(std::string &b, std::size_t i, unsigned v) {
  b[i] = (char)(v & 0xFF);
  b[i + 1] = (char)((v >> 8) & 0xFF);
}
inline void put32(std::string &b, std::size_t i, uint32 v) {
  put16(b, i, v & 0xFFFF);
  put16(b, i + 2, v >> 16);
}

// a parse failure: the reference's `failwith` inside of_binary_string
struct Failure {
  std::string msg;
};
[[noreturn]] void fail(const char *m) { throw Failure{m}; }

}  // namespace

// ---- compression zipc.ml:23-35
compression compression::of_int(int c) {
  compression r;
  switch (c) {
  case 0: r.kind = Stored; break;
  case 8: r.kind = Deflate; break;
  case 12: r.kind = Bzip2; break;
  case 14: r.kind = Lzma; break;
  case 93: r.kind = Zstd; break;
  case 95: r.kind = Xz; break;
  default: r.kind = Other; r.other = c; break;
  }
  return r;
}
int compression::to_int() const {
  switch (kind) {
  case Stored: return 0;
  case Deflate: return 8;
  case Bzip2: return 12;
  case Lzma: return 14;
  case Zstd: return 93;
  case Xz: return 95;
  default: return other;
  }
}
std::string compression::to_string() const {
  switch (kind) {
  case Bzip2: return "bz2";
  case Deflate: return "defl";
  case Lzma: return "lzma";
  case Stored: return "none";
  case Xz: return "xz";
  case Zstd: return "zst";
  default: return strf("%04d", other);
  }
}

// ---- Fpath zipc.ml:39-62
namespace Fpath {
t ensure_unix(const t &p) {
  t r = p;
  for (char &c : r)
    if (c == '\\') c = '/';
  return r;
}
t ensure_directoryness(const t &p) {
  if (p.empty()) return "./";
  return p.back() == '/' ? p : p + "/";
}
t sanitize(const t &p) {
  t out;
  std::size_t i = 0;
  bool first = true;
  while (i <= p.size()) {
    std::size_t j = i;
    while (j < p.size() && p[j] != '/' && p[j] != '\\') j++;
    const t seg = p.substr(i, j - i);
    if (!(seg.empty() || seg == ".." || seg == ".")) {
      if (!first) out += '/';
      out += seg;
      first = false;
    }
    i = j + 1;
  }
  return out;
}
std::string pp_mode(int m) {
  std::string r;
  for (int shift : {6, 3, 0}) {
    const int e = m >> shift;
    r += (e & 4) ? 'r' : '-';
    r += (e & 2) ? 'w' : '-';
    r += (e & 1) ? 'x' : '-';
  }
  return r;
}
}  // namespace Fpath

// ---- Ptime zipc.ml:64-124 (C++ integer division truncates like OCaml's)
namespace Ptime {
static constexpr long long jd_posix_epoch = 2440588;

std::tuple<int, int, int, int, int, int> to_date_time(t s) {
  const long long jd = s / 86400 + jd_posix_epoch;
  const long long jd_rem = s % 86400;
  const long long hh = jd_rem / 3600, hh_rem = jd_rem % 3600;
  const long long mm = hh_rem / 60, ss = hh_rem % 60;
  const long long a = jd + 32044;
  const long long b = (4 * a + 3) / 146097;
  const long long c = a - (146097 * b) / 4;
  const long long d = (4 * c + 3) / 1461;
  const long long e = c - (1461 * d) / 4;
  const long long m = (5 * e + 2) / 153;
  const long long day = e - (153 * m + 2) / 5 + 1;
  const long long month = m + 3 - 12 * (m / 10);
  const long long year = 100 * b + d - 4800 + m / 10;
  return std::make_tuple((int)year, (int)month, (int)day, (int)hh, (int)mm, (int)ss);
}
std::string pp(t s) {
  int y, mo, d, hh, mm, ss;
  std::tie(y, mo, d, hh, mm, ss) = to_date_time(s);
  char b[64];
  snprintf(b, sizeof b, "%04d-%02d-%02d %02d:%02d:%02dZ", y, mo, d, hh, mm, ss);
  return b;
}
t of_dos_date_time(int dos_date, int dos_time) {
  if (dos_date < 0x21) return dos_epoch;  // before 1980-01-01
  const long long hh = dos_time >> 11, mm = (dos_time >> 5) & 0x3F, ss = (dos_time & 0x1F) * 2;
  const long long year = ((dos_date >> 9) & 0x7F) + 1980, month = (dos_date >> 5) & 0xF, day = dos_date & 0x1F;
  const long long a = (14 - month) / 12;
  const long long y = year + 4800 - a;
  const long long m = month + 12 * a - 3;
  const long long jd = day + (153 * m + 2) / 5 + 365 * y + y / 4 - y / 100 + y / 400 - 32045;
  return (jd - jd_posix_epoch) * 86400 + hh * 3600 + mm * 60 + ss;
}
std::pair<int, int> to_dos_date_time(t s) {
  int y, mo, d, hh, mm, ss;
  std::tie(y, mo, d, hh, mm, ss) = to_date_time(s);
  if (y < 1980) { y = 1980; mo = 1; d = 1; hh = mm = ss = 0; }
  else if (y > 2107) { y = 2107; mo = 12; d = 31; hh = 23; mm = 59; ss = 59; }
  return {d | (mo << 5) | ((y - 1980) << 9), (ss / 2) | (mm << 5) | (hh << 11)};
}
}  // namespace Ptime

// ---- File zipc.ml:127-226
Result<File> File::make(compression c, std::shared_ptr<const std::string> bytes, long long decompressed_size,
                        uint32 crc, const make_args &a) {
  const long long compressed_size = a.compressed_size ? *a.compressed_size : (long long)bytes->size() - (long long)a.start;
  if (compressed_size < 0) throw std::invalid_argument(strf("compressed_size is negative (%lld)", compressed_size));
  if (decompressed_size < 0) throw std::invalid_argument(strf("decompressed_size is negative (%lld)", decompressed_size));
  if (compressed_size > max_size || decompressed_size > max_size)
    return Result<File>::Error(strf2("Maximum ZIP byte size 4294967295 exceeded by compressed (%lld) or decompressed "
                                     "(%lld) file size", compressed_size, decompressed_size));
  File f;
  f.made_by_ = a.version_made_by;
  f.needed_ = a.version_needed_to_extract;
  f.gp_ = a.gp_flags;
  f.compression__ = c;
  f.start_ = a.start;
  f.compressed_size_ = compressed_size;
  f.bytes_ = std::move(bytes);
  f.decompressed_size_ = decompressed_size;
  f.crc_ = crc;
  return Result<File>::Ok(std::move(f));
}

Result<File> File::stored_of_binary_string(const std::string &s, std::size_t start, std::size_t len) {
  const auto r = zipc_deflate::range(s, start, len);
  const uint32 crc = zipc_deflate::Crc_32::string(s, r.first, r.second);
  make_args a;
  a.start = r.first;
  a.compressed_size = (long long)r.second;
  return make(compression::of_int(0), std::make_shared<const std::string>(s), (long long)r.second, crc, a);
}

Result<File> File::deflate_of_binary_string(const std::string &s, std::optional<zipc_deflate::level> level,
                                            std::size_t start, std::size_t len) {
  const auto r = zipc_deflate::range(s, start, len);
  auto d = zipc_deflate::crc_32_and_deflate(s, level, r.first, r.second);
  if (!d.ok) return Result<File>::Error(d.error);
  return make(compression::of_int(8), std::make_shared<const std::string>(std::move(d.value.second)),
              (long long)r.second, d.value.first);
}

std::string File::compressed_bytes_to_binary_string() const { return bytes_->substr(start_, (std::size_t)compressed_size_); }

bool File::can_extract() const {
  return !is_encrypted() && (compression__.kind == compression::Stored || compression__.kind == compression::Deflate);
}

Result<std::pair<std::string, uint32>> File::to_binary_string_no_crc_check() const {
  typedef Result<std::pair<std::string, uint32>> R;
  if (is_encrypted()) return R::Error("Encrypted file not supported");
  if (compression__.kind == compression::Stored) {
    std::string s = compressed_bytes_to_binary_string();
    const uint32 crc = zipc_deflate::Crc_32::string(s);
    return R::Ok({std::move(s), crc});
  }
  if (compression__.kind == compression::Deflate) {
    auto r = zipc_deflate::inflate_and_crc_32(*bytes_, (std::size_t)decompressed_size_, start_, (std::size_t)compressed_size_);
    if (!r.ok) return R::Error("deflate: " + r.error);
    return R::Ok(std::move(r.value));
  }
  return R::Error("Compression " + compression__.to_string() + " not supported");
}

Result<std::string> File::to_binary_string() const {
  auto r = to_binary_string_no_crc_check();
  if (!r.ok) return Result<std::string>::Error(r.error);
  auto c = zipc_deflate::Crc_32::check(crc_, r.value.second);
  if (!c.ok) return Result<std::string>::Error(c.error);
  return Result<std::string>::Ok(std::move(r.value.first));
}

// ---- Member zipc.ml:233-290
Result<Member> Member::make(const Fpath::t &path, std::optional<File> file_kind, std::optional<Ptime::t> mtime,
                            std::optional<int> mode) {
  Member m;
  m.path_ = Fpath::ensure_unix(path);
  if (!file_kind) m.path_ = Fpath::ensure_directoryness(m.path_);
  if (m.path_.size() > (std::size_t)max_path_length)
    return Result<Member>::Error(strf2("Maximum ZIP path length %lld exceeded (%lld)", max_path_length,
                                       (long long)m.path_.size()));
  m.mode_ = mode ? *mode : (file_kind ? 0644 : 0755);
  const Ptime::t t = mtime ? *mtime : Ptime::dos_epoch;
  m.mtime_ = t < Ptime::dos_epoch ? Ptime::dos_epoch : t;
  m.file_ = std::move(file_kind);
  return Result<Member>::Ok(std::move(m));
}

std::string Member::pp(bool long_form) const {
  const char is_dir_c = is_dir() ? 'd' : '-';
  char comp[16];
  snprintf(comp, sizeof comp, "%4s", is_dir() ? "none" : file_->compression_().to_string().c_str());
  const char enc = (!is_dir() && file_->is_encrypted()) ? 'X' : ' ';
  const long long size = is_dir() ? 0 : file_->decompressed_size();
  char pct[16] = "    ";
  if (!is_dir()) {
    const double r = (double)file_->compressed_size() / (double)file_->decompressed_size();
    const double v = r * 100.0;
    snprintf(pct, sizeof pct, "%3d%%", (v == v && v < 2147483648.0) ? (int)v : 0);
  }
  char crc[16] = "";
  if (long_form) {
    if (is_dir()) snprintf(crc, sizeof crc, "        ");
    else snprintf(crc, sizeof crc, "%08x", file_->decompressed_crc_32());
  }
  char head[96];
  snprintf(head, sizeof head, "%c%s %s%c%s %8lld %s ", is_dir_c, Fpath::pp_mode(mode_).c_str(), comp, enc, crc, size, pct);
  return std::string(head) + Ptime::pp(mtime_) + " " + path_;
}

// ---- archive
const Member *Archive::find(const Fpath::t &p) const {
  auto it = members_.find(p);
  return it == members_.end() ? nullptr : &it->second;
}

// ---- decoding zipc.ml:314-446
namespace {
constexpr uint32 lfh_sig = 0x04034b50u, cdfh_sig = 0x02014b50u, eocd_sig = 0x06054b50u;
constexpr std::size_t lfh_min_size = 30, cdfh_min_size = 46, eocd_min_size = 22;

std::size_t decode_data_start_of_lfh(const std::string &s, std::size_t i, std::size_t compressed_size) {
  if (i + lfh_min_size > s.size() || u32(s, i) != lfh_sig) fail("Corrupted local file header");
  const std::size_t data_start = i + lfh_min_size + u16(s, i + 26) + u16(s, i + 28);
  if (data_start + compressed_size > s.size()) fail("Corrupted local file header");
  return data_start;
}
}  // namespace

bool Archive::string_has_magic(const std::string &s) {
  if (s.size() < 4) return false;
  const uint32 m = u32(s, 0);
  return m == lfh_sig || m == eocd_sig;
}

Result<Archive> Archive::of_binary_string(std::shared_ptr<const std::string> sp) {
  const std::string &s = *sp;
  try {
    // find_cd_info_in_eocd zipc.ml:401-432: the record ends with a comment of up
    // to 65535 bytes, so it is searched for from the end
    const long long len = (long long)s.size();
    long long start = len - (long long)eocd_min_size;
    if (start < 0) fail("File too short to be a ZIP archive");
    const long long min_start = len - 65535 - (long long)eocd_min_size;
    for (;;) {
      if (start < min_start || start < 0) fail("Likely not a ZIP archive: no end of central directory record found");
      if (u32(s, (std::size_t)start) == eocd_sig) break;
      start--;
    }
    const std::size_t e = (std::size_t)start;
    const unsigned disk_num = u16(s, e + 4), disk_cd = u16(s, e + 6);
    if (disk_num == 0xFFFF) fail("ZIP64 archives are not supported");
    if (disk_num != 0 || disk_cd != 0) fail("Multipart archives are not supported");
    std::size_t count = u16(s, e + 10);
    const unsigned long long cd_size = u32(s, e + 12), cd_start = u32(s, e + 16);
    if (cd_start + cd_size > (unsigned long long)len) fail("Corrupted end of central directory record");
    const long long cd_max = (long long)(cd_start + cd_size) - 1;

    Archive z;
    long long i = (long long)cd_start;
    for (; count != 0; count--) {  // decode_cd_members zipc.ml:393-397
      if (i > cd_max) fail("Truncated central directory");
      // decode_member_of_cd zipc.ml:344-391
      if (i + (long long)cdfh_min_size - 1 > cd_max || u32(s, (std::size_t)i) != cdfh_sig)
        fail("Corrupted central directory file header");
      const std::size_t h = (std::size_t)i;
      const std::size_t path_len = u16(s, h + 28);
      const long long next = i + (long long)cdfh_min_size + (long long)path_len + u16(s, h + 30) + u16(s, h + 32);
      if (next - 1 > cd_max) fail("Corrupted central directory file header");
      Member m;
      m.path_ = s.substr(h + 46, path_len);
      m.mtime_ = Ptime::of_dos_date_time((int)u16(s, h + 14), (int)u16(s, h + 12));
      bool is_dir;
      const unsigned ext_hi = u16(s, h + 40);
      if (ext_hi != 0) {  // unix permissions
        is_dir = (ext_hi & 070000) == 040000;
        m.mode_ = (int)(ext_hi & 07777);
      } else if (u8(s, h + 38) & 0x10) {  // MS-DOS directory bit
        is_dir = true;
        m.mode_ = 0755;
      } else {
        is_dir = false;
        m.mode_ = 0644;
      }
      if (!is_dir) {
        File f;
        f.compression__ = compression::of_int((int)u16(s, h + 10));
        f.made_by_ = (int)u16(s, h + 4);
        f.needed_ = (int)u16(s, h + 6);
        f.gp_ = (int)u16(s, h + 8);
        f.compressed_size_ = (long long)u32(s, h + 20);
        f.decompressed_size_ = (long long)u32(s, h + 24);
        f.crc_ = u32(s, h + 16);
        const std::size_t start_local = u32(s, h + 42);
        if (start_local >= s.size()) fail("Corrupted central directory file header");
        f.start_ = decode_data_start_of_lfh(s, start_local, (std::size_t)f.compressed_size_);
        if (f.crc_ == 0) f.crc_ = u32(s, start_local + 14);  // CRC 0 in the directory: take the local header's
        f.bytes_ = sp;
        m.file_ = std::move(f);
      }
      z.add(m);
      i = next;
    }
    return Result<Archive>::Ok(std::move(z));
  } catch (const Failure &f) {
    return Result<Archive>::Error(f.msg);
  }
}

// ---- encoding zipc.ml:448-588
std::size_t Archive::encoding_size() const {
  std::size_t n = eocd_min_size;
  for (const auto &kv : members_) {
    const Member &m = kv.second;
    const std::size_t data = m.is_dir() ? 0 : (std::size_t)m.file().compressed_size();
    n += lfh_min_size + m.path().size() + data + cdfh_min_size + m.path().size();
  }
  return n;
}

namespace {
int cleaned_gp_flags(const File &f) { return f.gp_flags() & ~(1 << 3) & 0xFFFF; }  // no data descriptors are written

Result<Unit> encode_eocd(std::string &b, std::size_t start, std::size_t member_count, unsigned long long cd_start,
                         unsigned long long cd_size) {
  if (cd_start > 0xFFFFFFFFull)
    return Result<Unit>::Error(strf("Maximum ZIP central directory offset 4294967295 exceeded (%lld)", (long long)cd_start));
  if (cd_size > 0xFFFFFFFFull)
    return Result<Unit>::Error(strf("Maximum ZIP central directory size 4294967295 exceeded (%lld)", (long long)cd_size));
  put32(b, start, eocd_sig);
  put16(b, start + 4, 0);   // number of this disk
  put16(b, start + 6, 0);   // disk where the directory starts
  put16(b, start + 8, (unsigned)member_count);
  put16(b, start + 10, (unsigned)member_count);
  put32(b, start + 12, (uint32)cd_size);
  put32(b, start + 16, (uint32)cd_start);
  put16(b, start + 20, 0);  // comment length
  return Result<Unit>::Ok(Unit{});
}
}  // namespace

Result<Unit> Archive::write_bytes(std::string &b, std::size_t start, const Fpath::t &first) const {
  if (b.size() < start + encoding_size()) throw std::invalid_argument("index out of bounds");
  if (is_empty()) return encode_eocd(b, start, 0, 0, 0);
  const std::size_t count = member_count();
  if (count > (std::size_t)Member::max)
    return Result<Unit>::Error(strf2("Maximum ZIP member count %lld exceeded (%lld)", Member::max, (long long)count));
  // write order: `first` if present, then the others by increasing path
  std::vector<const Member *> order;
  order.reserve(count);
  if (const Member *f = find(first)) order.push_back(f);
  for (const auto &kv : members_)
    if (kv.first != first) order.push_back(&kv.second);

  std::vector<std::size_t> lfh_at(order.size());
  std::size_t pos = start;
  for (std::size_t k = 0; k < order.size(); k++) {  // encode_member zipc.ml:465-506
    const Member &m = *order[k];
    const std::string &path = m.path();
    const auto dt = Ptime::to_dos_date_time(m.mtime());
    lfh_at[k] = pos;
    put32(b, pos, lfh_sig);
    put16(b, pos + 10, (unsigned)dt.second);
    put16(b, pos + 12, (unsigned)dt.first);
    put16(b, pos + 26, (unsigned)path.size());
    put16(b, pos + 28, 0);  // extra field length
    memcpy(&b[pos + 30], path.data(), path.size());
    if (m.is_dir()) {
      put16(b, pos + 4, (unsigned)File::version_needed_to_extract_default);
      put16(b, pos + 6, (unsigned)File::gp_default);
      put16(b, pos + 8, 0);
      put32(b, pos + 14, 0);
      put32(b, pos + 18, 0);
      put32(b, pos + 22, 0);
      pos += 30 + path.size();
    } else {
      const File &f = m.file();
      put16(b, pos + 4, (unsigned)f.version_needed_to_extract());
      put16(b, pos + 6, (unsigned)cleaned_gp_flags(f));
      put16(b, pos + 8, (unsigned)f.compression_().to_int());
      put32(b, pos + 14, f.decompressed_crc_32());
      put32(b, pos + 18, (uint32)f.compressed_size());
      put32(b, pos + 22, (uint32)f.decompressed_size());
      pos += 30 + path.size();
      memcpy(&b[pos], f.compressed_bytes().data() + f.start(), (std::size_t)f.compressed_size());
      pos += (std::size_t)f.compressed_size();
    }
  }
  const std::size_t cd_start = pos;
  for (std::size_t k = 0; k < order.size(); k++) {  // encode_cd_member zipc.ml:508-560
    const Member &m = *order[k];
    const std::string &path = m.path();
    const auto dt = Ptime::to_dos_date_time(m.mtime());
    const unsigned ext_hi = (m.is_dir() ? 040000u : 0100000u) | ((unsigned)m.mode() & 07777u);
    put32(b, pos, cdfh_sig);
    put16(b, pos + 12, (unsigned)dt.second);
    put16(b, pos + 14, (unsigned)dt.first);
    put16(b, pos + 28, (unsigned)path.size());
    put16(b, pos + 30, 0);  // extra field length
    put16(b, pos + 32, 0);  // file comment length
    put16(b, pos + 34, 0);  // disk number start
    put16(b, pos + 36, 0);  // internal file attributes
    put16(b, pos + 38, m.is_dir() ? 0x10u : 0u);
    put16(b, pos + 40, ext_hi & 0xFFFF);
    put32(b, pos + 42, (uint32)lfh_at[k]);
    memcpy(&b[pos + 46], path.data(), path.size());
    if (m.is_dir()) {
      put16(b, pos + 4, (unsigned)File::version_made_by_default);
      put16(b, pos + 6, (unsigned)File::version_needed_to_extract_default);
      put16(b, pos + 8, (unsigned)File::gp_default);
      put16(b, pos + 10, 0);
      put32(b, pos + 16, 0);
      put32(b, pos + 20, 0);
      put32(b, pos + 24, 0);
    } else {
      const File &f = m.file();
      put16(b, pos + 4, (unsigned)f.version_made_by());
      put16(b, pos + 6, (unsigned)f.version_needed_to_extract());
      put16(b, pos + 8, (unsigned)cleaned_gp_flags(f));
      put16(b, pos + 10, (unsigned)f.compression_().to_int());
      put32(b, pos + 16, f.decompressed_crc_32());
      put32(b, pos + 20, (uint32)f.compressed_size());
      put32(b, pos + 24, (uint32)f.decompressed_size());
    }
    pos += 46 + path.size();
  }
  return encode_eocd(b, pos, count, cd_start, pos - cd_start);
}

Result<std::string> Archive::to_binary_string(const Fpath::t &first) const {
  std::string b(encoding_size(), '\0');
  auto r = write_bytes(b, 0, first);
  if (!r.ok) return Result<std::string>::Error(r.error);
  return Result<std::string>::Ok(std::move(b));
}

// ---- all members at once
Result<Unit> Archive::add_deflated_files(const std::vector<NewFile> &files, std::optional<zipc_deflate::level> level) {
  std::vector<zipc_deflate::ManyItem> items(files.size());
  for (std::size_t i = 0; i < files.size(); i++) {
    items[i].data = files[i].data->data();
    items[i].len = files[i].data->size();
  }
  auto res = zipc_deflate::crc_32_and_deflate_many(items, level);
  std::vector<Member> made;
  made.reserve(files.size());
  for (std::size_t i = 0; i < files.size(); i++) {
    if (!res[i].ok) return Result<Unit>::Error(res[i].error);
    auto f = File::make(compression::of_int(8), std::make_shared<const std::string>(std::move(res[i].value)),
                        (long long)files[i].data->size(), res[i].checksum);
    if (!f.ok) return Result<Unit>::Error(f.error);
    auto m = Member::make(files[i].path, std::move(f.value), files[i].mtime, files[i].mode);
    if (!m.ok) return Result<Unit>::Error(m.error);
    made.push_back(std::move(m.value));
  }
  for (const Member &m : made) add(m);
  return Result<Unit>::Ok(Unit{});
}

std::vector<std::pair<Fpath::t, Result<std::string>>> Archive::extract_all() const {
  std::vector<std::pair<Fpath::t, Result<std::string>>> out;
  std::vector<zipc_deflate::ManyItem> items;
  std::vector<std::size_t> slot;  // index in out of each batched (Deflate) item
  for (const auto &kv : members_) {
    const Member &m = kv.second;
    if (m.is_dir()) continue;
    const File &f = m.file();
    if (f.compression_().kind == compression::Deflate && !f.is_encrypted()) {
      zipc_deflate::ManyItem it;
      it.data = f.compressed_bytes().data() + f.start();
      it.len = (std::size_t)f.compressed_size();
      it.decompressed_size = (std::size_t)f.decompressed_size();
      items.push_back(it);
      slot.push_back(out.size());
      out.push_back({m.path(), Result<std::string>::Error("")});
    } else {
      out.push_back({m.path(), f.to_binary_string()});  // stored, encrypted, unsupported: the single-member path
    }
  }
  auto res = zipc_deflate::inflate_and_crc_32_many(items);
  std::size_t k = 0;
  for (const auto &kv : members_) {
    const Member &m = kv.second;
    if (m.is_dir()) continue;
    const File &f = m.file();
    if (!(f.compression_().kind == compression::Deflate && !f.is_encrypted())) continue;
    auto &dst = out[slot[k]].second;
    if (!res[k].ok) dst = Result<std::string>::Error("deflate: " + res[k].error);
    else {
      auto c = zipc_deflate::Crc_32::check(f.decompressed_crc_32(), res[k].checksum);
      dst = c.ok ? Result<std::string>::Ok(std::move(res[k].value)) : Result<std::string>::Error(c.error);
    }
    k++;
  }
  return out;
}

std::vector<std::pair<Fpath::t, Result<Unit>>> Archive::test_all() const {
  std::vector<std::pair<Fpath::t, Result<Unit>>> out;
  std::vector<zipc_deflate::ManyItem> items;
  std::vector<std::size_t> slot;
  std::vector<const File *> files;
  for (const auto &kv : members_) {
    const Member &m = kv.second;
    if (m.is_dir()) continue;
    const File &f = m.file();
    if (f.compression_().kind == compression::Deflate && !f.is_encrypted()) {
      zipc_deflate::ManyItem it;
      it.data = f.compressed_bytes().data() + f.start();
      it.len = (std::size_t)f.compressed_size();
      it.decompressed_size = (std::size_t)f.decompressed_size();
      items.push_back(it);
      slot.push_back(out.size());
      files.push_back(&f);
      out.push_back({m.path(), Result<Unit>::Error("")});
    } else {
      auto r = f.to_binary_string();  // stored, encrypted, unsupported: the single-member path
      out.push_back({m.path(), r.ok ? Result<Unit>::Ok(Unit{}) : Result<Unit>::Error(r.error)});
    }
  }
  auto res = zipc_deflate::inflate_and_crc_32_many_check(items);
  for (std::size_t k = 0; k < items.size(); k++) {
    auto &dst = out[slot[k]].second;
    if (!res[k].ok) dst = Result<Unit>::Error("deflate: " + res[k].error);
    else {
      auto c = zipc_deflate::Crc_32::check(files[k]->decompressed_crc_32(), res[k].checksum);
      dst = c.ok ? Result<Unit>::Ok(Unit{}) : Result<Unit>::Error(c.error);
    }
  }
  return out;
}

}  // namespace zipc
