// zipc_tool.cpp -- `zipc-hip`: a command line over the host layer, after the
// reference's test/zipc_tool.ml.  Same command names and
// the options that matter for driving the codec over files; I/O-bound, no kernel
// of its own.  Reads whole inputs in memory like the reference's tool does.
//
//   zipc-hip crc [-a|--adler-32] [INPUT]                 hex checksum (CRC-32 default)
//   zipc-hip compress [--zlib] [--level LEVEL] [-o OUT] [INPUT] deflate (RFC 1951) / zlib
//   zipc-hip decompress [--zlib] [-o OUT] [INPUT]
//   zipc-hip list [-s|-l] ARCHIVE                         members, Member.pp / pp_long lines
//   zipc-hip sniff [-0] [-r] [-P] PATH...                 the paths that begin with a ZIP magic number, one a line (-0: NUL
//                                                         separated); -r: directories are walked, -P: symlinks not followed.
//                                                         A FILE argument without the magic: a message and exit 4   (zipc_tool.ml:556-579)
//   zipc-hip unzip [-t] [--skip] [-v] [-d DIR] ARCHIVE    -t: decode and CRC-check every member, write nothing; exit 2 when a
//                                                         member is corrupted, 3 when one cannot be decoded (encrypted, a format
//                                                         other than stored / deflate) unless --skip     (zipc_tool.ml:635-660,727-749)
//   zipc-hip zip [--level LEVEL] -o ARCHIVE PATH...       files / directories -> archive
//   zipc-hip recode [--deflate | -u | --as-is] [--level LEVEL] [-t [--check-cmd CMD]] [-v] [-o OUT] ARCHIVE
//                                                         every member the tool can decode is written again deflated (--deflate)
//                                                         or stored (-u); the others, and all of them by default (--as-is), as
//                                                         they are.  A member whose CRC-32 changed is an error.  -t: nothing is
//                                                         written; the recoded archive is decoded again in memory, or handed to
//                                                         `CMD tmpfile` (exit 0 is success; for a CMD that starts with "unzip"
//                                                         also 1 -- empty -- and 82 -- encrypted)        (zipc_tool.ml:485-545)
// LEVEL: none | fast | default | best (absent = best, like the reference).
// `-` or no INPUT is stdin; `-o -` or no -o is stdout.  unzip / zip / recode hand all
// members to the GPU as one batch (Archive::extract_all, add_deflated_files).
// Exit codes (zipc_tool.ml:282-311): 0; 1 a requested path does not exist; 2 a member is corrupted; 3 a compression format
// is unsupported; 4 no ZIP magic; 123 some other error; 124 a command line error.  The corpus procedure of the
// reference's DEVEL.md:7-31 (sniff -0 -P -r ... | xargs -0 -L1 zipc unzip --skip -t / recode --deflate -t --check-cmd=...) is
// tools/corpus_box.py.
#include <dirent.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <set>
#include <sstream>

#include "zipc.hpp"

namespace {

enum Exit { ok = 0, err_path = 1, err_corrupted = 2, err_unsupported = 3, err_no_magic = 4, err_some = 123, err_cli = 124 };

[[noreturn]] void die(const std::string &m, int code = err_some) {
  std::cerr << "zipc-hip: " << m << "\n";
  exit(code);
}

std::string read_file(const std::string &p) {
  if (p == "-") return std::string(std::istreambuf_iterator<char>(std::cin), {});
  // (one read of the file's size: a character at a time through a stream iterator took longer than the GPU's work on the file)
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) die(p + ": cannot read");
  std::string out;
  struct stat st;
  if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
    out.resize((std::size_t)st.st_size);
    const std::size_t got = fread(&out[0], 1, out.size(), f);
    out.resize(got);
  }
  char buf[1 << 16];  // (what a pipe holds, or a file that grew)
  for (std::size_t k; (k = fread(buf, 1, sizeof buf, f)) > 0;) out.append(buf, k);
  fclose(f);
  return out;
}
void write_file(const std::string &p, const std::string &s) {
  if (p == "-") { std::cout.write(s.data(), (std::streamsize)s.size()); return; }
  FILE *f = fopen(p.c_str(), "wb");
  if (!f || fwrite(s.data(), 1, s.size(), f) != s.size() || fclose(f) != 0) die(p + ": cannot write");
}
std::optional<zipc_deflate::level> parse_level(const std::string &l) {
  if (l == "none") return zipc_deflate::level::None;
  if (l == "fast") return zipc_deflate::level::Fast;
  if (l == "default") return zipc_deflate::level::Default;
  if (l == "best") return zipc_deflate::level::Best;
  die("unknown deflate level '" + l + "' (none, fast, default or best)", err_cli);
}

struct Args {
  bool zlib = false, adler = false, test = false, short_out = false, long_out = false;
  bool nul_sep = false, recurse = false, no_follow = false, skip = false, verbose = false;
  int recode_as = 0;  // 0 as is, 1 deflate, 2 stored
  std::optional<std::string> check_cmd;
  std::optional<zipc_deflate::level> level;
  std::string out = "-", dir = ".";
  std::vector<std::string> pos;
};
Args parse(int argc, char **argv, int from) {
  Args a;
  for (int i = from; i < argc; i++) {
    const std::string s = argv[i];
    auto value = [&]() -> std::string {
      if (i + 1 >= argc) die("option " + s + " needs a value", err_cli);
      return argv[++i];
    };
    if (s == "--zlib") a.zlib = true;
    else if (s == "-a" || s == "--adler-32") a.adler = true;
    else if (s == "-z" || s == "--zip-crc-32") a.adler = false;
    else if (s == "-t" || s == "--test" || s == "--check") a.test = true;
    else if (s == "-0") a.nul_sep = true;
    else if (s == "-r" || s == "--recurse") a.recurse = true;
    else if (s == "-P" || s == "--no-dereference") a.no_follow = true;
    else if (s == "--skip") a.skip = true;
    else if (s == "-v" || s == "--verbose") a.verbose = true;
    else if (s == "--deflate") a.recode_as = 1;
    else if (s == "-u" || s == "--stored") a.recode_as = 2;
    else if (s == "--as-is") a.recode_as = 0;
    else if (s == "--check-cmd") a.check_cmd = value();
    else if (s.rfind("--check-cmd=", 0) == 0) a.check_cmd = s.substr(12);
    else if (s == "-s" || s == "--short") a.short_out = true;
    else if (s == "-l" || s == "--long") a.long_out = true;
    else if (s == "--level") a.level = parse_level(value());
    else if (s == "-o") a.out = value();
    else if (s == "-d") a.dir = value();
    else if (s.size() > 1 && s[0] == '-') die("unknown option " + s, err_cli);
    else a.pos.push_back(s);
  }
  return a;
}

// the first four bytes of a file ("" when it is shorter or cannot be read: read_magic, zipc_tool.ml:217-224)
bool read_magic(const std::string &path, std::string &magic, std::string &err) {
  magic.clear();
  if (path == "-") { char b[4]; std::cin.read(b, 4); if (std::cin.gcount() == 4) magic.assign(b, 4); return true; }
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = path + ": " + strerror(errno); return false; }
  char b[4];
  f.read(b, 4);
  if (f.gcount() == 4) magic.assign(b, 4);
  return true;
}

// sniff (zipc_tool.ml:556-579): regular files below `dir` that begin with the magic; dot files too, directories that
// cannot be read are passed over, symlinks followed unless -P
void sniff_dir(const std::string &dir, const Args &a, std::set<std::pair<dev_t, ino_t>> &seen) {
  DIR *d = opendir(dir.c_str());
  if (!d) return;  // (Os.dir_prune_denied)
  std::vector<std::string> names;
  while (dirent *e = readdir(d))
    if (strcmp(e->d_name, ".") && strcmp(e->d_name, "..")) names.push_back(e->d_name);
  closedir(d);
  std::sort(names.begin(), names.end());
  for (const auto &n : names) {
    const std::string p = dir + (dir.size() && dir.back() == '/' ? "" : "/") + n;
    struct stat st;
    if ((a.no_follow ? lstat(p.c_str(), &st) : stat(p.c_str(), &st)) != 0) continue;
    if (S_ISDIR(st.st_mode)) {
      if (!a.recurse) continue;
      if (!seen.insert({st.st_dev, st.st_ino}).second) continue;  // (a symlink cycle)
      sniff_dir(p, a, seen);
    } else if (S_ISREG(st.st_mode)) {
      std::string magic, err;
      if (!read_magic(p, magic, err)) { std::cerr << "zipc-hip: " << err << "\n"; continue; }
      if (zipc::Archive::string_has_magic(magic)) { std::cout << p << (a.nul_sep ? '\0' : '\n'); }
    }
  }
}

std::string pct(long long num, long long den) {
  char b[32];
  snprintf(b, sizeof b, "%d%%", den ? (int)((double)num / (double)den * 100.0) : 0);
  return b;
}

// the recoded archive decoded again in memory (recode_check_in_memory, zipc_tool.ml:503-521)
int recode_check_in_memory(const std::string &archive, std::size_t oldlen, const std::string &recoded, bool verbose) {
  auto z = zipc::Archive::of_binary_string(recoded);
  if (!z.ok) die("recode check: " + z.error);
  int exit_code = ok;
  for (const auto &r : z.value.test_all())
    if (!r.second.ok && z.value.find(r.first) && z.value.find(r.first)->file().can_extract()) {
      if (verbose) std::cerr << archive << ": " << r.first << ": " << r.second.error << "\n";
      exit_code = err_corrupted;
    }
  if (exit_code != ok) std::cerr << "zipc-hip: " << archive << ": Some recoded archive members had errors\n";
  else if (verbose) std::cerr << "No errors in " << archive << " recode (" << pct((long long)recoded.size(), (long long)oldlen) << " of old size)\n";
  return exit_code;
}

// ... or handed to a command (recode_check_with_cmd, zipc_tool.ml:485-501)
int recode_check_with_cmd(const std::string &archive, std::size_t oldlen, const std::string &recoded, const std::string &cmd, bool verbose) {
  char tmpl[] = "/tmp/zipcXXXXXX.zip";
  const int fd = mkstemps(tmpl, 4);
  if (fd < 0) die("recode check: cannot make a temporary file");
  close(fd);
  write_file(tmpl, recoded);
  const int rc = system((cmd + " " + tmpl).c_str());
  unlink(tmpl);
  const int code = rc == -1 ? -1 : (WIFEXITED(rc) ? WEXITSTATUS(rc) : 128 + WTERMSIG(rc));
  const bool is_unzip = cmd.rfind("unzip", 0) == 0;
  if (code == 0 || (is_unzip && (code == 1 /* empty */ || code == 82 /* encrypted */))) {
    if (verbose) std::cerr << "No errors in " << archive << " recode (" << pct((long long)recoded.size(), (long long)oldlen) << " of old size)\n";
    return ok;
  }
  std::cerr << "zipc-hip: " << archive << ": check command returned " << code << "\n";
  return err_some;
}

zipc::Archive load_archive(const std::string &path) {
  auto r = zipc::Archive::of_binary_string(read_file(path));
  if (!r.ok) die(path + ": " + r.error);
  return r.value;
}

void mkdirs(const std::string &p) {
  for (std::size_t i = 1; i <= p.size(); i++)
    if (i == p.size() || p[i] == '/') mkdir(p.substr(0, i).c_str(), 0755);
}

void collect(const std::string &fs_path, const std::string &zip_path, zipc::Archive &z,
             std::vector<std::string> &datas, std::vector<zipc::Archive::NewFile> &files) {
  struct stat st;
  if (stat(fs_path.c_str(), &st) != 0) die(fs_path + ": cannot stat");
  if (S_ISDIR(st.st_mode)) {
    auto m = zipc::Member::make(zip_path, std::nullopt, (zipc::Ptime::t)st.st_mtime, (int)(st.st_mode & 07777));
    if (!m.ok) die(m.error);
    z.add(m.value);
    DIR *d = opendir(fs_path.c_str());
    if (!d) die(fs_path + ": cannot open directory");
    std::vector<std::string> names;
    while (dirent *e = readdir(d))
      if (strcmp(e->d_name, ".") && strcmp(e->d_name, "..")) names.push_back(e->d_name);
    closedir(d);
    for (const auto &n : names) collect(fs_path + "/" + n, zipc::Fpath::ensure_directoryness(zip_path) + n, z, datas, files);
  } else {
    datas.push_back(read_file(fs_path));
    zipc::Archive::NewFile f;
    f.path = zip_path;
    f.mtime = (zipc::Ptime::t)st.st_mtime;
    f.mode = (int)(st.st_mode & 07777);
    files.push_back(f);
  }
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) die("usage: zipc-hip crc|compress|decompress|list|sniff|unzip|zip|recode ... (see zipc_tool.cpp)", 124);
  const std::string cmd = argv[1];
  const Args a = parse(argc, argv, 2);
  const std::string in = a.pos.empty() ? "-" : a.pos[0];
  try {
    if (cmd == "crc") {
      const std::string s = read_file(in);
      std::cout << (a.adler ? zipc_deflate::Adler_32::pp(zipc_deflate::Adler_32::string(s))
                            : zipc_deflate::Crc_32::pp(zipc_deflate::Crc_32::string(s)))
                << "\n";
    } else if (cmd == "compress") {
      const std::string s = read_file(in);
      auto r = a.zlib ? zipc_deflate::zlib_compress(s, a.level) : zipc_deflate::deflate(s, a.level);
      if (!r.ok) die(r.error);
      write_file(a.out, r.value);
    } else if (cmd == "decompress") {
      const std::string s = read_file(in);
      if (a.zlib) {
        auto r = zipc_deflate::zlib_decompress(s);
        if (!r.ok) die(r.error.message);
        write_file(a.out, r.value);
      } else {
        auto r = zipc_deflate::inflate(s);
        if (!r.ok) die(r.error);
        write_file(a.out, r.value);
      }
    } else if (cmd == "sniff") {
      int exit_code = ok;
      std::set<std::pair<dev_t, ino_t>> seen;
      for (const auto &p : a.pos.empty() ? std::vector<std::string>{"-"} : a.pos) {
        struct stat st;
        if (p != "-" && stat(p.c_str(), &st) != 0) { std::cerr << "zipc-hip: " << p << ": " << strerror(errno) << "\n"; exit_code = err_some; continue; }
        if (p != "-" && S_ISDIR(st.st_mode)) { sniff_dir(p, a, seen); continue; }
        std::string magic, err;
        if (!read_magic(p, magic, err)) { std::cerr << "zipc-hip: " << err << "\n"; exit_code = err_some; continue; }
        if (zipc::Archive::string_has_magic(magic)) std::cout << p << (a.nul_sep ? '\0' : '\n');
        else { if (exit_code == ok) exit_code = err_no_magic; std::cerr << "zipc-hip: " << p << ": Not a ZIP archive\n"; }
      }
      return exit_code;
    } else if (cmd == "list") {
      const zipc::Archive z = load_archive(in);
      z.fold([&](const zipc::Member &m) { std::cout << (a.short_out ? m.path() : m.pp(a.long_out)) << "\n"; });
    } else if (cmd == "unzip") {
      const zipc::Archive z = load_archive(in);
      int bad = 0;
      if (a.test) {  // check_archive, zipc_tool.ml:635-660: every member decoded and CRC-checked on the GPU, only the verdicts come back
        const auto tested = z.test_all();
        std::map<std::string, const zipc_deflate::Result<zipc_deflate::Unit> *> by_path;
        for (const auto &r : tested) by_path[r.first] = &r.second;
        int exit_code = ok;
        std::size_t files = 0;
        z.fold([&](const zipc::Member &m) {
          if (m.is_dir()) { if (a.verbose) std::cerr << "[----] " << m.path() << "\n"; return; }
          files++;
          const zipc::File &f = m.file();
          if (f.can_extract()) {
            const auto it = by_path.find(m.path());
            if (it != by_path.end() && it->second->ok) { if (a.verbose) std::cerr << "[ OK ] " << m.path() << "\n"; return; }
            const std::string e = it == by_path.end() ? "not decoded" : it->second->error;
            std::cerr << (a.verbose ? "[FAIL] " : "") << m.path() << ": " << e << "\n";
            exit_code = err_corrupted;
            return;
          }
          if (a.verbose) std::cerr << "[ ?? ] " << m.path() << (f.is_encrypted() ? " encrypted " : " ") << f.compression_().to_string() << "\n";
          if (a.skip) return;
          if (f.is_encrypted()) std::cerr << "zipc-hip: " << in << ": " << m.path() << ": Cannot decompress encrypted file\n";
          else std::cerr << "zipc-hip: " << in << ": " << m.path() << ": Cannot decompress format " << f.compression_().to_string() << "\n";
          if (exit_code == ok) exit_code = err_unsupported;
        });
        if (exit_code != ok) std::cerr << "zipc-hip: " << in << ": Some archive members had errors\n";
        std::cout << (exit_code != ok ? "Errors detected in " : "No errors detected in ") << in << " (" << files << " files)\n";
        return exit_code;
      }
      const auto res = z.extract_all();  // all file members as one batch on the GPU
      z.fold([&](const zipc::Member &m) { if (m.is_dir()) mkdirs(a.dir + "/" + zipc::Fpath::sanitize(m.path())); });
      for (const auto &r : res) {
        if (!r.second.ok) { std::cerr << r.first << ": " << r.second.error << "\n"; bad++; continue; }
        const std::string p = a.dir + "/" + zipc::Fpath::sanitize(r.first);
        const std::size_t slash = p.rfind('/');
        if (slash != std::string::npos) mkdirs(p.substr(0, slash));
        write_file(p, r.second.value);
        if (const zipc::Member *m = z.find(r.first)) chmod(p.c_str(), (mode_t)m->mode());
      }
      return bad ? (int)err_corrupted : (int)ok;
    } else if (cmd == "zip") {
      if (a.out == "-" && isatty(1)) die("refusing to write an archive to a terminal (use -o)", 124);
      zipc::Archive z;
      std::vector<std::string> datas;
      std::vector<zipc::Archive::NewFile> files;
      for (const auto &p : a.pos) {
        std::string zp = zipc::Fpath::sanitize(p);
        collect(p, zp, z, datas, files);
      }
      for (std::size_t i = 0; i < files.size(); i++) files[i].data = &datas[i];
      auto r = z.add_deflated_files(files, a.level);  // one batch on the GPU
      if (!r.ok) die(r.error);
      auto enc = z.to_binary_string();
      if (!enc.ok) die(enc.error);
      write_file(a.out, enc.value);
    } else if (cmd == "recode") {  // zipc_tool.ml:437-545
      const std::string raw = read_file(in);
      auto zr = zipc::Archive::of_binary_string(raw);
      if (!zr.ok) die(in + ": " + zr.error);
      const zipc::Archive &z = zr.value;
      zipc::Archive out = z;
      if (a.recode_as != 0) {
        std::vector<std::string> datas;
        std::vector<zipc::Archive::NewFile> files;
        for (const auto &r : z.extract_all()) {  // (members that cannot be decoded are kept as they are; so are directories)
          const zipc::Member *m = z.find(r.first);
          if (!m || m->is_dir() || !m->file().can_extract()) continue;  // (File.can_extract: kept as it is, zipc_tool.ml:432)
          if (!r.second.ok) die(r.first + ": " + r.second.error);
          datas.push_back(r.second.value);
          zipc::Archive::NewFile f;
          f.path = r.first;
          f.mtime = m->mtime();
          f.mode = m->mode();
          files.push_back(f);
        }
        for (std::size_t i = 0; i < files.size(); i++) files[i].data = &datas[i];
        if (a.recode_as == 1) {
          auto r = out.add_deflated_files(files, a.level);  // one batch on the GPU
          if (!r.ok) die(r.error);
        } else {
          for (const auto &f : files) {
            auto file = zipc::File::stored_of_binary_string(*f.data);
            if (!file.ok) die(f.path + ": " + file.error);
            auto m = zipc::Member::make(f.path, file.value, f.mtime, f.mode);
            if (!m.ok) die(m.error);
            out.add(m.value);
          }
        }
        for (const auto &f : files) {  // err_checksum, zipc_tool.ml:426-428
          const zipc_deflate::uint32 was = z.find(f.path)->file().decompressed_crc_32(), now = out.find(f.path)->file().decompressed_crc_32();
          if (was != now)
            die(f.path + ": Recoding changed the checksum from " + zipc_deflate::Crc_32::pp(was) + " to " + zipc_deflate::Crc_32::pp(now) + " (zipc bug)");
          if (a.verbose) {
            const zipc::File &o = z.find(f.path)->file(), &n = out.find(f.path)->file();
            std::cerr << "Recode " << pct(n.compressed_size(), n.decompressed_size()) << " (was " << pct(o.compressed_size(), o.decompressed_size()) << ") " << f.path << "\n";
          }
        }
      }
      auto enc = out.to_binary_string();
      if (!enc.ok) die(in + ": " + enc.error);
      if (a.test) return a.check_cmd ? recode_check_with_cmd(in, raw.size(), enc.value, *a.check_cmd, a.verbose) : recode_check_in_memory(in, raw.size(), enc.value, a.verbose);
      write_file(a.out, enc.value);
    } else {
      die("unknown command " + cmd, 124);
    }
  } catch (const std::exception &e) {
    die(e.what(), 123);
  }
  return 0;
}
