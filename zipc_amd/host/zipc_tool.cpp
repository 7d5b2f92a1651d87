// zipc_tool.cpp -- `zipc-hip`: a command line over the host layer, after the
// reference's test/zipc_tool.ml.  Same command names and
// the options that matter for driving the codec over files; I/O-bound, no kernel
// of its own.  Reads whole inputs in memory like the reference's tool does.
//
//   zipc-hip crc [-a|--adler-32] [INPUT]                 hex checksum (CRC-32 default)
//   zipc-hip compress [--zlib] [--level LEVEL] [-o OUT] [INPUT] deflate (RFC 1951) / zlib
//   zipc-hip decompress [--zlib] [-o OUT] [INPUT]
//   zipc-hip list [-s|-l] ARCHIVE                         members, Member.pp / pp_long lines
//   zipc-hip sniff FILE                                   exit 0 when FILE looks like a ZIP
//   zipc-hip unzip [-t] [-d DIR] ARCHIVE                  test (CRC check) or extract every member
//   zipc-hip zip [--level LEVEL] -o ARCHIVE PATH...       files / directories -> archive
//   zipc-hip recode [--level LEVEL] -o OUT ARCHIVE        re-deflate every extractable file member
// LEVEL: none | fast | default | best (absent = best, like the reference).
// `-` or no INPUT is stdin; `-o -` or no -o is stdout.  unzip / zip / recode hand all
// members to the GPU as one batch (Archive::extract_all, add_deflated_files).
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>

#include "zipc.hpp"

namespace {

[[noreturn]] void die(const std::string &m, int code = 1) {
  std::cerr << "zipc-hip: " << m << "\n";
  exit(code);
}

std::string read_file(const std::string &p) {
  if (p == "-") return std::string(std::istreambuf_iterator<char>(std::cin), {});
  std::ifstream f(p, std::ios::binary);
  if (!f) die(p + ": cannot read");
  return std::string(std::istreambuf_iterator<char>(f), {});
}
void write_file(const std::string &p, const std::string &s) {
  if (p == "-") { std::cout.write(s.data(), (std::streamsize)s.size()); return; }
  std::ofstream f(p, std::ios::binary);
  if (!f || !f.write(s.data(), (std::streamsize)s.size())) die(p + ": cannot write");
}
std::optional<zipc_deflate::level> parse_level(const std::string &l) {
  if (l == "none") return zipc_deflate::level::None;
  if (l == "fast") return zipc_deflate::level::Fast;
  if (l == "default") return zipc_deflate::level::Default;
  if (l == "best") return zipc_deflate::level::Best;
  die("unknown deflate level '" + l + "' (none, fast, default or best)", 124);
}

struct Args {
  bool zlib = false, adler = false, test = false, short_out = false, long_out = false;
  std::optional<zipc_deflate::level> level;
  std::string out = "-", dir = ".";
  std::vector<std::string> pos;
};
Args parse(int argc, char **argv, int from) {
  Args a;
  for (int i = from; i < argc; i++) {
    const std::string s = argv[i];
    auto value = [&]() -> std::string {
      if (i + 1 >= argc) die("option " + s + " needs a value", 124);
      return argv[++i];
    };
    if (s == "--zlib") a.zlib = true;
    else if (s == "-a" || s == "--adler-32") a.adler = true;
    else if (s == "-z" || s == "--zip-crc-32") a.adler = false;
    else if (s == "-t" || s == "--test") a.test = true;
    else if (s == "-s" || s == "--short") a.short_out = true;
    else if (s == "-l" || s == "--long") a.long_out = true;
    else if (s == "--level") a.level = parse_level(value());
    else if (s == "-o") a.out = value();
    else if (s == "-d") a.dir = value();
    else if (s.size() > 1 && s[0] == '-') die("unknown option " + s, 124);
    else a.pos.push_back(s);
  }
  return a;
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
      return zipc::Archive::string_has_magic(read_file(in)) ? 0 : 1;
    } else if (cmd == "list") {
      const zipc::Archive z = load_archive(in);
      z.fold([&](const zipc::Member &m) { std::cout << (a.short_out ? m.path() : m.pp(a.long_out)) << "\n"; });
    } else if (cmd == "unzip") {
      const zipc::Archive z = load_archive(in);
      int bad = 0;
      const auto res = z.extract_all();  // all file members as one batch on the GPU
      if (!a.test) z.fold([&](const zipc::Member &m) { if (m.is_dir()) mkdirs(a.dir + "/" + zipc::Fpath::sanitize(m.path())); });
      for (const auto &r : res) {
        if (!r.second.ok) { std::cerr << r.first << ": " << r.second.error << "\n"; bad++; continue; }
        if (a.test) continue;
        const std::string p = a.dir + "/" + zipc::Fpath::sanitize(r.first);
        const std::size_t slash = p.rfind('/');
        if (slash != std::string::npos) mkdirs(p.substr(0, slash));
        write_file(p, r.second.value);
        if (const zipc::Member *m = z.find(r.first)) chmod(p.c_str(), (mode_t)m->mode());
      }
      if (a.test) std::cout << (bad ? "Errors detected in " : "No errors detected in ") << in << " (" << res.size() << " files)\n";
      return bad ? 1 : 0;
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
    } else if (cmd == "recode") {
      const zipc::Archive z = load_archive(in);
      zipc::Archive out = z;
      std::vector<std::string> datas;
      std::vector<zipc::Archive::NewFile> files;
      for (const auto &r : z.extract_all()) {
        if (!r.second.ok) continue;  // members that cannot be extracted are kept as they are
        const zipc::Member *m = z.find(r.first);
        datas.push_back(r.second.value);
        zipc::Archive::NewFile f;
        f.path = r.first;
        f.mtime = m->mtime();
        f.mode = m->mode();
        files.push_back(f);
      }
      for (std::size_t i = 0; i < files.size(); i++) files[i].data = &datas[i];
      auto r = out.add_deflated_files(files, a.level);
      if (!r.ok) die(r.error);
      auto enc = out.to_binary_string();
      if (!enc.ok) die(enc.error);
      write_file(a.out, enc.value);
    } else {
      die("unknown command " + cmd, 124);
    }
  } catch (const std::exception &e) {
    die(e.what(), 123);
  }
  return 0;
}
