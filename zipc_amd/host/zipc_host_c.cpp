// zipc_host_c.cpp -- the C view of zipc.hpp declared in include/zipc_host.h.
#include "../../include/zipc_host.h"

#include <cstring>

#include "zipc.hpp"

struct zipc_host_archive {
  zipc::Archive z;
};
struct zipc_host_extraction {
  std::vector<std::pair<std::string, zipc::Result<std::string>>> items;
};

namespace {
void set_err(char *err, size_t cap, const std::string &m) {
  if (!err || !cap) return;
  const size_t n = m.size() < cap - 1 ? m.size() : cap - 1;
  memcpy(err, m.data(), n);
  err[n] = 0;
}
size_t put_str(const std::string &s, char *buf, size_t cap) {
  if (buf && cap) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return s.size();
}
// runs f, mapping the exceptions of the C++ layer to the C return codes
template <class F>
int guarded(char *err, size_t cap, F f) {
  try {
    return f();
  } catch (const std::invalid_argument &e) {
    set_err(err, cap, e.what());
    return ZIPC_HOST_INVALID;
  } catch (const std::exception &e) {
    set_err(err, cap, e.what());
    return ZIPC_HOST_FAILURE;
  }
}
const zipc::Member *member_at(const zipc_host_archive *a, size_t index) {
  if (!a || index >= a->z.member_count()) return nullptr;
  auto it = a->z.to_string_map().begin();
  std::advance(it, (long)index);
  return &it->second;
}
std::optional<zipc::Ptime::t> opt_mtime(const zipc_host_member_opts *o) {
  return (o && o->has_mtime) ? std::optional<zipc::Ptime::t>(o->mtime) : std::nullopt;
}
std::optional<int> opt_mode(const zipc_host_member_opts *o) {
  return (o && o->has_mode) ? std::optional<int>(o->mode) : std::nullopt;
}
std::optional<zipc_deflate::level> opt_level(int level) {
  if (level < 0) return std::nullopt;
  return (zipc_deflate::level)level;
}
int add_member(zipc_host_archive *a, const std::string &path, zipc::Result<zipc::File> f, const zipc_host_member_opts *o,
               char *err, size_t errcap) {
  if (!f.ok) { set_err(err, errcap, f.error); return ZIPC_HOST_ERROR; }
  auto m = zipc::Member::make(path, std::move(f.value), opt_mtime(o), opt_mode(o));
  if (!m.ok) { set_err(err, errcap, m.error); return ZIPC_HOST_ERROR; }
  a->z.add(m.value);
  return ZIPC_HOST_OK;
}
}  // namespace

extern "C" {

zipc_host_archive *zipc_host_empty(void) { return new zipc_host_archive(); }

int zipc_host_of_binary_string(const void *s, size_t len, zipc_host_archive **out, char *err, size_t errcap) {
  if (!out || (!s && len)) return ZIPC_HOST_INVALID;
  *out = nullptr;
  return guarded(err, errcap, [&] {
    auto r = zipc::Archive::of_binary_string(std::make_shared<const std::string>((const char *)s, len));
    if (!r.ok) { set_err(err, errcap, r.error); return (int)ZIPC_HOST_ERROR; }
    *out = new zipc_host_archive{std::move(r.value)};
    return (int)ZIPC_HOST_OK;
  });
}
void zipc_host_free(zipc_host_archive *a) { delete a; }
int zipc_host_string_has_magic(const void *s, size_t len) {
  return zipc::Archive::string_has_magic(std::string((const char *)s, len)) ? 1 : 0;
}

size_t zipc_host_member_count(const zipc_host_archive *a) { return a ? a->z.member_count() : 0; }

int zipc_host_member_at(const zipc_host_archive *a, size_t index, zipc_host_member *m) {
  const zipc::Member *mm = member_at(a, index);
  if (!mm || !m) return ZIPC_HOST_INVALID;
  memset(m, 0, sizeof *m);
  m->path = mm->path().data();
  m->path_len = mm->path().size();
  m->is_dir = mm->is_dir();
  m->mode = mm->mode();
  m->mtime = mm->mtime();
  if (!mm->is_dir()) {
    const zipc::File &f = mm->file();
    m->compression = f.compression_().to_int();
    m->gp_flags = f.gp_flags();
    m->version_made_by = f.version_made_by();
    m->version_needed_to_extract = f.version_needed_to_extract();
    m->start = f.start();
    m->compressed_size = (uint64_t)f.compressed_size();
    m->decompressed_size = (uint64_t)f.decompressed_size();
    m->decompressed_crc_32 = f.decompressed_crc_32();
    m->is_encrypted = f.is_encrypted();
    m->can_extract = f.can_extract();
  }
  return ZIPC_HOST_OK;
}

int zipc_host_find(const zipc_host_archive *a, const char *path, size_t path_len, size_t *index) {
  if (!a || !index) return ZIPC_HOST_INVALID;
  const auto &m = a->z.to_string_map();
  auto it = m.find(std::string(path, path_len));
  if (it == m.end()) return ZIPC_HOST_ERROR;
  *index = (size_t)std::distance(m.begin(), it);
  return ZIPC_HOST_OK;
}
int zipc_host_remove(zipc_host_archive *a, const char *path, size_t path_len) {
  if (!a) return ZIPC_HOST_INVALID;
  a->z.remove(std::string(path, path_len));
  return ZIPC_HOST_OK;
}
size_t zipc_host_member_pp(const zipc_host_archive *a, size_t index, int long_form, char *buf, size_t cap) {
  const zipc::Member *mm = member_at(a, index);
  if (!mm) return 0;
  return put_str(mm->pp(long_form != 0), buf, cap);
}

int zipc_host_add_dir(zipc_host_archive *a, const char *path, size_t path_len, const zipc_host_member_opts *o,
                      char *err, size_t errcap) {
  if (!a) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    auto m = zipc::Member::make(std::string(path, path_len), std::nullopt, opt_mtime(o), opt_mode(o));
    if (!m.ok) { set_err(err, errcap, m.error); return (int)ZIPC_HOST_ERROR; }
    a->z.add(m.value);
    return (int)ZIPC_HOST_OK;
  });
}

int zipc_host_add_file_made(zipc_host_archive *a, const char *path, size_t path_len, int compression,
                            const void *bytes, size_t bytes_len, size_t start, int64_t compressed_size,
                            int64_t decompressed_size, uint32_t crc, int gp_flags, int version_made_by,
                            int version_needed_to_extract, const zipc_host_member_opts *o, char *err, size_t errcap) {
  if (!a) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    zipc::File::make_args ma;
    ma.gp_flags = gp_flags;
    ma.version_made_by = version_made_by;
    ma.version_needed_to_extract = version_needed_to_extract;
    ma.start = start;
    if (compressed_size >= 0) ma.compressed_size = compressed_size;
    auto f = zipc::File::make(zipc::compression::of_int(compression),
                              std::make_shared<const std::string>((const char *)bytes, bytes_len), decompressed_size, crc, ma);
    return add_member(a, std::string(path, path_len), std::move(f), o, err, errcap);
  });
}

int zipc_host_add_file_stored(zipc_host_archive *a, const char *path, size_t path_len, const void *data, size_t len,
                              const zipc_host_member_opts *o, char *err, size_t errcap) {
  if (!a) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    return add_member(a, std::string(path, path_len),
                      zipc::File::stored_of_binary_string(std::string((const char *)data, len)), o, err, errcap);
  });
}
int zipc_host_add_file_deflate(zipc_host_archive *a, const char *path, size_t path_len, const void *data, size_t len,
                               int level, const zipc_host_member_opts *o, char *err, size_t errcap) {
  if (!a || level > 3) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    return add_member(a, std::string(path, path_len),
                      zipc::File::deflate_of_binary_string(std::string((const char *)data, len), opt_level(level)), o,
                      err, errcap);
  });
}
int zipc_host_add_files_deflate(zipc_host_archive *a, size_t n, const char *const *paths, const size_t *path_lens,
                                const void *const *datas, const size_t *lens, int level, char *err, size_t errcap) {
  if (!a || level > 3) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    std::vector<std::string> data(n);
    std::vector<zipc::Archive::NewFile> files(n);
    for (size_t i = 0; i < n; i++) {
      data[i].assign((const char *)datas[i], lens[i]);
      files[i].path.assign(paths[i], path_lens[i]);
      files[i].data = &data[i];
    }
    auto r = a->z.add_deflated_files(files, opt_level(level));
    if (!r.ok) { set_err(err, errcap, r.error); return (int)ZIPC_HOST_ERROR; }
    return (int)ZIPC_HOST_OK;
  });
}

size_t zipc_host_encoding_size(const zipc_host_archive *a) { return a ? a->z.encoding_size() : 0; }

int zipc_host_to_binary_string(const zipc_host_archive *a, const char *first, size_t first_len, void *dst, size_t cap,
                               size_t *out_len, char *err, size_t errcap) {
  if (!a || !out_len) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    auto r = first ? a->z.to_binary_string(std::string(first, first_len)) : a->z.to_binary_string();
    if (!r.ok) { set_err(err, errcap, r.error); return (int)ZIPC_HOST_ERROR; }
    *out_len = r.value.size();
    if (r.value.size() > cap) { set_err(err, errcap, "destination too small"); return (int)ZIPC_HOST_INVALID; }
    memcpy(dst, r.value.data(), r.value.size());
    return (int)ZIPC_HOST_OK;
  });
}

int zipc_host_member_to_binary_string(const zipc_host_archive *a, size_t index, int check_crc, void *dst, size_t cap,
                                      size_t *out_len, uint32_t *crc, char *err, size_t errcap) {
  const zipc::Member *mm = member_at(a, index);
  if (!mm || mm->is_dir() || !out_len) return ZIPC_HOST_INVALID;
  return guarded(err, errcap, [&] {
    std::string s;
    if (check_crc) {
      auto r = mm->file().to_binary_string();
      if (!r.ok) { set_err(err, errcap, r.error); return (int)ZIPC_HOST_ERROR; }
      s = std::move(r.value);
      if (crc) *crc = mm->file().decompressed_crc_32();
    } else {
      auto r = mm->file().to_binary_string_no_crc_check();
      if (!r.ok) { set_err(err, errcap, r.error); return (int)ZIPC_HOST_ERROR; }
      s = std::move(r.value.first);
      if (crc) *crc = r.value.second;
    }
    *out_len = s.size();
    if (s.size() > cap) { set_err(err, errcap, "destination too small"); return (int)ZIPC_HOST_INVALID; }
    if (!s.empty()) memcpy(dst, s.data(), s.size());
    return (int)ZIPC_HOST_OK;
  });
}

int zipc_host_extract_all(const zipc_host_archive *a, zipc_host_extraction **out, char *err, size_t errcap) {
  if (!a || !out) return ZIPC_HOST_INVALID;
  *out = nullptr;
  return guarded(err, errcap, [&] {
    auto *x = new zipc_host_extraction();
    x->items = a->z.extract_all();
    *out = x;
    return (int)ZIPC_HOST_OK;
  });
}
size_t zipc_host_extraction_count(const zipc_host_extraction *x) { return x ? x->items.size() : 0; }
int zipc_host_extraction_at(const zipc_host_extraction *x, size_t i, const char **path, size_t *path_len, int *ok,
                            const char **data, size_t *len) {
  if (!x || i >= x->items.size()) return ZIPC_HOST_INVALID;
  const auto &it = x->items[i];
  *path = it.first.data();
  *path_len = it.first.size();
  *ok = it.second.ok;
  const std::string &s = it.second.ok ? it.second.value : it.second.error;
  *data = s.data();
  *len = s.size();
  return ZIPC_HOST_OK;
}
void zipc_host_extraction_free(zipc_host_extraction *x) { delete x; }

void zipc_host_ptime_to_date_time(int64_t t, int o[6]) {
  std::tie(o[0], o[1], o[2], o[3], o[4], o[5]) = zipc::Ptime::to_date_time(t);
}
int64_t zipc_host_ptime_of_dos_date_time(int dos_date, int dos_time) {
  return zipc::Ptime::of_dos_date_time(dos_date, dos_time);
}
void zipc_host_ptime_to_dos_date_time(int64_t t, int *dos_date, int *dos_time) {
  const auto r = zipc::Ptime::to_dos_date_time(t);
  *dos_date = r.first;
  *dos_time = r.second;
}
size_t zipc_host_ptime_pp(int64_t t, char *buf, size_t cap) { return put_str(zipc::Ptime::pp(t), buf, cap); }
size_t zipc_host_fpath(int which, const char *p, size_t len, char *out, size_t cap) {
  const std::string s(p, len);
  const std::string r = which == 0 ? zipc::Fpath::ensure_unix(s)
                        : which == 1 ? zipc::Fpath::ensure_directoryness(s) : zipc::Fpath::sanitize(s);
  return put_str(r, out, cap);
}
size_t zipc_host_fpath_pp_mode(int mode, char *buf, size_t cap) { return put_str(zipc::Fpath::pp_mode(mode), buf, cap); }

int zipc_host_set_devices(const int *devices, size_t n) {
  try {
    zipc_deflate::set_devices(std::vector<int>(devices, devices + (devices ? n : 0)));
    return ZIPC_HOST_OK;
  } catch (...) {
    return ZIPC_HOST_FAILURE;
  }
}
size_t zipc_host_devices(int *devices, size_t cap) {
  const std::vector<int> d = zipc_deflate::devices();
  for (size_t i = 0; i < d.size() && i < cap && devices; i++) devices[i] = d[i];
  return d.size();
}
void zipc_host_set_thread_device(int device) { zipc_deflate::set_thread_device(device); }
void zipc_host_partition(const size_t *sizes, size_t n, size_t n_devices, size_t *bounds) {
  std::vector<zipc_deflate::ManyItem> items(n);
  for (size_t i = 0; i < n; i++) items[i].len = sizes[i];
  const auto parts = zipc_deflate::partition_items(items, n_devices ? n_devices : 1);
  bounds[0] = 0;
  for (size_t k = 0; k < parts.size(); k++) bounds[k + 1] = parts[k].second;
}

}  // extern "C"
