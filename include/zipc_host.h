/* zipc_host.h -- C interface of the host layer zipc_amd/host (libzipc_host.so):
 * the reference's `Zipc` module (src/zipc.mli; SURVEY.md section 8(f) rows 1-2: member
 * glue and ZIP container) over the MI355X codec of zipc_hip.h.
 *
 * The host layer itself is C++ (zipc_amd/host/zipc.hpp mirrors the OCaml
 * signature: compression, Fpath, Ptime, File, Member, the archive map and its
 * binary codec).  This header is the plain-C view of it that bindings and the
 * tests use: an archive handle, member records, and the operations of
 * src/zipc.mli by name.
 *
 * Return codes: ZIPC_HOST_OK; ZIPC_HOST_ERROR = the reference's `Error msg` (msg
 * copied to err); ZIPC_HOST_INVALID = the reference's Invalid_argument;
 * ZIPC_HOST_FAILURE = the MI355X library failed (no device, HIP error). */
#ifndef ZIPC_HOST_H
#define ZIPC_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ZIPC_HOST_OK = 0, ZIPC_HOST_ERROR = 1, ZIPC_HOST_INVALID = 2, ZIPC_HOST_FAILURE = 3 };

typedef struct zipc_host_archive zipc_host_archive; /* Zipc.t (zipc.mli:287) */

/* Member.t + File.t accessors (zipc.mli:139-183,251-264); path points into the archive */
typedef struct {
  const char *path;
  size_t path_len;
  int is_dir;   /* Member.kind = Dir */
  int mode;     /* Member.mode */
  int64_t mtime; /* Member.mtime, POSIX seconds */
  /* the rest is File.* and is zero for directories */
  int compression; /* the ZIP method number (compression_to_int) */
  int gp_flags, version_made_by, version_needed_to_extract;
  uint64_t start, compressed_size, decompressed_size;
  uint32_t decompressed_crc_32;
  int is_encrypted, can_extract;
} zipc_host_member;

/* optional arguments of Member.make (zipc.mli:232-249) */
typedef struct {
  int has_mtime;
  int64_t mtime;
  int has_mode;
  int mode;
} zipc_host_member_opts;

/* Zipc.empty / of_binary_string (zipc.mli:290,333) */
zipc_host_archive *zipc_host_empty(void);
int zipc_host_of_binary_string(const void *s, size_t len, zipc_host_archive **out, char *err, size_t errcap);
void zipc_host_free(zipc_host_archive *a);
int zipc_host_string_has_magic(const void *s, size_t len); /* zipc.mli:329 */

/* member_count, fold order (increasing path), find, remove (zipc.mli:296-315) */
size_t zipc_host_member_count(const zipc_host_archive *a);
int zipc_host_member_at(const zipc_host_archive *a, size_t index, zipc_host_member *m);
int zipc_host_find(const zipc_host_archive *a, const char *path, size_t path_len, size_t *index);
int zipc_host_remove(zipc_host_archive *a, const char *path, size_t path_len);
/* Member.pp / pp_long (zipc.mli:267-272); returns the length written (without NUL) */
size_t zipc_host_member_pp(const zipc_host_archive *a, size_t index, int long_form, char *buf, size_t cap);

/* Member.make ~path Dir |> add */
int zipc_host_add_dir(zipc_host_archive *a, const char *path, size_t path_len, const zipc_host_member_opts *o,
                      char *err, size_t errcap);
/* File.make (zipc.mli:100-121) |> Member.make |> add: bytes already compressed by the caller.
 * compressed_size < 0: absent (the rest of bytes from start). */
int zipc_host_add_file_made(zipc_host_archive *a, const char *path, size_t path_len, int compression,
                            const void *bytes, size_t bytes_len, size_t start, int64_t compressed_size,
                            int64_t decompressed_size, uint32_t decompressed_crc_32, int gp_flags,
                            int version_made_by, int version_needed_to_extract, const zipc_host_member_opts *o,
                            char *err, size_t errcap);
/* File.stored_of_binary_string / deflate_of_binary_string (zipc.mli:123-137) |> Member.make |> add.
 * level: 0..3 = `None `Fast `Default `Best, -1 = absent (= `Best).  The CRC-32 and
 * the compression run on the GPU. */
int zipc_host_add_file_stored(zipc_host_archive *a, const char *path, size_t path_len, const void *data, size_t len,
                              const zipc_host_member_opts *o, char *err, size_t errcap);
int zipc_host_add_file_deflate(zipc_host_archive *a, const char *path, size_t path_len, const void *data, size_t len,
                               int level, const zipc_host_member_opts *o, char *err, size_t errcap);
/* the same for n files as ONE batch on the GPU (Archive::add_deflated_files) */
int zipc_host_add_files_deflate(zipc_host_archive *a, size_t n, const char *const *paths, const size_t *path_lens,
                                const void *const *datas, const size_t *lens, int level, char *err, size_t errcap);

/* encoding_size / to_binary_string ?first (zipc.mli:353-356).  first = NULL: "mimetype". */
size_t zipc_host_encoding_size(const zipc_host_archive *a);
int zipc_host_to_binary_string(const zipc_host_archive *a, const char *first, size_t first_len, void *dst, size_t cap,
                               size_t *out_len, char *err, size_t errcap);

/* File.to_binary_string / to_binary_string_no_crc_check (zipc.mli:190-208) of member `index`:
 * dst needs decompressed_size bytes. */
int zipc_host_member_to_binary_string(const zipc_host_archive *a, size_t index, int check_crc, void *dst, size_t cap,
                                      size_t *out_len, uint32_t *crc, char *err, size_t errcap);
/* File.to_binary_string of every file member as one batch on the GPU (Archive::extract_all).
 * Results are read back per file member, in path order. */
typedef struct zipc_host_extraction zipc_host_extraction;
int zipc_host_extract_all(const zipc_host_archive *a, zipc_host_extraction **out, char *err, size_t errcap);

/* ---- devices.  The reference handles one member per call on one core; its tool notes that the members of an
 * archive are "trivially parallelizable" (test/zipc_tool.ml:6-8).  zipc_host_add_files_deflate and
 * zipc_host_extract_all hand ALL members to the GPUs as one batch: with several devices the members are cut into
 * contiguous ranges of about equal bytes (the order Zipc writes them, src/zipc.ml:575-581), one range per
 * device, each through a context and a host thread of its own; nothing is exchanged between the devices.
 * The list is every visible device by default, ZIPC_HIP_DEVICES="0,2,3" from the environment, or this call
 * (n = 0: back to the default).  A device may be named twice (two contexts on it). */
int zipc_host_set_devices(const int *devices, size_t n);
size_t zipc_host_devices(int *devices, size_t cap);
/* the device of the calling thread's single-stream calls (the first of the list unless set) */
void zipc_host_set_thread_device(int device);
/* bounds[0 .. n_devices] of the ranges n sizes are cut into (what the two calls above use) */
void zipc_host_partition(const size_t *sizes, size_t n, size_t n_devices, size_t *bounds);
size_t zipc_host_extraction_count(const zipc_host_extraction *x);
/* ok != 0: data/len are the member's bytes; else data/len are the error message */
int zipc_host_extraction_at(const zipc_host_extraction *x, size_t i, const char **path, size_t *path_len, int *ok,
                            const char **data, size_t *len);
void zipc_host_extraction_free(zipc_host_extraction *x);

/* Ptime (zipc.mli:71-87, zipc.ml:93-124) and Fpath (zipc.mli:34-67) */
void zipc_host_ptime_to_date_time(int64_t t, int ymdhms[6]);
int64_t zipc_host_ptime_of_dos_date_time(int dos_date, int dos_time);
void zipc_host_ptime_to_dos_date_time(int64_t t, int *dos_date, int *dos_time);
size_t zipc_host_ptime_pp(int64_t t, char *buf, size_t cap);
/* which: 0 ensure_unix, 1 ensure_directoryness, 2 sanitize; returns the length (out needs len + 2) */
size_t zipc_host_fpath(int which, const char *p, size_t len, char *out, size_t cap);
size_t zipc_host_fpath_pp_mode(int mode, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
