(* zipc_deflate.ml -- drop-in replacement for the reference's src/zipc_deflate.ml,
   same signature (src/zipc_deflate.mli), every function backed by the MI355X
   library libzipc_hip.so through ocaml-ctypes (no C stubs).

   NOT COMPILED in this repository's build image (no OCaml toolchain there); kept
   tiny on purpose.  See INTEGRATION.md for how a zipc maintainer links it:
   replace src/zipc_deflate.ml by this file and add `ctypes.foreign` to the
   package's dependencies; src/zipc.ml is unchanged (it only uses the signature,
   src/zipc.ml:174,183,210,216,223). *)

open Ctypes
open Foreign

type uint16 = int
type uint32 = int32

let lib = Dl.dlopen ~filename:"libzipc_hip.so" ~flags:[Dl.RTLD_NOW]
let ctx_t = ptr void
let f name ty = foreign ~from:lib name ty

let c_create = f "zipc_hip_create" (ptr ctx_t @-> int @-> returning int)
let c_strerror = f "zipc_hip_strerror" (int @-> returning string)
let c_crc32 = f "zipc_hip_crc32" (ctx_t @-> ocaml_string @-> size_t @-> ptr uint32_t @-> returning int)
let c_adler32 = f "zipc_hip_adler32" (ctx_t @-> ocaml_string @-> size_t @-> ptr uint32_t @-> returning int)
let c_inflate =
  f "zipc_hip_inflate"
    (ctx_t @-> ocaml_string @-> size_t @-> int @-> size_t @-> int @-> ocaml_bytes @-> size_t @->
     ptr size_t @-> ptr uint32_t @-> returning int)
let c_deflate_bound = f "zipc_hip_deflate_bound" (size_t @-> returning size_t)
let c_deflate =
  f "zipc_hip_deflate"
    (ctx_t @-> ocaml_string @-> size_t @-> int @-> int @-> ocaml_bytes @-> size_t @->
     ptr size_t @-> ptr uint32_t @-> returning int)

let c_zlib_decompress =
  f "zipc_hip_zlib_decompress"
    (ctx_t @-> ocaml_string @-> size_t @-> int @-> size_t @-> ocaml_bytes @-> size_t @->
     ptr size_t @-> ptr uint32_t @-> ptr uint32_t @-> ptr uint32_t @-> returning int)
let c_zlib_bound = f "zipc_hip_zlib_bound" (size_t @-> returning size_t)
let c_zlib_compress =
  f "zipc_hip_zlib_compress"
    (ctx_t @-> ocaml_string @-> size_t @-> int @-> ocaml_bytes @-> size_t @->
     ptr size_t @-> ptr uint32_t @-> returning int)

(* status codes and enums of include/zipc_hip.h *)
let ok = 0 and err_zlib_method = 3 and err_checksum = 6 and err_dst_too_small = 16
let crc_nop = 0 and crc_crc32 = 1 and crc_adler32 = 2

(* ONE context for the whole program: a zipc_hip context owns one HIP stream and staging buffers
   that every call reuses, so it serves one thread at a time.  The reference module is re-entrant;
   this shim is not domain-safe as written -- a multi-domain program keeps a context per domain
   (OCaml 5: a Domain.DLS key holding this lazy value), as the C++ host layer and the Python
   mirror of this repository do per thread.  (Also: ?start on the encode side selects exactly
   [start, start + len) here; the reference's Lz77.compress does not when start > 0 -- see
   INTEGRATION.md.  And Adler_32 is the reference's signed-remainder value unless
   zipc_hip_set_adler_rfc1950 is called on the context.) *)
(* The device: ZIPC_HIP_DEVICE from the environment (a process per GPU is how a node's GPUs are used through
   this seam: members are independent, test/zipc_tool.ml:6-8; the C++ host layer of this repository also
   spreads one batch of members over all devices from one process, include/zipc_host.h), else device 0. *)
let device =
  match Sys.getenv_opt "ZIPC_HIP_DEVICE" with
  | None -> 0
  | Some v -> (try int_of_string (String.trim v) with Failure _ -> 0)

let ctx = lazy begin
  let p = allocate ctx_t null in
  let st = c_create p device in
  if st <> ok then failwith ("zipc_hip_create: " ^ c_strerror st);
  !@ p
end

(* ?start ?len -> the selected bytes (a copy only when a proper sub-range is asked);
   out-of-range arguments raise Invalid_argument like the reference's bounds checks *)
let range ?(start = 0) ?len s =
  let len = match len with None -> String.length s - start | Some l -> l in
  if start = 0 && len = String.length s then s else String.sub s start len

let u32 p = Unsigned.UInt32.to_int32 (!@ p)
let sz = Unsigned.Size_t.of_int

let crc_error e f =
  Error (Printf.sprintf "Checksum mismatch, expected %lx found %lx)" e f)

module type CHECKSUM = sig
  type t = uint32
  val equal : t -> t -> bool
  val check : expect:t -> found:t -> (unit, string) result
  val pp : Format.formatter -> t -> unit
  val string : ?start:int -> ?len:int -> string -> t
end

let checksum c_fn : (module CHECKSUM) = (module struct
  type t = uint32
  let equal = Int32.equal
  let pp ppf crc = Format.fprintf ppf "%lx" crc
  let check ~expect:e ~found:f = if equal e f then Ok () else crc_error e f
  let string ?start ?len s =
    let s = range ?start ?len s in
    let out = allocate uint32_t Unsigned.UInt32.zero in
    let st = c_fn (Lazy.force ctx) (ocaml_string_start s) (sz (String.length s)) out in
    if st <> ok then failwith (c_strerror st);
    u32 out
end)

module Crc_32 = (val checksum c_crc32)
module Adler_32 = (val checksum c_adler32)

(* inflate: with ?decompressed_size the destination is exactly that large
   (Buf.make ~fixed:true, zipc_deflate.ml:554); without it start at 3x the input
   and double on ZIPC_HIP_ERR_DST_TOO_SMALL (Buf.grow, zipc_deflate.ml:27-38) *)
let inflate_and_crc ?decompressed_size ?start ?len s ~crc_op =
  let s = range ?start ?len s in
  let n = String.length s in
  let has_limit, limit = match decompressed_size with None -> 0, 0 | Some d -> 1, d in
  let rec go cap =
    let dst = Bytes.create cap in
    let out_len = allocate size_t Unsigned.Size_t.zero in
    let crc = allocate uint32_t Unsigned.UInt32.zero in
    let st =
      c_inflate (Lazy.force ctx) (ocaml_string_start s) (sz n) has_limit (sz limit) crc_op
        (ocaml_bytes_start dst) (sz cap) out_len crc
    in
    if st = err_dst_too_small && has_limit = 0 then go (2 * cap) else
    if st <> ok then Error (c_strerror st) else
    Ok (Bytes.sub_string dst 0 (Unsigned.Size_t.to_int (!@ out_len)), u32 crc)
  in
  go (if has_limit = 1 then limit else max (3 * n) 1024)

let inflate_and_crc_32 ?decompressed_size ?start ?len s =
  inflate_and_crc ?decompressed_size ?start ?len s ~crc_op:crc_crc32

let inflate_and_adler_32 ?decompressed_size ?start ?len s =
  inflate_and_crc ?decompressed_size ?start ?len s ~crc_op:crc_adler32

let inflate ?decompressed_size ?start ?len s =
  Result.map fst (inflate_and_crc ?decompressed_size ?start ?len s ~crc_op:crc_nop)

(* zlib_decompress zipc_deflate.ml:720-740: the library's own entry point (header checks, body, trailer and the
   Adler-32 comparison all behind zipc_hip_zlib_decompress); only the messages are put together here *)
let zlib_decompress ?decompressed_size ?start ?len s =
  let s = range ?start ?len s in
  let n = String.length s in
  let has_limit, limit = match decompressed_size with None -> 0, 0 | Some d -> 1, d in
  let rec go cap =
    let dst = Bytes.create cap in
    let out_len = allocate size_t Unsigned.Size_t.zero in
    let adler = allocate uint32_t Unsigned.UInt32.zero in
    let expect = allocate uint32_t Unsigned.UInt32.zero in
    let found = allocate uint32_t Unsigned.UInt32.zero in
    let st =
      c_zlib_decompress (Lazy.force ctx) (ocaml_string_start s) (sz n) has_limit (sz limit)
        (ocaml_bytes_start dst) (sz cap) out_len adler expect found
    in
    if st = err_dst_too_small && has_limit = 0 then go (2 * cap) else
    if st = ok then Ok (Bytes.sub_string dst 0 (Unsigned.Size_t.to_int (!@ out_len)), u32 adler) else
    if st = err_checksum then begin
      let e = u32 expect and f = u32 found in
      Error (Some (e, f), Printf.sprintf "Checksum mismatch, expected %lx found %lx)" e f)
    end else
    if st = err_zlib_method
    then Error (None, Printf.sprintf "Unknown compression method (%d)" (String.get_uint8 s 0 land 0x0F))
    else Error (None, c_strerror st)
  in
  go (if has_limit = 1 then limit else max (3 * n) 1024)

type level = [ `None | `Fast | `Default | `Best ]

let level_code = function `None -> 0 | `Fast -> 1 | `Default -> 2 | `Best -> 3

(* the reference's effective default is `Best (make_encoder, zipc_deflate.ml:817) *)
let crc_and_deflate ?(level = `Best) ?start ?len s ~crc_op =
  let s = range ?start ?len s in
  let n = String.length s in
  let cap = Unsigned.Size_t.to_int (c_deflate_bound (sz n)) in
  let dst = Bytes.create cap in
  let out_len = allocate size_t Unsigned.Size_t.zero in
  let crc = allocate uint32_t Unsigned.UInt32.zero in
  let st =
    c_deflate (Lazy.force ctx) (ocaml_string_start s) (sz n) (level_code level) crc_op
      (ocaml_bytes_start dst) (sz cap) out_len crc
  in
  if st <> ok then Error (c_strerror st) else
  Ok (u32 crc, Bytes.sub_string dst 0 (Unsigned.Size_t.to_int (!@ out_len)))

let crc_32_and_deflate ?level ?start ?len s = crc_and_deflate ?level ?start ?len s ~crc_op:crc_crc32
let adler_32_and_deflate ?level ?start ?len s = crc_and_deflate ?level ?start ?len s ~crc_op:crc_adler32
let deflate ?level ?start ?len s = Result.map snd (crc_and_deflate ?level ?start ?len s ~crc_op:crc_nop)

(* zlib_compress zipc_deflate.ml:1262-1277: header, body and trailer by zipc_hip_zlib_compress *)
let zlib_compress ?(level = `Best) ?start ?len s =
  let s = range ?start ?len s in
  let n = String.length s in
  let cap = Unsigned.Size_t.to_int (c_zlib_bound (sz n)) in
  let dst = Bytes.create cap in
  let out_len = allocate size_t Unsigned.Size_t.zero in
  let adler = allocate uint32_t Unsigned.UInt32.zero in
  let st =
    c_zlib_compress (Lazy.force ctx) (ocaml_string_start s) (sz n) (level_code level)
      (ocaml_bytes_start dst) (sz cap) out_len adler
  in
  if st <> ok then Error (c_strerror st) else
  Ok (u32 adler, Bytes.sub_string dst 0 (Unsigned.Size_t.to_int (!@ out_len)))
