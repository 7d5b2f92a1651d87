// zd_common.h -- shared definitions for the HIP kernels of the Zipc_deflate hot path.
//
// RFC 1951 constants as the reference packs them (src/zipc_deflate.ml:237-313),
// status codes of include/zipc_hip.h, and small helpers.  Functions marked ZD_HD
// are lane-serial logic: they compile as __device__ code under hipcc and as
// plain inline C++ under g++, where tests/host_sim runs them against the oracle
// (test tooling only -- the product always runs them on the GPU).
#pragma once

#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZD_HD __host__ __device__ __forceinline__
#define ZD_DEV __device__ __forceinline__
#define ZD_CONST __device__ __constant__
#else
#define ZD_HD inline
#define ZD_DEV inline
#define ZD_CONST static const
#endif

namespace zd {

// status codes (include/zipc_hip.h)
enum : uint32_t {
  ST_OK = 0,
  ST_CORRUPTED = 1,
  ST_SIZE_EXCEEDED = 2,
  ST_DST_TOO_SMALL = 16,
  ST_HIP = 17,  // a device-side check of the library itself failed (lz_chain: deflate.hip chain_check)
  ST_INVALID_ARG = 18,
};
enum : int { CRC_NOP = 0, CRC_CRC32 = 1, CRC_ADLER32 = 2, CRC_ADLER32_RFC = 3 };  // 3: RFC 1950's Adler-32, not the reference's (Q6/Q7)
enum : int { LEVEL_NONE = 0, LEVEL_FAST = 1, LEVEL_DEFAULT = 2, LEVEL_BEST = 3 };

constexpr int LITLEN_SYM_MAX = 285;    // zd.ml:237
constexpr int LITLEN_EOB = 256;        // zd.ml:240
constexpr int LITLEN_FIRST_LEN = 257;  // zd.ml:241
constexpr int DIST_SYM_MAX = 29;       // zd.ml:271
constexpr int CODELEN_SYM_MAX = 18;    // zd.ml:310
// Longest stream (and largest destination capacity) the kernels take: positions are 32-bit and are
// advanced in tile-sized steps (64 .. 16 Ki positions and a chunk of overshoot), so the limit keeps
// 64 KiB of headroom below 2^32.  Longer streams report ST_INVALID_ARG.  (include/zipc_hip.h:
// ZIPC_HIP_MAX_STREAM_LEN.)
constexpr uint64_t MAX_STREAM_LEN = 0xFFFF0000ull;
constexpr int MAX_BLOCK_SRC_LEN = 65534;  // zd.ml:747-750
constexpr int MIN_MATCH_LEN = 4;       // zd.ml:1141
constexpr int MAX_MATCH_LEN = 258;     // zd.ml:1142
constexpr int MAX_MATCH_DIST = 32768;  // zd.ml:1143
constexpr uint32_t ADLER_BASE = 65521; // zd.ml:172
constexpr uint32_t ADLER_CHUNK = 5552; // zd.ml:180,196
constexpr uint32_t CRC_POLY = 0xedb88320u;  // zd.ml:113

// stream descriptor / result: binary-identical to zipc_hip_stream_desc /
// zipc_hip_stream_result of include/zipc_hip.h
struct StreamDesc {
  uint64_t src_off, src_len, dst_off, dst_cap, limit;
  uint32_t flags, reserved;
};
struct StreamResult {
  uint32_t status, checksum;
  uint64_t out_len;
};
constexpr uint32_t STREAM_HAS_LIMIT = 1u;
// (never a caller's: the library's own copy of a call's descriptors, for streams that went by blocks -- api.hip)
constexpr uint32_t STREAM_DONE = 1u << 31;
constexpr int CRC_OP_MARKED = 0x100;  // in a kernel's crc_op argument: the descriptors are the library's marked copy (inflate.hip)

// (base << 4) | extra_bits, zd.ml:245-255 and zd.ml:277-288
#define ZD_V(bits, len) (uint16_t)(((len) << 4) | (bits))
ZD_CONST uint16_t k_length_value_of_sym[29] = {
    ZD_V(0, 3),   ZD_V(0, 4),   ZD_V(0, 5),   ZD_V(0, 6),   ZD_V(0, 7),   ZD_V(0, 8),
    ZD_V(0, 9),   ZD_V(0, 10),  ZD_V(1, 11),  ZD_V(1, 13),  ZD_V(1, 15),  ZD_V(1, 17),
    ZD_V(2, 19),  ZD_V(2, 23),  ZD_V(2, 27),  ZD_V(2, 31),  ZD_V(3, 35),  ZD_V(3, 43),
    ZD_V(3, 51),  ZD_V(3, 59),  ZD_V(4, 67),  ZD_V(4, 83),  ZD_V(4, 99),  ZD_V(4, 115),
    ZD_V(5, 131), ZD_V(5, 163), ZD_V(5, 195), ZD_V(5, 227), ZD_V(0, 258)};
// dist bases go up to 24577: (base << 4) needs 19 bits
#define ZD_VD(bits, len) (uint32_t)(((len) << 4) | (bits))
ZD_CONST uint32_t k_dist_value_of_sym[30] = {
    ZD_VD(0, 1),     ZD_VD(0, 2),     ZD_VD(0, 3),      ZD_VD(0, 4),      ZD_VD(1, 5),
    ZD_VD(1, 7),     ZD_VD(2, 9),     ZD_VD(2, 13),     ZD_VD(3, 17),     ZD_VD(3, 25),
    ZD_VD(4, 33),    ZD_VD(4, 49),    ZD_VD(5, 65),     ZD_VD(5, 97),     ZD_VD(6, 129),
    ZD_VD(6, 193),   ZD_VD(7, 257),   ZD_VD(7, 385),    ZD_VD(8, 513),    ZD_VD(8, 769),
    ZD_VD(9, 1025),  ZD_VD(9, 1537),  ZD_VD(10, 2049),  ZD_VD(10, 3073),  ZD_VD(11, 4097),
    ZD_VD(11, 6145), ZD_VD(12, 8193), ZD_VD(12, 12289), ZD_VD(13, 16385), ZD_VD(13, 24577)};
// zd.ml:312-313
ZD_CONST uint8_t k_codelen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The two tables above in closed form (no memory access: a table read in the
// middle of a symbol loop is a global load, and its s_waitcnt also drains every
// store the wave has in flight).  sym: litlen symbol 257..285 / dist symbol 0..29.
ZD_HD void length_sym_value(int sym, uint32_t &base, uint32_t &extra) {
  const uint32_t idx = (uint32_t)(sym - LITLEN_FIRST_LEN);
  if (idx < 4) { base = 3 + idx; extra = 0; return; }
  if (idx == 28) { base = 258; extra = 0; return; }
  extra = (idx - 4) >> 2;
  base = 3 + ((4 + (idx & 3)) << extra);
}
ZD_HD void dist_sym_value(int sym, uint32_t &base, uint32_t &extra) {
  if (sym < 4) { base = 1 + (uint32_t)sym; extra = 0; return; }
  extra = ((uint32_t)sym >> 1) - 1;
  base = 1 + ((2 + ((uint32_t)sym & 1)) << extra);
}

// length value -> litlen symbol (zd.ml:260-267; later rows overwrite, so 258 -> 285)
ZD_HD int length_to_sym(int len) {
  if (len == 258) return 285;
  if (len <= 10) return 254 + len;
  // rows with e extra bits (e = 1..5) start at base(e) = 3 + (1 << (e + 2)),
  // four symbols each, first symbol 261 + 4e
  int x = len - 3;                       // >= 8
  int e = 29 - __builtin_clz((unsigned)x);  // floor(log2 x) - 2
  return 261 + 4 * e + ((x - (1 << (e + 2))) >> e);
}

// dist value -> dist symbol (zd.ml:290-302)
ZD_HD int dist_to_sym(int dist) {
  if (dist <= 4) return dist - 1;
  int x = dist - 1;                         // >= 4
  int e = 30 - __builtin_clz((unsigned)x);  // floor(log2 x) - 1  (extra bits)
  return 2 * e + 2 + ((x >> e) & 1);
}

ZD_HD uint32_t load_u32_le(const uint8_t *p) {
  typedef uint32_t __attribute__((aligned(1), may_alias)) u32u;
  return *(const u32u *)p;
}
ZD_HD void store_u16_le(uint8_t *p, uint16_t v) {
  typedef uint16_t __attribute__((aligned(1), may_alias)) u16u;
  *(u16u *)p = v;
}
ZD_HD uint64_t load_u64_le(const uint8_t *p) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  return *(const u64u *)p;
}
ZD_HD void store_u64_le(uint8_t *p, uint64_t v) {
  typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
  *(u64u *)p = v;
}

// bits [sh, sh + 32) of hi:lo, sh taken modulo 32 (v_alignbit_b32)
ZD_HD uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
#endif
}
// `width` bits of v from bit `off` on (v_bfe_u32: off and width are the operands' low 5 bits)
ZD_HD uint32_t bit_field(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ubfe(v, off, width);
#else
  return (v >> (off & 31u)) & ((1u << (width & 31u)) - 1u);  // operands taken modulo 32 like the instruction
#endif
}

// 8 bytes at s + pos read as three ALIGNED words (s 4-byte aligned; touches up
// to 11 bytes past pos & ~3): the form for LDS, where a misaligned 8-byte read
// stalls the pipe (SQ_LDS_UNALIGNED_STALL)
ZD_HD uint64_t load_u64_words(const uint8_t *s, uint32_t pos) {
  const uint32_t *w = (const uint32_t *)(s + (pos & ~3u));
  const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], sh = (pos & 3u) * 8u;
  return ((uint64_t)funnel32(d2, d1, sh) << 32) | funnel32(d1, d0, sh);
}

// low `len` bits of v, reversed
ZD_HD uint32_t bitrev(uint32_t v, int len) {
#if defined(__clang__)
  return __builtin_bitreverse32(v) >> (32 - len);
#else
  uint32_t r = 0;
  for (int i = 0; i < len; i++) r |= ((v >> i) & 1u) << (len - 1 - i);
  return r;
#endif
}

// Lz77.hash4 zd.ml:1145-1148
ZD_HD uint32_t hash4(uint32_t le32) { return (le32 * 0x9E3779B1u) >> 17; }

// ---------------------------------------------------------------------------
// CRC-32 in GF(2): the reference's table CRC (zd.ml:113-156) is the reflected
// CRC-32 with polynomial 0xedb88320; splitting a buffer over lanes needs the
// standard combination rule crc(A||B) = crc(A) * x^(8|B|) + crc(B) (mod P) on
// raw (init 0, no final xor) values.  Bit 31 is x^0 in this representation.

// a(x) * b(x) mod P
ZD_HD uint32_t gf2_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll 4
  for (int i = 0; i < 32; i++) {
    p ^= b & (0u - ((a >> 31) & 1u));
    a <<= 1;
    b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
  }
  return p;
}

// x^(8 * nbytes) mod P, by square and multiply on the exponent bits
ZD_HD uint32_t gf2_xpow8n(uint64_t nbytes) {
  uint32_t r = 0x80000000u;    // x^0
  uint32_t sq = 0x00800000u;   // x^8
  while (nbytes) {
    if (nbytes & 1) r = gf2_mul(r, sq);
    sq = gf2_mul(sq, sq);
    nbytes >>= 1;
  }
  return r;
}

// a(x) * M(x) mod P for a CONSTANT M from nibble tables: multiplication by M is linear
// over GF(2), so a * M is the XOR over a's 8 nibbles of tab[j][nibble j], where
// tab[j][v] = (v << 4j) * M.  8 lookups in 16-entry rows (16 consecutive words: no LDS
// bank conflicts) instead of gf2_mul's 32 dependent bit steps.
constexpr int GF2_NIB_WORDS = 8 * 16;  // words of one constant's table
ZD_HD void gf2_nib_table(uint32_t m, uint32_t *tab) {
  for (int j = 0; j < 8; j++)
    for (uint32_t v = 0; v < 16; v++) tab[j * 16 + v] = gf2_mul(v << (4 * j), m);
}
ZD_HD uint32_t gf2_mul_nib(uint32_t a, const uint32_t *tab) {
  uint32_t r = tab[a & 15u];
#pragma unroll
  for (int j = 1; j < 8; j++) r ^= tab[j * 16 + ((a >> (4 * j)) & 15u)];
  return r;
}

// state after feeding nbytes with raw CRC `raw` to running state `state`
// (state is the reference's un-finished value: init 0xFFFFFFFF, zd.ml:135)
ZD_HD uint32_t crc_state_advance(uint32_t state, uint32_t raw, uint32_t xpow) {
  return gf2_mul(state, xpow) ^ raw;
}

// ---------------------------------------------------------------------------
// Adler-32 chunk chain (zd.ml:175-198).  A chunk of n <= 5552 bytes is
// summarised by S1 = sum b_i and S2 = sum (n - i) b_i; the running (s1, s2)
// then advance with the reference's wrapping int32 arithmetic and its SIGNED
// remainder (Int32.rem, zd.ml:95,196).
// rfc: RFC 1950's arithmetic instead (unsigned remainder; with s1, s2 < 65521 and n <= 5552 the
// sums stay below 2^32, which is what zlib's NMAX = 5552 is chosen for) -- the value every other
// zlib implementation computes, whatever the chunking.
ZD_HD void adler_chunk_step(uint32_t &s1, uint32_t &s2, uint32_t n, uint32_t S1, uint32_t S2, bool rfc = false) {
  uint32_t t2 = s2 + n * s1 + S2;  // wraps like int32
  uint32_t t1 = s1 + S1;
  if (rfc) {
    s1 = t1 % ADLER_BASE;
    s2 = t2 % ADLER_BASE;
    return;
  }
  s1 = (uint32_t)((int32_t)t1 % (int32_t)ADLER_BASE);
  s2 = (uint32_t)((int32_t)t2 % (int32_t)ADLER_BASE);
}
// unpack / pack of the running value across calls (zd.ml:178,198)
ZD_HD void adler_unpack(uint32_t a, uint32_t &s1, uint32_t &s2) { s1 = a & 0xFFFFu; s2 = a >> 16; }
ZD_HD uint32_t adler_pack(uint32_t s1, uint32_t s2) { return (s2 << 16) + s1; }

}  // namespace zd
