/* zd_oracle.c -- CPU oracle for the Zipc_deflate hot path.  TEST INFRASTRUCTURE
 * ONLY (see zd_oracle.h).  A plain-C restatement of the reference's
 * src/zipc_deflate.ml ("zd.ml" below); every function cites the lines it follows
 * and keeps the reference's order of side effects, so that error precedence and
 * the compressed bytes come out the same.
 *
 * OCaml `int` is 63-bit: int64_t here.  `Uint32.t` is int32 with wrapping
 * arithmetic: uint32_t here, with the one signed operation (Int32.rem, zd.ml:95)
 * done on an int32_t cast.
 */
#include "zd_oracle.h"

#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* errors: the reference raises Failure and catches it at the API boundary    */
/* (zd.ml:709,740,1250); longjmp plays that role here.                        */

typedef struct { jmp_buf jb; } zd_exn;
#define ZD_RAISE(x, code) longjmp((x)->jb, (code))

const char *zd_strerror(int status) {
  switch (status) {
  case ZD_OK: return "";
  case ZD_ERR_CORRUPTED: return "Corrupted data stream";
  case ZD_ERR_SIZE_EXCEEDED: return "Expected decompression size exceeded";
  case ZD_ERR_ZLIB_METHOD: return "Unknown compression method (%d)";
  case ZD_ERR_ZLIB_WINDOW: return "Window size too large";
  case ZD_ERR_ZLIB_DICT: return "Preset dictionary unsupported";
  case ZD_ERR_CHECKSUM: return "Checksum mismatch, expected %lx found %lx)";
  case ZD_ERR_NOMEM: return "out of memory";
  default: return "unknown status";
  }
}

void zd_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* Buf  zd.ml:16-76                                                          */

typedef struct {
  uint8_t *b;
  size_t cap;
  size_t len;
  int fixed;    /* zd.ml:17 */
  int external; /* caller-provided storage: cannot grow */
  zd_exn *x;
} zd_buf;

/* Buf.make zd.ml:18-20 */
static void buf_make(zd_buf *buf, zd_exn *x, int fixed, size_t sz) {
  size_t cap = (sz == 0 && !fixed) ? 1024 : sz;
  buf->b = (uint8_t *)malloc(cap ? cap : 1);
  buf->cap = cap;
  buf->len = 0;
  buf->fixed = fixed;
  buf->external = 0;
  buf->x = x;
  if (!buf->b) ZD_RAISE(x, ZD_ERR_NOMEM);
}

/* Buf.grow zd.ml:27-38 ("OCaml string size exceeded" cannot happen on 64-bit) */
static void buf_grow(zd_buf *buf, size_t ensure) {
  if (buf->fixed) ZD_RAISE(buf->x, ZD_ERR_SIZE_EXCEEDED);
  if (buf->external) ZD_RAISE(buf->x, ZD_ERR_NOMEM);
  size_t newlen = buf->cap;
  while (newlen < ensure) newlen = 2 * newlen;
  uint8_t *nb = (uint8_t *)realloc(buf->b, newlen);
  if (!nb) ZD_RAISE(buf->x, ZD_ERR_NOMEM);
  buf->b = nb;
  buf->cap = newlen;
}

/* Buf.add_uint8 zd.ml:43-46 */
static inline void buf_add_uint8(zd_buf *buf, int64_t byte) {
  size_t len1 = buf->len + 1;
  if (len1 > buf->cap) buf_grow(buf, len1);
  buf->b[buf->len] = (uint8_t)byte;
  buf->len = len1;
}

/* Buf.add_uint16_le zd.ml:48-51 */
static inline void buf_add_uint16_le(zd_buf *buf, int64_t u16) {
  size_t len1 = buf->len + 2;
  if (len1 > buf->cap) buf_grow(buf, len1);
  buf->b[buf->len] = (uint8_t)(u16 & 0xFF);
  buf->b[buf->len + 1] = (uint8_t)((u16 >> 8) & 0xFF);
  buf->len = len1;
}

/* Buf.add_uint32_be zd.ml:53-56 */
static inline void buf_add_uint32_be(zd_buf *buf, uint32_t u32) {
  size_t len1 = buf->len + 4;
  if (len1 > buf->cap) buf_grow(buf, len1);
  buf->b[buf->len] = (uint8_t)(u32 >> 24);
  buf->b[buf->len + 1] = (uint8_t)(u32 >> 16);
  buf->b[buf->len + 2] = (uint8_t)(u32 >> 8);
  buf->b[buf->len + 3] = (uint8_t)u32;
  buf->len = len1;
}

/* Buf.add_string zd.ml:58-61 */
static void buf_add_string(zd_buf *buf, const uint8_t *s, size_t start, size_t len) {
  size_t len1 = buf->len + len;
  if (len1 > buf->cap) buf_grow(buf, len1);
  memcpy(buf->b + buf->len, s + start, len);
  buf->len = len1;
}

/* Buf.recopy zd.ml:63-75 */
static void buf_recopy(zd_buf *buf, size_t start, size_t len) {
  size_t len1 = buf->len + len;
  if (len1 > buf->cap) buf_grow(buf, len1);
  if (start + len <= buf->len) {
    memcpy(buf->b + buf->len, buf->b + start, len);
    buf->len = len1;
  } else { /* overlapping, work bytewise */
    uint8_t *b = buf->b;
    size_t dst_start = buf->len;
    for (size_t i = 0; i < len; i++) b[dst_start + i] = b[start + i];
    buf->len = len1;
  }
}

/* ------------------------------------------------------------------------ */
/* Crc_32  zd.ml:106-164                                                     */

static uint32_t crc_table[4][256];
static int crc_table_ready = 0;

/* Crc_32.table zd.ml:114-133 (slice-by-4; rows 1-3 derived from row 0) */
static void crc_table_init(void) {
  if (crc_table_ready) return;
  const uint32_t poly = 0xedb88320u; /* zd.ml:113 */
  for (int i = 0; i <= 0xFF; i++) {
    uint32_t c = (uint32_t)i;
    for (int k = 0; k <= 7; k++) c = (c & 1u) ? (poly ^ (c >> 1)) : (c >> 1);
    crc_table[0][i] = c;
  }
  for (int i = 0; i <= 0xFF; i++) {
    for (int k = 0; k < 3; k++) {
      uint32_t v = crc_table[k][i];
      crc_table[k + 1][i] = (v >> 8) ^ crc_table[0][v & 0xFF];
    }
  }
  crc_table_ready = 1;
}

/* Crc_32.string_update zd.ml:137-156 */
uint32_t zd_crc32_update(uint32_t c, const uint8_t *s, size_t len) {
  crc_table_init();
  size_t i = 0;
  /* while i <= (start+len-1)-3: one little-endian word per iteration */
  while (i + 4 <= len) {
    uint32_t u = (uint32_t)s[i] | ((uint32_t)s[i + 1] << 8) |
                 ((uint32_t)s[i + 2] << 16) | ((uint32_t)s[i + 3] << 24);
    u = c ^ u;
    c = crc_table[3][u & 0xFF] ^ crc_table[2][(u >> 8) & 0xFF] ^
        crc_table[1][(u >> 16) & 0xFF] ^ crc_table[0][u >> 24];
    i += 4;
  }
  for (; i < len; i++) { /* byte tail zd.ml:151-155 */
    uint32_t k = (c ^ (uint32_t)s[i]) & 0xFF;
    c = (c >> 8) ^ crc_table[0][k];
  }
  return c;
}

/* Crc_32.string zd.ml:161-163 with init/finish zd.ml:135-136 */
uint32_t zd_crc32(const uint8_t *s, size_t len) {
  return zd_crc32_update(0xFFFFFFFFu, s, len) ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------ */
/* Adler_32  zd.ml:166-206                                                   */

/* Uint32.Syntax.(mod) = Int32.rem: SIGNED, truncating (zd.ml:95).  C99 `%` on
 * int32_t truncates toward zero, like Int32.rem. */
static inline uint32_t u32_rem_signed(uint32_t v, int32_t base) {
  return (uint32_t)((int32_t)v % base);
}

/* Adler_32.string_update zd.ml:175-198 */
uint32_t zd_adler32_update(uint32_t a, const uint8_t *s, size_t len) {
  const int32_t base = 65521; /* zd.ml:172 */
  uint32_t s1 = a & 0xFFFFu, s2 = a >> 16; /* zd.ml:178 (lsr is logical) */
  size_t start = 0;
  size_t block_len = len % 5552; /* zd.ml:180: FIRST chunk is len mod 5552 */
  /* while !start <= max, max = start+len-1  (empty range: no iteration) */
  while (start < len) {
    size_t i = start, block_end = start + block_len; /* block_max+1 */
    while (i + 8 <= block_end) { /* zd.ml:183-193 */
      s1 += s[i];     s2 += s1;
      s1 += s[i + 1]; s2 += s1;
      s1 += s[i + 2]; s2 += s1;
      s1 += s[i + 3]; s2 += s1;
      s1 += s[i + 4]; s2 += s1;
      s1 += s[i + 5]; s2 += s1;
      s1 += s[i + 6]; s2 += s1;
      s1 += s[i + 7]; s2 += s1;
      i += 8;
    }
    while (i < block_end) { s1 += s[i]; s2 += s1; i++; } /* zd.ml:194-195 */
    s1 = u32_rem_signed(s1, base); /* zd.ml:196 */
    s2 = u32_rem_signed(s2, base);
    start = i;
    block_len = 5552;
  }
  return (s2 << 16) + s1; /* zd.ml:198 */
}

/* Adler_32.string zd.ml:203-205 (init = 1, finish = id) */
uint32_t zd_adler32(const uint8_t *s, size_t len) {
  return zd_adler32_update(1u, s, len);
}

/* crc_op_init / crc_op_finish zd.ml:212-216 */
static uint32_t crc_op_init(int op) {
  return op == ZD_CRC_ADLER32 ? 1u : op == ZD_CRC_CRC32 ? 0xFFFFFFFFu : 0u;
}
static uint32_t crc_op_finish(int op, uint32_t crc) {
  return op == ZD_CRC_ADLER32 ? crc : op == ZD_CRC_CRC32 ? (crc ^ 0xFFFFFFFFu) : 0u;
}
static uint32_t crc_op_update(int op, uint32_t crc, const uint8_t *s, size_t len) {
  switch (op) {
  case ZD_CRC_ADLER32: return zd_adler32_update(crc, s, len);
  case ZD_CRC_CRC32: return zd_crc32_update(crc, s, len);
  default: return crc;
  }
}

/* ------------------------------------------------------------------------ */
/* Deflate format data  zd.ml:235-313                                        */

enum {
  LITLEN_SYM_MAX = 285,       /* zd.ml:237 */
  MAX_LITLEN_SYM_COUNT = 286, /* zd.ml:238 */
  LITLEN_SYM_FIXED_MAX = 287, /* zd.ml:239 */
  LITLEN_EOB = 256,           /* zd.ml:240 */
  LITLEN_FIRST_LEN_SYM = 257, /* zd.ml:241 */
  LENGTH_VALUE_MAX = 258,     /* zd.ml:242 */
  DIST_SYM_MAX = 29,          /* zd.ml:271 */
  MAX_DIST_SYM_COUNT = 30,    /* zd.ml:272 */
  DIST_SYM_FIXED_MAX = 31,    /* zd.ml:273 */
  CODELEN_SYM_MAX = 18,       /* zd.ml:310 */
  MAX_CODELEN_SYM_COUNT = 19, /* zd.ml:311 */
  HUFF_MAX_SYMBOL_COUNT = 288, /* zd.ml:319 */
  HUFF_MAX_CODE_BIT_LENGTH = 15 /* zd.ml:320 */
};

#define V(bits, len) (((len) << 4) | (bits))
/* length_value_of_sym_table zd.ml:245-255 */
static const int length_value_of_sym_table[29] = {
    V(0, 3),   V(0, 4),   V(0, 5),   V(0, 6),   V(0, 7),   V(0, 8),   V(0, 9),
    V(0, 10),  V(1, 11),  V(1, 13),  V(1, 15),  V(1, 17),  V(2, 19),  V(2, 23),
    V(2, 27),  V(2, 31),  V(3, 35),  V(3, 43),  V(3, 51),  V(3, 59),  V(4, 67),
    V(4, 83),  V(4, 99),  V(4, 115), V(5, 131), V(5, 163), V(5, 195), V(5, 227),
    V(0, 258)};
/* dist_value_of_sym zd.ml:277-288 */
static const int dist_value_of_sym[30] = {
    V(0, 1),      V(0, 2),      V(0, 3),     V(0, 4),     V(1, 5),
    V(1, 7),      V(2, 9),      V(2, 13),    V(3, 17),    V(3, 25),
    V(4, 33),     V(4, 49),     V(5, 65),    V(5, 97),    V(6, 129),
    V(6, 193),    V(7, 257),    V(7, 385),   V(8, 513),   V(8, 769),
    V(9, 1025),   V(9, 1537),   V(10, 2049), V(10, 3073), V(11, 4097),
    V(11, 6145),  V(12, 8193),  V(12, 12289), V(13, 16385), V(13, 24577)};
#undef V
#define VALUE_BASE(v) ((v) >> 4)        /* zd.ml:243,275 */
#define VALUE_EXTRA_BITS(v) ((v) & 0xF) /* zd.ml:244,276 */

/* codelen_order_of_sym_lengths zd.ml:312-313 */
static const int codelen_order_of_sym_lengths[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                                     11, 4,  12, 3, 13, 2, 14, 1, 15};

static int length_value_to_sym[LENGTH_VALUE_MAX + 1]; /* zd.ml:260-267 */
static int dist_value_to_sym_table[512];              /* zd.ml:290-299 */
static int format_tables_ready = 0;

static void format_tables_init(void) {
  if (format_tables_ready) return;
  /* iter order is important, higher indexes overwrite lower ones (zd.ml:266):
   * length 258 ends up as symbol 285, not 284. */
  for (int i = 0; i < 29; i++) {
    int v = length_value_of_sym_table[i];
    int base = VALUE_BASE(v), extra = VALUE_EXTRA_BITS(v);
    for (int len = base; len <= base + (1 << extra) - 1; len++)
      if (len <= LENGTH_VALUE_MAX) length_value_to_sym[len] = 257 + i;
  }
  for (int i = 0; i < 30; i++) {
    int v = dist_value_of_sym[i];
    int base = VALUE_BASE(v), extra = VALUE_EXTRA_BITS(v);
    for (int dist = base; dist <= base + (1 << extra) - 1; dist++) {
      int k = dist <= 256 ? dist - 1 : 256 + ((dist - 1) >> 7);
      dist_value_to_sym_table[k] = i;
    }
  }
  format_tables_ready = 1;
}

/* dist_value_to_sym zd.ml:301-302 */
static inline int dist_value_to_sym(int dist) {
  return dist_value_to_sym_table[dist <= 256 ? dist - 1 : 256 + ((dist - 1) >> 7)];
}

/* ------------------------------------------------------------------------ */
/* Huffman decoding  zd.ml:324-391                                           */

typedef struct {
  int counts[HUFF_MAX_CODE_BIT_LENGTH + 1]; /* counts[i]: codes of length i */
  int symbols[HUFF_MAX_SYMBOL_COUNT];       /* symbols sorted by code */
  int max_sym;
} huff_decoder;

/* fixed_litlen_decoder zd.ml:334-342 */
static void fixed_litlen_decoder(huff_decoder *t) {
  memset(t, 0, sizeof *t);
  t->counts[7] = 24; t->counts[8] = 152; t->counts[9] = 112;
  for (int i = 0; i <= 23; i++) t->symbols[i] = 256 + i;
  for (int i = 24; i <= 167; i++) t->symbols[i] = i - 24;
  for (int i = 168; i <= 175; i++) t->symbols[i] = 112 + i;
  for (int i = 176; i <= 287; i++) t->symbols[i] = i - 32;
  t->max_sym = LITLEN_SYM_MAX; /* 286 and 287 are unused */
}

/* fixed_dist_decoder zd.ml:344-349 */
static void fixed_dist_decoder(huff_decoder *t) {
  memset(t, 0, sizeof *t);
  t->counts[5] = 32;
  for (int i = 0; i <= 31; i++) t->symbols[i] = i;
  t->max_sym = DIST_SYM_MAX; /* 30 and 31 are unused */
}

/* Huffman.init_decoder zd.ml:355-391 */
static void init_decoder(zd_exn *x, huff_decoder *t, const int *lengths, int start,
                         int lengths_len) {
  int offs[16];
  int *counts = t->counts;
  memset(counts, 0, sizeof t->counts);
  t->max_sym = -1;
  for (int i = 0; i < lengths_len; i++) {
    int len = lengths[start + i];
    if (len != 0) { t->max_sym = i; counts[len]++; }
  }
  int64_t available = 1;
  int num_codes = 0;
  for (int i = 0; i <= 15; i++) {
    int used = counts[i];
    if (used > available) ZD_RAISE(x, ZD_ERR_CORRUPTED); /* over-subscribed */
    available = 2 * (available - used);
    offs[i] = num_codes;
    num_codes += used;
  }
  /* all codes used, or if only one that its length is one (zd.ml:377-378) */
  if ((num_codes > 1 && available > 0) || (num_codes == 1 && counts[1] != 1))
    ZD_RAISE(x, ZD_ERR_CORRUPTED);
  for (int i = 0; i < lengths_len; i++) {
    int leni = lengths[start + i];
    if (leni != 0) { t->symbols[offs[leni]] = i; offs[leni]++; }
  }
  /* single code: add a code 1 that yields a too-large symbol (zd.ml:389-390) */
  if (num_codes == 1) { counts[1] = 2; t->symbols[1] = t->max_sym + 1; }
}

/* ------------------------------------------------------------------------ */
/* Huffman encoding  zd.ml:393-528                                           */

#define SYM_INFO(code, len) (((code) << 5) | (len)) /* zd.ml:396 */
#define SYM_CODE(v) ((v) >> 5)                       /* zd.ml:397 */
#define SYM_CODE_LENGTH(v) ((v) & 0x1F)              /* zd.ml:398 */

typedef struct { int e[HUFF_MAX_SYMBOL_COUNT]; } huff_encoder;

#define NODE_FREQ(v) ((v) >> 10)   /* zd.ml:418 */
#define NODE_LINK(v) ((v) & 0x3FF) /* zd.ml:419 */
#define NODE(freq, link) (((int64_t)(freq) << 10) | (int64_t)(link)) /* zd.ml:420 */

/* heapdown zd.ml:408-417 (head at 1, one-based) */
static void heapdown(int64_t *h, int max, int i) {
  for (;;) {
    int l = 2 * i, r = l + 1;
    if (l > max) return;
    int k = (r > max) ? l : (h[l] < h[r] ? l : r);
    if (h[i] > h[k]) {
      int64_t v = h[i]; h[i] = h[k]; h[k] = v;
      i = k;
    } else return;
  }
}

/* test instrumentation: how often the flatten-and-retry branch (zd.ml:470-473) ran, for codes
   limited to 7 bits (the code-length code) [0] and to 15 bits (litlen, dist) [1] */
int zd_huffman_retries[2];

/* Huffman.lengths_of_freqs zd.ml:404-473.  Writes plain lengths into e. */
static void lengths_of_freqs(int64_t *heap, int *e, const int64_t *freqs, int max_sym,
                             int max_code_len) {
  int64_t freq_cap = 65535; /* zd.ml:405 */
  for (;;) {
    /* create_and_sort_nodes zd.ml:421-431 */
    int max = 0;
    for (int sym = 0; sym <= max_sym; sym++) {
      int64_t freq = freqs[sym];
      if (freq == 0) continue;
      if (freq > freq_cap) freq = freq_cap;
      max++;
      heap[max] = NODE(freq, max_sym + 1 + max);
    }
    for (int i = max / 2; i >= 1; i--) heapdown(heap, max, i);
    if (max < 2) { /* trivial_codeword_lengths zd.ml:462-466 */
      for (int sym = 0; sym <= max_sym; sym++) e[sym] = freqs[sym] == 0 ? 0 : 1;
      return;
    }
    /* make_huffman_tree zd.ml:432-445 */
    for (int m = max; m > 1; m--) {
      int new_max = m - 1;
      int64_t p = heap[1]; /* node with least frequency */
      heap[1] = heap[m];
      heapdown(heap, new_max, 1);
      int64_t q = heap[1]; /* next lowest frequency node */
      int nlink = m;       /* slot m is unused now: it names the new node */
      int64_t freq = NODE_FREQ(p) + NODE_FREQ(q);
      heap[1] = NODE(freq, nlink);
      heap[NODE_LINK(p)] = nlink;
      heap[NODE_LINK(q)] = nlink;
      heapdown(heap, new_max, 1);
    }
    /* code_lengths_of_tree zd.ml:446-461 */
    int overflow = 0, rank = 0;
    for (int sym = 0; sym <= max_sym; sym++) {
      if (freqs[sym] == 0) { e[sym] = 0; continue; }
      rank++;
      int64_t p = heap[max_sym + 1 + rank];
      int len = 1;
      while (p != 2) { len++; p = heap[p]; } /* root has link 2 */
      if (len > max_code_len) { overflow = 1; break; }
      e[sym] = len;
    }
    if (!overflow) return;
    zd_huffman_retries[max_code_len > 7]++;
    freq_cap = freq_cap / 2; /* flatten distribution and retry zd.ml:470-473 */
  }
}

void zd_huffman_lengths_of_freqs(const int64_t *freqs, int max_sym, int max_code_len,
                                 int *lengths) {
  int64_t heap[HUFF_MAX_SYMBOL_COUNT * 2 + 1];
  memset(heap, 0, sizeof heap);
  lengths_of_freqs(heap, lengths, freqs, max_sym, max_code_len);
}

/* reverse_16 zd.ml:481-487 */
static inline int reverse_16(int b) {
  b = ((b & 0xFF00) >> 8) | ((b & 0x00FF) << 8);
  b = ((b & 0xF0F0) >> 4) | ((b & 0x0F0F) << 4);
  b = ((b & 0xCCCC) >> 2) | ((b & 0x3333) << 2);
  b = ((b & 0xAAAA) >> 1) | ((b & 0x5555) << 1);
  return b;
}

/* Huffman.init_with_lengths zd.ml:477-506 */
static void init_with_lengths(int *e, int max_sym) {
  int count[HUFF_MAX_CODE_BIT_LENGTH + 1], code[HUFF_MAX_CODE_BIT_LENGTH + 1];
  memset(count, 0, sizeof count);
  memset(code, 0, sizeof code);
  for (int sym = 0; sym <= max_sym; sym++) count[SYM_CODE_LENGTH(e[sym])]++;
  count[0] = 0;
  code[0] = 0;
  for (int len = 1; len <= HUFF_MAX_CODE_BIT_LENGTH; len++)
    code[len] = (code[len - 1] + count[len - 1]) << 1;
  for (int sym = 0; sym <= max_sym; sym++) {
    int len = SYM_CODE_LENGTH(e[sym]);
    if (len != 0) {
      int c = code[len];
      int bits = reverse_16(c) >> (16 - len);
      e[sym] = SYM_INFO(bits, len);
      code[len] = c + 1;
    }
  }
}

static huff_encoder fixed_litlen_enc, fixed_dist_enc;
static int fixed_encoders_ready = 0;

/* fixed_litlen_encoder / fixed_dist_encoder zd.ml:514-527 */
static void fixed_encoders_init(void) {
  if (fixed_encoders_ready) return;
  int *e = fixed_litlen_enc.e;
  for (int i = 0; i <= 143; i++) e[i] = 8;
  for (int i = 144; i <= 255; i++) e[i] = 9;
  for (int i = 256; i <= 279; i++) e[i] = 7;
  for (int i = 280; i <= 287; i++) e[i] = 8;
  init_with_lengths(e, LITLEN_SYM_FIXED_MAX);
  e = fixed_dist_enc.e;
  memset(e, 0, sizeof fixed_dist_enc.e);
  for (int i = 0; i <= 31; i++) e[i] = 5;
  init_with_lengths(e, DIST_SYM_FIXED_MAX);
  fixed_encoders_ready = 1;
}

/* ------------------------------------------------------------------------ */
/* Inflate  zd.ml:530-718                                                    */

typedef struct {
  const uint8_t *src;
  int64_t src_max;      /* zd.ml:534 (index of last byte; -1 when empty) */
  int64_t src_pos;      /* zd.ml:535 */
  uint64_t src_bits;    /* zd.ml:536 */
  int src_bits_len;     /* zd.ml:537 */
  zd_buf dst;
  huff_decoder dyn_litlen, dyn_dist;
  int crc_op;
  uint32_t crc;
  size_t crc_next;
  int scratch_lengths[HUFF_MAX_SYMBOL_COUNT + DIST_SYM_FIXED_MAX + 1]; /* zd.ml:352 */
  zd_exn *x;
} decoder;

/* read_bits zd.ml:564-579: refills ONE BYTE AT A TIME, only when needed */
static inline int64_t read_bits(decoder *d, int count) {
  uint64_t bits = d->src_bits;
  int bits_len = d->src_bits_len;
  while (bits_len < count) {
    if (d->src_pos > d->src_max) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    bits |= (uint64_t)d->src[d->src_pos] << bits_len;
    d->src_pos++;
    bits_len += 8;
  }
  int64_t ret = (int64_t)(bits & (((uint64_t)1 << count) - 1));
  d->src_bits = bits >> count;
  d->src_bits_len = bits_len - count;
  return ret;
}

/* read_int zd.ml:581-582 */
static inline int64_t read_int(decoder *d, int64_t base, int bit_count) {
  return base + (bit_count == 0 ? 0 : read_bits(d, bit_count));
}

/* read_symbol zd.ml:584-591: canonical walk, one bit per code bit.  The
 * reference indexes counts.(16) out of bounds (Invalid_argument, which is NOT
 * caught as Failure) when the decoder holds an incomplete/empty code; the
 * boundary maps that to "Corrupted data stream" (SURVEY.md 8b.4). */
static int read_symbol(decoder *d, const huff_decoder *huff) {
  int len = 1, base = 0;
  int64_t offs = 0;
  for (;;) {
    offs = 2 * offs + read_bits(d, 1);
    if (len > HUFF_MAX_CODE_BIT_LENGTH) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    int count = huff->counts[len];
    if (offs < count) return huff->symbols[base + offs];
    len++;
    base += count;
    offs -= count;
  }
}

/* read_block_symbols zd.ml:593-616 */
static void read_block_symbols(decoder *d, const huff_decoder *hlitlen,
                               const huff_decoder *hdist) {
  for (;;) {
    int sym = read_symbol(d, hlitlen);
    if (sym < LITLEN_EOB) { buf_add_uint8(&d->dst, sym); continue; }
    if (sym == LITLEN_EOB) return;
    if (sym > hlitlen->max_sym || sym > LITLEN_SYM_MAX || hlitlen->max_sym == -1)
      ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    int lv = length_value_of_sym_table[sym - LITLEN_FIRST_LEN_SYM];
    int64_t length = read_int(d, VALUE_BASE(lv), VALUE_EXTRA_BITS(lv));
    int dsym = read_symbol(d, hdist);
    if (dsym > hdist->max_sym || dsym > DIST_SYM_MAX) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    int dv = dist_value_of_sym[dsym];
    int64_t dist = read_int(d, VALUE_BASE(dv), VALUE_EXTRA_BITS(dv));
    if (dist > (int64_t)d->dst.len) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    buf_recopy(&d->dst, d->dst.len - (size_t)dist, (size_t)length);
  }
}

/* read_fixed_block zd.ml:618-621 */
static void read_fixed_block(decoder *d) {
  huff_decoder litlen, dist;
  fixed_litlen_decoder(&litlen);
  fixed_dist_decoder(&dist);
  read_block_symbols(d, &litlen, &dist);
}

/* read_dynamic_block zd.ml:623-669 */
static void read_dynamic_block(decoder *d) {
  /* read_dynamic_codes zd.ml:638-667 */
  int hlit = (int)read_int(d, 257, 5);
  int hdist = (int)read_int(d, 1, 5);
  if (hlit > MAX_LITLEN_SYM_COUNT || hdist > MAX_DIST_SYM_COUNT)
    ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
  /* read_codelen_code zd.ml:624-636 */
  int hclen = (int)read_int(d, 4, 4);
  int *lengths = d->scratch_lengths;
  for (int i = 0; i < MAX_CODELEN_SYM_COUNT; i++) lengths[i] = 0;
  for (int i = 0; i < hclen; i++)
    lengths[codelen_order_of_sym_lengths[i]] = (int)read_bits(d, 3);
  huff_decoder *huff = &d->dyn_litlen; /* temporarily used for that code */
  init_decoder(d->x, huff, lengths, 0, MAX_CODELEN_SYM_COUNT);
  if (huff->max_sym == -1) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
  /* decode code lengths for the dynamic litlen and dist codes */
  int num = 0;
  while (num < hlit + hdist) {
    int sym = read_symbol(d, huff);
    if (sym > huff->max_sym) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    int repeat = 0;
    switch (sym) {
    case 16:
      if (num == 0) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
      repeat = (int)read_int(d, 3, 2);
      sym = lengths[num - 1];
      break;
    case 17: repeat = (int)read_int(d, 3, 3); sym = 0; break;
    case 18: repeat = (int)read_int(d, 11, 7); sym = 0; break;
    default: repeat = 1; break;
    }
    if (repeat > hlit + hdist - num) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
    while (repeat > 0) { repeat--; lengths[num] = sym; num++; }
  }
  if (lengths[256] == 0) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
  init_decoder(d->x, &d->dyn_litlen, lengths, 0, hlit);
  init_decoder(d->x, &d->dyn_dist, lengths, hlit, hdist);
  read_block_symbols(d, &d->dyn_litlen, &d->dyn_dist);
}

/* read_uncompressed_block zd.ml:671-680 */
static void read_uncompressed_block(decoder *d) {
  if (d->src_max - d->src_pos + 1 < 4) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
  int64_t length = d->src[d->src_pos] | (d->src[d->src_pos + 1] << 8);
  int64_t inv_length = d->src[d->src_pos + 2] | (d->src[d->src_pos + 3] << 8);
  if (length != ((~inv_length) & 0xFFFF)) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
  d->src_pos += 4;
  if (d->src_max - d->src_pos + 1 < length) ZD_RAISE(d->x, ZD_ERR_CORRUPTED);
  buf_add_string(&d->dst, d->src, (size_t)d->src_pos, (size_t)length);
  d->src_pos += length;
  d->src_bits = 0;
  d->src_bits_len = 0;
}

/* inflated_block_crc zd.ml:682-690: ONE update call per deflate block */
static void inflated_block_crc(decoder *d) {
  size_t crc_next = d->dst.len, start = d->crc_next, len = crc_next - start;
  d->crc_next = crc_next;
  d->crc = crc_op_update(d->crc_op, d->crc, d->dst.b + start, len);
}

/* inflate_and_crc zd.ml:692-709 with make_decoder zd.ml:548-562 */
static int inflate_common(const uint8_t *src, size_t len, int has_limit, size_t limit,
                          int crc_op, uint8_t *ext_dst, size_t ext_cap, uint8_t **out,
                          size_t *out_len, uint32_t *crc) {
  format_tables_init();
  zd_exn x;
  decoder *d = (decoder *)calloc(1, sizeof *d);
  if (!d) return ZD_ERR_NOMEM;
  d->x = &x;
  d->dst.b = NULL;
  int status = setjmp(x.jb);
  if (status != 0) {
    if (!d->dst.external) free(d->dst.b);
    free(d);
    if (out) *out = NULL;
    if (out_len) *out_len = 0;
    return status;
  }
  d->src = src;
  d->src_max = (int64_t)len - 1;
  d->src_pos = 0;
  d->src_bits = 0;
  d->src_bits_len = 0;
  if (ext_dst) { /* caller storage: same accept/reject, no growth */
    d->dst.b = ext_dst;
    d->dst.cap = has_limit ? (limit < ext_cap ? limit : ext_cap) : ext_cap;
    d->dst.len = 0;
    d->dst.fixed = has_limit && limit <= ext_cap;
    d->dst.external = 1;
    d->dst.x = &x;
  } else if (has_limit) {
    buf_make(&d->dst, &x, 1, limit); /* zd.ml:554 */
  } else {
    buf_make(&d->dst, &x, 0, len * 3); /* zd.ml:553 */
  }
  d->crc_op = crc_op;
  d->crc = crc_op_init(crc_op);
  d->crc_next = 0;
  for (;;) { /* inflate_loop zd.ml:694-705 */
    int final = read_bits(d, 1) == 1;
    int btype = (int)read_bits(d, 2);
    switch (btype) {
    case 0: read_uncompressed_block(d); break;
    case 1: read_fixed_block(d); break;
    case 2: read_dynamic_block(d); break;
    default: ZD_RAISE(&x, ZD_ERR_CORRUPTED);
    }
    inflated_block_crc(d);
    if (final) break;
  }
  if (crc) *crc = crc_op_finish(crc_op, d->crc);
  if (out_len) *out_len = d->dst.len;
  if (out) *out = d->dst.b; /* Buf.contents */
  else if (!d->dst.external) free(d->dst.b);
  free(d);
  return ZD_OK;
}

int zd_inflate(const uint8_t *src, size_t len, int has_limit, size_t limit, int crc_op,
               uint8_t **out, size_t *out_len, uint32_t *crc) {
  return inflate_common(src, len, has_limit, limit, crc_op, NULL, 0, out, out_len, crc);
}

int zd_inflate_into(const uint8_t *src, size_t len, int has_limit, size_t limit,
                    int crc_op, uint8_t *dst, size_t dst_cap, size_t *out_len,
                    uint32_t *crc) {
  static uint8_t dummy;
  return inflate_common(src, len, has_limit, limit, crc_op, dst ? dst : &dummy, dst_cap,
                        NULL, out_len, crc);
}

/* zlib_decompress zd.ml:720-740 (start = 0) */
int zd_zlib_decompress(const uint8_t *s, size_t len, int has_limit, size_t limit,
                       uint8_t **out, size_t *out_len, uint32_t *adler, uint32_t *expect,
                       uint32_t *found) {
  if (out) *out = NULL;
  if (out_len) *out_len = 0;
  if (len < 6) return ZD_ERR_CORRUPTED; /* header and trailer */
  int cmf = s[0], flg = s[1];
  if ((256 * cmf + flg) % 31 != 0) return ZD_ERR_CORRUPTED;
  int cm = cmf & 0x0F;
  if (cm != 8) return ZD_ERR_ZLIB_METHOD;
  if ((cmf >> 4) > 7) return ZD_ERR_ZLIB_WINDOW;
  if ((flg & 0x20) != 0) return ZD_ERR_ZLIB_DICT;
  uint32_t e = ((uint32_t)s[len - 4] << 24) | ((uint32_t)s[len - 3] << 16) |
               ((uint32_t)s[len - 2] << 8) | (uint32_t)s[len - 1];
  /* start = 2, len = len - 4: the 4 trailer bytes stay in range (zd.ml:732);
   * harmless since inflate stops at the final block */
  uint8_t *o = NULL;
  size_t ol = 0;
  uint32_t f = 0;
  int st = zd_inflate(s + 2, len - 4, has_limit, limit, ZD_CRC_ADLER32, &o, &ol, &f);
  if (st != ZD_OK) return st;
  if (expect) *expect = e;
  if (found) *found = f;
  if (e != f) { free(o); return ZD_ERR_CHECKSUM; }
  if (adler) *adler = f;
  if (out) *out = o; else free(o);
  if (out_len) *out_len = ol;
  return ZD_OK;
}

/* ------------------------------------------------------------------------ */
/* Deflate  zd.ml:742-1277                                                   */

enum {
  LZ77_NO_POS = -1,          /* zd.ml:744 */
  LZ77_WINDOW_SIZE = 32768,  /* zd.ml:745 */
  LZ77_HASH_BIT_SIZE = 15,   /* zd.ml:746 */
  MAX_BLOCK_SRC_LEN = 65534, /* zd.ml:747-750 */
  MIN_MATCH_LEN = 4,         /* zd.ml:1141 */
  MAX_MATCH_LEN = 258,       /* zd.ml:1142 */
  MAX_MATCH_DIST = 32768     /* zd.ml:1143 */
};

/* backref zd.ml:766-776 */
#define MAKE_BACKREF(dist, len) ((int32_t)(((dist) << 9) | (len)))
#define BACKREF_DIST(v) ((v) >> 9)
#define BACKREF_LEN(v) ((v) & 0x1FF)

typedef struct {
  int level;
  const uint8_t *src;
  int64_t src_len;
  zd_buf dst;
  uint64_t dst_bits;
  int dst_bits_len;
  int32_t *block_syms; /* MAX_BLOCK_SRC_LEN + 1 */
  int64_t block_syms_len;
  int64_t block_src_start, block_src_len;
  int64_t litlen_sym_freqs[LITLEN_SYM_MAX + 1];
  int64_t dist_sym_freqs[DIST_SYM_MAX + 1];
  int codelen_syms[LITLEN_SYM_MAX + DIST_SYM_MAX + 2];
  int codelen_syms_len;
  int64_t codelen_sym_freqs[CODELEN_SYM_MAX + 1]; /* NEVER reset between blocks (Q1) */
  int good_match, max_chain_len;
  int32_t *hash_head; /* 1 << 15, init -1 */
  int32_t *hash_prev; /* 32768 */
  huff_encoder dyn_litlen, dyn_dist, dyn_codelen;
  int hlit, hdist, hclen;
  int crc_op;
  uint32_t crc;
  int64_t scratch_heap[HUFF_MAX_SYMBOL_COUNT * 2 + 1];
  zd_exn *x;
  /* trace */
  zd_block_info *blocks;
  size_t max_blocks, n_blocks;
} encoder;

/* new_block zd.ml:849-854 (codelen_sym_freqs is NOT cleared: Q1) */
static void new_block(encoder *e) {
  e->block_syms_len = 0;
  e->block_src_start += e->block_src_len;
  e->block_src_len = 0;
  memset(e->litlen_sym_freqs, 0, sizeof e->litlen_sym_freqs);
  memset(e->dist_sym_freqs, 0, sizeof e->dist_sym_freqs);
}

/* flush zd.ml:856-858 */
static void enc_flush(encoder *e) {
  if (e->dst_bits_len > 0) {
    buf_add_uint8(&e->dst, (int64_t)(e->dst_bits & 0xFF));
    e->dst_bits = 0;
    e->dst_bits_len = 0;
  }
}

/* write_bits zd.ml:864-871 */
static inline void write_bits(encoder *e, uint64_t v, int count) {
  e->dst_bits = (v << e->dst_bits_len) | e->dst_bits;
  e->dst_bits_len += count;
  while (e->dst_bits_len >= 8) {
    buf_add_uint8(&e->dst, (int64_t)(e->dst_bits & 0xFF));
    e->dst_bits >>= 8;
    e->dst_bits_len -= 8;
  }
}

/* write_non_compressed_block zd.ml:873-877 */
static void write_non_compressed_block(encoder *e, int final) {
  int64_t len = e->block_src_len;
  write_bits(e, final ? 1 : 0, 3);
  enc_flush(e);
  buf_add_uint16_le(&e->dst, len);
  buf_add_uint16_le(&e->dst, ~len);
  buf_add_string(&e->dst, e->src, (size_t)e->block_src_start, (size_t)len);
}

/* write_block_symbols zd.ml:879-910 */
static void write_block_symbols(encoder *e, const int *huffman_litlen,
                                const int *huffman_dist) {
  for (int64_t i = 0; i < e->block_syms_len; i++) {
    int32_t bref = e->block_syms[i];
    int dist = BACKREF_DIST(bref), len = BACKREF_LEN(bref);
    if (dist == 0) { /* a literal or the end of block */
      int si = huffman_litlen[len];
      write_bits(e, (uint64_t)SYM_CODE(si), SYM_CODE_LENGTH(si));
    } else {
      int litlen_sym = length_value_to_sym[len];
      int si = huffman_litlen[litlen_sym];
      uint64_t bits = (uint64_t)SYM_CODE(si);
      int count = SYM_CODE_LENGTH(si);
      int ll = length_value_of_sym_table[litlen_sym - LITLEN_FIRST_LEN_SYM];
      uint64_t extra_bits = (uint64_t)(len - VALUE_BASE(ll));
      int extra_bits_count = VALUE_EXTRA_BITS(ll);
      bits = (extra_bits << count) | bits;
      write_bits(e, bits, count + extra_bits_count);
      int dist_sym = dist_value_to_sym(dist);
      si = huffman_dist[dist_sym];
      bits = (uint64_t)SYM_CODE(si);
      count = SYM_CODE_LENGTH(si);
      int dv = dist_value_of_sym[dist_sym];
      extra_bits = (uint64_t)(dist - VALUE_BASE(dv));
      bits = (extra_bits << count) | bits;
      extra_bits_count = VALUE_EXTRA_BITS(dv);
      write_bits(e, bits, count + extra_bits_count);
    }
  }
}

/* write_fixed_huffman_block zd.ml:912-916 */
static void write_fixed_huffman_block(encoder *e, int final) {
  write_bits(e, final ? 3 : 2, 3);
  write_block_symbols(e, fixed_litlen_enc.e, fixed_dist_enc.e);
}

/* write_dynamic_huffman_block zd.ml:918-945 */
static void write_dynamic_huffman_block(encoder *e, int final) {
  write_bits(e, final ? 5 : 4, 3);
  write_bits(e, (uint64_t)e->hlit, 5);
  write_bits(e, (uint64_t)e->hdist, 5);
  write_bits(e, (uint64_t)e->hclen, 4);
  for (int o = 0; o < e->hclen + 4; o++) {
    int sym = codelen_order_of_sym_lengths[o];
    write_bits(e, (uint64_t)SYM_CODE_LENGTH(e->dyn_codelen.e[sym]), 3);
  }
  for (int l = 0; l < e->codelen_syms_len; l++) {
    int symref = e->codelen_syms[l];
    int sym = symref & 0xFF; /* zd.ml:308 */
    int si = e->dyn_codelen.e[sym];
    uint64_t bits = (uint64_t)SYM_CODE(si);
    int count = SYM_CODE_LENGTH(si);
    if (sym <= 15) write_bits(e, bits, count);
    else {
      uint64_t repeat_bits = (uint64_t)(symref >> 8); /* zd.ml:309 */
      bits = (repeat_bits << count) | bits;
      int rb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
      write_bits(e, bits, count + rb);
    }
  }
  write_block_symbols(e, e->dyn_litlen.e, e->dyn_dist.e);
}

/* huffman_init_with_freqs zd.ml:947-951 = lengths_of_freqs; init_with_lengths */
static void huffman_init_with_freqs(encoder *e, huff_encoder *h, const int64_t *freqs,
                                    int max_sym, int max_code_len) {
  lengths_of_freqs(e->scratch_heap, h->e, freqs, max_sym, max_code_len);
  init_with_lengths(h->e, max_sym);
}

/* make_dynamic_huffman zd.ml:953-957 */
static void make_dynamic_huffman(encoder *e) {
  huffman_init_with_freqs(e, &e->dyn_litlen, e->litlen_sym_freqs, LITLEN_SYM_MAX, 15);
  huffman_init_with_freqs(e, &e->dyn_dist, e->dist_sym_freqs, DIST_SYM_MAX, 15);
}

/* code_length_count zd.ml:966-970 */
static int code_length_count(const int *h, int max_sym) {
  int sym = max_sym;
  while (sym >= 0 && SYM_CODE_LENGTH(h[sym]) == 0) sym--;
  return sym + 1;
}

/* make_dynamic_huffman_encoding zd.ml:959-1043 */
static void make_dynamic_huffman_encoding(encoder *e) {
  /* gather_dynamic_huffman_code_lengths zd.ml:963-988 */
  int litlen_count = code_length_count(e->dyn_litlen.e, LITLEN_SYM_MAX);
  int dist_count = code_length_count(e->dyn_dist.e, DIST_SYM_MAX);
  if (dist_count == 0) { /* HDIST 0 means 1: patch symbol 0 to length 1, code 0 */
    e->dyn_dist.e[0] = SYM_INFO(0, 1);
    dist_count = 1;
  }
  e->hlit = litlen_count - 257;
  e->hdist = dist_count - 1;
  int *l = e->codelen_syms;
  for (int i = 0; i < litlen_count; i++) l[i] = SYM_CODE_LENGTH(e->dyn_litlen.e[i]);
  for (int i = 0; i < dist_count; i++)
    l[litlen_count + i] = SYM_CODE_LENGTH(e->dyn_dist.e[i]);
  int length_count = litlen_count + dist_count;
  /* compute_codelen_syms zd.ml:989-1030: reads and writes e->codelen_syms in
   * place (the encoding never expands) */
  int *lengths = e->codelen_syms;
  int len_max = length_count - 1;
  int k = 0, i = 0;
  while (i <= len_max) {
    if (lengths[i] == 0) {
      int max = len_max < i + 138 - 1 ? len_max : i + 138 - 1;
      int j = i + 1;
      while (j <= max && lengths[j] == 0) j++;
      int zcount = j - i, next;
      if (zcount < 3) { /* ONE zero, advance by one */
        e->codelen_syms[k] = 0; e->codelen_sym_freqs[0]++; next = i + 1;
      } else if (zcount <= 10) {
        e->codelen_syms[k] = ((zcount - 3) << 8) | 17; e->codelen_sym_freqs[17]++; next = j;
      } else {
        e->codelen_syms[k] = ((zcount - 11) << 8) | 18; e->codelen_sym_freqs[18]++; next = j;
      }
      k++;
      i = next;
    } else {
      int sym = lengths[i];
      e->codelen_syms[k] = sym; e->codelen_sym_freqs[sym]++;
      int max = len_max < i + 6 ? len_max : i + 6;
      int j = i + 1;
      while (j <= max && lengths[j] == sym) j++;
      int scount = j - i;
      if (scount <= 3) { k++; i = i + 1; }
      else {
        e->codelen_syms[k + 1] = ((scount - 3 - 1) << 8) | 16;
        e->codelen_sym_freqs[16]++;
        k += 2;
        i = j;
      }
    }
  }
  e->codelen_syms_len = k;
  /* the codelen code uses the ACCUMULATED codelen_sym_freqs (Q1) */
  huffman_init_with_freqs(e, &e->dyn_codelen, e->codelen_sym_freqs, CODELEN_SYM_MAX, 7);
  /* codelen_length_count zd.ml:1032-1036 */
  int o = CODELEN_SYM_MAX;
  while (o > 0 && SYM_CODE_LENGTH(e->dyn_codelen.e[codelen_order_of_sym_lengths[o]]) == 0) o--;
  e->hclen = (o + 1) - 4;
}

/* bit_length_of_non_compressed_block zd.ml:1045-1047 (Q3: 8, not 0, when aligned) */
static int64_t bit_length_of_non_compressed_block(const encoder *e) {
  int alignment_loss = 8 - ((e->dst_bits_len + 3) % 8);
  return 3 + alignment_loss + (4 + e->block_src_len) * 8;
}

/* bit_length_of_block_symbols zd.ml:1049-1064 */
static int64_t bit_length_of_block_symbols(const encoder *e, const int *hlitlen,
                                           const int *hdist) {
  int64_t acc = 0;
  for (int sym = 0; sym <= LITLEN_SYM_MAX; sym++) {
    int code_length = SYM_CODE_LENGTH(hlitlen[sym]);
    int extra_bits = sym < LITLEN_FIRST_LEN_SYM
                         ? 0
                         : VALUE_EXTRA_BITS(length_value_of_sym_table[sym - 257]);
    acc += e->litlen_sym_freqs[sym] * (code_length + extra_bits);
  }
  for (int sym = 0; sym <= DIST_SYM_MAX; sym++) {
    int code_length = SYM_CODE_LENGTH(hdist[sym]);
    int extra_bits = VALUE_EXTRA_BITS(dist_value_of_sym[sym]);
    acc += e->dist_sym_freqs[sym] * (code_length + extra_bits);
  }
  return acc;
}

/* bit_length_of_fixed_huffman_block zd.ml:1066-1069 */
static int64_t bit_length_of_fixed_huffman_block(const encoder *e) {
  return 3 + bit_length_of_block_symbols(e, fixed_litlen_enc.e, fixed_dist_enc.e);
}

/* bit_length_of_dynamic_huffman_block zd.ml:1071-1079 */
static int64_t bit_length_of_dynamic_huffman_block(const encoder *e) {
  int codelen_length_count = e->hclen + 4;
  int64_t acc = 3 + 5 + 5 + 4 + 3 * codelen_length_count;
  for (int sym = 0; sym <= CODELEN_SYM_MAX; sym++) {
    int len = SYM_CODE_LENGTH(e->dyn_codelen.e[sym]);
    int repeat_bits = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
    acc += e->codelen_sym_freqs[sym] * (len + repeat_bits);
  }
  return acc + bit_length_of_block_symbols(e, e->dyn_litlen.e, e->dyn_dist.e);
}

/* deflated_block_src_crc zd.ml:1081-1086: one update call per block */
static void deflated_block_src_crc(encoder *e) {
  e->crc = crc_op_update(e->crc_op, e->crc, e->src + e->block_src_start,
                         (size_t)e->block_src_len);
}

/* add_end_of_block_sym zd.ml:1088-1092 */
static void add_end_of_block_sym(encoder *e) {
  e->block_syms[e->block_syms_len] = LITLEN_EOB;
  e->block_syms_len++;
  e->litlen_sym_freqs[LITLEN_EOB] = 1;
}

/* write_block zd.ml:1094-1104 */
static void write_block(encoder *e, int final) {
  deflated_block_src_crc(e);
  add_end_of_block_sym(e);
  make_dynamic_huffman(e);
  make_dynamic_huffman_encoding(e);
  int64_t nlen = bit_length_of_non_compressed_block(e);
  int64_t flen = bit_length_of_fixed_huffman_block(e);
  int64_t dlen = bit_length_of_dynamic_huffman_block(e);
  int kind;
  if (nlen <= dlen && nlen <= flen) kind = ZD_BLOCK_STORED;
  else if (flen <= dlen) kind = ZD_BLOCK_FIXED;
  else kind = ZD_BLOCK_DYNAMIC;
  if (e->blocks && e->n_blocks < e->max_blocks) {
    zd_block_info *b = &e->blocks[e->n_blocks];
    b->kind = kind; b->final = final;
    b->src_start = (uint32_t)e->block_src_start;
    b->src_len = (uint32_t)e->block_src_len;
    b->n_syms = (uint32_t)e->block_syms_len;
    b->nlen = nlen; b->flen = flen; b->dlen = dlen;
  }
  e->n_blocks++;
  if (kind == ZD_BLOCK_STORED) write_non_compressed_block(e, final);
  else if (kind == ZD_BLOCK_FIXED) write_fixed_huffman_block(e, final);
  else write_dynamic_huffman_block(e, final);
}

/* write_all_non_compressed zd.ml:1106-1116 (`None level fast path) */
static void write_all_non_compressed(encoder *e) {
  int64_t src_max = e->src_len - 1;
  for (;;) {
    int64_t start = e->block_src_start;
    int64_t block_max = src_max < start + MAX_BLOCK_SRC_LEN - 1 ? src_max
                                                               : start + MAX_BLOCK_SRC_LEN - 1;
    int64_t len = block_max - start + 1;
    int final = block_max == src_max;
    e->block_src_len = len;
    deflated_block_src_crc(e);
    if (e->blocks && e->n_blocks < e->max_blocks) {
      zd_block_info *b = &e->blocks[e->n_blocks];
      memset(b, 0, sizeof *b);
      b->kind = ZD_BLOCK_STORED; b->final = final;
      b->src_start = (uint32_t)start; b->src_len = (uint32_t)len;
    }
    e->n_blocks++;
    write_non_compressed_block(e, final);
    if (final) return;
    e->block_src_start = start + len;
  }
}

/* write_block_symbol zd.ml:1118-1123 */
static inline void write_block_symbol(encoder *e, int32_t sym, int64_t src_len) {
  if (e->block_src_len + src_len > MAX_BLOCK_SRC_LEN) {
    write_block(e, 0);
    new_block(e);
  }
  e->block_syms[e->block_syms_len] = sym;
  e->block_syms_len++;
  e->block_src_len += src_len;
}

/* write_lit_symbol zd.ml:1125-1128 */
static inline void write_lit_symbol(encoder *e, int byte) {
  write_block_symbol(e, (int32_t)byte, 1);
  e->litlen_sym_freqs[byte]++;
}

/* write_backref_symbol zd.ml:1130-1136 */
static inline void write_backref_symbol(encoder *e, int32_t bref) {
  int len = BACKREF_LEN(bref);
  write_block_symbol(e, bref, len);
  e->litlen_sym_freqs[length_value_to_sym[len]]++;
  e->dist_sym_freqs[dist_value_to_sym(BACKREF_DIST(bref))]++;
}

/* Lz77.hash4 zd.ml:1145-1148 */
static inline int hash4(const uint8_t *s, int64_t i) {
  uint32_t v = (uint32_t)s[i] | ((uint32_t)s[i + 1] << 8) | ((uint32_t)s[i + 2] << 16) |
               ((uint32_t)s[i + 3] << 24);
  return (int)((v * 0x9E3779B1u) >> (32 - LZ77_HASH_BIT_SIZE));
}

/* Lz77.insert_hash zd.ml:1150-1152 */
static inline void insert_hash(encoder *e, int hash, int64_t pos) {
  e->hash_prev[pos % LZ77_WINDOW_SIZE] = e->hash_head[hash];
  e->hash_head[hash] = (int32_t)pos;
}

/* Lz77.find_match_length zd.ml:1154-1174 */
static inline int find_match_length(const uint8_t *s, int64_t i, int64_t j,
                                    int prev_match_len, int max_match_len) {
  int64_t a = i + prev_match_len, b = j + prev_match_len;
  int len = prev_match_len;
  /* match_bwd: compare positions +prev_match_len down to +0 */
  while (len >= 0 && s[a] == s[b]) { a--; b--; len--; }
  if (len >= 0) return 0;
  /* match_fwd from +prev_match_len+1 */
  a = i + prev_match_len + 1; b = j + prev_match_len + 1;
  len = prev_match_len + 1;
  while (len < max_match_len && s[a] == s[b]) { a++; b++; len++; }
  return len;
}

/* Lz77.find_backref zd.ml:1176-1201 */
static int32_t find_backref(encoder *e, int64_t pos, int hash, int prev_match_len,
                            int max_match_len) {
  if (prev_match_len == 0) prev_match_len = MIN_MATCH_LEN - 1;
  if (prev_match_len >= max_match_len) return 0;
  int chain_steps = e->max_chain_len;
  if (prev_match_len >= e->good_match) chain_steps = chain_steps / 4;
  int64_t i = e->hash_head[hash];
  int64_t match_pos = LZ77_NO_POS;
  for (;;) {
    if (i == LZ77_NO_POS || chain_steps == 0 || pos - i > MAX_MATCH_DIST) {
      if (match_pos == LZ77_NO_POS) return 0;
      return MAKE_BACKREF((int32_t)(pos - match_pos), prev_match_len);
    }
    chain_steps--;
    int len = find_match_length(e->src, i, pos, prev_match_len, max_match_len);
    if (len == max_match_len) return MAKE_BACKREF((int32_t)(pos - i), len);
    if (len != 0) { match_pos = i; prev_match_len = len; }
    i = e->hash_prev[i % LZ77_WINDOW_SIZE];
  }
}

/* Lz77.compress zd.ml:1203-1244 (src_start = 0) */
static void lz77_compress(encoder *e) {
  if (e->level == ZD_LEVEL_NONE) { write_all_non_compressed(e); return; }
  const uint8_t *s = e->src;
  int64_t max_pos = e->src_len - MIN_MATCH_LEN;
  int64_t i = 0;
  int32_t prev_backref = 0;
  for (;;) {
    int prev_match_len = BACKREF_LEN(prev_backref);
    if (i > max_pos) {
      if (prev_match_len != 0) { /* write pending previous match */
        write_backref_symbol(e, prev_backref);
        i = max_pos + prev_match_len;
      }
      for (int64_t k = i; k <= e->src_len - 1; k++) write_lit_symbol(e, s[k]);
      write_block(e, 1);
      enc_flush(e);
      return;
    }
    int hash = hash4(s, i);
    int max_match_len = MAX_MATCH_LEN < e->src_len - i ? MAX_MATCH_LEN : (int)(e->src_len - i);
    int32_t bref = find_backref(e, i, hash, prev_match_len, max_match_len);
    int match_len = BACKREF_LEN(bref);
    insert_hash(e, hash, i);
    if (prev_match_len != 0 && prev_match_len > match_len) {
      /* previous match at least as good: write it and move past it */
      write_backref_symbol(e, prev_backref);
      int64_t next = (i - 1) + prev_match_len;
      int64_t last = next - 1 < max_pos ? next - 1 : max_pos;
      for (int64_t j = i + 1; j <= last; j++) insert_hash(e, hash4(s, j), j);
      i = next;
      prev_backref = 0;
    } else if (match_len == 0) { /* no match and no previous match */
      write_lit_symbol(e, s[i]);
      i = i + 1;
      prev_backref = 0;
    } else { /* current better than previous: defer it */
      if (prev_match_len != 0) write_lit_symbol(e, s[i - 1]);
      i = i + 1;
      prev_backref = bref;
    }
  }
}

/* level_params zd.ml:754-764 (max_lazy and nice_length are never read) */
static void level_params(int level, int *good_match, int *max_chain_len) {
  switch (level) {
  case ZD_LEVEL_NONE: *good_match = 0; *max_chain_len = 0; break;
  case ZD_LEVEL_FAST: *good_match = 4; *max_chain_len = 4; break;
  case ZD_LEVEL_DEFAULT: *good_match = 8; *max_chain_len = 128; break;
  default: *good_match = 32; *max_chain_len = 4096; break;
  }
}

static void encoder_free(encoder *e) {
  if (!e) return;
  free(e->block_syms);
  free(e->hash_head);
  free(e->hash_prev);
  free(e);
}

/* make_encoder zd.ml:817-847 (level is explicit at this boundary) */
static encoder *make_encoder(zd_exn *x, const uint8_t *src, size_t len, int level,
                             int crc_op) {
  format_tables_init();
  fixed_encoders_init();
  encoder *e = (encoder *)calloc(1, sizeof *e);
  if (!e) return NULL;
  e->x = x;
  e->level = level;
  e->src = src;
  e->src_len = (int64_t)len;
  e->block_syms = (int32_t *)calloc(MAX_BLOCK_SRC_LEN + 1, sizeof(int32_t));
  e->hash_head = (int32_t *)malloc(sizeof(int32_t) << LZ77_HASH_BIT_SIZE);
  e->hash_prev = (int32_t *)calloc(LZ77_WINDOW_SIZE, sizeof(int32_t));
  if (!e->block_syms || !e->hash_head || !e->hash_prev) { encoder_free(e); return NULL; }
  for (int i = 0; i < (1 << LZ77_HASH_BIT_SIZE); i++) e->hash_head[i] = LZ77_NO_POS;
  level_params(level, &e->good_match, &e->max_chain_len);
  e->crc_op = crc_op;
  e->crc = crc_op_init(crc_op);
  return e;
}

int zd_deflate_trace(const uint8_t *src, size_t len, int level, int crc_op, uint8_t **out,
                     size_t *out_len, uint32_t *crc, zd_block_info *blocks,
                     size_t max_blocks, size_t *n_blocks) {
  zd_exn x;
  encoder *volatile ev = NULL;
  int status = setjmp(x.jb);
  if (status != 0) {
    if (ev) { free(ev->dst.b); encoder_free(ev); }
    if (out) *out = NULL;
    if (out_len) *out_len = 0;
    return status;
  }
  encoder *e = make_encoder(&x, src, len, level, crc_op);
  if (!e) return ZD_ERR_NOMEM;
  ev = e;
  e->blocks = blocks; e->max_blocks = max_blocks; e->n_blocks = 0;
  buf_make(&e->dst, &x, 0, len); /* zd.ml:819 */
  lz77_compress(e);
  if (crc) *crc = crc_op_finish(crc_op, e->crc);
  if (out_len) *out_len = e->dst.len;
  if (n_blocks) *n_blocks = e->n_blocks;
  if (out) *out = e->dst.b; else free(e->dst.b);
  encoder_free(e);
  return ZD_OK;
}

int zd_deflate(const uint8_t *src, size_t len, int level, int crc_op, uint8_t **out,
               size_t *out_len, uint32_t *crc) {
  return zd_deflate_trace(src, len, level, crc_op, out, out_len, crc, NULL, 0, NULL);
}

/* zlib_compress zd.ml:1262-1277 (start = 0) */
int zd_zlib_compress(const uint8_t *src, size_t len, int level, uint8_t **out,
                     size_t *out_len, uint32_t *adler) {
  zd_exn x;
  encoder *volatile ev = NULL;
  int status = setjmp(x.jb);
  if (status != 0) {
    if (ev) { free(ev->dst.b); encoder_free(ev); }
    if (out) *out = NULL;
    if (out_len) *out_len = 0;
    return status;
  }
  encoder *e = make_encoder(&x, src, len, level, ZD_CRC_ADLER32);
  if (!e) return ZD_ERR_NOMEM;
  ev = e;
  buf_make(&e->dst, &x, 0, len);
  int cmf = (7 << 4) | 8; /* 32k window, deflate */
  int flevel = level; /* `None 0, `Fast 1, `Default 2, `Best 3 */
  int header = (cmf << 8) | (flevel << 6);
  int flg = (header + 31 - (header % 31)) & 0xFF;
  buf_add_uint8(&e->dst, cmf);
  buf_add_uint8(&e->dst, flg);
  lz77_compress(e);
  uint32_t crc = crc_op_finish(ZD_CRC_ADLER32, e->crc);
  buf_add_uint32_be(&e->dst, crc);
  if (adler) *adler = crc;
  if (out_len) *out_len = e->dst.len;
  if (out) *out = e->dst.b; else free(e->dst.b);
  encoder_free(e);
  return ZD_OK;
}

size_t zd_deflate_bound(size_t len) {
  /* every block stored: 5 bytes of header per <= 65534 source bytes; a
   * compressed block is only chosen when it is strictly smaller than the
   * stored estimate, and an empty input is 2 bytes (fixed EOB) or 5 (`None) */
  size_t blocks = len / MAX_BLOCK_SRC_LEN + 1;
  return len + 6 * blocks + 8; /* +1/block: the stored estimate may be 8 bits high (Q3) */
}
