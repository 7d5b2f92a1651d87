"""A SECOND, independent reading of the reference's encode side and of its Adler-32, in naive Python.

Written in round 6 from /root/reference/src/zipc_deflate.ml (`zd.ml`) alone -- lines 166-206 (Adler_32), 404-528
(Huffman encoder), 742-1277 (deflate) -- WITHOUT consulting oracle/zd_oracle.c, so that the two restatements are two
readings that must agree; a disagreement is a finding against a zd.ml line number, not a bug to paper over.  It is
test infrastructure of the slowest kind (a minute per 100 KB of text at `Best): only tests/golden/make_deflate_vectors.py
runs it, in the build container, and what travels is the JSON it writes.

The structure follows the OCaml one function at a time (same names, same mutable record, same arrays doubling as
scratch), not the oracle's and not the kernels'.  OCaml's `int` is 63 bits: Python's integers never wrap where OCaml's
would not; the places where the reference computes in Int32 (the checksums, hash4) are wrapped by hand.
"""
import struct

# ---- Int32 arithmetic (zd.ml:83-99: Uint32 is int32, `mod` is Int32.rem) ------------------------------------------


def i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def i32_rem(a, b):
    """Int32.rem: truncated division, the sign of the dividend"""
    q = abs(a) // abs(b)
    if (a < 0) != (b < 0):
        q = -q
    return a - q * b


def i32_lsr(a, n):
    return (a & 0xFFFFFFFF) >> n


# ---- Crc_32 (zd.ml:106-164), byte at a time: the slice-by-4 loop computes the same polynomial division ---------------

def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (0xEDB88320 ^ (c >> 1)) if c & 1 else (c >> 1)
        t.append(c)
    return t


_CRC_T = _crc_table()


def crc_32_string_update(c, s, start, length):
    for j in range(start, start + length):
        c = (c >> 8) ^ _CRC_T[(c ^ s[j]) & 0xFF]
    return c


def crc_32_string(s):
    return crc_32_string_update(0xFFFFFFFF, s, 0, len(s)) ^ 0xFFFFFFFF


# ---- Adler_32 (zd.ml:166-206) ---------------------------------------------------------------------------------------

ADLER_BASE = 65521


def adler_32_string_update(a, s, start, length):
    """a: the running value as an int32 (signed).  zd.ml:175-198"""
    s1 = a & 0xFFFF                     # a land 0xFFFFl
    s2 = i32_lsr(a, 16)                 # a lsr 16 (logical)
    mx = start + length - 1
    block_len = length % 5552           # Stdlib mod on a non-negative int
    while start <= mx:
        i = start
        block_max = start + block_len - 1
        while i <= block_max:           # the 8-fold unrolled loop and its tail are one loop here
            s1 = i32(s1 + s[i])
            s2 = i32(s2 + s1)
            i += 1
        s1 = i32_rem(s1, ADLER_BASE)
        s2 = i32_rem(s2, ADLER_BASE)
        start = i
        block_len = 5552
    return i32(i32(s2 << 16) + s1)


def adler_32_string(s):
    return adler_32_string_update(1, s, 0, len(s)) & 0xFFFFFFFF


# ---- format tables (zd.ml:237-313) ------------------------------------------------------------------------------------

LITLEN_SYM_MAX = 285
LITLEN_FIRST_LEN_SYM = 257
DIST_SYM_MAX = 29
CODELEN_SYM_MAX = 18


def _v(bits, base):
    return (base << 4) | bits


LENGTH_VALUE_OF_SYM_TABLE = [
    _v(0, 3), _v(0, 4), _v(0, 5), _v(0, 6), _v(0, 7), _v(0, 8), _v(0, 9), _v(0, 10),
    _v(1, 11), _v(1, 13), _v(1, 15), _v(1, 17), _v(2, 19), _v(2, 23), _v(2, 27), _v(2, 31),
    _v(3, 35), _v(3, 43), _v(3, 51), _v(3, 59), _v(4, 67), _v(4, 83), _v(4, 99), _v(4, 115),
    _v(5, 131), _v(5, 163), _v(5, 195), _v(5, 227), _v(0, 258)]

DIST_VALUE_OF_SYM = [
    _v(0, 1), _v(0, 2), _v(0, 3), _v(0, 4), _v(1, 5), _v(1, 7), _v(2, 9), _v(2, 13),
    _v(3, 17), _v(3, 25), _v(4, 33), _v(4, 49), _v(5, 65), _v(5, 97), _v(6, 129), _v(6, 193),
    _v(7, 257), _v(7, 385), _v(8, 513), _v(8, 769), _v(9, 1025), _v(9, 1537), _v(10, 2049),
    _v(10, 3073), _v(11, 4097), _v(11, 6145), _v(12, 8193), _v(12, 12289), _v(13, 16385), _v(13, 24577)]

CODELEN_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def _length_value_to_sym():
    t = [0] * 259
    for i, v in enumerate(LENGTH_VALUE_OF_SYM_TABLE):   # later rows overwrite earlier ones (zd.ml:266)
        base, extra = v >> 4, v & 0xF
        for ln in range(base, base + (1 << extra)):
            t[ln] = 257 + i
    return t


def _dist_value_to_sym_table():
    t = [0] * 512
    for i, v in enumerate(DIST_VALUE_OF_SYM):
        base, extra = v >> 4, v & 0xF
        for d in range(base, base + (1 << extra)):
            t[d - 1 if d <= 256 else 256 + ((d - 1) >> 7)] = i
    return t


LENGTH_VALUE_TO_SYM = _length_value_to_sym()
DIST_VALUE_TO_SYM_TABLE = _dist_value_to_sym_table()


def dist_value_to_sym(dist):
    return DIST_VALUE_TO_SYM_TABLE[dist - 1 if dist <= 256 else 256 + ((dist - 1) >> 7)]


# ---- Huffman encoder (zd.ml:393-528) ---------------------------------------------------------------------------------

MAX_SYMBOL_COUNT = 288


class _Exit(Exception):
    pass


def lengths_of_freqs(e, freqs, max_sym, max_code_len, heap, freq_cap=65535, stats=None):
    """zd.ml:404-473.  `heap` is the encoder's scratch array (one-based heap in front, parent links behind)."""

    def heapdown(h, mx, i):
        while True:
            l = 2 * i
            r = l + 1
            if l > mx:
                return
            k = l if r > mx else (l if h[l] < h[r] else r)
            if h[i] > h[k]:
                h[i], h[k] = h[k], h[i]
                i = k
            else:
                return

    mx = 0
    for sym in range(max_sym + 1):
        f = freqs[sym]
        if f != 0:
            if f > freq_cap:
                f = freq_cap
            mx += 1
            heap[mx] = (f << 10) | (max_sym + 1 + mx)
    for i in range(mx // 2, 0, -1):
        heapdown(heap, mx, i)
    if mx < 2:
        for sym in range(max_sym + 1):
            e[sym] = 0 if freqs[sym] == 0 else 1
        return
    try:
        m = mx
        while m > 1:
            new_max = m - 1
            p = heap[1]
            heap[1] = heap[m]
            heapdown(heap, new_max, 1)
            q = heap[1]
            nlink = m
            heap[1] = (((p >> 10) + (q >> 10)) << 10) | nlink
            heap[p & 0x3FF] = nlink
            heap[q & 0x3FF] = nlink
            heapdown(heap, new_max, 1)
            m = new_max
        k = 0
        for sym in range(max_sym + 1):
            if freqs[sym] == 0:
                e[sym] = 0
            else:
                k += 1
                p = heap[max_sym + 1 + k]
                ln = 1
                while p != 2:
                    ln += 1
                    p = heap[p]
                if ln > max_code_len:
                    raise _Exit()
                e[sym] = ln
    except _Exit:
        if stats is not None:
            stats["huffman_retries"] = stats.get("huffman_retries", 0) + 1
        lengths_of_freqs(e, freqs, max_sym, max_code_len, heap, freq_cap // 2, stats)


def _reverse_16(b):
    b = ((b & 0xFF00) >> 8) | ((b & 0x00FF) << 8)
    b = ((b & 0xF0F0) >> 4) | ((b & 0x0F0F) << 4)
    b = ((b & 0xCCCC) >> 2) | ((b & 0x3333) << 2)
    b = ((b & 0xAAAA) >> 1) | ((b & 0x5555) << 1)
    return b


def init_with_lengths(e, max_sym):
    """zd.ml:477-506: e holds code lengths on entry, sym_info = (reversed code << 5) | length on exit"""
    count = [0] * 16
    code = [0] * 16
    for sym in range(max_sym + 1):
        count[e[sym] & 0x1F] += 1
    count[0] = 0
    for ln in range(1, 16):
        code[ln] = (code[ln - 1] + count[ln - 1]) << 1
    for sym in range(max_sym + 1):
        ln = e[sym] & 0x1F
        if ln != 0:
            c = code[ln]
            e[sym] = ((_reverse_16(c) >> (16 - ln)) << 5) | ln
            code[ln] = c + 1


def _fixed_litlen_encoder():
    e = [0] * MAX_SYMBOL_COUNT
    for i in range(0, 144):
        e[i] = 8
    for i in range(144, 256):
        e[i] = 9
    for i in range(256, 280):
        e[i] = 7
    for i in range(280, 288):
        e[i] = 8
    init_with_lengths(e, 287)
    return e


def _fixed_dist_encoder():
    e = [0] * MAX_SYMBOL_COUNT
    for i in range(32):
        e[i] = 5
    init_with_lengths(e, 31)
    return e


FIXED_LITLEN_ENCODER = _fixed_litlen_encoder()
FIXED_DIST_ENCODER = _fixed_dist_encoder()

# ---- the encoder record (zd.ml:742-847) -------------------------------------------------------------------------------

MAX_BLOCK_SRC_LEN = 65534
LEVEL_PARAMS = {"none": (0, 0), "fast": (4, 4), "default": (8, 128), "best": (32, 4096)}   # good_match, max_chain_len
NO_POS = -1
WINDOW = 32768
MIN_MATCH, MAX_MATCH, MAX_DIST = 4, 258, 32768
NOP, CRC_32, ADLER_32 = 0, 1, 2


class Encoder:
    def __init__(self, src, level="best", crc_op=NOP):
        self.level = level
        self.src = src
        self.src_start = 0
        self.src_len = len(src)
        self.dst = bytearray()
        self.dst_bits = 0
        self.dst_bits_len = 0
        self.block_syms = [0] * (MAX_BLOCK_SRC_LEN + 1)
        self.block_syms_len = 0
        self.block_src_start = 0
        self.block_src_len = 0
        self.litlen_sym_freqs = [0] * (LITLEN_SYM_MAX + 1)
        self.dist_sym_freqs = [0] * (DIST_SYM_MAX + 1)
        self.codelen_syms = [0] * (LITLEN_SYM_MAX + DIST_SYM_MAX + 2)
        self.codelen_syms_len = 0
        self.codelen_sym_freqs = [0] * (CODELEN_SYM_MAX + 1)      # created once, never cleared (zd.ml:827, 849-854)
        self.good_match, self.max_chain_len = LEVEL_PARAMS[level]
        self.hash_head = [NO_POS] * (1 << 15)
        self.hash_prev = [0] * WINDOW
        self.dyn_litlen = [0] * MAX_SYMBOL_COUNT
        self.dyn_dist = [0] * MAX_SYMBOL_COUNT
        self.dyn_codelen = [0] * MAX_SYMBOL_COUNT
        self.hlit = self.hdist = self.hclen = 0
        self.crc_op = crc_op
        self.crc = {NOP: 0, CRC_32: 0xFFFFFFFF, ADLER_32: 1}[crc_op]
        self.scratch_heap = [0] * (MAX_SYMBOL_COUNT * 2 + 1)
        self.stats = {"blocks": [], "block_src_lens": []}


def new_block(e):
    e.block_syms_len = 0
    e.block_src_start += e.block_src_len
    e.block_src_len = 0
    for i in range(len(e.litlen_sym_freqs)):
        e.litlen_sym_freqs[i] = 0
    for i in range(len(e.dist_sym_freqs)):
        e.dist_sym_freqs[i] = 0


def flush(e):
    if e.dst_bits_len > 0:
        e.dst.append(e.dst_bits & 0xFF)
        e.dst_bits = 0
        e.dst_bits_len = 0


def write_bits(e, v, count):
    e.dst_bits = (v << e.dst_bits_len) | e.dst_bits
    e.dst_bits_len += count
    while e.dst_bits_len >= 8:
        e.dst.append(e.dst_bits & 0xFF)
        e.dst_bits >>= 8
        e.dst_bits_len -= 8


def write_non_compressed_block(e, final):
    ln = e.block_src_len
    write_bits(e, 0b001 if final else 0b000, 3)
    flush(e)
    e.dst += struct.pack("<H", ln & 0xFFFF)
    e.dst += struct.pack("<H", (~ln) & 0xFFFF)
    e.dst += e.src[e.block_src_start:e.block_src_start + ln]


def write_block_symbols(e, huffman_litlen, huffman_dist):
    for i in range(e.block_syms_len):
        bref = e.block_syms[i]
        dist, ln = bref >> 9, bref & 0x1FF
        if dist == 0:
            info = huffman_litlen[ln]
            write_bits(e, info >> 5, info & 0x1F)
        else:
            litlen_sym = LENGTH_VALUE_TO_SYM[ln]
            info = huffman_litlen[litlen_sym]
            bits, count = info >> 5, info & 0x1F
            lv = LENGTH_VALUE_OF_SYM_TABLE[litlen_sym - LITLEN_FIRST_LEN_SYM]
            extra_bits = ln - (lv >> 4)
            write_bits(e, (extra_bits << count) | bits, count + (lv & 0xF))
            dist_sym = dist_value_to_sym(dist)
            info = huffman_dist[dist_sym]
            bits, count = info >> 5, info & 0x1F
            dv = DIST_VALUE_OF_SYM[dist_sym]
            extra_bits = dist - (dv >> 4)
            write_bits(e, (extra_bits << count) | bits, count + (dv & 0xF))


def write_fixed_huffman_block(e, final):
    write_bits(e, 0b011 if final else 0b010, 3)
    write_block_symbols(e, FIXED_LITLEN_ENCODER, FIXED_DIST_ENCODER)


def write_dynamic_huffman_block(e, final):
    write_bits(e, 0b101 if final else 0b100, 3)
    write_bits(e, e.hlit, 5)
    write_bits(e, e.hdist, 5)
    write_bits(e, e.hclen, 4)
    for o in range(e.hclen + 4):
        write_bits(e, e.dyn_codelen[CODELEN_ORDER[o]] & 0x1F, 3)
    for l in range(e.codelen_syms_len):
        symref = e.codelen_syms[l]
        sym = symref & 0xFF
        info = e.dyn_codelen[sym]
        bits, count = info >> 5, info & 0x1F
        if sym <= 15:
            write_bits(e, bits, count)
        else:
            repeat_bits = symref >> 8
            rb = {16: 2, 17: 3, 18: 7}[sym]
            write_bits(e, (repeat_bits << count) | bits, count + rb)
    write_block_symbols(e, e.dyn_litlen, e.dyn_dist)


def huffman_init_with_freqs(e, huff, freqs, max_sym, max_code_len):
    lengths_of_freqs(huff, freqs, max_sym, max_code_len, e.scratch_heap, stats=e.stats)
    init_with_lengths(huff, max_sym)


def make_dynamic_huffman(e):
    huffman_init_with_freqs(e, e.dyn_litlen, e.litlen_sym_freqs, LITLEN_SYM_MAX, 15)
    huffman_init_with_freqs(e, e.dyn_dist, e.dist_sym_freqs, DIST_SYM_MAX, 15)


def make_dynamic_huffman_encoding(e):
    def code_length_count(h, max_sym):
        sym = max_sym
        while sym >= 0 and (h[sym] & 0x1F) == 0:
            sym -= 1
        return sym + 1

    litlen_count = code_length_count(e.dyn_litlen, LITLEN_SYM_MAX)
    dist_count = code_length_count(e.dyn_dist, DIST_SYM_MAX)
    if dist_count == 0:                         # zd.ml:974-979: symbol 0 gets a codeword of length 1, code 0
        e.dyn_dist[0] = (0 << 5) | 1
        dist_count = 1
    e.hlit = litlen_count - 257
    e.hdist = dist_count - 1
    l = e.codelen_syms
    for i in range(litlen_count):
        l[i] = e.dyn_litlen[i] & 0x1F
    for i in range(dist_count):
        l[litlen_count + i] = e.dyn_dist[i] & 0x1F
    length_count = litlen_count + dist_count

    # compute_codelen_syms (zd.ml:989-1030): reads the lengths from and writes the symbols to the same array
    lengths = e.codelen_syms
    len_max = length_count - 1
    i = k = 0
    while i <= len_max:
        cur = lengths[i]
        if cur == 0:
            mx = min(len_max, i + 138 - 1)
            j = i + 1
            while j <= mx and lengths[j] == 0:
                j += 1
            zcount = j - i
            if zcount < 3:
                e.codelen_syms[k] = 0
                e.codelen_sym_freqs[0] += 1
                nxt = i + 1
            elif zcount <= 10:
                e.codelen_syms[k] = ((zcount - 3) << 8) | 17
                e.codelen_sym_freqs[17] += 1
                nxt = j
            else:
                e.codelen_syms[k] = ((zcount - 11) << 8) | 18
                e.codelen_sym_freqs[18] += 1
                nxt = j
            k += 1
            i = nxt
        else:
            sym = cur
            e.codelen_syms[k] = sym
            e.codelen_sym_freqs[sym] += 1
            mx = min(len_max, i + 6)
            j = i + 1
            while j <= mx and lengths[j] == sym:
                j += 1
            scount = j - i
            if scount <= 3:
                k += 1
                i += 1
            else:
                e.codelen_syms[k + 1] = ((scount - 3 - 1) << 8) | 16
                e.codelen_sym_freqs[16] += 1
                k += 2
                i = j
    e.codelen_syms_len = k
    huffman_init_with_freqs(e, e.dyn_codelen, e.codelen_sym_freqs, CODELEN_SYM_MAX, 7)
    o = CODELEN_SYM_MAX
    while o > 0 and (e.dyn_codelen[CODELEN_ORDER[o]] & 0x1F) == 0:
        o -= 1
    e.hclen = (o + 1) - 4


def bit_length_of_non_compressed_block(e):
    alignment_loss = 8 - ((e.dst_bits_len + 3) % 8)
    return 3 + alignment_loss + (4 + e.block_src_len) * 8


def bit_length_of_block_symbols(e, hlitlen, hdist):
    acc = 0
    for sym in range(LITLEN_SYM_MAX + 1):
        code_length = hlitlen[sym] & 0x1F
        extra = 0 if sym < LITLEN_FIRST_LEN_SYM else (LENGTH_VALUE_OF_SYM_TABLE[sym - LITLEN_FIRST_LEN_SYM] & 0xF)
        acc += e.litlen_sym_freqs[sym] * (code_length + extra)
    for sym in range(DIST_SYM_MAX + 1):
        code_length = hdist[sym] & 0x1F
        acc += e.dist_sym_freqs[sym] * (code_length + (DIST_VALUE_OF_SYM[sym] & 0xF))
    return acc


def bit_length_of_dynamic_huffman_block(e):
    acc = 3 + 5 + 5 + 4 + 3 * (e.hclen + 4)
    for sym in range(CODELEN_SYM_MAX + 1):
        ln = e.dyn_codelen[sym] & 0x1F
        rb = {16: 2, 17: 3, 18: 7}.get(sym, 0)
        acc += e.codelen_sym_freqs[sym] * (ln + rb)
    return acc + bit_length_of_block_symbols(e, e.dyn_litlen, e.dyn_dist)


def deflated_block_src_crc(e):
    if e.crc_op == ADLER_32:
        e.crc = adler_32_string_update(e.crc, e.src, e.block_src_start, e.block_src_len)
    elif e.crc_op == CRC_32:
        e.crc = crc_32_string_update(e.crc, e.src, e.block_src_start, e.block_src_len)


def write_block(e, final):
    deflated_block_src_crc(e)
    e.stats["block_src_lens"].append(e.block_src_len)
    e.block_syms[e.block_syms_len] = 256
    e.block_syms_len += 1
    e.litlen_sym_freqs[256] = 1
    make_dynamic_huffman(e)
    make_dynamic_huffman_encoding(e)
    nlen = bit_length_of_non_compressed_block(e)
    flen = 3 + bit_length_of_block_symbols(e, FIXED_LITLEN_ENCODER, FIXED_DIST_ENCODER)
    dlen = bit_length_of_dynamic_huffman_block(e)
    if nlen <= dlen and nlen <= flen:
        e.stats["blocks"].append("none")
        write_non_compressed_block(e, final)
    elif flen <= dlen:
        e.stats["blocks"].append("fixed")
        write_fixed_huffman_block(e, final)
    else:
        e.stats["blocks"].append("dynamic")
        write_dynamic_huffman_block(e, final)


def write_all_non_compressed(e):
    src_max = e.src_start + e.src_len - 1
    while True:
        start = e.block_src_start
        block_max = min(src_max, start + MAX_BLOCK_SRC_LEN - 1)
        ln = block_max - start + 1
        final = block_max == src_max
        e.block_src_len = ln
        deflated_block_src_crc(e)
        e.stats["blocks"].append("none")
        e.stats["block_src_lens"].append(ln)
        write_non_compressed_block(e, final)
        if final:
            return
        e.block_src_start = start + ln


def write_block_symbol(e, sym, src_len):
    if e.block_src_len + src_len > MAX_BLOCK_SRC_LEN:
        write_block(e, False)
        new_block(e)
    e.block_syms[e.block_syms_len] = sym
    e.block_syms_len += 1
    e.block_src_len += src_len


def write_lit_symbol(e, byte):
    write_block_symbol(e, byte, 1)
    e.litlen_sym_freqs[byte] += 1


def write_backref_symbol(e, bref):
    ln = bref & 0x1FF
    write_block_symbol(e, bref, ln)
    e.litlen_sym_freqs[LENGTH_VALUE_TO_SYM[ln]] += 1
    e.dist_sym_freqs[dist_value_to_sym(bref >> 9)] += 1


# ---- Lz77 (zd.ml:1140-1245) -------------------------------------------------------------------------------------------

def hash4(s, i):
    v = s[i] | (s[i + 1] << 8) | (s[i + 2] << 16) | (s[i + 3] << 24)
    return ((v * 0x9E3779B1) & 0xFFFFFFFF) >> 17


def insert_hash(e, h, pos):
    e.hash_prev[pos % WINDOW] = e.hash_head[h]
    e.hash_head[h] = pos


def find_match_length(s, i, j, prev_match_len, max_match_len):
    """zd.ml:1154-1174: the bytes at offsets prev_match_len .. 0 backwards, then forward from prev_match_len + 1"""
    k = prev_match_len
    while k >= 0:
        if s[i + k] != s[j + k]:
            return 0
        k -= 1
    ln = prev_match_len + 1
    while ln < max_match_len and s[i + ln] == s[j + ln]:
        ln += 1
    return ln


def find_backref(e, pos, h, prev_match_len, max_match_len):
    if prev_match_len == 0:
        prev_match_len = MIN_MATCH - 1
    if prev_match_len >= max_match_len:
        return 0
    chain_steps = e.max_chain_len // 4 if prev_match_len >= e.good_match else e.max_chain_len
    s = e.src
    prev = e.hash_prev
    i = e.hash_head[h]
    match_pos = NO_POS
    while True:
        if i == NO_POS or chain_steps == 0 or pos - i > MAX_DIST:
            return 0 if match_pos == NO_POS else ((pos - match_pos) << 9) | prev_match_len
        chain_steps -= 1
        # the first byte find_match_length looks at, tested here so that the common miss costs one comparison
        if s[i + prev_match_len] == s[pos + prev_match_len]:
            ln = find_match_length(s, i, pos, prev_match_len, max_match_len)
            if ln == max_match_len:
                return ((pos - i) << 9) | ln
            if ln != 0:
                match_pos = i
                prev_match_len = ln
        i = prev[i % WINDOW]


def compress(e):
    if e.level == "none":
        write_all_non_compressed(e)
        return
    s = e.src
    max_pos = e.src_len - MIN_MATCH
    i = e.src_start
    prev_backref = 0
    while True:
        prev_match_len = prev_backref & 0x1FF
        if i > max_pos:
            if prev_match_len != 0:
                write_backref_symbol(e, prev_backref)
                i = max_pos + prev_match_len
            for k in range(i, e.src_len):
                write_lit_symbol(e, s[k])
            write_block(e, True)
            flush(e)
            return
        h = hash4(s, i)
        max_match_len = min(MAX_MATCH, e.src_len - i)
        bref = find_backref(e, i, h, prev_match_len, max_match_len)
        match_len = bref & 0x1FF
        insert_hash(e, h, i)
        if prev_match_len != 0 and prev_match_len > match_len:
            write_backref_symbol(e, prev_backref)
            nxt = (i - 1) + prev_match_len
            last = min(nxt - 1, max_pos)
            for j in range(i + 1, last + 1):
                insert_hash(e, hash4(s, j), j)
            i = nxt
            prev_backref = 0
        elif match_len == 0:
            write_lit_symbol(e, s[i])
            i += 1
            prev_backref = 0
        else:
            if prev_match_len != 0:
                write_lit_symbol(e, s[i - 1])
            i += 1
            prev_backref = bref


def crc_and_deflate(src, level="best", crc_op=NOP):
    """-> (checksum as an unsigned 32-bit number, compressed bytes, stats)   zd.ml:1247-1251"""
    e = Encoder(bytes(src), level, crc_op)
    compress(e)
    crc = {NOP: 0, CRC_32: e.crc ^ 0xFFFFFFFF, ADLER_32: e.crc & 0xFFFFFFFF}[crc_op]
    return crc & 0xFFFFFFFF, bytes(e.dst), e.stats


def zlib_compress(src, level="best"):
    """zd.ml:1262-1277"""
    e = Encoder(bytes(src), level, ADLER_32)
    cmf = (7 << 4) | 8
    flevel = {"none": 0, "fast": 1, "default": 2, "best": 3}[level]
    header = (cmf << 8) | (flevel << 6)
    flg = (header + 31 - (header % 31)) & 0xFF
    e.dst.append(cmf)
    e.dst.append(flg)
    compress(e)
    crc = e.crc & 0xFFFFFFFF
    e.dst += struct.pack(">I", crc)
    return crc, bytes(e.dst), e.stats
