/*
 * oo_encoding.c -- ORACLE (test infrastructure only): ORC chunk framing + value decoders.
 *
 * Plain-C, streaming, single threaded restatement of the reference's behaviour:
 *   compression.rs:113-123,238-347   chunk header, DecompressorIter, Decompressor: Read
 *   encoding/util.rs:26-38           read_u8 / try_read_u8
 *   encoding/rle.rs:62-107           GenericRle::decode (leftover queue)
 *   encoding/integer/util.rs         read_ints, width tables, varint, zigzag, signed-msb
 *   encoding/integer/mod.rs:218-318  NInt (i16/i32/i64) checked arithmetic
 *   encoding/integer/rle_v2/*        SHORT_REPEAT / DIRECT / PATCHED_BASE / DELTA
 *   encoding/integer/rle_v1.rs       RLE v1
 *   encoding/byte.rs:228-247         byte RLE
 *   encoding/boolean.rs:33-114       boolean (MSB-first bits over byte RLE)
 *   encoding/decimal.rs:28-52        unbounded zigzag varint -> i128
 *   encoding/timestamp.rs:121-192    timestamp combine
 * Where the reference panics on malformed input (index out of bounds, usize underflow) this
 * oracle returns OO_OUT_OF_SPEC; DESIGN.md lists those places.
 */
#include <stdlib.h>
#include <string.h>

#include "orc_oracle.h"

void oo_free(void* p) { free(p); }

/* ---------------------------------------------------------------------------------------- */
/* compression.rs:113-123                                                                     */
uint32_t oo_decode_chunk_header(const uint8_t b[3], int* is_original) {
  uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
  *is_original = (int)(v & 1);
  return v >> 1;
}

static long codec_block(int kind, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  switch (kind) {
    case OO_COMP_ZLIB: return oo_inflate_raw(src, n, dst, cap);
    case OO_COMP_SNAPPY: return oo_snappy_raw(src, n, dst, cap);
    case OO_COMP_LZ4: return oo_lz4_block(src, n, dst, cap);
    case OO_COMP_LZO: return oo_lzo1x(src, n, dst, cap);
    case OO_COMP_ZSTD: return oo_zstd_frame(src, n, dst, cap);
    default: return -1;
  }
}

/* std::io::Read over the chunked stream (compression.rs:287-347). */
struct oo_reader {
  const uint8_t* src;
  size_t n, pos;   /* compressed cursor                      */
  int kind;
  size_t block_size;
  uint8_t* cur;    /* current plain chunk (owned iff cur_owned) */
  const uint8_t* curp;
  size_t cur_len, cur_off;
  int cur_owned;
  int status;      /* sticky decode error                    */
};

oo_reader* oo_reader_new(const uint8_t* src, size_t n, int kind, size_t block_size) {
  oo_reader* r = (oo_reader*)calloc(1, sizeof(*r));
  r->src = src;
  r->n = n;
  r->kind = kind;
  r->block_size = block_size ? block_size : 262144;
  return r;
}
void oo_reader_free(oo_reader* r) {
  if (!r) return;
  if (r->cur_owned) free(r->cur);
  free(r);
}

/* DecompressorIter::advance (compression.rs:244-275); returns 0 at end of stream */
static int reader_advance(oo_reader* r) {
  if (r->cur_owned) free(r->cur);
  r->cur = NULL;
  r->cur_owned = 0;
  r->curp = NULL;
  r->cur_len = r->cur_off = 0;
  if (r->pos >= r->n) return 0;
  if (r->kind == OO_COMP_NONE) {
    r->curp = r->src + r->pos;
    r->cur_len = r->n - r->pos;
    r->pos = r->n;
    return 1;
  }
  if (r->pos + 3 > r->n) { /* reference: split_to panics on a truncated header */
    r->status = OO_IO_ERROR;
    r->pos = r->n;
    return 0;
  }
  int orig;
  uint32_t len = oo_decode_chunk_header(r->src + r->pos, &orig);
  r->pos += 3;
  if (r->pos + len > r->n) {
    r->status = OO_IO_ERROR;
    r->pos = r->n;
    return 0;
  }
  if (orig) {
    r->curp = r->src + r->pos;
    r->cur_len = len;
  } else {
    /* lz4_flex is handed max_decompressed_block_size; the other crates grow the scratch
     * buffer as needed, so give them head-room and only LZ4 the hard limit. */
    size_t cap = r->kind == OO_COMP_LZ4 ? r->block_size : (r->block_size > (1u << 22) ? r->block_size : (1u << 22));
    r->cur = (uint8_t*)malloc(cap + 16);
    r->cur_owned = 1;
    long got = codec_block(r->kind, r->src + r->pos, len, r->cur, cap);
    if (got < 0) {
      r->status = OO_BUILD_DECODER;
      r->pos = r->n;
      return 0;
    }
    r->curp = r->cur;
    r->cur_len = (size_t)got;
  }
  r->pos += len;
  return 1;
}

/* read exactly n bytes; returns number of bytes read (short on EOF) */
static size_t reader_read(oo_reader* r, uint8_t* buf, size_t n) {
  size_t got = 0;
  while (got < n) {
    if (r->cur_off == r->cur_len) {
      if (!reader_advance(r)) break;
      continue;
    }
    size_t take = r->cur_len - r->cur_off;
    if (take > n - got) take = n - got;
    if (buf) memcpy(buf + got, r->curp + r->cur_off, take);
    r->cur_off += take;
    got += take;
  }
  return got;
}

/* encoding/util.rs:26-38 */
static int read_u8(oo_reader* r, uint8_t* b) { return reader_read(r, b, 1) == 1 ? OO_OK : (r->status ? r->status : OO_IO_ERROR); }
/* returns 1 byte read, 0 EOF, <0 error */
static int try_read_u8(oo_reader* r, uint8_t* b) {
  if (reader_read(r, b, 1) == 1) return 1;
  return r->status ? -r->status : 0;
}

int oo_stream_decompress(const uint8_t* src, size_t n, int kind, size_t block_size, uint8_t** out, size_t* out_len) {
  oo_reader* r = oo_reader_new(src, n, kind, block_size);
  size_t cap = n * 4 + 1024, len = 0;
  uint8_t* buf = (uint8_t*)malloc(cap);
  for (;;) {
    if (len == cap) {
      cap *= 2;
      buf = (uint8_t*)realloc(buf, cap);
    }
    size_t got = reader_read(r, buf + len, cap - len);
    len += got;
    if (got == 0) break;
  }
  int st = r->status;
  oo_reader_free(r);
  *out = buf;
  *out_len = len;
  return st;
}

/* ---------------------------------------------------------------------------------------- */
/* NInt helpers: values are carried as sign-extended int64_t; nbits in {16,32,64}.            */
static int64_t trunc_n(int64_t v, int nbits) {
  if (nbits == 64) return v;
  if (nbits == 32) return (int64_t)(int32_t)v;
  return (int64_t)(int16_t)v;
}
static int in_range_n(int64_t v, int nbits) { return trunc_n(v, nbits) == v; }

/* integer/util.rs:536-546 signed_zigzag_decode in N bits */
static int64_t zigzag_n(int64_t enc, int nbits) {
  uint64_t u = (uint64_t)enc;
  if (nbits < 64) u &= ((uint64_t)1 << nbits) - 1;
  uint64_t v = (u >> 1) ^ (uint64_t)(-(int64_t)(u & 1));
  return trunc_n((int64_t)v, nbits);
}

/* integer/util.rs:559-569 signed_msb_decode (i64) */
static int64_t signed_msb_decode(int64_t enc, int byte_size) {
  uint64_t mask = (uint64_t)1 << (byte_size * 8 - 1);
  uint64_t u = (uint64_t)enc;
  int positive = (u & mask) == 0;
  u &= ~mask;
  return positive ? (int64_t)u : (int64_t)(0 - u);
}

/* integer/util.rs:475-498 read_varint::<N> + zigzag (util.rs:522-527) */
int oo_read_varint(oo_reader* r, int nbits, int is_signed, int64_t* out) {
  uint64_t num = 0;
  unsigned offset = 0;
  for (;;) {
    uint8_t b;
    int st = read_u8(r, &b);
    if (st) return st;
    if (offset >= (unsigned)nbits) return OO_VARINT_TOO_LARGE; /* checked_shl fails */
    num |= ((uint64_t)(b & 0x7f)) << offset;                   /* bits past the top are dropped */
    offset += 7;
    if (!(b & 0x80)) break;
  }
  int64_t v = trunc_n((int64_t)num, nbits);
  *out = is_signed ? zigzag_n(v, nbits) : v;
  return OO_OK;
}

/* integer/mod.rs:154-168 read_big_endian::<N>(byte_size); byte_size > sizeof(N) panics there */
static int read_big_endian(oo_reader* r, int nbits, int byte_size, int64_t* out) {
  if (byte_size > nbits / 8) return OO_OUT_OF_SPEC;
  uint8_t b[8];
  if (reader_read(r, b, (size_t)byte_size) != (size_t)byte_size) return r->status ? r->status : OO_IO_ERROR;
  uint64_t v = 0;
  for (int i = 0; i < byte_size; i++) v = (v << 8) | b[i];
  *out = trunc_n((int64_t)v, nbits);
  return OO_OK;
}

/* integer/util.rs:370-384 */
static int decode_bit_width(int enc) {
  static const int tail[8] = {26, 28, 30, 32, 40, 48, 56, 64};
  return enc <= 23 ? enc + 1 : tail[enc - 24];
}
/* integer/util.rs:407-421 */
static int closest_fixed_bits(int n) {
  if (n == 0) return 1;
  if (n <= 24) return n;
  if (n <= 26) return 26;
  if (n <= 28) return 28;
  if (n <= 30) return 30;
  if (n <= 32) return 32;
  if (n <= 40) return 40;
  if (n <= 48) return 48;
  if (n <= 56) return 56;
  return 64;
}

/* integer/util.rs:44-218 read_ints::<N>: big-endian MSB-first bit packing, results kept in N bits */
static int read_ints(oo_reader* r, int64_t* out, size_t n, int width, int nbits) {
  if (width == 1 || width == 2 || width == 4) {
    size_t per = (size_t)(8 / width);
    uint8_t byte = 0;
    for (size_t i = 0; i < n; i++) {
      if (i % per == 0) {
        int st = read_u8(r, &byte);
        if (st) return st;
      }
      int shift = 8 - width * (int)(i % per + 1);
      out[i] = (byte >> shift) & ((1 << width) - 1);
    }
    return OO_OK;
  }
  if (width % 8 == 0) {
    for (size_t i = 0; i < n; i++) {
      int st = read_big_endian(r, nbits, width / 8, &out[i]);
      if (st) return st;
    }
    return OO_OK;
  }
  /* unrolled_unpack_unaligned: arithmetic in N, i.e. modulo 2^nbits */
  int bits_left = 0;
  uint64_t cur = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t result = 0;
    int to_read = width;
    while (to_read > bits_left) {
      result <<= bits_left;
      result |= cur & (((uint64_t)1 << bits_left) - 1);
      to_read -= bits_left;
      uint8_t b;
      int st = read_u8(r, &b);
      if (st) return st;
      cur = b;
      bits_left = 8;
    }
    if (to_read > 0) {
      result <<= to_read;
      bits_left -= to_read;
      result |= (cur >> bits_left) & (((uint64_t)1 << to_read) - 1);
    }
    out[i] = trunc_n((int64_t)result, nbits);
  }
  return OO_OK;
}

/* NInt::add_i64 / sub_i64 (integer/mod.rs:236-317): i64 checked op, then try_into N */
static int add_i64_n(int64_t a, int64_t d, int nbits, int64_t* out) {
  int64_t r;
  if (__builtin_add_overflow(a, d, &r)) return 0;
  if (!in_range_n(r, nbits)) return 0;
  *out = r;
  return 1;
}
static int sub_i64_n(int64_t a, int64_t d, int nbits, int64_t* out) {
  int64_t r;
  if (__builtin_sub_overflow(a, d, &r)) return 0;
  if (!in_range_n(r, nbits)) return 0;
  *out = r;
  return 1;
}

/* ---------------------------------------------------------------------------------------- */
#define MAX_RUN 512
struct oo_int_rle {
  oo_reader* r;
  int version, is_signed, nbits;
  int64_t buf[MAX_RUN + 8];
  size_t len, head;
};

oo_int_rle* oo_int_rle_new(oo_reader* r, int version, int is_signed, int nbits) {
  oo_int_rle* d = (oo_int_rle*)calloc(1, sizeof(*d));
  d->r = r;
  d->version = version;
  d->is_signed = is_signed;
  d->nbits = nbits;
  return d;
}
void oo_int_rle_free(oo_int_rle* d) { free(d); }

static int run_length(uint8_t h0, uint8_t h1) { return (((int)h0 & 1) << 8 | h1) + 1; } /* util.rs:37-40 */

/* rle_v2/short_repeat.rs:29-63 */
static int v2_short_repeat(oo_int_rle* d, uint8_t h) {
  int bw = ((h >> 3) & 7) + 1;
  if (d->nbits / 8 < bw) return OO_OUT_OF_SPEC;
  int n = (h & 7) + 3;
  int64_t v;
  int st = read_big_endian(d->r, d->nbits, bw, &v);
  if (st) return st;
  if (d->is_signed) v = zigzag_n(v, d->nbits);
  for (int i = 0; i < n; i++) d->buf[i] = v;
  d->len = (size_t)n;
  return OO_OK;
}

/* rle_v2/direct.rs:39-65 */
static int v2_direct(oo_int_rle* d, uint8_t h) {
  int width = decode_bit_width((h >> 1) & 31);
  if (d->nbits < width) return OO_OUT_OF_SPEC;
  uint8_t h1;
  int st = read_u8(d->r, &h1);
  if (st) return st;
  int n = run_length(h, h1);
  st = read_ints(d->r, d->buf, (size_t)n, width, d->nbits);
  if (st) return st;
  if (d->is_signed)
    for (int i = 0; i < n; i++) d->buf[i] = zigzag_n(d->buf[i], d->nbits);
  d->len = (size_t)n;
  return OO_OK;
}

/* rle_v2/patched_base.rs:38-151 */
static int v2_patched(oo_int_rle* d, uint8_t h) {
  int nb = d->nbits;
  int W = decode_bit_width((h >> 1) & 31);
  uint8_t b1, b2, b3;
  int st;
  if ((st = read_u8(d->r, &b1))) return st;
  int n = run_length(h, b1);
  if ((st = read_u8(d->r, &b2))) return st;
  if ((st = read_u8(d->r, &b3))) return st;
  int BW = ((b2 >> 5) & 7) + 1;
  int PW = decode_bit_width(b2 & 31);
  int PGW = ((b3 >> 5) & 7) + 1;
  if (PW + PGW > 64) return OO_OUT_OF_SPEC;
  int PL = b3 & 31;
  int64_t base;
  if ((st = read_big_endian(d->r, 64, BW, &base))) return st;
  if (d->is_signed) base = signed_msb_decode(base, BW);
  base = trunc_n(base, nb);
  if ((st = read_ints(d->r, d->buf, (size_t)n, W, nb))) return st;
  int64_t patches[32];
  int cw = closest_fixed_bits(PW + PGW);
  if ((st = read_ints(d->r, patches, (size_t)PL, cw, 64))) return st;
  if (PL == 0) return OO_OUT_OF_SPEC; /* reference: patches[0] index panic */
  uint64_t pmask = PW >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << PW) - 1);
  int pi = 0;
  uint64_t gap = (uint64_t)patches[pi] >> PW, patch = (uint64_t)patches[pi] & pmask;
  int64_t actual = 0;
  while (gap == 255 && patch == 0) {
    actual += 255;
    pi++;
    if (pi >= PL) return OO_OUT_OF_SPEC; /* reference: index panic */
    gap = (uint64_t)patches[pi] >> PW;
    patch = (uint64_t)patches[pi] & pmask;
  }
  actual += (int64_t)gap;
  for (int idx = 0; idx < n; idx++) {
    if ((int64_t)idx == actual) {
      if (W >= 64) return OO_OUT_OF_SPEC; /* checked_shl */
      int64_t pbits = trunc_n((int64_t)(patch << W), nb);
      int64_t pv = d->buf[idx] | pbits;
      d->buf[idx] = trunc_n((int64_t)((uint64_t)pv + (uint64_t)base), nb); /* wrapping_add in N */
      pi++;
      if (pi < PL) {
        gap = (uint64_t)patches[pi] >> PW;
        patch = (uint64_t)patches[pi] & pmask;
        actual = 0;
        while (gap == 255 && patch == 0) {
          actual += 255;
          pi++;
          if (pi >= PL) return OO_OUT_OF_SPEC;
          gap = (uint64_t)patches[pi] >> PW;
          patch = (uint64_t)patches[pi] & pmask;
        }
        actual += (int64_t)gap;
        actual += idx;
      }
    } else {
      int64_t r;
      if (__builtin_add_overflow(d->buf[idx], base, &r) || !in_range_n(r, nb)) return OO_OUT_OF_SPEC; /* checked_add in N */
      d->buf[idx] = r;
    }
  }
  d->len = (size_t)n;
  return OO_OK;
}

/* rle_v2/delta.rs:44-116 */
static int v2_delta(oo_int_rle* d, uint8_t h) {
  int nb = d->nbits;
  int enc = (h >> 1) & 31;
  int width = enc == 0 ? 0 : decode_bit_width(enc);
  uint8_t h1;
  int st;
  if ((st = read_u8(d->r, &h1))) return st;
  int n = run_length(h, h1);
  int64_t base, db;
  if ((st = oo_read_varint(d->r, nb, d->is_signed, &base))) return st;
  d->buf[0] = base;
  d->len = 1;
  if ((st = oo_read_varint(d->r, 64, 1, &db))) return st;
  int add = db > 0;
  int64_t mag = db < 0 ? (int64_t)(0 - (uint64_t)db) : db; /* abs(), wrapping for i64::MIN */
  if (width == 0) {
    int64_t acc = base;
    for (int i = 1; i < n; i++) {
      if (!(add ? add_i64_n(acc, mag, nb, &acc) : sub_i64_n(acc, mag, nb, &acc))) return OO_OUT_OF_SPEC;
      d->buf[d->len++] = acc;
    }
  } else {
    int64_t acc;
    if (!(add ? add_i64_n(base, mag, nb, &acc) : sub_i64_n(base, mag, nb, &acc))) return OO_OUT_OF_SPEC;
    d->buf[d->len++] = acc;
    if (n < 2) return OO_OUT_OF_SPEC; /* reference: `length - 2` underflows */
    int64_t deltas[MAX_RUN];
    if ((st = read_ints(d->r, deltas, (size_t)(n - 2), width, 64))) return st;
    for (int i = 0; i < n - 2; i++) {
      if (!(add ? add_i64_n(acc, deltas[i], nb, &acc) : sub_i64_n(acc, deltas[i], nb, &acc))) return OO_OUT_OF_SPEC;
      d->buf[d->len++] = acc;
    }
  }
  return OO_OK;
}

/* rle_v2/mod.rs:112-146 decode_batch */
static int v2_decode_batch(oo_int_rle* d) {
  d->head = 0;
  d->len = 0;
  uint8_t h;
  int t = try_read_u8(d->r, &h);
  if (t == 0) return OO_OUT_OF_SPEC; /* "not enough values to decode in RLE v2" */
  if (t < 0) return -t;
  int st;
  switch (h >> 6) {
    case 0: st = v2_short_repeat(d, h); break;
    case 1: st = v2_direct(d, h); break;
    case 2: st = v2_patched(d, h); break;
    default: st = v2_delta(d, h); break;
  }
  return st;
}

/* rle_v1.rs:54-68,90-159 */
static int v1_decode_batch(oo_int_rle* d) {
  d->head = 0;
  d->len = 0;
  int nb = d->nbits;
  uint8_t hb;
  int t = try_read_u8(d->r, &hb);
  if (t == 0) return OO_OUT_OF_SPEC; /* "not enough values to decode" */
  if (t < 0) return -t;
  int8_t h = (int8_t)hb;
  int st;
  if (h < 0) {
    int n = -(int)h;
    for (int i = 0; i < n; i++) {
      int64_t v;
      if ((st = oo_read_varint(d->r, nb, d->is_signed, &v))) return st;
      d->buf[d->len++] = v;
    }
    return OO_OK;
  }
  int n = (int)hb + 3;
  uint8_t db;
  if ((st = read_u8(d->r, &db))) return st;
  int8_t delta = (int8_t)db;
  int64_t base;
  if ((st = oo_read_varint(d->r, nb, d->is_signed, &base))) return st;
  d->buf[d->len++] = base;
  for (int i = 1; i < n; i++) {
    int64_t r;
    /* checked_add / checked_sub in N with |delta| */
    int64_t mag = delta < 0 ? -(int64_t)delta : (int64_t)delta;
    if (delta < 0) {
      if (__builtin_sub_overflow(base, mag, &r) || !in_range_n(r, nb)) return OO_OUT_OF_SPEC;
    } else {
      if (__builtin_add_overflow(base, mag, &r) || !in_range_n(r, nb)) return OO_OUT_OF_SPEC;
    }
    base = r;
    d->buf[d->len++] = base;
  }
  return OO_OK;
}

/* encoding/rle.rs:68-107 GenericRle::decode */
int oo_int_rle_decode(oo_int_rle* d, int64_t* out, size_t n) {
  size_t copied = 0;
  while (copied < n) {
    if (d->head == d->len) {
      int st = d->version == 1 ? v1_decode_batch(d) : v2_decode_batch(d);
      if (st) {
        d->head = d->len = 0; /* decoded_ints is left cleared / partial; nothing consumable */
        return st;
      }
    }
    size_t avail = d->len - d->head;
    size_t c = avail < n - copied ? avail : n - copied;
    memcpy(out + copied, d->buf + d->head, c * sizeof(int64_t));
    d->head += c;
    copied += c;
  }
  return OO_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* byte.rs:228-247 */
struct oo_byte_rle {
  oo_reader* r;
  uint8_t buf[256];
  size_t len, head;
};
oo_byte_rle* oo_byte_rle_new(oo_reader* r) {
  oo_byte_rle* d = (oo_byte_rle*)calloc(1, sizeof(*d));
  d->r = r;
  return d;
}
void oo_byte_rle_free(oo_byte_rle* d) { free(d); }

static int byte_decode_batch(oo_byte_rle* d) {
  d->head = d->len = 0;
  uint8_t h;
  int st = read_u8(d->r, &h); /* byte.rs:232 uses read_u8: EOF is an IoError here */
  if (st) return st;
  if (h < 0x80) {
    size_t n = (size_t)h + 3;
    uint8_t v;
    if ((st = read_u8(d->r, &v))) return st;
    memset(d->buf, v, n);
    d->len = n;
  } else {
    size_t n = 0x100 - (size_t)h;
    if (reader_read(d->r, d->buf, n) != n) return d->r->status ? d->r->status : OO_IO_ERROR;
    d->len = n;
  }
  return OO_OK;
}

int oo_byte_rle_decode(oo_byte_rle* d, int8_t* out, size_t n) {
  size_t copied = 0;
  while (copied < n) {
    if (d->head == d->len) {
      int st = byte_decode_batch(d);
      if (st) {
        d->head = d->len = 0;
        return st;
      }
    }
    size_t avail = d->len - d->head;
    size_t c = avail < n - copied ? avail : n - copied;
    memcpy(out + copied, d->buf + d->head, c);
    d->head += c;
    copied += c;
  }
  return OO_OK;
}

/* boolean.rs:33-114 */
struct oo_bool_dec {
  oo_byte_rle* br;
  uint8_t data;
  int bits;
};
oo_bool_dec* oo_bool_new(oo_reader* r) {
  oo_bool_dec* d = (oo_bool_dec*)calloc(1, sizeof(*d));
  d->br = oo_byte_rle_new(r);
  return d;
}
void oo_bool_free(oo_bool_dec* d) {
  if (!d) return;
  oo_byte_rle_free(d->br);
  free(d);
}
int oo_bool_decode(oo_bool_dec* d, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    if (d->bits == 0) {
      int8_t b;
      int st = oo_byte_rle_decode(d->br, &b, 1);
      if (st) return st;
      d->data = (uint8_t)b;
      d->bits = 8;
    }
    out[i] = (d->data & 0x80) != 0;
    d->data <<= 1;
    d->bits--;
  }
  return OO_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* encoding/decimal.rs:28-52: read_varint_zigzagged::<i128, _, SignedEncoding>                */
int oo_varint128_decode(oo_reader* r, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    unsigned __int128 num = 0;
    unsigned offset = 0;
    for (;;) {
      uint8_t b;
      int st = read_u8(r, &b);
      if (st) return st;
      if (offset >= 128) return OO_VARINT_TOO_LARGE;
      num |= ((unsigned __int128)(b & 0x7f)) << offset;
      offset += 7;
      if (!(b & 0x80)) break;
    }
    unsigned __int128 v = (num >> 1) ^ (unsigned __int128)(-(__int128)(num & 1));
    out[2 * i] = (uint64_t)v;
    out[2 * i + 1] = (uint64_t)(v >> 64);
  }
  return OO_OK;
}

/* encoding/timestamp.rs:121-192 */
int oo_decode_timestamp(int64_t base, int64_t seconds_since_orc_base, int64_t nanos_in, int unit, int64_t* out) {
  uint64_t nanos = (uint64_t)nanos_in;
  uint64_t zeros = nanos & 7;
  nanos >>= 3;
  if (zeros != 0) {
    uint64_t p = 1;
    for (uint64_t i = 0; i < zeros + 1; i++) p *= 10;
    nanos *= p; /* u64 multiply: wraps in release builds */
  }
  int64_t sse = (int64_t)((uint64_t)seconds_since_orc_base + (uint64_t)base);
  int64_t seconds = (sse < 0 && nanos > 999999) ? sse - 1 : sse;
  __int128 ns = (__int128)seconds * 1000000000 + (__int128)nanos;
  static const int64_t per[4] = {1000000000, 1000000, 1000, 1};
  __int128 p = per[unit];
  if (ns % p != 0) return OO_DECODE_TIMESTAMP;
  __int128 q = ns / p;
  if (q > (__int128)INT64_MAX || q < (__int128)INT64_MIN) return OO_DECODE_TIMESTAMP;
  *out = (int64_t)q;
  return OO_OK;
}

/* array_decoder/decimal.rs:138-166 (release-build wrapping semantics for pow / mul) */
void oo_fix_i128_scale(const uint64_t in[2], uint32_t fixed_scale, int32_t varying_scale, uint64_t out[2]) {
  __int128 v = (__int128)(((unsigned __int128)in[1] << 64) | in[0]);
  uint32_t vs = (uint32_t)varying_scale;
  if (fixed_scale < vs) {
    uint32_t k = vs - fixed_scale;
    unsigned __int128 f = 1;
    for (uint32_t i = 0; i < k && i < 200; i++) f *= 10;
    __int128 sf = (__int128)f;
    if (sf != 0) v = v / sf;
  } else if (fixed_scale > vs) {
    uint32_t k = fixed_scale - vs;
    unsigned __int128 f = 1;
    for (uint32_t i = 0; i < k && i < 200; i++) f *= 10;
    v = (__int128)((unsigned __int128)v * f);
  }
  out[0] = (uint64_t)(unsigned __int128)v;
  out[1] = (uint64_t)((unsigned __int128)v >> 64);
}

/* internal hook for oo_column.c (Read::read_exact / take().read_to_end on a stream) */
size_t oo__reader_read(oo_reader* r, uint8_t* buf, size_t n) { return reader_read(r, buf, n); }
/* sticky reader status (framing IoError / rejected block): raw byte readers report it instead of a plain short read */
int oo__reader_status(const oo_reader* r) { return r->status; }
