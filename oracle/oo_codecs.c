/*
 * oo_codecs.c -- ORACLE (test infrastructure only): block codecs of the ORC chunk layer.
 *
 * The reference delegates block decompression to un-vendored crates pinned only by semver
 * (Cargo.toml:41-49, no Cargo.lock): flate2 "1" (raw DEFLATE, compression.rs:142-149),
 * snap "1.1" (raw Snappy, :161-172), lz4_flex "0.11" (LZ4 block, :185-195), zstd "0.13"
 * (Zstandard frame, :151-159).  Their sources are absent from /root/reference, so this file
 * restates the PUBLISHED formats: RFC 1951 (DEFLATE), the Snappy format description
 * (google/snappy format_description.txt), the LZ4 block format description
 * (lz4/doc/lz4_Block_format.md) and RFC 8878 (Zstandard).  Pinned in tests against Python's
 * zlib and pyarrow.Codec and through the reference's compressed fixture files.
 */
#include <stdlib.h>
#include <string.h>

#include "orc_oracle.h"

/* ------------------------------------------------------------------------------------------
 * RFC 1951 raw DEFLATE
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* src;
  size_t n, pos;
  uint32_t bitbuf;
  int bitcnt;
  int err;
} bitrd;

static int br_bits(bitrd* b, int need) {
  uint32_t val = b->bitbuf;
  while (b->bitcnt < need) {
    if (b->pos >= b->n) {
      b->err = 1;
      return 0;
    }
    val |= (uint32_t)b->src[b->pos++] << b->bitcnt;
    b->bitcnt += 8;
  }
  b->bitbuf = need == 32 ? 0 : (val >> need);
  b->bitcnt -= need;
  return (int)(val & ((need == 32) ? 0xffffffffu : ((1u << need) - 1)));
}

typedef struct {
  uint16_t count[16];
  uint16_t symbol[320];
} huff;

static int huff_build(huff* h, const uint8_t* lens, int n) {
  uint16_t offs[16];
  memset(h->count, 0, sizeof(h->count));
  for (int i = 0; i < n; i++) h->count[lens[i]]++;
  if (h->count[0] == n) return 0; /* no codes: legal, decode will fail if used */
  int left = 1;
  for (int len = 1; len < 16; len++) {
    left <<= 1;
    left -= h->count[len];
    if (left < 0) return -1; /* over-subscribed */
  }
  offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
  for (int i = 0; i < n; i++)
    if (lens[i]) h->symbol[offs[lens[i]]++] = (uint16_t)i;
  return left; /* >0: incomplete */
}

static int huff_decode(bitrd* b, const huff* h) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len < 16; len++) {
    code |= br_bits(b, 1);
    if (b->err) return -1;
    int count = h->count[len];
    if (code - count < first) return h->symbol[index + (code - first)];
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint16_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static int inflate_codes(bitrd* b, uint8_t* dst, size_t cap, size_t* outp, const huff* lc, const huff* dc) {
  size_t out = *outp;
  for (;;) {
    int sym = huff_decode(b, lc);
    if (sym < 0) return -1;
    if (sym < 256) {
      if (out >= cap) return -1;
      dst[out++] = (uint8_t)sym;
    } else if (sym == 256) {
      break;
    } else {
      sym -= 257;
      if (sym >= 29) return -1;
      int len = LBASE[sym] + br_bits(b, LEXT[sym]);
      int ds = huff_decode(b, dc);
      if (ds < 0 || ds >= 30) return -1;
      size_t dist = (size_t)DBASE[ds] + (size_t)br_bits(b, DEXT[ds]);
      if (b->err || dist > out || out + (size_t)len > cap) return -1;
      for (int i = 0; i < len; i++, out++) dst[out] = dst[out - dist];
    }
  }
  *outp = out;
  return 0;
}

long oo_inflate_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  bitrd b = {src, n, 0, 0, 0, 0};
  size_t out = 0;
  int last;
  do {
    last = br_bits(&b, 1);
    int type = br_bits(&b, 2);
    if (b.err) return -1;
    if (type == 0) {
      b.bitbuf = 0;
      b.bitcnt = 0;
      if (b.pos + 4 > n) return -1;
      unsigned len = src[b.pos] | (src[b.pos + 1] << 8);
      unsigned nlen = src[b.pos + 2] | (src[b.pos + 3] << 8);
      b.pos += 4;
      if ((len ^ 0xffffu) != nlen) return -1;
      if (b.pos + len > n || out + len > cap) return -1;
      memcpy(dst + out, src + b.pos, len);
      b.pos += len;
      out += len;
    } else if (type == 1) {
      huff lc, dc;
      uint8_t lens[288];
      int i = 0;
      for (; i < 144; i++) lens[i] = 8;
      for (; i < 256; i++) lens[i] = 9;
      for (; i < 280; i++) lens[i] = 7;
      for (; i < 288; i++) lens[i] = 8;
      huff_build(&lc, lens, 288);
      for (i = 0; i < 30; i++) lens[i] = 5;
      huff_build(&dc, lens, 30);
      if (inflate_codes(&b, dst, cap, &out, &lc, &dc)) return -1;
    } else if (type == 2) {
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t lens[320];
      huff lc, dc;
      int nlen = br_bits(&b, 5) + 257, ndist = br_bits(&b, 5) + 1, ncode = br_bits(&b, 4) + 4;
      if (b.err || nlen > 286 || ndist > 30) return -1;
      int i;
      for (i = 0; i < ncode; i++) lens[order[i]] = (uint8_t)br_bits(&b, 3);
      for (; i < 19; i++) lens[order[i]] = 0;
      if (huff_build(&lc, lens, 19) != 0) return -1;
      i = 0;
      while (i < nlen + ndist) {
        int sym = huff_decode(&b, &lc);
        if (sym < 0) return -1;
        if (sym < 16) {
          lens[i++] = (uint8_t)sym;
        } else {
          int len = 0, rep;
          if (sym == 16) {
            if (i == 0) return -1;
            len = lens[i - 1];
            rep = 3 + br_bits(&b, 2);
          } else if (sym == 17) {
            rep = 3 + br_bits(&b, 3);
          } else {
            rep = 11 + br_bits(&b, 7);
          }
          if (b.err || i + rep > nlen + ndist) return -1;
          while (rep--) lens[i++] = (uint8_t)len;
        }
      }
      if (lens[256] == 0) return -1;
      int r = huff_build(&lc, lens, nlen);
      if (r < 0 || (r > 0 && nlen - lc.count[0] != 1)) return -1;
      r = huff_build(&dc, lens + nlen, ndist);
      if (r < 0 || (r > 0 && ndist - dc.count[0] != 1)) return -1;
      if (inflate_codes(&b, dst, cap, &out, &lc, &dc)) return -1;
    } else {
      return -1;
    }
  } while (!last);
  return (long)out;
}

/* ------------------------------------------------------------------------------------------
 * Snappy raw format
 * ---------------------------------------------------------------------------------------- */
long oo_snappy_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  size_t pos = 0, out = 0;
  uint64_t ulen = 0;
  int shift = 0;
  for (;;) {
    if (pos >= n || shift > 28) return -1;
    uint8_t c = src[pos++];
    ulen |= (uint64_t)(c & 0x7f) << shift;
    shift += 7;
    if (!(c & 0x80)) break;
  }
  if (ulen > cap) return -1;
  while (pos < n) {
    uint8_t tag = src[pos++];
    size_t len, off;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2);
        if (len >= 60) {
          int nb = (int)len - 59;
          if (pos + nb > n) return -1;
          len = 0;
          for (int i = 0; i < nb; i++) len |= (size_t)src[pos + i] << (8 * i);
          pos += nb;
        }
        len += 1;
        if (pos + len > n || out + len > ulen) return -1;
        memcpy(dst + out, src + pos, len);
        pos += len;
        out += len;
        continue;
      }
      case 1:
        if (pos + 1 > n) return -1;
        len = 4 + ((tag >> 2) & 7);
        off = ((size_t)(tag >> 5) << 8) | src[pos];
        pos += 1;
        break;
      case 2:
        if (pos + 2 > n) return -1;
        len = 1 + (tag >> 2);
        off = src[pos] | ((size_t)src[pos + 1] << 8);
        pos += 2;
        break;
      default:
        if (pos + 4 > n) return -1;
        len = 1 + (tag >> 2);
        off = src[pos] | ((size_t)src[pos + 1] << 8) | ((size_t)src[pos + 2] << 16) | ((size_t)src[pos + 3] << 24);
        pos += 4;
        break;
    }
    if (off == 0 || off > out || out + len > ulen) return -1;
    for (size_t i = 0; i < len; i++, out++) dst[out] = dst[out - off];
  }
  if (out != ulen) return -1;
  return (long)out;
}

/* ------------------------------------------------------------------------------------------
 * LZ4 block format
 * ---------------------------------------------------------------------------------------- */
long oo_lz4_block(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  size_t pos = 0, out = 0;
  if (n == 0) return -1;
  for (;;) {
    if (pos >= n) return -1;
    uint8_t tok = src[pos++];
    size_t lit = tok >> 4;
    if (lit == 15) {
      uint8_t c;
      do {
        if (pos >= n) return -1;
        c = src[pos++];
        lit += c;
      } while (c == 255);
    }
    if (pos + lit > n || out + lit > cap) return -1;
    memcpy(dst + out, src + pos, lit);
    pos += lit;
    out += lit;
    if (pos == n) break; /* last sequence: literals only */
    if (pos + 2 > n) return -1;
    size_t off = src[pos] | ((size_t)src[pos + 1] << 8);
    pos += 2;
    size_t ml = tok & 15;
    if (ml == 15) {
      uint8_t c;
      do {
        if (pos >= n) return -1;
        c = src[pos++];
        ml += c;
      } while (c == 255);
    }
    ml += 4;
    if (off == 0 || off > out || out + ml > cap) return -1;
    for (size_t i = 0; i < ml; i++, out++) dst[out] = dst[out - off];
  }
  return (long)out;
}

/* ------------------------------------------------------------------------------------------
 * LZO1X (compression.rs:174-183: lzokay_native::decompress_all; lzokay-native "0.1" is a port of lzokay's
 * decompress(), itself written from the format description in the Linux kernel, Documentation/staging/lzo.rst).
 * The crate is absent from /root/reference (Cargo.toml:44): the algorithm is restated from the format --
 *   first byte 18..21: 1..4 "state" literals follow, >= 22: (byte - 17) literals, state 4;
 *   instructions  1LLDDDSS / 01LDDDSS + H      M2: length (inst >> 5) + 1, distance (H << 3) + D + 1
 *                 001LLLLL [+ zero bytes + byte] + LE16   M3: length 2 + L (L = 0: 31 + 255 z + byte), distance (LE16 >> 2) + 1
 *                 0001HLLL [+ zero bytes + byte] + LE16   M4: length 2 + L (L = 0: 7 + 255 z + byte), distance 16384 + (H << 14) + (LE16 >> 2);
 *                                                          distance 16384 ends the stream
 *                 0000xxxx                                 by the state: 0 -> literal run of 3 + L (L = 0: 15 + 255 z + byte), state 4;
 *                                                          1..3 -> 2 bytes from (inst >> 2) + (H << 2) + 1; 4 -> 3 bytes from (inst >> 2) + (H << 2) + 2049
 *   S (two low bits of the instruction / of the LE16): that many literals follow the match and become the state.
 * Every failure of the crate (input overrun, output overrun, look-behind before the output, a stream that does not end with
 * the M4 marker, input left over) is an Err: BuildLzoDecoder.  Pinned by the reference's two LZO fixture files
 * (tests/golden/data/alltypes.lzo.orc, TestVectorOrcFile.testLzo.orc).
 * ---------------------------------------------------------------------------------------- */
long oo_lzo1x(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  size_t ip = 0, op = 0, state = 0, nstate = 0, lblen = 0, lbdist = 0;
  if (n < 3) return -1;
#define LZO_IN(k) do { if (n - ip < (size_t)(k)) return -1; } while (0)
#define LZO_OUT(k) do { if (cap - op < (size_t)(k)) return -1; } while (0)
#define LZO_ZEROS(z) do { size_t z0 = ip; while (ip < n && src[ip] == 0) ip++; (z) = ip - z0; if ((z) > ((~(size_t)0) / 255 - 2)) return -1; } while (0)
  if (src[ip] >= 22) {
    size_t len = (size_t)src[ip++] - 17;
    LZO_IN(len);
    LZO_OUT(len);
    memcpy(dst + op, src + ip, len);
    ip += len;
    op += len;
    state = 4;
  } else if (src[ip] >= 18) {
    nstate = (size_t)src[ip++] - 17;
    state = nstate;
    LZO_IN(nstate);
    LZO_OUT(nstate);
    memcpy(dst + op, src + ip, nstate);
    ip += nstate;
    op += nstate;
  }
  for (;;) {
    LZO_IN(1);
    const uint8_t inst = src[ip++];
    if (inst & 0xC0) {
      LZO_IN(1);
      lbdist = ((size_t)src[ip++] << 3) + ((inst >> 2) & 7) + 1;
      lblen = (size_t)(inst >> 5) + 1;
      nstate = inst & 3;
    } else if (inst & 0x20) {
      lblen = (size_t)(inst & 0x1f) + 2;
      if (lblen == 2) {
        size_t z;
        LZO_ZEROS(z);
        LZO_IN(1);
        lblen += z * 255 + 31 + src[ip++];
      }
      LZO_IN(2);
      nstate = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
      ip += 2;
      lbdist = (nstate >> 2) + 1;
      nstate &= 3;
    } else if (inst & 0x10) {
      lblen = (size_t)(inst & 7) + 2;
      if (lblen == 2) {
        size_t z;
        LZO_ZEROS(z);
        LZO_IN(1);
        lblen += z * 255 + 7 + src[ip++];
      }
      LZO_IN(2);
      nstate = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
      ip += 2;
      lbdist = (((size_t)inst & 8) << 11) + (nstate >> 2);
      nstate &= 3;
      if (lbdist == 0) break; /* stream finished */
      lbdist += 16384;
    } else if (state == 0) {
      size_t len = (size_t)inst + 3;
      if (len == 3) {
        size_t z;
        LZO_ZEROS(z);
        LZO_IN(1);
        len += z * 255 + 15 + src[ip++];
      }
      LZO_IN(len);
      LZO_OUT(len);
      memcpy(dst + op, src + ip, len);
      ip += len;
      op += len;
      state = 4;
      continue;
    } else if (state != 4) {
      LZO_IN(1);
      nstate = inst & 3;
      lbdist = (size_t)(inst >> 2) + ((size_t)src[ip++] << 2) + 1;
      lblen = 2;
    } else {
      LZO_IN(1);
      nstate = inst & 3;
      lbdist = (size_t)(inst >> 2) + ((size_t)src[ip++] << 2) + 2049;
      lblen = 3;
    }
    if (lbdist > op) return -1; /* look-behind before the start of the output */
    LZO_IN(nstate);
    LZO_OUT(lblen + nstate);
    for (size_t i = 0; i < lblen; i++, op++) dst[op] = dst[op - lbdist];
    state = nstate;
    memcpy(dst + op, src + ip, nstate);
    ip += nstate;
    op += nstate;
  }
#undef LZO_IN
#undef LZO_OUT
#undef LZO_ZEROS
  if (lblen != 3) return -1; /* the terminating M4 has length 3 */
  if (ip != n) return -1;    /* input not consumed / overrun */
  return (long)op;
}

/* ------------------------------------------------------------------------------------------
 * RFC 8878 Zstandard frame (no dictionary; the content checksum, when the frame flags one, is verified)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* p;
  long bits; /* number of unread bits; reading proceeds from bit (bits-1) downwards */
  int err;
} rbits;   /* backward bit stream */

static void rb_init(rbits* r, const uint8_t* p, size_t n) {
  r->p = p;
  r->err = 0;
  if (n == 0 || p[n - 1] == 0) {
    r->err = 1;
    r->bits = 0;
    return;
  }
  int hb = 7;
  while (!((p[n - 1] >> hb) & 1)) hb--;
  r->bits = (long)(n - 1) * 8 + hb;
}
/* read nb bits (<=57); bits below the start of the stream read as zero, flagged via bits<0 */
static uint64_t rb_read(rbits* r, int nb) {
  if (nb == 0) return 0;
  long start = r->bits - nb; /* lowest bit index */
  uint64_t v = 0;
  for (int i = nb - 1; i >= 0; i--) {
    long bi = start + i;
    uint64_t bit = 0;
    if (bi >= 0) bit = (r->p[bi >> 3] >> (bi & 7)) & 1;
    v = (v << 1) | bit;
  }
  r->bits = start;
  return v;
}

typedef struct {
  uint8_t sym;
  uint8_t nb;
  uint16_t base;
} fse_ent;
typedef struct {
  int log;
  fse_ent t[512];
} fse_tab;

static int hibit(uint32_t v) {
  int r = -1;
  while (v) {
    v >>= 1;
    r++;
  }
  return r;
}

static int fse_build(fse_tab* ft, const int16_t* norm, int nsym, int log) {
  int size = 1 << log;
  int high = size - 1;
  uint16_t next[256];
  ft->log = log;
  for (int s = 0; s < nsym; s++) {
    if (norm[s] == -1) {
      ft->t[high--].sym = (uint8_t)s;
      next[s] = 1;
    } else {
      next[s] = (uint16_t)norm[s];
    }
  }
  int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
  for (int s = 0; s < nsym; s++) {
    for (int i = 0; i < norm[s]; i++) {
      ft->t[pos].sym = (uint8_t)s;
      do {
        pos = (pos + step) & mask;
      } while (pos > high);
    }
  }
  if (pos != 0) return -1;
  for (int i = 0; i < size; i++) {
    int s = ft->t[i].sym;
    uint32_t ns = next[s]++;
    int nb = log - hibit(ns);
    ft->t[i].nb = (uint8_t)nb;
    ft->t[i].base = (uint16_t)((ns << nb) - size);
  }
  return 0;
}

/* forward-bitstream FSE table description; returns bytes consumed or -1 */
static long fse_read_ncount(const uint8_t* p, size_t n, int16_t* norm, int* nsym_io, int* log_out, int maxlog) {
  size_t pos = 0;
  uint64_t bb = 0;
  int bc = 0;
#define NEED(k)                                   \
  while (bc < (k)) {                              \
    uint64_t byte_ = pos < n ? p[pos] : 0;        \
    if (pos >= n + 8) return -1;                  \
    pos++;                                        \
    bb |= byte_ << bc;                            \
    bc += 8;                                      \
  }
  NEED(4);
  int log = (int)(bb & 15) + 5;
  bb >>= 4;
  bc -= 4;
  if (log > maxlog) return -1;
  int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1;
  int sym = 0, maxsym = *nsym_io;
  int prev0 = 0;
  while (remaining > 1 && sym < maxsym) {
    if (prev0) {
      for (;;) {
        NEED(2);
        int rep = (int)(bb & 3);
        bb >>= 2;
        bc -= 2;
        for (int i = 0; i < rep && sym < maxsym; i++) norm[sym++] = 0;
        if (rep != 3) break;
      }
      prev0 = 0;
      if (sym >= maxsym) break;
      continue;
    }
    int max = (2 * threshold - 1) - remaining;
    NEED(nbits);
    int count;
    if ((int)(bb & (uint64_t)(threshold - 1)) < max) {
      count = (int)(bb & (uint64_t)(threshold - 1));
      bb >>= (nbits - 1);
      bc -= (nbits - 1);
    } else {
      count = (int)(bb & (uint64_t)(2 * threshold - 1));
      if (count >= threshold) count -= max;
      bb >>= nbits;
      bc -= nbits;
    }
    count--;
    remaining -= count < 0 ? -count : count;
    norm[sym++] = (int16_t)count;
    prev0 = (count == 0);
    while (remaining < threshold) {
      nbits--;
      threshold >>= 1;
    }
  }
#undef NEED
  if (remaining != 1) return -1;
  *nsym_io = sym;
  *log_out = log;
  /* bytes consumed = ceil(bits consumed / 8) */
  size_t bits_used = pos * 8 - (size_t)bc;
  size_t used = (bits_used + 7) / 8;
  if (used > n) return -1;
  return (long)used;
}

typedef struct {
  int maxbits;
  uint8_t sym[2048];
  uint8_t nb[2048];
} huf_tab;

static int huf_build(huf_tab* h, const uint8_t* weights, int nw) {
  /* weights for nw symbols, last one implied */
  uint32_t sum = 0;
  for (int i = 0; i < nw; i++) {
    if (weights[i] > 11) return -1;
    if (weights[i]) sum += 1u << (weights[i] - 1);
  }
  if (sum == 0) return -1;
  int maxbits = hibit(sum) + 1;
  if (maxbits > 11) return -1;
  uint32_t left = (1u << maxbits) - sum;
  if (left & (left - 1)) return -1;
  uint8_t w[256];
  memcpy(w, weights, (size_t)nw);
  w[nw] = (uint8_t)(hibit(left) + 1);
  nw++;
  h->maxbits = maxbits;
  uint32_t rankstart[13] = {0};
  uint32_t cnt[13] = {0};
  for (int i = 0; i < nw; i++) cnt[w[i]]++;
  uint32_t pos = 0;
  for (int wt = 1; wt <= maxbits; wt++) {
    rankstart[wt] = pos;
    pos += cnt[wt] << (wt - 1);
  }
  if (pos != (1u << maxbits)) return -1;
  for (int s = 0; s < nw; s++) {
    if (!w[s]) continue;
    uint32_t len = 1u << (w[s] - 1);
    uint32_t st = rankstart[w[s]];
    for (uint32_t i = 0; i < len; i++) {
      h->sym[st + i] = (uint8_t)s;
      h->nb[st + i] = (uint8_t)(maxbits + 1 - w[s]);
    }
    rankstart[w[s]] += len;
  }
  return 0;
}

static int huf_decode_stream(const huf_tab* h, const uint8_t* p, size_t n, uint8_t* out, size_t outn) {
  rbits r;
  rb_init(&r, p, n);
  if (r.err) return -1;
  int mb = h->maxbits;
  uint32_t state = (uint32_t)rb_read(&r, mb);
  for (size_t i = 0; i < outn; i++) {
    out[i] = h->sym[state];
    int nb = h->nb[state];
    state = ((state << nb) & ((1u << mb) - 1)) | (uint32_t)rb_read(&r, nb);
  }
  /* all bits must be consumed exactly: after the last symbol we over-read by maxbits */
  if (r.bits != -(long)mb) return -1;
  return 0;
}

typedef struct {
  huf_tab huf;
  int huf_valid;
  fse_tab ll, of, ml;
  int ll_valid, of_valid, ml_valid;
  uint32_t rep[3];
} zctx;

static const int16_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

static long z_literals(zctx* z, const uint8_t* p, size_t n, uint8_t* lit, size_t litcap, size_t* litn) {
  if (n < 1) return -1;
  int type = p[0] & 3, sf = (p[0] >> 2) & 3;
  size_t regen, comp = 0, hdr;
  int streams = 1;
  if (type < 2) {
    if (sf == 0 || sf == 2) {
      regen = p[0] >> 3;
      hdr = 1;
    } else if (sf == 1) {
      if (n < 2) return -1;
      regen = (p[0] >> 4) | ((size_t)p[1] << 4);
      hdr = 2;
    } else {
      if (n < 3) return -1;
      regen = (p[0] >> 4) | ((size_t)p[1] << 4) | ((size_t)p[2] << 12);
      hdr = 3;
    }
    if (regen > litcap) return -1;
    if (type == 0) {
      if (hdr + regen > n) return -1;
      memcpy(lit, p + hdr, regen);
      *litn = regen;
      return (long)(hdr + regen);
    }
    if (hdr + 1 > n) return -1;
    memset(lit, p[hdr], regen);
    *litn = regen;
    return (long)(hdr + 1);
  }
  if (sf == 0 || sf == 1) {
    if (n < 3) return -1;
    uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16);
    regen = (v >> 4) & 0x3ff;
    comp = (v >> 14) & 0x3ff;
    hdr = 3;
    streams = sf == 0 ? 1 : 4;
  } else if (sf == 2) {
    if (n < 4) return -1;
    uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    regen = (v >> 4) & 0x3fff;
    comp = (v >> 18) & 0x3fff;
    hdr = 4;
    streams = 4;
  } else {
    if (n < 5) return -1;
    uint64_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint64_t)p[3] << 24) | ((uint64_t)p[4] << 32);
    regen = (v >> 4) & 0x3ffff;
    comp = (v >> 22) & 0x3ffff;
    hdr = 5;
    streams = 4;
  }
  if (regen > litcap || hdr + comp > n) return -1;
  const uint8_t* q = p + hdr;
  size_t qn = comp;
  if (type == 2) {
    /* Huffman tree description */
    if (qn < 1) return -1;
    uint8_t weights[256];
    int nw;
    int hb = q[0];
    size_t used;
    if (hb >= 128) {
      nw = hb - 127;
      size_t nbytes = (size_t)(nw + 1) / 2;
      if (1 + nbytes > qn) return -1;
      for (int i = 0; i < nw; i++) weights[i] = (i & 1) ? (q[1 + i / 2] & 15) : (q[1 + i / 2] >> 4);
      used = 1 + nbytes;
    } else {
      if (1 + (size_t)hb > qn) return -1;
      int16_t norm[256];
      int nsym = 256, log;
      long c = fse_read_ncount(q + 1, (size_t)hb, norm, &nsym, &log, 6);
      if (c < 0) return -1;
      fse_tab ft;
      if (fse_build(&ft, norm, nsym, log)) return -1;
      rbits r;
      rb_init(&r, q + 1 + c, (size_t)hb - (size_t)c);
      if (r.err) return -1;
      uint32_t s1 = (uint32_t)rb_read(&r, log), s2 = (uint32_t)rb_read(&r, log);
      nw = 0;
      for (;;) {
        if (nw >= 255) return -1;
        weights[nw++] = ft.t[s1].sym;
        if (r.bits < ft.t[s1].nb) { /* not enough bits to update s1 */
          if (nw >= 255) return -1;
          weights[nw++] = ft.t[s2].sym;
          break;
        }
        s1 = ft.t[s1].base + (uint32_t)rb_read(&r, ft.t[s1].nb);
        if (nw >= 255) return -1;
        weights[nw++] = ft.t[s2].sym;
        if (r.bits < ft.t[s2].nb) {
          if (nw >= 255) return -1;
          weights[nw++] = ft.t[s1].sym;
          break;
        }
        s2 = ft.t[s2].base + (uint32_t)rb_read(&r, ft.t[s2].nb);
      }
      used = 1 + (size_t)hb;
    }
    if (huf_build(&z->huf, weights, nw)) return -1;
    z->huf_valid = 1;
    q += used;
    qn -= used;
  } else if (!z->huf_valid) {
    return -1;
  }
  if (streams == 1) {
    if (huf_decode_stream(&z->huf, q, qn, lit, regen)) return -1;
  } else {
    if (qn < 6) return -1;
    size_t s1 = q[0] | (q[1] << 8), s2 = q[2] | (q[3] << 8), s3 = q[4] | (q[5] << 8);
    if (6 + s1 + s2 + s3 > qn) return -1;
    size_t s4 = qn - 6 - s1 - s2 - s3;
    size_t seg = (regen + 3) / 4;
    if (seg * 3 > regen) return -1;
    const uint8_t* b = q + 6;
    if (huf_decode_stream(&z->huf, b, s1, lit, seg)) return -1;
    if (huf_decode_stream(&z->huf, b + s1, s2, lit + seg, seg)) return -1;
    if (huf_decode_stream(&z->huf, b + s1 + s2, s3, lit + 2 * seg, seg)) return -1;
    if (huf_decode_stream(&z->huf, b + s1 + s2 + s3, s4, lit + 3 * seg, regen - 3 * seg)) return -1;
  }
  *litn = regen;
  return (long)(hdr + comp);
}

static long z_seq_table(fse_tab* ft, int* valid, int mode, const uint8_t* p, size_t n, const int16_t* def, int defn,
                        int deflog, int maxsym, int maxlog) {
  if (mode == 0) {
    fse_build(ft, def, defn, deflog);
    *valid = 1;
    return 0;
  }
  if (mode == 1) {
    if (n < 1) return -1;
    ft->log = 0;
    ft->t[0].sym = p[0];
    ft->t[0].nb = 0;
    ft->t[0].base = 0;
    *valid = 1;
    return 1;
  }
  if (mode == 2) {
    int16_t norm[64];
    int nsym = maxsym, log;
    long c = fse_read_ncount(p, n, norm, &nsym, &log, maxlog);
    if (c < 0) return -1;
    if (fse_build(ft, norm, nsym, log)) return -1;
    *valid = 1;
    return c;
  }
  return *valid ? 0 : -1;
}

static long z_block(zctx* z, const uint8_t* p, size_t n, uint8_t* dst, size_t cap, size_t out) {
  uint8_t* litbuf = (uint8_t*)malloc(128 * 1024 + 16);
  if (!litbuf) return -1;
  size_t litn = 0;
  long used = z_literals(z, p, n, litbuf, 128 * 1024, &litn);
  long ret = -1;
  if (used < 0) goto done;
  {
    const uint8_t* q = p + used;
    size_t qn = n - (size_t)used;
    if (qn < 1) goto done;
    size_t nseq;
    if (q[0] < 128) {
      nseq = q[0];
      q += 1;
      qn -= 1;
    } else if (q[0] < 255) {
      if (qn < 2) goto done;
      nseq = ((size_t)(q[0] - 128) << 8) + q[1];
      q += 2;
      qn -= 2;
    } else {
      if (qn < 3) goto done;
      nseq = (size_t)q[1] + ((size_t)q[2] << 8) + 0x7f00;
      q += 3;
      qn -= 3;
    }
    size_t lp = 0;
    if (nseq) {
      if (qn < 1) goto done;
      int modes = q[0];
      if (modes & 3) goto done;
      q++;
      qn--;
      long c = z_seq_table(&z->ll, &z->ll_valid, (modes >> 6) & 3, q, qn, LL_DEF, 36, 6, 36, 9);
      if (c < 0) goto done;
      q += c;
      qn -= (size_t)c;
      c = z_seq_table(&z->of, &z->of_valid, (modes >> 4) & 3, q, qn, OF_DEF, 29, 5, 32, 8);
      if (c < 0) goto done;
      q += c;
      qn -= (size_t)c;
      c = z_seq_table(&z->ml, &z->ml_valid, (modes >> 2) & 3, q, qn, ML_DEF, 53, 6, 53, 9);
      if (c < 0) goto done;
      q += c;
      qn -= (size_t)c;
      rbits r;
      rb_init(&r, q, qn);
      if (r.err) goto done;
      uint32_t sl = (uint32_t)rb_read(&r, z->ll.log);
      uint32_t so = (uint32_t)rb_read(&r, z->of.log);
      uint32_t sm = (uint32_t)rb_read(&r, z->ml.log);
      for (size_t i = 0; i < nseq; i++) {
        int oc = z->of.t[so].sym, mc = z->ml.t[sm].sym, lc = z->ll.t[sl].sym;
        if (oc > 31 || mc > 52 || lc > 35) goto done;
        uint64_t ofv = ((uint64_t)1 << oc) + rb_read(&r, oc);
        uint32_t mlen = ML_BASE[mc] + (uint32_t)rb_read(&r, ML_BITS[mc]);
        uint32_t llen = LL_BASE[lc] + (uint32_t)rb_read(&r, LL_BITS[lc]);
        if (r.bits < 0) goto done;
        uint64_t offset;
        if (ofv > 3) {
          offset = ofv - 3;
          z->rep[2] = z->rep[1];
          z->rep[1] = z->rep[0];
          z->rep[0] = (uint32_t)offset;
        } else {
          uint32_t idx = (uint32_t)ofv - 1 + (llen == 0 ? 1 : 0);
          if (idx == 0) {
            offset = z->rep[0];
          } else {
            offset = idx < 3 ? z->rep[idx] : z->rep[0] - 1;
            if (idx > 1) z->rep[2] = z->rep[1];
            z->rep[1] = z->rep[0];
            z->rep[0] = (uint32_t)offset;
          }
        }
        if (offset == 0) goto done;
        if (lp + llen > litn || out + llen + mlen > cap) goto done;
        memcpy(dst + out, litbuf + lp, llen);
        lp += llen;
        out += llen;
        if (offset > out) goto done;
        for (uint32_t k = 0; k < mlen; k++, out++) dst[out] = dst[out - offset];
        if (i + 1 < nseq) {
          /* update order: LL, ML, OF */
          sl = z->ll.t[sl].base + (uint32_t)rb_read(&r, z->ll.t[sl].nb);
          sm = z->ml.t[sm].base + (uint32_t)rb_read(&r, z->ml.t[sm].nb);
          so = z->of.t[so].base + (uint32_t)rb_read(&r, z->of.t[so].nb);
          if (r.bits < 0) goto done;
        }
      }
      if (r.bits != 0) goto done;
    }
    if (out + (litn - lp) > cap) goto done;
    memcpy(dst + out, litbuf + lp, litn - lp);
    out += litn - lp;
    ret = (long)out;
  }
done:
  free(litbuf);
  return ret;
}

/* XXH64 (seed 0) of the frame's regenerated content: RFC 8878 3.1.1 Content_Checksum = its low 32 bits.  libzstd behind the
 * reference's zstd crate (src/compression.rs:151-159) verifies it when the frame header flags one; a mismatch fails the
 * decoder.  The algorithm is the published xxHash specification (four 64-bit lanes over 32-byte stripes, then the tail). */
#define XXP1 0x9E3779B185EBCA87ull
#define XXP2 0xC2B2AE3D27D4EB4Full
#define XXP3 0x165667B19E3779F9ull
#define XXP4 0x85EBCA77C2B2AE63ull
#define XXP5 0x27D4EB2F165667C5ull
static uint64_t xx_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t xx_rd64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
  return v;
}
static uint64_t xx_round(uint64_t acc, uint64_t in) { return xx_rotl(acc + in * XXP2, 31) * XXP1; }
static uint64_t xx_merge(uint64_t h, uint64_t v) { return (h ^ xx_round(0, v)) * XXP1 + XXP4; }
uint64_t oo_xxh64(const uint8_t* p, size_t n) {
  const uint8_t* end = p + n;
  uint64_t h;
  if (n >= 32) {
    uint64_t v1 = XXP1 + XXP2, v2 = XXP2, v3 = 0, v4 = 0 - XXP1;
    for (; p + 32 <= end; p += 32) {
      v1 = xx_round(v1, xx_rd64(p));
      v2 = xx_round(v2, xx_rd64(p + 8));
      v3 = xx_round(v3, xx_rd64(p + 16));
      v4 = xx_round(v4, xx_rd64(p + 24));
    }
    h = xx_rotl(v1, 1) + xx_rotl(v2, 7) + xx_rotl(v3, 12) + xx_rotl(v4, 18);
    h = xx_merge(h, v1);
    h = xx_merge(h, v2);
    h = xx_merge(h, v3);
    h = xx_merge(h, v4);
  } else {
    h = XXP5;
  }
  h += (uint64_t)n;
  for (; p + 8 <= end; p += 8) h = xx_rotl(h ^ xx_round(0, xx_rd64(p)), 27) * XXP1 + XXP4;
  if (p + 4 <= end) {
    uint64_t k = (uint64_t)p[0] | (uint64_t)p[1] << 8 | (uint64_t)p[2] << 16 | (uint64_t)p[3] << 24;
    h = xx_rotl(h ^ k * XXP1, 23) * XXP2 + XXP3;
    p += 4;
  }
  for (; p < end; p++) h = xx_rotl(h ^ *p * XXP5, 11) * XXP1;
  h ^= h >> 33;
  h *= XXP2;
  h ^= h >> 29;
  h *= XXP3;
  h ^= h >> 32;
  return h;
}

long oo_zstd_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  size_t pos = 0, out = 0;
  /* the zstd crate's streaming Decoder reads concatenated frames until EOF */
  while (pos < n) {
    if (pos + 4 > n) return -1;
    uint32_t magic = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
    if ((magic & 0xfffffff0u) == 0x184d2a50u) { /* skippable frame */
      if (pos + 8 > n) return -1;
      uint32_t sz = src[pos + 4] | (src[pos + 5] << 8) | (src[pos + 6] << 16) | ((uint32_t)src[pos + 7] << 24);
      if (pos + 8 + sz > n) return -1;
      pos += 8 + sz;
      continue;
    }
    if (magic != 0xfd2fb528u) return -1;
    pos += 4;
    if (pos >= n) return -1;
    uint8_t fhd = src[pos++];
    int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, has_ck = (fhd >> 2) & 1, did_flag = fhd & 3;
    if (fhd & 0x08) return -1; /* reserved bit */
    if (!single) {
      if (pos >= n) return -1;
      pos++; /* window descriptor: irrelevant, whole frame is decoded into dst */
    }
    static const int DID[4] = {0, 1, 2, 4};
    if (did_flag) {
      uint32_t did = 0;
      if (pos + (size_t)DID[did_flag] > n) return -1;
      for (int i = 0; i < DID[did_flag]; i++) did |= (uint32_t)src[pos + i] << (8 * i);
      pos += (size_t)DID[did_flag];
      if (did != 0) return -1; /* dictionaries unsupported */
    }
    int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    uint64_t fcs = 0;
    if (pos + (size_t)fcs_bytes > n) return -1;
    for (int i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
    if (fcs_bytes == 2) fcs += 256;
    pos += (size_t)fcs_bytes;
    size_t frame_start = out;
    zctx* z = (zctx*)calloc(1, sizeof(zctx));
    if (!z) return -1;
    z->rep[0] = 1;
    z->rep[1] = 4;
    z->rep[2] = 8;
    int last;
    do {
      if (pos + 3 > n) {
        free(z);
        return -1;
      }
      uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
      pos += 3;
      last = bh & 1;
      int bt = (bh >> 1) & 3;
      size_t bs = bh >> 3;
      if (bt == 0) {
        if (pos + bs > n || out + bs > cap) {
          free(z);
          return -1;
        }
        memcpy(dst + out, src + pos, bs);
        pos += bs;
        out += bs;
      } else if (bt == 1) {
        if (pos + 1 > n || out + bs > cap) {
          free(z);
          return -1;
        }
        memset(dst + out, src[pos], bs);
        pos += 1;
        out += bs;
      } else if (bt == 2) {
        if (pos + bs > n || bs > 128 * 1024) {
          free(z);
          return -1;
        }
        /* matches may reach back to the start of this frame only */
        long r = z_block(z, src + pos, bs, dst + frame_start, cap - frame_start, out - frame_start);
        if (r < 0) {
          free(z);
          return -1;
        }
        out = frame_start + (size_t)r;
        pos += bs;
      } else {
        free(z);
        return -1;
      }
    } while (!last);
    free(z);
    if (fcs_bytes && (uint64_t)(out - frame_start) != fcs) return -1;
    if (has_ck) {
      if (pos + 4 > n) return -1;
      uint32_t want = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
      if ((uint32_t)oo_xxh64(dst + frame_start, out - frame_start) != want) return -1; /* ZSTD_error_checksum_wrong */
      pos += 4;
    }
  }
  return (long)out;
}
