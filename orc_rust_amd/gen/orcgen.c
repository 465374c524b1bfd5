/*
 * orcgen.c -- synthetic ORC stream generator (host side, plain C).
 *
 * The decode path needs realistic encoded inputs at sizes no fixture can hold (BASELINE.md
 * configs C2-C5).  This file holds the WRITE side needed for that and nothing else: an
 * Integer RLE v2 encoder with the sub-encoding chooser of the ORC specification (the same rules
 * as the reference's RleV2Encoder, src/encoding/integer/rle_v2/mod.rs:255-531, and of the
 * Java/C++ writers: SHORT_REPEAT for 3..10 repeats, DELTA for fixed/monotonic sequences,
 * PATCHED_BASE when the 100th and 90th percentile widths differ, else DIRECT), RLE v1, byte RLE
 * and boolean encoders, ORC chunk framing with greedy Snappy / LZ4 block compressors, and
 * zigzag varints for Decimal DATA.  It is NOT on the decode path and is never used to check
 * results: parity is always GPU vs oracle on the streams produced here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint8_t* p;
  size_t len, cap;
} obuf;

static void ob_reserve(obuf* b, size_t extra) {
  if (b->len + extra <= b->cap) return;
  size_t nc = b->cap ? b->cap * 2 : 4096;
  while (nc < b->len + extra) nc *= 2;
  b->p = (uint8_t*)realloc(b->p, nc);
  b->cap = nc;
}
static inline void ob_put(obuf* b, uint8_t v) {
  ob_reserve(b, 1);
  b->p[b->len++] = v;
}
static void ob_write(obuf* b, const void* src, size_t n) {
  ob_reserve(b, n);
  memcpy(b->p + b->len, src, n);
  b->len += n;
}

static inline uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
static void put_uvarint(obuf* b, uint64_t u) {
  while (u >= 0x80) {
    ob_put(b, (uint8_t)(u | 0x80));
    u >>= 7;
  }
  ob_put(b, (uint8_t)u);
}
static int bits_used(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }
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
static int closest_aligned_bits(int n) {
  if (n <= 1) return 1;
  if (n <= 2) return 2;
  if (n <= 4) return 4;
  if (n <= 8) return 8;
  if (n <= 16) return 16;
  if (n <= 24) return 24;
  if (n <= 32) return 32;
  if (n <= 40) return 40;
  if (n <= 48) return 48;
  if (n <= 56) return 56;
  return 64;
}
static int encode_width(int w) {
  if (w <= 24) return w - 1;
  switch (w) {
    case 26: return 24;
    case 28: return 25;
    case 30: return 26;
    case 32: return 27;
    case 40: return 28;
    case 48: return 29;
    case 56: return 30;
    default: return 31;
  }
}

/* MSB-first bit packing */
static void pack_bits(obuf* b, const uint64_t* v, int n, int w) {
  uint64_t acc = 0;
  int nbits = 0;
  for (int i = 0; i < n; i++) {
    uint64_t x = w == 64 ? v[i] : (v[i] & (((uint64_t)1 << w) - 1));
    int left = w;
    while (left > 0) {
      int take = 8 - nbits < left ? 8 - nbits : left;
      acc = (acc << take) | ((x >> (left - take)) & (((uint64_t)1 << take) - 1));
      nbits += take;
      left -= take;
      if (nbits == 8) {
        ob_put(b, (uint8_t)acc);
        acc = 0;
        nbits = 0;
      }
    }
  }
  if (nbits) ob_put(b, (uint8_t)(acc << (8 - nbits)));
}

/* ------------------------------------------------------------------------------------------ */
/* Integer RLE v2 encoder                                                                      */
#define MAXLIT 512
typedef struct {
  obuf* out;
  int is_signed, aligned;
  int64_t lit[MAXLIT];
  int n;
  int fixed_run, var_run;
  int64_t prev_delta;
  uint64_t stats[4]; /* runs emitted per sub-encoding */
} rle2;

static int pct_bits(const uint64_t* v, int n, double p) {
  int hist[65] = {0};
  for (int i = 0; i < n; i++) hist[closest_fixed_bits(bits_used(v[i]))]++;
  int per = (int)((1.0 - p) * n);
  for (int i = 64; i >= 0; i--) {
    per -= hist[i];
    if (per < 0) return i;
  }
  return 0;
}

static void w_short_repeat(rle2* e, int64_t v, int count) {
  uint64_t u = e->is_signed ? zigzag(v) : (uint64_t)v;
  int nb = (bits_used(u) + 7) / 8;
  if (nb == 0) nb = 1;
  ob_put(e->out, (uint8_t)(((nb - 1) << 3) | (count - 3)));
  for (int i = nb - 1; i >= 0; i--) ob_put(e->out, (uint8_t)(u >> (8 * i)));
  e->stats[0]++;
}
static void w_direct(rle2* e, const uint64_t* zz, int n) {
  int w = 0;
  for (int i = 0; i < n; i++) {
    int b = bits_used(zz[i]);
    if (b > w) w = b;
  }
  w = e->aligned ? closest_aligned_bits(w) : closest_fixed_bits(w);
  ob_put(e->out, (uint8_t)(0x40 | (encode_width(w) << 1) | (((n - 1) >> 8) & 1)));
  ob_put(e->out, (uint8_t)((n - 1) & 0xff));
  pack_bits(e->out, zz, n, w);
  e->stats[1]++;
}
static void w_delta(rle2* e, const int64_t* lit, int n, int fixed, int64_t first_delta, const uint64_t* adj, uint64_t maxdelta) {
  int w = 0;
  if (!fixed) {
    w = closest_fixed_bits(bits_used(maxdelta));
    if (e->aligned) w = closest_aligned_bits(w);
    if (w == 1) w = 2;
  }
  ob_put(e->out, (uint8_t)(0xc0 | ((w ? encode_width(w) : 0) << 1) | (((n - 1) >> 8) & 1)));
  ob_put(e->out, (uint8_t)((n - 1) & 0xff));
  put_uvarint(e->out, e->is_signed ? zigzag(lit[0]) : (uint64_t)lit[0]);
  put_uvarint(e->out, zigzag(first_delta));
  if (!fixed) pack_bits(e->out, adj + 1, n - 2, w);
  e->stats[3]++;
}

static void w_patched(rle2* e, const int64_t* lit, int n, int64_t min, int br95, int br100) {
  uint64_t red[MAXLIT];
  for (int i = 0; i < n; i++) red[i] = (uint64_t)lit[i] - (uint64_t)min;
  int W = br95;
  int pw = closest_fixed_bits(br100 - br95);
  if (pw == 64) {
    pw = 56;
    W = 8;
  }
  uint64_t mask = W == 64 ? ~(uint64_t)0 : (((uint64_t)1 << W) - 1);
  uint64_t gaps[MAXLIT], patches[MAXLIT];
  int np = 0, prev = 0;
  uint64_t maxgap = 0;
  for (int i = 0; i < n; i++) {
    if (red[i] > mask) {
      uint64_t gap = (uint64_t)(i - prev);
      if (gap > maxgap) maxgap = gap;
      prev = i;
      gaps[np] = gap;
      patches[np] = red[i] >> W;
      np++;
      red[i] &= mask;
    }
  }
  int pgw = maxgap == 0 ? 1 : closest_fixed_bits(bits_used(maxgap));
  if (pgw > 8) pgw = 8;
  /* entries: gaps above 255 are split into (255, 0) continuation entries */
  uint64_t ent[MAXLIT * 3];
  int ne = 0;
  for (int i = 0; i < np; i++) {
    uint64_t g = gaps[i];
    while (g > 255) {
      ent[ne++] = (uint64_t)255 << pw;
      g -= 255;
    }
    ent[ne++] = (g << pw) | patches[i];
  }
  if (ne > 31 || ne == 0) { /* cannot be represented: fall back to DIRECT */
    uint64_t zz[MAXLIT];
    for (int i = 0; i < n; i++) zz[i] = e->is_signed ? zigzag(lit[i]) : (uint64_t)lit[i];
    w_direct(e, zz, n);
    return;
  }
  uint64_t amin = min < 0 ? (uint64_t)0 - (uint64_t)min : (uint64_t)min;
  int bb = bits_used(amin) + 1; /* sign bit */
  int base_bytes = (bb + 7) / 8;
  if (base_bytes == 0) base_bytes = 1;
  uint64_t base_enc = amin | (min < 0 ? ((uint64_t)1 << (base_bytes * 8 - 1)) : 0);
  ob_put(e->out, (uint8_t)(0x80 | (encode_width(W) << 1) | (((n - 1) >> 8) & 1)));
  ob_put(e->out, (uint8_t)((n - 1) & 0xff));
  ob_put(e->out, (uint8_t)(((base_bytes - 1) << 5) | encode_width(pw)));
  ob_put(e->out, (uint8_t)(((pgw - 1) << 5) | ne));
  for (int i = base_bytes - 1; i >= 0; i--) ob_put(e->out, (uint8_t)(base_enc >> (8 * i)));
  pack_bits(e->out, red, n, W);
  pack_bits(e->out, ent, ne, closest_fixed_bits(pw + pgw));
  e->stats[2]++;
}

static void determine_and_write(rle2* e) {
  int n = e->n;
  const int64_t* lit = e->lit;
  uint64_t zz[MAXLIT];
  for (int i = 0; i < n; i++) zz[i] = e->is_signed ? zigzag(lit[i]) : (uint64_t)lit[i];
  if (n <= 3) {
    w_direct(e, zz, n);
    goto done;
  }
  {
    int inc = 1, dec = 1, fixed = 1, ovf = 0;
    int64_t min = lit[0], max = lit[0];
    int64_t first_delta;
    uint64_t adj[MAXLIT];
    uint64_t maxdelta = 0;
    ovf |= __builtin_sub_overflow(lit[1], lit[0], &first_delta);
    for (int i = 1; i < n; i++) {
      int64_t d;
      ovf |= __builtin_sub_overflow(lit[i], lit[i - 1], &d);
      if (lit[i] < min) min = lit[i];
      if (lit[i] > max) max = lit[i];
      inc &= d >= 0;
      dec &= d <= 0;
      fixed &= d == first_delta;
      uint64_t a = d < 0 ? (uint64_t)0 - (uint64_t)d : (uint64_t)d;
      adj[i - 1] = a;
      if (i > 1 && a > maxdelta) maxdelta = a;
    }
    int64_t range;
    if (ovf || __builtin_sub_overflow(max, min, &range)) {
      w_direct(e, zz, n);
      goto done;
    }
    if (fixed) {
      w_delta(e, lit, n, 1, first_delta, adj, 0);
      goto done;
    }
    if (first_delta != 0 && (inc || dec)) {
      w_delta(e, lit, n, 0, first_delta, adj, maxdelta);
      goto done;
    }
    int zz90 = pct_bits(zz, n, 0.9), zz100 = pct_bits(zz, n, 1.0);
    if (zz100 - zz90 > 1) {
      uint64_t red[MAXLIT];
      for (int i = 0; i < n; i++) red[i] = (uint64_t)lit[i] - (uint64_t)min;
      int br95 = pct_bits(red, n, 0.95), br100 = pct_bits(red, n, 1.0);
      uint64_t amin = min < 0 ? (uint64_t)0 - (uint64_t)min : (uint64_t)min;
      if (br100 != br95 && amin < ((uint64_t)1 << 56)) {
        w_patched(e, lit, n, min, br95, br100);
        goto done;
      }
    }
    w_direct(e, zz, n);
  }
done:
  e->n = 0;
  e->fixed_run = e->var_run = 0;
  e->prev_delta = 0;
}

static void write_fixed_run(rle2* e) {
  if (e->fixed_run <= 10) {
    w_short_repeat(e, e->lit[0], e->fixed_run);
  } else {
    uint64_t adj[1] = {0};
    w_delta(e, e->lit, e->fixed_run, 1, 0, adj, 0);
  }
  e->n = 0;
  e->fixed_run = e->var_run = 0;
  e->prev_delta = 0;
}

static void rle2_put(rle2* e, int64_t v) {
  if (e->n == 0) {
    e->lit[e->n++] = v;
    e->fixed_run = 1;
    e->var_run = 1;
    return;
  }
  if (e->n == 1) {
    e->prev_delta = (int64_t)((uint64_t)v - (uint64_t)e->lit[0]);
    e->lit[e->n++] = v;
    if (v == e->lit[0]) {
      e->fixed_run = 2;
      e->var_run = 0;
    } else {
      e->fixed_run = 0;
      e->var_run = 2;
    }
    return;
  }
  int64_t cur = (int64_t)((uint64_t)v - (uint64_t)e->lit[e->n - 1]);
  if (e->prev_delta == 0 && cur == 0) {
    e->lit[e->n++] = v;
    if (e->var_run > 0) e->fixed_run = 2;
    e->fixed_run += 1;
    if (e->fixed_run >= 3 && e->var_run > 0) {
      /* flush the variable part, keep the 3 repeats */
      e->n -= 3;
      e->var_run -= 2;
      int64_t keep = v;
      int fr = e->fixed_run;
      determine_and_write(e);
      e->lit[0] = e->lit[1] = e->lit[2] = keep;
      e->n = 3;
      e->fixed_run = fr;
      e->var_run = 0;
      e->prev_delta = 0;
    }
    if (e->fixed_run == MAXLIT) write_fixed_run(e);
    return;
  }
  if (e->fixed_run >= 3) write_fixed_run(e);
  if (e->fixed_run > 0 && e->fixed_run < 3 && e->n > 0 && v != e->lit[e->n - 1]) {
    e->var_run = e->fixed_run;
    e->fixed_run = 0;
  }
  if (e->n == 0) {
    e->lit[e->n++] = v;
    e->fixed_run = 1;
    e->var_run = 1;
  } else {
    e->prev_delta = (int64_t)((uint64_t)v - (uint64_t)e->lit[e->n - 1]);
    e->lit[e->n++] = v;
    e->var_run += 1;
    if (e->var_run == MAXLIT) determine_and_write(e);
  }
}

static void rle2_flush(rle2* e) {
  if (e->n == 0) return;
  if (e->var_run != 0) {
    determine_and_write(e);
  } else if (e->fixed_run != 0) {
    if (e->fixed_run < 3) {
      e->var_run = e->fixed_run;
      e->fixed_run = 0;
      determine_and_write(e);
    } else {
      write_fixed_run(e);
    }
  }
}

/* Public: encode n values; returns malloc'd buffer via *out / *out_len; stats[4] = runs per sub-encoding */
int orcgen_rle2(const int64_t* vals, size_t n, int is_signed, int aligned, uint8_t** out, size_t* out_len, uint64_t* stats) {
  obuf b = {0, 0, 0};
  rle2* e = (rle2*)calloc(1, sizeof(rle2));
  e->out = &b;
  e->is_signed = is_signed;
  e->aligned = aligned;
  for (size_t i = 0; i < n; i++) rle2_put(e, vals[i]);
  rle2_flush(e);
  if (stats) memcpy(stats, e->stats, sizeof(e->stats));
  free(e);
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

/* Public: like orcgen_rle2, but the encoder is flushed behind every segment (seg_lens[k] values each): run boundaries where the
 * caller wants them, e.g. run lengths a writer would not normally produce */
int orcgen_rle2_segments(const int64_t* vals, size_t n, const uint32_t* seg_lens, size_t n_seg, int is_signed, int aligned, uint8_t** out,
                         size_t* out_len, uint64_t* stats) {
  obuf b = {0, 0, 0};
  rle2* e = (rle2*)calloc(1, sizeof(rle2));
  e->out = &b;
  e->is_signed = is_signed;
  e->aligned = aligned;
  size_t i = 0;
  for (size_t k = 0; k < n_seg && i < n; k++) {
    for (uint32_t t = 0; t < seg_lens[k] && i < n; t++) rle2_put(e, vals[i++]);
    rle2_flush(e);
  }
  for (; i < n; i++) rle2_put(e, vals[i]);
  rle2_flush(e);
  if (stats) memcpy(stats, e->stats, sizeof(e->stats));
  free(e);
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

/* Public: orcgen_rle2_segments (n_seg = 0: no forced flushes) that also records, as a writer does for its ROW_INDEX streams, the
 * position of every `stride`-th value: positions[2g] = bytes written when value g * stride arrives, positions[2g + 1] = values
 * the encoder holds at that moment (they open the run(s) that start at that byte).  ceil(n / stride) pairs. */
int orcgen_rle2_marked(const int64_t* vals, size_t n, const uint32_t* seg_lens, size_t n_seg, int is_signed, int aligned, uint32_t stride,
                       const uint64_t* marks, size_t n_marks, uint64_t* positions, uint8_t** out, size_t* out_len);
int orcgen_rle2_indexed(const int64_t* vals, size_t n, const uint32_t* seg_lens, size_t n_seg, int is_signed, int aligned, uint32_t stride,
                        uint64_t* positions, uint8_t** out, size_t* out_len) {
  return orcgen_rle2_marked(vals, n, seg_lens, n_seg, is_signed, aligned, stride, NULL, 0, positions, out, out_len);
}
/* ... or of the values marks[0] < marks[1] < ... (a column with nulls: the value a row group starts with is the count of
 * non-null rows before it); a mark == n (no value behind it) gets the end of the stream */
int orcgen_rle2_marked(const int64_t* vals, size_t n, const uint32_t* seg_lens, size_t n_seg, int is_signed, int aligned, uint32_t stride,
                       const uint64_t* marks, size_t n_marks, uint64_t* positions, uint8_t** out, size_t* out_len) {
  obuf b = {0, 0, 0};
  rle2* e = (rle2*)calloc(1, sizeof(rle2));
  e->out = &b;
  e->is_signed = is_signed;
  e->aligned = aligned;
  size_t i = 0, k = 0, m = 0;
  uint32_t in_seg = 0;
  for (; i < n; i++) {
    if (marks) {
      while (m < n_marks && marks[m] == i) {
        positions[2 * m] = b.len;
        positions[2 * m + 1] = (uint64_t)e->n;
        m++;
      }
    } else if (stride && i % stride == 0) {
      positions[2 * (i / stride)] = b.len;
      positions[2 * (i / stride) + 1] = (uint64_t)e->n;
    }
    rle2_put(e, vals[i]);
    if (k < n_seg && ++in_seg == seg_lens[k]) {
      rle2_flush(e);
      in_seg = 0;
      k++;
    }
  }
  rle2_flush(e);
  for (; marks && m < n_marks; m++) {
    positions[2 * m] = b.len;
    positions[2 * m + 1] = 0;
  }
  free(e);
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

/* Public: where the values marks[0] < marks[1] < ... of a byte-RLE stream lie: positions[2m] = offset of the header of the run /
 * literal group that holds value marks[m], positions[2m + 1] = values of that group in front of it (what a writer records for
 * a byte-RLE stream: row_index.rs:42-50).  A mark behind the last value gets the end of the stream. */
int orcgen_byte_rle_positions(const uint8_t* s, size_t len, const uint64_t* marks, size_t n_marks, uint64_t* positions) {
  size_t p = 0, m = 0;
  uint64_t v = 0;
  while (p < len && m < n_marks) {
    const uint8_t c = s[p];
    const uint64_t cnt = c < 0x80 ? (uint64_t)c + 3 : 256 - (uint64_t)c;
    const size_t size = c < 0x80 ? 2 : 1 + (size_t)cnt;
    while (m < n_marks && marks[m] < v + cnt) {
      positions[2 * m] = p;
      positions[2 * m + 1] = marks[m] - v;
      m++;
    }
    v += cnt;
    p += size;
  }
  for (; m < n_marks; m++) {
    positions[2 * m] = len;
    positions[2 * m + 1] = 0;
  }
  return 0;
}

/* RLE v1 (spec: runs of 3..130 with delta -128..127, literal groups up to 128) */
int orcgen_rle1(const int64_t* vals, size_t n, int is_signed, uint8_t** out, size_t* out_len) {
  obuf b = {0, 0, 0};
  size_t i = 0;
  while (i < n) {
    /* find a run starting at i */
    size_t run = 1;
    int64_t delta = 0;
    if (i + 1 < n) {
      int64_t d;
      if (!__builtin_sub_overflow(vals[i + 1], vals[i], &d) && d >= -128 && d <= 127) {
        delta = d;
        run = 2;
        while (i + run < n && run < 130) {
          int64_t d2;
          if (__builtin_sub_overflow(vals[i + run], vals[i + run - 1], &d2) || d2 != delta) break;
          run++;
        }
      }
    }
    if (run >= 3) {
      ob_put(&b, (uint8_t)(run - 3));
      ob_put(&b, (uint8_t)(int8_t)delta);
      put_uvarint(&b, is_signed ? zigzag(vals[i]) : (uint64_t)vals[i]);
      i += run;
      continue;
    }
    /* literals until the next run of >= 3 or 128 values */
    size_t start = i, cnt = 0;
    while (i < n && cnt < 128) {
      if (i + 2 < n) {
        int64_t d1, d2;
        if (!__builtin_sub_overflow(vals[i + 1], vals[i], &d1) && !__builtin_sub_overflow(vals[i + 2], vals[i + 1], &d2) && d1 == d2 &&
            d1 >= -128 && d1 <= 127)
          break;
      }
      i++;
      cnt++;
    }
    if (cnt == 0) { /* a run starts right here but was shorter than 3: emit one literal */
      i++;
      cnt = 1;
    }
    ob_put(&b, (uint8_t)(int8_t)(-(int)cnt));
    for (size_t k = 0; k < cnt; k++) put_uvarint(&b, is_signed ? zigzag(vals[start + k]) : (uint64_t)vals[start + k]);
  }
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

/* byte RLE (runs 3..130, literals 1..128) */
int orcgen_byte_rle(const uint8_t* v, size_t n, uint8_t** out, size_t* out_len) {
  obuf b = {0, 0, 0};
  size_t i = 0;
  while (i < n) {
    size_t run = 1;
    while (i + run < n && run < 130 && v[i + run] == v[i]) run++;
    if (run >= 3) {
      ob_put(&b, (uint8_t)(run - 3));
      ob_put(&b, v[i]);
      i += run;
      continue;
    }
    size_t start = i, cnt = 0;
    while (i < n && cnt < 128) {
      if (i + 2 < n && v[i] == v[i + 1] && v[i] == v[i + 2]) break;
      i++;
      cnt++;
    }
    ob_put(&b, (uint8_t)(0x100 - cnt));
    ob_write(&b, v + start, cnt);
  }
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

/* booleans (one byte each) -> MSB-first bit bytes -> byte RLE */
int orcgen_bool(const uint8_t* bools, size_t n, uint8_t** out, size_t* out_len) {
  size_t nb = (n + 7) / 8;
  uint8_t* bits = (uint8_t*)calloc(nb + 1, 1);
  for (size_t i = 0; i < n; i++)
    if (bools[i]) bits[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
  int rc = orcgen_byte_rle(bits, nb, out, out_len);
  free(bits);
  return rc;
}

/* zigzag varints of 128-bit values given as (lo, hi) pairs */
int orcgen_varint128(const uint64_t* lohi, size_t n, uint8_t** out, size_t* out_len) {
  obuf b = {0, 0, 0};
  for (size_t i = 0; i < n; i++) {
    unsigned __int128 v = ((unsigned __int128)lohi[2 * i + 1] << 64) | lohi[2 * i];
    __int128 s = (__int128)v;
    unsigned __int128 u = ((unsigned __int128)s << 1) ^ (unsigned __int128)(s >> 127);
    while (u >= 0x80) {
      ob_put(&b, (uint8_t)((uint8_t)u | 0x80));
      u >>= 7;
    }
    ob_put(&b, (uint8_t)u);
  }
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* greedy block compressors (valid streams, modest ratios -- enough to exercise the decoders)  */
static size_t snappy_block(const uint8_t* src, size_t n, uint8_t* dst) {
  size_t o = 0;
  uint64_t u = n;
  while (u >= 0x80) {
    dst[o++] = (uint8_t)(u | 0x80);
    u >>= 7;
  }
  dst[o++] = (uint8_t)u;
  enum { HB = 14 };
  static __thread int32_t table[1 << HB];
  memset(table, -1, sizeof(table));
  size_t i = 0, lit = 0;
#define EMIT_LIT(upto)                                            \
  while (lit < (upto)) {                                          \
    size_t l = (upto)-lit;                                        \
    if (l > 65536) l = 65536;                                     \
    if (l <= 60) dst[o++] = (uint8_t)((l - 1) << 2);              \
    else if (l <= 256) {                                          \
      dst[o++] = 60 << 2;                                         \
      dst[o++] = (uint8_t)(l - 1);                                \
    } else {                                                      \
      dst[o++] = 61 << 2;                                         \
      dst[o++] = (uint8_t)((l - 1) & 0xff);                       \
      dst[o++] = (uint8_t)((l - 1) >> 8);                         \
    }                                                             \
    memcpy(dst + o, src + lit, l);                                \
    o += l;                                                       \
    lit += l;                                                     \
  }
  while (i + 4 <= n) {
    uint32_t w;
    memcpy(&w, src + i, 4);
    uint32_t h = (w * 2654435761u) >> (32 - HB);
    int32_t cand = table[h];
    table[h] = (int32_t)i;
    uint32_t cw = 0;
    if (cand >= 0) memcpy(&cw, src + cand, 4);
    if (cand >= 0 && cw == w && i - (size_t)cand <= 65535) {
      size_t len = 4;
      while (i + len < n && src[cand + len] == src[i + len] && len < 64) len++;
      EMIT_LIT(i);
      size_t off = i - (size_t)cand;
      if (len <= 11 && off < 2048) {
        dst[o++] = (uint8_t)(1 | ((len - 4) << 2) | ((off >> 8) << 5));
        dst[o++] = (uint8_t)(off & 0xff);
      } else {
        dst[o++] = (uint8_t)(2 | ((len - 1) << 2));
        dst[o++] = (uint8_t)(off & 0xff);
        dst[o++] = (uint8_t)(off >> 8);
      }
      i += len;
      lit = i;
    } else {
      i++;
    }
  }
  EMIT_LIT(n);
#undef EMIT_LIT
  return o;
}

static size_t lz4_block(const uint8_t* src, size_t n, uint8_t* dst) {
  enum { HB = 14 };
  static __thread int32_t table[1 << HB];
  memset(table, -1, sizeof(table));
  size_t o = 0, i = 0, anchor = 0;
  /* the last 5 bytes are always literals, a match must not start within the last 12 bytes */
  size_t mflimit = n > 12 ? n - 12 : 0;
  while (i < mflimit) {
    uint32_t w;
    memcpy(&w, src + i, 4);
    uint32_t h = (w * 2654435761u) >> (32 - HB);
    int32_t cand = table[h];
    table[h] = (int32_t)i;
    uint32_t cw = 0;
    if (cand >= 0) memcpy(&cw, src + cand, 4);
    if (cand >= 0 && cw == w && i - (size_t)cand <= 65535) {
      size_t len = 4;
      size_t maxl = n - 5 - i;
      while (len < maxl && src[cand + len] == src[i + len]) len++;
      size_t litlen = i - anchor;
      size_t tokpos = o++;
      uint8_t tok = (uint8_t)((litlen >= 15 ? 15 : litlen) << 4);
      if (litlen >= 15) {
        size_t r = litlen - 15;
        while (r >= 255) {
          dst[o++] = 255;
          r -= 255;
        }
        dst[o++] = (uint8_t)r;
      }
      memcpy(dst + o, src + anchor, litlen);
      o += litlen;
      size_t off = i - (size_t)cand;
      dst[o++] = (uint8_t)(off & 0xff);
      dst[o++] = (uint8_t)(off >> 8);
      size_t ml = len - 4;
      tok |= (uint8_t)(ml >= 15 ? 15 : ml);
      if (ml >= 15) {
        size_t r = ml - 15;
        while (r >= 255) {
          dst[o++] = 255;
          r -= 255;
        }
        dst[o++] = (uint8_t)r;
      }
      dst[tokpos] = tok;
      i += len;
      anchor = i;
    } else {
      i++;
    }
  }
  size_t litlen = n - anchor;
  size_t tokpos = o++;
  dst[tokpos] = (uint8_t)((litlen >= 15 ? 15 : litlen) << 4);
  if (litlen >= 15) {
    size_t r = litlen - 15;
    while (r >= 255) {
      dst[o++] = 255;
      r -= 255;
    }
    dst[o++] = (uint8_t)r;
  }
  memcpy(dst + o, src + anchor, litlen);
  o += litlen;
  return o;
}

/* ORC chunk framing (compression.rs:113-123): kind 2 = snappy, 4 = lz4.  Chunks that do not
 * shrink are stored "original", like the ORC writers do. */
int orcgen_compress_stream(const uint8_t* src, size_t n, int kind, size_t block_size, uint8_t** out, size_t* out_len) {
  obuf b = {0, 0, 0};
  uint8_t* tmp = (uint8_t*)malloc(block_size + block_size / 4 + 1024);
  for (size_t p = 0; p < n; p += block_size) {
    size_t len = n - p < block_size ? n - p : block_size;
    size_t c = kind == 2 ? snappy_block(src + p, len, tmp) : lz4_block(src + p, len, tmp);
    uint32_t hdr;
    if (c < len) {
      hdr = (uint32_t)(c << 1);
      uint8_t h3[3] = {(uint8_t)hdr, (uint8_t)(hdr >> 8), (uint8_t)(hdr >> 16)};
      ob_write(&b, h3, 3);
      ob_write(&b, tmp, c);
    } else {
      hdr = (uint32_t)((len << 1) | 1);
      uint8_t h3[3] = {(uint8_t)hdr, (uint8_t)(hdr >> 8), (uint8_t)(hdr >> 16)};
      ob_write(&b, h3, 3);
      ob_write(&b, src + p, len);
    }
  }
  free(tmp);
  ob_reserve(&b, 64);
  *out = b.p;
  *out_len = b.len;
  return 0;
}

void orcgen_free(void* p) { free(p); }

/* splitmix64, for the seeded synthetic columns of BASELINE.md */
void orcgen_splitmix64(uint64_t seed, uint64_t* out, size_t n) {
  uint64_t x = seed;
  for (size_t i = 0; i < n; i++) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    out[i] = z ^ (z >> 31);
  }
}
