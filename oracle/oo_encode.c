/*
 * oo_encode.c -- ORACLE (test infrastructure only): a CPU restatement of the reference's value ENCODERS, one value at a time
 * as the reference runs them.  It exists so that the bytes of the device encoder (orc_rust_amd/csrc/device/rle_encode.hip) can
 * be compared with what the reference's writer would have produced for the same values; nothing in the product includes,
 * links or calls it.
 *
 * What is restated (reference file:line):
 *   RleV2Encoder::{process_value, flush}            src/encoding/integer/rle_v2/mod.rs:281-394
 *   determine_variable_run_encoding                 src/encoding/integer/rle_v2/mod.rs:422-531
 *   delta_encoding_check                            src/encoding/integer/rle_v2/mod.rs:186-239
 *   write_direct                                    src/encoding/integer/rle_v2/direct.rs:69-95
 *   write_short_repeat                              src/encoding/integer/rle_v2/short_repeat.rs:65-81
 *   write_fixed_delta / write_varying_delta         src/encoding/integer/rle_v2/delta.rs:118-182
 *   write_patched_base / derive_patches             src/encoding/integer/rle_v2/patched_base.rs:162-284
 *   write_packed_ints, write_varint, bit-width maps,
 *   calculate_percentile_bits                       src/encoding/integer/util.rs:222-365, 391-473, 501-520, 584-610
 *   ByteRleEncoder                                  src/encoding/byte.rs:38-197
 *   BooleanEncoder::finish                          src/encoding/boolean.rs:157-169
 *
 * The integer type N of the reference (i16 / i32 / i64: Int16 / Int32 / Int64 columns, i32 / i64 offsets of strings,
 * writer/column.rs:396-402) matters: bits_used() counts within N's width (a negative N is N's full width), zigzag wraps in N,
 * max.checked_sub(min) overflows in N.  Values travel here as int64_t, sign-extended from N; `nbits` is N's width.
 *
 * Pinned by tests/test_oracle_encode.py on the reference's own writer vectors (rle_v2/mod.rs:559-591), on the ORC
 * specification's examples the reference's reader tests hold, and by round trips through the (pinned) decoders of oo_encoding.c.
 *
 * Two inputs make the reference panic (an underflowing `brl_100p_bit_width - brl_95p_bit_width` when more than 5 % of the
 * base-reduced values sit in the top width class and that class is wider than the largest value, patched_base.rs:235; and
 * `base.abs()` of i64::MIN).  Runs for which that happens are written DIRECT here and counted in *ref_panics: there is no reference
 * byte string to match for them.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_oracle.h"

typedef struct {
  uint8_t* p;
  size_t len, cap;
} ebuf;

static void eb_put(ebuf* b, uint8_t v) {
  if (b->len == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 256;
    b->p = (uint8_t*)realloc(b->p, b->cap);
  }
  b->p[b->len++] = v;
}

/* VarintSerde::bits_used (integer/mod.rs:124-126) of a value of N = nbits wide held sign-extended */
static int bits_used_n(int64_t v, int nbits) { return v < 0 ? nbits : (v ? 64 - __builtin_clzll((uint64_t)v) : 0); }
/* signed_zigzag_encode (util.rs:550-553) in N; UnsignedEncoding leaves the value alone (integer/mod.rs:108-110) */
static int64_t zigzag_n(int64_t v, int nbits, int is_signed) {
  if (!is_signed) return v;
  uint64_t z = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); /* v is sign-extended: v >> (nbits - 1) == v >> 63 */
  /* wrap into N, sign-extended again */
  if (nbits == 64) return (int64_t)z;
  return (int64_t)(z << (64 - nbits)) >> (64 - nbits);
}
/* get_closest_fixed_bits (util.rs:407-421) */
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
/* encode_bit_width (util.rs:423-437) == rle_v2_encode_bit_width (util.rs:391-405) on the widths that one accepts */
static int encode_bit_width(int n) {
  n = closest_fixed_bits(n);
  if (n <= 24) return n - 1;
  if (n <= 26) return 24;
  if (n <= 28) return 25;
  if (n <= 30) return 26;
  if (n <= 32) return 27;
  if (n <= 40) return 28;
  if (n <= 48) return 29;
  if (n <= 56) return 30;
  return 31;
}
/* decode_bit_width (util.rs:439-453) */
static int decode_bit_width(int n) {
  static const int t[8] = {26, 28, 30, 32, 40, 48, 56, 64};
  return n <= 23 ? n + 1 : t[n - 24];
}
/* get_closest_aligned_bit_width (util.rs:456-472) */
static int closest_aligned_bit_width(int w) {
  if (w <= 1) return 1;
  if (w == 2) return 2;
  if (w <= 4) return 4;
  if (w <= 8) return 8;
  if (w <= 16) return 16;
  if (w <= 24) return 24;
  if (w <= 32) return 32;
  if (w <= 40) return 40;
  if (w <= 48) return 48;
  if (w <= 54) return 56;
  return 64;
}
/* calculate_percentile_bits (util.rs:584-610): f32 arithmetic as there */
static int percentile_bits(const int64_t* v, int n, int nbits, float percentile) {
  int hist[32] = {0};
  for (int i = 0; i < n; i++) hist[encode_bit_width(bits_used_n(v[i], nbits))]++;
  float count = (float)n;
  volatile float frac = 1.0f - percentile;
  volatile float prod = frac * count;
  size_t per_len = (size_t)prod;
  for (int i = 31; i >= 0; i--) {
    if (per_len >= (size_t)hist[i])
      per_len -= (size_t)hist[i];
    else
      return decode_bit_width(i);
  }
  return 1; /* unreachable!() */
}
/* write_packed_ints (util.rs:237-291; the aligned forms 293-365 are the same bytes for values that fit their width): the low
 * `w` bits of every value, most significant bit first, the last byte padded with zeros */
static void pack_ints(ebuf* b, int w, const int64_t* v, int n) {
  uint8_t cur = 0;
  int left = 8;
  for (int i = 0; i < n; i++) {
    uint64_t x = (uint64_t)v[i];
    int todo = w;
    while (todo > left) {
      int shift = todo - left;
      cur |= (uint8_t)((x >> shift) & (0xffu >> (8 - left)));
      todo -= left;
      eb_put(b, cur);
      cur = 0;
      left = 8;
    }
    left -= todo;
    cur |= (uint8_t)(((x & (todo == 64 ? ~0ull : ((1ull << todo) - 1))) << left) & 0xff);
    if (left == 0) {
      eb_put(b, cur);
      cur = 0;
      left = 8;
    }
  }
  if (left != 8) eb_put(b, cur);
}
/* write_varint (util.rs:501-520): `value >> shift` is N's ARITHMETIC shift -- a value with N's top bit set (a zigzag of a
 * large negative) gets a last byte of sign bits, as the reference writes it */
static void put_varint_n(ebuf* b, int64_t value, int nbits) {
  int size = (bits_used_n(value, nbits) + 6) / 7;
  if (size < 1) size = 1;
  for (int i = 0; i < size; i++) {
    int shift = i * 7;
    uint8_t byte = (uint8_t)((shift >= 64 ? (value >> 63) : (value >> shift)) & 0x7f);
    eb_put(b, (uint8_t)(byte | (i + 1 < size ? 0x80 : 0)));
  }
}

typedef struct {
  ebuf* out;
  int nbits, is_signed;
  uint64_t* stats; /* runs per sub-encoding: SHORT_REPEAT, DIRECT, PATCHED_BASE, DELTA; [4] = runs the reference panics on */
} enc2;

/* write_direct (direct.rs:69-95); values are zigzagged N already */
static void w_direct(enc2* e, const int64_t* zz, int n) {
  int mb = 0;
  for (int i = 0; i < n; i++) {
    int bu = bits_used_n(zz[i], e->nbits);
    if (bu > mb) mb = bu;
  }
  int w = closest_aligned_bit_width(mb);
  eb_put(e->out, (uint8_t)(0x40 | (encode_bit_width(w) << 1) | ((n - 1) >> 8)));
  eb_put(e->out, (uint8_t)((n - 1) & 0xff));
  pack_ints(e->out, w, zz, n);
  e->stats[1]++;
}
static void w_direct_of(enc2* e, const int64_t* lit, int n) {
  int64_t zz[512];
  for (int i = 0; i < n; i++) zz[i] = zigzag_n(lit[i], e->nbits, e->is_signed);
  w_direct(e, zz, n);
}
/* write_short_repeat (short_repeat.rs:65-81) */
static void w_short_repeat(enc2* e, int64_t value, int count) {
  int64_t z = zigzag_n(value, e->nbits, e->is_signed);
  int bytes = (bits_used_n(z, e->nbits) + 7) / 8;
  if (bytes < 1) bytes = 1;
  eb_put(e->out, (uint8_t)(((bytes - 1) << 3) | (count - 3)));
  for (int i = bytes - 1; i >= 0; i--) eb_put(e->out, (uint8_t)((uint64_t)z >> (8 * i)));
  e->stats[0]++;
}
/* derive_delta_header (delta.rs:161-182) */
static void delta_header(enc2* e, int width, int run_length) {
  int ew = width ? encode_bit_width(width) : 0;
  eb_put(e->out, (uint8_t)(0xc0 | (ew << 1) | ((run_length - 1) >> 8)));
  eb_put(e->out, (uint8_t)((run_length - 1) & 0xff));
}
/* write_fixed_delta (delta.rs:146-159) */
static void w_fixed_delta(enc2* e, int64_t base, int64_t delta, int subsequent) {
  delta_header(e, 0, subsequent + 2);
  put_varint_n(e->out, zigzag_n(base, e->nbits, e->is_signed), e->nbits);
  put_varint_n(e->out, zigzag_n(delta, 64, 1), 64);
  e->stats[3]++;
}
/* write_varying_delta (delta.rs:118-144) */
static void w_varying_delta(enc2* e, int64_t base, int64_t first_delta, int64_t max_delta, const int64_t* adj, int n_adj) {
  int w = closest_aligned_bit_width(bits_used_n(max_delta, 64));
  if (w == 1) w = 2;
  delta_header(e, w, n_adj + 2);
  put_varint_n(e->out, zigzag_n(base, e->nbits, e->is_signed), e->nbits);
  put_varint_n(e->out, zigzag_n(first_delta, 64, 1), 64);
  pack_ints(e->out, w, adj, n_adj);
  e->stats[3]++;
}
/* write_patched_base + derive_patches (patched_base.rs:162-284); brl: base-reduced literals */
static void w_patched_base(enc2* e, int64_t* brl, int n, int64_t base, int w100, int w95) {
  int pbw = closest_fixed_bits(w100 - w95);
  if (pbw == 64) {
    pbw = 56;
    w95 = 8;
  }
  /* derive_patches */
  int64_t mask = (int64_t)(((uint64_t)1 << w95) - 1);
  int64_t jump = (int64_t)255 << pbw;
  int64_t patches[40];
  int np = 0, last = 0, max_gap = 0;
  for (int idx = 0; idx < n; idx++) {
    if (brl[idx] <= mask) continue;
    uint64_t patch_bits = (uint64_t)brl[idx] >> w95;
    int gap = idx - last;
    if (gap == 511) {
      max_gap = 255;
      patches[np++] = jump;
      patches[np++] = jump;
      gap = 1;
    } else if (gap > 255) {
      max_gap = 255;
      patches[np++] = jump;
      gap -= 255;
    } else if (gap > max_gap) {
      max_gap = gap;
    }
    patches[np++] = (int64_t)(patch_bits | ((uint64_t)gap << pbw));
    last = idx;
    brl[idx] &= mask;
  }
  int pgw = max_gap == 0 ? 1 : bits_used_n(max_gap, 16);
  uint64_t abs_base = base < 0 ? (uint64_t)0 - (uint64_t)base : (uint64_t)base;
  int base_bits = closest_fixed_bits(bits_used_n((int64_t)abs_base, 64) + 1);
  int base_bytes = (base_bits + 7) / 8;
  if (base_bytes < 1) base_bytes = 1;
  uint64_t msb = abs_base | ((uint64_t)(base < 0) << (base_bytes * 8 - 1));
  eb_put(e->out, (uint8_t)(0x80 | (encode_bit_width(w95) << 1) | ((n - 1) >> 8)));
  eb_put(e->out, (uint8_t)((n - 1) & 0xff));
  eb_put(e->out, (uint8_t)(((base_bytes - 1) << 5) | encode_bit_width(pbw)));
  eb_put(e->out, (uint8_t)(((pgw - 1) << 5) | np));
  for (int i = base_bytes - 1; i >= 0; i--) eb_put(e->out, (uint8_t)(msb >> (8 * i)));
  pack_ints(e->out, closest_fixed_bits(w95), brl, n);
  pack_ints(e->out, closest_fixed_bits(pgw + pbw), patches, np);
  e->stats[2]++;
}

/* checked_sub in N */
static int sub_overflows_n(int64_t a, int64_t b, int nbits) {
  int64_t r;
  if (__builtin_sub_overflow(a, b, &r)) return 1;
  if (nbits == 64) return 0;
  int64_t lim = (int64_t)1 << (nbits - 1);
  return r >= lim || r < -lim;
}
static int64_t sat_sub(int64_t a, int64_t b) {
  int64_t r;
  if (__builtin_sub_overflow(a, b, &r)) return b > 0 ? INT64_MIN : INT64_MAX;
  return r;
}

/* determine_variable_run_encoding (rle_v2/mod.rs:422-531) with delta_encoding_check (:186-239) */
static void determine(enc2* e, const int64_t* lit, int n) {
  if (n <= 3) {
    w_direct_of(e, lit, n);
    return;
  }
  int64_t base_value = lit[0];
  int64_t min = lit[0] < lit[1] ? lit[0] : lit[1];
  int64_t max = lit[0] > lit[1] ? lit[0] : lit[1];
  int64_t first_delta = sat_sub(lit[1], lit[0]);
  int64_t max_delta = 0;
  int inc = first_delta > 0, dec = first_delta < 0, fixed = 1;
  int64_t adj[512];
  int n_adj = 0;
  for (int i = 2; i < n; i++) {
    if (lit[i] < min) min = lit[i];
    if (lit[i] > max) max = lit[i];
    int64_t cur = sat_sub(lit[i], lit[i - 1]);
    inc &= cur >= 0;
    dec &= cur <= 0;
    fixed &= cur == first_delta;
    cur = cur == INT64_MIN ? INT64_MAX : (cur < 0 ? -cur : cur); /* saturating_abs */
    adj[n_adj++] = cur;
    if (cur > max_delta) max_delta = cur;
  }
  if (sub_overflows_n(max, min, e->nbits)) {
    w_direct_of(e, lit, n);
    return;
  }
  if (fixed) {
    w_fixed_delta(e, lit[0], first_delta, n - 2);
    return;
  }
  if (first_delta != 0 && (inc || dec)) {
    w_varying_delta(e, base_value, first_delta, max_delta, adj, n_adj);
    return;
  }
  /* `min.abs() >= BASE_VALUE_LIMIT && min != i64::MIN`: of i64::MIN the release build's abs() is i64::MIN again (the debug build
   * panics there): the test is false and the run goes on */
  if (min != INT64_MIN && (min < 0 ? -min : min) >= ((int64_t)1 << 56)) {
    w_direct_of(e, lit, n);
    return;
  }
  int64_t zz[512];
  for (int i = 0; i < n; i++) zz[i] = zigzag_n(lit[i], e->nbits, e->is_signed);
  int z90 = percentile_bits(zz, n, e->nbits, 0.90f);
  int z100 = percentile_bits(zz, n, e->nbits, 1.00f);
  if ((z100 > z90 ? z100 - z90 : 0) <= 1) {
    w_direct(e, zz, n);
    return;
  }
  int64_t brl[512];
  int64_t max_data = 0;
  for (int i = 0; i < n; i++) {
    brl[i] = lit[i] - min;
    if (brl[i] > max_data) max_data = brl[i];
  }
  int w100 = bits_used_n(max_data, 64);
  int w95 = percentile_bits(brl, n, 64, 0.95f);
  if (w100 != w95) {
    if (w100 < w95 || min == INT64_MIN) { /* `brl_100p_bit_width - brl_95p_bit_width` underflows (patched_base.rs:235) / the base's
                                            abs() has 65 bits (:259): the reference panics */
      e->stats[4]++;
      w_direct(e, zz, n);
      return;
    }
    w_patched_base(e, brl, n, min, w100, w95);
  } else {
    w_direct(e, zz, n);
  }
}

/* RleV2Encoder::process_value / flush (rle_v2/mod.rs:281-394) over n values of N.  stats: 5 counters or NULL. */
int oo_enc_rle2(const int64_t* vals, uint64_t n, int int_bytes, int is_signed, uint8_t** out, uint64_t* out_len, uint64_t* stats) {
  if (int_bytes != 2 && int_bytes != 4 && int_bytes != 8) return OO_UNEXPECTED;
  ebuf b = {0, 0, 0};
  uint64_t st[5] = {0, 0, 0, 0, 0};
  enc2 e = {&b, int_bytes * 8, is_signed, st};
  enum { EMPTY, ONE, FIXED, VARIABLE } state = EMPTY;
  int64_t value = 0; /* One(value) / FixedRun.value */
  int count = 0;     /* FixedRun.count */
  int64_t lit[512];  /* VariableRun.literals */
  int nl = 0;
  for (uint64_t i = 0; i < n; i++) {
    int64_t v = vals[i];
    switch (state) {
      case EMPTY:
        state = ONE;
        value = v;
        break;
      case ONE:
        if (v == value) {
          state = FIXED;
          count = 2;
        } else {
          lit[0] = value;
          lit[1] = v;
          nl = 2;
          state = VARIABLE;
        }
        break;
      case FIXED:
        if (v == value) {
          count++;
          if (count == 512) {
            w_fixed_delta(&e, v, 0, count - 2);
            state = EMPTY;
          }
        } else if (count == 2) {
          lit[0] = lit[1] = value;
          lit[2] = v;
          nl = 3;
          state = VARIABLE;
        } else if (count <= 10) {
          w_short_repeat(&e, value, count);
          state = ONE;
          value = v;
        } else {
          w_fixed_delta(&e, value, 0, count - 2);
          state = ONE;
          value = v;
        }
        break;
      case VARIABLE:
        if (v == lit[nl - 1] && v == lit[nl - 2]) {
          nl -= 2;
          determine(&e, lit, nl);
          state = FIXED;
          value = v;
          count = 3;
        } else {
          lit[nl++] = v;
          if (nl == 512) {
            determine(&e, lit, nl);
            state = EMPTY;
          }
        }
        break;
    }
  }
  /* flush */
  switch (state) {
    case EMPTY:
      break;
    case ONE:
      w_direct_of(&e, &value, 1);
      break;
    case FIXED:
      if (count == 2) {
        int64_t two[2] = {value, value};
        w_direct_of(&e, two, 2);
      } else if (count <= 10) {
        w_short_repeat(&e, value, count);
      } else {
        w_fixed_delta(&e, value, 0, count - 2);
      }
      break;
    case VARIABLE:
      determine(&e, lit, nl);
      break;
  }
  *out = b.p;
  *out_len = b.len;
  if (stats) memcpy(stats, st, sizeof st);
  return OO_OK;
}

/* determine_variable_run_encoding alone: what the reference's writer tests call (rle_v2/mod.rs:559-591) */
int oo_enc_rle2_variable_run(const int64_t* lit, uint32_t n, int int_bytes, int is_signed, uint8_t** out, uint64_t* out_len) {
  if (n < 1 || n > 512) return OO_UNEXPECTED;
  ebuf b = {0, 0, 0};
  uint64_t st[5] = {0, 0, 0, 0, 0};
  enc2 e = {&b, int_bytes * 8, is_signed, st};
  determine(&e, lit, (int)n);
  *out = b.p;
  *out_len = b.len;
  return OO_OK;
}

/* ByteRleEncoder::{process_value, flush} (byte.rs:53-150), write_run / write_literals (byte.rs:176-197) */
int oo_enc_byte_rle(const uint8_t* vals, uint64_t n, uint8_t** out, uint64_t* out_len) {
  ebuf b = {0, 0, 0};
  uint8_t literals[128];
  int num_literals = 0, tail_run_length = 0, in_run = 0;
  uint8_t run_value = 0;
  for (uint64_t i = 0; i < n; i++) {
    uint8_t value = vals[i];
    if (num_literals == 0) {
      in_run = 0;
      literals[0] = value;
      num_literals = 1;
      tail_run_length = 1;
    } else if (in_run) {
      if (value == run_value) {
        num_literals++;
        if (num_literals == 130) {
          eb_put(&b, (uint8_t)(130 - 3));
          eb_put(&b, run_value);
          in_run = 0;
          tail_run_length = 0;
          num_literals = 0;
        }
      } else {
        eb_put(&b, (uint8_t)(num_literals - 3));
        eb_put(&b, run_value);
        in_run = 0;
        literals[0] = value;
        num_literals = 1;
        tail_run_length = 1;
      }
    } else {
      if (value == literals[num_literals - 1])
        tail_run_length++;
      else
        tail_run_length = 1;
      if (tail_run_length == 3) {
        if (num_literals + 1 == 3) {
          in_run = 1;
          run_value = value;
          num_literals++;
        } else {
          int len = num_literals - 2;
          eb_put(&b, (uint8_t)(-len));
          for (int k = 0; k < len; k++) eb_put(&b, literals[k]);
          in_run = 1;
          run_value = value;
          num_literals = 3;
        }
      } else {
        literals[num_literals++] = value;
        if (num_literals == 128) {
          eb_put(&b, (uint8_t)(-128));
          for (int k = 0; k < 128; k++) eb_put(&b, literals[k]);
          in_run = 0;
          tail_run_length = 0;
          num_literals = 0;
        }
      }
    }
  }
  if (num_literals != 0) {
    if (in_run) {
      eb_put(&b, (uint8_t)(num_literals - 3));
      eb_put(&b, run_value);
    } else {
      eb_put(&b, (uint8_t)(-num_literals));
      for (int k = 0; k < num_literals; k++) eb_put(&b, literals[k]);
    }
  }
  *out = b.p;
  *out_len = b.len;
  return OO_OK;
}

/* BooleanEncoder::finish (boolean.rs:157-169): the Arrow bitmap's bytes (least significant bit first, the last byte's spare bits
 * zero) with their bits reversed, through the byte encoder */
int oo_enc_boolean(const uint8_t* bits_lsb, uint64_t n_bits, uint8_t** out, uint64_t* out_len) {
  uint64_t nb = (n_bits + 7) / 8;
  uint8_t* rev = (uint8_t*)malloc(nb ? nb : 1);
  for (uint64_t i = 0; i < nb; i++) {
    uint8_t x = bits_lsb[i];
    if (i == nb - 1 && (n_bits & 7)) x &= (uint8_t)((1u << (n_bits & 7)) - 1);
    x = (uint8_t)((x >> 4) | (x << 4));
    x = (uint8_t)(((x & 0xcc) >> 2) | ((x & 0x33) << 2));
    x = (uint8_t)(((x & 0xaa) >> 1) | ((x & 0x55) << 1));
    rev[i] = x;
  }
  int rc = oo_enc_byte_rle(rev, nb, out, out_len);
  free(rev);
  return rc;
}

void oo_enc_free(void* p) { free(p); }
