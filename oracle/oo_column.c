/*
 * oo_column.c -- ORACLE (test infrastructure only): the reference's array decoders for flat
 * ORC types, batch by batch, exactly as NaiveStripeDecoder drives them.
 *
 *   array_decoder/mod.rs:87-139    PrimitiveArrayDecoder::next_primitive_batch
 *   array_decoder/mod.rs:149-190   BooleanArrayDecoder
 *   array_decoder/mod.rs:192-252   PresentDecoder / derive_present_vec (None when no nulls)
 *   array_decoder/mod.rs:390-511   array_decoder_factory (stream wiring per ORC type)
 *   encoding/mod.rs:64-91          decode_spaced (null slots keep the zero fill)
 *   array_decoder/string.rs        direct + dictionary strings, Binary
 *   array_decoder/decimal.rs       Decimal128 with per-value scale repair
 *   array_decoder/timestamp.rs     Timestamp / TimestampInstant (writer tz None/UTC/GMT only:
 *                                  the base is supplied by the caller, conversion is identity)
 *   encoding/float.rs:53-75        Float/Double raw little-endian copy
 */
#include <stdlib.h>
#include <string.h>

#include "orc_oracle.h"

/* internal symbols from oo_encoding.c reached through the public API only */

struct oo_column {
  oo_column_desc d;
  oo_reader *r_present, *r_data, *r_length, *r_secondary, *r_dict;
  oo_bool_dec* present;
  oo_bool_dec* bool_data;
  oo_byte_rle* byte_data;
  oo_int_rle *int_data, *int_length, *int_secondary;
  /* dictionary (string.rs:65-82) */
  uint8_t* dict_bytes;
  int32_t* dict_offsets;
  uint64_t dict_n;
  int dict_status;
  /* batch output storage */
  uint8_t* validity;
  uint8_t* values;
  int32_t* offsets;
  size_t validity_cap, values_cap, offsets_cap;
};

static const oo_stream_t* find_stream(const oo_column_desc* d, int kind) {
  for (uint32_t i = 0; i < d->n_streams; i++)
    if (d->streams[i].kind == kind) return &d->streams[i];
  return NULL;
}

/* StreamMap::get: a missing stream is an empty stream (stripe.rs:319-326) */
static oo_reader* open_stream(const oo_column_desc* d, int kind, int* exists) {
  const oo_stream_t* s = find_stream(d, kind);
  if (exists) *exists = s != NULL;
  if (!s) return oo_reader_new(NULL, 0, OO_COMP_NONE, 0);
  return oo_reader_new(s->ptr, (size_t)s->len, d->compression, (size_t)d->block_size);
}

static void* grow(void* p, size_t* cap, size_t need) {
  if (need <= *cap && p) return p;
  size_t nc = need + need / 2 + 64;
  p = realloc(p, nc);
  *cap = nc;
  return p;
}

static int is_v2(const oo_column_desc* d) { return d->encoding == 2 || d->encoding == 3; } /* column.rs:52-59 */
static int is_dict(const oo_column_desc* d) { return d->encoding == 1 || d->encoding == 3; }

static int utf8_valid(const uint8_t* s, size_t n) {
  size_t i = 0;
  while (i < n) {
    uint8_t c = s[i];
    if (c < 0x80) {
      i++;
    } else if (c >= 0xc2 && c <= 0xdf) {
      if (i + 1 >= n || (s[i + 1] & 0xc0) != 0x80) return 0;
      i += 2;
    } else if (c >= 0xe0 && c <= 0xef) {
      if (i + 2 >= n) return 0;
      uint8_t c1 = s[i + 1], c2 = s[i + 2];
      if ((c1 & 0xc0) != 0x80 || (c2 & 0xc0) != 0x80) return 0;
      if (c == 0xe0 && c1 < 0xa0) return 0;
      if (c == 0xed && c1 > 0x9f) return 0;
      i += 3;
    } else if (c >= 0xf0 && c <= 0xf4) {
      if (i + 3 >= n) return 0;
      uint8_t c1 = s[i + 1], c2 = s[i + 2], c3 = s[i + 3];
      if ((c1 & 0xc0) != 0x80 || (c2 & 0xc0) != 0x80 || (c3 & 0xc0) != 0x80) return 0;
      if (c == 0xf0 && c1 < 0x90) return 0;
      if (c == 0xf4 && c1 > 0x8f) return 0;
      i += 4;
    } else {
      return 0;
    }
  }
  return 1;
}

/* GenericByteArrayDecoder::next_byte_batch without PRESENT (dictionary load, string.rs:70-72) */
static int load_dictionary(oo_column* c) {
  uint64_t n = c->d.dictionary_size;
  int64_t* lens = (int64_t*)malloc((n + 1) * sizeof(int64_t));
  int st = oo_int_rle_decode(c->int_length, lens, (size_t)n);
  if (st) {
    free(lens);
    return st;
  }
  int64_t total = 0;
  for (uint64_t i = 0; i < n; i++) total += lens[i];
  if (total > INT32_MAX) {
    free(lens);
    return OO_OFFSET_OVERFLOW;
  }
  c->dict_offsets = (int32_t*)malloc((n + 1) * sizeof(int32_t));
  int64_t acc = 0;
  c->dict_offsets[0] = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (lens[i] < 0) {
      free(lens);
      return OO_ARROW;
    }
    acc += lens[i];
    c->dict_offsets[i + 1] = (int32_t)acc;
  }
  free(lens);
  c->dict_bytes = (uint8_t*)malloc((size_t)total + 1);
  /* take(total).read_to_end: a short stream is not an IoError, try_new then rejects the offsets */
  uint8_t* all = NULL;
  size_t all_len = 0;
  const oo_stream_t* s = find_stream(&c->d, OO_S_DICTIONARY_DATA);
  if (s) {
    st = oo_stream_decompress(s->ptr, (size_t)s->len, c->d.compression, (size_t)c->d.block_size, &all, &all_len);
    if (st && all_len < (size_t)total) {
      free(all);
      return st;
    }
  }
  if (all_len < (size_t)total) {
    free(all);
    return OO_ARROW;
  }
  memcpy(c->dict_bytes, all, (size_t)total);
  free(all);
  if (c->d.orc_type != OO_T_BINARY) {
    if (!utf8_valid(c->dict_bytes, (size_t)total)) return OO_ARROW;
    for (uint64_t i = 0; i <= n; i++) {
      int32_t o = c->dict_offsets[i];
      if (o < total && (c->dict_bytes[o] & 0xc0) == 0x80) return OO_ARROW;
    }
  }
  c->dict_n = n;
  return OO_OK;
}

oo_column* oo_column_new(const oo_column_desc* d, int* status) {
  oo_column* c = (oo_column*)calloc(1, sizeof(*c));
  c->d = *d;
  *status = OO_OK;
  int has_present = 0;
  c->r_present = open_stream(d, OO_S_PRESENT, &has_present);
  if (has_present) c->present = oo_bool_new(c->r_present); /* PresentDecoder::from_stripe: get_opt */
  int v = is_v2(d) ? 2 : 1;
  switch (d->orc_type) {
    case OO_T_BOOLEAN:
      c->r_data = open_stream(d, OO_S_DATA, NULL);
      c->bool_data = oo_bool_new(c->r_data);
      break;
    case OO_T_BYTE:
      c->r_data = open_stream(d, OO_S_DATA, NULL);
      c->byte_data = oo_byte_rle_new(c->r_data);
      break;
    case OO_T_SHORT:
    case OO_T_INT:
    case OO_T_LONG:
    case OO_T_DATE:
      c->r_data = open_stream(d, OO_S_DATA, NULL);
      c->int_data = oo_int_rle_new(c->r_data, v, 1, d->orc_type == OO_T_SHORT ? 16 : (d->orc_type == OO_T_LONG ? 64 : 32));
      break;
    case OO_T_FLOAT:
    case OO_T_DOUBLE:
      c->r_data = open_stream(d, OO_S_DATA, NULL);
      break;
    case OO_T_STRING:
    case OO_T_VARCHAR:
    case OO_T_CHAR:
    case OO_T_BINARY:
      c->r_length = open_stream(d, OO_S_LENGTH, NULL);
      c->int_length = oo_int_rle_new(c->r_length, v, 0, 64);
      if (d->orc_type != OO_T_BINARY && is_dict(d)) {
        c->dict_status = load_dictionary(c);
        if (c->dict_status) *status = c->dict_status; /* new_string_decoder fails: the stripe decoder cannot be built */
        c->r_data = open_stream(d, OO_S_DATA, NULL);
        c->int_data = oo_int_rle_new(c->r_data, v, 0, 64);
      } else {
        c->r_data = open_stream(d, OO_S_DATA, NULL);
      }
      break;
    case OO_T_DECIMAL:
      c->r_data = open_stream(d, OO_S_DATA, NULL);
      c->r_secondary = open_stream(d, OO_S_SECONDARY, NULL);
      c->int_secondary = oo_int_rle_new(c->r_secondary, v, 1, 32);
      break;
    case OO_T_TIMESTAMP:
    case OO_T_TIMESTAMP_INSTANT:
      c->r_data = open_stream(d, OO_S_DATA, NULL);
      c->int_data = oo_int_rle_new(c->r_data, v, 1, 64);
      c->r_secondary = open_stream(d, OO_S_SECONDARY, NULL);
      c->int_secondary = oo_int_rle_new(c->r_secondary, v, 0, 64);
      break;
    default:
      *status = OO_UNSUPPORTED;
  }
  return c;
}

void oo_column_free(oo_column* c) {
  if (!c) return;
  oo_bool_free(c->present);
  oo_bool_free(c->bool_data);
  oo_byte_rle_free(c->byte_data);
  oo_int_rle_free(c->int_data);
  oo_int_rle_free(c->int_length);
  oo_int_rle_free(c->int_secondary);
  oo_reader_free(c->r_present);
  oo_reader_free(c->r_data);
  oo_reader_free(c->r_length);
  oo_reader_free(c->r_secondary);
  oo_reader_free(c->r_dict);
  free(c->dict_bytes);
  free(c->dict_offsets);
  free(c->validity);
  free(c->values);
  free(c->offsets);
  free(c);
}

static size_t value_width(int orc_type) {
  switch (orc_type) {
    case OO_T_BYTE: return 1;
    case OO_T_SHORT: return 2;
    case OO_T_INT:
    case OO_T_DATE:
    case OO_T_FLOAT: return 4;
    case OO_T_LONG:
    case OO_T_DOUBLE:
    case OO_T_TIMESTAMP:
    case OO_T_TIMESTAMP_INSTANT: return 8;
    case OO_T_DECIMAL: return 16;
    default: return 0;
  }
}

/* decode k dense values of the column's primitive type into `dense` (k * width bytes) */
static int decode_dense(oo_column* c, uint8_t* dense, size_t k) {
  int st = OO_OK;
  switch (c->d.orc_type) {
    case OO_T_BYTE: return oo_byte_rle_decode(c->byte_data, (int8_t*)dense, k);
    case OO_T_SHORT:
    case OO_T_INT:
    case OO_T_DATE:
    case OO_T_LONG: {
      int64_t* tmp = (int64_t*)malloc((k + 1) * sizeof(int64_t));
      st = oo_int_rle_decode(c->int_data, tmp, k);
      if (!st) {
        size_t w = value_width(c->d.orc_type);
        for (size_t i = 0; i < k; i++) memcpy(dense + i * w, &tmp[i], w); /* little endian host */
      }
      free(tmp);
      return st;
    }
    case OO_T_FLOAT:
    case OO_T_DOUBLE: {
      /* float.rs:70-74 read_exact */
      size_t w = value_width(c->d.orc_type), want = k * w;
      uint8_t* p = dense;
      /* read through the public reader API: oo_stream pull of `want` bytes */
      extern size_t oo__reader_read(oo_reader*, uint8_t*, size_t);
      extern int oo__reader_status(const oo_reader*);
      size_t got = oo__reader_read(c->r_data, p, want);
      /* a block the codec rejects panics in the reference (compression.rs:317,322 unwrap); here it is
       * reported as Build*Decoder whatever the column type */
      if (got != want && oo__reader_status(c->r_data) == OO_BUILD_DECODER) return OO_BUILD_DECODER;
      return got == want ? OO_OK : OO_IO_ERROR;
    }
    case OO_T_DECIMAL: {
      uint64_t* v = (uint64_t*)malloc((k + 1) * 16);
      int64_t* sc = (int64_t*)malloc((k + 1) * sizeof(int64_t));
      st = oo_varint128_decode(c->r_data, v, k); /* decimal.rs:131-132: varint first, then scale */
      if (!st) st = oo_int_rle_decode(c->int_secondary, sc, k);
      if (!st)
        for (size_t i = 0; i < k; i++) {
          uint64_t o[2];
          oo_fix_i128_scale(&v[2 * i], c->d.scale, (int32_t)sc[i], o);
          memcpy(dense + i * 16, o, 16);
        }
      free(v);
      free(sc);
      /* array_decoder/decimal.rs:96-100: the batch then goes through with_precision_and_scale, which arrow-rs refuses (ArrowError) for a
       * precision outside 1..=38, a scale above 38 or -- positive -- above the precision; the reference takes the footer's numbers
       * `as u8` / `as i8` (decimal.rs:59-60).  A stream failure of the batch comes first. */
      if (!st) {
        const uint32_t pr = c->d.precision & 0xffu;
        const int sc8 = (int8_t)c->d.scale;
        if (pr == 0 || pr > 38 || sc8 > 38 || (sc8 > 0 && (uint32_t)sc8 > pr)) st = OO_ARROW;
      }
      return st;
    }
    case OO_T_TIMESTAMP:
    case OO_T_TIMESTAMP_INSTANT: {
      int64_t* a = (int64_t*)malloc((k + 1) * sizeof(int64_t));
      int64_t* b = (int64_t*)malloc((k + 1) * sizeof(int64_t));
      st = oo_int_rle_decode(c->int_data, a, k);
      if (!st) st = oo_int_rle_decode(c->int_secondary, b, k);
      if (!st)
        for (size_t i = 0; i < k && !st; i++) {
          int64_t o;
          st = oo_decode_timestamp(c->d.ts_base, a[i], b[i], c->d.ts_unit, &o);
          memcpy(dense + i * 8, &o, 8);
        }
      free(a);
      free(b);
      return st;
    }
    default: return OO_UNSUPPORTED;
  }
}

/* PresentDecoder::next_buffer + derive_present_vec (array_decoder/mod.rs:199-252): returns null_count, fills c->validity
 * (LSB-first).  `bools` receives one byte per row.  parent: the Struct / Union arm above the column says which of the n rows
 * exist at all (one byte per row; NULL: no parent, or a parent without nulls) --
 *   (own stream, parent)     the stream holds one bit per row the PARENT has: read that many, deal them out (merge_parent_present)
 *   (own stream, no parent)  n bits
 *   (no stream, parent)      the parent's
 *   (neither)                none
 * and in every case: a buffer without nulls is dropped, and an ERROR is dropped too (`_ => None`, mod.rs:247-251): the batch is
 * then decoded as if every row were present. */
static int next_present(oo_column* c, size_t n, const uint8_t* parent, uint8_t** bools_out, uint64_t* null_count) {
  *bools_out = NULL;
  *null_count = 0;
  if (!c->present && !parent) return OO_OK;
  uint8_t* bools = (uint8_t*)malloc(n + 1);
  if (c->present && parent) {
    size_t have = 0;
    for (size_t i = 0; i < n; i++) have += parent[i] != 0;
    uint8_t* own = (uint8_t*)malloc(have + 1);
    int st = oo_bool_decode(c->present, own, have);
    if (st) {
      free(own);
      free(bools);
      return OO_OK;
    }
    size_t j = 0;
    for (size_t i = 0; i < n; i++) bools[i] = parent[i] ? own[j++] : 0;
    free(own);
  } else if (c->present) {
    int st = oo_bool_decode(c->present, bools, n);
    if (st) {
      free(bools);
      return OO_OK;
    }
  } else {
    memcpy(bools, parent, n);
  }
  uint64_t nulls = 0;
  for (size_t i = 0; i < n; i++) nulls += !bools[i];
  if (nulls == 0) {
    free(bools);
    return OO_OK;
  }
  c->validity = (uint8_t*)grow(c->validity, &c->validity_cap, (n + 7) / 8 + 8);
  memset(c->validity, 0, (n + 7) / 8);
  for (size_t i = 0; i < n; i++)
    if (bools[i]) c->validity[i >> 3] |= (uint8_t)(1u << (i & 7));
  *bools_out = bools;
  *null_count = nulls;
  return OO_OK;
}

int oo_column_next_batch(oo_column* c, uint64_t batch_size, oo_batch* out) { return oo_column_next_batch_under(c, batch_size, NULL, out); }

int oo_column_next_batch_under(oo_column* c, uint64_t batch_size, const uint8_t* parent_present, oo_batch* out) {
  memset(out, 0, sizeof(*out));
  size_t n = (size_t)batch_size;
  out->length = n;
  if (c->dict_status) return out->status = c->dict_status;
  uint8_t* bools = NULL;
  uint64_t nulls = 0;
  int st = next_present(c, n, parent_present, &bools, &nulls);
  if (st) return out->status = st;
  size_t k = n - (size_t)nulls;
  out->null_count = nulls;
  out->validity = bools ? c->validity : NULL;
  int t = c->d.orc_type;

  if (t == OO_T_BOOLEAN) {
    /* BooleanArrayDecoder: dense bools spaced out, then packed LSB-first */
    uint8_t* dense = (uint8_t*)malloc(k + 1);
    st = k ? oo_bool_decode(c->bool_data, dense, k) : OO_OK;
    c->values = (uint8_t*)grow(c->values, &c->values_cap, (n + 7) / 8 + 8);
    memset(c->values, 0, (n + 7) / 8);
    if (!st) {
      size_t di = 0;
      for (size_t i = 0; i < n; i++) {
        int valid = bools ? bools[i] : 1;
        if (valid) {
          if (dense[di]) c->values[i >> 3] |= (uint8_t)(1u << (i & 7));
          di++;
        }
      }
    }
    free(dense);
    free(bools);
    out->values = c->values;
    out->values_len = (n + 7) / 8;
    return out->status = st;
  }

  size_t w = value_width(t);
  if (w) {
    c->values = (uint8_t*)grow(c->values, &c->values_cap, n * w + 16);
    memset(c->values, 0, n * w);
    if (!bools) {
      st = decode_dense(c, c->values, n);
    } else if (k) {
      uint8_t* dense = (uint8_t*)malloc(k * w + 16);
      st = decode_dense(c, dense, k);
      if (!st) {
        size_t di = 0;
        for (size_t i = 0; i < n; i++)
          if (bools[i]) memcpy(c->values + i * w, dense + (di++) * w, w);
      }
      free(dense);
    }
    free(bools);
    out->values = c->values;
    out->values_len = n * w;
    return out->status = st;
  }

  /* strings / binary */
  if (t == OO_T_STRING || t == OO_T_VARCHAR || t == OO_T_CHAR || t == OO_T_BINARY) {
    int64_t* vals = (int64_t*)calloc(n + 1, sizeof(int64_t)); /* lengths or keys, spaced, 0 at nulls */
    oo_int_rle* dec = (t != OO_T_BINARY && is_dict(&c->d)) ? c->int_data : c->int_length;
    if (!bools) {
      st = oo_int_rle_decode(dec, vals, n);
    } else if (k) {
      int64_t* dense = (int64_t*)malloc((k + 1) * sizeof(int64_t));
      st = oo_int_rle_decode(dec, dense, k);
      if (!st) {
        size_t di = 0;
        for (size_t i = 0; i < n; i++)
          if (bools[i]) vals[i] = dense[di++];
      }
      free(dense);
    }
    if (st) {
      free(vals);
      free(bools);
      return out->status = st;
    }
    c->offsets = (int32_t*)grow(c->offsets, &c->offsets_cap, (n + 1) * sizeof(int32_t));
    if (t != OO_T_BINARY && is_dict(&c->d)) {
      /* DictionaryArray::try_new bounds-checks non-null keys, cast gathers (string.rs:204-224) */
      int64_t total = 0;
      for (size_t i = 0; i < n && !st; i++) {
        int valid = bools ? bools[i] : 1;
        if (valid) {
          if (vals[i] < 0 || (uint64_t)vals[i] >= c->dict_n) st = OO_ARROW;
          else total += c->dict_offsets[vals[i] + 1] - c->dict_offsets[vals[i]];
        }
      }
      if (!st && total > INT32_MAX) st = OO_ARROW; /* arrow cast: offset overflow */
      if (!st) {
        c->values = (uint8_t*)grow(c->values, &c->values_cap, (size_t)total + 16);
        int32_t acc = 0;
        c->offsets[0] = 0;
        for (size_t i = 0; i < n; i++) {
          int valid = bools ? bools[i] : 1;
          if (valid) {
            int32_t a = c->dict_offsets[vals[i]], b = c->dict_offsets[vals[i] + 1];
            memcpy(c->values + acc, c->dict_bytes + a, (size_t)(b - a));
            acc += b - a;
          }
          c->offsets[i + 1] = acc;
        }
        out->values_len = (uint64_t)acc;
      }
    } else {
      int64_t total = 0;
      for (size_t i = 0; i < n; i++) total += vals[i];
      if (total > INT32_MAX) st = OO_OFFSET_OVERFLOW;
      if (!st) {
        for (size_t i = 0; i < n; i++)
          if (vals[i] < 0) st = OO_ARROW;
      }
      if (!st) {
        c->values = (uint8_t*)grow(c->values, &c->values_cap, (size_t)total + 16);
        extern size_t oo__reader_read(oo_reader*, uint8_t*, size_t);
        size_t got = oo__reader_read(c->r_data, c->values, (size_t)total);
        int64_t acc = 0;
        c->offsets[0] = 0;
        for (size_t i = 0; i < n; i++) {
          acc += vals[i];
          c->offsets[i + 1] = (int32_t)acc;
        }
        extern int oo__reader_status(const oo_reader*);
        if (got < (size_t)total) {
          /* a stream cut short by its container (rejected block / broken framing) reports that; a stream that is
           * simply too short is not an IoError for read_to_end -- try_new then rejects offsets past the buffer */
          int rs = oo__reader_status(c->r_data);
          st = rs ? rs : OO_ARROW;
        }
        if (!st && t != OO_T_BINARY) {
          if (!utf8_valid(c->values, (size_t)total)) st = OO_ARROW;
          for (size_t i = 0; i <= n && !st; i++) {
            int32_t o = c->offsets[i];
            if (o < total && (c->values[o] & 0xc0) == 0x80) st = OO_ARROW;
          }
        }
        out->values_len = (uint64_t)total;
      }
    }
    free(vals);
    free(bools);
    out->values = c->values;
    out->offsets = c->offsets;
    return out->status = st;
  }
  free(bools);
  return out->status = OO_UNSUPPORTED;
}

/* ---- writer time zone (array_decoder/timestamp.rs:236-291, :316-349) ------------------------------------ */
static int32_t tz_offset(const int64_t* at, const int32_t* offs, uint32_t n_at, int32_t offs0, int64_t fold_at, int64_t sec) {
  /* chrono-tz: the span whose start is <= the instant (binary search over the zone's timespans).  A zone that ends in a
   * daylight-saving RULE has spans without end; the table lists them up to fold_at, and an instant at or behind fold_at is
   * looked up whole 400-year cycles earlier (the Gregorian calendar repeats after 146 097 days, and so does every rule that
   * names a month, a week and a weekday); INT64_MAX for a zone whose last offset holds for ever. */
  if (sec >= fold_at) sec -= ((sec - fold_at) / (146097ll * 86400) + 1) * (146097ll * 86400);
  uint32_t lo = 0, hi = n_at;
  while (lo < hi) {
    uint32_t mid = lo + (hi - lo) / 2;
    if (at[mid] <= sec) lo = mid + 1;
    else hi = mid;
  }
  return lo ? offs[lo - 1] : offs0;
}

static int64_t floor_div(int64_t a, int64_t b) {
  int64_t q = a / b;
  return (a % b < 0) ? q - 1 : q;
}

uint64_t oo_timestamps_to_utc(int64_t* values, const uint8_t* validity, uint64_t n, int unit, const int64_t* at, const int32_t* offs,
                              uint32_t n_at, int32_t offs0, int64_t fold_at, uint8_t* validity_out) {
  /* DateTime::<Utc>::MIN_UTC / MAX_UTC = NaiveDate::MIN (-262143-01-01) 00:00:00 / NaiveDate::MAX (+262142-12-31) 23:59:59.999999999 */
  const int64_t chrono_min = -8334601315200ll, chrono_max = 8210266876799ll;
  uint64_t nulls = 0;
  memset(validity_out, 0, (n + 7) / 8);
  for (uint64_t i = 0; i < n; i++) {
    if (validity && !((validity[i >> 3] >> (i & 7)) & 1)) {
      nulls++;
      continue; /* try_unary / unary_opt only visit valid slots */
    }
    int64_t ts = values[i], out = 0;
    int ok;
    if (unit == 3) {
      /* writer_tz.timestamp_nanos(ts).naive_local().and_utc().timestamp_nanos_opt() */
      int64_t sec = floor_div(ts, 1000000000);
      __int128 r = (__int128)ts + (__int128)tz_offset(at, offs, n_at, offs0, fold_at, sec) * 1000000000;
      ok = r <= (__int128)INT64_MAX && r >= (__int128)INT64_MIN;
      out = (int64_t)r;
    } else {
      /* writer_tz.timestamp_micros(ts * k).single().map(|dt| dt.naive_local().and_utc().timestamp_micros() / k) */
      int64_t k = unit == 0 ? 1000000 : (unit == 1 ? 1000 : 1);
      int64_t m = (int64_t)((uint64_t)ts * (uint64_t)k); /* release build: wrapping */
      int64_t sec = floor_div(m, 1000000);
      ok = sec >= chrono_min && sec <= chrono_max;
      if (ok) out = (m + (int64_t)tz_offset(at, offs, n_at, offs0, fold_at, sec) * 1000000) / k;
    }
    if (ok) {
      values[i] = out;
      validity_out[i >> 3] |= (uint8_t)(1u << (i & 7));
    } else {
      values[i] = 0;
      nulls++;
    }
  }
  return nulls;
}

void oo_timestamp_decimals_to_utc(uint64_t* values, uint64_t n, const int64_t* at, const int32_t* offs, uint32_t n_at, int32_t offs0, int64_t fold_at) {
  for (uint64_t i = 0; i < n; i++) {
    __int128 ts = (__int128)(((unsigned __int128)values[2 * i + 1] << 64) | values[2 * i]);
    __int128 q = ts / 1000000000;
    if (ts % 1000000000 < 0) q--; /* div_euclid */
    __int128 r = ts + (__int128)tz_offset(at, offs, n_at, offs0, fold_at, (int64_t)q) * 1000000000;
    values[2 * i] = (uint64_t)(unsigned __int128)r;
    values[2 * i + 1] = (uint64_t)((unsigned __int128)r >> 64);
  }
}
