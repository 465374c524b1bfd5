// column_kernels.hip -- PRESENT -> Arrow validity, null spacing and the per-type finishers.
//
//   PresentDecoder / BooleanDecoder bits are MSB-first (boolean.rs:48-54, :101-113); Arrow
//   validity is LSB-first, omitted per batch when the batch has no nulls (array_decoder/mod.rs:247-251).
//   decode_spaced leaves null slots at the caller's zero fill (encoding/mod.rs:64-91).
//   Timestamp combine: encoding/timestamp.rs:121-192.  Decimal scale repair: array_decoder/decimal.rs:138-166.
#include "rle_kernels.h"
#include "rle_parse.h"

struct ErrSlot {
  unsigned long long v;  // min over (row or value index << 8 | code)
};

__device__ __forceinline__ void report_row(unsigned long long* err, uint64_t idx, uint32_t code) {
  atomicMin(err, ((unsigned long long)idx << 8) | code);
}

// ------------------------------------------------------------------------------------------------
// ---- PRESENT streams of ALL columns of a call, one launch per step ----------------------------------------
// (hundreds of columns x stripes per call: per-column launches of these tiny kernels would cost more
// than the work; blockIdx.y selects the column, blockIdx.x the piece of it)
struct PresJob {
  const uint8_t* pbytes;           // decoded PRESENT bytes (MSB first)
  unsigned long long* vbits;       // stripe-wide validity words (LSB first)
  uint32_t* wpop;                  // popcount per word
  uint32_t* rank;                  // exclusive scan of wpop: non-null rows before each word
  uint32_t* rtiles;                // per 1024-word tile sums
  unsigned long long* validity;    // per-batch bitmaps in the result
  unsigned long long* null_counts; // per batch
  uint64_t* nonnull_out;           // scalar: non-null rows of the column
  uint64_t* ceil8_out;             // BOOLEAN columns: ceil(nonnull / 8) (bytes of the DATA job), else null
  const RleJob* job;               // the PRESENT byte-RLE job (its error word says where decoding stopped); null: the column has no PRESENT stream of its own
  uint64_t n_rows, n_words, n_rank_tiles, n_out_words;
  uint32_t batch, words_per_batch;
  // a child of a Struct: its PRESENT stream has one bit per row in which the PARENT is present (array_decoder/mod.rs:216-252,
  // merge_parent_present); the column's validity over the stripe's rows is those bits dealt out to the parent's set bits
  const unsigned long long* parent_vbits;  // the parent's stripe-wide validity words (null: a root column)
  const uint32_t* parent_rank;             // ... and the non-null rows before each of them
  uint64_t* ceil8b_out;            // Struct columns: ceil(nonnull / 8) = bytes of their children's PRESENT streams, else null
  // one arm of a Union (array_decoder/union.rs:69-136): "present" where the Union is and its tag names this arm -- the validity
  // the arm's child is decoded under.  tags: the Union's dense tags (one per row in which the Union is present: parent_vbits /
  // parent_rank are then the Union's own words and ranks, null when it has no nulls)
  const int8_t* tags;
  int32_t tag;
  // streams entered at a row group in mid-byte (orcgpu_stream::skip_bits): bits of the first byte of this column's PRESENT stream
  // that belong to the rows before (bit0), of its Boolean DATA stream (data_bits), of its fields' PRESENT streams (child_bits)
  uint32_t bit0, data_bits, child_bits;
};

__device__ __forceinline__ void present_word(const PresJob& j, uint64_t w) {
  unsigned long long v;
  uint64_t rows_here = j.n_rows - w * 64;
  if (j.tags) {
    const unsigned long long pv = j.parent_vbits ? j.parent_vbits[w] : ~0ull;
    uint64_t at = j.parent_rank ? j.parent_rank[w] : w * 64;
    const uint32_t nrow = rows_here < 64 ? (uint32_t)rows_here : 64u;
    v = 0;
    for (uint32_t b = 0; b < nrow; b++)
      if ((pv >> b) & 1) {
        if (j.tags[at] == (int8_t)j.tag) v |= 1ull << b;
        at++;
      }
  } else if (!j.parent_vbits) {
    uint64_t x = ld_u64(j.pbytes + w * 8);
    // reverse the bits inside each byte: bitreverse64 reverses everything, bswap restores byte order
    v = __builtin_bswap64(__builtin_bitreverse64(x));
    if (j.bit0) v = (v >> j.bit0) | (__builtin_bswap64(__builtin_bitreverse64(ld_u64(j.pbytes + w * 8 + 8))) << (64 - j.bit0));  // (slack behind the stream)
    const unsigned long long e = j.job->err;
    if (e != RLE_NO_ERR) {
      const uint64_t got = (e >> 8) * 8;
      const uint64_t bits = got > j.bit0 ? got - j.bit0 : 0;
      if (bits < j.n_rows) {
        const uint64_t cutoff = bits / j.batch * j.batch;  // first row of the batch that fails
        if (w * 64 + 64 > cutoff) v |= w * 64 >= cutoff ? ~0ull : ~0ull << (cutoff - w * 64);
      }
    }
  } else {
    const unsigned long long pv = j.parent_vbits[w];
    if (!j.pbytes) {
      v = pv;  // no PRESENT stream of its own: null exactly where the parent is (derive_present_vec: (None, Some(parent)))
    } else {
      // the column's bits for this word: popcount(pv) of them, from bit parent_rank[w] of its own stream on
      const uint64_t r = (uint64_t)j.parent_rank[w] + j.bit0;
      const uint64_t a = __builtin_bswap64(__builtin_bitreverse64(ld_u64(j.pbytes + (r >> 3))));
      const uint64_t b = __builtin_bswap64(__builtin_bitreverse64(ld_u64(j.pbytes + (r >> 3) + 8)));  // (slack behind the stream)
      const uint32_t sh = (uint32_t)(r & 7);
      unsigned long long c = sh ? (a >> sh) | (b << (64 - sh)) : a;
      const unsigned long long e = j.job->err;
      if (e != RLE_NO_ERR) {  // (the stream failed to decode: the bits it did not deliver read as present, like a root column's)
        const uint64_t bits = (e >> 8) * 8;  // (counted from the stream's first byte, like r)
        if (bits < r + 64) c |= bits <= r ? ~0ull : ~0ull << (bits - r);
      }
      // deal them out to the set bits of the parent's word, lowest first
      v = 0;
      for (unsigned long long m = pv; m; m &= m - 1, c >>= 1)
        if (c & 1) v |= m & (0 - m);
    }
  }
  if (rows_here < 64) v &= (1ull << rows_here) - 1;
  j.vbits[w] = v;
  j.wpop[w] = (uint32_t)__builtin_popcountll(v);
}
// PRESENT bytes -> validity words + popcounts.  A PRESENT stream that fails to decode does not fail the
// column: the reference swallows the error and decodes the batch -- and, its decoder being at the end
// of its input, every later batch -- as if there were no PRESENT stream (derive_present_vec,
// array_decoder/mod.rs:228-251, `_ => None`).  job->err holds the number of bytes decoded before the
// failing run: rows from the first batch those bits do not cover are valid.
extern "C" __global__ void __launch_bounds__(256) pres_words_kernel(const PresJob* jobs) {
  const PresJob j = jobs[blockIdx.y];
  uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w < j.n_words) present_word(j, w);
}
extern "C" __global__ void __launch_bounds__(256) pres_scan_tiles_kernel(const PresJob* jobs) {
  __shared__ uint32_t wsum[4];
  const PresJob j = jobs[blockIdx.y];
  if (blockIdx.x >= j.n_rank_tiles) return;
  const uint64_t n = j.n_words;
  uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t v[4];
  for (int k = 0; k < 4; k++) v[k] = base + k < n ? j.wpop[base + k] : 0;
  uint32_t s = v[0] + v[1] + v[2] + v[3];
  uint32_t incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
  uint32_t e = wbase + incl - s;
  for (int k = 0; k < 4; k++) {
    if (base + k < n) j.rank[base + k] = e;
    e += v[k];
  }
  if (threadIdx.x == 255) j.rtiles[blockIdx.x] = e;
}
// one workgroup per column: exclusive scan of the tile sums in place, non-null total
extern "C" __global__ void __launch_bounds__(256) pres_scan_sums_kernel(const PresJob* jobs) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  const PresJob j = jobs[blockIdx.x];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t s = 0; s < j.n_rank_tiles; s += 256) {
    uint64_t i = s + threadIdx.x;
    uint64_t v = i < j.n_rank_tiles ? j.rtiles[i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    if (i < j.n_rank_tiles) j.rtiles[i] = (uint32_t)(wbase + incl - v);
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *j.nonnull_out = carry_s;
    if (j.ceil8_out) *j.ceil8_out = (carry_s + j.data_bits + 7) / 8;
    if (j.ceil8b_out) *j.ceil8b_out = (carry_s + j.child_bits + 7) / 8;
  }
}
extern "C" __global__ void __launch_bounds__(256) pres_scan_apply_kernel(const PresJob* jobs) {
  const PresJob j = jobs[blockIdx.y];
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < j.n_words) j.rank[i] += j.rtiles[i >> 10];
}

// Generic 2-level exclusive scan of u32 counts (tile = 1024 entries / workgroup).
__device__ __forceinline__ void scan_tiles_body(const uint32_t* in, uint32_t* out, uint32_t* tile_sum, uint64_t n) {
  __shared__ uint32_t wsum[4];
  uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t v[4];
  for (int k = 0; k < 4; k++) v[k] = base + k < n ? in[base + k] : 0;
  uint32_t s = v[0] + v[1] + v[2] + v[3];
  uint32_t incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
  uint32_t e = wbase + incl - s;
  for (int k = 0; k < 4; k++) {
    if (base + k < n) out[base + k] = e;
    e += v[k];
  }
  if (threadIdx.x == 255) tile_sum[blockIdx.x] = e;
}
// single workgroup: exclusive scan of tile sums in place; total -> *total_out (u64)
__device__ __forceinline__ void scan_sums_body(uint32_t* tile_sum, uint64_t ntiles, uint64_t* total_out) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t s = 0; s < ntiles; s += 256) {
    uint64_t i = s + threadIdx.x;
    uint64_t v = i < ntiles ? tile_sum[i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    if (i < ntiles) tile_sum[i] = (uint32_t)(wbase + incl - v);
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}
__device__ __forceinline__ void scan_apply_body(uint32_t* out, const uint32_t* tile_sum, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] += tile_sum[i >> 10];
}

// Per-batch validity bitmaps + null counts of all columns of a call (blockIdx.y = column).  One thread per
// output word of a batch; out layout: batch b at word offset b * words_per_batch; one atomic per wavefront
// when its 64 words belong to one batch (the common case).
extern "C" __global__ void __launch_bounds__(256) pres_validity_kernel(const PresJob* jobs) {
  const PresJob j = jobs[blockIdx.y];
  if ((uint64_t)blockIdx.x * 256 >= j.n_out_words) return;
  uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < j.n_out_words;
  if (!live) t = j.n_out_words - 1;  // stays in the wavefront for the reduction below, contributes nothing
  uint64_t b = t / j.words_per_batch, w = t % j.words_per_batch;
  uint64_t row0 = b * j.batch + w * 64;
  uint64_t bend = (b + 1) * (uint64_t)j.batch;
  if (bend > j.n_rows) bend = j.n_rows;
  unsigned long long v = 0;
  uint32_t rows = 0, nulls = 0;
  if (row0 < bend) {
    rows = bend - row0 < 64 ? (uint32_t)(bend - row0) : 64;
    uint64_t sw = row0 >> 6;
    uint32_t sh = row0 & 63;
    unsigned long long lo = j.vbits[sw];
    unsigned long long hi = (sh && ((sw + 1) * 64 < j.n_rows)) ? j.vbits[sw + 1] : 0;
    v = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    if (rows < 64) v &= (1ull << rows) - 1;
    nulls = live ? rows - (uint32_t)__builtin_popcountll(v) : 0;
  }
  if (live) j.validity[t] = v;
  const uint64_t b0 = __shfl((unsigned long long)b, 0);
  if (__ballot(b != b0) == 0) {
    for (int o = 32; o; o >>= 1) nulls += __shfl_xor(nulls, o);
    if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(&j.null_counts[b0], (unsigned long long)nulls);
  } else if (nulls) {
    atomicAdd(&j.null_counts[b], (unsigned long long)nulls);
  }
}

// Null spacing for fixed-width values: out[i] = valid(i) ? dense[rank(i)] : 0   (encoding/mod.rs:64-91)
// `limit`: dense values there are to read.  (Float / Double values are spaced straight out of their stream: a stream that is shorter
// than its column's non-null rows -- or missing: a stripe footer that lost it, tests/test_gpu_containers.py -- fails the column
// (float_check_body), and nothing behind its end may be touched on the way there.)
template <typename T>
__device__ __forceinline__ void space_body(const T* dense, const unsigned long long* vbits, const uint32_t* rank, T* out, uint64_t n_rows, uint64_t limit) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  unsigned long long word = vbits[i >> 6];
  uint32_t bit = i & 63;
  T v = T(0);
  if ((word >> bit) & 1) {
    const uint64_t at = (uint64_t)rank[i >> 6] + __builtin_popcountll(word & ((1ull << bit) - 1));
    if (at < limit) v = dense[at];
  }
  out[i] = v;
}

// all fixed-width columns of a call in one launch (blockIdx.y = column)
struct SpaceJob {
  const void* dense;
  const unsigned long long* vbits;
  const uint32_t* rank;
  void* out;
  uint64_t n_rows;
  uint32_t width, pad;
  uint64_t limit;  // dense values that may be read (a dense buffer of the decoders holds one per row: ~0)
};
extern "C" __global__ void __launch_bounds__(256) space_multi_kernel(const SpaceJob* jobs) {
  const SpaceJob j = jobs[blockIdx.y];
  if ((uint64_t)blockIdx.x * 256 >= j.n_rows) return;
  if (j.width == 8) space_body((const int64_t*)j.dense, j.vbits, j.rank, (int64_t*)j.out, j.n_rows, j.limit);
  else if (j.width == 4) space_body((const int32_t*)j.dense, j.vbits, j.rank, (int32_t*)j.out, j.n_rows, j.limit);
  else if (j.width == 2) space_body((const int16_t*)j.dense, j.vbits, j.rank, (int16_t*)j.out, j.n_rows, j.limit);
  else space_body((const int8_t*)j.dense, j.vbits, j.rank, (int8_t*)j.out, j.n_rows, j.limit);
}

// The summary of a decode call (scalars, job records, null counts: some KB) goes to the host through this kernel -- stores into
// the pinned, device-visible mirror -- instead of a device-to-host copy: a copy would queue behind the copies back of earlier
// results on the same DMA engine (hundreds of MB each) and hold the decode up for as long as they take.
extern "C" __global__ void __launch_bounds__(256) summary_to_host_kernel(const unsigned long long* src, unsigned long long* host, uint64_t n_words) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_words) __builtin_nontemporal_store(src[i], host + i);
}

// Union: a type id that names no arm cannot become a UnionArray (union.rs:126-129: UnionArray::try_new -> ArrowError)
// (tags: the dense ones, one per row in which the Union is present; the error names the first such value)
__device__ __forceinline__ void union_tags_body(const int8_t* tags, const uint64_t* n_tags, int32_t n_arms, unsigned long long* err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < *n_tags && (tags[i] < 0 || tags[i] >= n_arms)) atomicMin(err, ((unsigned long long)i << 8) | ORC_E_ARROW);
}

// Float/Double without nulls: plain copy of the raw little-endian stream (float.rs:70-74).
__device__ __forceinline__ void copy_bytes_body(const uint8_t* src, uint8_t* dst, uint64_t n) {
  uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i + 16 <= n) {
    uint4 v;
    __builtin_memcpy(&v, src + i, 16);
    *reinterpret_cast<uint4*>(dst + i) = v;
  } else {
    for (uint64_t k = i; k < n; k++) dst[k] = src[k];
  }
}

// Boolean DATA: dense MSB-first bit bytes -> per-batch LSB-first value bitmaps (BooleanArrayDecoder,
// array_decoder/mod.rs:163-183).  One thread per output word; with nulls the dense bits are
// deposited into the valid positions.
__device__ __forceinline__ void bool_values_body(const uint8_t* dbytes, const unsigned long long* vbits, const uint32_t* rank,
                                                                      uint64_t n_rows, uint32_t batch, uint32_t words_per_batch,
                                                                      unsigned long long* out, uint64_t n_out_words, uint32_t bit0) {
  uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < n_out_words;
  if (!live) t = n_out_words - 1;  // stays in the wavefront for the reduction below, contributes nothing
  uint64_t b = t / words_per_batch, w = t % words_per_batch;
  uint64_t row0 = b * batch + w * 64;
  uint64_t bend = (b + 1) * (uint64_t)batch;
  if (bend > n_rows) bend = n_rows;
  unsigned long long v = 0;
  if (row0 < bend) {
    uint32_t rows = bend - row0 < 64 ? (uint32_t)(bend - row0) : 64;
    for (uint32_t k = 0; k < rows; k++) {
      uint64_t row = row0 + k;
      uint64_t d = row;
      bool valid = true;
      // (bit0: a stream entered at a row group in mid-byte -- the bits of its first byte that belong to the rows before)
      if (vbits) {
        unsigned long long word = vbits[row >> 6];
        uint32_t bit = row & 63;
        valid = (word >> bit) & 1;
        d = (uint64_t)rank[row >> 6] + __builtin_popcountll(word & ((1ull << bit) - 1));
      }
      d += bit0;
      if (valid && ((dbytes[d >> 3] >> (7 - (d & 7))) & 1)) v |= 1ull << k;
    }
  }
  out[t] = v;
}

// Timestamp combine (encoding/timestamp.rs:121-192) fused with null spacing.
// unit: 0 s, 1 ms, 2 us, 3 ns.
__device__ __forceinline__ void timestamp_body(const int64_t* secs, const int64_t* nanos, const unsigned long long* vbits,
                                                                    const uint32_t* rank, int64_t* out, uint64_t n_rows, int64_t base,
                                                                    int unit, unsigned long long* err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  uint64_t d = i;
  bool valid = true;
  if (vbits) {
    unsigned long long word = vbits[i >> 6];
    uint32_t bit = i & 63;
    valid = (word >> bit) & 1;
    d = (uint64_t)rank[i >> 6] + __builtin_popcountll(word & ((1ull << bit) - 1));
  }
  int64_t r = 0;
  if (valid) {
    uint64_t nn = (uint64_t)nanos[d];
    uint32_t zeros = nn & 7;
    nn >>= 3;
    if (zeros) {
      uint64_t p = 10;
      for (uint32_t k = 0; k < zeros; k++) p *= 10;
      nn *= p;  // wrapping, as the release build of the reference
    }
    int64_t sse = (int64_t)((uint64_t)secs[d] + (uint64_t)base);
    int64_t s = (sse < 0 && nn > 999999) ? sse - 1 : sse;
    // ns = s * 1e9 + nn, then ns / per with `per` (1, 1e3, 1e6 or 1e9 ns per unit) dividing 1e9: the
    // remainder of ns is that of nn, and an exact quotient is s * (1e9 / per) + nn / per -- no 128-bit
    // division (which costs hundreds of instructions per row), constants per unit
    uint64_t qn, rem;
    int64_t m;
    if (unit == 4) {
      // Decimal128(38, 9) target: the raw nanoseconds as i128, nothing to lose or overflow (encoding/timestamp.rs:78-119)
      reinterpret_cast<__int128*>(out)[i] = (__int128)s * 1000000000 + (__int128)nn;
      return;
    }
    if (unit == 3) {
      qn = nn, rem = 0, m = 1000000000;
    } else if (unit == 2) {
      qn = nn / 1000, rem = nn - qn * 1000, m = 1000000;
    } else if (unit == 1) {
      qn = nn / 1000000, rem = nn - qn * 1000000, m = 1000;
    } else {
      qn = nn / 1000000000, rem = nn - qn * 1000000000, m = 1;
    }
    const __int128 q = (__int128)s * m + (__int128)qn;
    bool bad = rem != 0 || q > (__int128)INT64_MAX || q < (__int128)INT64_MIN;
    if (bad) report_row(err, i, ORC_E_TIMESTAMP);
    r = (int64_t)q;
  }
  if (unit == 4) reinterpret_cast<__int128*>(out)[i] = 0;  // null slot
  else out[i] = r;
}

// Float/Double: read_exact of `needed` values must fit in the stream (float.rs:70-74 -> IoError).
__device__ __forceinline__ void float_check_body(const uint64_t* scalars, uint32_t len_idx, uint32_t needed_idx, uint32_t width, uint64_t* err) {
  if (threadIdx.x == 0) {
    uint64_t have = scalars[len_idx] / width;
    if (have < scalars[needed_idx]) atomicMin((unsigned long long*)err, ((unsigned long long)have << 8) | ORC_E_IO);
  }
}

// Writer time zone -> UTC re-labelling of decoded TIMESTAMP values (array_decoder/timestamp.rs:236-291, :316-349): the instant
// is looked at in the writer's zone and its wall clock read as UTC, i.e. out = ts + offset(ts) with the zone's UTC offset at
// that instant (table of transitions, binary search).  Per unit, as the release build of the reference computes it:
//   s / ms / us  m = ts * k microseconds (wrapping); chrono must hold the instant (years -262143 ..= 262142) or the value
//                becomes a null (try_unary fails -> unary_opt); out = (m + offset * 1e6) / k, truncating.
//   ns           out = ts + offset * 1e9, a null when that leaves i64 (timestamp_nanos_opt).
//   Decimal128   i128 nanoseconds + offset * 1e9 (no nulls).
// A null clears the row's validity bit and raises the batch's null count; the validity words exist for every column that gets
// here (all ones when the column has no PRESENT stream).
struct TzJob {
  void* values;
  const unsigned long long* vbits;   // stripe-wide PRESENT bitmap or null
  unsigned long long* validity;      // per batch
  unsigned long long* null_counts;
  const long long* at;
  const int* offs;
  uint64_t n_rows;
  uint32_t n_at;
  int offs0;
  int unit;                          // 0 s, 1 ms, 2 us, 3 ns, 4 Decimal128(38, 9)
  uint32_t batch, words_per_batch, pad;
  long long fold_at;                 // instants at or behind it are looked up whole 400-year cycles earlier (orcgpu_tz.inc: TzTable::fold_at)
};

__device__ __forceinline__ int tz_offset_at(const TzJob& j, long long sec) {
  if (sec >= j.fold_at) sec -= ((sec - j.fold_at) / (146097ll * 86400) + 1) * (146097ll * 86400);
  uint32_t lo = 0, hi = j.n_at;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (j.at[mid] <= sec) lo = mid + 1;
    else hi = mid;
  }
  return lo ? j.offs[lo - 1] : j.offs0;
}

extern "C" __global__ void __launch_bounds__(256) tz_shift_kernel(TzJob j) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= j.n_rows) return;
  if (j.vbits && !((j.vbits[i >> 6] >> (i & 63)) & 1)) return;  // null row: stays 0
  bool ok = true;
  if (j.unit == 4) {
    __int128* v = reinterpret_cast<__int128*>(j.values) + i;
    const __int128 ts = *v;
    __int128 q = ts / 1000000000;
    if (ts % 1000000000 < 0) q--;
    *v = ts + (__int128)tz_offset_at(j, (long long)q) * 1000000000;
    return;
  }
  long long* v = reinterpret_cast<long long*>(j.values) + i;
  const long long ts = *v;
  if (j.unit == 3) {
    long long sec = ts / 1000000000;
    if (ts % 1000000000 < 0) sec--;
    const __int128 r = (__int128)ts + (__int128)tz_offset_at(j, sec) * 1000000000;
    ok = r <= (__int128)INT64_MAX && r >= (__int128)INT64_MIN;
    if (ok) *v = (long long)r;
  } else {
    const long long k = j.unit == 0 ? 1000000 : (j.unit == 1 ? 1000 : 1);
    const long long m = (long long)((unsigned long long)ts * (unsigned long long)k);
    long long sec = m / 1000000;
    if (m % 1000000 < 0) sec--;
    // chrono's NaiveDate::MIN = -262143-01-01 and MAX = +262142-12-31
    ok = sec >= -8334601315200ll && sec <= 8210266876799ll;
    if (ok) *v = (m + (long long)tz_offset_at(j, sec) * 1000000) / k;
  }
  if (!ok) {
    *v = 0;
    const uint64_t b = i / j.batch, l = i % j.batch;
    atomicAnd(&j.validity[b * j.words_per_batch + (l >> 6)], ~(1ull << (l & 63)));
    atomicAdd(&j.null_counts[b], 1ull);
  }
}

// all-ones validity words for a column without PRESENT whose values may still turn into nulls (tz_shift_kernel)
__device__ __forceinline__ void validity_ones_body(unsigned long long* validity, uint64_t n_rows, uint32_t batch, uint32_t words_per_batch,
                                                                        uint64_t n_words) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_words) return;
  const uint64_t b = t / words_per_batch, w = t % words_per_batch;
  const uint64_t row0 = b * batch + w * 64;
  uint64_t bend = (b + 1) * (uint64_t)batch;
  if (bend > n_rows) bend = n_rows;
  unsigned long long v = 0;
  if (row0 < bend) v = bend - row0 >= 64 ? ~0ull : (1ull << (bend - row0)) - 1;
  validity[t] = v;
}
