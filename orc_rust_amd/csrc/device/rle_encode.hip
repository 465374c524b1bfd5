// rle_encode.hip -- the reference's value ENCODERS on the device, byte for byte ("GPU ORC encode").
//
// Replaces RleV2Encoder<N, S>::{write_slice, take_inner} (src/encoding/integer/rle_v2/mod.rs:255-531), ByteRleEncoder
// (src/encoding/byte.rs:38-197) and BooleanEncoder::finish (src/encoding/boolean.rs:157-169); the seam is PrimitiveValueEncoder
// (src/encoding/mod.rs:36-50), driven per column by src/writer/column.rs and flushed per stripe by src/writer/stripe.rs:109-165.
//
// Both encoders are greedy state machines over single values.  What makes them parallel: behind every run they write, their
// state is "empty at position p" for some p (One(v) is Empty one value earlier, FixedRun{v, 3} is Empty two values earlier), so
// where a run started at p ends is a function next(p) of the values at p .. p + 513 alone:
//   E(p)  = how many equal values stand at p;  T(p) = the first t >= p with three equal values at t, t + 1, t + 2
//   E(p) >= 3:  a run of repeats, next = p + min(E(p), MAXFIX)                 (SHORT_REPEAT / fixed DELTA; byte RLE: a Run)
//   else:       literals, next = T(p) when T(p) - p <= MAXVAR - 3, else p + MAXVAR   (never past the end of the values)
// with MAXFIX = MAXVAR = 512 for Integer RLE v2, 130 / 128 for byte RLE.  The runs of a column are the orbit of 0 under next:
//   1. enc_next_kernel: a workgroup per tile of 512 positions finds E and T from two bitmaps (ballots) and next(p); pointer
//      doubling in LDS turns the tile into its EXIT MAP (enter at offset e -> the offset in the following tile the chain lands on);
//   2. enc_compose_kernel: 32 maps at a time are composed into the map of their span, level by level, until <= 32 are left;
//      enc_entries_kernel walks back down: every tile learns where the chain enters it -- exactly, whatever the values are;
//   3. enc_runs_kernel: a wavefront per tile follows the chain inside its tile (next(p) from LDS): first counted, then listed;
//   4. Integer RLE v2 -- enc2_plan_kernel: a wavefront per run applies determine_variable_run_encoding's rules (delta check,
//      percentile histograms in f32 as the reference computes them, patch list) -> sub-encoding and size;
//      a scan gives every run its place; enc2_emit_kernel writes it: lane l packs values 8 l .. 8 l + 7 = `width` whole bytes.
//      Byte RLE -- sizes follow from the run list alone; enc1_emit_kernel copies literals / writes runs.
// The integer type N of the reference matters (bits_used() and zigzag work within N's width, max - min overflows in N): values
// are read as N = int_bytes wide and held sign-extended.  The reference's quirks are kept (write_varint's arithmetic shift of a
// value with N's top bit set; f32 percentile lengths); on the two inputs on which it panics (oracle/oo_encode.c's header) a run
// is written DIRECT.  tests/test_gpu_encode.py compares the bytes with the restated reference encoder (oracle/oo_encode.c).
#pragma once

#define ENC_TILE 512u
#define ENC_FAN 32u

template <int KIND>
struct EncK;
template <>
struct EncK<0> {  // Integer RLE v2
  static constexpr uint32_t MAXFIX = 512, MAXVAR = 512;
};
template <>
struct EncK<1> {  // byte RLE
  static constexpr uint32_t MAXFIX = 130, MAXVAR = 128;
};

__device__ __forceinline__ int64_t enc_ld(const void* values, uint64_t i, int int_bytes) {
  switch (int_bytes) {
    case 1: return (int64_t)((const uint8_t*)values)[i];
    case 2: return (int64_t)((const int16_t*)values)[i];
    case 4: return (int64_t)((const int32_t*)values)[i];
    default: return ((const int64_t*)values)[i];
  }
}

__device__ __forceinline__ uint32_t enc_first_set(const uint64_t* words, uint32_t p) {  // first set bit at index >= p of 16 words; 1024: none
  uint32_t w = p >> 6;
  uint64_t m = words[w] >> (p & 63);
  if (m) return p + (uint32_t)__builtin_ctzll(m);
  for (w++; w < 16; w++) {
    m = words[w];
    if (m) return w * 64 + (uint32_t)__builtin_ctzll(m);
  }
  return 1024;
}

// 1. next(p) of a tile's positions and the tile's exit map
template <int KIND>
__global__ void __launch_bounds__(512) enc_next_kernel(const void* values, uint64_t n, int int_bytes, uint16_t* next16, uint16_t* map0) {
  __shared__ int64_t sv[1026];
  __shared__ uint64_t eqw[18], neqw[16], triw[16];
  __shared__ uint16_t jump[512];
  const uint32_t t = threadIdx.x, wave = t >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * ENC_TILE;
  for (uint32_t i = t; i < 1025; i += 512) sv[i] = base + i < n ? enc_ld(values, base + i, int_bytes) : 0;
  if (t < 2) eqw[16 + t] = 0;
  __syncthreads();
  {
    const bool e0 = base + t + 1 < n && sv[t] == sv[t + 1];
    const bool e1 = base + t + 513 < n && sv[t + 512] == sv[t + 513];
    const uint64_t b0 = __ballot(e0), b1 = __ballot(e1);
    if ((t & 63) == 0) {
      eqw[wave] = b0;
      eqw[8 + wave] = b1;
    }
  }
  __syncthreads();
  if (t < 16) {
    triw[t] = eqw[t] & ((eqw[t] >> 1) | (eqw[t + 1] << 63));
    neqw[t] = ~eqw[t];
  }
  __syncthreads();
  const uint64_t pos = base + t;
  uint32_t L = ENC_TILE, fixed = 0;
  if (pos < n) {
    fixed = (uint32_t)(triw[t >> 6] >> (t & 63)) & 1u;
    if (fixed) {
      const uint32_t z = enc_first_set(neqw, t);  // values t .. z are equal
      L = z - t + 1;
      if (L > EncK<KIND>::MAXFIX) L = EncK<KIND>::MAXFIX;
    } else {
      const uint32_t tt = enc_first_set(triw, t);
      L = tt - t <= EncK<KIND>::MAXVAR - 3 ? tt - t : EncK<KIND>::MAXVAR;
    }
    if ((uint64_t)L > n - pos) L = (uint32_t)(n - pos);
    next16[pos] = (uint16_t)(L | (fixed << 15));
  }
  jump[t] = (uint16_t)(t + L);
  __syncthreads();
  for (int r = 0; r < 10; r++) {
    const uint16_t j = jump[t];
    const uint16_t j2 = j < ENC_TILE ? jump[j] : j;
    __syncthreads();
    jump[t] = j2;
    __syncthreads();
  }
  map0[(uint64_t)blockIdx.x * ENC_TILE + t] = (uint16_t)(jump[t] - ENC_TILE);
}

// 2. the maps of spans of ENC_FAN maps; and back down, where the chain enters every map's span
extern "C" __global__ void __launch_bounds__(512) enc_compose_kernel(const uint16_t* src, uint32_t count, uint16_t* dst) {
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = g * ENC_FAN, hi = lo + ENC_FAN < count ? lo + ENC_FAN : count;
  uint32_t x = t;
  for (uint32_t j = lo; j < hi; j++) x = src[(uint64_t)j * ENC_TILE + x];
  dst[(uint64_t)g * ENC_TILE + t] = (uint16_t)x;
}
extern "C" __global__ void __launch_bounds__(64) enc_entries_kernel(const uint16_t* src, uint32_t count, const uint16_t* parent_entry, uint32_t n_groups,
                                                                    uint16_t* child_entry) {
  const uint32_t g = blockIdx.x * 64 + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t lo = g * ENC_FAN, hi = lo + ENC_FAN < count ? lo + ENC_FAN : count;
  uint32_t x = parent_entry ? parent_entry[g] : 0;
  for (uint32_t j = lo; j < hi; j++) {
    child_entry[j] = (uint16_t)x;
    x = src[(uint64_t)j * ENC_TILE + x];
  }
}

// 3. the runs that start in a tile: counted (runs == nullptr), then listed.  A wavefront per tile; KIND 1 also states the sizes.
template <int KIND>
__global__ void __launch_bounds__(256) enc_runs_kernel(const uint16_t* next16, uint64_t n, const uint16_t* entry, uint32_t n_tiles, uint32_t* tile_runs,
                                                       const uint64_t* tile_off, uint32_t* runs, uint32_t* run_bytes) {
  __shared__ uint16_t nx[4][ENC_TILE];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t k = blockIdx.x * 4 + wave;
  if (k >= n_tiles) return;
  const uint64_t base = (uint64_t)k * ENC_TILE;
  const uint32_t have = n - base < ENC_TILE ? (uint32_t)(n - base) : ENC_TILE;
  for (uint32_t i = lane; i < ENC_TILE; i += 64) nx[wave][i] = i < have ? next16[base + i] : (uint16_t)ENC_TILE;
  __builtin_amdgcn_wave_barrier();
  if (lane) return;
  uint32_t x = entry[k], cnt = 0;
  const uint64_t off = runs ? tile_off[k] : 0;
  while (x < have) {
    const uint32_t e = nx[wave][x];
    if (runs) {
      runs[off + cnt] = (uint32_t)(base + x);
      if (KIND == 1) run_bytes[off + cnt] = (e & 0x8000u) ? 2u : 1u + (e & 0x3ffu);
    }
    cnt++;
    x += e & 0x3ffu;
  }
  if (!runs) tile_runs[k] = cnt;
}

// exclusive scan of a u32 array into u64 (run counts of tiles, sizes of runs): sums of tiles of 2048, their scan, the rest
extern "C" __global__ void __launch_bounds__(256) enc_scan_tiles_kernel(const uint32_t* in, uint64_t count, uint64_t* sums) {
  __shared__ uint64_t part[4];
  const uint64_t lo = (uint64_t)blockIdx.x * 2048 + threadIdx.x * 8;
  uint64_t s = 0;
  for (uint32_t i = 0; i < 8; i++) s += lo + i < count ? in[lo + i] : 0;
  for (int o = 32; o; o >>= 1) s += (uint64_t)__shfl_xor((long long)s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
extern "C" __global__ void __launch_bounds__(1024) enc_scan_sums_kernel(uint64_t* sums, uint32_t n_sums, uint64_t* total) {
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (n_sums + 1023) / 1024;
  const uint32_t lo = t * per < n_sums ? t * per : n_sums, hi = lo + per < n_sums ? lo + per : n_sums;
  uint64_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += sums[k];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    uint64_t acc = 0;
    for (uint32_t k = 0; k < 1024; k++) {
      const uint64_t x = part[k];
      part[k] = acc;
      acc += x;
    }
    *total = acc;
  }
  __syncthreads();
  uint64_t acc = part[t];
  for (uint32_t k = lo; k < hi; k++) {
    const uint64_t x = sums[k];
    sums[k] = acc;
    acc += x;
  }
}
extern "C" __global__ void __launch_bounds__(256) enc_scan_apply_kernel(const uint32_t* in, uint64_t count, const uint64_t* sums, uint64_t* out) {
  __shared__ uint64_t wsum[4];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint64_t lo = (uint64_t)blockIdx.x * 2048 + t * 8;
  uint32_t v[8];
  uint64_t s = 0;
  for (uint32_t i = 0; i < 8; i++) {
    v[i] = lo + i < count ? in[lo + i] : 0;
    s += v[i];
  }
  uint64_t inc = s;  // inclusive scan over the wavefront's lanes
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = (uint64_t)__shfl_up((long long)inc, o);
    if (lane >= (uint32_t)o) inc += y;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint64_t acc = sums[blockIdx.x] + inc - s;
  for (uint32_t w = 0; w < wave; w++) acc += wsum[w];
  for (uint32_t i = 0; i < 8; i++) {
    if (lo + i < count) out[lo + i] = acc;
    acc += v[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Integer RLE v2: what a run is written as

struct EncRun {     // 24 bytes
  uint32_t start;   // position of the run's first value
  uint16_t len;
  uint8_t mode;     // 0 SHORT_REPEAT, 1 DIRECT, 2 PATCHED_BASE, 3 DELTA with a fixed step, 4 DELTA with varying steps
  uint8_t w;        // bits per packed value (DIRECT, PATCHED_BASE: the reduced values, DELTA: the steps); SHORT_REPEAT: bytes
  uint8_t pbw, pgw; // PATCHED_BASE: patch width, gap width
  uint8_t ne;       // PATCHED_BASE: patch list entries
  uint8_t w95;      // PATCHED_BASE: bits kept in the reduced values (before rounding to a width the format can state)
  uint32_t panic;   // 1: the reference panics on this run (written DIRECT)
  int64_t base;     // PATCHED_BASE: the minimum
};

// VarintSerde::bits_used within N (integer/mod.rs:124-126)
__device__ __forceinline__ uint32_t enc_bits_n(int64_t v, uint32_t nbits) { return v < 0 ? nbits : (v ? 64u - (uint32_t)__builtin_clzll((uint64_t)v) : 0u); }
// signed_zigzag_encode in N (util.rs:550-553); UnsignedEncoding: the value itself
__device__ __forceinline__ int64_t enc_zigzag_n(int64_t v, uint32_t nbits, int is_signed) {
  if (!is_signed) return v;
  const uint64_t z = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
  return nbits == 64 ? (int64_t)z : (int64_t)(z << (64 - nbits)) >> (64 - nbits);
}
// get_closest_fixed_bits (util.rs:407-421)
__device__ __forceinline__ uint32_t enc_fixed_bits(uint32_t n) {
  if (n == 0) return 1;
  if (n <= 24) return n;
  if (n <= 32) return (n + 1) & ~1u;
  return (n + 7) & ~7u;
}
// encode_bit_width (util.rs:423-437)
__device__ __forceinline__ uint32_t enc_width_code(uint32_t n) {
  n = enc_fixed_bits(n);
  return n <= 24 ? n - 1 : (n <= 32 ? 24 + (n - 26) / 2 : 28 + (n - 40) / 8);
}
// decode_bit_width (util.rs:439-453)
__device__ __forceinline__ uint32_t enc_code_width(uint32_t c) { return c <= 23 ? c + 1 : (c <= 27 ? 26 + (c - 24) * 2 : 40 + (c - 28) * 8); }
// get_closest_aligned_bit_width (util.rs:456-472)
__device__ __forceinline__ uint32_t enc_aligned_bits(uint32_t w) {
  if (w <= 1) return 1;
  if (w == 2) return 2;
  if (w <= 4) return 4;
  if (w <= 48) return (w + 7) & ~7u;
  return w <= 54 ? 56 : 64;
}
// write_varint (util.rs:501-520): the shift is N's arithmetic shift
__device__ __forceinline__ uint32_t enc_varint_len(int64_t value, uint32_t nbits) {
  const uint32_t s = (enc_bits_n(value, nbits) + 6) / 7;
  return s ? s : 1;
}
__device__ __forceinline__ uint32_t enc_put_varint(uint8_t* p, int64_t value, uint32_t nbits) {
  const uint32_t size = enc_varint_len(value, nbits);
  for (uint32_t i = 0; i < size; i++) {
    const uint32_t shift = i * 7;
    const uint32_t b = (uint32_t)((shift >= 64 ? (value >> 63) : (value >> shift)) & 0x7f);
    p[i] = (uint8_t)(b | (i + 1 < size ? 0x80u : 0u));
  }
  return size;
}
__device__ __forceinline__ int64_t enc_sat_sub(int64_t a, int64_t b) {
  int64_t r;
  if (__builtin_sub_overflow(a, b, &r)) return b > 0 ? INT64_MIN : INT64_MAX;
  return r;
}
__device__ __forceinline__ int64_t enc_sat_abs(int64_t a) { return a == INT64_MIN ? INT64_MAX : (a < 0 ? -a : a); }

__device__ __forceinline__ int64_t wave_min_i64(int64_t x) {
  for (int o = 32; o; o >>= 1) {
    const int64_t y = (int64_t)__shfl_xor((long long)x, o);
    x = y < x ? y : x;
  }
  return x;
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t x) {
  for (int o = 32; o; o >>= 1) {
    const int64_t y = (int64_t)__shfl_xor((long long)x, o);
    x = y > x ? y : x;
  }
  return x;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
  for (int o = 32; o; o >>= 1) {
    const uint32_t y = (uint32_t)__shfl_xor((int)x, o);
    x = y > x ? y : x;
  }
  return x;
}

// calculate_percentile_bits (util.rs:584-610) over a run's values: `code` of this lane's `cnt` values -> the histogram in LDS,
// then the walk from the top.  hist: 32 words of this wavefront.  per_len is computed in f32 as the reference does.
__device__ __forceinline__ void enc_hist(uint32_t* hist, const uint32_t code[8], uint32_t cnt, uint32_t lane) {
  if (lane < 32) hist[lane] = 0;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 8; i++)
    if ((uint32_t)i < cnt) atomicAdd(&hist[code[i]], 1u);
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t enc_percentile(const uint32_t* hist, uint32_t len, float percentile) {
  const float frac = 1.0f - percentile;
  uint32_t per_len = (uint32_t)__fmul_rn(frac, (float)len);
  for (int i = 31; i >= 0; i--) {
    const uint32_t h = hist[i];
    if (per_len >= h)
      per_len -= h;
    else
      return enc_code_width((uint32_t)i);
  }
  return 1;
}

// the low `w` bits of `cnt` (<= 8) values, most significant bit first, into ceil(cnt w / 8) bytes at q (write_packed_ints, util.rs:237-291)
__device__ __forceinline__ void enc_pack8(uint8_t* q, const uint64_t u[8], uint32_t cnt, uint32_t w) {
  unsigned __int128 acc = 0;
  uint32_t bits = 0, k = 0;
  const uint64_t mask = w == 64 ? ~0ull : ((1ull << w) - 1);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if ((uint32_t)i < cnt) {
      acc = (acc << w) | (u[i] & mask);
      bits += w;
      while (bits >= 8) {
        q[k++] = (uint8_t)(acc >> (bits - 8));
        bits -= 8;
      }
    }
  }
  if (bits) q[k] = (uint8_t)((uint64_t)acc << (8 - bits));
}

struct EncVals {
  int64_t v[8];
  uint32_t cnt;
};
__device__ __forceinline__ EncVals enc_load_run(const void* values, int int_bytes, uint64_t first, uint32_t len, uint32_t lane) {
  EncVals e;
  const uint32_t lo = lane * 8;
  e.cnt = lo < len ? (len - lo < 8 ? len - lo : 8) : 0;
#pragma unroll
  for (int i = 0; i < 8; i++) e.v[i] = (uint32_t)i < e.cnt ? enc_ld(values, first + lo + i, int_bytes) : 0;
  return e;
}


// PATCHED_BASE's patch list (derive_patches, patched_base.rs:162-226): the indexes of the values above `mask`, in order, into
// idx[] (LDS of this wavefront); returns how many.  flags: bit i = this lane's value i is patched.
__device__ __forceinline__ uint32_t enc_patch_indexes(uint32_t* idx, uint32_t flags, uint32_t lane) {
  const uint32_t c = (uint32_t)__builtin_popcount(flags);
  uint32_t inc = c;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)inc, o);
    if (lane >= (uint32_t)o) inc += y;
  }
  const uint32_t total = (uint32_t)__shfl((int)inc, 63);
  uint32_t at = inc - c;
  for (uint32_t f = flags; f; f &= f - 1) {
    if (at < 64) idx[at] = lane * 8 + (uint32_t)__builtin_ctz(f);
    at++;
  }
  __builtin_amdgcn_wave_barrier();
  return total;
}

// 4a. what a run is written as.  A wavefront takes 64 runs: runs of repeats (SHORT_REPEAT, a DELTA of step 0: rle_v2/mod.rs:303-337,
// :361-384) and literals of up to three values (DIRECT, :426-432) are settled by their lane alone; longer literals take the
// wavefront, one run after the other -- determine_variable_run_encoding (rle_v2/mod.rs:422-531) with lane l holding values
// 8 l .. 8 l + 7 of the run.
__device__ __forceinline__ void enc2_plan_coop(const void* values, int int_bytes, uint32_t nbits, int is_signed, uint32_t start, uint32_t len, uint32_t lane,
                                               uint32_t* hist, uint32_t* idx, EncRun& rec, uint32_t& bytes) {
  const EncVals e = enc_load_run(values, int_bytes, start, len, lane);
  int64_t zz[8];
  uint32_t zbits = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    zz[i] = enc_zigzag_n(e.v[i], nbits, is_signed);
    if ((uint32_t)i < e.cnt) {
      const uint32_t b = enc_bits_n(zz[i], nbits);
      zbits = b > zbits ? b : zbits;
    }
  }
  zbits = wave_max_u32(zbits);
  const uint32_t direct_w = enc_aligned_bits(zbits);
  rec.mode = 1;
  rec.w = (uint8_t)direct_w;
  bytes = 2 + (len * direct_w + 7) / 8;
  // delta_encoding_check (rle_v2/mod.rs:186-239)
  const int64_t v0 = (int64_t)__shfl((long long)e.v[0], 0), v1 = (int64_t)__shfl((long long)e.v[1], 0);
  const int64_t first_delta = enc_sat_sub(v1, v0);
  const int64_t below = (int64_t)__shfl_up((long long)e.v[7], 1);
  int64_t mn = INT64_MAX, mx = INT64_MIN, maxd = 0;
  bool inc = true, dec = true, fx = true;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if ((uint32_t)i < e.cnt) {
      mn = e.v[i] < mn ? e.v[i] : mn;
      mx = e.v[i] > mx ? e.v[i] : mx;
      const uint32_t gi = lane * 8 + i;
      if (gi >= 2) {
        const int64_t cur = enc_sat_sub(e.v[i], i ? e.v[i ? i - 1 : 0] : below);
        inc = inc && cur >= 0;
        dec = dec && cur <= 0;
        fx = fx && cur == first_delta;
        const int64_t a = enc_sat_abs(cur);
        maxd = a > maxd ? a : maxd;
      }
    }
  }
  mn = wave_min_i64(mn);
  mx = wave_max_i64(mx);
  maxd = wave_max_i64(maxd);
  const bool is_inc = first_delta > 0 && !__ballot(!inc), is_dec = first_delta < 0 && !__ballot(!dec), is_fixed = !__ballot(!fx);
  int64_t range;
  bool ovf = __builtin_sub_overflow(mx, mn, &range);
  if (!ovf && nbits < 64) ovf = range >= ((int64_t)1 << (nbits - 1));
  const int64_t zbase = enc_zigzag_n(v0, nbits, is_signed);
  const int64_t zfirst = enc_zigzag_n(first_delta, 64, 1);
  if (ovf) return;  // DIRECT
  if (is_fixed) {
    rec.mode = 3;
    rec.w = 0;
    bytes = 2 + enc_varint_len(zbase, nbits) + enc_varint_len(zfirst, 64);
    return;
  }
  if (first_delta != 0 && (is_inc || is_dec)) {
    uint32_t w = enc_aligned_bits(enc_bits_n(maxd, 64));
    w = w == 1 ? 2 : w;
    rec.mode = 4;
    rec.w = (uint8_t)w;
    bytes = 2 + enc_varint_len(zbase, nbits) + enc_varint_len(zfirst, 64) + ((len - 2) * w + 7) / 8;
    return;
  }
  if (mn != INT64_MIN && (mn < 0 ? -mn : mn) >= ((int64_t)1 << 56)) return;  // DIRECT
  uint32_t code[8];
#pragma unroll
  for (int i = 0; i < 8; i++) code[i] = enc_width_code(enc_bits_n(zz[i], nbits));
  enc_hist(hist, code, e.cnt, lane);
  const uint32_t z90 = enc_percentile(hist, len, 0.90f), z100 = enc_percentile(hist, len, 1.00f);
  __builtin_amdgcn_wave_barrier();
  if (z100 <= z90 + 1) return;  // DIRECT
  int64_t brl[8], maxb = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    brl[i] = (uint32_t)i < e.cnt ? e.v[i] - mn : 0;
    maxb = brl[i] > maxb ? brl[i] : maxb;
    code[i] = enc_width_code(enc_bits_n(brl[i], 64));
  }
  maxb = wave_max_i64(maxb);
  enc_hist(hist, code, e.cnt, lane);
  const uint32_t w100 = enc_bits_n(maxb, 64);
  uint32_t w95 = enc_percentile(hist, len, 0.95f);
  __builtin_amdgcn_wave_barrier();
  if (w100 == w95) return;  // DIRECT
  if (w100 < w95 || mn == INT64_MIN) {
    rec.panic = 1;  // the reference panics here (patched_base.rs:235 / :259): DIRECT
    return;
  }
  uint32_t pbw = enc_fixed_bits(w100 - w95);
  if (pbw == 64) {
    pbw = 56;
    w95 = 8;
  }
  const int64_t mask = (int64_t)(((uint64_t)1 << w95) - 1);
  uint32_t flags = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if ((uint32_t)i < e.cnt && brl[i] > mask) flags |= 1u << i;
  const uint32_t np = enc_patch_indexes(idx, flags, lane);
  uint32_t gap = 0, extra = 0;
  if (lane < np) {
    gap = idx[lane] - (lane ? idx[lane - 1] : 0);
    extra = gap == 511 ? 2 : (gap > 255 ? 1 : 0);
  }
  const bool jumps = __ballot(extra != 0) != 0;
  const uint32_t ne = np + (uint32_t)__builtin_popcountll(__ballot(extra == 1)) + 2 * (uint32_t)__builtin_popcountll(__ballot(extra == 2));
  const uint32_t max_gap = jumps ? 255 : wave_max_u32(lane < np ? gap : 0);
  const uint32_t pgw = max_gap ? 32u - (uint32_t)__builtin_clz(max_gap) : 1u;
  const uint64_t amin = mn < 0 ? (uint64_t)0 - (uint64_t)mn : (uint64_t)mn;
  uint32_t bb = (enc_fixed_bits(enc_bits_n((int64_t)amin, 64) + 1) + 7) / 8;
  bb = bb ? bb : 1;
  rec.mode = 2;
  rec.w = (uint8_t)enc_fixed_bits(w95);
  rec.w95 = (uint8_t)w95;
  rec.pbw = (uint8_t)pbw;
  rec.pgw = (uint8_t)pgw;
  rec.ne = (uint8_t)ne;
  rec.base = mn;
  bytes = 4 + bb + (len * rec.w + 7) / 8 + (ne * enc_fixed_bits(pgw + pbw) + 7) / 8;
  __builtin_amdgcn_wave_barrier();
}

extern "C" __global__ void __launch_bounds__(256) enc2_plan_kernel(const void* values, int int_bytes, int is_signed, const uint16_t* next16, const uint32_t* runs,
                                                                   uint32_t n_runs, EncRun* recs, uint32_t* run_bytes, uint32_t rpw) {
  __shared__ uint32_t lds[4][96];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* hist = lds[wave];
  uint32_t* idx = lds[wave] + 32;
  const uint32_t nbits = (uint32_t)int_bytes * 8;
  // (rpw runs per wavefront, 4 .. 64: many short runs -- a lane each --, or few long ones -- the wavefront one after the other)
  const uint32_t r0 = (blockIdx.x * 4 + wave) * rpw, r = r0 + lane;
  const bool have = lane < rpw && r < n_runs;
  uint32_t start = 0, e16 = 0;
  if (have) {
    start = runs[r];
    e16 = next16[start];
  }
  const uint32_t len = e16 & 0x3ffu;
  const bool coop = have && !(e16 & 0x8000u) && len > 3;
  if (have && !coop) {
    EncRun rec;
    rec.start = start;
    rec.len = (uint16_t)len;
    rec.pbw = rec.pgw = rec.ne = rec.w95 = 0;
    rec.panic = 0;
    rec.base = 0;
    uint32_t bytes;
    if (e16 & 0x8000u) {
      const int64_t z = enc_zigzag_n(enc_ld(values, start, int_bytes), nbits, is_signed);
      if (len <= 10) {
        uint32_t b = (enc_bits_n(z, nbits) + 7) / 8;
        b = b ? b : 1;
        rec.mode = 0;
        rec.w = (uint8_t)b;
        bytes = 1 + b;
      } else {
        rec.mode = 3;
        rec.w = 0;
        bytes = 2 + enc_varint_len(z, nbits) + 1;
      }
    } else {  // up to three literals: DIRECT at the width of the widest
      uint32_t zbits = 0;
      for (uint32_t i = 0; i < len; i++) {
        const uint32_t b = enc_bits_n(enc_zigzag_n(enc_ld(values, (uint64_t)start + i, int_bytes), nbits, is_signed), nbits);
        zbits = b > zbits ? b : zbits;
      }
      const uint32_t w = enc_aligned_bits(zbits);
      rec.mode = 1;
      rec.w = (uint8_t)w;
      bytes = 2 + (len * w + 7) / 8;
    }
    recs[r] = rec;
    run_bytes[r] = bytes;
  }
  unsigned long long m = __ballot(coop);
  while (m) {
    const int src = __builtin_ctzll(m);
    m &= m - 1;
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)start, src), l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, src);
    EncRun rec;
    rec.start = s0;
    rec.len = (uint16_t)l0;
    rec.pbw = rec.pgw = rec.ne = rec.w95 = 0;
    rec.panic = 0;
    rec.base = 0;
    uint32_t bytes = 0;
    enc2_plan_coop(values, int_bytes, nbits, is_signed, s0, l0, lane, hist, idx, rec, bytes);
    if (lane == 0) {
      recs[r0 + src] = rec;
      run_bytes[r0 + src] = bytes;
    }
  }
}

// 4b. the runs written: write_short_repeat (short_repeat.rs:65-81), write_direct (direct.rs:69-95), write_fixed_delta /
// write_varying_delta (delta.rs:118-182), write_patched_base (patched_base.rs:228-284).  As in the plan: repeats, fixed steps and
// up to three literals by their lane, the rest by the wavefront.
__device__ __forceinline__ void enc2_emit_coop(const void* values, int int_bytes, uint32_t nbits, int is_signed, const EncRun& rec, uint8_t* p, uint32_t lane,
                                               uint32_t* idx, uint64_t* ent) {
  const uint32_t len = rec.len, w = rec.w;
  if (rec.mode == 1) {
    if (lane == 0) {
      p[0] = (uint8_t)(0x40u | (enc_width_code(w) << 1) | ((len - 1) >> 8));
      p[1] = (uint8_t)((len - 1) & 0xff);
    }
    const EncVals e = enc_load_run(values, int_bytes, rec.start, len, lane);
    uint64_t u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) u[i] = (uint64_t)enc_zigzag_n(e.v[i], nbits, is_signed);
    if (e.cnt) enc_pack8(p + 2 + (uint64_t)lane * w, u, e.cnt, w);
    return;
  }
  if (rec.mode == 4) {
    const int64_t v0 = enc_ld(values, rec.start, int_bytes), v1 = enc_ld(values, rec.start + 1, int_bytes);
    const int64_t zbase = enc_zigzag_n(v0, nbits, is_signed), zfirst = enc_zigzag_n(enc_sat_sub(v1, v0), 64, 1);
    const uint32_t head = 2 + enc_varint_len(zbase, nbits) + enc_varint_len(zfirst, 64);
    if (lane == 0) {
      p[0] = (uint8_t)(0xc0u | (enc_width_code(w) << 1) | ((len - 1) >> 8));
      p[1] = (uint8_t)((len - 1) & 0xff);
      uint32_t k = 2;
      k += enc_put_varint(p + k, zbase, nbits);
      k += enc_put_varint(p + k, zfirst, 64);
    }
    // steps 2 .. len - 1: lane l packs |v[j + 2] - v[j + 1]| for j = 8 l .. 8 l + 7
    const uint32_t n_adj = len - 2, lo = lane * 8;
    const uint32_t cnt = lo < n_adj ? (n_adj - lo < 8 ? n_adj - lo : 8) : 0;
    if (cnt) {
      uint64_t u[8];
      int64_t prev = enc_ld(values, (uint64_t)rec.start + lo + 1, int_bytes);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        u[i] = 0;
        if ((uint32_t)i < cnt) {
          const int64_t cur = enc_ld(values, (uint64_t)rec.start + lo + 2 + i, int_bytes);
          u[i] = (uint64_t)enc_sat_abs(enc_sat_sub(cur, prev));
          prev = cur;
        }
      }
      enc_pack8(p + head + (uint64_t)lane * w, u, cnt, w);
    }
    return;
  }
  // PATCHED_BASE
  const int64_t mn = rec.base;
  const uint32_t w95 = rec.w95, pbw = rec.pbw, pgw = rec.pgw;
  const uint64_t amin = mn < 0 ? (uint64_t)0 - (uint64_t)mn : (uint64_t)mn;
  uint32_t bb = (enc_fixed_bits(enc_bits_n((int64_t)amin, 64) + 1) + 7) / 8;
  bb = bb ? bb : 1;
  if (lane == 0) {
    p[0] = (uint8_t)(0x80u | (enc_width_code(w95) << 1) | ((len - 1) >> 8));
    p[1] = (uint8_t)((len - 1) & 0xff);
    p[2] = (uint8_t)(((bb - 1) << 5) | enc_width_code(pbw));
    p[3] = (uint8_t)(((pgw - 1) << 5) | rec.ne);
    const uint64_t msb = amin | ((uint64_t)(mn < 0) << (bb * 8 - 1));
    for (uint32_t k = 0; k < bb; k++) p[4 + k] = (uint8_t)(msb >> (8 * (bb - 1 - k)));
  }
  const EncVals e = enc_load_run(values, int_bytes, rec.start, len, lane);
  const int64_t mask = (int64_t)(((uint64_t)1 << w95) - 1);
  uint64_t u[8];
  uint32_t flags = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int64_t b = (uint32_t)i < e.cnt ? e.v[i] - mn : 0;
    if (b > mask) flags |= 1u << i;
    u[i] = (uint64_t)(b & mask);
  }
  if (e.cnt) enc_pack8(p + 4 + bb + (uint64_t)lane * w, u, e.cnt, w);
  const uint32_t np = enc_patch_indexes(idx, flags, lane);
  uint32_t gap = 0, extra = 0;
  uint64_t patch_bits = 0;
  if (lane < np) {
    const uint32_t at = idx[lane];
    gap = at - (lane ? idx[lane - 1] : 0);
    extra = gap == 511 ? 2 : (gap > 255 ? 1 : 0);
    gap = gap == 511 ? 1 : (gap > 255 ? gap - 255 : gap);
    patch_bits = (uint64_t)(enc_ld(values, (uint64_t)rec.start + at, int_bytes) - mn) >> w95;
  }
  uint32_t inc = extra + (lane < np ? 1u : 0u);
  const uint32_t mine = inc;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)inc, o);
    if (lane >= (uint32_t)o) inc += y;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < np) {
    uint32_t at = inc - mine;
    const uint64_t jump = (uint64_t)255 << pbw;
    for (uint32_t k = 0; k < extra; k++) ent[at++] = jump;
    ent[at] = patch_bits | ((uint64_t)gap << pbw);
  }
  __builtin_amdgcn_wave_barrier();
  const uint32_t ne = rec.ne, pw = enc_fixed_bits(pgw + pbw), lo = lane * 8;
  const uint32_t cnt = lo < ne ? (ne - lo < 8 ? ne - lo : 8) : 0;
  if (cnt) {
#pragma unroll
    for (int i = 0; i < 8; i++) u[i] = (uint32_t)i < cnt ? ent[lo + i] : 0;
    enc_pack8(p + 4 + bb + (len * w + 7) / 8 + (uint64_t)lane * pw, u, cnt, pw);
  }
  __builtin_amdgcn_wave_barrier();
}

extern "C" __global__ void __launch_bounds__(256) enc2_emit_kernel(const void* values, int int_bytes, int is_signed, const EncRun* recs, const uint64_t* offsets,
                                                                   uint32_t n_runs, uint8_t* out, uint32_t rpw) {
  __shared__ uint32_t lds32[4][64];
  __shared__ uint64_t lds64[4][32];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t nbits = (uint32_t)int_bytes * 8;
  const uint32_t r0 = (blockIdx.x * 4 + wave) * rpw, r = r0 + lane;
  EncRun rec;
  rec.mode = 0xff;
  rec.len = 0;
  bool coop = false;
  if (lane < rpw && r < n_runs) {
    rec = recs[r];
    const uint32_t len = rec.len;
    uint8_t* p = out + offsets[r];
    if (rec.mode == 0) {
      const int64_t z = enc_zigzag_n(enc_ld(values, rec.start, int_bytes), nbits, is_signed);
      p[0] = (uint8_t)(((rec.w - 1u) << 3) | (len - 3u));
      for (uint32_t k = 0; k < rec.w; k++) p[1 + k] = (uint8_t)((uint64_t)z >> (8 * (rec.w - 1 - k)));
    } else if (rec.mode == 3) {
      const int64_t v0 = enc_ld(values, rec.start, int_bytes), v1 = enc_ld(values, rec.start + 1, int_bytes);
      p[0] = (uint8_t)(0xc0u | ((len - 1) >> 8));
      p[1] = (uint8_t)((len - 1) & 0xff);
      uint32_t k = 2;
      k += enc_put_varint(p + k, enc_zigzag_n(v0, nbits, is_signed), nbits);
      k += enc_put_varint(p + k, enc_zigzag_n(enc_sat_sub(v1, v0), 64, 1), 64);
    } else if (rec.mode == 1 && len <= 3) {
      p[0] = (uint8_t)(0x40u | (enc_width_code(rec.w) << 1));
      p[1] = (uint8_t)(len - 1);
      uint64_t u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t i = 0; i < len; i++) u[i] = (uint64_t)enc_zigzag_n(enc_ld(values, (uint64_t)rec.start + i, int_bytes), nbits, is_signed);
      enc_pack8(p + 2, u, len, rec.w);
    } else {
      coop = true;
    }
  }
  unsigned long long m = __ballot(coop);
  while (m) {
    const int src = __builtin_ctzll(m);
    m &= m - 1;
    const EncRun rc = recs[r0 + src];
    enc2_emit_coop(values, int_bytes, nbits, is_signed, rc, out + offsets[r0 + src], lane, lds32[wave], lds64[wave]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// byte RLE: write_run / write_literals (byte.rs:176-197); a wavefront per 64 runs, literals copied by the wavefront
extern "C" __global__ void __launch_bounds__(256) enc1_emit_kernel(const uint8_t* values, const uint16_t* next16, const uint32_t* runs, const uint64_t* offsets,
                                                                   uint32_t n_runs, uint8_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  uint32_t start = 0, e16 = 0;
  uint64_t off = 0;
  if (r < n_runs) {
    start = runs[r];
    e16 = next16[start];
    off = offsets[r];
    if (e16 & 0x8000u) {
      out[off] = (uint8_t)((e16 & 0x3ffu) - 3);
      out[off + 1] = values[start];
    } else {
      out[off] = (uint8_t)(0u - (e16 & 0x3ffu));
    }
  }
  uint64_t lit = __ballot(r < n_runs && !(e16 & 0x8000u));
  while (lit) {
    const int src = __builtin_ctzll(lit);
    lit &= lit - 1;
    const uint32_t s = (uint32_t)__shfl((int)start, src), len = (uint32_t)__shfl((int)e16, src) & 0x3ffu;
    const uint64_t o = (uint64_t)__shfl((long long)off, src) + 1;
    for (uint32_t i = lane; i < len; i += 64) out[o + i] = values[s + i];
  }
}

// BooleanEncoder::finish (boolean.rs:157-169): the bitmap's bytes with their bits reversed (ORC counts from the top bit), the
// last byte's spare bits zero
extern "C" __global__ void __launch_bounds__(256) enc_bool_bytes_kernel(const uint8_t* bits, uint64_t n_bits, uint8_t* bytes) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t nb = (n_bits + 7) / 8;
  if (i >= nb) return;
  uint32_t x = bits[i];
  if (i == nb - 1 && (n_bits & 7)) x &= (1u << (n_bits & 7)) - 1;
  bytes[i] = (uint8_t)(__builtin_bitreverse32(x) >> 24);
}

// what a column's writer does before its values reach an encoder (writer/column.rs:103-139, :196-232, :304-359): only the valid
// rows' values are written.  Positions of the valid rows by a scan of the bitmap's popcounts, then a gather.
extern "C" __global__ void __launch_bounds__(256) enc_valid_counts_kernel(const uint8_t* validity, uint64_t n_rows, uint32_t* counts) {
  // one count per 64 rows
  const uint64_t wi = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n_words = (n_rows + 63) / 64;
  if (wi >= n_words) return;
  uint64_t word = 0;
  const uint64_t nb = (n_rows + 7) / 8;
  for (uint32_t k = 0; k < 8; k++) word |= wi * 8 + k < nb ? (uint64_t)validity[wi * 8 + k] << (8 * k) : 0;
  if (wi == n_words - 1 && (n_rows & 63)) word &= (1ull << (n_rows & 63)) - 1;
  counts[wi] = (uint32_t)__builtin_popcountll(word);
}
// elem_bytes 1 / 2 / 4 / 8: fixed-width values;  0: a bitmap's bits (Boolean values) gathered into bytes_out as 0 / 1 bytes
extern "C" __global__ void __launch_bounds__(256) enc_gather_valid_kernel(const uint8_t* validity, uint64_t n_rows, const uint64_t* word_off, const void* values,
                                                                          int elem_bytes, void* out) {
  const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  if (!((validity[row >> 3] >> (row & 7)) & 1)) return;
  const uint64_t wi = row >> 6;
  uint64_t word = 0;
  const uint64_t nb = (n_rows + 7) / 8;
  for (uint32_t k = 0; k < 8; k++) word |= wi * 8 + k < nb ? (uint64_t)validity[wi * 8 + k] << (8 * k) : 0;
  const uint64_t at = word_off[wi] + (uint64_t)__builtin_popcountll(word & ((1ull << (row & 63)) - 1));
  switch (elem_bytes) {
    case 0: ((uint8_t*)out)[at] = (((const uint8_t*)values)[row >> 3] >> (row & 7)) & 1; break;
    case 1: ((uint8_t*)out)[at] = ((const uint8_t*)values)[row]; break;
    case 2: ((uint16_t*)out)[at] = ((const uint16_t*)values)[row]; break;
    case 4: ((uint32_t*)out)[at] = ((const uint32_t*)values)[row]; break;
    default: ((uint64_t*)out)[at] = ((const uint64_t*)values)[row]; break;
  }
}
// 0 / 1 bytes -> a bitmap, least significant bit first (the valid rows' Boolean values, gathered)
extern "C" __global__ void __launch_bounds__(256) enc_bytes_to_bits_kernel(const uint8_t* bytes, uint64_t n, uint8_t* bits) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (n + 7) / 8) return;
  uint32_t x = 0;
  for (uint32_t k = 0; k < 8; k++) x |= i * 8 + k < n ? (uint32_t)(bytes[i * 8 + k] & 1) << k : 0;
  bits[i] = (uint8_t)x;
}
// the lengths of strings from their offsets (writer/column.rs:334-343) in the offsets' own width N (the length encoder is
// RleV2Encoder<T::Offset, UnsignedEncoding>), and as u32, zero for null rows: their scan places the valid rows' bytes
// `bad` (one word, zeroed by the host): set when a pair of offsets is not ascending or a value is 4 GiB or longer -- the Arrow
// offsets are the caller's; a negative length cast to u32 would send the copy kernel far outside the values buffer
extern "C" __global__ void __launch_bounds__(256) enc_lengths_kernel(const void* offsets, int offset_bytes, const uint8_t* validity, uint64_t n_rows, void* lengths,
                                                                     uint32_t* vlen, uint32_t* bad) {
  const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  int64_t len;
  if (offset_bytes == 4) {
    len = (int64_t)((const int32_t*)offsets)[row + 1] - ((const int32_t*)offsets)[row];
    ((int32_t*)lengths)[row] = (int32_t)len;
  } else {
    len = ((const int64_t*)offsets)[row + 1] - ((const int64_t*)offsets)[row];
    ((int64_t*)lengths)[row] = len;
  }
  const bool valid = !validity || ((validity[row >> 3] >> (row & 7)) & 1);
  if (len < 0 || len > 0xffffffffll) {
    *bad = 1;
    len = 0;
  }
  vlen[row] = valid ? (uint32_t)len : 0u;
}
// the valid rows' bytes one behind the other: a wavefront per row (row_dst: exclusive scan of vlen)
extern "C" __global__ void __launch_bounds__(256) enc_copy_strings_kernel(const uint8_t* validity, const void* offsets, int offset_bytes, uint64_t n_rows,
                                                                          const uint64_t* row_dst, const uint8_t* data, uint8_t* out) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  if (!((validity[row >> 3] >> (row & 7)) & 1)) return;
  const int64_t lo = offset_bytes == 4 ? ((const int32_t*)offsets)[row] : ((const int64_t*)offsets)[row];
  const int64_t hi = offset_bytes == 4 ? ((const int32_t*)offsets)[row + 1] : ((const int64_t*)offsets)[row + 1];
  const uint64_t d = row_dst[row];
  for (int64_t i = lane; i < hi - lo; i += 64) out[d + i] = data[lo + i];
}
