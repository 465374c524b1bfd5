// rle_encode.hip -- Integer RLE v2 ENCODING of an Int64 column on the device ("GPU ORC encode").
//
// Replaces RleV2Encoder<i64, S>::{write_slice, take_inner} (src/encoding/integer/rle_v2/mod.rs:403-531; the seam is
// PrimitiveValueEncoder, src/encoding/mod.rs:36-50, driven per column by src/writer/column.rs and flushed per stripe by
// src/writer/stripe.rs:109-165).  The reference's encoder is a greedy state machine over single values (a run is cut when a
// repeat of three shows up, literals are re-examined for DELTA / PATCHED_BASE / DIRECT when 512 have gathered); its choices are
// one valid encoding among many -- the format only fixes what a run MEANS.  Here runs are cut at fixed boundaries, 512 values
// each, so that every run is independent: ONE WAVEFRONT PER RUN, twice --
//   1. plan: the run's zigzag width (wave maximum) and whether its values are an arithmetic progression (every lane checks its
//      eight steps, a ballot decides) -> sub-encoding and byte size;
//      (a scan of the sizes gives every run its place in the stream)
//   2. emit: the header, then SHORT_REPEAT (3..10 equal values: rle_v2/short_repeat.rs), DELTA with a fixed step (header
//      width 0, base as (zigzag) varint, step as signed varint: delta.rs:44-116), or DIRECT (direct.rs:39-65): values bit-packed
//      big-endian at the run's width, lane l packs values 8 l .. 8 l + 7 = `width` whole bytes.
// PATCHED_BASE is never chosen (it is an optimisation of DIRECT for outliers, not a different meaning).  What the decoders make
// of the stream -- the kernels of rle_expand.hip and the CPU oracle alike -- is the input, value for value (tests/test_gpu_encode.py).
#pragma once

struct EncRun {
  uint32_t bytes;    // size of the run in the stream
  uint8_t mode;      // 0 DIRECT, 1 DELTA (fixed step), 2 SHORT_REPEAT
  uint8_t width;     // DIRECT: bits per value; SHORT_REPEAT: bytes of the value
  uint16_t pad;
};

__device__ __forceinline__ uint64_t enc_zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
// widths the format can state (integer/util.rs:370-384): 1..24, 26, 28, 30, 32, 40, 48, 56, 64
__device__ __forceinline__ uint32_t enc_fixed_width(uint32_t bits) {
  if (bits <= 1) return 1;
  if (bits <= 24) return bits;
  if (bits <= 32) return (bits + 1) & ~1u;
  return (bits + 7) & ~7u;
}
__device__ __forceinline__ uint32_t enc_width_code(uint32_t w) { return w <= 24 ? w - 1 : (w <= 32 ? 24 + (w - 26) / 2 : 28 + (w - 40) / 8); }
__device__ __forceinline__ uint32_t enc_varint_len(uint64_t u) {
  uint32_t n = 1;
  while (u >= 0x80) {
    u >>= 7;
    n++;
  }
  return n;
}
__device__ __forceinline__ uint32_t enc_put_varint(uint8_t* p, uint64_t u) {
  uint32_t n = 0;
  while (u >= 0x80) {
    p[n++] = (uint8_t)(u | 0x80);
    u >>= 7;
  }
  p[n++] = (uint8_t)u;
  return n;
}

// One wavefront per run of up to 512 values: lane l holds values 8 l .. 8 l + 7 of the run.
struct EncLoad {
  uint64_t u[8];   // the values as the stream stores them (zigzag when signed)
  int64_t v[8];
  uint32_t n;      // values of this lane (0..8)
};
__device__ __forceinline__ EncLoad enc_load(const int64_t* values, uint64_t first, uint32_t len, int is_signed, uint32_t lane) {
  EncLoad e;
  const uint32_t lo = lane * 8;
  e.n = lo < len ? (len - lo < 8 ? len - lo : 8) : 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    e.v[i] = (uint32_t)i < e.n ? values[first + lo + i] : 0;
    e.u[i] = is_signed ? enc_zigzag(e.v[i]) : (uint64_t)e.v[i];
  }
  return e;
}

extern "C" __global__ void __launch_bounds__(64) rle2_enc_plan_kernel(const int64_t* values, uint64_t n, int is_signed, EncRun* runs, uint32_t n_runs) {
  const uint32_t run = blockIdx.x, lane = threadIdx.x;
  if (run >= n_runs) return;
  const uint64_t first = (uint64_t)run * 512;
  const uint32_t len = n - first < 512 ? (uint32_t)(n - first) : 512u;
  const EncLoad e = enc_load(values, first, len, is_signed, lane);
  uint64_t m = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) m |= e.u[i];
  for (int o = 32; o; o >>= 1) m |= (uint64_t)__shfl_xor((long long)m, o);
  const uint32_t bits = m ? 64u - (uint32_t)__builtin_clzll(m) : 1u;
  // an arithmetic progression?  step = v[1] - v[0] without overflow, and every later step the same
  const int64_t v0 = (int64_t)__shfl((long long)e.v[0], 0), v1 = (int64_t)__shfl((long long)e.v[1], 0);
  int64_t step = 0;
  bool ok = len >= 3 && !__builtin_sub_overflow(v1, v0, &step);
  const int64_t next0 = (int64_t)__shfl_down((long long)e.v[0], 1);  // first value of the lane above
  bool mine = true;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t idx = lane * 8 + i;
    if (idx + 1 < len) {
      const int64_t a = e.v[i], b = i < 7 ? e.v[i + 1 < 8 ? i + 1 : 7] : next0;
      int64_t d;
      if (__builtin_sub_overflow(b, a, &d) || d != step) mine = false;
    }
  }
  ok = ok && !__ballot(!mine);
  if (lane == 0) {
    EncRun r;
    r.pad = 0;
    if (ok && step == 0 && len <= 10) {
      const uint64_t u = is_signed ? enc_zigzag(v0) : (uint64_t)v0;
      const uint32_t w = u ? (64u - (uint32_t)__builtin_clzll(u) + 7) / 8 : 1u;
      r.mode = 2;
      r.width = (uint8_t)w;
      r.bytes = 1 + w;
    } else if (ok) {
      r.mode = 1;
      r.width = 0;
      r.bytes = 2 + enc_varint_len(is_signed ? enc_zigzag(v0) : (uint64_t)v0) + enc_varint_len(enc_zigzag(step));
    } else {
      const uint32_t w = enc_fixed_width(bits);
      r.mode = 0;
      r.width = (uint8_t)w;
      r.bytes = 2 + (uint32_t)(((uint64_t)len * w + 7) / 8);
    }
    runs[run] = r;
  }
}

// where every run starts: an exclusive scan of the sizes by one workgroup (a stripe's column is some thousands of runs)
extern "C" __global__ void __launch_bounds__(1024) rle2_enc_scan_kernel(const EncRun* runs, uint32_t n_runs, uint64_t* offsets) {
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (n_runs + 1023) / 1024;
  const uint32_t lo = t * per, hi = lo + per < n_runs ? lo + per : n_runs;
  uint64_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += runs[k].bytes;
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    uint64_t acc = 0;
    for (uint32_t k = 0; k < 1024; k++) {
      const uint64_t x = part[k];
      part[k] = acc;
      acc += x;
    }
    offsets[n_runs] = acc;
  }
  __syncthreads();
  uint64_t acc = part[t];
  for (uint32_t k = lo; k < hi; k++) {
    offsets[k] = acc;
    acc += runs[k].bytes;
  }
}

extern "C" __global__ void __launch_bounds__(64) rle2_enc_emit_kernel(const int64_t* values, uint64_t n, int is_signed, const EncRun* runs, const uint64_t* offsets,
                                                                     uint32_t n_runs, uint8_t* out) {
  const uint32_t run = blockIdx.x, lane = threadIdx.x;
  if (run >= n_runs) return;
  const uint64_t first = (uint64_t)run * 512;
  const uint32_t len = n - first < 512 ? (uint32_t)(n - first) : 512u;
  const EncRun r = runs[run];
  uint8_t* p = out + offsets[run];
  if (r.mode == 2) {
    if (lane == 0) {
      const int64_t v0 = values[first];
      const uint64_t u = is_signed ? enc_zigzag(v0) : (uint64_t)v0;
      p[0] = (uint8_t)(((r.width - 1u) << 3) | (len - 3u));
      for (uint32_t k = 0; k < r.width; k++) p[1 + k] = (uint8_t)(u >> (8 * (r.width - 1 - k)));  // big-endian
    }
    return;
  }
  if (r.mode == 1) {
    if (lane == 0) {
      const int64_t v0 = values[first], v1 = values[first + 1];
      p[0] = (uint8_t)((3u << 6) | ((len - 1) >> 8));  // width code 0: a fixed step
      p[1] = (uint8_t)((len - 1) & 0xff);
      uint32_t k = 2;
      k += enc_put_varint(p + k, is_signed ? enc_zigzag(v0) : (uint64_t)v0);
      k += enc_put_varint(p + k, enc_zigzag(v1 - v0));
    }
    return;
  }
  const uint32_t w = r.width;
  if (lane == 0) {
    p[0] = (uint8_t)((1u << 6) | (enc_width_code(w) << 1) | ((len - 1) >> 8));
    p[1] = (uint8_t)((len - 1) & 0xff);
  }
  const EncLoad e = enc_load(values, first, len, is_signed, lane);
  if (!e.n) return;
  // this lane's eight values are `w` whole bytes of the payload (the last lane of a short run: ceil(n w / 8) of them); byte j
  // holds bits [8 j, 8 j + 8) of the lane's big-endian bit string
  uint8_t* q = p + 2 + (uint64_t)lane * w;
  const uint32_t nbytes = (e.n * w + 7) / 8;
  for (uint32_t j = 0; j < nbytes; j++) {
    uint32_t byte = 0;
    // bit b of the string belongs to value b / w, its bit w - 1 - b % w
    const uint32_t b0 = 8 * j;
    uint32_t i = b0 / w;
    uint32_t used = b0 - i * w;  // bits of value i that lie before this byte
    uint32_t filled = 0;
    while (filled < 8 && i < 8) {
      const uint32_t take = w - used < 8 - filled ? w - used : 8 - filled;
      const uint64_t u = i < e.n ? (i == 0 ? e.u[0] : i == 1 ? e.u[1] : i == 2 ? e.u[2] : i == 3 ? e.u[3] : i == 4 ? e.u[4] : i == 5 ? e.u[5] : i == 6 ? e.u[6] : e.u[7]) : 0;
      const uint32_t piece = (uint32_t)((u >> (w - used - take)) & ((1u << take) - 1u));
      byte = (byte << take) | piece;
      filled += take;
      used += take;
      if (used == w) {
        used = 0;
        i++;
      }
    }
    byte <<= 8 - filled;
    q[j] = (uint8_t)byte;
  }
}
