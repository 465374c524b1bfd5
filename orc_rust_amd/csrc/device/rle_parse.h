// rle_parse.h -- device-side run-header parsing for the ORC run-length codecs (gfx950).
//
// Every RLE-family stream (Integer RLE v2 / v1, byte RLE) is a serial chain of runs whose
// boundaries are data dependent.  Both passes of the decoder (the block walk that finds run
// boundaries, rle_scan.hip, and the cooperative expansion, rle_expand.hip) parse run headers
// with the functions below so that they always agree on run sizes and value counts.
//
// Behaviour follows the reference decoders:
//   RLE v2  src/encoding/integer/rle_v2/{mod.rs:112-146, short_repeat.rs:29-63, direct.rs:39-65,
//           patched_base.rs:38-151, delta.rs:44-116}, width tables integer/util.rs:370-421
//   RLE v1  src/encoding/integer/rle_v1.rs:54-159
//   byte    src/encoding/byte.rs:228-247
//   varint  src/encoding/integer/util.rs:475-527, zigzag :536-546, signed-msb :559-569
//
// All stream buffers in HBM carry >= ORC_PAD bytes of readable slack after their last byte, so
// 8-byte unaligned loads may run past the logical end (their extra bytes are never used).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ORC_PAD 32

// status codes shared with include/orcgpu.h (1:1 with OrcError variants, error.rs:31-174)
#define ORC_E_OK 0u
#define ORC_E_IO 1u
#define ORC_E_OUT_OF_SPEC 2u
#define ORC_E_VARINT 3u
#define ORC_E_TIMESTAMP 4u
#define ORC_E_OFFSET_OVERFLOW 5u
#define ORC_E_ARROW 8u
#define ORC_E_CODEC 9u
// flag on a reported code: the error is "the stream ran dry" whatever its kind (ORC_E_IO always means that).  The host
// turns such an error into the container's own one when the stream was cut short by a rejected / mis-framed chunk.
#define ORC_E_EOF 0x40u

enum : int { CODEC_RLE2 = 0, CODEC_RLE1 = 1, CODEC_BYTE = 2 };
enum : int { RT_SR = 0, RT_DIRECT = 1, RT_PATCHED = 2, RT_DELTA = 3,   // RLE v2 sub-encodings
             RT_V1_RUN = 4, RT_V1_LIT = 5, RT_B_RUN = 6, RT_B_LIT = 7 };

// Pointers read out of job structures are generic to the compiler, and generic ("flat") accesses
// count against the LDS counter as well as the memory counter: every wait for an LDS result would
// also wait for the stream loads and value stores in flight.  as_global() states what the host
// knows -- the pointer is device global memory -- in a way the optimiser cannot
// fold away: the cast to the global address space passes through an empty asm.
__device__ __forceinline__ const uint8_t* as_global(const uint8_t* p) {
  const __attribute__((address_space(1))) uint8_t* g = (const __attribute__((address_space(1))) uint8_t*)p;
  asm volatile("" : "+v"(g));
  return (const uint8_t*)g;
}
__device__ __forceinline__ void* as_global(void* p) {
  __attribute__((address_space(1))) uint8_t* g = (__attribute__((address_space(1))) uint8_t*)p;
  asm volatile("" : "+v"(g));
  return (void*)g;
}

template <typename T>
__device__ __forceinline__ T* glob(T* p) {  // typed as_global()
  return (T*)as_global((void*)const_cast<typename std::remove_const<T>::type*>(p));
}

// n (at most 8) bytes of w into LDS at q, any alignment: one to four stores instead of a store per byte (a lane's loop over its
// bytes runs as long as the longest of the wavefront's 64)
__device__ __forceinline__ void lds_put8(uint8_t* q, uint64_t w, uint32_t n) {
  if (n >= 8) {
    __builtin_memcpy(q, &w, 8);
    return;
  }
  if (n & 4u) {
    const uint32_t x = (uint32_t)w;
    __builtin_memcpy(q, &x, 4);
    q += 4;
    w >>= 32;
  }
  if (n & 2u) {
    const uint16_t x = (uint16_t)w;
    __builtin_memcpy(q, &x, 2);
    q += 2;
    w >>= 16;
  }
  if (n & 1u) *q = (uint8_t)w;
}

__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// big-endian 8 bytes at p
__device__ __forceinline__ uint64_t ld_be64(const uint8_t* p) { return __builtin_bswap64(ld_u64(p)); }

// Value `idx` of an MSB-first bit-packed array of `w`-bit values starting at byte pointer p
// (integer/util.rs:44-218).  Widths above 32 are byte multiples in every ORC width table, so
// w + (bit offset & 7) <= 64 always holds.
__device__ __forceinline__ uint64_t unpack_be(const uint8_t* p, uint32_t idx, uint32_t w) {
  uint64_t bit = (uint64_t)idx * w;
  uint64_t v = ld_be64(p + (bit >> 3));
  v <<= (bit & 7);
  return v >> (64 - w);
}

// integer/util.rs:370-384
__device__ __forceinline__ uint32_t rle2_width(uint32_t enc) {
  return enc <= 23 ? enc + 1 : (enc <= 27 ? 26 + 2 * (enc - 24) : 40 + 8 * (enc - 28));
}
// integer/util.rs:407-421
__device__ __forceinline__ uint32_t closest_fixed_bits(uint32_t n) {
  if (n == 0) return 1;
  if (n <= 24) return n;
  if (n <= 32) return (n + 1) & ~1u;
  return (n + 7) & ~7u;
}

__device__ __forceinline__ int64_t trunc_n(int64_t v, int nbits) {
  return nbits == 64 ? v : (nbits == 32 ? (int64_t)(int32_t)v : (nbits == 16 ? (int64_t)(int16_t)v : (int64_t)(int8_t)v));
}
// signed_zigzag_decode carried out in N bits (integer/util.rs:536-546)
__device__ __forceinline__ int64_t zigzag_n(uint64_t u, int nbits) {
  if (nbits < 64) u &= (1ull << nbits) - 1;
  return trunc_n((int64_t)((u >> 1) ^ (0 - (u & 1))), nbits);
}

// Base-128 varint as read_varint::<N> does it (integer/util.rs:475-498), byte by byte: a byte
// at index i is first read (EOF -> IoError), then rejected with VarintTooLarge if its offset
// 7*i >= nbits, then accumulated (groups shifted past the top of N are dropped).
// `avail` = bytes left in the stream from p.  Returns the number of bytes consumed; 0 means the
// stream ended first (IoError); *err = ORC_E_VARINT on VarintTooLarge.
__device__ __forceinline__ uint32_t varint_n(const uint8_t* p, uint64_t avail, int nbits, uint64_t* out, uint32_t* err) {
  uint64_t lo = ld_u64(p), hi = ld_u64(p + 8);
  uint64_t tl = ~lo & 0x8080808080808080ull, th = ~hi & 0x8080808080808080ull;
  uint32_t t;  // index of the terminator byte (first byte with the top bit clear)
  if (tl) t = (uint32_t)(__builtin_ctzll(tl) >> 3);
  else if (th) t = (uint32_t)(__builtin_ctzll(th) >> 3) + 8;
  else t = 16;
  uint32_t max_groups = (uint32_t)(nbits + 6) / 7;  // indices 0..max_groups-1 are acceptable
  uint32_t lim = avail < max_groups ? (uint32_t)avail : max_groups;
  if (t >= lim) {
    if (avail > max_groups) {
      *err = ORC_E_VARINT;
      return max_groups + 1;
    }
    return 0;
  }
  uint64_t v = 0;
  for (uint32_t i = 0; i <= t; i++) {  // t <= 9
    uint64_t b = (i < 8 ? (lo >> (8 * i)) : (hi >> (8 * (i - 8)))) & 0x7f;
    v |= b << (7 * i);
  }
  if (nbits < 64) v &= (1ull << nbits) - 1;
  *out = v;
  return t + 1;
}

// 24-byte window of the stream at a run header: all header fields (2-4 bytes + up to two varints)
// are extracted from registers, so a header costs ONE memory latency instead of a chain of
// dependent byte loads.
struct Win24 {
  uint64_t w0, w1, w2;
};
__device__ __forceinline__ Win24 ld_win24(const uint8_t* p) {
  Win24 w;
  w.w0 = ld_u64(p);
  w.w1 = ld_u64(p + 8);
  w.w2 = ld_u64(p + 16);
  return w;
}
__device__ __forceinline__ uint32_t win_byte(const Win24& w, uint32_t i) {
  uint64_t v = i < 8 ? w.w0 : (i < 16 ? w.w1 : w.w2);
  return (uint32_t)(v >> (8 * (i & 7))) & 0xff;
}
// 8 bytes of the window starting at byte offset off (<= 16); bytes past the window read as 0
__device__ __forceinline__ uint64_t win_u64(const Win24& w, uint32_t off) {
  uint64_t a = off < 8 ? w.w0 : (off < 16 ? w.w1 : w.w2);
  uint64_t b = off < 8 ? w.w1 : (off < 16 ? w.w2 : 0ull);
  uint32_t sh = 8 * (off & 7);
  return sh ? (a >> sh) | (b << (64 - sh)) : a;
}
// varint_n on a window: the varint starts at window byte `off` (off <= 12 so that 10 bytes are visible)
__device__ __forceinline__ uint32_t varint_win(const Win24& w, uint32_t off, uint64_t avail, int nbits, uint64_t* out, uint32_t* err) {
  uint64_t lo = win_u64(w, off), hi = win_u64(w, off + 8);
  uint64_t tl = ~lo & 0x8080808080808080ull, th = ~hi & 0x8080808080808080ull;
  uint32_t t;
  if (tl) t = (uint32_t)(__builtin_ctzll(tl) >> 3);
  else if (th) t = (uint32_t)(__builtin_ctzll(th) >> 3) + 8;
  else t = 16;
  uint32_t max_groups = (uint32_t)(nbits + 6) / 7;
  uint32_t lim = avail < max_groups ? (uint32_t)avail : max_groups;
  if (t >= lim) {
    if (avail > max_groups) {
      *err = ORC_E_VARINT;
      return max_groups + 1;
    }
    return 0;
  }
  uint64_t v = 0;
  for (uint32_t i = 0; i <= t; i++) {
    uint64_t b = (i < 8 ? (lo >> (8 * i)) : (hi >> (8 * (i - 8)))) & 0x7f;
    v |= b << (7 * i);
  }
  if (nbits < 64) v &= (1ull << nbits) - 1;
  *out = v;
  return t + 1;
}

// One parsed run.  `size` and `n` are what the block walk needs; the rest feeds expansion.
struct RunHdr {
  uint32_t size;      // bytes of the whole run (header + payload); clamped to `avail` when truncated
  uint32_t n;         // values in the run
  uint32_t type;      // RT_*
  uint32_t width;     // bit width of packed values (SR: byte width * 8, DELTA: 0 = fixed delta)
  uint32_t payload;   // byte offset of the packed values from the run start
  uint32_t err;       // ORC_E_* detected while parsing (truncation = ORC_E_IO)
  int64_t base;       // SR value / DELTA base / PATCHED base / v1 run base / byte-run value
  int64_t delta;      // DELTA delta_base (signed) / v1 run delta
  // PATCHED_BASE extras
  uint32_t pw, pgw, pl, cw, patch_off;
};

// Parse the RLE v2 run whose header byte is p[0]; avail >= 1 bytes remain from p.
template <bool FULL>
__device__ __forceinline__ void rle2_parse(const uint8_t* p, uint64_t avail, bool is_signed, int nbits, RunHdr& h) {
  const Win24 win = ld_win24(p);
  uint32_t h0 = win_byte(win, 0);
  h.err = 0;
  h.type = h0 >> 6;
  h.base = 0;
  h.delta = 0;
  if (h.type == RT_SR) {
    uint32_t bw = ((h0 >> 3) & 7) + 1;
    h.n = (h0 & 7) + 3;
    h.width = bw * 8;
    h.payload = 1;
    h.size = 1 + bw;
    if ((uint32_t)nbits < h.width) h.err = ORC_E_OUT_OF_SPEC;          // short_repeat.rs:46-52 (before any read)
    else if (h.size > avail) h.err = ORC_E_IO;
    if (FULL && !h.err) {
      uint64_t v = __builtin_bswap64(win_u64(win, 1)) >> (64 - 8 * bw);
      h.base = is_signed ? zigzag_n(v, nbits) : trunc_n((int64_t)v, nbits);
    }
  } else if (h.type == RT_DIRECT) {
    uint32_t w = rle2_width((h0 >> 1) & 31);
    h.width = w;
    h.payload = 2;
    if ((uint32_t)nbits < w) {                                           // direct.rs:47-52 (before the 2nd header byte)
      h.err = ORC_E_OUT_OF_SPEC;
      h.n = 0;
      h.size = 1;
    } else if (avail < 2) {
      h.err = ORC_E_IO;
      h.n = 0;
      h.size = 1;
    } else {
      h.n = (((h0 & 1) << 8) | win_byte(win, 1)) + 1;
      h.size = 2 + ((h.n * w + 7) >> 3);
      if (h.size > avail) h.err = ORC_E_IO;
    }
  } else if (h.type == RT_PATCHED) {
    uint32_t w = rle2_width((h0 >> 1) & 31);
    h.width = w;
    if (avail < 4) {
      h.err = ORC_E_IO;
      h.n = 0;
      h.size = (uint32_t)avail;
    } else {
      h.n = (((h0 & 1) << 8) | win_byte(win, 1)) + 1;
      uint32_t b2 = win_byte(win, 2), b3 = win_byte(win, 3);
      uint32_t bw = ((b2 >> 5) & 7) + 1;
      h.pw = rle2_width(b2 & 31);
      h.pgw = ((b3 >> 5) & 7) + 1;
      h.pl = b3 & 31;
      h.cw = closest_fixed_bits(h.pw + h.pgw);
      h.payload = 4 + bw;
      h.patch_off = h.payload + ((h.n * w + 7) >> 3);
      h.size = h.patch_off + ((h.pl * h.cw + 7) >> 3);
      // in the order the reference meets them: header, base bytes, first value, payload, patch list
      if (h.pw + h.pgw > 64) {                                           // patched_base.rs:61-67 (after 4 header bytes)
        h.err = ORC_E_OUT_OF_SPEC;
        h.size = 4;
      } else if (4 + (uint64_t)bw > avail) {
        h.err = ORC_E_IO;                                                // i64::read_big_endian(base_byte_width)
      } else if ((w & 7) == 0 && w > (uint32_t)nbits) {
        h.err = ORC_E_OUT_OF_SPEC;                                       // read_big_endian::<N> with too many bytes (panics in the reference)
      } else if (h.size > avail) {
        h.err = ORC_E_IO;
      } else if (h.pl == 0) {
        h.err = ORC_E_OUT_OF_SPEC;                                       // patches[0] (index panic in the reference)
      }
      if (FULL && !h.err) {
        uint64_t b = __builtin_bswap64(win_u64(win, 4)) >> (64 - 8 * bw);
        int64_t base;
        if (is_signed) {                                                 // signed_msb_decode (integer/util.rs:559-569)
          uint64_t m = 1ull << (bw * 8 - 1);
          base = (b & m) ? (int64_t)(0 - (b & ~m)) : (int64_t)b;
        } else {
          base = (int64_t)b;
        }
        h.base = trunc_n(base, nbits);
      }
    }
  } else {  // RT_DELTA
    uint32_t enc = (h0 >> 1) & 31;
    uint32_t w = enc == 0 ? 0 : rle2_width(enc);
    h.width = w;
    h.n = 0;
    if (avail < 2) {
      h.err = ORC_E_IO;
      h.size = (uint32_t)avail;
    } else {
      uint32_t n = (((h0 & 1) << 8) | win_byte(win, 1)) + 1;
      uint64_t ub = 0, ud = 0;
      uint32_t e1 = 0, e2 = 0;
      uint32_t l1 = varint_win(win, 2, avail - 2, nbits, &ub, &e1);
      uint32_t l2 = (l1 && !e1) ? varint_win(win, 2 + l1, avail - 2 - l1, 64, &ud, &e2) : 0;
      if (!l1 || (!e1 && !l2)) {
        h.err = ORC_E_IO;
      } else if (e1 || e2) {
        h.err = ORC_E_VARINT;
      } else {
        h.n = n;
        h.payload = 2 + l1 + l2;
        h.size = h.payload + (w ? (((n - 2) * w + 7) >> 3) : 0);
        if (w && n < 2) {
          h.err = ORC_E_OUT_OF_SPEC;                                     // delta.rs:95 `length - 2` underflow
          h.size = h.payload;
        } else if (h.size > avail) {
          // delta.rs:80-93 takes the first step base +/- |delta_base| (checked in N) before it reads the packed deltas
          h.err = ORC_E_IO;
          const int64_t b0 = is_signed ? zigzag_n(ub, nbits) : trunc_n((int64_t)ub, nbits);
          const int64_t db = zigzag_n(ud, 64);
          const int64_t mag = db < 0 ? (int64_t)(0 - (uint64_t)db) : db;
          const int64_t v1 = (int64_t)(db > 0 ? (uint64_t)b0 + (uint64_t)mag : (uint64_t)b0 - (uint64_t)mag);
          const bool ovf = db > 0 ? ((b0 ^ v1) & (mag ^ v1)) < 0 : ((b0 ^ mag) & (b0 ^ v1)) < 0;
          if (ovf || trunc_n(v1, nbits) != v1) h.err = ORC_E_OUT_OF_SPEC;
        }
        if (FULL) {
          h.base = is_signed ? zigzag_n(ub, nbits) : trunc_n((int64_t)ub, nbits);
          h.delta = zigzag_n(ud, 64);
        }
      }
    }
  }
  if (h.err) {
    // a failing run produces no values and the reference stops there: end the chain
    h.n = 0;
    h.size = (uint32_t)avail;
  }
}

// RLE v1 (rle_v1.rs:54-68, :90-132)
template <bool FULL>
__device__ __forceinline__ void rle1_parse(const uint8_t* p, uint64_t avail, bool is_signed, int nbits, RunHdr& h) {
  const Win24 win = ld_win24(p);
  int32_t h0 = (int8_t)win_byte(win, 0);
  h.err = 0;
  h.base = 0;
  h.delta = 0;
  h.width = 0;
  if (h0 >= 0) {
    h.type = RT_V1_RUN;
    h.n = (uint32_t)h0 + 3;
    if (avail < 2) {
      h.err = ORC_E_IO;
    } else {
      uint64_t ub = 0;
      uint32_t e = 0;
      uint32_t l = varint_win(win, 2, avail - 2, nbits, &ub, &e);
      if (!l) h.err = ORC_E_IO;
      else if (e) h.err = ORC_E_VARINT;
      h.size = 2 + l;
      h.payload = 2;
      if (FULL && !h.err) {
        h.base = is_signed ? zigzag_n(ub, nbits) : trunc_n((int64_t)ub, nbits);
        h.delta = (int8_t)win_byte(win, 1);
      }
    }
  } else {
    h.type = RT_V1_LIT;
    h.n = (uint32_t)(-h0);
    h.payload = 1;
    // literal varints: walk to the end of the n-th terminator byte (read_literals, rle_v1.rs:90-100)
    uint32_t pos = 1, left = h.n;
    uint32_t max_groups = (uint32_t)(nbits + 6) / 7, cur = 0;
    while (left && pos < avail && !h.err) {
      uint64_t v = ld_u64(p + pos);
      uint32_t take = (uint32_t)((avail - pos) < 8 ? (avail - pos) : 8);
      for (uint32_t i = 0; i < take && left; i++) {
        cur++;
        if (cur > max_groups) {
          h.err = ORC_E_VARINT;
          break;
        }
        if (!((v >> (8 * i)) & 0x80)) {
          cur = 0;
          left--;
        }
        pos++;
      }
    }
    if (left && !h.err) h.err = ORC_E_IO;
    h.size = pos;
  }
  if (h.err) {
    h.n = 0;
    h.size = (uint32_t)avail;
  }
}

// byte RLE (byte.rs:228-247)
template <bool FULL>
__device__ __forceinline__ void byte_parse(const uint8_t* p, uint64_t avail, RunHdr& h) {
  const uint64_t w0 = ld_u64(p);
  uint32_t h0 = (uint32_t)w0 & 0xff;
  h.err = 0;
  h.width = 8;
  h.delta = 0;
  h.payload = 1;
  if (h0 < 0x80) {
    h.type = RT_B_RUN;
    h.n = h0 + 3;
    h.size = 2;
    h.base = 0;
    if (avail < 2) h.err = ORC_E_IO;
    else if (FULL) h.base = (int8_t)((w0 >> 8) & 0xff);
  } else {
    h.type = RT_B_LIT;
    h.n = 0x100 - h0;
    h.size = 1 + h.n;
    h.base = 0;
    if (h.size > avail) h.err = ORC_E_IO;
  }
  if (h.err) {
    h.n = 0;
    h.size = (uint32_t)avail;
  }
}

template <int CODEC, bool FULL>
__device__ __forceinline__ void run_parse(const uint8_t* p, uint64_t avail, bool is_signed, int nbits, RunHdr& h) {
  if (CODEC == CODEC_RLE2) rle2_parse<FULL>(p, avail, is_signed, nbits, h);
  else if (CODEC == CODEC_RLE1) rle1_parse<FULL>(p, avail, is_signed, nbits, h);
  else byte_parse<FULL>(p, avail, h);
}

// Lean parse of an RLE v2 header out of a 24-byte window (no memory access): size and count of the
// run.  Returns false for anything unusual (the caller then takes the full parse).
__device__ __forceinline__ bool rle2_lean(const Win24& win, int nbits, uint32_t& sz, uint32_t& cnt) {
  const uint32_t h0 = (uint32_t)win.w0 & 0xff, b1 = (uint32_t)(win.w0 >> 8) & 0xff;
  const uint32_t t = h0 >> 6, enc = (h0 >> 1) & 31;
  const uint32_t nn = (((h0 & 1) << 8) | b1) + 1;
  bool ok = false;
  sz = 0, cnt = 0;
  if (t == RT_SR) {
    uint32_t bw = ((h0 >> 3) & 7) + 1;
    sz = 1 + bw;
    cnt = (h0 & 7) + 3;
    ok = bw * 8 <= (uint32_t)nbits;
  } else if (t == RT_DIRECT) {
    uint32_t w = rle2_width(enc);
    sz = 2 + ((nn * w + 7) >> 3);
    cnt = nn;
    ok = w <= (uint32_t)nbits;
  } else if (t == RT_PATCHED) {
    uint32_t w = rle2_width(enc);
    uint32_t b2 = (uint32_t)(win.w0 >> 16) & 0xff, b3 = (uint32_t)(win.w0 >> 24) & 0xff;
    uint32_t bw = (b2 >> 5) + 1, pw = rle2_width(b2 & 31), pgw = (b3 >> 5) + 1, pl = b3 & 31;
    uint32_t cw = closest_fixed_bits(pw + pgw);
    sz = 4 + bw + ((nn * w + 7) >> 3) + ((pl * cw + 7) >> 3);
    cnt = nn;
    ok = pw + pgw <= 64 && pl != 0 && !((w & 7) == 0 && w > (uint32_t)nbits);
  } else {
    uint32_t w = enc == 0 ? 0 : rle2_width(enc);
    // two varints starting at byte 2: lengths from the terminator bits of bytes 2..21
    uint64_t v0 = win_u64(win, 2), v1 = win_u64(win, 10);
    uint64_t t0 = ~v0 & 0x8080808080808080ull, t1 = ~v1 & 0x8080808080808080ull;
    uint32_t l1 = t0 ? (uint32_t)(__builtin_ctzll(t0) >> 3) + 1 : (t1 ? (uint32_t)(__builtin_ctzll(t1) >> 3) + 9 : 99);
    uint32_t maxg = (uint32_t)(nbits + 6) / 7;
    if (l1 <= maxg && l1 <= 10) {
      uint64_t u0 = win_u64(win, 2 + l1), u1 = win_u64(win, 10 + l1);
      uint64_t s0 = ~u0 & 0x8080808080808080ull, s1 = ~u1 & 0x8080808080808080ull;
      uint32_t l2 = s0 ? (uint32_t)(__builtin_ctzll(s0) >> 3) + 1 : (s1 ? (uint32_t)(__builtin_ctzll(s1) >> 3) + 9 : 99);
      if (l2 <= 10 && 2 + l1 + l2 <= 22) {
        sz = 2 + l1 + l2 + (w ? (((nn - 2) * w + 7) >> 3) : 0);
        cnt = nn;
        ok = !(w && nn < 2);
      }
    }
  }
  return ok;
}

// The two commonest RLE v2 headers -- SHORT_REPEAT and DIRECT -- from their first two bytes alone, branch-free: size and value
// count of the run.  False for the other sub-encodings and for anything the full parse would reject (the caller then takes it).
// What the expansion's chain walk costs per run in short-run streams (dictionary keys, order keys): ~20 instructions instead of
// the full parse's few hundred, executed for the handful of lanes that own a block.
__device__ __forceinline__ bool rle2_hop2(uint32_t h0, uint32_t h1, int nbits, uint64_t avail, uint32_t& sz, uint32_t& n) {
  const uint32_t t = h0 >> 6;
  const uint32_t bw = ((h0 >> 3) & 7) + 1;
  const uint32_t w = rle2_width((h0 >> 1) & 31);
  const uint32_t nn = (((h0 & 1) << 8) | h1) + 1;
  const bool sr = t == RT_SR;
  sz = sr ? 1 + bw : 2 + ((nn * w + 7) >> 3);
  n = sr ? (h0 & 7) + 3 : nn;
  const uint32_t width = sr ? bw * 8 : w;
  return t <= RT_DIRECT && width <= (uint32_t)nbits && sz <= avail;
}

// ---- lean hop: (size, n) of the run at p, for the block walks ------------------------------------------
// Identical to run_parse<CODEC, false> whenever that succeeds; anything unusual (errors, the last
// bytes of the stream) is delegated to run_parse so that both always agree.
template <int CODEC>
__device__ __forceinline__ void hop_parse(const uint8_t* p, uint64_t avail, bool is_signed, int nbits, uint32_t& size, uint32_t& n, uint32_t& err) {
  if (CODEC == CODEC_RLE2 && avail >= 32) {
    const Win24 win = ld_win24(p);
    uint32_t sz = 0, cnt = 0;
    if (rle2_lean(win, nbits, sz, cnt) && sz <= avail) {
      size = sz;
      n = cnt;
      err = 0;
      return;
    }
  }
  if (CODEC == CODEC_BYTE && avail >= 130) {
    uint32_t h0 = p[0];
    size = h0 < 0x80 ? 2u : 1u + (0x100u - h0);
    n = h0 < 0x80 ? h0 + 3 : 0x100u - h0;
    err = 0;
    return;
  }
  RunHdr h;
  run_parse<CODEC, false>(p, avail, is_signed, nbits, h);
  size = h.size;
  n = h.n;
  err = h.err;
}
