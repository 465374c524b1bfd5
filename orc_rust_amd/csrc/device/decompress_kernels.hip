// decompress_kernels.hip -- ORC chunk block decompressors, one wavefront per chunk.
//
// Replaces DecompressorVariant::decompress_block (src/compression.rs:142-195), i.e. the crates
// flate2 (raw DEFLATE), snap (raw Snappy), lz4_flex (LZ4 block) and zstd (Zstandard frame), with
// decoders written against the published formats (RFC 1951, Snappy format description, LZ4 block
// format, RFC 8878).  Chunk framing (3-byte headers, compression.rs:113-123, :244-267) is scanned
// on the host while staging; every chunk -- compressed or "original" -- becomes one ChunkDesc.
//
// All four formats are byte-serial LZ77 variants: the token stream is parsed wave-uniformly (every
// lane runs the same scalar parse), literal runs and matches are copied by all 64 lanes.  A match
// may read bytes the same wave stored a moment ago: a workgroup-scope fence before each match
// makes them visible (the wave stays on one CU, whose L1 is write-through).
#pragma once
#include "rle_parse.h"

struct ChunkDesc {
  const uint8_t* src;   // compressed payload
  uint8_t* dst;         // output slot
  uint8_t* scratch;     // per-chunk scratch (zstd literals), may be null
  uint32_t src_len;
  uint32_t dst_cap;
  uint32_t kind;        // 0 = original (copy), else ORCGPU_COMP_*
  uint32_t stream;      // index into the per-stream tables
  uint32_t out_len;     // written by the kernel
  uint32_t status;      // 0 ok, else ORC_E_CODEC
};

struct StreamDesc {
  uint32_t first_chunk, n_chunks;
  uint32_t len_idx;     // scalar receiving the plain length
  uint32_t err_idx;     // scalar receiving a codec error flag
  uint8_t* base;        // start of the stream's plain buffer
  uint32_t framing_error;
  uint32_t pad;
};

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ void st_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

// dst <- src (no overlap with anything the wave wrote), n bytes, all lanes
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t lane) {
  uint32_t k = lane * 8;
  for (; k + 8 <= n; k += 512) st_u64(dst + k, ld_u64(src + k));
  // tail: the last (n & 7) bytes plus whatever 8-byte slots were not reached
  uint32_t done = n & ~7u;
  for (uint32_t t = done + lane; t < n; t += 64) dst[t] = src[t];
}

// LZ77 match: copy len bytes from (out - off) to out inside dst; sources are < out
__device__ __forceinline__ void wave_match(uint8_t* dst, uint64_t out, uint32_t off, uint32_t len, uint32_t lane) {
  const uint8_t* s = dst + out - off;
  uint8_t* d = dst + out;
  if (off >= len) {
    if (off >= 8 && len >= 64) {
      uint32_t k = lane * 8;
      // 8-byte pieces are safe while a piece never reads what this call writes: off >= len
      for (; k + 8 <= len; k += 512) st_u64(d + k, ld_u64(s + k));
      for (uint32_t t = (len & ~7u) + lane; t < len; t += 64) d[t] = s[t];
    } else {
      for (uint32_t k = lane; k < len; k += 64) d[k] = s[k];
    }
  } else {
    for (uint32_t k = lane; k < len; k += 64) d[k] = s[k % off];
  }
}

// ---- Snappy raw (compression.rs:161-172) --------------------------------------------------------------
__device__ __forceinline__ int snappy_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t lane, uint32_t* out_len) {
  uint32_t pos = 0;
  uint64_t ulen = 0;
  int shift = 0;
  for (;;) {
    if (pos >= n || shift > 28) return 1;
    uint32_t c = src[pos++];
    ulen |= (uint64_t)(c & 0x7f) << shift;
    shift += 7;
    if (!(c & 0x80)) break;
  }
  if (ulen > cap) return 1;
  uint64_t out = 0;
  while (pos < n) {
    uint32_t tag = src[pos++];
    uint32_t len, off;
    uint32_t t = tag & 3;
    if (t == 0) {
      len = tag >> 2;
      if (len >= 60) {
        uint32_t nb = len - 59;
        if (pos + nb > n) return 1;
        len = 0;
        for (uint32_t i = 0; i < nb; i++) len |= (uint32_t)src[pos + i] << (8 * i);
        pos += nb;
      }
      len += 1;
      if ((uint64_t)pos + len > n || out + len > ulen) return 1;
      wave_copy(dst + out, src + pos, len, lane);
      pos += len;
      out += len;
      continue;
    }
    if (t == 1) {
      if (pos + 1 > n) return 1;
      len = 4 + ((tag >> 2) & 7);
      off = ((tag >> 5) << 8) | src[pos];
      pos += 1;
    } else if (t == 2) {
      if (pos + 2 > n) return 1;
      len = 1 + (tag >> 2);
      off = src[pos] | ((uint32_t)src[pos + 1] << 8);
      pos += 2;
    } else {
      if (pos + 4 > n) return 1;
      len = 1 + (tag >> 2);
      off = ld_u32(src + pos);
      pos += 4;
    }
    if (off == 0 || off > out || out + len > ulen) return 1;
    wave_fence();
    wave_match(dst, out, off, len, lane);
    out += len;
  }
  if (out != ulen) return 1;
  *out_len = (uint32_t)out;
  return 0;
}

// ---- LZ4 block (compression.rs:185-195) ------------------------------------------------------------------
__device__ __forceinline__ int lz4_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t lane, uint32_t* out_len) {
  if (n == 0) return 1;
  uint32_t pos = 0;
  uint64_t out = 0;
  for (;;) {
    if (pos >= n) return 1;
    uint32_t tok = src[pos++];
    uint32_t lit = tok >> 4;
    if (lit == 15) {
      uint32_t c;
      do {
        if (pos >= n) return 1;
        c = src[pos++];
        lit += c;
      } while (c == 255);
    }
    if ((uint64_t)pos + lit > n || out + lit > cap) return 1;
    wave_copy(dst + out, src + pos, lit, lane);
    pos += lit;
    out += lit;
    if (pos == n) break;
    if (pos + 2 > n) return 1;
    uint32_t off = src[pos] | ((uint32_t)src[pos + 1] << 8);
    pos += 2;
    uint32_t ml = tok & 15;
    if (ml == 15) {
      uint32_t c;
      do {
        if (pos >= n) return 1;
        c = src[pos++];
        ml += c;
      } while (c == 255);
    }
    ml += 4;
    if (off == 0 || off > out || out + ml > cap) return 1;
    wave_fence();
    wave_match(dst, out, off, ml, lane);
    out += ml;
  }
  *out_len = (uint32_t)out;
  return 0;
}

#include "inflate_device.h"
#include "zstd_device.h"

extern "C" __global__ void __launch_bounds__(64) decompress_chunks_kernel(ChunkDesc* chunks, uint32_t n_chunks) {
  __shared__ DecompLds lds;
  uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  uint32_t lane = threadIdx.x;
  ChunkDesc d = chunks[c];
  d.src = as_global(d.src);  // plain global memory, not generic: see as_global()
  d.dst = (uint8_t*)as_global((void*)d.dst);
  d.scratch = (decltype(d.scratch))as_global((void*)d.scratch);
  uint32_t out_len = 0;
  int bad = 0;
  switch (d.kind) {
    case 0:
      if (d.src_len > d.dst_cap) bad = 1;
      else {
        wave_copy(d.dst, d.src, d.src_len, lane);
        out_len = d.src_len;
      }
      break;
    case 1: bad = inflate_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len, lds); break;
    case 2: bad = snappy_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len); break;
    case 4: bad = lz4_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len); break;
    case 5: bad = zstd_wave(d.src, d.src_len, d.dst, d.dst_cap, d.scratch, lane, &out_len, lds); break;
    default: bad = 1;
  }
  if (lane == 0) {
    chunks[c].out_len = bad ? 0 : out_len;
    chunks[c].status = bad ? ORC_E_CODEC : 0;
  }
}

// One workgroup per stream: make the plain chunks contiguous (they already are unless a chunk in
// the middle came out shorter than its slot), publish the plain length, stop at the first bad chunk.
extern "C" __global__ void __launch_bounds__(256) decompress_finalize_kernel(const ChunkDesc* chunks, const StreamDesc* streams, uint64_t* scalars) {
  StreamDesc s = streams[blockIdx.x];
  uint64_t total = 0;
  uint32_t err = s.framing_error ? ORC_E_IO : 0;
  for (uint32_t i = 0; i < s.n_chunks; i++) {
    const ChunkDesc& c = chunks[s.first_chunk + i];
    if (c.status) {
      err = c.status;
      break;
    }
    uint8_t* want = s.base + total;
    if (c.dst != want && c.out_len) {
      // move left, ascending addresses: safe for overlapping ranges because want < c.dst
      for (uint32_t k = 0; k < c.out_len; k += 256 * 8) {
        uint32_t p = k + threadIdx.x * 8;
        uint64_t v = 0;
        uint32_t nb = p < c.out_len ? (c.out_len - p < 8 ? c.out_len - p : 8) : 0;
        for (uint32_t t = 0; t < nb; t++) v |= (uint64_t)c.dst[p + t] << (8 * t);
        __syncthreads();
        for (uint32_t t = 0; t < nb; t++) want[p + t] = (uint8_t)(v >> (8 * t));
        __syncthreads();
      }
    }
    total += c.out_len;
  }
  if (threadIdx.x == 0) {
    scalars[s.len_idx] = total;
    if (err) scalars[s.err_idx] = err;
  }
}
