// decompress_kernels.hip -- ORC chunk block decompressors.
//
// Replaces DecompressorVariant::decompress_block (src/compression.rs:142-195), i.e. the crates
// flate2 (raw DEFLATE), snap (raw Snappy), lz4_flex (LZ4 block) and zstd (Zstandard frame), with
// decoders written against the published formats (RFC 1951, Snappy format description, LZ4 block
// format, RFC 8878).  Chunk framing (3-byte headers, compression.rs:113-123, :244-267) is scanned
// on the host while staging; every chunk -- compressed or "original" -- becomes one ChunkDesc.
//
//   Zstandard     zstd_entropy.h (one wavefront per block: FSE / Huffman) -> lz_exec.h (one workgroup per chunk: the copies);
//                 at table scale the sequences go one LANE per block: zstd_lanes.h
//   Snappy, LZ4   lz_parse.h (one workgroup per chunk: the tokens)          -> lz_exec.h
//   DEFLATE       inflate_device.h: one wavefront per chunk, Huffman decode and copies through an LDS ring (this file)
//   LZO           lzo_device.h: one wavefront per chunk, through the same LDS ring
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
  uint32_t status;      // 0 ok, else ORC_E_CODEC (Zstandard: preset by the host when the frame / block headers do not parse)
  uint32_t first_item, n_items;  // Zstandard: the chunk's blocks in the ZItem table (lz_exec.h)
  uint32_t diag;        // diagnostics: where a rejected chunk failed
  uint32_t pad;
};

struct StreamDesc {
  uint32_t first_chunk, n_chunks;
  uint32_t len_idx;     // scalar receiving the plain length
  uint32_t err_idx;     // scalar receiving a codec error flag
  uint8_t* base;        // start of the stream's plain buffer
  uint32_t framing_error;
  uint32_t skip;        // a stream entered at a row group: bytes of its first chunk that belong to the rows before (the plain length
                        // published is what lies behind them; the consumers start there)
};

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
__device__ __forceinline__ void lds_order() {  // LDS accesses of one wavefront execute in order; this only pins the compiler
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

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

// ---- output window in LDS ---------------------------------------------------------------------------------
// A match that reads what the same wavefront stored a moment ago would need the stores to reach
// L2 and a fence before every match (microseconds per token).  Instead the most recent 32-64 Ki
// bytes of the output live in an LDS ring: literals and matches are written there, matches read
// from there (LDS accesses of one wavefront are in order), and the ring is flushed to the chunk's
// slot in HBM in 4 KiB pieces of coalesced stores.  Only a match that reaches further back than the
// ring goes to memory (flush + fence first).
#define LZ_FLUSH 4096u
#define LZ_STAGE 4096u
// (ring size: 32 KiB next to the DEFLATE tables -- a DEFLATE match never reaches further back)
template <uint32_t RING>
struct LzStore {
  uint8_t ring[RING];
  uint8_t stage[LZ_STAGE + 16];
};
struct LzLds {  // what the decoders see of it
  uint8_t* ring;
  uint32_t rsize;
  uint8_t* stage;
};
template <uint32_t RING>
__device__ __forceinline__ LzLds lz_view(LzStore<RING>& s) {
  return LzLds{s.ring, RING, s.stage};
}
struct LzOut {
  uint8_t* ring;     // LDS
  uint32_t rmask;    // ring size - 1
  uint8_t* dst;      // the chunk's output slot
  uint64_t out;      // bytes produced
  uint64_t flushed;  // bytes already in dst
};
__device__ __forceinline__ void lz_flush(LzOut& o, uint32_t lane) {
  lds_order();
  const uint32_t n = (uint32_t)(o.out - o.flushed);
  for (uint32_t k = lane * 8; k < n; k += 512) {
    const uint32_t ri = (uint32_t)(o.flushed + k) & o.rmask;
    if (k + 8 <= n && ri + 8 <= o.rmask + 1) {
      uint64_t v;
      __builtin_memcpy(&v, o.ring + ri, 8);
      st_u64(o.dst + o.flushed + k, v);
    } else {
      for (uint32_t t = k; t < n && t < k + 8; t++) o.dst[o.flushed + t] = o.ring[(uint32_t)(o.flushed + t) & o.rmask];
    }
  }
  o.flushed = o.out;
}
__device__ __forceinline__ void lz_maybe_flush(LzOut& o, uint32_t lane) {
  if (o.out - o.flushed >= LZ_FLUSH) lz_flush(o, lane);
}
// one byte (DEFLATE / Huffman literals)
__device__ __forceinline__ void lz_byte(LzOut& o, uint32_t b, uint32_t lane) {
  if (lane == 0) o.ring[(uint32_t)o.out & o.rmask] = (uint8_t)b;
  o.out++;
  lz_maybe_flush(o, lane);
}
// len literal bytes from src (global memory or LDS, not the ring)
__device__ __forceinline__ void lz_literal(LzOut& o, const uint8_t* src, uint32_t len, uint32_t lane) {
  for (uint32_t done = 0; done < len;) {
    const uint32_t piece = len - done < LZ_FLUSH ? len - done : LZ_FLUSH;
    for (uint32_t k = lane; k < piece; k += 64) o.ring[(uint32_t)(o.out + k) & o.rmask] = src[done + k];
    o.out += piece;
    done += piece;
    lz_maybe_flush(o, lane);
  }
}
// len copies of one byte (Zstandard RLE blocks)
__device__ __forceinline__ void lz_fill(LzOut& o, uint32_t v, uint32_t len, uint32_t lane) {
  for (uint32_t done = 0; done < len;) {
    const uint32_t piece = len - done < LZ_FLUSH ? len - done : LZ_FLUSH;
    for (uint32_t k = lane; k < piece; k += 64) o.ring[(uint32_t)(o.out + k) & o.rmask] = (uint8_t)v;
    o.out += piece;
    done += piece;
    lz_maybe_flush(o, lane);
  }
}
// len bytes from `off` bytes back (1 <= off <= out)
__device__ __forceinline__ void lz_match(LzOut& o, uint32_t off, uint32_t len, uint32_t lane) {
  lds_order();
  for (uint32_t done = 0; done < len;) {
    const uint32_t piece = len - done < LZ_FLUSH ? len - done : LZ_FLUSH;
    if (off <= o.rmask + 1 - 2 * LZ_FLUSH) {
      // the source is still in the ring (at most LZ_FLUSH + piece unflushed bytes are ahead of it)
      for (uint32_t k = lane; k < piece; k += 64) {
        const uint32_t s = off >= piece ? k : k % off;
        o.ring[(uint32_t)(o.out + k) & o.rmask] = o.ring[(uint32_t)(o.out - off + s) & o.rmask];
      }
    } else {
      // far match: everything produced so far goes to memory first, then the bytes come from there
      if (o.flushed != o.out) lz_flush(o, lane);
      wave_fence();
      for (uint32_t k = lane; k < piece; k += 64) {
        const uint32_t s = off >= piece ? k : k % off;
        o.ring[(uint32_t)(o.out + k) & o.rmask] = o.dst[o.out - off + s];
      }
    }
    lds_order();
    o.out += piece;
    done += piece;
    lz_maybe_flush(o, lane);
  }
}

// ---- staged input: the token stream is parsed out of LDS ------------------------------------------------
struct LzIn {
  const uint8_t* src;
  uint32_t n;
  uint8_t* stage;  // LDS, LZ_STAGE + 16 bytes
  uint32_t sb;     // stage holds src[sb, sb + LZ_STAGE) (clipped at n; zero behind it)
};
__device__ __forceinline__ void lzin_stage(LzIn& in, uint32_t pos, uint32_t lane) {
  lds_order();
  in.sb = pos;
  for (uint32_t k = lane * 16; k < LZ_STAGE + 16; k += 1024) {
    uint64_t v[2] = {0, 0};
    if ((uint64_t)pos + k + 16 <= in.n) {
      __builtin_memcpy(v, in.src + pos + k, 16);
    } else {
      for (uint32_t t = 0; t < 16; t++)
        if ((uint64_t)pos + k + t < in.n) reinterpret_cast<uint8_t*>(v)[t] = in.src[pos + k + t];
    }
    __builtin_memcpy(in.stage + k, v, 16);
  }
  lds_order();
}
// 8 bytes at pos (zero beyond n); restages when the window does not hold them
__device__ __forceinline__ uint64_t lzin_peek(LzIn& in, uint32_t pos, uint32_t lane) {
  if (pos < in.sb || pos + 8 > in.sb + LZ_STAGE + 16) lzin_stage(in, pos, lane);
  uint64_t v;
  __builtin_memcpy(&v, in.stage + (pos - in.sb), 8);
  return v;
}
__device__ __forceinline__ void lzin_literal(LzIn& in, LzOut& o, uint32_t pos, uint32_t len, uint32_t lane) {
  if (pos >= in.sb && pos + len <= in.sb + LZ_STAGE + 16) lz_literal(o, in.stage + (pos - in.sb), len, lane);
  else lz_literal(o, in.src + pos, len, lane);
}

#include "inflate_device.h"
#include "zstd_device.h"
#include "zstd_entropy.h"
#include "zstd_lanes.h"
#include "lz_parse.h"
#include "lz_exec.h"
#include "inflate_parse.h"
#include "lzo_device.h"

// DEFLATE chunks: one wavefront each ("original" chunks ride along when no other launch has taken them: copy_too).
// (`only_deferred`: the chunks inflate_parse_kernel / lz_exec_kernel left alone -- diag == LZX_DEFERRED; this decoder is the authority on
// malformed streams)
extern "C" __global__ void __launch_bounds__(64) decompress_deflate_kernel(ChunkDesc* chunks, uint32_t n_chunks, int copy_too, int only_deferred) {
  __shared__ DecompLds lds;
  __shared__ __attribute__((aligned(16))) LzStore<32768> lz;
  uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  uint32_t lane = threadIdx.x;
  PROF_BEGIN();
  ChunkDesc d = chunks[c];
  if (!(d.kind == 1 || (d.kind == 0 && copy_too))) return;
  if (d.kind == 1 && only_deferred && d.diag != LZX_DEFERRED) return;
  d.src = as_global(d.src);  // plain global memory, not generic: see as_global()
  d.dst = (uint8_t*)as_global((void*)d.dst);
  uint32_t out_len = 0;
  int bad = 0;
  if (d.kind == 0) {
    if (d.src_len > d.dst_cap) bad = 1;
    else {
      wave_copy(d.dst, d.src, d.src_len, lane);
      out_len = d.src_len;
    }
  } else {
    bad = inflate_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len, lds, lz_view(lz));
  }
  PROF_MARK(5);
  PROF_END_AT(48);
  if (lane == 0) {
    chunks[c].out_len = bad ? 0 : out_len;
    chunks[c].status = bad ? ORC_E_CODEC : 0;
  }
}

// LZO1X chunks: one wavefront each (compression.rs:174-183).
extern "C" __global__ void __launch_bounds__(64) decompress_lzo_kernel(ChunkDesc* chunks, uint32_t n_chunks) {
  __shared__ __attribute__((aligned(16))) LzStore<65536> lz;  // (an M4 match reaches 48 KiB back)
  const uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  const uint32_t lane = threadIdx.x;
  ChunkDesc d = chunks[c];
  if (d.kind != 3) return;
  d.src = as_global(d.src);
  d.dst = (uint8_t*)as_global((void*)d.dst);
  uint32_t out_len = 0;
  const int bad = lzo_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len, lz_view(lz));
  if (lane == 0) {
    chunks[c].out_len = bad ? 0 : out_len;
    chunks[c].status = bad ? ORC_E_CODEC : 0;
    chunks[c].diag = (uint32_t)bad;
  }
}

// One WAVEFRONT per stream: make the plain chunks contiguous (they already are unless a chunk in the middle came out shorter
// than its slot), publish the plain length, stop at the first bad chunk.
// chunk_start (optional): per chunk, where its plain bytes start in the stream (for the verified run starts of rle_hint_kernel)
// The chunks of a stream are looked at 64 at a time, one per lane: a stream of hundreds of chunks costs a few memory round trips,
// not one per chunk (round 5: a serial loop).  A workgroup of ONE wavefront, because this kernel starts while the other column
// lane's execution kernel -- single-wavefront workgroups, 24 a CU -- still fills the device: a 256-thread workgroup waits for four
// free slots on one CU, which that kernel never leaves (round 5 and the first parallel version alike: 3.7 ms in the queue at SF 12.5
// for 0.05 ms of work).  Only a chunk that has to move is handled by the whole wavefront, in stream order.
extern "C" __global__ void __launch_bounds__(64) decompress_finalize_kernel(const ChunkDesc* chunks, const StreamDesc* streams, uint64_t* scalars,
                                                                            uint32_t* chunk_start) {
  const StreamDesc s = streams[blockIdx.x];
  const uint32_t t = threadIdx.x;
  uint64_t total = 0;
  uint32_t err = s.framing_error ? ORC_E_IO : 0;
  for (uint32_t i0 = 0; i0 < s.n_chunks; i0 += 64) {
    const uint32_t i = i0 + t;
    uint32_t status = 0, out_len = 0;
    const uint8_t* dst = nullptr;
    if (i < s.n_chunks) {
      const ChunkDesc& c = chunks[s.first_chunk + i];
      status = c.status;
      out_len = c.out_len;
      dst = c.dst;
    }
    const unsigned long long bad_m = __ballot(status != 0);
    const uint32_t first_bad = bad_m ? (uint32_t)__builtin_ctzll(bad_m) : 64u;
    // inclusive scan of the tile's plain lengths (a chunk is at most a block size: 64 of them fit 32 bits)
    uint32_t incl = out_len;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      if (t >= d) incl += v;
    }
    const uint64_t start = total + incl - out_len;
    if (i < s.n_chunks && t <= first_bad && chunk_start) chunk_start[s.first_chunk + i] = (uint32_t)(start > 0xffffffffull ? 0xffffffffull : start);
    const unsigned long long move_m = __ballot(i < s.n_chunks && t < first_bad && out_len && dst != s.base + start);
    const uint32_t n_here = s.n_chunks - i0 < 64 ? s.n_chunks - i0 : 64;
    const uint32_t n_ok = first_bad < n_here ? first_bad : n_here;
    if (move_m) {
      // the rare case: chunk by chunk from the first one that is out of place
      const uint32_t first_move = (uint32_t)__builtin_ctzll(move_m);
      uint64_t at = total + (first_move ? (uint32_t)__shfl((int)incl, first_move - 1) : 0u);
      for (uint32_t k = first_move; k < n_ok; k++) {
        const uint32_t n = (uint32_t)__shfl((int)out_len, k);
        const uint8_t* from = chunks[s.first_chunk + i0 + k].dst;
        uint8_t* want = s.base + at;
        if (from != want && n) {
          // move left, ascending addresses: safe for overlapping ranges because want < from (a piece is read by every lane before
          // any lane writes it: loads and stores of one wavefront leave in order, the fence keeps the compiler from mixing them)
          for (uint32_t b = 0; b < n; b += 64 * 8) {
            const uint32_t p = b + t * 8;
            uint64_t v = 0;
            const uint32_t nb = p < n ? (n - p < 8 ? n - p : 8) : 0;
            if (nb == 8) v = ld_u64(from + p);
            else
              for (uint32_t q = 0; q < nb; q++) v |= (uint64_t)from[p + q] << (8 * q);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (nb == 8) st_u64(want + p, v);
            else
              for (uint32_t q = 0; q < nb; q++) want[p + q] = (uint8_t)(v >> (8 * q));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
          }
        }
        at += n;
      }
    }
    total += n_ok ? (uint32_t)__shfl((int)incl, n_ok - 1) : 0u;
    if (first_bad < n_here) {
      err = (uint32_t)__shfl((int)status, first_bad);
      break;
    }
  }
  if (t == 0) {
    scalars[s.len_idx] = total > s.skip ? total - s.skip : 0;
    if (err) scalars[s.err_idx] = err;
  }
}
