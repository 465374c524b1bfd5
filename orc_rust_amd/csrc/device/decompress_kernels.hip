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
  uint32_t pad;
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
// (ring size per kernel: 64 KiB for Snappy / LZ4, whose matches reach 64 KiB back; 32 KiB next to the
// DEFLATE / Zstandard tables -- DEFLATE never reaches further)
template <uint32_t RING>
struct LzStore {
  uint8_t ring[RING];
  uint8_t stage[LZ_STAGE + 16];
  uint32_t g_len[64], g_off[64], g_src[64];  // elements waiting for lz_group_run (Snappy)
};
struct LzLds {  // what the decoders see of it
  uint8_t* ring;
  uint32_t rsize;
  uint8_t* stage;
  uint32_t *g_len, *g_off, *g_src;
};
template <uint32_t RING>
__device__ __forceinline__ LzLds lz_view(LzStore<RING>& s) {
  return LzLds{s.ring, RING, s.stage, s.g_len, s.g_off, s.g_src};
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

// ---- element groups: up to 64 short literals / matches executed together ----------------------------------
// The token stream is parsed wave-uniformly out of a 512-byte register window (lane l holds bytes
// 8l..8l+7; a token is fetched with two cross-lane reads, no memory latency) and each element is
// filed in one lane.  A group of 64 elements is then executed at once: a wave scan gives the output
// positions, every lane copies its own element (literals from the staged input, matches from the
// LDS ring).  Matches whose source overlaps the group's own output -- a few per cent in practice --
// wait and are then done one by one, in order; so are matches that reach behind the ring.
struct LzGroup {
  uint32_t len, off, src;  // this lane's element: off == 0: literal from input position src
  uint32_t n;              // elements filed (wave uniform)
};
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
  return lo | ((uint64_t)hi << 32);
}
struct LzWin {
  const uint8_t* src;
  uint32_t n;
  uint32_t wb;  // window base (wave uniform)
  uint64_t w;   // this lane's 8 bytes: src[wb + 8*lane ..]
};
__device__ __forceinline__ void lzwin_load(LzWin& W, uint32_t pos, uint32_t lane) {
  W.wb = pos;
  const uint64_t p = (uint64_t)pos + 8ull * lane;
  uint64_t v = 0;
  if (p + 8 <= W.n) {
    v = ld_u64(W.src + p);
  } else {
    for (uint32_t t = 0; t < 8; t++)
      if (p + t < W.n) v |= (uint64_t)W.src[p + t] << (8 * t);
  }
  W.w = v;
}
// 8 bytes at pos (zero behind the end of the input)
__device__ __forceinline__ uint64_t lzwin_peek(LzWin& W, uint32_t pos, uint32_t lane) {
  if (pos < W.wb || pos - W.wb > 496) lzwin_load(W, pos, lane);
  const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pos - W.wb));
  const uint32_t i = o >> 3, s = (o & 7) * 8;
  const uint64_t a = readlane_u64(W.w, i), b = readlane_u64(W.w, i + 1);
  return s ? (a >> s) | (b << (64 - s)) : a;
}

// execute the filed elements; returns nonzero on a malformed element
__device__ __forceinline__ int lz_group_run(LzGroup& G, LzIn& in, LzOut& o, uint64_t limit, uint32_t lane PROF_PARM) {
  if (!G.n) return 0;
  PROF_MARK(0);
  const bool act = lane < G.n;
  const uint32_t len = act ? G.len : 0;
  const uint32_t incl = wave_incl_scan_u32(len, lane);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint64_t base = o.out;
  const uint64_t my = base + incl - len;
  const bool is_match = act && G.off != 0;
  bool bad = is_match && (G.off > my);
  if (__ballot(bad) || base + total > limit) return 1;
#ifdef EXP_PARSEONLY
  o.out = base + total;
  o.flushed = o.out;
  G.n = 0;
  return 0;
#endif
  // literals come from the staged input: (re)stage when this group's literals are not all inside it
  const bool lit = act && G.off == 0;
  const bool lit_out = lit && (G.src < in.sb || G.src + len > in.sb + LZ_STAGE + 16);
  if (__ballot(lit_out)) {
    uint32_t lo = lit ? G.src : 0xffffffffu;
    for (int s = 32; s; s >>= 1) {
      uint32_t t = __shfl_xor(lo, s);
      lo = t < lo ? t : lo;
    }
    lzin_stage(in, lo, lane);
  }
  const bool lit_far = lit && (G.src < in.sb || G.src + len > in.sb + LZ_STAGE + 16);  // still outside: done one by one below
  const bool dep = is_match && (my - G.off + (len < G.off ? len : G.off) > base);
  const bool far = is_match && G.off > o.rmask + 1 - 2 * LZ_FLUSH;
  const bool later = dep || far || lit_far;
  lds_order();
  PROF_MARK(1);
  if (act && !later) {
    // eight bytes per step while neither side wraps around the ring (a match needs off >= 8 for that)
    const uint32_t d0 = (uint32_t)my & o.rmask;
    uint32_t k = 0;
    if (lit) {
      const uint8_t* s = in.stage + (G.src - in.sb);
      if (d0 + len <= o.rmask + 1) {
        for (; k + 8 <= len; k += 8) {
          uint64_t v;
          __builtin_memcpy(&v, s + k, 8);
          __builtin_memcpy(o.ring + d0 + k, &v, 8);
        }
      }
      for (; k < len; k++) o.ring[(uint32_t)(my + k) & o.rmask] = s[k];
    } else {
      const uint32_t s0 = (uint32_t)(my - G.off) & o.rmask;
      if (G.off >= 8 && d0 + len <= o.rmask + 1 && s0 + len <= o.rmask + 1) {
        for (; k + 8 <= len; k += 8) {
          uint64_t v;
          __builtin_memcpy(&v, o.ring + s0 + k, 8);
          __builtin_memcpy(o.ring + d0 + k, &v, 8);
        }
      }
      for (; k < len; k++) {
        const uint32_t sidx = len <= G.off ? k : k % G.off;
        o.ring[(uint32_t)(my + k) & o.rmask] = o.ring[(uint32_t)(my - G.off + sidx) & o.rmask];
      }
    }
  }
  lds_order();
  PROF_MARK(2);
  // the rest in order, all lanes on one element at a time
  unsigned long long m = __ballot(later);
  while (m) {
    const uint32_t i = (uint32_t)__builtin_ctzll(m);
    m &= m - 1;
    const uint32_t elen = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)i);
    const uint32_t eoff = (uint32_t)__builtin_amdgcn_readlane((int)G.off, (int)i);
    const uint32_t esrc = (uint32_t)__builtin_amdgcn_readlane((int)G.src, (int)i);
    const uint64_t eo = base + (uint32_t)__builtin_amdgcn_readlane((int)(incl - len), (int)i);
    if (eoff == 0) {
      for (uint32_t k = lane; k < elen; k += 64) o.ring[(uint32_t)(eo + k) & o.rmask] = in.src[esrc + k];
    } else if (eoff <= o.rmask + 1 - 2 * LZ_FLUSH) {
      for (uint32_t k = lane; k < elen; k += 64) {
        const uint32_t sidx = elen <= eoff ? k : k % eoff;
        o.ring[(uint32_t)(eo + k) & o.rmask] = o.ring[(uint32_t)(eo - eoff + sidx) & o.rmask];
      }
    } else {
      // behind the ring: the flushed output in memory has the bytes (the group itself is < 4 KiB)
      if (o.flushed != o.out) lz_flush(o, lane);
      wave_fence();
      for (uint32_t k = lane; k < elen; k += 64) o.ring[(uint32_t)(eo + k) & o.rmask] = o.dst[eo - eoff + k];
    }
    lds_order();
  }
  PROF_MARK(3);
  o.out = base + total;
  G.n = 0;
  lz_maybe_flush(o, lane);
  PROF_MARK(4);
  return 0;
}
__device__ __forceinline__ void lz_group_add(LzGroup& G, uint32_t len, uint32_t off, uint32_t src, uint32_t lane) {
  if (lane == G.n) {
    G.len = len;
    G.off = off;
    G.src = src;
  }
  G.n++;
}

// ---- Snappy raw (compression.rs:161-172) --------------------------------------------------------------
// A single wavefront issues one instruction every four cycles, so a token-at-a-time scalar parse
// costs ~500 cycles per element whatever the code looks like.  Instead all 64 lanes decode, at
// once, the element that WOULD start at each of the next 64 input bytes (size, length, offset);
// the real chain is then followed through those results with one cross-lane read per element, its
// members are compacted into the group table and executed 32-64 at a time by lz_group_run.
__device__ __forceinline__ int snappy_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t lane, uint32_t* out_len, LzLds Z PROF_PARM) {
  LzIn in{src, n, Z.stage, 0};
  lzin_stage(in, 0, lane);
  uint32_t pos = 0;
  uint64_t ulen = 0;
  int shift = 0;
  for (;;) {
    if (pos >= n || shift > 28) return 1;
    uint32_t c = Z.stage[pos];  // the preamble is at most 5 bytes
    pos++;
    ulen |= (uint64_t)(c & 0x7f) << shift;
    shift += 7;
    if (!(c & 0x80)) break;
  }
  if (ulen > cap) return 1;
  LzOut o{Z.ring, Z.rsize - 1, dst, 0, 0};
  LzGroup G{0, 0, 0, 0};
  uint32_t gn = 0;  // elements in the group table (wave uniform)
  auto run_group = [&]() -> int {
    lds_order();
    G.len = lane < gn ? Z.g_len[lane] : 0;
    G.off = lane < gn ? Z.g_off[lane] : 0;
    G.src = lane < gn ? Z.g_src[lane] : 0;
    G.n = gn;
    gn = 0;
    return lz_group_run(G, in, o, ulen, lane PROF_ARG);
  };
  pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
  while (pos < n) {
    PROF_MARK(9);
    if (pos < in.sb || pos + 72 > in.sb + LZ_STAGE + 16) lzin_stage(in, pos, lane);
    // ---- every lane: the element that would start at byte pos + lane ----
    const uint32_t q = pos + lane;
    uint64_t w;
    __builtin_memcpy(&w, in.stage + (q - in.sb), 8);  // zero behind the end of the input
    const uint32_t tag = (uint32_t)w & 0xff, t = tag & 3, L = tag >> 2;
    const uint32_t x = (uint32_t)(w >> 8);  // the four bytes after the tag
    const bool is_lit = t == 0;
    const uint32_t nb = L >= 60 ? L - 59 : 0;  // literal: extra length bytes
    const uint32_t ext = nb ? (x & (0xffffffffu >> (32 - 8 * nb))) : 0;
    const uint32_t len = is_lit ? (L >= 60 ? ext : L) + 1 : (t == 1 ? 4 + (L & 7) : L + 1);
    const uint32_t off = is_lit ? 0 : (t == 1 ? ((tag >> 5) << 8) | (x & 0xff) : (t == 2 ? (x & 0xffff) : x));
    const uint32_t hdr = is_lit ? 1 + nb : (t == 1 ? 2 : (t == 2 ? 3 : 5));
    const uint64_t end = (uint64_t)q + hdr + (is_lit ? len : 0);
    const bool bad = end > n || (!is_lit && off == 0) || (is_lit && nb && ext == 0xffffffffu);
    // what the chain walk needs per lane: bytes to the next element, and a stop flag
    const uint32_t flag = q >= n ? 4u : (bad ? 1u : ((is_lit && len > 64) ? 2u : 0u));
    const uint32_t hop = flag ? (flag << 16) : (uint32_t)(end - q);  // < 70 for the elements the walk steps over
    PROF_MARK(6);
    // ---- follow the chain from lane 0 ----
    unsigned long long members = 0;
    uint32_t cur = 0, stop = 0;
    {
      // (one short basic block per hop -- taken branches are what a lone wavefront pays most for: a
      // flagged lane's hop is >= 65536, which ends the loop by itself and is undone afterwards)
      uint32_t last = 0, h = 0;
      while (cur < 64) {
        h = (uint32_t)__builtin_amdgcn_readlane((int)hop, (int)cur);
        members |= 1ull << cur;
        last = cur;
        cur += h;
      }
      if (h >> 16) {
        stop = h >> 16;
        members &= ~(1ull << last);
        cur = last;
      }
    }
    PROF_MARK(7);
    // ---- members -> group table (stream order = lane order) ----
    if ((members >> lane) & 1) {
      const uint32_t slot = gn + (uint32_t)__popcll(members & ((1ull << lane) - 1));
      Z.g_len[slot] = len;
      Z.g_off[slot] = off;
      Z.g_src[slot] = q + hdr;
    }
    gn += (uint32_t)__popcll(members);
    if (stop & 1) return 1;
    if (stop & 2) {
      // long literal at lane `cur`: everything filed so far first, then all lanes on it
      const uint32_t llen = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)cur);
      const uint32_t lhdr = (uint32_t)__builtin_amdgcn_readlane((int)hdr, (int)cur);
      if (run_group()) return 1;
      if (o.out + llen > ulen) return 1;
      lz_literal(o, src + pos + cur + lhdr, llen, lane);
      pos += cur + lhdr + llen;
    } else {
      pos += cur;  // >= 64, or the end of the input
    }
    PROF_MARK(8);
    if (gn > 32 && run_group()) return 1;  // a window adds at most 32 elements (2 bytes each at least)
  }
  if (run_group()) return 1;
  if (o.out != ulen) return 1;
  lz_flush(o, lane);
  *out_len = (uint32_t)o.out;
  return 0;
}

// ---- LZ4 block (compression.rs:185-195) ------------------------------------------------------------------
// Same scheme as Snappy: every lane decodes the SEQUENCE (token, literal length, literals, offset,
// match length) that would start at its byte, the chain is followed with cross-lane reads, literal
// and match parts are filed as group elements.  Sequences with more than two length-extension bytes,
// literals or matches longer than 64 bytes take the one-at-a-time path.
__device__ __forceinline__ int lz4_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t lane, uint32_t* out_len, LzLds Z PROF_PARM) {
  if (n == 0) return 1;
  LzIn in{src, n, Z.stage, 0};
  lzin_stage(in, 0, lane);
  LzOut o{Z.ring, Z.rsize - 1, dst, 0, 0};
  LzGroup G{0, 0, 0, 0};
  uint32_t gn = 0;  // elements in the group table (wave uniform)
  auto run_group = [&]() -> int {
    lds_order();
    G.len = lane < gn ? Z.g_len[lane] : 0;
    G.off = lane < gn ? Z.g_off[lane] : 0;
    G.src = lane < gn ? Z.g_src[lane] : 0;
    G.n = gn;
    gn = 0;
    return lz_group_run(G, in, o, cap, lane PROF_ARG);
  };
  uint32_t pos = 0;
  bool done = false;
  while (!done) {
    if (pos >= n) return 1;  // a block ends with a literals-only sequence, never between sequences
    if (gn > 20 && run_group()) return 1;  // a 64-byte window holds at most 21 sequences = 42 elements
    // the stage must hold a short sequence starting anywhere in the window: 64 + token/extensions + 525 literal bytes + offset/extensions
    if (pos < in.sb || pos + 640 > in.sb + LZ_STAGE + 16) lzin_stage(in, pos, lane);
    const uint32_t q = pos + lane;
    uint32_t flag = 0, adv = 0, ll = 0, ml = 0, off = 0, lsrc = 0;
    if (q >= n) {
      flag = 4;
    } else {
      uint64_t w;
      __builtin_memcpy(&w, in.stage + (q - in.sb), 8);
      const uint32_t tok = (uint32_t)w & 0xff;
      ll = tok >> 4;
      uint32_t p = 1;  // bytes of the sequence consumed so far
      if (ll == 15) {
        const uint32_t e0 = (uint32_t)(w >> 8) & 0xff, e1 = (uint32_t)(w >> 16) & 0xff;
        ll += e0;
        p = 2;
        if (e0 == 255) {
          ll += e1;
          p = 3;
          if (e1 == 255) flag = 2;  // longer extension: one-at-a-time path
        }
      }
      lsrc = q + p;
      const uint64_t lit_end = (uint64_t)lsrc + ll;
      if (ll > 525) flag = 2;
      if (!flag) {
        if (lit_end > n) {
          flag = 1;
        } else if (lit_end == n) {
          flag = 8;  // last sequence: literals only
          adv = (uint32_t)(lit_end - q);
        } else if (lit_end + 2 > n) {
          flag = 1;
        } else {
          uint64_t w2;
          __builtin_memcpy(&w2, in.stage + ((uint32_t)lit_end - in.sb), 8);
          off = (uint32_t)w2 & 0xffff;
          ml = tok & 15;
          uint32_t p2 = 2;
          if (ml == 15) {
            const uint32_t e0 = (uint32_t)(w2 >> 16) & 0xff, e1 = (uint32_t)(w2 >> 24) & 0xff;
            ml += e0;
            p2 = 3;
            if (e0 == 255) {
              ml += e1;
              p2 = 4;
              if (e1 == 255) flag = 2;
            }
          }
          ml += 4;
          if (lit_end + p2 > n) flag = 1;  // extension bytes behind the end
          if (!flag && off == 0) flag = 1;
          adv = (uint32_t)(lit_end - q) + p2;
        }
      }
      if (!flag && (ll > 64 || ml > 64)) flag = 16;  // well-formed, but its parts are too long for a group element
    }
    const uint32_t hop = (flag & ~8u) ? (flag << 16) : adv;  // the last sequence (8) is a member; its hop ends the input
    // ---- follow the chain from lane 0 ----
    unsigned long long members = 0;
    uint32_t cur = 0, stop = 0;
    {
      uint32_t last = 0, h = 0;
      while (cur < 64) {
        h = (uint32_t)__builtin_amdgcn_readlane((int)hop, (int)cur);
        members |= 1ull << cur;
        last = cur;
        cur += h;
      }
      if (h >> 16) {
        stop = h >> 16;
        members &= ~(1ull << last);
        cur = last;
      }
    }
    // ---- members -> group table: literal part, then match part ----
    const bool mem = (members >> lane) & 1;
    const bool is_last = mem && (flag & 8);
    const uint32_t cnt = mem ? (ll ? 1u : 0u) + (is_last ? 0u : 1u) : 0u;
    const uint32_t incl = wave_incl_scan_u32(cnt, lane);
    if (mem) {
      uint32_t slot = gn + incl - cnt;
      if (ll) {
        Z.g_len[slot] = ll;
        Z.g_off[slot] = 0;
        Z.g_src[slot] = lsrc;
        slot++;
      }
      if (!is_last) {
        Z.g_len[slot] = ml;
        Z.g_off[slot] = off;
        Z.g_src[slot] = 0;
      }
    }
    gn += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (__ballot(is_last)) done = true;
    pos += cur;
    if (stop & 1) return 1;
    if (stop & (2 | 16)) {
      // one sequence the slow way (all lanes on it): everything filed so far first
      if (run_group()) return 1;
      const uint32_t tok = src[pos];
      pos++;
      uint32_t lit = tok >> 4;
      if (lit == 15) {
        uint32_t c;
        do {
          if (pos >= n) return 1;
          c = src[pos++];
          lit += c;
        } while (c == 255);
      }
      if ((uint64_t)pos + lit > n || o.out + lit > cap) return 1;
      lz_literal(o, src + pos, lit, lane);
      pos += lit;
      if (pos == n) {
        done = true;
      } else {
        if (pos + 2 > n) return 1;
        const uint32_t moff = src[pos] | ((uint32_t)src[pos + 1] << 8);
        pos += 2;
        uint32_t mlen = tok & 15;
        if (mlen == 15) {
          uint32_t c;
          do {
            if (pos >= n) return 1;
            c = src[pos++];
            mlen += c;
          } while (c == 255);
        }
        mlen += 4;
        if (moff == 0 || moff > o.out || o.out + mlen > cap) return 1;
        lz_match(o, moff, mlen, lane);
      }
    }
  }
  if (run_group()) return 1;
  if (pos != n) return 1;
  lz_flush(o, lane);
  *out_len = (uint32_t)o.out;
  return 0;
}

#include "inflate_device.h"
#include "zstd_device.h"
#include "zstd_entropy.h"
#include "lz_parse.h"
#include "lz_exec.h"

// One kernel per codec family (a single kernel with all of them inlined runs out of scalar registers and
// carries every family's LDS): `family` 0 = Snappy / LZ4, 1 = DEFLATE; Zstandard has its own two kernels
// (zstd_entropy.h, lz_exec.h).  Each launch covers the whole chunk table and takes the chunks of its
// family; "original" chunks (plain copies) belong to whichever family the host launches first (copy_too).
template <int FAMILY>
__device__ __forceinline__ void decompress_chunks_body(ChunkDesc* chunks, uint32_t n_chunks, int copy_too, DecompLds* tables, LzLds lz) {
  uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  uint32_t lane = threadIdx.x;
  PROF_BEGIN();
  ChunkDesc d = chunks[c];
  const bool mine = (d.kind == 0 && copy_too) || (FAMILY == 0 && (d.kind == 2 || d.kind == 4)) || (FAMILY == 1 && d.kind == 1) ||
                    (copy_too && d.kind != 0 && d.kind != 1 && d.kind != 2 && d.kind != 4 && d.kind != 5);
  if (!mine) return;
  d.src = as_global(d.src);  // plain global memory, not generic: see as_global()
  d.dst = (uint8_t*)as_global((void*)d.dst);
  d.scratch = (decltype(d.scratch))as_global((void*)d.scratch);
  uint32_t out_len = 0;
  int bad = 0;
  if (d.kind == 0) {
    if (d.src_len > d.dst_cap) bad = 1;
    else {
      wave_copy(d.dst, d.src, d.src_len, lane);
      out_len = d.src_len;
    }
  } else if (FAMILY == 0 && d.kind == 2) {
    bad = snappy_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len, lz PROF_ARG);
  } else if (FAMILY == 0 && d.kind == 4) {
    bad = lz4_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len, lz PROF_ARG);
  } else if (FAMILY == 1 && d.kind == 1) {
    bad = inflate_wave(d.src, d.src_len, d.dst, d.dst_cap, lane, &out_len, *tables, lz);
  } else {
    bad = 1;  // unknown compression kind
  }
  PROF_MARK(5);
  PROF_END_AT(48);
  if (lane == 0) {
    chunks[c].out_len = bad ? 0 : out_len;
    chunks[c].status = bad ? ORC_E_CODEC : 0;
  }
}
extern "C" __global__ void __launch_bounds__(64) decompress_lz_kernel(ChunkDesc* chunks, uint32_t n_chunks, int copy_too) {
  __shared__ __attribute__((aligned(16))) LzStore<65536> lz;
  decompress_chunks_body<0>(chunks, n_chunks, copy_too, nullptr, lz_view(lz));
}
extern "C" __global__ void __launch_bounds__(64) decompress_deflate_kernel(ChunkDesc* chunks, uint32_t n_chunks, int copy_too) {
  __shared__ DecompLds lds;
  __shared__ __attribute__((aligned(16))) LzStore<32768> lz;
  decompress_chunks_body<1>(chunks, n_chunks, copy_too, &lds, lz_view(lz));
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
