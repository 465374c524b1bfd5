// lz_parse.h -- token stage of the Snappy / LZ4 block decompressors: ONE WORKGROUP PER ORC CHUNK.
//
// Replaces the parsing half of snap::raw::Decoder::decompress_vec and lz4_flex::block::decompress
// (compression.rs:161-172, :185-195).  Both formats are chains of variable-length elements (Snappy: a
// literal or a copy; LZ4: a sequence = literals + match) with no way to tell where an element starts but
// to follow the chain from the first one.  The work that does NOT depend on the chain -- what the element
// WOULD be at every byte position -- is done for all positions by all lanes; the chain itself is a scalar
// walk over those results (one cross-lane read per element).  The LZP_WAVES wavefronts of the workgroup
// take the chunk's 512-byte blocks in turn: a wavefront parses all 512 positions of its block while the
// chain is still busy in the blocks before it, waits for the position where the chain enters its block
// (handed from wavefront to wavefront through LDS), walks, hands on, and only then writes its tokens out.
//
// Output: one 8-byte record per token, in stream order -- {length | copy flag, copy offset or position of
// the literal bytes in the payload} -- executed by lz_exec_kernel (lz_exec.h), and per chunk the number of
// records, the declared length (Snappy) and a status.
#pragma once

#ifndef LZP_WAVES
#define LZP_WAVES 16
#endif
#ifndef LZP_SLEEP
#define LZP_SLEEP 1
#endif
#define LZP_BLK 512u
#define LZP_WIN 8
#define LZP_RING 32u
#define LZP_STOP 0x80000000u     // hop flag: the chain ends here (low bits: why)
#define LZP_END_POS 0xffffffffu  // handed on when the chain has ended (or failed)
#define LZ_COPY 0x80000000u      // record flag

struct LzpLds {
  uint32_t pos[LZP_RING];   // where the chain enters the block behind block b (slot b % LZP_RING)
  uint32_t cnt[LZP_RING];   // records before that block
  uint32_t tag[LZP_RING];   // b + 1 once the two above are valid
  uint32_t err;
};

// ---- what would start at byte q ----
// The 8 bytes at q (zero behind the end of the input); issued for all windows of a block before any is looked at.
__device__ __forceinline__ uint64_t lzp_fetch(const uint8_t* src, uint32_t n, uint64_t q) {
  return ld_u64(src + (q < n ? q : n));  // (ORC_PAD bytes of slack behind every stream)
}
__device__ __forceinline__ uint64_t lzp_mask(uint64_t w, uint32_t n, uint64_t q) {
  if (q >= n) return 0;
  return q + 8 > n ? w & (~0ull >> (8 * (q + 8 - n))) : w;
}
// hop: bytes to the next element (or LZP_STOP | reason); a: record word 0; b: record word 1
__device__ __forceinline__ void snappy_at(uint64_t w, uint32_t n, uint32_t q, uint32_t& hop, uint32_t& a, uint32_t& b) {
  const uint32_t tag = (uint32_t)w & 0xff, t = tag & 3, L = tag >> 2;
  const uint32_t x = (uint32_t)(w >> 8);  // the four bytes after the tag
  const bool is_lit = t == 0;
  const uint32_t nb = L >= 60 ? L - 59 : 0;  // literal: extra length bytes
  const uint32_t ext = nb ? (x & (0xffffffffu >> (32 - 8 * nb))) : 0;
  const uint32_t len = is_lit ? (L >= 60 ? ext : L) + 1 : (t == 1 ? 4 + (L & 7) : L + 1);
  const uint32_t off = is_lit ? 0 : (t == 1 ? ((tag >> 5) << 8) | (x & 0xff) : (t == 2 ? (x & 0xffff) : x));
  const uint32_t hdr = is_lit ? 1 + nb : (t == 1 ? 2 : (t == 2 ? 3 : 5));
  const uint64_t end = (uint64_t)q + hdr + (is_lit ? len : 0);
  const bool bad = end > n || (!is_lit && off == 0) || (is_lit && nb && ext == 0xffffffffu) || len >= LZ_COPY || off >= LZ_COPY;
  hop = q >= n ? (LZP_STOP | 2)  // the end of the input: a chain that lands here is complete
               : (bad ? (LZP_STOP | 1) : (uint32_t)(end - q));
  a = is_lit ? len : (len | LZ_COPY);
  b = is_lit ? q + hdr : off;
}

// LZ4: a sequence gives two records (literals, match); the block's last sequence only the first.
// First half (token, literal length): returns where the literals end (~0: no sequence can start here); a0/b0: literal record
__device__ __forceinline__ uint64_t lz4_head(const uint8_t* src, uint64_t w, uint32_t n, uint32_t q, uint32_t& a0, uint32_t& b0) {
  a0 = b0 = 0;
  if (q >= n) return ~0ull;
  uint64_t ll = (uint32_t)w >> 4 & 15;
  uint64_t p = (uint64_t)q + 1;
  if (ll == 15) {
    uint32_t c = (uint32_t)(w >> 8) & 0xff;
    if (p >= n) return ~0ull;
    ll += c;
    p++;
    if (c == 255) {
      // more extension bytes: rare, byte by byte
      do {
        if (p >= n) return ~0ull;
        c = src[p++];
        ll += c;
      } while (c == 255 && ll < (1ull << 31));
      if (ll >= (1ull << 31)) return ~0ull;
    }
  }
  const uint64_t lit_end = p + ll;
  if (lit_end > n) return ~0ull;
  a0 = (uint32_t)ll;
  b0 = (uint32_t)p;
  return lit_end;
}
// Second half (offset, match length) from the 8 bytes at lit_end; a1/b1: match record (a1 == 0: this is the last sequence)
__device__ __forceinline__ void lz4_tail(const uint8_t* src, uint64_t w, uint64_t w2, uint64_t lit_end, uint32_t n, uint32_t q, uint32_t& hop, uint32_t& a1,
                                         uint32_t& b1) {
  a1 = b1 = 0;
  hop = q >= n ? (LZP_STOP | 2) : (LZP_STOP | 1);
  if (lit_end == ~0ull) return;
  if (lit_end == n) {
    hop = (uint32_t)(lit_end - q);  // last sequence: literals only, the chain lands on the end of the input
    return;
  }
  if (lit_end + 2 > n) return;
  const uint32_t off = (uint32_t)w2 & 0xffff;
  uint64_t ml = (uint32_t)w & 15;
  uint64_t r = lit_end + 2;
  if (ml == 15) {
    uint32_t c = (uint32_t)(w2 >> 16) & 0xff;
    if (r >= n) return;
    ml += c;
    r++;
    if (c == 255) {
      do {
        if (r >= n) return;
        c = src[r++];
        ml += c;
      } while (c == 255 && ml < (1ull << 31));
      if (ml >= (1ull << 31)) return;
    }
  }
  ml += 4;
  // a sequence with a match is never the last one: something must follow it
  if (off == 0 || r >= n || ml >= LZ_COPY) return;
  a1 = (uint32_t)ml | LZ_COPY;
  b1 = off;
  hop = (uint32_t)(r - q);
}

template <int KIND>  // 2 Snappy, 4 LZ4
__device__ __forceinline__ void lz_parse_chunk(LzpLds& L, ChunkDesc* cd, const ChunkDesc& d, uint32_t tid) {
  const uint32_t lane = tid & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint8_t* src = as_global(d.src);
  const uint32_t n = d.src_len;
  uint2* recs = reinterpret_cast<uint2*>(as_global((void*)d.scratch));
  // ---- preamble ----
  uint32_t p0 = 0;
  uint64_t ulen = 0;
  int st = 0;
  if (KIND == 2) {
    int shift = 0;
    for (;;) {
      if (p0 >= n || shift > 28) {
        st = 1;
        break;
      }
      const uint32_t c = src[p0++];
      ulen |= (uint64_t)(c & 0x7f) << shift;
      shift += 7;
      if (!(c & 0x80)) break;
    }
    if (!st && ulen > d.dst_cap) st = 1;
  } else if (n == 0) {
    st = 1;
  }
  st = __builtin_amdgcn_readfirstlane(st);
  p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0);
  if (st) {
    if (tid == 0) {
      cd->status = ORC_E_CODEC;
      cd->diag = 20;
      cd->n_items = 0;
    }
    return;
  }
  // blocks cover [p0, n]: position n itself must be visited (that is where a complete chain lands)
  const uint32_t nblk = (n - p0) / LZP_BLK + 1;
  PROF_BEGIN();
  for (uint32_t b = wv; b < nblk; b += LZP_WAVES) {
    const uint32_t bstart = p0 + b * LZP_BLK;
    // ---- every position of the block ----
    uint32_t hop[LZP_WIN], ra[LZP_WIN], rb[LZP_WIN], rc[LZP_WIN], rd[LZP_WIN];
    {
      uint64_t w[LZP_WIN];
#pragma unroll
      for (int i = 0; i < LZP_WIN; i++) w[i] = lzp_fetch(src, n, (uint64_t)bstart + 64u * i + lane);
      if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < LZP_WIN; i++) {
          const uint32_t q = bstart + 64u * i + lane;
          snappy_at(lzp_mask(w[i], n, q), n, q, hop[i], ra[i], rb[i]);
          rc[i] = rd[i] = 0;
        }
      } else {
        uint64_t le[LZP_WIN], w2[LZP_WIN];
#pragma unroll
        for (int i = 0; i < LZP_WIN; i++) {
          const uint32_t q = bstart + 64u * i + lane;
          w[i] = lzp_mask(w[i], n, q);
          le[i] = lz4_head(src, w[i], n, q, ra[i], rb[i]);
        }
#pragma unroll
        for (int i = 0; i < LZP_WIN; i++) w2[i] = lzp_fetch(src, n, le[i]);
#pragma unroll
        for (int i = 0; i < LZP_WIN; i++) {
          const uint32_t q = bstart + 64u * i + lane;
          lz4_tail(src, w[i], lzp_mask(w2[i], n, le[i]), le[i], n, q, hop[i], rc[i], rd[i]);
        }
      }
    }
    // ---- per window: where the chain leaves it, and over how many elements, from EVERY position (pointer doubling across
    // the lanes: an element takes two bytes or more, so five rounds cover a window).  top: offset from the window's start
    // (>= 64), or the stop flag of the element the chain ends at; cnt: records of the elements on the way ----
    uint32_t top[LZP_WIN], cnt[LZP_WIN];
#pragma unroll
    for (int i = 0; i < LZP_WIN; i++) {
      top[i] = (hop[i] & LZP_STOP) ? hop[i] : lane + hop[i];
      cnt[i] = (hop[i] & LZP_STOP) ? 0u : (KIND == 2 || rc[i] == 0 ? 1u : 2u);  // records: LZ4 sequences give two, but for the last one
    }
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
      for (int i = 0; i < LZP_WIN; i++) {
        const int idx = (int)((top[i] & 63u) << 2);
        const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)top[i]);
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)cnt[i]);
        const bool in = top[i] < 64u;
        cnt[i] += in ? c : 0u;
        top[i] = in ? t : top[i];
      }
    }
    // ... and through the whole block for the entries in its first window (where the chain comes in unless an element longer
    // than 64 bytes straddles the border): one lookup on the critical path instead of one per window
    uint32_t bx = top[0], bc = cnt[0];
#pragma unroll
    for (int i = 1; i < LZP_WIN; i++) {
      const bool in = bx - 64u * i < 64u;  // (a stop flag is >= 2^31)
      const int idx = (int)(((bx - 64u * i) & 63u) << 2);
      const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)top[i]);
      const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)cnt[i]);
      bc += in ? c : 0u;
      bx = in ? ((t & LZP_STOP) ? t : 64u * i + t) : bx;
    }
    PROF_MARK(0);
    // ---- where does the chain come in? ----
    uint32_t e = bstart, base = 0;
    if (b) {
      const uint32_t slot = (b - 1) % LZP_RING;
      while (__hip_atomic_load(&L.tag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != b) __builtin_amdgcn_s_sleep(LZP_SLEEP);
      e = L.pos[slot];
      base = L.cnt[slot];
    }
    e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    PROF_MARK(1);
    // ---- through the block, one step per window ----
    uint32_t out_pos = e, stop = 0, nrec = 0;
    if (e != LZP_END_POS) {
      uint32_t cur = e - bstart;  // >= LZP_BLK: the chain jumps over this block
      if (cur < 64u) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)bx, (int)cur);
        nrec = (uint32_t)__builtin_amdgcn_readlane((int)bc, (int)cur);
        if (x & LZP_STOP) stop = x & 3;
        else cur = x;
      } else {
#pragma unroll
        for (int i = 1; i < LZP_WIN; i++) {
          if (!stop && cur < 64u * (i + 1)) {
            const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)top[i], (int)(cur - 64u * i));
            nrec += (uint32_t)__builtin_amdgcn_readlane((int)cnt[i], (int)(cur - 64u * i));
            if (x & LZP_STOP) stop = x & 3;
            else cur = 64u * i + x;
          }
        }
      }
      out_pos = stop ? LZP_END_POS : bstart + cur;
    }
    PROF_MARK(2);
    // ---- hand on ----
    {
      const uint32_t slot = b % LZP_RING;
      if (lane == 0) {
        L.pos[slot] = out_pos;
        L.cnt[slot] = base + nrec;
        __hip_atomic_store(&L.tag[slot], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (stop) {
      // the chain ended in this block: at the end of the input (complete) or at an element that cannot be
      if (lane == 0) {
        if (stop == 2) {
          cd->n_items = base + nrec;
          cd->pad = KIND == 2 ? (uint32_t)ulen : 0xffffffffu;
        } else {
          cd->status = ORC_E_CODEC;
          cd->diag = 21 + stop;
          cd->n_items = 0;
        }
      }
    }
    // ---- which positions the chain visits (off the critical path: the blocks behind are already on their way).  A window
    // with few elements is walked (one cross-lane read per element); a dense one is marked by all lanes at once: with the
    // jump tables J_k (2^k elements ahead) and D (elements up to the window's end), position p is on the chain from e
    // iff it leaves the window where e does and J^(D(e) - D(p))(e) = p ----
    unsigned long long mem[LZP_WIN];
#pragma unroll
    for (int i = 0; i < LZP_WIN; i++) mem[i] = 0;
    if (e != LZP_END_POS) {
      uint32_t cur = e - bstart;
      bool ended = false;
#pragma unroll
      for (int i = 0; i < LZP_WIN; i++) {
        unsigned long long m = 0;
        if (!ended && cur < 64u * (i + 1)) {
          const uint32_t ew = cur - 64u * i;
          const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)top[i], (int)ew);
          const uint32_t nr = (uint32_t)__builtin_amdgcn_readlane((int)cnt[i], (int)ew);
          if (nr > (KIND == 2 ? 8u : 16u)) {
            uint32_t J[5], D = (hop[i] & LZP_STOP) ? 0u : 1u;
            uint32_t nx = (hop[i] & LZP_STOP) ? hop[i] : lane + hop[i];
#pragma unroll
            for (int r = 0; r < 5; r++) {
              J[r] = nx;
              const int idx = (int)((nx & 63u) << 2);
              const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)nx);
              const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)D);
              const bool in = nx < 64u;
              D += in ? c : 0u;
              nx = in ? t : nx;
            }
            const uint32_t De = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)ew);
            const uint32_t hops = De - D;
            uint32_t y = ew;
#pragma unroll
            for (int r = 0; r < 5; r++) {
              const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((y & 63u) << 2), (int)J[r]);
              y = ((hops >> r) & 1u) && y < 64u ? t : y;
            }
            const bool on = nx == x && D <= De && lane >= ew && y == lane && !(hop[i] & LZP_STOP);
            m = __ballot(on);
          } else {
            uint32_t c = ew, h = 0;
            while (c < 64) {
              h = (uint32_t)__builtin_amdgcn_readlane((int)hop[i], (int)c);
              if (h & LZP_STOP) break;
              m |= 1ull << c;
              c += h;
            }
          }
          if (x & LZP_STOP) ended = true;
          else cur = 64u * i + x;
        }
        mem[i] = m;
      }
    }
    PROF_MARK(3);
    // ---- the tokens of this block ----
#pragma unroll
    for (int i = 0; i < LZP_WIN; i++) {
      const unsigned long long m = mem[i];
      if (m) {
        if ((m >> lane) & 1) {
          const uint32_t k = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
          if (KIND == 2) {
            recs[base + k] = make_uint2(ra[i], rb[i]);
          } else {
            recs[base + 2 * k] = make_uint2(ra[i], rb[i]);
            recs[base + 2 * k + 1] = make_uint2(rc[i], rd[i]);  // (zeros behind the last sequence: lz_exec reads pairs)
          }
        }
        base += (KIND == 2 ? 1u : 2u) * (uint32_t)__builtin_popcountll(m);
      }
    }
    PROF_MARK(4);
  }
  PROF_END_AT(112);
}

extern "C" __global__ void __launch_bounds__(64 * LZP_WAVES) lz_parse_kernel(ChunkDesc* chunks, uint32_t n_chunks) {
  __shared__ LzpLds L;
  const uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  const uint32_t tid = threadIdx.x;
  const ChunkDesc d = chunks[c];
  if (d.kind != 2 && d.kind != 4) return;
  if (tid < LZP_RING) L.tag[tid] = 0;
  __syncthreads();
  if (d.kind == 2) lz_parse_chunk<2>(L, chunks + c, d, tid);
  else lz_parse_chunk<4>(L, chunks + c, d, tid);
}
