// inflate_parse.h -- token stage of the DEFLATE decompressor: ONE WAVEFRONT PER ORC CHUNK, its 64 lanes decoding
// 64 stretches of the Huffman stream at once.
//
// Replaces the Huffman half of flate2's raw DeflateDecoder (compression.rs:142-149; RFC 1951).  A DEFLATE block is one
// prefix-coded stream of tokens (literal | length + distance | end of block); where a token starts is known only by decoding
// everything before it.  But prefix codes self-synchronise: a decoder started at a wrong bit falls in step with the true token
// boundaries after a few tokens.  So the stream is taken a window at a time (64 segments of INF_SEG bits): every lane decodes
// its segment from a guessed start up to the first token boundary behind the segment's end; then each lane compares its start
// with the end its left neighbour reached and decodes again from there if they differ, until nothing moves (lane 0 starts at
// the true position, so what comes out is exactly the serial decode); a counting pass and a writing pass follow.  Block headers
// (code length tables) are decoded wave-uniformly by the code of inflate_device.h.
//
// Output per chunk: the literal bytes, back to back, and one {distance + 3, match length, literal length} record per match (the
// format of zstd_entropy.h: lz_exec.h executes it).  Anything unexpected -- a malformed stream, a code the fast path does not
// take -- leaves the chunk to the one-wavefront decoder (decompress_deflate_kernel), which is the authority on errors.
#pragma once

#ifndef INF_SEG
#define INF_SEG 256u                      // bits per thread and window
#endif
// threads per chunk (template parameter T below): 64 -- a wavefront per chunk, for calls with chunks enough to fill the device --, or 256:
// four wavefronts take four times the window at once (a call of few chunks lasts as long as its longest chunk's chain of windows)
#define INF_WIN_BYTES(T) ((T) * INF_SEG / 8u)  // 2 / 8 KiB
#define INF_SLACK 16u                     // a token reads at most 48 bits past its first one; a thread stops within them

// codes longer than the fast table's 10 bits: per length 11..15 the first canonical code, how many there are, and where their symbols
// start in HuffTab::symbol (filled per block by inf_long_codes)
struct InfLong {
  uint16_t first[5], count[5], index[5], pad;
};
template <int T>
struct InfLds {
  HuffTab lit, dist;
  uint8_t lens[320];
  __attribute__((aligned(16))) uint8_t win[INF_WIN_BYTES(T) + INF_SLACK + 24];
  uint32_t end_pos[T + 1];   // [i + 1]: where thread i's decode ended (bits from the window's first bit); [0]: the window's true start
  uint32_t n_lit[T], n_seq[T], tail_lit[T], carry_in[T];
  uint32_t flagv[T];         // what stopped a thread (1 end of block, 2 cannot decode)
  uint32_t wsum[2][T / 64];  // the wavefronts' totals of a scan (literals, matches)
  uint32_t stop_at;              // the first thread (in stream order) that met the end of the block or could not decode
  uint32_t any;                  // a round's "somebody moved"
  uint32_t hdr[8];               // what wavefront 0 found in a block header: {code, last, bit position lo / hi, literals so far, pending literals}
  uint16_t lbase[32], dbase[32];  // RFC 1951 3.2.5 (copies of LBASE / DBASE / LEXT / DEXT: an LDS read, not a trip to memory)
  uint8_t lext[32], dext[32];
  InfLong llong, dlong;
};
__device__ __forceinline__ void inf_long_codes(InfLong& g, const HuffTab& h, uint32_t lane) {
  if (lane == 0) {
    uint32_t code = 0, index = 0;
    for (int len = 1; len < 16; len++) {
      // (canonical codes: the first code of a length = (first of the length before + its count) << 1)
      if (len > HUFF_FAST_BITS) {
        g.first[len - HUFF_FAST_BITS - 1] = (uint16_t)code;
        g.count[len - HUFF_FAST_BITS - 1] = h.count[len];
        g.index[len - HUFF_FAST_BITS - 1] = (uint16_t)index;
      }
      code = (code + h.count[len]) << 1;
      index += h.count[len];
    }
  }
}

struct InfTok {
  uint32_t kind;   // 0 literal, 1 match, 2 end of block, 3 cannot be (the serial decoder decides what it is)
  uint32_t bits;   // bits the token takes
  uint32_t a, b;   // literal: byte; match: length, distance
};

// the code at the low end of `bits` when the fast table has no entry for it (a code of 11..15 bits): -1 = no such code
__device__ __forceinline__ int inf_slow(uint64_t bits, const HuffTab& h, const InfLong& g, uint32_t& used) {
  const uint32_t r = __builtin_bitreverse32((uint32_t)bits) >> 17;  // the first 15 bits, the first one on top
#pragma unroll
  for (int k = 0; k < 15 - HUFF_FAST_BITS; k++) {
    const uint32_t c = r >> (14 - HUFF_FAST_BITS - k);  // the first HUFF_FAST_BITS + 1 + k bits as a code
    const uint32_t d = c - g.first[k];
    if (d < g.count[k]) {
      used = HUFF_FAST_BITS + 1u + (uint32_t)k;
      return h.symbol[g.index[k] + d];
    }
  }
  return -1;
}

// the token at bit `pos` of the staged window (per lane: no cooperation)
template <int T>
__device__ __forceinline__ InfTok inf_token(const InfLds<T>& L, const uint8_t* win, uint32_t pos, const HuffTab& lc, const HuffTab& dc) {
  // 57 bits or more from bit `pos` on: two aligned 8-byte reads and a funnel shift (an unaligned 8-byte LDS read is eight byte reads)
  const uint64_t* w64 = reinterpret_cast<const uint64_t*>(win) + (pos >> 6);
  const uint64_t w0 = w64[0], w1 = w64[1];
  const uint32_t sh = pos & 63;
  uint64_t v = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
  InfTok t{3, 0, 0, 0};
  uint32_t used;
  int sym;
  uint32_t e = lc.fast[(uint32_t)v & (HUFF_FAST_SIZE - 1)];
  if (e) {
    used = e >> 12;
    sym = (int)(e & 0xfff);
  } else {
    sym = inf_slow(v, lc, L.llong, used);
    if (sym < 0) return t;
  }
  if (sym < 256) {
    t.kind = 0;
    t.bits = used;
    t.a = (uint32_t)sym;
    return t;
  }
  if (sym == 256) {
    t.kind = 2;
    t.bits = used;
    return t;
  }
  sym -= 257;
  if (sym >= 29) return t;
  const uint32_t le = L.lext[sym];
  const uint32_t len = L.lbase[sym] + ((uint32_t)(v >> used) & ((1u << le) - 1));
  used += le;
  uint32_t du;
  int ds;
  e = dc.fast[(uint32_t)(v >> used) & (HUFF_FAST_SIZE - 1)];
  if (e) {
    du = e >> 12;
    ds = (int)(e & 0xfff);
  } else {
    ds = inf_slow(v >> used, dc, L.dlong, du);
    if (ds < 0) return t;
  }
  if (ds >= 30) return t;
  used += du;
  const uint32_t de = L.dext[ds];
  const uint32_t dist = L.dbase[ds] + ((uint32_t)(v >> used) & ((1u << de) - 1));
  used += de;
  t.kind = 1;
  t.bits = used;  // <= 15 + 5 + 15 + 13 = 48
  t.a = len;
  t.b = dist;
  return t;
}

// One window of a Huffman block, by all T threads of the workgroup (thread t: segment t of the window).  `bitpos`: the
// window's first bit (absolute, in the chunk); tokens are decoded while they START before the end of the input.  Returns 0 = window
// done (bitpos advanced), 1 = end of block met (bitpos behind it), 2 = leave the chunk to the serial decoder.  lit / seq / pending:
// the chunk's output so far (uniform over the workgroup: every thread computes them from the same LDS words).
template <int T>
__device__ __forceinline__ void inf_sync() {
  if (T == 64) wave_sync();
  else __syncthreads();
}
template <int T>
__device__ __forceinline__ int inf_window(InfLds<T>& L, const uint8_t* src, uint32_t n, uint64_t& bitpos, uint8_t* lit_out, uint32_t* seq_out, uint32_t& litn,
                                          uint32_t& nseq, uint32_t& pending_ll, uint32_t lit_cap, uint32_t seq_cap, uint32_t tid PROF_PARM) {
  PROF_MARK(0);
  PROF_COUNT(4, 1);
  const uint32_t lane = tid & 63, wv = tid >> 6;
  // ---- stage the window (byte aligned at its first bit's byte) ----
  const uint32_t byte0 = (uint32_t)(bitpos >> 3), bit0 = (uint32_t)(bitpos & 7);
  for (uint32_t k = tid * 8; k < INF_WIN_BYTES(T) + INF_SLACK + 16; k += T * 8) {
    uint64_t v = 0;
    const uint64_t p = (uint64_t)byte0 + k;
    if (p + 8 <= n) v = ld_u64(src + p);
    else
      for (uint32_t t = 0; t < 8; t++)
        if (p + t < n) v |= (uint64_t)src[p + t] << (8 * t);
    __builtin_memcpy(L.win + k, &v, 8);
  }
  if (tid == 0) L.stop_at = T;
  const uint64_t total_bits = (uint64_t)n * 8;
  const uint64_t left = total_bits - bitpos;  // bits of input from the window's first bit
  const uint32_t seg_end = bit0 + (tid + 1) * INF_SEG;  // (positions are relative to the window's first BYTE)
  inf_sync<T>();
  // decode from `start` while tokens begin before this thread's segment end; what = 0 count only, 1 write
  uint32_t my_start = tid == 0 ? bit0 : bit0 + tid * INF_SEG;
  uint32_t my_end = 0, my_flag = 0;  // flag: 1 end of block inside, 2 cannot decode
  uint32_t c_lit = 0, c_seq = 0, c_tail = 0;
  auto run = [&](uint32_t start, bool write, uint32_t lit_base, uint32_t seq_base, uint32_t carry) {
    uint32_t pos = start, flag = 0;
    uint32_t nl = 0, ns = 0, run_ll = carry;
#pragma unroll 1
    while (pos < seg_end) {
      if ((uint64_t)pos - bit0 >= left) {
        flag = 2;  // ran off the end of the input without an end of block
        break;
      }
      const InfTok t = inf_token(L, L.win, pos, L.lit, L.dist);
      if (t.kind == 3 || (uint64_t)pos - bit0 + t.bits > left) {
        flag = 2;
        break;
      }
      pos += t.bits;
      if (t.kind == 2) {
        flag = 1;
        break;
      }
      if (t.kind == 0) {
        if (write) lit_out[lit_base + nl] = (uint8_t)t.a;
        nl++;
        run_ll++;
      } else {
        if (write) {
          uint32_t* q = seq_out + 3ull * (seq_base + ns);
          q[0] = t.b + 3;  // an offset value above 3: a new offset (RFC 8878 3.1.1.5), as lz_exec reads it
          q[1] = t.a;
          q[2] = run_ll;
        }
        ns++;
        run_ll = 0;
      }
    }
    my_end = pos;
    my_flag = flag;
    c_lit = nl;
    c_seq = ns;
    c_tail = run_ll - (ns ? 0u : carry);  // literals behind the thread's last match (all of its own when it has none)
  };
  PROF_MARK(1);
  // ---- every thread from its guess, then again from where its left neighbour ended, until nothing moves ----
  // (Measured and not kept: decodes that stop at a token boundary an earlier decode of the thread stood on -- a bitmap per thread in
  // LDS -- with the counting left to a pass of its own: the marks cost what the shorter re-runs save, lineitem / zlib 30.3 -> 34.0 ms.)
  run(my_start, false, 0, 0, 0);
  PROF_MARK(2);
  PROF_COUNT(6, c_lit + c_seq);
  for (uint32_t round = 0; round < T; round++) {
    L.end_pos[tid + 1] = my_end;
    if (tid == 0) {
      L.end_pos[0] = bit0;
      L.any = 0;
    }
    inf_sync<T>();
    const uint32_t want = L.end_pos[tid];
    // a thread behind an end of block / a failure has nothing of its own: it passes the position on
    const bool moved = want != my_start;
    if (moved) {
      my_start = want;
      if (want >= seg_end) {
        my_end = want;
        my_flag = 0;
        c_lit = c_seq = c_tail = 0;
      } else {
        run(want, false, 0, 0, 0);
      }
      L.any = 1;
    }
    inf_sync<T>();
    PROF_COUNT(5, 1);
    const bool any = L.any != 0;
    inf_sync<T>();  // (everybody has read the word before the next round clears it)
    if (!any) break;
    if (round == T - 1) return 2;
  }
  PROF_MARK(3);
  // ---- the first thread (in stream order) that met the end of the block or could not decode ends the window ----
  L.flagv[tid] = my_flag;
  if (my_flag) atomicMin(&L.stop_at, tid);
  inf_sync<T>();
  uint32_t last_t = T - 1;
  int result = 0;
  if (L.stop_at < T) {
    last_t = L.stop_at;
    if (L.flagv[last_t] == 2) return 2;
    result = 1;
  }
  const bool mine = tid <= last_t;
  // ---- positions: exclusive prefix sums of literals and matches; literals pending across threads ----
  const uint32_t nl = mine ? c_lit : 0, ns = mine ? c_seq : 0;
  uint32_t il = wave_incl_scan_u32(nl, lane), is = wave_incl_scan_u32(ns, lane);
  if (lane == 63) {
    L.wsum[0][wv] = il;
    L.wsum[1][wv] = is;
  }
  L.n_seq[tid] = ns;
  L.tail_lit[tid] = mine ? c_tail : 0;
  L.n_lit[tid] = nl;
  inf_sync<T>();
  uint32_t tot_l = 0, tot_s = 0;
#pragma unroll
  for (uint32_t w = 0; w < T / 64; w++) {
    if (w < wv) {
      il += L.wsum[0][w];
      is += L.wsum[1][w];
    }
    tot_l += L.wsum[0][w];
    tot_s += L.wsum[1][w];
  }
  if ((uint64_t)litn + tot_l > lit_cap || (uint64_t)nseq + tot_s > seq_cap) return 2;
  if (tid == 0) {
    uint32_t carry = pending_ll;
    for (uint32_t i = 0; i < T; i++) {
      L.carry_in[i] = carry;
      carry = L.n_seq[i] ? L.tail_lit[i] : carry + L.n_lit[i];
    }
    L.end_pos[0] = carry;  // (reused: literals pending behind the window)
  }
  inf_sync<T>();
  const uint32_t carry = L.carry_in[tid];
  const uint32_t new_pending = L.end_pos[0];
  const uint32_t end_rel = L.end_pos[last_t + 1];
  PROF_MARK(4);
  // ---- write ----
  if (mine && (nl || ns)) run(my_start, true, litn + il - nl, nseq + is - ns, carry);
  PROF_MARK(5);
  litn += tot_l;
  nseq += tot_s;
  pending_ll = new_pending;
  bitpos = (uint64_t)byte0 * 8 + end_rel;
  inf_sync<T>();
  return result;
}

// A block's header by ONE wavefront (the code of inflate_device.h: wave-uniform): the block's two code tables in L.lit / L.dist, or a
// stored block copied to the literals.  Returns 0 = a Huffman block follows at the reader's position, 1 = a stored block (handled),
// 2 = leave the chunk to the serial decoder.
template <int T>
__device__ __forceinline__ int inf_block_header(InfLds<T>& L, BitRd& b, const uint8_t* src, uint32_t n, uint8_t* lit_out, uint32_t lit_cap, uint32_t& litn,
                                                uint32_t& pending, uint32_t& last, uint32_t lane) {
  last = br_get(b, 1);
  const uint32_t type = br_get(b, 2);
  if (br_overrun(b)) return 2;
  if (type == 0) {
    const uint32_t drop = b.bc & 7;
    b.bb >>= drop;
    b.bc -= drop;
    uint32_t bytepos = b.pos - (b.bc >> 3);
    if (bytepos + 4 > n) return 2;
    const uint32_t len = src[bytepos] | (src[bytepos + 1] << 8);
    const uint32_t nlen = src[bytepos + 2] | (src[bytepos + 3] << 8);
    bytepos += 4;
    if ((len ^ 0xffffu) != nlen) return 2;
    if (bytepos + len > n || (uint64_t)litn + len > lit_cap) return 2;
    wave_copy(lit_out + litn, src + bytepos, len, lane);
    litn += len;
    pending += len;
    b.pos = bytepos + len;
    b.bb = 0;
    b.bc = 0;
    return 1;
  }
  if (type == 1) {
    for (uint32_t i = lane; i < 288; i += 64) L.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
    wave_sync();
    huff_build_dev(L.lit, L.lens, 288, lane);
    for (uint32_t i = lane; i < 30; i += 64) L.lens[i] = 5;
    wave_sync();
    huff_build_dev(L.dist, L.lens, 30, lane);
  } else if (type == 2) {
    const uint32_t nlen = br_get(b, 5) + 257, ndist = br_get(b, 5) + 1, ncode = br_get(b, 4) + 4;
    if (br_overrun(b) || nlen > 286 || ndist > 30) return 2;
    for (uint32_t i = lane; i < 19; i += 64) L.lens[i] = 0;
    wave_sync();
    for (uint32_t i = 0; i < ncode; i++) {
      const uint32_t v = br_get(b, 3);
      if (lane == 0) L.lens[CLORDER[i]] = (uint8_t)v;
    }
    wave_sync();
    if (huff_build_dev(L.lit, L.lens, 19, lane) != 0) return 2;
    uint32_t i = 0, prev = 0;
    while (i < nlen + ndist) {
      const int sym = huff_decode_dev(b, L.lit);
      if (sym < 0) return 2;
      if (sym < 16) {
        if (lane == 0) L.lens[i] = (uint8_t)sym;
        prev = (uint32_t)sym;
        i++;
      } else {
        uint32_t len = 0, rep;
        if (sym == 16) {
          if (i == 0) return 2;
          len = prev;
          rep = 3 + br_get(b, 2);
        } else if (sym == 17) {
          rep = 3 + br_get(b, 3);
        } else {
          rep = 11 + br_get(b, 7);
        }
        if (i + rep > nlen + ndist) return 2;
        for (uint32_t k = lane; k < rep; k += 64) L.lens[i + k] = (uint8_t)len;
        prev = len;
        i += rep;
      }
    }
    if (br_overrun(b)) return 2;
    wave_sync();
    if (L.lens[256] == 0) return 2;
    int r = huff_build_dev(L.dist, L.lens + nlen, (int)ndist, lane);
    if (r < 0 || (r > 0 && (int)ndist - (int)L.dist.count[0] != 1)) return 2;
    r = huff_build_dev(L.lit, L.lens, (int)nlen, lane);
    if (r < 0 || (r > 0 && (int)nlen - (int)L.lit.count[0] != 1)) return 2;
  } else {
    return 2;
  }
  inf_long_codes(L.llong, L.lit, lane);
  inf_long_codes(L.dlong, L.dist, lane);
  wave_sync();
  return 0;
}

// The chunk's blocks: wavefront 0 reads a block's header and builds its tables, all wavefronts take the block's tokens a window at
// a time.  Returns 0 (litn, nseq final; trailing literals are the sequence-less rest) or 2 (serial decoder).
template <int T>
__device__ __forceinline__ int inflate_parse_chunk(InfLds<T>& L, LzLds Z, const uint8_t* src, uint32_t n, uint8_t* lit_out, uint32_t* seq_out, uint32_t lit_cap,
                                                   uint32_t seq_cap, uint32_t& litn_out, uint32_t& nseq_out, uint32_t tid PROF_PARM) {
  const uint32_t lane = tid & 63, wv = tid >> 6;
  LzIn in{src, n, Z.stage, 0};
  BitRd b{src, n, 0, 0, 0, &in};
  if (wv == 0) lzin_stage(in, 0, lane);
  if (tid < 32) {
    L.lbase[tid] = tid < 29 ? LBASE[tid] : 0;
    L.lext[tid] = tid < 29 ? LEXT[tid] : 0;
    L.dbase[tid] = tid < 30 ? DBASE[tid] : 0;
    L.dext[tid] = tid < 30 ? DEXT[tid] : 0;
  }
  inf_sync<T>();
  uint32_t litn = 0, nseq = 0, pending = 0;
  uint32_t last = 0;
  do {
    if (wv == 0) {
      const int code = inf_block_header(L, b, src, n, lit_out, lit_cap, litn, pending, last, lane);
      if (lane == 0) {
        const uint64_t bp = (uint64_t)b.pos * 8 - b.bc;
        L.hdr[0] = (uint32_t)code;
        L.hdr[1] = last;
        L.hdr[2] = (uint32_t)bp;
        L.hdr[3] = (uint32_t)(bp >> 32);
        L.hdr[4] = litn;
        L.hdr[5] = pending;
      }
    }
    PROF_MARK(6);
    PROF_COUNT(7, 1);
    inf_sync<T>();
    const uint32_t code = L.hdr[0];
    last = L.hdr[1];
    uint64_t bitpos = (uint64_t)L.hdr[2] | ((uint64_t)L.hdr[3] << 32);
    litn = L.hdr[4];
    pending = L.hdr[5];
    inf_sync<T>();  // (the words are read: wavefront 0 may write the next header's)
    if (code == 2) return 2;
    if (code == 1) continue;  // a stored block: copied
    // ---- the block's tokens, a window at a time ----
    for (;;) {
      const int r = inf_window(L, src, n, bitpos, lit_out, seq_out, litn, nseq, pending, lit_cap, seq_cap, tid PROF_ARG);
      if (r == 2) return 2;
      if (r == 1) break;
    }
    if (bitpos > (uint64_t)n * 8) return 2;
    // the bit reader (wavefront 0's) goes on behind the block
    if (wv == 0) {
      b.pos = (uint32_t)(bitpos >> 3);
      b.bb = 0;
      b.bc = 0;
    }
    if (bitpos & 7) {
      if ((uint32_t)(bitpos >> 3) >= n) return 2;
      if (wv == 0) (void)br_get(b, (uint32_t)(bitpos & 7));  // (refills from b.pos: the bits of that byte below the position are dropped)
    }
  } while (!last);
  litn_out = litn;
  nseq_out = nseq;
  return 0;
}

// ChunkDesc of a DEFLATE chunk: scratch = [literal bytes: dst_cap + 16][records: 12 x (dst_cap / 3 + 2)]; the kernel leaves
// n_items = records, pad = literal bytes, diag = 0 -- or diag = LZX_DEFERRED: not decoded here, decompress_deflate_kernel takes it.
#ifndef INF_MIN_WAVES
#define INF_MIN_WAVES 3  // 168 registers, nothing spilled: three wavefronts per SIMD instead of two (lineitem / zlib SF 4: token stage 23.1 -> 20.3 ms)
#endif
// [min_src, max_src): the chunks (by compressed size) this launch takes -- a call of many chunks gives its long ones to the
// four-wavefront kernel and the rest to the one-wavefront kernel
template <int T>
__device__ __forceinline__ void inflate_parse_body(ChunkDesc* chunks, uint32_t n_chunks, uint32_t min_src, uint32_t max_src) {
  __shared__ InfLds<T> L;
  __shared__ __attribute__((aligned(16))) uint8_t stage[LZ_STAGE + 16];
  const uint32_t c = blockIdx.x;
  if (c >= n_chunks) return;
  const ChunkDesc d = chunks[c];
  if (d.kind != 1 || d.src_len < min_src || d.src_len >= max_src) return;
  const uint32_t tid = threadIdx.x;
  const uint8_t* src = as_global(d.src);
  uint8_t* sc = (uint8_t*)as_global((void*)d.scratch);
  const uint32_t lit_cap = d.dst_cap;
  const uint32_t seq_cap = d.dst_cap / 3 + 2;
  uint8_t* lit_out = sc;
  uint32_t* seq_out = reinterpret_cast<uint32_t*>(sc + ((lit_cap + 16 + 15) & ~15u));
  uint32_t litn = 0, nseq = 0;
  LzLds Z{nullptr, 0, stage};
  PROF_BEGIN();
  const int r = sc ? inflate_parse_chunk(L, Z, src, d.src_len, lit_out, seq_out, lit_cap, seq_cap, litn, nseq, tid PROF_ARG) : 2;
  PROF_MARK(7);
  PROF_END_AT(48);
  if (tid == 0) {
    chunks[c].n_items = r ? 0 : nseq;
    chunks[c].pad = r ? 0 : litn;
    chunks[c].diag = r ? LZX_DEFERRED : 0;
  }
}
extern "C" __global__ void __launch_bounds__(64, INF_MIN_WAVES) inflate_parse_kernel(ChunkDesc* chunks, uint32_t n_chunks, uint32_t min_src, uint32_t max_src) { inflate_parse_body<64>(chunks, n_chunks, min_src, max_src); }
extern "C" __global__ void __launch_bounds__(256, INF_MIN_WAVES) inflate_parse4_kernel(ChunkDesc* chunks, uint32_t n_chunks, uint32_t min_src, uint32_t max_src) { inflate_parse_body<256>(chunks, n_chunks, min_src, max_src); }
