// zstd_device.h -- building blocks of the Zstandard decoder (RFC 8878; replaces zstd::Decoder at
// compression.rs:151-159; no dictionaries; the content checksum is skipped, not verified): constant
// tables, the backward bit reader, FSE table descriptions and tables, Huffman tables and the
// many-lanes-per-stream Huffman decoder.  The decoder itself is split in two kernels:
// zstd_entropy.h (one wavefront per compressed block) and lz_exec.h (one workgroup per chunk).
#pragma once

__device__ const int16_t Z_LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ const int16_t Z_ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ const int16_t Z_OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__device__ const uint32_t Z_LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ const uint8_t Z_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint32_t Z_ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ const uint8_t Z_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

// backward bit stream: `bits` = number of unread bits, reading proceeds from bit (bits-1) down.
// A 64-bit window of the stream is kept in a register and refilled only when a read leaves it: a
// memory round trip per 7 bytes consumed instead of one per symbol.
struct RBits {
  const uint8_t* p;
  long bits;
  uint64_t win;   // stream bits [wbit, wbit + 64)
  long wbit;      // multiple of 8; -1: nothing loaded
};
__device__ __forceinline__ bool rb_init(RBits& r, const uint8_t* p, uint32_t n) {
  r.p = p;
  r.bits = 0;
  r.win = 0;
  r.wbit = -1;
  if (n == 0 || p[n - 1] == 0) return false;
  int hb = 31 - __builtin_clz((uint32_t)p[n - 1]);
  r.bits = (long)(n - 1) * 8 + hb;
  return true;
}
__device__ __forceinline__ uint64_t rb_read(RBits& r, uint32_t nb) {
  if (nb == 0) return 0;
  long start = r.bits - (long)nb;
  uint64_t v;
  uint64_t mask = nb >= 64 ? ~0ull : ((1ull << nb) - 1);
  if (start >= 0) {
    if (nb > 56) {
      v = (ld_u64(r.p + (start >> 3)) >> (start & 7)) & mask;  // (never: fields are at most 32 bits wide)
    } else {
      if (r.wbit < 0 || start < r.wbit || start + (long)nb > r.wbit + 64) {
        // put the window's top just above this read: the following (lower) reads find their bits in it
        long byte = ((start + (long)nb + 7) >> 3) - 8;
        if (byte < 0) byte = 0;
        r.win = ld_u64(r.p + byte);
        r.wbit = byte * 8;
      }
      v = (r.win >> (start - r.wbit)) & mask;
    }
  } else if (r.bits > 0) {
    // the low (-start) bits lie before the stream and read as zero
    uint64_t have = ld_u64(r.p) & ((1ull << r.bits) - 1);
    v = (have << (-start)) & mask;
  } else {
    v = 0;
  }
  r.bits = start;
  return v;
}

__device__ __forceinline__ int z_hibit(uint32_t v) { return 31 - __builtin_clz(v); }

// FSE decoding table from normalised counts (all lanes run this redundantly on the same LDS: benign)
__device__ __forceinline__ int fse_build_dev(FseEnt* t, const int16_t* norm, int nsym, int log, uint16_t* next, uint32_t lane) {
  int size = 1 << log;
  int bad = 0;
  if (lane == 0) {
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
      if (norm[s] == -1) {
        t[high--].sym = (uint8_t)s;
        next[s] = 1;
      } else {
        next[s] = (uint16_t)norm[s];
      }
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++) {
      for (int i = 0; i < norm[s]; i++) {
        t[pos].sym = (uint8_t)s;
        do {
          pos = (pos + step) & mask;
        } while (pos > high);
      }
    }
    if (pos != 0) bad = 1;
    for (int i = 0; i < size && !bad; i++) {
      int s = t[i].sym;
      uint32_t ns = next[s]++;
      int nb = log - z_hibit(ns);
      t[i].nb = (uint8_t)nb;
      t[i].base = (uint16_t)((ns << nb) - size);
    }
  }
  bad = __shfl(bad, 0);
  wave_sync();
  return bad;
}

// FSE table description (forward bit stream); returns bytes consumed or -1.  Uniform.
// `stg`: 128 bytes of LDS the description is read through (one load of the wavefront instead of a dependent memory access per byte).
__device__ __forceinline__ long fse_read_ncount_dev(const uint8_t* p, uint32_t n, int16_t* norm, int* nsym_io, int* log_out, int maxlog,
                                                    uint32_t lane, uint8_t* stg) {
  const uint32_t staged = n < 128u ? n : 128u;
  if (lane < 32 && 4 * lane < staged) reinterpret_cast<uint32_t*>(stg)[lane] = ld_u32(p + 4 * lane);  // (up to 3 bytes behind the description: slack of the arena)
  wave_sync();
  uint32_t pos = 0;
  uint64_t bb = 0;
  int bc = 0;
#define ZNEED(k)                                  \
  while (bc < (k)) {                              \
    uint64_t byte_ = pos < n ? (pos < staged ? stg[pos] : p[pos]) : 0; \
    if (pos >= n + 8) return -1;                  \
    pos++;                                        \
    bb |= byte_ << bc;                            \
    bc += 8;                                      \
  }
  ZNEED(4);
  int log = (int)(bb & 15) + 5;
  bb >>= 4;
  bc -= 4;
  if (log > maxlog) return -1;
  int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1;
  int sym = 0, maxsym = *nsym_io;
  int prev0 = 0;
  while (remaining > 1 && sym < maxsym) {
    if (prev0) {
      for (;;) {
        ZNEED(2);
        int rep = (int)(bb & 3);
        bb >>= 2;
        bc -= 2;
        for (int i = 0; i < rep && sym < maxsym; i++) {
          if (lane == 0) norm[sym] = 0;
          sym++;
        }
        if (rep != 3) break;
      }
      prev0 = 0;
      if (sym >= maxsym) break;
      continue;
    }
    int max = (2 * threshold - 1) - remaining;
    ZNEED(nbits);
    int count;
    if ((int)(bb & (uint64_t)(threshold - 1)) < max) {
      count = (int)(bb & (uint64_t)(threshold - 1));
      bb >>= (nbits - 1);
      bc -= (nbits - 1);
    } else {
      count = (int)(bb & (uint64_t)(2 * threshold - 1));
      if (count >= threshold) count -= max;
      bb >>= nbits;
      bc -= nbits;
    }
    count--;
    remaining -= count < 0 ? -count : count;
    if (lane == 0) norm[sym] = (int16_t)count;
    sym++;
    prev0 = (count == 0);
    while (remaining < threshold) {
      nbits--;
      threshold >>= 1;
    }
  }
#undef ZNEED
  if (remaining != 1) return -1;
  *nsym_io = sym;
  *log_out = log;
  uint32_t bits_used = pos * 8 - (uint32_t)bc;
  uint32_t used = (bits_used + 7) / 8;
  if (used > n) return -1;
  wave_sync();
  return (long)used;
}

// Huffman decoding table from weights (last weight implied), by the whole wavefront.  (One lane used to do it all -- three passes
// over the weights with counters in a private array, i.e. in scratch memory, and 2048 two-byte stores: 150 us a block, a sixth of
// the literals kernel.)  Symbols s = lane, lane + 64, ...: the sum of 2^(w - 1) by a wave reduction; per weight the symbols' count
// and every symbol's rank among the symbols of its weight by ballots; `order` = the symbols sorted by (weight, symbol); the table
// weight class by weight class, 64 entries a step.  scratch: 256 + 16 bytes of LDS.
__device__ __forceinline__ int huf_build_dev(uint16_t* tab, int* maxbits_out, uint8_t* w, int nw, uint32_t lane, uint8_t* scratch) {
  uint8_t* order = scratch;
  // this lane's symbols (nw <= 255 of them stated; the last one, nw, implied)
  uint32_t ws[4];
  uint32_t sum = 0;
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int sidx = r * 64 + (int)lane;
    ws[r] = sidx < nw ? w[sidx] : 0;
    if (ws[r] > 11) bad = true;
    else if (ws[r]) sum += 1u << (ws[r] - 1);
  }
  for (int o = 32; o; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o);
  if (__ballot(bad) || sum == 0) return 1;
  const int maxbits = z_hibit(sum) + 1;
  if (maxbits > 11) return 1;
  const uint32_t left = (1u << maxbits) - sum;
  if (left & (left - 1)) return 1;
  const uint32_t wlast = (uint32_t)z_hibit(left) + 1;
  {
    const int r = nw >> 6;
    if ((int)lane == (nw & 63)) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k == r) ws[k] = wlast;
    }
  }
  // per weight: how many symbols (cnt), where their entries start (rs) and where they stand in `order` (cb); uniform
  uint32_t total = 0, cbase = 0;
  uint32_t rs_of[4] = {0, 0, 0, 0}, ord_of[4] = {0, 0, 0, 0};
  uint32_t rs_w[13], cb_w[13];
#pragma unroll
  for (int wt = 1; wt <= 12; wt++) {
    rs_w[wt] = total;
    cb_w[wt] = cbase;
    if (wt > maxbits) continue;
    uint32_t seen = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const unsigned long long m = __ballot(ws[r] == (uint32_t)wt);
      if (ws[r] == (uint32_t)wt) {
        const uint32_t rank = seen + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
        ord_of[r] = cbase + rank;
        rs_of[r] = total + (rank << (wt - 1));
      }
      seen += (uint32_t)__builtin_popcountll(m);
    }
    total += seen << (wt - 1);
    cbase += seen;
  }
  if (total != (1u << maxbits)) return 1;
  (void)rs_of;
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (ws[r]) order[ord_of[r]] = (uint8_t)(r * 64 + (int)lane);
  wave_sync();
  // the table: the entries of weight wt are runs of 2^(wt - 1) cells, one run per symbol of that weight in symbol order
#pragma unroll
  for (int wt = 1; wt <= 11; wt++) {
    if (wt > maxbits) continue;
    const uint32_t lo = rs_w[wt], hi = wt == 11 ? total : rs_w[wt + 1];
    const uint32_t nbv = (uint32_t)(maxbits + 1 - wt) << 8;
    for (uint32_t i = lo + lane; i < hi; i += 64) tab[i] = (uint16_t)(order[cb_w[wt] + ((i - lo) >> (wt - 1))] | nbv);
  }
  *maxbits_out = maxbits;
  wave_sync();
  return 0;
}

// Huffman streams decoded by MANY lanes each (16 per stream for the usual four streams, 64 for a single
// one).  A stream is a chain of prefix codes read downwards from its top bit; lane k of a stream starts
// at bit top - k*B (not a code boundary in general), decodes down to the start of the next lane's
// segment and reports where it crossed it.  Prefix codes re-synchronise within a few symbols, so after
// handing every lane its predecessor's crossing point once or twice nothing changes any more (the
// loop runs until then: exact whatever the data, at worst as slow as one lane per stream).  Symbol
// counts are then prefix-summed per stream and a last pass writes the symbols.
// downward bit cursor for the Huffman streams: `pos` = code boundary (bits of the stream below it are
// unread), `w` holds the bits just below pos left-aligned (bit pos-1 at bit 63), `avail` of them valid;
// bits below the start of the stream read as zero.
struct HBits {
  const uint8_t* p;
  uint64_t lo, hi1;  // stream bits [wb64 - 64, wb64) and [wb64, wb64 + 64) (the latter kept shifted left by one)
  uint64_t nx[4];    // the 32 bytes below lo, highest word first: on their way while lo and hi are consumed
  uint32_t nxi;      // words of nx taken so far
  int wb64;          // multiple of 8; 0 <= pos - wb64 <= 63 always
  int pos;
};
// 64 stream bits from bit `wb` (a multiple of 8; may lie before the stream: those bits read as zero) of the stream at q
__device__ __forceinline__ uint64_t zl_word(const uint8_t* q, int wb) {
  const int bo = wb >> 3;
  const uint64_t v = ld_u64(q + (bo < 0 ? 0 : bo));
  const int neg = bo < 0 ? -bo : 0;  // bytes of the word that lie before the stream
  return neg >= 8 ? 0ull : v << (8 * neg);
}
// the four words below bit `wb`: nx[k] = stream bits [wb - 64 (k + 1), wb - 64 k).  ONE 32-byte piece of the stream per 256 bits of
// codes: with a word at a time (8 bytes per load, 64 lanes 750 bytes apart, 26 wavefronts a CU) every load was a cache line from
// memory -- the lines do not survive in L2 between a lane's loads: 13.7 GB fetched for 1.1 GB of Huffman streams (PMC, SF 12.5)
__device__ __forceinline__ void hb_fetch4(HBits& h, int wb) {
  const int bo = (wb - 256) >> 3;
  if (bo >= 0) {
    uint64_t w[4];
    __builtin_memcpy(w, h.p + bo, 32);
    h.nx[3] = w[0];
    h.nx[2] = w[1];
    h.nx[1] = w[2];
    h.nx[0] = w[3];
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) h.nx[k] = zl_word(h.p, wb - 64 * (k + 1));
  }
  h.nxi = 0;
}
// A seek costs four loads; stepping down the stream afterwards never waits for memory: the words below the window are requested
// when the last of them is taken and needed only when the window moves again (eight bytes of codes later).  The loop used to
// reload its window from the cursor's byte position -- a dependent load every five or six symbols in SOME lane of the wavefront,
// i.e. in every iteration of all of them.
__device__ __forceinline__ void hb_seek(HBits& h, int pos) {
  h.pos = pos;
  h.wb64 = (pos & ~7) - 56;
  h.hi1 = zl_word(h.p, h.wb64) << 1;
  h.lo = zl_word(h.p, h.wb64 - 64);
  hb_fetch4(h, h.wb64 - 64);
}
__device__ __forceinline__ uint32_t hb_peek32(const HBits& h) {  // the 32 bits below pos, bit 31 = the next unread bit
  const uint32_t s = (uint32_t)(h.pos - h.wb64);
  return (uint32_t)(((h.lo >> s) | (h.hi1 << (63u - s))) >> 32);
}
__device__ __forceinline__ void hb_skip(HBits& h, int nb) {
  h.pos -= nb;
  if (h.pos < h.wb64) {
    h.hi1 = h.lo << 1;
    h.lo = h.nxi == 0 ? h.nx[0] : (h.nxi == 1 ? h.nx[1] : (h.nxi == 2 ? h.nx[2] : h.nx[3]));
    h.wb64 -= 64;
    if (++h.nxi == 4) hb_fetch4(h, h.wb64 - 64);
  }
}

// The same cursor with its bytes in LDS (zstd_literals_kernel).  With the window refilled from memory where a lane needs it, SOME
// lane of the wavefront issues a load in nearly every trip of the symbol loop, and the wait in front of the next use of a
// refilled word -- one counter for the whole wavefront -- then waits for that load: a memory round trip per symbol, ~1-2 us
// (1 650 symbol steps per lane took 1.7 ms).  Here every lane copies ZL_CHUNK bytes of its segment to LDS at once -- all lanes in the
// same trip of an outer loop, ZL_CHUNK / 16 loads in flight per lane -- and the symbol loop touches LDS only; a lane whose chunk is
// used up leaves the inner loop and waits for the others, who are at most some symbols behind (same code lengths on average).
#ifndef ZL_CHUNK
#define ZL_CHUNK 64
#endif
struct HBitsL {
  const uint8_t* p;
  uint8_t* cb;       // this wavefront's chunk buffer: 64 x ZL_CHUNK bytes, piece-interleaved (piece j of lane l at (j * 64 + l) * 16)
  uint64_t lo, hi1;  // as HBits
  uint64_t pf[ZL_CHUNK / 8];  // the chunk below the one in LDS, on its way from memory while that one is decoded
  uint32_t idx;      // words of the chunk taken so far (from its top); ZL_CHUNK / 8: used up
  int e;             // the chunk in LDS: bytes [e - ZL_CHUNK, e) of the stream; -1: none (behind a seek)
  int wb64;
  int pos;
};
__device__ __forceinline__ void hb_seek(HBitsL& h, int pos) {
  h.pos = pos;
  h.wb64 = (pos & ~7) - 56;
  h.idx = ZL_CHUNK / 8;  // (nothing loaded: the refill in front of the loop does)
  h.e = -1;
}
__device__ __forceinline__ uint64_t hbl_word(const HBitsL& h, uint32_t idx, uint32_t lane) {  // word `idx` of the chunk, counted from its top
  const uint32_t o = ZL_CHUNK - 8 * (idx + 1);
  return *reinterpret_cast<const uint64_t*>(h.cb + (((o >> 4) * 64 + lane) << 4) + (o & 8));
}
// bytes [e - ZL_CHUNK, e) of the stream into pf (bytes before the stream read as zero)
__device__ __forceinline__ void hbl_load(HBitsL& h, int e) {
#pragma unroll
  for (int j = 0; j < ZL_CHUNK / 16; j++) {
    const int bo = e - ZL_CHUNK + 16 * j;
    uint64_t v[2] = {0, 0};
    if (bo >= 0) {
      __builtin_memcpy(v, h.p + bo, 16);
    } else if (bo > -16) {
      v[0] = zl_word(h.p, 8 * bo);
      v[1] = zl_word(h.p, 8 * bo + 64);
    }
    h.pf[2 * j] = v[0];
    h.pf[2 * j + 1] = v[1];
  }
}
// The next chunk: the window's two words are its top (a chunk used up leaves the window 16 bytes above its bottom: the chunks
// follow each other at a fixed distance, ZL_CHUNK - 16 bytes, so the one after is requested at once and has a whole chunk's
// decoding to arrive in).  Behind a seek the first chunk is waited for.
__device__ __forceinline__ void hb_refill(HBitsL& h, bool need, uint32_t lane) {
  if (need) {
    if (h.e < 0) {
      h.e = (h.wb64 + 64) >> 3;
      hbl_load(h, h.e);
    } else {
      h.e -= ZL_CHUNK - 16;
    }
#pragma unroll
    for (int j = 0; j < ZL_CHUNK / 16; j++) {
      const uint64_t v[2] = {h.pf[2 * j], h.pf[2 * j + 1]};
      __builtin_memcpy(h.cb + ((j * 64 + lane) << 4), v, 16);
    }
    hbl_load(h, h.e - (ZL_CHUNK - 16));
    h.hi1 = hbl_word(h, 0, lane) << 1;
    h.lo = hbl_word(h, 1, lane);
    h.idx = 2;
  }
}
__device__ __forceinline__ bool hb_ready(const HBitsL& h) { return h.idx < ZL_CHUNK / 8; }
__device__ __forceinline__ uint32_t hb_peek32(const HBitsL& h) {
  const uint32_t s = (uint32_t)(h.pos - h.wb64);
  return (uint32_t)(((h.lo >> s) | (h.hi1 << (63u - s))) >> 32);
}
__device__ __forceinline__ void hb_skip(HBitsL& h, int nb, uint32_t lane) {
  h.pos -= nb;
  if (h.pos < h.wb64) {
    h.hi1 = h.lo << 1;
    h.lo = hbl_word(h, h.idx, lane);
    h.wb64 -= 64;
    h.idx++;
  }
}
// (the cursor that reads memory: always ready, nothing to refill)
__device__ __forceinline__ void hb_refill(HBits&, bool, uint32_t) {}
__device__ __forceinline__ bool hb_ready(const HBits&) { return true; }
__device__ __forceinline__ void hb_skip(HBits& h, int nb, uint32_t) { hb_skip(h, nb); }

template <bool LW = false>
__device__ __forceinline__ int huf_decode_par(const uint16_t* tab, int mb, const uint8_t* sp, uint32_t sn, uint8_t* out, uint32_t outn, uint32_t k,
                                              uint32_t lps, bool on PROF_PARM, uint8_t* cbuf = nullptr) {
  int bad = 0;
  int top = 0;
  if (on) {
    RBits r;
    if (!rb_init(r, sp, sn)) bad = 1;
    top = (int)r.bits;
  }
  const uint32_t wl = threadIdx.x & 63;  // lane of the wavefront (k: lane of the stream)
  typename std::conditional<LW, HBitsL, HBits>::type h{};
  h.p = sp;
  if constexpr (LW) h.cb = cbuf;
  const int B = (top + (int)lps - 1) / (int)lps;
  int pk = top - (int)k * B, pn = k + 1 == lps ? 0 : top - (int)(k + 1) * B;
  if (pk < 0) pk = 0;
  if (pn < 0) pn = 0;
  int start = pk, endpos = pk;
  uint32_t cnt = 0;
  const bool work = on && !bad;
  // Round 0 decodes every segment from its nominal start; a lane then takes the point where its predecessor crossed into its
  // segment as its true start, a few bits further down.  The new path meets the old one within a few symbols (prefix codes
  // re-synchronise) and is the same from there on: round 0 notes where it stood after 16, 64, 256 and 1024 symbols, round 1
  // decodes only until it stands on one of those marks.  Lanes whose start did not change keep what they have.
  int ck0 = -1, ck1 = -1, ck2 = -1, ck3 = -1;
  bool redo = true;
  PROF_MARK(3);
  for (uint32_t round = 0; round < lps; round++) {
    if (work && redo) {
      const uint32_t cnt_old = cnt;
      const int end_old = endpos;
      cnt = 0;
      endpos = start;
      if (start > pn) {
        hb_seek(h, start);
        bool met = false;
        if (round == 0) {
          uint32_t limit = 16, jj = 0;
          for (bool stop = false;;) {
          hb_refill(h, !stop && h.pos > pn, wl);
          while (h.pos > pn && hb_ready(h)) {
            const uint32_t e = tab[hb_peek32(h) >> (32 - mb)];
            const int nb = (int)(e >> 8);
            if (nb == 0) {  // not a table zstd builds: no progress possible
              bad = 1;
              stop = true;
              break;
            }
            cnt++;
            hb_skip(h, nb, wl);
            if (cnt == limit) {
              if (jj == 0) ck0 = h.pos;
              else if (jj == 1) ck1 = h.pos;
              else if (jj == 2) ck2 = h.pos;
              else if (jj == 3) ck3 = h.pos;
              jj++;
              limit <<= 2;
            }
          }
          if (!LW || stop || h.pos <= pn) break;
          }
        } else {
          // (marks are only trusted in round 1: later rounds -- rare -- decode their segment whole)
          uint32_t jj = round == 1 ? 0u : 4u;
          for (bool stop = false;;) {
          hb_refill(h, !stop && h.pos > pn, wl);
          while (h.pos > pn && hb_ready(h)) {
            if (jj < 4) {
              int ck = jj == 0 ? ck0 : (jj == 1 ? ck1 : (jj == 2 ? ck2 : ck3));
              while (jj < 4 && h.pos < ck) {  // passed without standing on it (a mark never reached is -1: below everything)
                jj++;
                ck = jj == 1 ? ck1 : (jj == 2 ? ck2 : ck3);
              }
              if (jj < 4 && h.pos == ck) {
                met = true;
                stop = true;
                break;
              }
            }
            const uint32_t e = tab[hb_peek32(h) >> (32 - mb)];
            const int nb = (int)(e >> 8);
            if (nb == 0) {
              bad = 1;
              stop = true;
              break;
            }
            cnt++;
            hb_skip(h, nb, wl);
          }
          if (!LW || stop || h.pos <= pn) break;
          }
          if (met) {
            cnt += cnt_old - (16u << (2 * jj));  // the old path had 16 * 4^jj symbols above this mark
            endpos = end_old;
          }
        }
        if (!met) endpos = h.pos;
      }
    }
    // predecessor's crossing point -> this lane's start (lane 0 of a stream starts at the top)
    const int prev = __shfl_up(endpos, 1);
    redo = false;
    if (k > 0 && work && prev != start) {
      start = prev;
      redo = true;
    }
    PROF_MARK(round == 0 ? 4 : 5);
    if (!__ballot(redo)) break;
  }
  // symbols before this lane inside its stream
  uint32_t incl = cnt;
  for (uint32_t o = 1; o < lps; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o);
    if (k >= o) incl += t;
  }
  const uint32_t glast = ((threadIdx.x & 63) & ~(lps - 1)) + lps - 1;
  const uint32_t total = __shfl(incl, glast);
  const int last_end = __shfl(endpos, glast);
  if (work && (total != outn || last_end != 0)) bad = 1;
  if (work && !bad && start > pn) {
    // symbols leave SIXTEEN at a time, from the first 16-byte boundary of the lane's part of the output on: a lane's part lies
    // between its neighbours' -- an 8-byte store at any alignment was written to memory as the 32-byte piece around it (3.9 x the
    // literals in HBM writes by the counters), a whole aligned 16-byte store is half a piece
    uint32_t i = incl - cnt;
    uint64_t acc0 = 0, acc1 = 0;
    uint32_t nacc = 0;
    // (bytes up to the boundary one by one)
    uint32_t head = (uint32_t)((16u - ((uint32_t)(uintptr_t)(out + i) & 15u)) & 15u);
    hb_seek(h, start);
    for (;;) {
    hb_refill(h, h.pos > pn, wl);
    while (h.pos > pn && hb_ready(h)) {
      const uint32_t e = tab[hb_peek32(h) >> (32 - mb)];
      const int nb = (int)(e >> 8);
      if (head) {
        out[i++] = (uint8_t)e;
        head--;
      } else {
        if (nacc < 8) acc0 |= (uint64_t)(e & 0xff) << (8 * nacc);
        else acc1 |= (uint64_t)(e & 0xff) << (8 * (nacc - 8));
        if (++nacc == 16) {
          const uint64_t v[2] = {acc0, acc1};
          __builtin_memcpy(__builtin_assume_aligned(out + i, 16), v, 16);
          i += 16;
          nacc = 0;
          acc0 = acc1 = 0;
        }
      }
      hb_skip(h, nb, wl);
    }
    if (!LW || h.pos <= pn) break;
    }
    for (uint32_t t = 0; t < nacc; t++) out[i + t] = (uint8_t)((t < 8 ? acc0 >> (8 * t) : acc1 >> (8 * (t - 8))));
  }
  PROF_MARK(6);
  return bad;
}
