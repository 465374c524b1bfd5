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
// `have`: bytes of the description the caller has put there already (zstd_literals_job stages a literals section's first bytes in one go).
__device__ __forceinline__ long fse_read_ncount_dev(const uint8_t* p, uint32_t n, int16_t* norm, int* nsym_io, int* log_out, int maxlog,
                                                    uint32_t lane, uint8_t* stg, uint32_t have = 0) {
  const uint32_t staged = have ? (n < have ? n : have) : (n < 128u ? n : 128u);
  if (!have) {
    if (lane < 32 && 4 * lane < staged) reinterpret_cast<uint32_t*>(stg)[lane] = ld_u32(p + 4 * lane);  // (up to 3 bytes behind the description: slack of the arena)
    wave_sync();
  }
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
// 64 stream bits from bit `wb` (a multiple of 8; may lie before the stream: those bits read as zero) of the stream at q
__device__ __forceinline__ uint64_t zl_word(const uint8_t* q, int wb) {
  const int bo = wb >> 3;
  const uint64_t v = ld_u64(q + (bo < 0 ? 0 : bo));
  const int neg = bo < 0 ? -bo : 0;  // bytes of the word that lie before the stream
  return neg >= 8 ? 0ull : v << (8 * neg);
}
#ifndef ZL_CHUNK
#define ZL_CHUNK 64  // bytes of its segment a lane holds in LDS
#endif

// ---- the symbol loops of the literals, SIXTEEN SYMBOLS A TRIP (zstd_literals_kernel, zstd_entropy_kernel) -------------------------------
// Every lane holds ZL_CHUNK bytes of its segment in LDS (round 5: with a window reloaded from memory where a lane needs it SOME lane
// issues a load in nearly every trip of the symbol loop, and the wait in front of any lane's next use of a reloaded register -- one
// counter for the whole wavefront -- waits for that load: a memory round trip per symbol); all lanes refill their chunk in the same
// trip of an outer loop, the chunk below already on its way.  Round 6 replaced the symbol loop: one symbol per trip cost ~28 vector
// and ~24 scalar instructions (-DORC_PROF + the ISA: the kernel alone is bound by what it issues -- two symbols per table cell made the
// steps fewer and the kernel no faster) -- the window arithmetic on two 64-bit words, the accumulation of symbols at a run-time byte
// position, and a handful of per-lane branches (segment end, chunk used up, window word used up, 16 bytes ready, bytes in front of
// the first 16-byte boundary), each of them exec-mask bookkeeping.  Here a lane far from its segment's end decodes sixteen symbols
// without any of the checks, ~11 vector instructions a symbol:
//   * the window is ONE 64-bit register of unread bits, left-aligned: the table index is its high word shifted right, a symbol is
//     consumed by one 64-bit shift;
//   * behind every second symbol (2 x 11 bits at most out of the 32 or more the window holds) the next 32-bit word of the lane's
//     chunk in LDS is put below the valid bits if fewer than 32 are left -- without a branch (the word is read either way);
//   * symbol j of a trip goes to byte j of four registers (v_perm_b32 with a constant selector), which leave as one aligned store;
//   * position, count and chunk bookkeeping once per trip: 176 bits at most, so a lane runs while its segment's end is that far off
//     and while fewer than ten of its chunk's sixteen words are taken (six more at most in a trip); the chunks follow each other at
//     40 bytes (ten words).
// The symbols in front of the first 16-byte boundary of a lane's output, the last 176 bits of a segment and round 1 (which has to stand
// on every code boundary) take the same cursor one symbol at a time.  (Literals kernel of the headline, wavefront time in the three
// symbol loops: 14.5 M -> 8.2 M wave-us per lane of columns; the kernel beside the sequences kernel 8.3 / 9.7 -> 7.4 / 7.2 ms.)
#define HW_STRIDE 40
__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)a); }
__device__ __forceinline__ uint32_t lds_ld16(uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>((uintptr_t)a); }
__device__ __forceinline__ void lds_st32(uint32_t a, uint32_t v) { *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)a) = v; }
struct HWin {
  const uint8_t* p;
  uint32_t a0;      // LDS address of the word this lane takes FIRST from a chunk (the j-th: a0 + 256 j; bytes [e - 4 (j + 1), e - 4 j) of the stream)
  uint32_t a;       // ... of the word it takes next; a0 + 2560 or more: the chunk is used up
  uint64_t w;       // unread bits, left-aligned (bit 63 = stream bit pos - 1), zero below the valid ones
  int av;           // valid bits of w: 32 .. 63 between symbols
  int e;            // the chunk in LDS: stream bytes [e - 64, e); -1: none (behind a seek)
  int pos;
  uint64_t pf[8];   // the chunk below on its way from memory
  uint32_t last;    // the stream's last byte (its highest set bit ends the stream), requested by hw_begin
  int e_pre;        // hw_begin has requested stream bytes [e_pre - 64, e_pre) into pf: round 0's first chunk
};
__device__ __forceinline__ void hw_seek(HWin& h, int pos) {
  h.pos = pos;
  h.e = -1;
  h.a = h.a0 + 2560;  // (nothing there: the refill in front of the loops loads)
  h.av = 32;
  h.w = 0;
}
__device__ __forceinline__ bool hw_ready(const HWin& h) { return h.a < h.a0 + 2560; }
// bytes [e - 64, e) of the stream into pf (bytes before the stream read as zero; up to 4 behind it are read: ORC_PAD)
__device__ __forceinline__ void hw_load(HWin& h, int e) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int bo = e - 64 + 16 * j;
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
// In front of everything else a block needs (hw_begin is called before the block's Huffman table is built): the stream's last byte and
// the first chunk of the lane's segment are requested TOGETHER.  Where the segment starts depends on that byte (top = its highest set
// bit; lane k starts at top - k * ceil(top / lps)), but only by 7 + k bits: the chunk is taken from the highest start there can be
// and the window passes over what lies above the true one.  A wavefront's first chunks are 64 lines nobody has touched: ~100 us
// before the symbol loops could start (-DORC_PROF), now under the 50 - 100 us the table takes.
__device__ __forceinline__ void hw_begin(HWin& h, const uint8_t* sp, uint32_t sn, uint32_t k, uint32_t lps, uint8_t* cbuf) {
  h.p = sp;
  h.a0 = (uint32_t)(uintptr_t)cbuf + (threadIdx.x & 63) * 4;
  h.last = sn ? sp[sn - 1] : 0u;
  const int T = 8 * ((int)sn - 1) + 7;                          // the highest top there can be
  const int bmin = (T - 7 + (int)lps - 1) / (int)lps;         // the shortest segments
  const int U = T - (int)k * bmin;                             // >= the lane's start, by 7 + k bits at most
  h.e_pre = ((U + 32) >> 5) << 2;
  hw_load(h, h.e_pre);
}
__device__ __forceinline__ void hw_refill(HWin& h, bool need) {
  if (need) {
    int junk = -1;
    if (h.e < 0) {
      if (h.e_pre != -1) {
        h.e = h.e_pre;  // (requested by hw_begin)
        h.e_pre = -1;
      } else {
        h.e = ((h.pos + 32) >> 5) << 2;  // (a bit above pos at least: the window never holds 64 valid bits)
        hw_load(h, h.e);
      }
      junk = 8 * h.e - h.pos;  // bits of the chunk above pos: 1 .. 32, or up to 102 in hw_begin's chunk
    } else {
      h.e -= HW_STRIDE;
      h.a -= 2560;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      lds_st32(h.a0 + 256u * (uint32_t)(15 - 2 * k), (uint32_t)h.pf[k]);
      lds_st32(h.a0 + 256u * (uint32_t)(14 - 2 * k), (uint32_t)(h.pf[k] >> 32));
    }
    hw_load(h, h.e - HW_STRIDE);
    if (junk >= 0) {
      const uint32_t jw = (uint32_t)(junk - 1) >> 5, jr = (uint32_t)junk - 32u * jw;  // whole words to pass over, 1 .. 32 bits of the next one
      const uint32_t at = h.a0 + 256u * jw;
      const uint64_t t = ((uint64_t)lds_ld32(at) << 32) | lds_ld32(at + 256);
      h.w = jr < 32 ? t << jr : t << 32;
      h.av = 64 - (int)jr;
      h.a = at + 512;
    }
  }
}
// behind a symbol or two: the next word of the chunk below the valid bits when fewer than 32 are left (no branch)
__device__ __forceinline__ void hw_top(HWin& h, uint32_t word) {
  const bool need = h.av < 32;
  const uint64_t t = ((uint64_t)word << 32) >> (h.av & 63);  // av >= 32: nothing in the high word, the low word is dropped
  h.w |= ((t >> 32) << 32) | (need ? (uint32_t)t : 0u);
  h.av += need ? 32 : 0;
  h.a += need ? 256u : 0u;
}
// one symbol with the cursor's bookkeeping; returns the table cell (symbol | bits << 8)
__device__ __forceinline__ uint32_t hw_step1(HWin& h, uint32_t tbase, uint32_t sh) {
  const uint32_t word = lds_ld32(h.a);
  const uint32_t e = lds_ld16(tbase + (((uint32_t)(h.w >> 32) >> sh) << 1));
  const uint32_t nb = e >> 8;
  h.w <<= nb;
  h.av -= (int)nb;
  h.pos -= (int)nb;
  hw_top(h, word);
  return e;
}
// N symbols (16, or 4 near a segment's end): EMIT: their bytes in acc[0 .. N / 4); returns the bits they took
template <int N, bool EMIT>
__device__ __forceinline__ int hw_steps(HWin& h, uint32_t tbase, uint32_t sh, uint32_t* acc) {
  const int av0 = h.av;
  const uint32_t a_0 = h.a;
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const uint32_t word = lds_ld32(h.a);
    const uint32_t e0 = lds_ld16(tbase + (((uint32_t)(h.w >> 32) >> sh) << 1));
    const uint32_t n0 = e0 >> 8;
    h.w <<= n0;
    const uint32_t e1 = lds_ld16(tbase + (((uint32_t)(h.w >> 32) >> sh) << 1));
    const uint32_t n1 = e1 >> 8;
    h.w <<= n1;
    h.av -= (int)(n0 + n1);
    if constexpr (EMIT) {
      uint32_t& r = acc[j >> 2];
      if ((j & 3) == 0) {
        r = __builtin_amdgcn_perm(e1, e0, 0x0c0c0400u);  // byte 0 <- e0, byte 1 <- e1
      } else {
        r = __builtin_amdgcn_perm(e0, r, 0x0c040100u);   // byte 2 <- e0
        r = __builtin_amdgcn_perm(e1, r, 0x04020100u);   // byte 3 <- e1
      }
    }
    hw_top(h, word);
  }
  const int used = (av0 - h.av) + (int)((h.a - a_0) >> 3);
  h.pos -= used;
  return used;  // (N or more with a table zstd builds: every cell holds a code of one bit at least)
}
#define HW_FAR 176  // bits a trip of sixteen symbols may take
#define HW_FAR4 44  // ... of four

// `h`: begun for this lane's stream (hw_begin: sp, sn, k, lps)
__device__ __forceinline__ int huf_decode_w16(HWin& h, const uint16_t* tab, int mb, uint32_t sn, uint8_t* out, uint32_t outn, uint32_t k,
                                              uint32_t lps, bool on PROF_PARM) {
  int bad = 0;
  int top = 0;
  if (on) {
    if (sn == 0 || h.last == 0) bad = 1;  // (rb_init)
    else top = (int)(sn - 1) * 8 + (31 - __builtin_clz(h.last));
  }
  PROF_MARK(11);
  const uint32_t tbase = (uint32_t)(uintptr_t)tab;
  const uint32_t sh = 32u - (uint32_t)mb;
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
        hw_seek(h, start);
        bool met = false;
        if (round == 0) {
          uint32_t limit = 16, jj = 0;
          auto mark = [&]() {  // (selects, not an indexed store: the marks stay in registers)
            ck0 = jj == 0 ? h.pos : ck0;
            ck1 = jj == 1 ? h.pos : ck1;
            ck2 = jj == 2 ? h.pos : ck2;
            ck3 = jj == 3 ? h.pos : ck3;
            jj++;
            limit <<= 2;
          };
          for (;;) {
            hw_refill(h, h.pos > pn);
            while (h.pos - pn >= HW_FAR && hw_ready(h)) {
              if (hw_steps<16, false>(h, tbase, sh, nullptr) < 16) {  // not a table zstd builds: no progress possible
                bad = 1;
                h.pos = pn;
              }
              cnt += 16;
              if (cnt == limit) mark();
            }
            while (h.pos - pn >= HW_FAR4 && h.pos - pn < HW_FAR && hw_ready(h)) {  // (the segment's last bits: a lane that is far from them has used its chunk up)
              if (hw_steps<4, false>(h, tbase, sh, nullptr) < 4) {
                bad = 1;
                h.pos = pn;
              }
              cnt += 4;
              if (cnt == limit) mark();
            }
            while (h.pos > pn && h.pos - pn < HW_FAR4 && hw_ready(h)) {
              if (hw_step1(h, tbase, sh) < 256) {
                bad = 1;
                h.pos = pn;
              }
              cnt++;
              if (cnt == limit) mark();
            }
            if (h.pos <= pn) break;
          }
        } else {
          // (marks are only trusted in round 1: later rounds -- rare -- decode their segment whole)
          uint32_t jj = round == 1 ? 0u : 4u;
          for (bool stop = false;;) {
            hw_refill(h, !stop && h.pos > pn);
            while (h.pos > pn && hw_ready(h)) {
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
              if (hw_step1(h, tbase, sh) < 256) {
                bad = 1;
                h.pos = pn;
              }
              cnt++;
            }
            if (stop || h.pos <= pn) break;
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
    // literals in HBM writes by the counters), a whole aligned 16-byte store is half a piece; the bytes in front of the boundary
    // and the symbols of the segment's last bits one by one
    uint32_t i = incl - cnt;
    uint32_t head = (uint32_t)((16u - ((uint32_t)(uintptr_t)(out + i) & 15u)) & 15u);
    hw_seek(h, start);
    for (;;) {
      hw_refill(h, h.pos > pn);
      while (head && h.pos > pn && hw_ready(h)) {
        const uint32_t c = hw_step1(h, tbase, sh);
        if (c < 256) h.pos = pn;
        else out[i++] = (uint8_t)c;
        head--;
      }
      while (!head && h.pos - pn >= HW_FAR && hw_ready(h)) {
        uint32_t acc[4];
        if (hw_steps<16, true>(h, tbase, sh, acc) < 16) h.pos = pn;  // (no progress: not a table zstd builds -- round 0 has said so)
        __builtin_memcpy(__builtin_assume_aligned(out + i, 16), acc, 16);
        i += 16;
      }
      while (!head && h.pos - pn >= HW_FAR4 && h.pos - pn < HW_FAR && hw_ready(h)) {
        uint32_t acc;
        if (hw_steps<4, true>(h, tbase, sh, &acc) < 4) h.pos = pn;
        __builtin_memcpy(__builtin_assume_aligned(out + i, 4), &acc, 4);
        i += 4;
      }
      while (!head && h.pos > pn && h.pos - pn < HW_FAR4 && hw_ready(h)) {
        const uint32_t c = hw_step1(h, tbase, sh);
        if (c < 256) h.pos = pn;
        else out[i++] = (uint8_t)c;
      }
      if (h.pos <= pn) break;
    }
  }
  PROF_MARK(6);
  return bad;
}

