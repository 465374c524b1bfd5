// zstd_device.h -- Zstandard frame decoder (RFC 8878) for one wavefront.  Replaces zstd::Decoder at
// compression.rs:151-159.  No dictionaries; the content checksum is skipped, not verified.
//
// Per compressed block: the literals section is decoded first (4 Huffman streams -> lanes 0..3
// decode one stream each; raw literals are used in place), then the sequence section is decoded
// wave-uniformly (three interleaved FSE states read backwards) and each sequence is executed by
// all lanes: literal copy from the literal buffer + LZ77 match copy.  FSE/Huffman tables live in
// LDS; decoded literals of a block (<= 128 KiB) go to a per-chunk scratch area in HBM.
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
__device__ __forceinline__ long fse_read_ncount_dev(const uint8_t* p, uint32_t n, int16_t* norm, int* nsym_io, int* log_out, int maxlog,
                                                    uint32_t lane) {
  uint32_t pos = 0;
  uint64_t bb = 0;
  int bc = 0;
#define ZNEED(k)                                  \
  while (bc < (k)) {                              \
    uint64_t byte_ = pos < n ? p[pos] : 0;        \
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

// Huffman decoding table from weights (last weight implied); uniform, lane 0 writes
__device__ __forceinline__ int huf_build_dev(uint16_t* tab, int* maxbits_out, uint8_t* w, int nw, uint32_t lane) {
  int bad = 0, maxbits = 0;
  if (lane == 0) {
    uint32_t sum = 0;
    for (int i = 0; i < nw; i++) {
      if (w[i] > 11) bad = 1;
      else if (w[i]) sum += 1u << (w[i] - 1);
    }
    if (sum == 0) bad = 1;
    if (!bad) {
      maxbits = z_hibit(sum) + 1;
      if (maxbits > 11) bad = 1;
    }
    if (!bad) {
      uint32_t left = (1u << maxbits) - sum;
      if (left & (left - 1)) bad = 1;
      else {
        w[nw] = (uint8_t)(z_hibit(left) + 1);
        int n2 = nw + 1;
        uint32_t rankstart[13] = {0}, cnt[13] = {0};
        for (int i = 0; i < n2; i++) cnt[w[i]]++;
        uint32_t pos = 0;
        for (int wt = 1; wt <= maxbits; wt++) {
          rankstart[wt] = pos;
          pos += cnt[wt] << (wt - 1);
        }
        if (pos != (1u << maxbits)) bad = 1;
        for (int s = 0; s < n2 && !bad; s++) {
          if (!w[s]) continue;
          uint32_t len = 1u << (w[s] - 1);
          uint32_t st = rankstart[w[s]];
          uint16_t e = (uint16_t)(s | ((maxbits + 1 - w[s]) << 8));
          for (uint32_t i = 0; i < len; i++) tab[st + i] = e;
          rankstart[w[s]] += len;
        }
      }
    }
  }
  bad = __shfl(bad, 0);
  *maxbits_out = __shfl(maxbits, 0);
  wave_sync();
  return bad;
}

// one Huffman stream, executed by ONE lane (others predicated off by the caller)
__device__ __forceinline__ int huf_decode_stream_dev(const uint16_t* tab, int mb, const uint8_t* p, uint32_t n, uint8_t* out, uint32_t outn) {
  RBits r;
  if (!rb_init(r, p, n)) return 1;
  uint32_t state = (uint32_t)rb_read(r, (uint32_t)mb);
  uint32_t mask = (1u << mb) - 1;
  for (uint32_t i = 0; i < outn; i++) {
    uint32_t e = tab[state];
    out[i] = (uint8_t)e;
    uint32_t nb = e >> 8;
    state = ((state << nb) & mask) | (uint32_t)rb_read(r, nb);
  }
  return r.bits != -(long)mb;
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
  uint64_t w;
  int pos, avail;
};
__device__ __forceinline__ void hb_seek(HBits& h, int pos) {
  h.pos = pos;
  const int bytepos = (pos + 7) >> 3;  // first byte at or above pos
  uint64_t v;
  if (bytepos >= 8) v = ld_u64(h.p + bytepos - 8);
  else v = bytepos > 0 ? ld_u64(h.p) << (8 * (8 - bytepos)) : 0;
  const int waste = 8 * bytepos - pos;  // 0..7 bits at the top that lie above pos
  h.w = v << waste;
  h.avail = 64 - waste;
}

__device__ __forceinline__ int huf_decode_par(const uint16_t* tab, int mb, const uint8_t* sp, uint32_t sn, uint8_t* out, uint32_t outn, uint32_t k,
                                              uint32_t lps, bool on) {
  int bad = 0;
  int top = 0;
  if (on) {
    RBits r;
    if (!rb_init(r, sp, sn)) bad = 1;
    top = (int)r.bits;
  }
  HBits h{sp, 0, 0, 0};
  const int B = (top + (int)lps - 1) / (int)lps;
  int pk = top - (int)k * B, pn = k + 1 == lps ? 0 : top - (int)(k + 1) * B;
  if (pk < 0) pk = 0;
  if (pn < 0) pn = 0;
  int start = pk, endpos = pk;
  uint32_t cnt = 0;
  const bool work = on && !bad;
  for (uint32_t round = 0; round < lps; round++) {
    cnt = 0;
    endpos = start;
    if (work && start > pn) {
      hb_seek(h, start);
      while (h.pos > pn) {
        const uint32_t e = tab[h.w >> (64 - mb)];
        const int nb = (int)(e >> 8);
        if (nb == 0) {  // not a table zstd builds: no progress possible
          bad = 1;
          break;
        }
        cnt++;
        h.w <<= nb;
        h.pos -= nb;
        h.avail -= nb;
        if (h.avail < mb) hb_seek(h, h.pos);
      }
      endpos = h.pos;
    }
    // predecessor's crossing point -> this lane's start (lane 0 of a stream starts at the top)
    const int prev = __shfl_up(endpos, 1);
    bool changed = false;
    if (k > 0 && work && prev != start) {
      start = prev;
      changed = true;
    }
    if (!__ballot(changed)) break;
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
    // symbols leave eight at a time (one 8-byte store instead of eight scattered single-byte ones: the
    // stores of a block used to take longer to drain than the decoding itself)
    uint32_t i = incl - cnt, nacc = 0;
    uint64_t acc = 0;
    hb_seek(h, start);
    while (h.pos > pn) {
      const uint32_t e = tab[h.w >> (64 - mb)];
      const int nb = (int)(e >> 8);
      acc |= (uint64_t)(e & 0xff) << (8 * nacc);
      if (++nacc == 8) {
        __builtin_memcpy(out + i, &acc, 8);
        i += 8;
        nacc = 0;
        acc = 0;
      }
      h.w <<= nb;
      h.pos -= nb;
      h.avail -= nb;
      if (h.avail < mb) hb_seek(h, h.pos);
    }
    for (uint32_t t = 0; t < nacc; t++) out[i + t] = (uint8_t)(acc >> (8 * t));
  }
  return bad;
}

struct ZState {
  int huf_valid, huf_bits;
  int ll_valid, of_valid, ml_valid;
  int ll_log, of_log, ml_log;
  uint32_t rep[3];
};

// literals section: returns bytes consumed (<0 error); *lit/*litn describe the decoded literals
__device__ __forceinline__ long z_literals_dev(ZState& z, DecompLds& L, const uint8_t* p, uint32_t n, uint8_t* scratch, const uint8_t** lit,
                                               uint32_t* litn, uint32_t lane PROF_PARM) {
  if (n < 1) return -1;
  uint32_t type = p[0] & 3, sf = (p[0] >> 2) & 3;
  uint32_t regen, comp = 0, hdr;
  int streams = 1;
  if (type < 2) {
    if (sf == 0 || sf == 2) {
      regen = p[0] >> 3;
      hdr = 1;
    } else if (sf == 1) {
      if (n < 2) return -1;
      regen = (p[0] >> 4) | ((uint32_t)p[1] << 4);
      hdr = 2;
    } else {
      if (n < 3) return -1;
      regen = (p[0] >> 4) | ((uint32_t)p[1] << 4) | ((uint32_t)p[2] << 12);
      hdr = 3;
    }
    if (regen > 128 * 1024) return -1;
    if (type == 0) {
      if (hdr + regen > n) return -1;
      *lit = p + hdr;
      *litn = regen;
      return (long)(hdr + regen);
    }
    if (hdr + 1 > n || !scratch) return -1;
    uint8_t v = p[hdr];
    for (uint32_t k = lane; k < regen; k += 64) scratch[k] = v;
    wave_fence();
    *lit = scratch;
    *litn = regen;
    return (long)(hdr + 1);
  }
  if (sf == 0 || sf == 1) {
    if (n < 3) return -1;
    uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16);
    regen = (v >> 4) & 0x3ff;
    comp = (v >> 14) & 0x3ff;
    hdr = 3;
    streams = sf == 0 ? 1 : 4;
  } else if (sf == 2) {
    if (n < 4) return -1;
    uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    regen = (v >> 4) & 0x3fff;
    comp = (v >> 18) & 0x3fff;
    hdr = 4;
    streams = 4;
  } else {
    if (n < 5) return -1;
    uint64_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint64_t)p[3] << 24) | ((uint64_t)p[4] << 32);
    regen = (uint32_t)((v >> 4) & 0x3ffff);
    comp = (uint32_t)((v >> 22) & 0x3ffff);
    hdr = 5;
    streams = 4;
  }
  if (regen > 128 * 1024 || hdr + comp > n || !scratch) return -1;
  const uint8_t* q = p + hdr;
  uint32_t qn = comp;
  if (type == 2) {
    if (qn < 1) return -1;
    int nw;
    uint32_t hb = q[0];
    uint32_t used;
    if (hb >= 128) {
      nw = (int)hb - 127;
      uint32_t nbytes = (uint32_t)(nw + 1) / 2;
      if (1 + nbytes > qn) return -1;
      for (int i = (int)lane; i < nw; i += 64) L.z.weights[i] = (i & 1) ? (q[1 + i / 2] & 15) : (q[1 + i / 2] >> 4);
      used = 1 + nbytes;
      wave_sync();
    } else {
      if (1 + hb > qn) return -1;
      int nsym = 256, log;
      long c = fse_read_ncount_dev(q + 1, hb, L.z.norm, &nsym, &log, 6, lane);
      if (c < 0) return -1;
      if (fse_build_dev(L.z.wt, L.z.norm, nsym, log, L.z.next, lane)) return -1;
      RBits r;
      if (!rb_init(r, q + 1 + c, hb - (uint32_t)c)) return -1;
      uint32_t s1 = (uint32_t)rb_read(r, (uint32_t)log), s2 = (uint32_t)rb_read(r, (uint32_t)log);
      nw = 0;
      int fail = 0;
      for (;;) {
        if (nw >= 254) {
          fail = 1;
          break;
        }
        if (lane == 0) L.z.weights[nw] = L.z.wt[s1].sym;
        nw++;
        if (r.bits < (long)L.z.wt[s1].nb) {
          if (lane == 0) L.z.weights[nw] = L.z.wt[s2].sym;
          nw++;
          break;
        }
        s1 = L.z.wt[s1].base + (uint32_t)rb_read(r, L.z.wt[s1].nb);
        if (nw >= 254) {
          fail = 1;
          break;
        }
        if (lane == 0) L.z.weights[nw] = L.z.wt[s2].sym;
        nw++;
        if (r.bits < (long)L.z.wt[s2].nb) {
          if (lane == 0) L.z.weights[nw] = L.z.wt[s1].sym;
          nw++;
          break;
        }
        s2 = L.z.wt[s2].base + (uint32_t)rb_read(r, L.z.wt[s2].nb);
      }
      if (fail) return -1;
      used = 1 + hb;
      wave_sync();
    }
    PROF_MARK(5);
    if (huf_build_dev(L.z.huf, &z.huf_bits, L.z.weights, nw, lane)) return -1;
    PROF_MARK(6);
    z.huf_valid = 1;
    q += used;
    qn -= used;
  } else if (!z.huf_valid) {
    return -1;
  }
  PROF_MARK(7);
  int bad = 0;
  if (streams == 1) {
    bad = huf_decode_par(L.z.huf, z.huf_bits, q, qn, scratch, regen, lane, 64, true);
  } else {
    if (qn < 6) return -1;
    uint32_t s1 = q[0] | (q[1] << 8), s2 = q[2] | (q[3] << 8), s3 = q[4] | (q[5] << 8);
    if (6 + s1 + s2 + s3 > qn) return -1;
    uint32_t s4 = qn - 6 - s1 - s2 - s3;
    uint32_t seg = (regen + 3) / 4;
    if (seg * 3 > regen) return -1;
    const uint8_t* b = q + 6;
    const uint32_t st = lane >> 4;  // stream of this lane (16 lanes each)
    uint32_t so = st == 0 ? 0 : (st == 1 ? s1 : (st == 2 ? s1 + s2 : s1 + s2 + s3));
    uint32_t sl = st == 0 ? s1 : (st == 1 ? s2 : (st == 2 ? s3 : s4));
    uint32_t on = st < 3 ? seg : regen - 3 * seg;
    bad = huf_decode_par(L.z.huf, z.huf_bits, b + so, sl, scratch + st * seg, on, lane & 15, 16, true);
  }
  PROF_MARK(8);
  if (__ballot(bad != 0)) return -1;
  wave_fence();
  *lit = scratch;
  *litn = regen;
  return (long)(hdr + comp);
}

__device__ __forceinline__ long z_seq_table_dev(FseEnt* t, int* valid, int* log_io, int mode, const uint8_t* p, uint32_t n, const int16_t* def,
                                                int defn, int deflog, int maxsym, int maxlog, DecompLds& L, uint32_t lane) {
  if (mode == 0) {
    for (int i = (int)lane; i < defn; i += 64) L.z.norm[i] = def[i];
    wave_sync();
    if (fse_build_dev(t, L.z.norm, defn, deflog, L.z.next, lane)) return -1;
    *valid = 1;
    *log_io = deflog;
    return 0;
  }
  if (mode == 1) {
    if (n < 1) return -1;
    if (lane == 0) {
      t[0].sym = p[0];
      t[0].nb = 0;
      t[0].base = 0;
    }
    wave_sync();
    *valid = 1;
    *log_io = 0;
    return 1;
  }
  if (mode == 2) {
    int nsym = maxsym, log;
    long c = fse_read_ncount_dev(p, n, L.z.norm, &nsym, &log, maxlog, lane);
    if (c < 0) return -1;
    if (fse_build_dev(t, L.z.norm, nsym, log, L.z.next, lane)) return -1;
    *valid = 1;
    *log_io = log;
    return c;
  }
  return *valid ? 0 : -1;
}

// one compressed block; returns the new output size or -1
// (returns 0 or -1; the output goes through the LDS window `o`; match distances count from frame_start)
__device__ __forceinline__ long z_block_dev(ZState& z, DecompLds& L, const uint8_t* p, uint32_t n, LzOut& o, uint64_t frame_start, uint64_t cap,
                                            uint8_t* scratch, LzLds Z, uint32_t lane PROF_PARM) {
  const uint8_t* lit = nullptr;
  uint32_t litn = 0;
  PROF_MARK(10);
  long used = z_literals_dev(z, L, p, n, scratch, &lit, &litn, lane PROF_ARG);
  PROF_MARK(11);
  if (used < 0) return -1;
  const uint8_t* q = p + used;
  uint32_t qn = n - (uint32_t)used;
  if (qn < 1) return -1;
  uint32_t nseq;
  if (q[0] < 128) {
    nseq = q[0];
    q += 1;
    qn -= 1;
  } else if (q[0] < 255) {
    if (qn < 2) return -1;
    nseq = ((uint32_t)(q[0] - 128) << 8) + q[1];
    q += 2;
    qn -= 2;
  } else {
    if (qn < 3) return -1;
    nseq = (uint32_t)q[1] + ((uint32_t)q[2] << 8) + 0x7f00;
    q += 3;
    qn -= 3;
  }
  uint32_t lp = 0;
  if (nseq) {
    if (qn < 1) return -1;
    uint32_t modes = q[0];
    if (modes & 3) return -1;
    q++;
    qn--;
    long c = z_seq_table_dev(L.z.ll, &z.ll_valid, &z.ll_log, (modes >> 6) & 3, q, qn, Z_LL_DEF, 36, 6, 36, 9, L, lane);
    if (c < 0) return -1;
    q += c;
    qn -= (uint32_t)c;
    c = z_seq_table_dev(L.z.of, &z.of_valid, &z.of_log, (modes >> 4) & 3, q, qn, Z_OF_DEF, 29, 5, 32, 8, L, lane);
    if (c < 0) return -1;
    q += c;
    qn -= (uint32_t)c;
    c = z_seq_table_dev(L.z.ml, &z.ml_valid, &z.ml_log, (modes >> 2) & 3, q, qn, Z_ML_DEF, 53, 6, 53, 9, L, lane);
    if (c < 0) return -1;
    q += c;
    qn -= (uint32_t)c;
    RBits r;
    if (!rb_init(r, q, qn)) return -1;
    uint32_t sl = (uint32_t)rb_read(r, (uint32_t)z.ll_log);
    uint32_t so = (uint32_t)rb_read(r, (uint32_t)z.of_log);
    uint32_t sm = (uint32_t)rb_read(r, (uint32_t)z.ml_log);
    PROF_MARK(12);
    // The FSE state machine is serial, the copies are not: sequences are decoded one after the other into
    // the group table (literal part, match part) and executed 60+ elements at a time by lz_group_run
    // (wave scan -> output positions, independent copies in parallel).  Literals come from `lit`.
    LzIn lin{lit, litn, Z.stage, 0};
    lzin_stage(lin, lp, lane);
    LzGroup G{0, 0, 0, 0};
    uint32_t gn = 0;
    uint64_t vout = o.out;  // output position once everything filed has been executed
    auto run_group = [&]() -> int {
      lds_order();
      G.len = lane < gn ? Z.g_len[lane] : 0;
      G.off = lane < gn ? Z.g_off[lane] : 0;
      G.src = lane < gn ? Z.g_src[lane] : 0;
      G.n = gn;
      gn = 0;
      return lz_group_run(G, lin, o, cap, lane PROF_ARG);
    };
    for (uint32_t i = 0; i < nseq; i++) {
      FseEnt el = L.z.ll[sl], eo = L.z.of[so], em = L.z.ml[sm];
      uint32_t oc = eo.sym, mc = em.sym, lc = el.sym;
      if (oc > 31 || mc > 52 || lc > 35) return -1;
      uint64_t ofv = (1ull << oc) + rb_read(r, oc);
      uint32_t mlen = Z_ML_BASE[mc] + (uint32_t)rb_read(r, Z_ML_BITS[mc]);
      uint32_t llen = Z_LL_BASE[lc] + (uint32_t)rb_read(r, Z_LL_BITS[lc]);
      if (r.bits < 0) return -1;
      uint64_t offset;
      if (ofv > 3) {
        offset = ofv - 3;
        z.rep[2] = z.rep[1];
        z.rep[1] = z.rep[0];
        z.rep[0] = (uint32_t)offset;
      } else {
        uint32_t idx = (uint32_t)ofv - 1 + (llen == 0 ? 1 : 0);
        if (idx == 0) {
          offset = z.rep[0];
        } else {
          offset = idx < 3 ? z.rep[idx] : z.rep[0] - 1;
          if (idx > 1) z.rep[2] = z.rep[1];
          z.rep[1] = z.rep[0];
          z.rep[0] = (uint32_t)offset;
        }
      }
      if (offset == 0) return -1;
      if ((uint64_t)lp + llen > litn || vout + llen + mlen > cap) return -1;
      if (offset > vout + llen - frame_start) return -1;
      if (gn > 62 && run_group()) return -1;
      if (llen > 64) {
        if (run_group()) return -1;
        lz_literal(o, lit + lp, llen, lane);
      } else if (llen) {
        if (lane == 0) {
          Z.g_len[gn] = llen;
          Z.g_off[gn] = 0;
          Z.g_src[gn] = lp;
        }
        gn++;
      }
      lp += llen;
      if (mlen > 64 || offset > 0xffffffffull) {
        if (run_group()) return -1;
        if (offset > 0xffffffffull) return -1;
        lz_match(o, (uint32_t)offset, mlen, lane);
      } else {
        if (lane == 0) {
          Z.g_len[gn] = mlen;
          Z.g_off[gn] = (uint32_t)offset;
          Z.g_src[gn] = 0;
        }
        gn++;
      }
      vout += llen + mlen;
      if (i + 1 < nseq) {
        sl = el.base + (uint32_t)rb_read(r, el.nb);
        sm = em.base + (uint32_t)rb_read(r, em.nb);
        so = eo.base + (uint32_t)rb_read(r, eo.nb);
        if (r.bits < 0) return -1;
      }
    }
    if (run_group()) return -1;
    if (r.bits != 0) return -1;
  }
  PROF_MARK(13);
  if (o.out + (litn - lp) > cap) return -1;
  lz_literal(o, lit + lp, litn - lp, lane);
  PROF_MARK(14);
  return 0;
}

__device__ __forceinline__ int zstd_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint8_t* scratch, uint32_t lane,
                                          uint32_t* out_len, DecompLds& L, LzLds Z PROF_PARM) {
  uint32_t pos = 0;
  LzOut o{Z.ring, Z.rsize - 1, dst, 0, 0};
  while (pos < n) {
    if (pos + 4 > n) return 1;
    uint32_t magic = ld_u32(src + pos);
    if ((magic & 0xfffffff0u) == 0x184d2a50u) {
      if (pos + 8 > n) return 1;
      uint32_t sz = ld_u32(src + pos + 4);
      if ((uint64_t)pos + 8 + sz > n) return 1;
      pos += 8 + sz;
      continue;
    }
    if (magic != 0xfd2fb528u) return 1;
    pos += 4;
    if (pos >= n) return 1;
    uint32_t fhd = src[pos++];
    uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, has_ck = (fhd >> 2) & 1, did_flag = fhd & 3;
    if (fhd & 0x08) return 1;
    if (!single) {
      if (pos >= n) return 1;
      pos++;
    }
    uint32_t did_bytes = did_flag == 3 ? 4 : did_flag;
    if (did_bytes) {
      if (pos + did_bytes > n) return 1;
      uint32_t did = 0;
      for (uint32_t i = 0; i < did_bytes; i++) did |= (uint32_t)src[pos + i] << (8 * i);
      pos += did_bytes;
      if (did) return 1;
    }
    uint32_t fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    uint64_t fcs = 0;
    if (pos + fcs_bytes > n) return 1;
    for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
    if (fcs_bytes == 2) fcs += 256;
    pos += fcs_bytes;
    uint64_t frame_start = o.out;
    ZState z;
    z.huf_valid = z.ll_valid = z.of_valid = z.ml_valid = 0;
    z.huf_bits = z.ll_log = z.of_log = z.ml_log = 0;
    z.rep[0] = 1;
    z.rep[1] = 4;
    z.rep[2] = 8;
    uint32_t last;
    do {
      if (pos + 3 > n) return 1;
      uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
      pos += 3;
      last = bh & 1;
      uint32_t bt = (bh >> 1) & 3, bs = bh >> 3;
      if (bt == 0) {
        if ((uint64_t)pos + bs > n || o.out + bs > cap) return 1;
        lz_literal(o, src + pos, bs, lane);
        pos += bs;
      } else if (bt == 1) {
        if (pos + 1 > n || o.out + bs > cap) return 1;
        lz_fill(o, src[pos], bs, lane);
        pos += 1;
      } else if (bt == 2) {
        if ((uint64_t)pos + bs > n || bs > 128 * 1024) return 1;
        if (z_block_dev(z, L, src + pos, bs, o, frame_start, cap, scratch, Z, lane PROF_ARG) < 0) return 1;
        pos += bs;
      } else {
        return 1;
      }
    } while (!last);
    if (fcs_bytes && o.out - frame_start != fcs) return 1;
    if (has_ck) {
      if (pos + 4 > n) return 1;
      pos += 4;
    }
  }
  lz_flush(o, lane);
  *out_len = (uint32_t)o.out;
  return 0;
}
