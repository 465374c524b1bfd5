// inflate_device.h -- raw DEFLATE (RFC 1951) for one wavefront.  Replaces flate2's DeflateDecoder
// at compression.rs:142-149.  The Huffman token stream is decoded wave-uniformly through a
// 10-bit fast table in LDS (longer codes fall back to canonical bit-by-bit decoding); literal
// bytes are stored by lane 0, matches are copied by all lanes.
#pragma once

#ifndef HUFF_FAST_BITS
#define HUFF_FAST_BITS 10  // (11: fewer trips through the long-code search, but 4 KiB more LDS per chunk -- lineitem / zlib SF 4 30.3 -> 32.7 ms)
#endif
#define HUFF_FAST_SIZE (1u << HUFF_FAST_BITS)
struct HuffTab {
  uint16_t fast[HUFF_FAST_SIZE];  // (len << 12) | symbol for codes of <= HUFF_FAST_BITS bits, 0 = long code
  uint16_t count[16];
  uint16_t symbol[320];
};

struct FseEnt {
  uint8_t sym, nb;
  uint16_t base;
};

struct DecompLds {
  union {
    struct {
      HuffTab lit, dist;
      uint8_t lens[320];
    } inf;
    struct {
      FseEnt ll[512], ml[512], of[256], wt[64];
      uint16_t huf[2048];      // sym | nb << 8
      uint8_t weights[256];
      int16_t norm[256];
      uint16_t next[256];
    } z;
  };
};

struct BitRd {
  const uint8_t* src;
  uint32_t n, pos;
  uint64_t bb;
  uint32_t bc;
  LzIn* in;  // compressed bytes staged in LDS (refills cost an LDS access, not a memory round trip)
};
__device__ __forceinline__ void br_refill(BitRd& b) {
  // bytes past the end read as zero; overrun is detected at block boundaries via br_overrun()
  if (b.pos < b.in->sb || b.pos + 8 > b.in->sb + LZ_STAGE + 16) lzin_stage(*b.in, b.pos, threadIdx.x & 63);
  uint64_t v;
  __builtin_memcpy(&v, b.in->stage + (b.pos - b.in->sb), 8);
  b.bb |= v << b.bc;
  uint32_t adv = (63 - b.bc) >> 3;
  b.pos += adv;
  b.bc += adv * 8;
}
__device__ __forceinline__ uint32_t br_get(BitRd& b, uint32_t k) {
  if (b.bc < k) br_refill(b);
  uint32_t v = (uint32_t)(b.bb & ((1ull << k) - 1));
  b.bb >>= k;
  b.bc -= k;
  return v;
}
__device__ __forceinline__ bool br_overrun(const BitRd& b) { return (uint64_t)b.pos * 8 - b.bc > (uint64_t)b.n * 8; }

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t len) { return __builtin_bitreverse32(v) >> (32 - len); }

// canonical Huffman from code lengths; returns <0 over-subscribed, 0 complete, >0 incomplete.  By the whole wavefront: lane l
// holds the lengths of symbols l, l + 64, ...; per length the symbols' count and every symbol's rank among them by ballots (the
// canonical code of a symbol = the first code of its length + its rank); every cell of the fast table looks its code up -- the
// length whose first `len` bits, reversed, fall into that length's range of codes.  (One lane used to do it all, its running
// offsets in a private array -- scratch memory --: half a millisecond per dynamic block, most of the DEFLATE stage's time.)
__device__ __forceinline__ int huff_build_dev(HuffTab& h, const uint8_t* lens, int n, uint32_t lane) {
  uint32_t l[5], rank[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int i = (int)lane + 64 * r;
    l[r] = i < n ? lens[i] : 0u;
  }
  uint32_t cnt[16], offs[17], first[16];
  uint32_t used = 0;
#pragma unroll
  for (int len = 1; len < 16; len++) {
    uint32_t seen = 0;
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const unsigned long long m = __ballot(l[r] == (uint32_t)len);
      if (l[r] == (uint32_t)len) rank[r] = seen + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
      seen += (uint32_t)__builtin_popcountll(m);
    }
    cnt[len] = seen;
    used += seen;
  }
  cnt[0] = (uint32_t)n - used;
  if (lane < 16) {
    uint32_t c = 0;
#pragma unroll
    for (int len = 0; len < 16; len++)
      if ((int)lane == len) c = cnt[len];
    h.count[lane] = (uint16_t)c;
  }
  int left = 0;
  bool bad = false;
  if (used != 0) {
    left = 1;
#pragma unroll
    for (int len = 1; len < 16; len++) {
      if (!bad) {
        left <<= 1;
        left -= (int)cnt[len];
        if (left < 0) bad = true;
      }
    }
  }
  if (bad || used == 0) {
    for (uint32_t i = lane; i < HUFF_FAST_SIZE; i += 64) h.fast[i] = 0;
    wave_sync();
    return bad ? -1 : 0;
  }
  offs[1] = 0;
  uint32_t code = 0;
#pragma unroll
  for (int len = 1; len < 16; len++) {
    offs[len + 1] = offs[len] + cnt[len];
    first[len] = code;
    code = (code + cnt[len]) << 1;
  }
#pragma unroll
  for (int r = 0; r < 5; r++) {
    if (l[r]) {
      uint32_t o = 0;
#pragma unroll
      for (int len = 1; len < 16; len++)
        if (l[r] == (uint32_t)len) o = offs[len];
      h.symbol[o + rank[r]] = (uint16_t)(lane + 64u * r);
    }
  }
  wave_sync();
  for (uint32_t j = lane; j < HUFF_FAST_SIZE; j += 64) {
    uint32_t e = 0;
#pragma unroll
    for (int len = 1; len <= HUFF_FAST_BITS; len++) {
      const uint32_t d = bitrev(j & ((1u << len) - 1), (uint32_t)len) - first[len];
      if (e == 0 && d < cnt[len]) e = ((uint32_t)len << 12) | h.symbol[offs[len] + d];
    }
    h.fast[j] = (uint16_t)e;
  }
  wave_sync();
  return left;
}

__device__ __forceinline__ int huff_decode_dev(BitRd& b, const HuffTab& h) {
  if (b.bc < 15) br_refill(b);
  uint32_t e = h.fast[b.bb & (HUFF_FAST_SIZE - 1)];
  if (e) {
    uint32_t l = e >> 12;
    b.bb >>= l;
    b.bc -= l;
    return (int)(e & 0xfff);
  }
  // long code: canonical decoding one bit at a time
  int code = 0, first = 0, index = 0;
  uint64_t bits = b.bb;
  for (int len = 1; len < 16; len++) {
    code |= (int)(bits & 1);
    bits >>= 1;
    int count = h.count[len];
    if (code - count < first) {
      b.bb >>= len;
      b.bc -= len;
      return h.symbol[index + (code - first)];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

__device__ const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ int inflate_codes_dev(BitRd& b, LzOut& o, uint32_t cap, const HuffTab& lc, const HuffTab& dc,
                                                  uint32_t lane) {
  for (;;) {
    // Literals, up to four per trip: four chained table lookups (a code of the fast table is at most 10
    // bits), one 4-byte store into the ring by lane 0, one round of bookkeeping.  Anything else -- a
    // long code, a length symbol, end of block -- leaves the trip to the one-symbol path below.
    if (b.bc < 4 * HUFF_FAST_BITS) br_refill(b);
    const uint32_t rpos = (uint32_t)o.out & o.rmask;
    if (b.bc >= 4 * HUFF_FAST_BITS && o.out + 4 <= cap && rpos + 4 <= o.rmask + 1) {
      uint32_t acc = 0, k = 0, used = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t e = lc.fast[(uint32_t)(b.bb >> used) & (HUFF_FAST_SIZE - 1)];
        if (e == 0 || (e & 0xfff) >= 256) break;
        acc |= (e & 0xff) << (8 * k);
        k++;
        used += e >> 12;
      }
      if (k) {
        if (lane == 0) __builtin_memcpy(o.ring + rpos, &acc, 4);  // bytes beyond k are overwritten by what follows
        b.bb >>= used;
        b.bc -= used;
        o.out += k;
        lz_maybe_flush(o, lane);
        if (br_overrun(b)) return 1;
        continue;
      }
    }
    int sym = huff_decode_dev(b, lc);
    if (sym < 0) return 1;
    if (sym < 256) {
      if (o.out >= cap) return 1;
      lz_byte(o, (uint32_t)sym, lane);
    } else if (sym == 256) {
      return br_overrun(b) ? 1 : 0;
    } else {
      sym -= 257;
      if (sym >= 29) return 1;
      uint32_t len = LBASE[sym] + br_get(b, LEXT[sym]);
      int ds = huff_decode_dev(b, dc);
      if (ds < 0 || ds >= 30) return 1;
      uint32_t dist = DBASE[ds] + br_get(b, DEXT[ds]);
      if (dist > o.out || o.out + len > cap) return 1;
      lz_match(o, dist, len, lane);
    }
    if (br_overrun(b)) return 1;
  }
}

__device__ __forceinline__ int inflate_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t lane, uint32_t* out_len,
                                             DecompLds& L, LzLds Z) {
  LzIn in{src, n, Z.stage, 0};
  lzin_stage(in, 0, lane);
  BitRd b{src, n, 0, 0, 0, &in};
  LzOut o{Z.ring, Z.rsize - 1, dst, 0, 0};
  uint32_t last;
  do {
    last = br_get(b, 1);
    uint32_t type = br_get(b, 2);
    if (br_overrun(b)) return 1;
    if (type == 0) {
      // stored: skip to the byte boundary; bits still buffered belong to the bytes before pos
      uint32_t drop = b.bc & 7;
      b.bb >>= drop;
      b.bc -= drop;
      uint32_t bytepos = b.pos - (b.bc >> 3);
      if (bytepos + 4 > n) return 1;
      uint32_t len = src[bytepos] | (src[bytepos + 1] << 8);
      uint32_t nlen = src[bytepos + 2] | (src[bytepos + 3] << 8);
      bytepos += 4;
      if ((len ^ 0xffffu) != nlen) return 1;
      if (bytepos + len > n || o.out + len > cap) return 1;
      lz_literal(o, src + bytepos, len, lane);
      b.pos = bytepos + len;
      b.bb = 0;
      b.bc = 0;
    } else if (type == 1) {
      for (uint32_t i = lane; i < 288; i += 64) L.inf.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
      wave_sync();
      huff_build_dev(L.inf.lit, L.inf.lens, 288, lane);
      for (uint32_t i = lane; i < 30; i += 64) L.inf.lens[i] = 5;
      wave_sync();
      huff_build_dev(L.inf.dist, L.inf.lens, 30, lane);
      if (inflate_codes_dev(b, o, cap, L.inf.lit, L.inf.dist, lane)) return 1;
    } else if (type == 2) {
      uint32_t nlen = br_get(b, 5) + 257, ndist = br_get(b, 5) + 1, ncode = br_get(b, 4) + 4;
      if (br_overrun(b) || nlen > 286 || ndist > 30) return 1;
      // code-length code lengths: read uniformly, lane 0 stores
      for (uint32_t i = lane; i < 19; i += 64) L.inf.lens[i] = 0;
      wave_sync();
      for (uint32_t i = 0; i < ncode; i++) {
        uint32_t v = br_get(b, 3);
        if (lane == 0) L.inf.lens[CLORDER[i]] = (uint8_t)v;
      }
      wave_sync();
      if (huff_build_dev(L.inf.lit, L.inf.lens, 19, lane) != 0) return 1;
      // literal/length + distance code lengths (decoded uniformly into registers-by-LDS)
      uint32_t i = 0;
      uint32_t prev = 0;
      int fail = 0;
      // lens[] is being rewritten while lit (the code-length code) is in use: lit only reads its own tables
      while (i < nlen + ndist) {
        int sym = huff_decode_dev(b, L.inf.lit);
        if (sym < 0) {
          fail = 1;
          break;
        }
        if (sym < 16) {
          if (lane == 0) L.inf.lens[i] = (uint8_t)sym;
          prev = (uint32_t)sym;
          i++;
        } else {
          uint32_t len = 0, rep;
          if (sym == 16) {
            if (i == 0) {
              fail = 1;
              break;
            }
            len = prev;
            rep = 3 + br_get(b, 2);
          } else if (sym == 17) {
            rep = 3 + br_get(b, 3);
          } else {
            rep = 11 + br_get(b, 7);
          }
          if (i + rep > nlen + ndist) {
            fail = 1;
            break;
          }
          for (uint32_t k = lane; k < rep; k += 64) L.inf.lens[i + k] = (uint8_t)len;
          prev = len;
          i += rep;
        }
      }
      if (fail || br_overrun(b)) return 1;
      wave_sync();
      if (L.inf.lens[256] == 0) return 1;
      // the distance lengths follow the literal/length lengths: build dist first from lens + nlen, then lit
      int r = huff_build_dev(L.inf.dist, L.inf.lens + nlen, (int)ndist, lane);
      if (r < 0 || (r > 0 && (int)ndist - (int)L.inf.dist.count[0] != 1)) return 1;
      r = huff_build_dev(L.inf.lit, L.inf.lens, (int)nlen, lane);
      if (r < 0 || (r > 0 && (int)nlen - (int)L.inf.lit.count[0] != 1)) return 1;
      if (inflate_codes_dev(b, o, cap, L.inf.lit, L.inf.dist, lane)) return 1;
    } else {
      return 1;
    }
  } while (!last);
  lz_flush(o, lane);
  *out_len = (uint32_t)o.out;
  return 0;
}
