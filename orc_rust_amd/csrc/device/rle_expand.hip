// rle_expand.hip -- pass 2 of the RLE decoders: cooperative run expansion, one wavefront per
// group of `group_size` consecutive stream blocks.
//
// Replaces read_short_repeat_values / read_direct_values / read_patched_base /
// read_delta_values (rle_v2/{short_repeat.rs:29-63, direct.rs:39-65, patched_base.rs:38-151,
// delta.rs:44-116}), RLE v1 read_run / read_literals (rle_v1.rs:90-132) and byte RLE
// (byte.rs:228-247), plus the copy-out of GenericRle::decode (rle.rs:68-107).
//
// Per iteration every lane that still has a run starting in ITS block parses that run's header
// (up to 64 runs at once).  Random-access sub-encodings (SHORT_REPEAT, DIRECT, fixed DELTA, v1
// runs, byte runs/literals) are then expanded value-parallel: a wave prefix sum over the run
// lengths maps 64 consecutive output values per step onto lanes, whatever the run lengths, and
// each lane bit-unpacks its value straight from the stream (HBM/L2, 8-byte unaligned load).
// Sub-encodings with a dependency inside the run (varying DELTA: wave prefix sum of the deltas;
// PATCHED_BASE: <=31-entry patch list resolved with a wave scan + LDS bitmap; v1 literal
// varints: ballot on terminator bytes) are expanded run by run by the whole wave.
// Values go to the dense output (one value per non-null row) with consecutive lanes writing
// consecutive elements.  Overflow/width/EOF errors are recorded as
// min(first value index of the failing run << 8 | code), which is exactly the batch in which
// the reference's decode_batch would have failed.
#include "rle_kernels.h"
#include "rle_parse.h"

#ifndef ORC_FAST512
#define ORC_FAST512 0
#endif
#ifndef ORC_EXPAND_WAVES
#define ORC_EXPAND_WAVES 4
#endif

struct WaveLds {
  uint32_t start[65];
  uint32_t cstart[65];  // narrow keys: first 8-value chunk of every run slot (see the chunk pass of expand_group)
  uint32_t meta[64];    // type | width << 8 | n << 16
  uint32_t meta2[64];   // pw | pl << 8 | cw << 16
  uint32_t oidx[64];    // output index of the run's first value
  uint64_t pay[64];     // stream offset of the packed payload
  uint64_t pay2[64];    // stream offset of the patch list
  int64_t base[64];
  int64_t delta[64];
  unsigned long long bitmap[8];
  int64_t tile[576];    // 512 values + 1 pad per 8: varying-DELTA transpose (lane-major -> value-major)
  uint64_t spos[64];    // run slots of this iteration: stream offset of the run header (~0 = empty)
  uint32_t soi[64];     // ... and the output index of its first value
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Inclusive scan over the 64 lanes with DPP adds (row shifts inside each row of 16 lanes, then the
// row totals broadcast onwards): six VALU instructions, no trip through the LDS crossbar that
// `__shfl_up` (ds_bpermute) would take six times in a dependent chain.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane) {
  (void)lane;
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
  return (uint32_t)x;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {  // the two halves moved by the same DPP pattern (0 where it has no source)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWS, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWS, 0xf, false);
  return lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, uint32_t lane) {
  (void)lane;
  v += dpp_u64<0x111, 0xf>(v);
  v += dpp_u64<0x112, 0xf>(v);
  v += dpp_u64<0x114, 0xf>(v);
  v += dpp_u64<0x118, 0xf>(v);
  v += dpp_u64<0x142, 0xa>(v);
  v += dpp_u64<0x143, 0xc>(v);
  return v;
}

// ob < 8 under a 64-bit decoder (dictionary keys, stored 1 / 2 / 4 bytes wide by the size of the dictionary): values that
// cannot be a key -- negative, or at / above the all-ones pattern of the key width -- are stored as all ones; no key of that
// size is valid (DictionaryArray::try_new), and the consumer reports them like any other key out of bounds
template <bool NARROW = false>
__device__ __forceinline__ void store_val(void* out, uint32_t ob, uint64_t i, int64_t v) {
  if (ob == 8) ((int64_t*)out)[i] = v;
  else if (ob == 4) ((int32_t*)out)[i] = NARROW && (v < 0 || v > 0x7fffffffll) ? -1 : (int32_t)v;
  else if (ob == 2) ((int16_t*)out)[i] = NARROW && (v < 0 || v > 0xfffell) ? (int16_t)-1 : (int16_t)v;
  else ((int8_t*)out)[i] = NARROW && (v < 0 || v > 0xfell) ? (int8_t)-1 : (int8_t)v;
}

__device__ __forceinline__ bool in_range_n(int64_t v, int nbits) { return trunc_n(v, nbits) == v; }
// a + b == r (wrapping): did the signed addition overflow?
__device__ __forceinline__ bool add_ovf(int64_t a, int64_t b, int64_t r) { return ((a ^ r) & (b ^ r)) < 0; }
// a - b == r (wrapping): did the signed subtraction overflow?
__device__ __forceinline__ bool sub_ovf(int64_t a, int64_t b, int64_t r) { return ((a ^ b) & (a ^ r)) < 0; }

// The reference stops at the first failing run.  Here every wavefront reports what it meets, speculative runs behind
// a failure included, and two minima pick the first one out: it has the smallest value index (err) -- though runs
// behind a run that failed to parse share that index, since such a run yields no values -- and the smallest stream
// position (err_pos), which settles the error kind.  `pos`: the run's first byte for a run that fails to parse, the
// last byte of its header (payload - 1: a fixed DELTA run's payload is already the next run) for one that fails later.
__device__ __forceinline__ void report(RleJob* j, uint64_t needed, uint64_t oi, uint32_t code, uint64_t pos) {
  if (oi < needed) {
    atomicMin(&j->err, ((unsigned long long)oi << 8) | code);
    atomicMin(&j->err_pos, ((unsigned long long)pos << 8) | code);
  }
}

// One value of a random-access run (SHORT_REPEAT, DIRECT, fixed DELTA, v1 run, byte run / literal).
// `ld8(byte offset)`: the 8 stream bytes at that offset from the run's payload, as loaded (little endian).
template <typename LD8>
__device__ __forceinline__ int64_t decode_b1(uint32_t type, uint32_t w, int64_t base, int64_t dlt, LD8 ld8, uint32_t idx,
                                             bool is_signed, int nbits, bool& bad) {
  int64_t v;
  if (type == RT_SR || type == RT_B_RUN) {
    v = base;
  } else if (type == RT_DIRECT) {
    const uint64_t bit = (uint64_t)idx * w;
    uint64_t u = (__builtin_bswap64(ld8((uint32_t)(bit >> 3))) << (bit & 7)) >> (64 - w);  // unpack_be
    v = is_signed ? zigzag_n(u, nbits) : trunc_n((int64_t)u, nbits);
  } else if (type == RT_B_LIT) {
    v = (int8_t)(ld8(idx) & 0xff);
  } else if (type == RT_DELTA) {  // fixed delta (delta.rs:84-93)
    bool add = dlt > 0;
    int64_t mag = dlt < 0 ? (int64_t)(0 - (uint64_t)dlt) : dlt;
    uint64_t step = (uint64_t)idx * (uint64_t)mag;
    v = add ? (int64_t)((uint64_t)base + step) : (int64_t)((uint64_t)base - step);
    if (idx) {
      int64_t prev = add ? (int64_t)((uint64_t)v - (uint64_t)mag) : (int64_t)((uint64_t)v + (uint64_t)mag);
      bad |= (add ? add_ovf(prev, mag, v) : sub_ovf(prev, mag, v)) || !in_range_n(v, nbits);
    }
  } else {  // RT_V1_RUN (rle_v1.rs:102-132): checked add/sub of |delta| in N
    v = (int64_t)((uint64_t)base + (uint64_t)((int64_t)idx * dlt));
    if (idx) {
      int64_t prev = (int64_t)((uint64_t)v - (uint64_t)dlt);
      bad |= add_ovf(prev, dlt, v) || !in_range_n(v, nbits);
    }
  }
  return v;
}

// 16 payload bytes for each of a lane's four value pairs (values i0 + 128u + 2*lane, +1) of a DIRECT run
__device__ __forceinline__ void direct_pair_load(const uint8_t* pp, uint32_t i0, uint32_t w, uint32_t lane, uint64_t (*pf)[2]) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    uint64_t bit = (uint64_t)(i0 + u * 128 + 2 * lane) * w;
    __builtin_memcpy(pf[u], pp + (bit >> 3), 16);
  }
}

template <int CODEC, int OB, int NB = OB * 8>
__device__ __forceinline__ void expand_group(RleJob* j, const RleBlocks& blk, const uint64_t* scalars, uint32_t lg, WaveLds& L,
                                             uint32_t lane PROF_PARM) {
  const uint8_t* data = as_global(j->data);
  void* out = as_global(j->out);
  const uint64_t len = scalars[j->len_idx];
  const uint64_t needed = scalars[j->needed_idx] + j->skip;
  const bool is_signed = j->is_signed;
  constexpr int nbits = NB;       // the reference's NInt width: the Arrow value width, but for dictionary keys (64-bit decoder, 32-bit keys)
  constexpr bool narrow = NB != OB * 8;
  constexpr uint32_t ob = OB;
  const uint32_t G = j->group_size;
  const uint32_t eof_code = CODEC == CODEC_BYTE ? ORC_E_IO : ORC_E_OUT_OF_SPEC;

  auto ld8_at = [&](uint64_t off) -> uint64_t { return ld_u64(data + off); };  // 8 stream bytes at `off`
  PROF_MARK(8);
  uint32_t lb = lg * G + lane;
  bool active = false, tail_owner = false;
  bool clean = true;  // no failing run seen by this lane
  uint64_t pos = 0, end = 0;
  uint64_t oi = 0;
  if (lane < G && lb < j->nblocks && ((uint64_t)lb * RLE_BLK < len || lb == 0)) {
    uint32_t b = j->block0 + lb;
    uint32_t e = blk.entry[b];
    uint64_t bend = (uint64_t)(lb + 1) * RLE_BLK;
    end = bend < len ? bend : len;
    pos = (uint64_t)lb * RLE_BLK + e;
    oi = (uint64_t)blk.tile_base[b / RLE_TILE] + blk.voff[b];
    active = e < RLE_BLK && pos < end && oi < needed;
    tail_owner = (lb == 0 && len == 0) || (e < RLE_BLK && pos < end);
  }

  // Lane l < G owns block l of the group and walks its run chain; per iteration it hands out up to
  // K = 64 / G consecutive runs into "slots" l*K .. l*K+K-1, then every lane parses ONE slot in full.
  // G = 64: one run per block per iteration; G = 1 (streams that expand a lot, e.g. fixed DELTA):
  // 64 consecutive runs of the single block are expanded together.
  const uint32_t K = 64u / G;
  PROF_MARK(10);
  for (;;) {
    L.spos[lane] = ~0ull;
    wave_sync();
    if (active && K == 1) {
      // one run per block and trip: the owner parses its own slot in full below (one header fetch, not two)
      L.spos[lane] = pos;
      L.soi[lane] = (uint32_t)oi;
    } else if (active) {
      uint64_t p = pos, o = oi;
      for (uint32_t k = 0; k < K && p < end && o < needed; k++) {
        L.spos[lane * K + k] = p;
        L.soi[lane * K + k] = (uint32_t)o;
        if (CODEC == CODEC_RLE2) {
          // SHORT_REPEAT / DIRECT from the header's two bytes (rle2_hop2); whatever else there is takes the full parse
          // (Round 6, all measured on lineitem's dictionary keys -- runs of 3 - 4 values -- and all without effect: the group's bytes
          // staged in LDS, a 32-byte register window over the headers, groups of 16 or 64 blocks.  The kernel is bound by what it
          // ISSUES: a hop of the chain costs its ~30 instructions for the whole wavefront, however few lanes own a block.)
          const uint32_t hw = (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8);  // (the second byte may be the first of the stream's slack: ORC_PAD)
          uint32_t fsz, fn;
          if (rle2_hop2(hw & 0xff, hw >> 8, nbits, len - p, fsz, fn)) {
            p += fsz;
            o += fn;
            continue;
          }
        }
        RunHdr hh;
        run_parse<CODEC, false>(data + p, len - p, is_signed, nbits, hh);
        if (hh.err) clean = false;
        p += hh.size;
        o += hh.n;
      }
      pos = p;
      oi = o;
      active = clean && pos < end && oi < needed;
    }
    wave_sync();
    PROF_MARK(11);
    const uint64_t sp = L.spos[lane];
    const bool has = sp != ~0ull;
    if (!__ballot(has)) break;
    RunHdr h;
    h.n = 0;
    h.size = 0;
    h.err = 0;
    h.type = 0;
    bool is_b2 = false;
    uint32_t cnt = 0, chunks = 0;
    if (has) {
      const uint64_t soi = L.soi[lane];
      run_parse<CODEC, true>(data + sp, len - sp, is_signed, nbits, h);
      if (K == 1) {
        if (h.err) clean = false;
        pos = sp + h.size;
        oi = soi + h.n;
        active = clean && pos < end && oi < needed;
      }
      if (h.err) {
        report(j, needed, soi, h.err, sp);
      } else {
        is_b2 = (h.type == RT_DELTA && h.width != 0) || h.type == RT_PATCHED || h.type == RT_V1_LIT;
        cnt = is_b2 ? 0 : h.n;
        // narrow dictionary keys: bit-packed runs of at most 8 bits a value and repeats leave in chunks of 8 values (below)
        if (narrow && OB <= 2 && CODEC == CODEC_RLE2 && ((h.type == RT_DIRECT && h.width <= 8) || h.type == RT_SR)) {
          chunks = (h.n + 7) >> 3;
          cnt = 0;
        }
      }
      L.meta[lane] = h.type | (h.width << 8) | (h.n << 16);
      L.oidx[lane] = (uint32_t)soi;
      L.pay[lane] = sp + h.payload;
      L.base[lane] = h.base;
      L.delta[lane] = h.delta;
      if (CODEC == CODEC_RLE2 && h.type == RT_PATCHED) {
        L.meta2[lane] = h.pw | (h.pl << 8) | (h.cw << 16);
        L.pay2[lane] = sp + h.patch_off;
      }
    }
    uint32_t incl = wave_incl_scan_u32(cnt, lane);
    L.start[lane] = incl - cnt;
    uint32_t T = __shfl(incl, 63);
    if (lane == 63) L.start[64] = T;
    wave_sync();
    PROF_MARK(12);

    // ---- narrow keys (dictionary keys stored 1 or 2 bytes wide): the memory instructions are what a short run costs, so a
    // lane takes EIGHT consecutive values of a run -- one 8-byte load (8 x w <= 64 bits; w = 8 is byte aligned, narrower widths
    // leave room for the up to 7 bits in front of the first value), one 8- or 16-byte store -- and 64 such chunks of any mix
    // of runs leave per step ------------------------------------------------------------------------------------------------
    if (narrow && OB <= 2 && CODEC == CODEC_RLE2) {
      const uint32_t cincl = wave_incl_scan_u32(chunks, lane);
      L.cstart[lane] = cincl - chunks;
      const uint32_t CT = __shfl(cincl, 63);
      if (lane == 63) L.cstart[64] = CT;
      wave_sync();
      for (uint32_t c0 = 0; c0 < CT; c0 += 64) {
        const uint32_t q = c0 + lane;
        if (q >= CT) continue;
        uint32_t l2 = 0, h2 = 64;
        while (h2 - l2 > 1) {
          const uint32_t mid = (l2 + h2) >> 1;
          if (L.cstart[mid] <= q) l2 = mid;
          else h2 = mid;
        }
        const uint32_t r = l2, m = L.meta[r];
        const uint32_t w = (m >> 8) & 0xff, n = m >> 16, i0 = 8 * (q - L.cstart[r]);
        const uint32_t nv = n - i0 < 8 ? n - i0 : 8;
        constexpr uint64_t none = OB == 1 ? 0xffull : 0xffffull;  // "not a key" (store_val)
        uint64_t lo = 0, hi = 0;
        if ((m & 0xff) == RT_SR) {
          const int64_t b = L.base[r];
          const uint64_t v = b < 0 || (uint64_t)b >= none ? none : (uint64_t)b;
          lo = v * (OB == 1 ? 0x0101010101010101ull : 0x0001000100010001ull);
          hi = lo;
        } else {
          const uint64_t bit = (uint64_t)i0 * w;
          uint64_t raw = __builtin_bswap64(ld8_at(L.pay[r] + (bit >> 3))) << (bit & 7);  // MSB-first bit stream (integer/util.rs:44-218)
#pragma unroll
          for (int k = 0; k < 8; k++) {
            uint64_t v = raw >> (64 - w);  // unsigned, 1 <= w <= 8
            raw <<= w;
            if (v >= none) v = none;
            if (OB == 1) lo |= v << (8 * k);
            else if (k < 4) lo |= v << (16 * k);
            else hi |= v << (16 * (k - 4));
          }
        }
        const uint64_t oo = (uint64_t)L.oidx[r] + i0;
        uint8_t* o = (uint8_t*)out + oo * OB;
        if (nv == 8 && oo + 8 <= needed) {
          if (OB == 1) {
            __builtin_memcpy(o, &lo, 8);
          } else {
            uint64_t pr[2] = {lo, hi};
            __builtin_memcpy(o, pr, 16);
          }
        } else {
          for (uint32_t k = 0; k < nv && oo + k < needed; k++) {
            const uint64_t v = OB == 1 ? (lo >> (8 * k)) & 0xff : ((k < 4 ? lo >> (16 * k) : hi >> (16 * (k - 4))) & 0xffff);
            if (OB == 1) o[k] = (uint8_t)v;
            else ((uint16_t*)o)[k] = (uint16_t)v;
          }
        }
      }
    }

    // ---- B1: value-parallel expansion of the random-access runs -----------------------------
    {
      uint32_t cur = 0;  // run that contains q0 (wave uniform, only moves forward)
      for (uint32_t q0 = 0; q0 < T;) {
        while (L.start[cur + 1] <= q0) cur++;
        const uint32_t rend = L.start[cur + 1];
        if (ORC_FAST512 && rend - q0 >= 512) {
          // fast path: 512 values of ONE run, 8 per lane, run parameters wave-uniform
          const uint32_t m = L.meta[cur];
          const uint32_t type = m & 0xff, w = (m >> 8) & 0xff;
          const int64_t base = L.base[cur], dlt = L.delta[cur];
          const uint64_t o0 = L.oidx[cur];
          const uint8_t* pp = data + L.pay[cur];
          const uint32_t i0 = q0 - L.start[cur];
          bool bad = false;
          int64_t v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = decode_b1(type, w, base, dlt, [&](uint32_t o) { return ld_u64(pp + o); }, i0 + u * 64 + lane, is_signed, nbits, bad);
#pragma unroll
          for (int u = 0; u < 8; u++) {
            uint64_t oo = o0 + i0 + u * 64 + lane;
            if (oo < needed) store_val<narrow>(out, ob, oo, v[u]);
          }
          if (bad) report(j, needed, o0, ORC_E_OUT_OF_SPEC, L.pay[cur] - 1);
          q0 += 512;
          continue;
        }
        if (OB == 8 && rend - q0 >= 512 && (uint64_t)L.oidx[cur] + (q0 - L.start[cur]) + 512 <= needed) {
          // 512 int64 values of ONE run: every lane produces PAIRS of consecutive values, so one
          // 16-byte load feeds two values and one 16-byte store writes them (1 KiB per wave store)
          const uint32_t m = L.meta[cur];
          const uint32_t type = m & 0xff, w = (m >> 8) & 0xff;
          const int64_t base = L.base[cur], dlt = L.delta[cur];
          const uint64_t o0 = L.oidx[cur];
          const uint8_t* pp = data + L.pay[cur];
          const uint32_t i0 = q0 - L.start[cur];
          bool bad = false;
          int64_t va[4], vb[4];
          if (type == RT_DIRECT) {
            uint64_t pf[4][2];
            direct_pair_load(pp, i0, w, lane, pf);
#pragma unroll
            for (int u = 0; u < 4; u++) {
              uint32_t idx = i0 + u * 128 + 2 * lane;
              uint64_t bit = (uint64_t)idx * w;
              uint64_t hi = __builtin_bswap64(pf[u][0]), lo = __builtin_bswap64(pf[u][1]);
              uint32_t sh = bit & 7, t = sh + w;
              uint64_t a = (hi << sh) >> (64 - w);
              uint64_t top = t < 64 ? ((hi << t) | (lo >> (64 - t))) : (lo << (t - 64));
              uint64_t b = top >> (64 - w);
              va[u] = is_signed ? zigzag_n(a, 64) : (int64_t)a;
              vb[u] = is_signed ? zigzag_n(b, 64) : (int64_t)b;
            }
          } else if (type == RT_DELTA) {
            // fixed delta: an arithmetic progression.  The reference checks every step; the sequence
            // is monotonic, so the last value of this 512-value segment decides (exact, in 128 bit).
            const int64_t step = dlt;  // sign of delta_base picks add / subtract (delta.rs:77-82): +d or -|d| = d either way
            const uint64_t first = (uint64_t)base + (uint64_t)(i0 + 2 * lane) * (uint64_t)step;
#pragma unroll
            for (int u = 0; u < 4; u++) {
              va[u] = (int64_t)(first + (uint64_t)(u * 128) * (uint64_t)step);
              vb[u] = (int64_t)((uint64_t)va[u] + (uint64_t)step);
            }
            __int128 last = (__int128)base + (__int128)(i0 + 511) * (__int128)step;
            bad = last > (__int128)INT64_MAX || last < (__int128)INT64_MIN;
          } else if (type == RT_SR || type == RT_B_RUN) {
#pragma unroll
            for (int u = 0; u < 4; u++) va[u] = vb[u] = base;
          } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
              uint32_t idx = i0 + u * 128 + 2 * lane;
              va[u] = decode_b1(type, w, base, dlt, [&](uint32_t o) { return ld_u64(pp + o); }, idx, is_signed, nbits, bad);
              vb[u] = decode_b1(type, w, base, dlt, [&](uint32_t o) { return ld_u64(pp + o); }, idx + 1, is_signed, nbits, bad);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            int64_t pr[2] = {va[u], vb[u]};
            __builtin_memcpy((int64_t*)out + o0 + i0 + u * 128 + 2 * lane, pr, 16);
          }
          if (bad) report(j, needed, o0, ORC_E_OUT_OF_SPEC, L.pay[cur] - 1);
          q0 += 512;
          continue;
        }
        if (OB == 4 && !narrow && rend - q0 >= 512 && (uint64_t)L.oidx[cur] + (q0 - L.start[cur]) + 512 <= needed) {
          // 512 int32 values of ONE run (dates, decimal scales, 32-bit integers): every lane produces FOUR consecutive values per
          // step -- 24 payload bytes in, one 16-byte store out, 1 KiB per wave store -- instead of four 4-byte stores 256 bytes apart
          const uint32_t m = L.meta[cur];
          const uint32_t type = m & 0xff, w = (m >> 8) & 0xff;
          const int64_t base = L.base[cur], dlt = L.delta[cur];
          const uint64_t o0 = L.oidx[cur];
          const uint8_t* pp = data + L.pay[cur];
          const uint32_t i0 = q0 - L.start[cur];
          bool bad = false;
          int32_t v4[2][4];
          if (type == RT_DIRECT) {
            uint64_t pf[2][3];
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const uint64_t bit = (uint64_t)(i0 + u * 256 + 4 * lane) * w;
              const uint8_t* q = pp + (bit >> 3);
              pf[u][0] = ld_u64(q);
              pf[u][1] = ld_u64(q + 8);
              pf[u][2] = ld_u64(q + 16);  // (4 x 32 bits behind up to 7: 17 bytes at most; the rest is slack -- ORC_PAD)
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const uint64_t hi = __builtin_bswap64(pf[u][0]), mid = __builtin_bswap64(pf[u][1]), lo = __builtin_bswap64(pf[u][2]);
              const uint32_t sh = (uint32_t)(((uint64_t)(i0 + u * 256 + 4 * lane) * w) & 7);
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const uint32_t pos = sh + k * w;  // < 7 + 96 + 1
                const uint64_t a = pos < 64 ? hi : mid, b = pos < 64 ? mid : lo;
                const uint32_t r = pos & 63;
                const uint64_t x = r ? (a << r) | (b >> (64 - r)) : a;
                const uint64_t uval = x >> (64 - w);
                v4[u][k] = (int32_t)(is_signed ? zigzag_n(uval, nbits) : trunc_n((int64_t)uval, nbits));
              }
            }
          } else if (type == RT_DELTA) {
            // fixed delta: an arithmetic progression, monotonic -- the segment's last value decides whether a step left N's range
            const int64_t step = dlt;
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
              for (int k = 0; k < 4; k++) v4[u][k] = (int32_t)((uint64_t)base + (uint64_t)(i0 + u * 256 + 4 * lane + k) * (uint64_t)step);
            const __int128 last = (__int128)base + (__int128)(i0 + 511) * (__int128)step;
            bad = last > (__int128)INT32_MAX || last < (__int128)INT32_MIN;
          } else if (type == RT_SR || type == RT_B_RUN) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
              for (int k = 0; k < 4; k++) v4[u][k] = (int32_t)base;
          } else {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
              for (int k = 0; k < 4; k++)
                v4[u][k] = (int32_t)decode_b1(type, w, base, dlt, [&](uint32_t o) { return ld_u64(pp + o); }, i0 + u * 256 + 4 * lane + k, is_signed, nbits, bad);
          }
#pragma unroll
          for (int u = 0; u < 2; u++) __builtin_memcpy((int32_t*)out + o0 + i0 + u * 256 + 4 * lane, v4[u], 16);
          if (bad) report(j, needed, o0, ORC_E_OUT_OF_SPEC, L.pay[cur] - 1);
          q0 += 512;
          continue;
        }
        if (rend - q0 >= 256) {
          // 256 values of ONE run, 4 per lane
          const uint32_t m = L.meta[cur];
          const uint32_t type = m & 0xff, w = (m >> 8) & 0xff;
          const int64_t base = L.base[cur], dlt = L.delta[cur];
          const uint64_t o0 = L.oidx[cur];
          const uint8_t* pp = data + L.pay[cur];
          const uint32_t i0 = q0 - L.start[cur];
          bool bad = false;
          int64_t v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = decode_b1(type, w, base, dlt, [&](uint32_t o) { return ld_u64(pp + o); }, i0 + u * 64 + lane, is_signed, nbits, bad);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            uint64_t oo = o0 + i0 + u * 64 + lane;
            if (oo < needed) store_val<narrow>(out, ob, oo, v[u]);
          }
          if (bad) report(j, needed, o0, ORC_E_OUT_OF_SPEC, L.pay[cur] - 1);
          q0 += 256;
          continue;
        }
        // short runs: 256 values per step, lane l takes values q0 + l, + 64, + 128, + 192 -- each finds its run by bisection over
        // the run starts, then all four loads are issued together (a step costs one memory round trip, whatever it carries)
        uint32_t rr[4], ix[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t q = q0 + u * 64 + lane;
          in[u] = q < T;
          uint32_t r = cur;
          if (in[u] && q >= L.start[r + 1]) {
            uint32_t l2 = r, h2 = 64;
            while (h2 - l2 > 1) {
              uint32_t mid = (l2 + h2) >> 1;
              if (L.start[mid] <= q) l2 = mid;
              else h2 = mid;
            }
            r = l2;
          }
          rr[u] = r;
          ix[u] = in[u] ? q - L.start[r] : 0;
        }
        int64_t vv[4];
        bool badv[4] = {false, false, false, false};
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t m = L.meta[rr[u]];
          vv[u] = in[u] ? decode_b1(m & 0xff, (m >> 8) & 0xff, L.base[rr[u]], L.delta[rr[u]], [&](uint32_t o) { return ld8_at(L.pay[rr[u]] + o); }, ix[u], is_signed, nbits, badv[u]) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (!in[u]) continue;
          const uint64_t o0 = L.oidx[rr[u]];
          if (badv[u]) report(j, needed, o0, ORC_E_OUT_OF_SPEC, L.pay[rr[u]] - 1);
          const uint64_t oo = o0 + ix[u];
          if (oo < needed) store_val<narrow>(out, ob, oo, vv[u]);
        }
        q0 += 256;
      }
    }

    PROF_MARK(13);
    // ---- B2: runs with an internal dependency, one run at a time, whole wave ------------------
    unsigned long long m2 = __ballot(has && is_b2 && !h.err);
    // varying-DELTA runs with deltas of <= 8 bits: a lane's 8 deltas are `w` contiguous bytes; the
    // load for the NEXT run is issued before the current run is scanned and stored
    uint64_t raw_next = 0;
    auto delta_prefetch = [&](uint32_t rr) -> uint64_t {
      const uint32_t mt = L.meta[rr];
      const uint32_t ww = (mt >> 8) & 0xff, nn = mt >> 16;
      if (CODEC != CODEC_RLE2 || (mt & 0xff) != RT_DELTA || ww > 8 || lane * 8 >= nn - 2) return 0;
      return ld_u64(data + L.pay[rr] + lane * ww);
    };
    if (CODEC == CODEC_RLE2 && m2) raw_next = delta_prefetch((uint32_t)__builtin_ctzll(m2));
    while (m2) {
      uint32_t r = (uint32_t)__builtin_ctzll(m2);
      m2 &= m2 - 1;
      uint32_t mt = L.meta[r];
      uint32_t type = mt & 0xff, w = (mt >> 8) & 0xff, n = mt >> 16;
      int64_t base = L.base[r];
      uint64_t o0 = L.oidx[r];
      const uint8_t* pp = data + L.pay[r];
      bool bad = false;
      const uint64_t raw_cur = raw_next;
      if (CODEC == CODEC_RLE2 && m2) raw_next = delta_prefetch((uint32_t)__builtin_ctzll(m2));
      if (CODEC == CODEC_RLE2 && type == RT_DELTA) {
        // varying delta (delta.rs:94-113)
        int64_t db = L.delta[r];
        bool add = db > 0;
        int64_t mag = db < 0 ? (int64_t)(0 - (uint64_t)db) : db;
        int64_t v1 = add ? (int64_t)((uint64_t)base + (uint64_t)mag) : (int64_t)((uint64_t)base - (uint64_t)mag);
        bad = (add ? add_ovf(base, mag, v1) : sub_ovf(base, mag, v1)) || !in_range_n(v1, nbits);
        if (lane == 0 && o0 < needed) store_val<narrow>(out, ob, o0, base);
        if (lane == 1 && o0 + 1 < needed) store_val<narrow>(out, ob, o0 + 1, v1);
        // n - 2 <= 510 packed deltas: lane l owns deltas 8l .. 8l+7 (for 8-bit deltas that is one
        // 8-byte load), sums them locally, ONE wave scan over the lane totals gives every prefix,
        // and an LDS transpose turns the lane-major results into coalesced stores.
        const uint32_t nd = n - 2;
        // only the running sums are kept (a delta is the difference of two neighbours): registers
        uint64_t pre[8];
        uint64_t run = 0;
        if (w <= 8) {
          const uint64_t be = __builtin_bswap64(raw_cur);  // MSB-first bit stream (integer/util.rs:44-218)
#pragma unroll
          for (int k = 0; k < 8; k++) {
            uint64_t dk = w == 8 ? ((raw_cur >> (8 * k)) & 0xff) : ((be << (k * w)) >> (64 - w));
            run += lane * 8 + k < nd ? dk : 0;
            pre[k] = run;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) {
            uint32_t i = lane * 8 + k;
            run += i < nd ? unpack_be(pp, i, w) : 0;
            pre[k] = run;
          }
        }
        // narrow deltas: the 510 of them sum to less than 2^32, the cheaper 32-bit scan will do
        uint64_t incl = w <= 16 ? (uint64_t)wave_incl_scan_u32((uint32_t)run, lane) : wave_incl_scan_u64(run, lane);
        uint64_t excl = incl - run;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          uint32_t i = lane * 8 + k;
          uint64_t sfx = excl + pre[k];
          int64_t dk = (int64_t)(pre[k] - (k ? pre[k - 1] : 0));
          int64_t v = add ? (int64_t)((uint64_t)v1 + sfx) : (int64_t)((uint64_t)v1 - sfx);
          int64_t prev = add ? (int64_t)((uint64_t)v - (uint64_t)dk) : (int64_t)((uint64_t)v + (uint64_t)dk);
          if (i < nd) bad |= (add ? add_ovf(prev, dk, v) : sub_ovf(prev, dk, v)) || !in_range_n(v, nbits);
          L.tile[i + (i >> 3)] = v;
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; k++) {
          uint32_t i = k * 64 + lane;
          if (i < nd) {
            uint64_t oo = o0 + 2 + i;
            if (oo < needed) store_val<narrow>(out, ob, oo, L.tile[i + (i >> 3)]);
          }
        }
        wave_sync();
      } else if (CODEC == CODEC_RLE2 && type == RT_PATCHED) {
        uint32_t m2v = L.meta2[r];
        uint32_t pw = m2v & 0xff, pl = (m2v >> 8) & 0xff, cw = m2v >> 16;
        const uint8_t* plist = data + L.pay2[r];
        if (lane < 8) L.bitmap[lane] = 0;
        wave_sync();
        // patch list -> positions (patched_base.rs:93-143)
        uint64_t e = lane < pl ? unpack_be(plist, lane, cw) : 0;
        uint64_t gap = pw >= 64 ? 0 : (e >> pw);
        uint64_t patch = pw >= 64 ? e : (e & ((1ull << pw) - 1));
        bool cont = lane < pl && gap == 255 && patch == 0;
        bool real = lane < pl && !cont;
        uint64_t ppos = wave_incl_scan_u64(lane < pl ? gap : 0, lane);
        // position of the previous real entry
        unsigned long long realm = __ballot(real);
        unsigned long long below = realm & ((1ull << lane) - 1);
        int prev_real = below ? 63 - __builtin_clzll(below) : -1;
        uint64_t prev_pos = __shfl(ppos, prev_real < 0 ? 0 : prev_real);
        bool stuck = real && prev_real >= 0 && ppos == prev_pos;  // next gap 0: never matches again
        bool beyond = real && ppos >= n;                           // not reached inside this run
        unsigned long long stopm = __ballot(stuck || beyond);
        uint32_t first_stop = stopm ? (uint32_t)__builtin_ctzll(stopm) : 64;
        bool applied = real && lane < first_stop;
        // a trailing continuation entry that the reference walks into indexes past the list (panic)
        if (pl > 0) {
          bool last_cont = __shfl((int)cont, pl - 1);
          unsigned long long real_before = realm;  // all real entries precede a trailing continuation chain
          int last_real = real_before ? 63 - __builtin_clzll(real_before) : -1;
          if (last_cont && (last_real < 0 || (uint32_t)last_real < first_stop)) bad = true;
        }
        if (w >= 64 && __ballot(applied)) bad = true;  // checked_shl(value_bit_width)
        if (applied) atomicOr(&L.bitmap[ppos >> 6], 1ull << (ppos & 63));
        wave_sync();
        for (uint32_t c = 0; c < n; c += 64) {
          uint32_t i = c + lane;
          if (i < n) {
            bool patched = (L.bitmap[i >> 6] >> (i & 63)) & 1;
            if (!patched) {
              int64_t u = trunc_n((int64_t)unpack_be(pp, i, w), nbits);
              int64_t v = (int64_t)((uint64_t)u + (uint64_t)base);
              bad |= add_ovf(u, base, v) || !in_range_n(v, nbits);  // checked_add in N
              uint64_t oo = o0 + i;
              if (oo < needed) store_val<narrow>(out, ob, oo, v);
            }
          }
        }
        if (applied && w < 64) {
          int64_t u = trunc_n((int64_t)unpack_be(pp, (uint32_t)ppos, w), nbits);
          int64_t pbits = trunc_n((int64_t)(patch << w), nbits);
          int64_t v = trunc_n((int64_t)((uint64_t)(u | pbits) + (uint64_t)base), nbits);  // wrapping_add in N
          uint64_t oo = o0 + ppos;
          if (oo < needed) store_val<narrow>(out, ob, oo, v);
        }
        wave_sync();
      } else if (CODEC == CODEC_RLE1 && type == RT_V1_LIT) {
        // n literal varints (rle_v1.rs:90-100): ballot on terminator bytes
        uint64_t p0 = L.pay[r];
        uint64_t vstart = p0;  // start of the varint that straddles into this 64-byte step
        uint32_t done = 0;
        for (uint64_t c = p0; done < n && c < len; c += 64) {
          uint64_t bp = c + lane;
          bool term = bp < len && !(data[bp] & 0x80);
          unsigned long long tm = __ballot(term);
          unsigned long long below = tm & ((1ull << lane) - 1);
          uint32_t rank = done + (uint32_t)__builtin_popcountll(below);
          if (term && rank < n) {
            uint64_t s = below ? c + (63 - __builtin_clzll(below)) + 1 : vstart;
            uint64_t u = 0;
            uint32_t e = 0;
            varint_n(data + s, len - s, nbits, &u, &e);
            int64_t v = is_signed ? zigzag_n(u, nbits) : trunc_n((int64_t)u, nbits);
            uint64_t oo = o0 + rank;
            if (oo < needed) store_val<narrow>(out, ob, oo, v);
          }
          if (tm) vstart = c + (63 - __builtin_clzll(tm)) + 1;
          done += (uint32_t)__builtin_popcountll(tm);
        }
      }
      if (__ballot(bad)) {
        if (lane == 0) report(j, needed, o0, ORC_E_OUT_OF_SPEC, L.pay[r] - 1);
      }
    }
    wave_sync();
    PROF_MARK(14);
  }
  // clean end of stream before `needed` values: "not enough values to decode"
  if (tail_owner && clean && pos >= len) report(j, needed, oi, eof_code | ORC_E_EOF, len);
}

template <int CODEC>
__device__ __forceinline__ void expand_entry(RleJob* jobs, const uint32_t* group_job, RleBlocks blk, const uint64_t* scalars, uint32_t total_groups) {
  __shared__ WaveLds lds[4];
  uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t g = blockIdx.x * 4 + wv;
  if (g >= total_groups) return;
  PROF_BEGIN();
  RleJob* j = &jobs[group_job[g]];
  uint32_t lg = g - j->group0;
  if (lg >= j->ngroups) return;
  if (j->uniform_idx && scalars[j->uniform_idx]) return;  // (one value throughout: rle2_uniform_kernel; nobody reads the job's output)
  PROF_MARK(9);
  if (CODEC == CODEC_BYTE) {
    expand_group<CODEC, 1>(j, blk, scalars, lg, lds[wv], lane PROF_ARG);
  } else {
    // wave-uniform dispatch on the value width: the bodies are specialised at compile time
    switch (j->out_bytes) {
      case 8: expand_group<CODEC, 8>(j, blk, scalars, lg, lds[wv], lane PROF_ARG); break;
      case 4:
        if (j->nbits == 64) expand_group<CODEC, 4, 64>(j, blk, scalars, lg, lds[wv], lane PROF_ARG);
        else expand_group<CODEC, 4>(j, blk, scalars, lg, lds[wv], lane PROF_ARG);
        break;
      case 2:
        if (j->nbits == 64) expand_group<CODEC, 2, 64>(j, blk, scalars, lg, lds[wv], lane PROF_ARG);
        else expand_group<CODEC, 2>(j, blk, scalars, lg, lds[wv], lane PROF_ARG);
        break;
      default: expand_group<CODEC, 1, 64>(j, blk, scalars, lg, lds[wv], lane PROF_ARG); break;  // (one-byte dictionary keys)
    }
  }
  PROF_END();
}

// The scale of every value of a Decimal column is the column's scale in every file a writer made (decimal.rs:28-52 still reads it value
// by value): its SECONDARY stream then is one run over and over -- DELTA, fixed delta 0, 512 values: C1 FF zz 00 -- and a shorter last
// one (the same with a smaller count, or a SHORT_REPEAT of one byte).  Expanding it writes 4 bytes per value that the varint decoder
// reads back only to find the scale it was given (lineitem: 1.2 GB each way per step, 0.7 ms of expansion).  One workgroup per such job
// compares the stream with that pattern and counts its values: a stream that IS the pattern and holds the values the column needs
// sets the job's flag; anything else leaves the job to the expansion, which stays the authority on every other stream and every error.
extern "C" __global__ void __launch_bounds__(256) rle2_uniform_kernel(const RleJob* jobs, uint32_t n_jobs, uint64_t* scalars) {
  if (blockIdx.x >= n_jobs) return;
  const RleJob& j = jobs[blockIdx.x];
  if (!j.uniform_idx || j.codec != CODEC_RLE2 || !j.is_signed || j.skip || j.uniform_value >= 64) return;
  __shared__ uint32_t differs;
  if (threadIdx.x == 0) differs = 0;
  __syncthreads();
  const uint8_t* d = j.data;
  const uint64_t len = scalars[j.len_idx], needed = scalars[j.needed_idx];
  const uint32_t zz = j.uniform_value << 1;  // (zigzag of a non-negative value below 64: one varint byte)
  const uint32_t pat = 0xC1u | 0xFFu << 8 | zz << 16;
  const uint64_t nw = len / 4;
  bool bad = false;
  for (uint64_t i = threadIdx.x; i + 1 < nw; i += 256) bad = bad || ld_u32(d + 4 * i) != pat;
  if (bad) differs = 1;
  __syncthreads();
  if (threadIdx.x == 0 && !differs && len >= 2) {
    uint64_t total = 0;
    bool ok = true;
    if (nw) {
      const uint32_t w = ld_u32(d + 4 * (nw - 1));
      // DELTA, width code 0 (fixed delta), any count; base zz, delta 0
      if ((w & 0xFEu) == 0xC0u && ((w >> 16) & 0xffu) == zz && (w >> 24) == 0) total = 512 * (nw - 1) + ((((w & 1u) << 8) | ((w >> 8) & 0xffu)) + 1u);
      else ok = false;
    }
    const uint32_t rem = (uint32_t)(len & 3);
    if (rem == 2) {
      // SHORT_REPEAT of one byte: 00 | width - 1 = 0 | count - 3
      if ((d[len - 2] & 0xF8u) == 0 && d[len - 1] == zz) total += (d[len - 2] & 7u) + 3u;
      else ok = false;
    } else if (rem) ok = false;
    if (ok && total >= needed) scalars[j.uniform_idx] = 1;
  }
}

extern "C" __global__ void __launch_bounds__(256, ORC_EXPAND_WAVES) rle2_expand_kernel(RleJob* jobs, const uint32_t* group_job, RleBlocks blk, const uint64_t* scalars,
                                                                      uint32_t total_groups) {
  expand_entry<CODEC_RLE2>(jobs, group_job, blk, scalars, total_groups);
}
extern "C" __global__ void __launch_bounds__(256) rle1_expand_kernel(RleJob* jobs, const uint32_t* group_job, RleBlocks blk, const uint64_t* scalars,
                                                                      uint32_t total_groups) {
  expand_entry<CODEC_RLE1>(jobs, group_job, blk, scalars, total_groups);
}
extern "C" __global__ void __launch_bounds__(256) byte_expand_kernel(RleJob* jobs, const uint32_t* group_job, RleBlocks blk, const uint64_t* scalars,
                                                                      uint32_t total_groups) {
  expand_entry<CODEC_BYTE>(jobs, group_job, blk, scalars, total_groups);
}
