// zstd_entropy.h -- Zstandard (RFC 8878) entropy stage: ONE WAVEFRONT PER COMPRESSED BLOCK.
//
// Replaces the entropy half of zstd::Decoder (compression.rs:151-159).  The frame and block headers of
// every chunk are parsed on the host while the stripe is staged (orcgpu_zstd_host.inc): each
// Compressed_Block becomes a ZBlock that says where its literals section, its sequences section and
// the table descriptions it depends on live (Treeless literals and Repeat_Mode tables point at the
// block of the same frame that defined them, so no block waits for another one here).
//
// Per block this kernel
//   1. builds the Huffman table and decodes the literal streams into the block's literal buffer
//      (16 lanes per stream, self-synchronising restarts: huf_decode_w16 in zstd_device.h);
//   2. builds the three FSE tables as 8-byte entries {next state, state bits, extra bits, base value};
//   3. runs the FSE state machine -- the one serial chain of the format -- with the six bit fields of a
//      sequence spread over six lanes (offset / match-length / literal-length extra bits, then the three
//      state updates): ONE LDS lookup, a 3-step DPP prefix sum of the field widths, and one shift pair
//      per lane extract all fields at once from a wave-uniform 64-bit window.  The bit stream itself is
//      held in a vector register (lane j = dword j of the current 256-byte segment), so the window is
//      rebuilt with three v_readlane per sequence and no memory access sits on the chain.
// It writes offset value, match length and literal length per sequence (three arrays) and one status word per block (0 = ok, else a
// diagnostic code: any nonzero value rejects the chunk); repeat offsets are resolved, and
// the LZ77 copies executed, by lz_exec_kernel (lz_exec.h), one workgroup per chunk.
#pragma once

struct ZBlock {
  const uint8_t* src;    // chunk payload in the staged arena
  uint8_t* lit_out;      // decoded literals (lit_type >= 2): lit_regen bytes (+ 8 bytes of slack)
  uint32_t* seq_out;     // three arrays of ((nseq + 3) & ~3) entries, one behind the other: offset values, match lengths, literal lengths
  uint32_t chunk;        // index into the chunk table
  uint32_t content_off, content_end;  // block content inside the payload
  uint32_t lit_type, lit_streams, lit_hdr, lit_regen, lit_comp;
  uint32_t nseq, seq_off;    // seq_off: offset of the byte behind Number_of_Sequences (the modes byte)
  uint32_t huf_off, huf_end; // Huffman tree description: own, or (Treeless) the defining block's; end of that literals section
  uint32_t tab_off[3];   // per table LL, OF, ML: offset of the modes byte of the block that defines it (own: seq_off)
  uint32_t tab_end[3];   // ... and the end of that block
  uint32_t pad[2];
};

struct ZSeqHdr {      // written by zstd_entropy_kernel (tables mode) per block with sequences
  uint32_t bit_off;   // where the sequences' bit stream starts in the chunk payload
  uint32_t logs;      // table logs: LL | OF << 8 | ML << 16
  uint32_t status;    // nonzero: a table description is broken (the code zstd_entropy_kernel would report)
  uint32_t pad;
};
// A sequence as zstd_seq_quads_kernel leaves it for the execution kernel: 8 bytes {offset value : 29, match length : 18, literal
// length : 17}.  Match lengths end at 65 539 + 65 535, literal lengths at 65 536 + 65 535 (RFC 8878 3.1.1.3.2.1.1); an offset
// value of 2^29 or more cannot be met by any chunk (a chunk header holds 23 bits of length, a plain chunk is bounded by the
// block size): it is kept as 2^29 - 1, which the execution kernel rejects like every offset that reaches before the frame.
__device__ __forceinline__ uint2 zseq_pack(uint32_t ofv, uint32_t ml, uint32_t ll) {
  const uint32_t o = ofv < 0x1fffffffu ? ofv : 0x1fffffffu;
  return make_uint2(o | ml << 29, ml >> 3 | ll << 15);
}
__device__ __forceinline__ void zseq_unpack(uint2 p, uint32_t& ofv, uint32_t& ml, uint32_t& ll) {
  ofv = p.x & 0x1fffffffu;
  ml = p.x >> 29 | (p.y & 0x7fffu) << 3;
  ll = p.y >> 15;
}
#define ZL_CELLS 1280u  // cells per block: LL [0, 512), ML [512, 1024), OF [1024, 1280)
#define ZL_LL 0u
#define ZL_ML 512u
#define ZL_OF 1024u

struct ZFse {  // one decoding table cell
  uint16_t next;   // base of the next state
  uint8_t nb;      // bits to read for the next state
  uint8_t add;     // extra bits of the symbol's value
  uint32_t base;   // base value of the symbol (offset: 1 << code)
};

struct ZEntLds {
  union {
    struct {  // literals phase
      uint16_t huf[2048];  // sym | nb << 8
      uint8_t weights[256];
      FseEnt wt[64];
    } h;
    struct {  // literals phase, behind the Huffman table and its description: every lane's piece of its stream (HWin, zstd_device.h)
      __attribute__((aligned(16))) uint8_t lit_pad[4096 + 256 + 64 * 8];
      __attribute__((aligned(16))) uint8_t chunk[64 * ZL_CHUNK];
    };
    struct {  // sequences phase
      ZFse ll[512], ml[512], of[256];
      ZFse zero;           // all-zero cell for the lanes without a field
      union {
        struct {
          uint32_t seqbuf[64 * 3 + 64];  // 64 decoded sequences waiting for their coalesced store (+ a sink for the idle lanes)
          __attribute__((aligned(8))) uint8_t bits[2048 + 16];  // the piece of the bit stream the state machine is working in
        };
        __attribute__((aligned(16))) uint16_t ctab[1280];  // tables mode (zstd_lanes.h): the block's tables as 2-byte cells {symbol : 6, state number : 10}
      };
    } s;
  };
  int16_t norm[256];
  uint16_t next[256];
  uint8_t sym[512];        // symbol of every cell while a table is being built
  __attribute__((aligned(4))) uint8_t stage[208];  // a table description on its way from memory to the parser (ZL_STAGE + the jump table of a Treeless block)
};

// FSE decoding table with the symbol's extra bits and base value folded into every cell.  which: 0 LL, 1 OF, 2 ML.
// Lane s works for symbol s (at most 53 symbols).  The serial construction of RFC 8878 4.1.1 -- spread the symbols over the
// table with a fixed stride, skipping the cells taken from the top by the "less than one" probabilities, then number the
// cells of every symbol in ascending order -- is done by the whole wavefront:
//   * the stride visits every cell once (it is odd): visit t goes to cell u(t) = t * step mod size.  With `high` the last
//     cell below the low-probability ones, the j-th PLACED symbol lands in the j-th visit whose cell is <= high: a running
//     count of such visits (ballot + popcount, 64 visits per turn) gives every visit its j, `expand[j]` its symbol;
//   * cells are numbered 64 at a time in ascending order: for every distinct symbol among the 64 (a loop over ballots) a
//     cell's number is the symbol's running count (kept in the symbol's lane) plus the cells of that symbol below it.
// A table took ~80 us of a wavefront the serial way (tens of thousands of blocks per call: milliseconds), now a few.
__device__ __forceinline__ int zfse_build(ZFse* t, uint8_t* symtab, const int16_t* norm, int nsym, int log, uint16_t* next, int which, uint32_t lane,
                                          uint16_t* ct = nullptr) {
  const uint32_t size = 1u << log;
  const uint32_t step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  const int cnt = (int)lane < nsym ? (int)norm[lane] : 0;
  const bool low = cnt == -1;
  const unsigned long long lowm = __ballot(low);
  const unsigned long long below = (1ull << lane) - 1;
  const uint32_t nlow = (uint32_t)__popcll(lowm);
  const uint32_t c = cnt > 0 ? (uint32_t)cnt : 0u;
  const uint32_t incl = wave_incl_scan_u32(c, lane);
  if ((uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + nlow != size) return 1;  // (the serial walk would not end on cell 0)
  const uint32_t high = size - 1 - nlow;
  uint8_t* expand = reinterpret_cast<uint8_t*>(next);  // 512 bytes: the symbol of the j-th placed cell
  if (low) symtab[size - 1 - (uint32_t)__popcll(lowm & below)] = (uint8_t)lane;
  for (uint32_t k = incl - c; k < incl; k++) expand[k] = (uint8_t)lane;
  wave_sync();
  uint32_t placed = 0;
  for (uint32_t t0 = 0; t0 < size; t0 += 64) {
    const uint32_t tt = t0 + lane;
    const uint32_t u = (tt * step) & mask;
    const bool f = tt < size && u <= high;
    const unsigned long long m = __ballot(f);
    if (f) symtab[u] = expand[placed + (uint32_t)__popcll(m & below)];
    placed += (uint32_t)__popcll(m);
  }
  wave_sync();
  // this lane's symbol: what its cells carry besides the state
  const uint32_t my_add = which == 0 ? (lane < 36 ? Z_LL_BITS[lane] : 0u) : (which == 1 ? lane : (lane < 53 ? Z_ML_BITS[lane] : 0u));
  const uint32_t my_base = which == 0 ? (lane < 36 ? Z_LL_BASE[lane] : 0u) : (which == 1 ? 1u << (lane & 31u) : (lane < 53 ? Z_ML_BASE[lane] : 0u));
  uint32_t nxt = low ? 1u : c;  // next state number of symbol `lane`
  for (uint32_t i0 = 0; i0 < size; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool valid = i < size;
    const uint32_t s = valid ? (uint32_t)symtab[i] : 0xffu;
    uint32_t ns = 0, add = 0, base = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int l0 = __builtin_ctzll(todo);
      const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)s, l0);
      const unsigned long long m = __ballot(valid && s == s0);
      const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)s0);
      const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)my_add, (int)s0);
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)my_base, (int)s0);
      if (valid && s == s0) {
        ns = first + (uint32_t)__popcll(m & below);
        add = a0;
        base = b0;
      }
      if (lane == s0) nxt += (uint32_t)__popcll(m);
      todo &= ~m;
    }
    if (valid) {
      const int nb = log - z_hibit(ns | (ns == 0));
      ZFse e;
      e.next = (uint16_t)((ns << nb) - size);
      e.nb = (uint8_t)nb;
      e.add = (uint8_t)add;
      e.base = base;
      t[i] = e;
      if (ct) ct[i] = (uint16_t)(s | ns << 6);
    }
  }
  wave_sync();
  return 0;
}

// One table description (Predefined / RLE / FSE_Compressed) at p: builds it when `build`, returns the bytes it takes or -1.
__device__ __forceinline__ long zfse_table(ZEntLds& L, ZFse* t, int* log_out, int mode, const uint8_t* p, uint32_t n, int which, bool build,
                                            uint32_t lane, uint16_t* ct = nullptr) {
  const int16_t* def = which == 0 ? Z_LL_DEF : (which == 1 ? Z_OF_DEF : Z_ML_DEF);
  const int defn = which == 0 ? 36 : (which == 1 ? 29 : 53), deflog = which == 1 ? 5 : 6;
  const int maxsym = which == 0 ? 36 : (which == 1 ? 32 : 53), maxlog = which == 1 ? 8 : 9;
  if (mode == 0) {
    if (build) {
      for (int i = (int)lane; i < defn; i += 64) L.norm[i] = def[i];
      wave_sync();
      if (zfse_build(t, L.sym, L.norm, defn, deflog, L.next, which, lane, ct)) return -1;
      *log_out = deflog;
    }
    return 0;
  }
  if (mode == 1) {
    if (n < 1) return -1;
    if (build) {
      const uint32_t s = p[0];
      if (s >= (uint32_t)maxsym) return -1;  // the state machine meets the symbol at once: code out of range
      if (lane == 0) {
        ZFse e;
        e.next = 0;
        e.nb = 0;
        e.add = which == 0 ? Z_LL_BITS[s] : (which == 1 ? (uint8_t)s : Z_ML_BITS[s]);
        e.base = which == 0 ? Z_LL_BASE[s] : (which == 1 ? 1u << s : Z_ML_BASE[s]);
        t[0] = e;
        if (ct) ct[0] = (uint16_t)(s | 1u << 6);
      }
      wave_sync();
      *log_out = 0;
    }
    return 1;
  }
  if (mode == 2) {
    int nsym = maxsym, log;
    const long c = fse_read_ncount_dev(p, n, L.norm, &nsym, &log, maxlog, lane, L.stage);
    if (c < 0) return -1;
    if (build) {
      if (zfse_build(t, L.sym, L.norm, nsym, log, L.next, which, lane, ct)) return -1;
      *log_out = log;
    }
    return c;
  }
  return -1;  // Repeat_Mode never reaches here: the host resolved it to the defining block
}

// The table `which` (0 LL, 1 OF, 2 ML) as described in the block whose modes byte sits at src[moff] (block ends at src[mend]).
// Returns the bytes the description takes (what the OWN block's parse advances by), or -1.
__device__ __forceinline__ long zfse_from_block(ZEntLds& L, ZFse* t, int* log_out, const uint8_t* src, uint32_t moff, uint32_t mend, int which,
                                                 uint32_t lane, uint16_t* ct = nullptr) {
  if (moff >= mend) return -1;
  const uint32_t modes = src[moff];
  uint32_t p = moff + 1;
  for (int w = 0; w <= which; w++) {
    const int mode = (modes >> (6 - 2 * w)) & 3;
    if (w == which) return zfse_table(L, t, log_out, mode, src + p, mend - p, w, true, lane, ct);
    if (mode == 3) continue;  // a repeated table takes no bytes
    const long c = zfse_table(L, t, log_out, mode, src + p, mend - p, w, false, lane);
    if (c < 0) return -1;
    p += (uint32_t)c;
  }
  return -1;
}

// wave-uniform values: tell the compiler (scalar registers, scalar branches)
__device__ __forceinline__ uint32_t zuni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int zuni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Huffman tree description at q (qn bytes available) -> L.h.huf; returns the bytes it takes or -1.  Its first min(qn, ZL_STAGE) bytes
// are in L.stage already (zstd_literals_job: one round trip for the header byte, the description and the jump table behind it).
#define ZL_STAGE 192u
// bytes a description takes, from its header byte (RFC 8878 4.2.1.1)
__device__ __forceinline__ uint32_t zhuf_desc_bytes(uint32_t hb) { return hb >= 128 ? 1u + (hb - 127u + 1u) / 2u : 1u + hb; }
template <class LDS>
__device__ __forceinline__ long zhuf_tree(LDS& L, const uint8_t* q, uint32_t qn, int* maxbits, uint32_t lane PROF_PARM) {
  if (qn < 1) return -1;
  int nw;
  const uint32_t hb = L.stage[0];
  uint32_t used;
  if (hb >= 128) {
    nw = (int)hb - 127;
    const uint32_t nbytes = (uint32_t)(nw + 1) / 2;
    if (1 + nbytes > qn) return -1;
    for (int i = (int)lane; i < nw; i += 64) L.h.weights[i] = (i & 1) ? (L.stage[1 + i / 2] & 15) : (L.stage[1 + i / 2] >> 4);  // (65 bytes at most: staged)
    used = 1 + nbytes;
    wave_sync();
  } else {
    if (1 + hb > qn) return -1;
    int nsym = 256, log;
    const long c = fse_read_ncount_dev(q + 1, hb, L.norm, &nsym, &log, 6, lane, L.stage + 1, ZL_STAGE - 1);
    PROF_MARK(8);
    if (c < 0) return -1;
    if (fse_build_dev(L.h.wt, L.norm, nsym, log, L.next, lane)) return -1;
    PROF_MARK(9);
    // The weights are ONE chain: two FSE states take turns on a backward bit stream, 100 - 250 steps a table.  With the cells in LDS
    // and the stream read through a window in LDS every step was two dependent LDS reads and ~60 instructions (63 us a block, a
    // tenth of the literals kernel, -DORC_PROF).  The table has 64 cells at most and the stream 127 bytes: lane i keeps cell i, lane j
    // dword j of the stream, a step reads both with v_readlane (the state and the bit position are wave-uniform: scalar registers)
    // and files the weight in the lane of its number -- no memory access in the loop.
    const uint32_t sn = hb - (uint32_t)c;
    const uint8_t* sb = L.stage + 1 + c;
    if (sn == 0 || sb[sn - 1] == 0) return -1;
    const uint32_t cellv = lane < (1u << log) ? *reinterpret_cast<const uint32_t*>(&L.h.wt[lane]) : 0u;  // sym | nb << 8 | base << 16
    uint32_t dwv = 0;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++)
      if (4 * lane + t < sn) dwv |= (uint32_t)sb[4 * lane + t] << (8 * t);
    int P = (int)zuni((sn - 1) * 8 + (uint32_t)(31 - __builtin_clz((uint32_t)sb[sn - 1])));  // unread bits (RBits::bits)
    auto cell = [&](uint32_t st) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)cellv, (int)zuni(st)); };
    auto rd = [&](uint32_t nb) -> uint32_t {  // rb_read: the nb bits below bit P; bits before the stream read as zero
      if (nb == 0) return 0u;
      const int start = P - (int)nb;
      uint32_t v = 0;
      if (start >= 0) {
        const uint32_t lo = (uint32_t)start >> 5;
        const uint64_t two = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dwv, (int)lo) |
                             (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dwv, (int)(lo + 1)) << 32;  // (lane 32 at most: zero there)
        v = (uint32_t)(two >> (start & 31)) & ((1u << nb) - 1u);
      } else if (P > 0) {
        const uint32_t have = (uint32_t)__builtin_amdgcn_readlane((int)dwv, 0) & ((1u << P) - 1u);  // (P < nb <= 6)
        v = (have << (-start)) & ((1u << nb) - 1u);
      }
      P = start;
      return v;
    };
    uint32_t s1 = rd((uint32_t)log), s2 = rd((uint32_t)log);
    uint32_t wsv[4] = {0, 0, 0, 0};  // weight number r * 64 + lane
    auto put = [&](int at, uint32_t w) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        if ((at >> 6) == r) wsv[r] = (int)lane == (at & 63) ? w : wsv[r];
    };
    nw = 0;
    int fail = 0;
    for (;;) {
      if (nw >= 254) {
        fail = 1;
        break;
      }
      const uint32_t c1 = cell(s1);
      put(nw, c1 & 0xffu);
      nw++;
      if (P < (int)((c1 >> 8) & 0xffu)) {
        put(nw, cell(s2) & 0xffu);
        nw++;
        break;
      }
      s1 = (c1 >> 16) + rd((c1 >> 8) & 0xffu);
      if (nw >= 254) {
        fail = 1;
        break;
      }
      const uint32_t c2 = cell(s2);
      put(nw, c2 & 0xffu);
      nw++;
      if (P < (int)((c2 >> 8) & 0xffu)) {
        put(nw, cell(s1) & 0xffu);
        nw++;
        break;
      }
      s2 = (c2 >> 16) + rd((c2 >> 8) & 0xffu);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) L.h.weights[r * 64 + (int)lane] = (uint8_t)wsv[r];
    if (fail) return -1;
    used = 1 + hb;
    wave_sync();
    PROF_MARK(10);
  }
  if (huf_build_dev(L.h.huf, maxbits, L.h.weights, nw, lane, reinterpret_cast<uint8_t*>(L.norm))) return -1;  // (norm: free once the weights are decoded)
  return (long)used;
}

// ---- the FSE state machine --------------------------------------------------------------------------------
// Lane roles inside every group of 8 lanes (only lanes 0..7 matter; the others run along harmlessly):
//   0 offset extra bits   1 match-length extra bits   2 literal-length extra bits   3, 4 nothing
//   5 literal-length state bits   6 match-length state bits   7 offset state bits
// = the order in which a sequence's fields follow each other in the backward bit stream (RFC 8878 3.1.1.3.2.1.1),
// so an inclusive prefix sum of the field widths over the 8 lanes gives every lane the position of its field.
// Lane k and lane 7-k share a table (row_half_mirror hands the new state from the state lane to the value lane).
#define ZDPP(x, ctrl) __builtin_amdgcn_update_dpp(0, (int)(x), ctrl, 0xf, 0xf, false)

// Decodes nseq sequences of one block.  Returns 0, or a nonzero diagnostic code.
// The backward bit stream q[0 .. qn) is read from its last set bit downwards.  It is held in registers: lane j of `cur`
// has dword (top - 63 + j) counted from `base` (q rounded down to 4 bytes), `nxt` the segment 48 dwords further down.
__device__ __forceinline__ int zfse_sequences(ZEntLds& L, const uint8_t* q_, uint32_t qn_, uint32_t nseq_, int ll_log_, int of_log_, int ml_log_,
                                              uint32_t* seq_out_, uint32_t* dump_, uint32_t* progress, uint32_t lane) {
  const uint8_t* q = q_;      // (pointers stay what as_global() made them: global address space, vector registers)
  const uint32_t qn = zuni(qn_), nseq = zuni(nseq_);
  const int ll_log = zuni(ll_log_), of_log = zuni(of_log_), ml_log = zuni(ml_log_);
  uint32_t* seq_out = seq_out_;
  uint32_t* dump = dump_;
  if (qn == 0) return 21;
  const uint32_t lastb = zuni((uint32_t)q[qn - 1]);
  if (lastb == 0) return 21;
  const uint32_t qo = zuni((uint32_t)(reinterpret_cast<uintptr_t>(q) & 3));
  const uint8_t* base = q - qo;
  const int nbytes = (int)(qn + qo);           // bytes from base to the end of the stream
  const int hb = 31 - __builtin_clz(lastb);
  int P = (int)(qn - 1) * 8 + hb;              // unread bits; bit b of the stream is bit g0 + b counted from base
  const int g0 = 8 * (int)qo;
  int top = (g0 + P) >> 5;                     // dword held by lane 63 of `cur`
  auto seg_load = [&](int t) -> uint32_t {
    const int d = t - 63 + (int)lane;
    // dwords below the stream (d < 0) or wholly behind its end are never looked at: zero
    return (d >= 0 && d * 4 < nbytes + 4) ? *reinterpret_cast<const uint32_t*>(base + (long)d * 4) : 0u;
  };
  // (no load stays in flight across iterations: vmcnt counts loads and stores alike on this target, so waiting for a
  // prefetched segment would also wait for every store issued since)
  uint32_t cur = seg_load(top);

  const uint32_t r = lane < 8 ? lane : 3;  // lanes 8..63 have no field: they look at the all-zero cell like lanes 3 and 4
  const bool is_x = r < 3;
  // table of this lane's role and where its field width sits in the cell's first word
  ZFse* tb = (r == 0 || r == 7) ? L.s.of : ((r == 1 || r == 6) ? L.s.ml : ((r == 2 || r == 5) ? L.s.ll : &L.s.zero));
  const uint32_t sh = is_x ? 24u : 16u;
  const bool has_tab = r != 3 && r != 4;

  // 64 unread bits below position p (wave uniform), left aligned (bit 63 = stream bit p - 1); bits below the stream read as zero
  auto window = [&](int p) -> uint64_t {
    if (p <= 0) return 0;
    const int gt = g0 + p - 1;            // top unread bit, counted from base
    int rel = (gt >> 5) - (top - 63);     // lane of `cur` that holds it
    if (rel < 3) {                        // keep three dwords (rel, rel-1, rel-2) inside the segment: next 61 dwords down
      top -= 61;
      rel += 61;
      cur = seg_load(top);
    }
    const uint32_t t = (uint32_t)(gt & 31) + 1;  // 1..32 bits of the top dword are unread
    const uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane((int)cur, rel);
    const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)cur, rel - 1);
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)cur, rel - 2);
    const uint64_t hi = (((uint64_t)d2 << 32) | d1) >> t;  // low 32 bits: window bits 63..32
    const uint64_t lo = (((uint64_t)d1 << 32) | d0) >> t;
    uint64_t w = (hi << 32) | (lo & 0xffffffffu);
    if (p < 64) w &= ~0ull << (64 - p);  // nothing below bit 0 of the stream
    return w;
  };

  // initial states: LL, OF, ML (RFC 8878 3.1.1.3.2.1.1)
  uint32_t state;
  {
    uint64_t w = window(P);
    const uint32_t sl = ll_log ? (uint32_t)(w >> (64 - ll_log)) : 0;
    w <<= ll_log;
    const uint32_t so = of_log ? (uint32_t)(w >> (64 - of_log)) : 0;
    w <<= of_log;
    const uint32_t sm = ml_log ? (uint32_t)(w >> (64 - ml_log)) : 0;
    P -= ll_log + of_log + ml_log;
    state = (r == 0 || r == 7) ? so : ((r == 1 || r == 6) ? sm : ((r == 2 || r == 5) ? sl : 0));
  }
  if (P < 0) return 22;

  // lanes 0..2 file {offset value, match length, literal length} in LDS; every 64 sequences leave with three coalesced stores
  (void)dump;
  uint32_t* const sb = L.s.seqbuf;
  const uint32_t slot0 = lane < 3 ? lane : 192u + lane;
  const uint32_t slot_step = lane < 3 ? 3u : 0u;
  // The execution kernel runs beside this one and takes the sequences as they come (lz_exec.h): they are stored write-through
  // (agent scope, nothing stays dirty in this XCD's L2) and the count of sequences that have LANDED is published one flush
  // late -- by then the stores of the flush before have long been acknowledged, so the wait costs nothing.
  uint32_t published = 0;
  const uint32_t np = (nseq + 3u) & ~3u;
  auto flush = [&](uint32_t first, uint32_t count) {  // sequences [first, first + count) are in seqbuf
    wave_sync();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (first > published) {
      if (lane == 0) __hip_atomic_store(progress, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      published = first;
    }
    // (three arrays of `np` entries: offset values, match lengths, literal lengths)
    if (lane < count) {
      uint32_t* o = seq_out + first + lane;
      __hip_atomic_store(o, sb[3 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + np, sb[3 * lane + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 2 * np, sb[3 * lane + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_sync();
  };
  const uint32_t tbase = (uint32_t)(uintptr_t)tb;  // LDS byte address of this lane's table
  uint32_t i = 0;
  // ---- groups of 64 sequences far from the end of the stream: no bit below the stream can be touched (a sequence takes
  // at most 89 bits), none of them is the block's last sequence.  The bit stream is staged in LDS 2 KiB at a time; per
  // sequence ONE 8-byte word that holds the next 57..64 unread bits is fetched together with the table cells (one LDS
  // latency), and every lane cuts its field out of it. ----
  if (nseq > 64 && P >= 64 * 89 + 128) {
    int T = P;                    // unread bits (stream bit T - 1 is the next one)
    int origin = 0x7fffffff;      // stream byte held by bits[0] (nothing staged yet)
    const uint32_t bits_lds = (uint32_t)(uintptr_t)L.s.bits;
    while (nseq - i > 64 && T >= 64 * 89 + 128) {
      // the stage must hold the words of the next 64 sequences: bytes [(T - 64 * 89 - 57) / 8, T / 8]
      if ((T - 64 * 89 - 64) >> 3 < origin) {
        wave_sync();
        origin = ((T - 57) >> 3) + 8 - 2048;
        if (origin < 0) origin = 0;
        for (uint32_t o = lane * 8; o < 2048 + 8; o += 512) {
          const uint64_t v = ld_u64(q + origin + o);  // (at most 15 bytes behind the stream: slack of the staged arena)
          __builtin_memcpy(L.s.bits + o, &v, 8);
        }
        wave_sync();
      }
      uint32_t slot = slot0;
      uint32_t k = 0;
      bool wide = false;  // a sequence of more than 57 bits came up: it is decoded by the careful loop below
      for (; k < 64; k++) {
        const int B = (T - 57) >> 3;       // the word q[B .. B + 8) holds stream bits [8B, 8B + 64): at least 57 unread ones
        const uint32_t waddr = bits_lds + (uint32_t)(B - origin);
        const uint32_t addr = tbase + (has_tab ? state << 3 : 0u);
        const uint64_t e = *reinterpret_cast<const __attribute__((address_space(3))) uint64_t*>((uintptr_t)addr);
        uint64_t word = *reinterpret_cast<const __attribute__((address_space(3))) uint64_t*>((uintptr_t)waddr);
        uint32_t e0 = (uint32_t)e, e1 = (uint32_t)(e >> 32);
        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(word));  // both reads are in flight together: ONE LDS latency per sequence
        const uint32_t cnt = (e0 >> sh) & 0xffu;
        // inclusive prefix sum of the field widths over lanes 0..7: fused DPP adds (two wait states between a VALU write
        // and a DPP read of the same register)
        uint32_t incl = cnt;
        asm volatile(
            "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 0"
            : "+v"(incl));
        const int total = __builtin_amdgcn_readlane((int)incl, 7);
        if (__builtin_expect(total > 57, 0)) {
          wide = true;
          break;
        }
        // the field's lowest bit is stream bit T - incl, i.e. bit (T - 8B) - incl of the word
        const uint32_t x = (uint32_t)(word >> ((uint32_t)(T - 8 * B) - incl));
        const uint32_t val = x & ~(~0u << cnt);  // cnt <= 31
        const uint32_t nv = (is_x ? e1 : (e0 & 0xffffu)) + val;
        sb[slot] = nv;
        slot += slot_step;
        const uint32_t mirrored = (uint32_t)ZDPP(nv, 0x141);  // row_half_mirror: lane k <- lane 7 - k
        state = is_x ? mirrored : nv;
        T -= total;
      }
      if (k) flush(i, k);
      i += k;
      if (wide) break;
    }
    P = T;
    top = (g0 + P) >> 5;
    cur = seg_load(top);
  }
  // ---- the rest, with every check ----
  uint32_t slot = slot0;
  const uint32_t first_rest = i;
  for (; i < nseq; i++) {
    const bool last = i + 1 == nseq;
    const uint64_t w = window(P);
    const ZFse* cell = tb + (has_tab ? state : 0);
    const uint32_t e0 = reinterpret_cast<const uint32_t*>(cell)[0];
    const uint32_t e1 = reinterpret_cast<const uint32_t*>(cell)[1];
    uint32_t cnt = (e0 >> sh) & 0xffu;
    if (last && !is_x) cnt = 0;  // no state update behind the last sequence
    uint32_t incl = cnt;
    incl += (uint32_t)ZDPP(incl, 0x111);  // row_shr:1
    incl += (uint32_t)ZDPP(incl, 0x112);  // row_shr:2
    incl += (uint32_t)ZDPP(incl, 0x114);  // row_shr:4
    const int total = __builtin_amdgcn_readlane((int)incl, 7);
    uint32_t val;
    if (total <= 64) {
      const uint64_t x = w << (incl - cnt);
      val = cnt ? (uint32_t)(x >> (64 - cnt)) : 0u;
    } else {
      // more than 64 bits in one sequence (only with offsets / lengths near the format's limits): every lane
      // fetches its own field bit by bit from memory
      val = 0;
      const int topbit = P - (int)(incl - cnt);
      for (uint32_t k = 0; k < cnt; k++) {
        const int bi = topbit - 1 - (int)k;
        val = (val << 1) | (bi >= 0 ? (uint32_t)(q[bi >> 3] >> (bi & 7)) & 1u : 0u);
      }
    }
    const uint32_t nv = (is_x ? e1 : (e0 & 0xffffu)) + val;
    sb[slot] = nv;
    slot += slot_step;
    if (((i - first_rest) & 63u) == 63u) {
      flush(i - 63u, 64u);
      slot = slot0;
    }
    const uint32_t mirrored = (uint32_t)ZDPP(nv, 0x141);  // row_half_mirror: lane k <- lane 7 - k
    state = is_x ? mirrored : nv;
    P -= total;
    if (P < 0) return 23;  // the stream ran dry
  }
  if ((nseq - first_rest) & 63u) flush(first_rest + ((nseq - first_rest) & ~63u), (nseq - first_rest) & 63u);
  return P == 0 ? 0 : 24;  // every bit must be used
}

// The literals of one block (Huffman coded: lit_type 2, or 3 = Treeless) -> B.lit_out.  Returns 0 or a diagnostic code.
// LDS: ZEntLds or ZLitLds (the same members; the latter without the room of the sequences phase).
template <class LDS>
__device__ __forceinline__ int zstd_literals_job(LDS& L, const ZBlock& B, const uint8_t* src, uint8_t* lit_out, uint32_t lane PROF_PARM) {
  int st = 0;
  int mb = 0;
  PROF_MARK(0);
  const uint8_t* q = src + B.content_off + B.lit_hdr;
  uint32_t qn = B.lit_comp;
  // ONE round trip for everything in front of the streams: the tree description (the block's own, or -- Treeless -- that of the block
  // that last described one, same frame) and this block's jump table (behind its own description; Treeless: at q) go to L.stage
  // together; the header byte tells how long the description is, i.e. where the streams are, BEFORE the table is built:
  // their last bytes and the lanes' first chunks are requested at once (hw_begin) and arrive while the table is built.
  const bool own = B.lit_type == 2;
  const uint8_t* td = own ? q : src + B.huf_off;
  const uint32_t tdn = own ? qn : B.huf_end - B.huf_off;
  {
    const uint32_t nst = tdn < ZL_STAGE ? tdn : ZL_STAGE;
    uint32_t* st32 = reinterpret_cast<uint32_t*>(L.stage);
    if (4 * lane < nst) st32[lane] = ld_u32(td + 4 * lane);  // (up to 3 bytes behind the section: slack of the arena)
    if (!own && lane >= ZL_STAGE / 4 && lane < ZL_STAGE / 4 + 2 && 4 * (lane - ZL_STAGE / 4) < qn) st32[lane] = ld_u32(q + 4 * (lane - ZL_STAGE / 4));
    wave_sync();
  }
  PROF_MARK(7);
  const uint32_t used = own && qn ? zhuf_desc_bytes(L.stage[0]) : 0u;  // (134 at most with its jump table: staged)
  const uint32_t jt = own ? used : ZL_STAGE;                             // where the jump table's six bytes stand in L.stage
  HWin h{};
  h.e_pre = -1;
  uint32_t sn = 0, sk = lane, lps = 64, sout = 0, son = B.lit_regen;  // this lane's stream: bytes, lane in it, lanes per stream, output
  if (used > qn) st = 11;
  else {
    q += used;
    qn -= used;
    if (B.lit_streams == 1) {
      sn = qn;
      hw_begin(h, q, sn, sk, lps, L.chunk);
    } else if (qn < 6) st = 13;
    else {
      const uint32_t s1 = L.stage[jt] | (L.stage[jt + 1] << 8), s2 = L.stage[jt + 2] | (L.stage[jt + 3] << 8), s3 = L.stage[jt + 4] | (L.stage[jt + 5] << 8);
      const uint32_t seg = (B.lit_regen + 3) / 4;
      if (6 + s1 + s2 + s3 > qn) st = 14;
      else if (seg * 3 > B.lit_regen) st = 15;
      else {
        const uint32_t s4 = qn - 6 - s1 - s2 - s3;
        const uint32_t k = lane >> 4;  // stream of this lane (16 lanes each)
        const uint32_t so = k == 0 ? 0 : (k == 1 ? s1 : (k == 2 ? s1 + s2 : s1 + s2 + s3));
        sn = k == 0 ? s1 : (k == 1 ? s2 : (k == 2 ? s3 : s4));
        sk = lane & 15;
        lps = 16;
        sout = k * seg;
        son = k < 3 ? seg : B.lit_regen - 3 * seg;
        hw_begin(h, q + 6 + so, sn, sk, lps, L.chunk);
      }
    }
  }
  if (!st) {
    if (own) {
      if (zhuf_tree(L, td, tdn, &mb, lane PROF_ARG) < 0) st = 11;
    } else {
      // Treeless: the table of the block that last described one (same frame)
      if (zhuf_tree(L, td, tdn, &mb, lane PROF_ARG) < 0) st = 12;
    }
  }
  PROF_MARK(1);
  if (!st) {
    const int bad = huf_decode_w16(h, L.h.huf, mb, sn, lit_out + sout, son, sk, lps, true PROF_ARG);
    if (__ballot(bad != 0)) st = 16;
  }
  wave_sync();
  PROF_MARK(2);
  return st;
}

// ---- the kernel -------------------------------------------------------------------------------------------------
// Two independent jobs per compressed block, each a workgroup of one wavefront: blockIdx.x < n_blocks decodes the
// SEQUENCES of block blockIdx.x (the long serial chain: those workgroups come first), blockIdx.x >= n_blocks decodes the
// LITERALS of block blockIdx.x - n_blocks.  status_out[job] = 0, or a diagnostic code (any nonzero value rejects the chunk).
// Both are read by the execution kernel while this one runs: status words start as ZSTD_PENDING, progress[b] counts the sequences
// of block b that are in memory.
#define ZSTD_PENDING 0xffffffffu
// Tables mode (ztab != nullptr; zstd_lanes.h): a sequences job only builds the block's tables and leaves them in ztab / zhdr;
// zstd_seq_lanes_kernel, behind this kernel, decodes the sequences and writes the job's status word.
extern "C" __global__ void __launch_bounds__(64) zstd_entropy_kernel(const ZBlock* __restrict__ blocks, uint32_t n_blocks, uint32_t* dump_words,
                                                                     uint32_t* status_out, uint32_t* progress, uint16_t* ztab, ZSeqHdr* zhdr) {
  __shared__ ZEntLds L;
  const uint32_t job = blockIdx.x;
  if (job >= 2 * n_blocks) return;
  const bool lit_job = job >= n_blocks;
  const uint32_t b = lit_job ? job - n_blocks : job;
  const uint32_t lane = threadIdx.x;
  PROF_BEGIN();
  const ZBlock& B = *glob(blocks + b);  // (read in place: a local copy indexed by `w` below would live in scratch)
  const uint8_t* src = as_global(B.src);
  uint8_t* lit_out = (uint8_t*)as_global((void*)B.lit_out);
  uint32_t* seq_out = (uint32_t*)as_global((void*)B.seq_out);
  uint32_t* dump = (uint32_t*)as_global((void*)dump_words) + (size_t)(b & 1023u) * 64;
  int st = 0;
  // ---- literals ----
  if (lit_job && B.lit_type >= 2) st = zstd_literals_job(L, B, src, lit_out, lane PROF_ARG);
  // ---- sequences ----
  if (!lit_job && B.nseq) {
    int ll_log = 0, of_log = 0, ml_log = 0;
    uint32_t p = B.seq_off + 1;  // own descriptions follow the modes byte
    const uint32_t end = B.content_end;
    if (lane == 0) L.s.zero = ZFse{0, 0, 0, 0};
    for (int w = 0; w < 3 && !st; w++) {
      ZFse* t = w == 0 ? L.s.ll : (w == 1 ? L.s.of : L.s.ml);
      int* lg = w == 0 ? &ll_log : (w == 1 ? &of_log : &ml_log);
      uint16_t* ct = ztab ? L.s.ctab + (w == 0 ? ZL_LL : (w == 1 ? ZL_OF : ZL_ML)) : nullptr;
      if (B.tab_off[w] == B.seq_off) {
        const int mode = (src[B.seq_off] >> (6 - 2 * w)) & 3;
        const long c = p <= end ? zfse_table(L, t, lg, mode, src + p, end - p, w, true, lane, ct) : -1;
        if (c < 0) st = 17 + w;
        else p += (uint32_t)c;
      } else {
        if (zfse_from_block(L, t, lg, src, B.tab_off[w], B.tab_end[w], w, lane, ct) < 0) st = 17 + w;
      }
    }
    wave_sync();
    PROF_MARK(3);
    if (ztab) {
      // tables mode: the cells go to memory (16 bytes per lane and step), the sequences are another kernel's
      uint4* g = reinterpret_cast<uint4*>((uint16_t*)as_global((void*)ztab) + (size_t)b * ZL_CELLS);
      const uint4* l = reinterpret_cast<const uint4*>(L.s.ctab);
      for (uint32_t k = lane; k < ZL_CELLS * 2 / 16; k += 64) g[k] = l[k];
      if (lane == 0) {
        ZSeqHdr h;
        h.bit_off = p;
        h.logs = (uint32_t)ll_log | (uint32_t)of_log << 8 | (uint32_t)ml_log << 16;
        h.status = (uint32_t)st;
        h.pad = 0;
        *(ZSeqHdr*)as_global((void*)(zhdr + b)) = h;
      }
      PROF_MARK(5);
      PROF_END_AT(112);
      return;
    }
    if (!st) {
      if (p > end) st = 20;
      else st = zfse_sequences(L, src + p, end - p, B.nseq, ll_log, of_log, ml_log, seq_out, dump, (uint32_t*)as_global((void*)(progress + b)), lane);
    }
    PROF_MARK(4);
    PROF_COUNT(4, B.nseq);
  }
  PROF_END_AT(112);
  // everything this job wrote must be in memory before its status says so (the literals went out as ordinary stores)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) {
    __hip_atomic_store(&status_out[job], (uint32_t)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // jobs that have ended (progress[n_blocks] is the execution kernel's queue head): once all have, the workgroups of the
    // execution kernel beside this one stop taking chunks and leave the rest to a launch that may fill the machine
    __hip_atomic_fetch_add(progress + n_blocks + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- table scale: the literals as a kernel of their own (zstd_lanes.h runs the sequences) -------------------------------
// The Huffman phase needs a third of the LDS of zstd_entropy_kernel (whose allocation is sized by the FSE tables of the sequences
// phase): 28 wavefronts per CU instead of 10.  The decoder is a chain of table lookups per lane -- what it needs is wavefronts
// to switch to.  Job j = the literals of block j; status_out[n_blocks + j] as zstd_entropy_kernel writes it.
struct ZLitLds {
  union {
    struct {
      struct {
        uint16_t huf[2048];  // sym | nb << 8
        uint8_t weights[256];
        FseEnt wt[64];
      } h;
      int16_t norm[256];
      uint16_t next[256];
      __attribute__((aligned(4))) uint8_t stage[208];
    };
    struct {  // ... behind the table, once it is built: every lane's piece of its Huffman stream (HWin)
      uint16_t huf_[2048];
      __attribute__((aligned(16))) uint8_t chunk[64 * ZL_CHUNK];
    };
  };
};
extern "C" __global__ void __launch_bounds__(64) zstd_literals_kernel(const ZBlock* __restrict__ blocks, uint32_t n_blocks, uint32_t* status_out, uint32_t* progress) {
  __shared__ ZLitLds L;
  const uint32_t b = blockIdx.x;
  if (b >= n_blocks) return;
  const uint32_t lane = threadIdx.x;
  PROF_BEGIN();
  const ZBlock& B = *glob(blocks + b);
  int st = 0;
  if (B.lit_type >= 2) st = zstd_literals_job(L, B, as_global(B.src), (uint8_t*)as_global((void*)B.lit_out), lane PROF_ARG);
  PROF_END_AT(112);
  // (as zstd_entropy_kernel ends a job: the execution kernel may be running beside the kernel that follows this one)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) {
    __hip_atomic_store(&status_out[n_blocks + b], (uint32_t)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(progress + n_blocks + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
