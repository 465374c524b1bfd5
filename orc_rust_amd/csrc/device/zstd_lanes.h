// zstd_lanes.h -- Zstandard sequences at table scale: FOUR LANES PER COMPRESSED BLOCK, sixteen blocks per wavefront.
//
// The FSE state chain of a block is serial (RFC 8878 3.1.1.3.2.1.1): one table lookup and one variable-length bit read
// per sequence, each depending on the one before.  zstd_entropy.h gives a block a whole wavefront -- right when a call
// holds few blocks (a stripe: the call lasts as long as its longest chain), wasteful when it holds tens of thousands (a
// table: 8 of 64 lanes busy, the stage is bound by the instructions it issues per sequence).  Here the same instructions
// advance 16 blocks:
//
//   * zstd_entropy_kernel (tables mode) builds the three FSE tables of every block and leaves them in memory as 2-byte cells
//     {symbol : 6, next-state number : 10}; the number of state bits and the base of the next state follow from the number
//     (nb = log - floor(log2 n), next = (n << nb) - size), a symbol's extra bits and base value from 64-entry tables held in
//     vector registers (ds_bpermute: lane s holds symbol s).  A block's tables are 2560 bytes: LL 512 + ML 512 + OF 256 cells.
//   * zstd_seq_quads_kernel: a wavefront loads the tables of its 16 chains into LDS (40 KiB).  A chain is a QUAD of lanes: lane 0
//     follows the offset table, 1 the match-length table, 2 the literal-length table (3 stands by as a second literal-length
//     lane).  A lane reads one cell, decodes one state, cuts its own extra bits and its own state bits out of a 64-bit window
//     of the backward bit stream and keeps its own value; the six field widths of a sequence reach the other lanes of the quad
//     by DPP quad broadcasts.
//   * The bit stream comes through a 512-byte RING per chain in LDS.  This target counts loads and stores in ONE in-order
//     counter per wavefront: with 16 chains refilling their windows from memory whenever they ran low, some chain did so in
//     nearly every step and every such wait was for the load issued a step before -- loads cost 10 of the kernel's 18 ms,
//     stores 5 (measured by launching it again without them).  Now memory is touched every SIXTEEN steps, by all lanes at
//     once: a quad writes the 64 bytes fetched a visit ago into its ring, fetches the next 64 bytes of its stream (16 per
//     lane) into registers and stores the sixteen values each lane has collected (three arrays per block -- offset values,
//     match lengths, literal lengths --, so a lane's values are contiguous: four 16-byte stores) -- in that order: the wait
//     for the fetched bytes is a wait for everything issued before them, and at the top of a visit all of it is 16 steps old.
//     A step reads its window from the ring (one unaligned 8-byte LDS read) a step AHEAD, beside the step before.
// LDS per wavefront: 16 x (2560 + 528) bytes + 3840 for the values on their way out + 512 of base values = 52.5 KiB: three wavefronts (48 chains) per CU.
// Blocks are dealt out in the order of their sequence counts (the host sorts them), so the chains of a wavefront end together.
// Status words are those of zstd_entropy_kernel; the execution kernel (lz_exec_wave_kernel) runs behind this kernel.
#pragma once

#define ZQ_BCAST(v, k) ((uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (k) * 0x55, 0xf, 0xf, false))  // lane k of the quad (every lane is written: the old value is a don't-care, naming the source saves the move that would clear it)
#define ZQ_STEPS 16  // steps between two visits to memory (16 values = one 64-byte run of a lane's array: put_vals)
#define ZQ_RING (ZQ_STEPS > 8 ? 512 : 256)  // bytes of a chain's stream held in LDS: stream byte b lives at ring[b & (ZQ_RING - 1)]
#define ZQ_RING_BYTES (ZQ_RING + 16)        // (+ the first 8 bytes again behind the end: a window may start at the last byte; 16-byte aligned)
#define ZQ_NEED ((ZQ_STEPS * 89 + 7) / 8 + 15)  // bytes below the cursor a group of steps may take (89 bits a step at the very most) + a window
#define ZQ_ROOM (ZQ_RING - 66)              // the next 64 bytes fit below `fill` once the cursor is within this many bytes of it
#define ZQ_XROW 80                          // bytes of a row of the values on their way out (16 values + padding: see put_vals)

__device__ __forceinline__ void zstd_seq_quads_body(const ZBlock* __restrict__ blocks, uint32_t n_chains, const uint16_t* ztab_, const ZSeqHdr* zhdr_, uint32_t* status_out_,
                                                    uint16_t* tabs, uint8_t* rings, uint8_t* xpose) {
  const uint32_t lane = threadIdx.x;
  const uint32_t first = blockIdx.x * 16u;
  if (first >= n_chains) return;
  const uint32_t nch = n_chains - first < 16u ? n_chains - first : 16u;
  const uint16_t* ztab = glob(ztab_);
  const ZSeqHdr* zhdr = glob(zhdr_);
  uint32_t* status_out = glob(status_out_);
  {
    // the tables of this wavefront's chains are one contiguous piece of ztab
    const uint4* g = reinterpret_cast<const uint4*>(ztab + (size_t)first * ZL_CELLS);
    uint4* l = reinterpret_cast<uint4*>(tabs);
    const uint32_t n16 = nch * (ZL_CELLS * 2 / 16);
    for (uint32_t k = lane; k < n16; k += 64) l[k] = g[k];
  }
  __syncthreads();
  const uint32_t cw = lane >> 2, r = lane & 3u;  // chain of the wavefront; role: 0 OF, 1 ML, 2 LL (3: LL again, stores nothing)
  const bool has = cw < nch;
  const uint32_t c = first + (has ? cw : 0u);    // (idle quads follow chain 0 and store nothing)
  const ZBlock* B = glob(blocks + c);
  const ZSeqHdr H = zhdr[c];
  const uint8_t* q = as_global(B->src) + H.bit_off;
  const uint32_t end = B->content_end;
  int st = (int)H.status;
  uint32_t qn = 0;
  if (!st) {
    if (H.bit_off > end) st = 20;
    else qn = end - H.bit_off;
  }
  uint32_t lastb = 0;
  if (!st) {
    if (qn == 0) st = 21;
    else {
      lastb = q[qn - 1];
      if (lastb == 0) st = 21;
    }
  }
  if (st) {  // nothing to follow: a stream of its own keeps the lanes harmless
    qn = 8;
    lastb = 1;
  }
  const uint32_t nseq_b = B->nseq;
  uint32_t nseq = has && !st ? nseq_b : 0u;
  // the block's output: its sequences as 8-byte records {offset value : 29, match length : 18, literal length : 17} (ZSEQ_PACK)
  uint2* const so8 = reinterpret_cast<uint2*>(as_global((void*)B->seq_out));
  const uint32_t llog = H.logs & 0xffu, olog = (H.logs >> 8) & 0xffu, mlog = (H.logs >> 16) & 0xffu;
  const uint32_t log = r == 0 ? olog : (r == 1 ? mlog : llog);
  const uint32_t size = 1u << log, c0 = 31u - log;  // state bits of a cell = clz(its state number) - (31 - log)
  // base values per symbol: lane s holds symbol s (read by ds_bpermute, off the chain)
  const uint32_t cA = r == 1 ? 32u : 16u, cC = r == 1 ? 43u : 25u, cD = r == 1 ? 36u : 19u;  // (extra bits of a length code: see the loop)
  // base values per length code: in LDS, looked up when the values leave (put_vals) -- off the chain, sixteen steps at a time
  uint32_t* const btab = reinterpret_cast<uint32_t*>(xpose + 48 * ZQ_XROW);  // [0, 64) literal lengths, [64, 128) match lengths
  btab[lane] = lane < 36 ? Z_LL_BASE[lane] : 0u;
  btab[64 + lane] = lane < 53 ? Z_ML_BASE[lane] : 0u;
  // this lane's share of the field positions: its extra bits end behind those of the roles below it, its state bits behind
  // the extra bits of all and the state bits of the roles above it (the states follow each other LL, ML, OF)
  const uint32_t mx1 = r >= 1 ? ~0u : 0u, mx2 = r >= 2 ? ~0u : 0u, ms1 = r <= 1 ? ~0u : 0u, ms0 = r == 0 ? ~0u : 0u;

  // The bit stream is read from its last set bit downwards; P = unread bits.  Stream bytes [fill, fill + 256) are in the quad's
  // ring (bytes before the stream are zeros); lane r of the quad moves bytes [16 r, 16 r + 16) of every 64-byte block.
  int P = (int)(qn - 1) * 8 + (31 - __builtin_clz(lastb));
  uint8_t* const rg = rings + cw * ZQ_RING_BYTES;
  auto load16 = [&](int b0) -> uint4 {  // stream bytes [b0, b0 + 16): zeros outside the stream (at most 15 bytes behind it are read: slack of the arena)
    uint4 v = make_uint4(0, 0, 0, 0);
    if (b0 >= 0 && (uint32_t)b0 < qn) {
      const uint64_t a = ld_u64(q + b0), b = ld_u64(q + b0 + 8);
      v = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
    return v;
  };
  auto ring_put = [&](int b0, uint4 v) {
    const uint32_t o = (uint32_t)b0 & (ZQ_RING - 1u);
    *reinterpret_cast<uint4*>(rg + o) = v;
    if (o == 0) *reinterpret_cast<uint2*>(rg + ZQ_RING) = make_uint2(v.x, v.y);
  };
  int fill = (((int)qn - 1) & ~63) + 64;  // (nothing yet: the block that holds the last byte comes first)
  for (int k = 0; k < ZQ_RING / 64; k++) {
    fill -= 64;
    ring_put(fill + 16 * (int)r, load16(fill + 16 * (int)r));
  }
  lds_order();
  uint4 pend = make_uint4(0, 0, 0, 0);
  bool pending = false;
  // the 64 bits that end 57..64 bits above... the word that holds stream bits [8 B, 8 B + 64), B = (top - 57) >> 3: at least 57 bits below `top`
  auto window_at = [&](int top, int& t8) -> uint64_t {
    const int B = (top - 57) >> 3;
    t8 = top - 8 * B;
    uint64_t w;
    __builtin_memcpy(&w, rg + ((uint32_t)B & (ZQ_RING - 1u)), 8);
    return w;
  };
  auto mask = [](uint32_t cnt) -> uint32_t { return ~(~0u << (cnt & 31u)); };

  const uint32_t tb = (has ? cw : 0u) * ZL_CELLS + (r == 0 ? ZL_OF : (r == 1 ? ZL_ML : ZL_LL));
  uint32_t s;
  {
    // initial states: LL, OF, ML (at most 26 bits)
    int t8;
    const uint64_t x = window_at(P, t8);
    const uint32_t e = r == 0 ? llog + olog : (r == 1 ? llog + olog + mlog : llog);
    s = (uint32_t)(x >> ((uint32_t)t8 - e)) & mask(log);
    P -= (int)(llog + olog + mlog);
    if (P < 0 && !st) {
      st = 22;
      nseq = 0;
    }
  }
  uint32_t maxn = nseq;
  for (int o = 32; o; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)maxn, o);
    maxn = t > maxn ? t : maxn;
  }
  maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
  int Pfin = nseq ? 1 : 0;  // unread bits behind the last sequence (must be none)
  // A step issues TWO LDS reads: its cell, then its window (an unaligned 8-byte read: slow in the LDS pipeline).  In that order: the
  // cell is what the chain waits for, the window is only needed ~35 instructions later, when the fields are cut out.  (Read a step
  // ahead -- as soon as the widths of the step before are added up -- the window sat in FRONT of the next cell read in the in-order
  // LDS queue: 6.2 instead of 4.2 ms for the kernel without its stores.)

  // The sixteen values a lane has collected leave at the NEXT visit to memory, behind its fetch: loads and stores share one
  // in-order counter, so the wait for a fetched block is also a wait for every store issued before it -- stores issued at the end
  // of a group were waited for at once, at the top of the next (2.3 of the kernel's 8.7 ms); now both are sixteen steps old then.
  // They leave PACKED, 8 bytes a sequence, 128 contiguous bytes a quad: a store instruction costs what its lanes are spread over
  // (each lane on a 64-byte run of its own: ~110 cycles of the CU per instruction, four per visit -- a third of the kernel once
  // the chain took ~150 ns a step).  The three lanes of a quad put their values side by side through LDS (rows of 80 bytes: the
  // eight rows a 16-byte LDS store serves at once lie in different banks), every lane packs four sequences, and two instructions
  // write 64 contiguous bytes per quad each.  (The same transposition WITHOUT packing -- three arrays, whole 64-byte runs -- was
  // no faster than the scattered stores.)
  uint32_t val[ZQ_STEPS];
  uint8_t* const xrow = xpose + (cw * 3u + (r < 3 ? r : 0u)) * ZQ_XROW;  // this lane's row (values of steps 0..15)
  const uint8_t* const xq = xpose + cw * 3u * ZQ_XROW + r * 8u;           // the quad's rows, at this lane's two steps of each half
  auto put_vals = [&](uint32_t at) {  // the values of steps [at, at + ZQ_STEPS)
    if (r < 3) {
#pragma unroll
      for (int k = 0; k < ZQ_STEPS / 4; k++) *reinterpret_cast<uint4*>(xrow + 16 * k) = make_uint4(val[4 * k], val[4 * k + 1], val[4 * k + 2], val[4 * k + 3]);
    }
    lds_order();
#pragma unroll
    for (int h = 0; h < 2; h++) {
      // lane j of the quad: sequences at + 8 h + 2 j and the one behind it
      const uint2 o2 = *reinterpret_cast<const uint2*>(xq + 32 * h);
      const uint2 m2 = *reinterpret_cast<const uint2*>(xq + ZQ_XROW + 32 * h);
      const uint2 l2 = *reinterpret_cast<const uint2*>(xq + 2 * ZQ_XROW + 32 * h);
      const uint32_t ma = btab[64 + (m2.x >> 16)] + (m2.x & 0xffffu), mb = btab[64 + (m2.y >> 16)] + (m2.y & 0xffffu);
      const uint32_t la = btab[l2.x >> 16] + (l2.x & 0xffffu), lb = btab[l2.y >> 16] + (l2.y & 0xffffu);
      const uint2 a = zseq_pack(o2.x, ma, la), b = zseq_pack(o2.y, mb, lb);
      const uint32_t i = at + 8u * (uint32_t)h + 2u * r;
      // (an array is padded to an even number of records: a pair that starts inside it ends inside its padding)
      if (i < nseq) *reinterpret_cast<uint4*>(so8 + i) = make_uint4(a.x, a.y, b.x, b.y);
    }
    lds_order();
  };
  // (a quad without a chain, or behind the end of its chain, goes on decoding whatever it finds in its ring and stores nothing)
  for (uint32_t i0 = 0; i0 < maxn; i0 += ZQ_STEPS) {
    // ---- memory: every lane at once.  The block fetched a visit ago goes into the ring; the next one is fetched when the
    // ring has room for it (its top 64 bytes are no longer needed).  A step takes 89 bits at the very most: a quad that would
    // have fewer than ZQ_NEED bytes below its cursor fetches at once (never seen with real data: a step takes ~2.5 bytes).
    if (i0 < nseq) {
      if (pending) {
        fill -= 64;
        ring_put(fill + 16 * (int)r, pend);
        pending = false;
      }
      while ((P >> 3) - fill < ZQ_NEED) {
        fill -= 64;
        ring_put(fill + 16 * (int)r, load16(fill + 16 * (int)r));
      }
    }
    if (i0 < nseq && (P >> 3) - fill < ZQ_ROOM) {
      pend = load16(fill - 64 + 16 * (int)r);
      pending = true;
    }
    if (i0) put_vals(i0 - ZQ_STEPS);  // (behind the fetch: the registers it lands in are only free once the counter is at zero)
    lds_order();
#pragma unroll
    for (int k = 0; k < ZQ_STEPS; k++) {
      const uint32_t i = i0 + (uint32_t)k;
      const uint32_t cell = tabs[tb + s];
      int t8 = 0;
      const uint64_t x = window_at(P, t8);
      const uint32_t sym = cell & 63u, ns = cell >> 6;
      const uint32_t nb = (uint32_t)__builtin_clz(ns) - c0;
      const uint32_t next = (ns << (nb & 31u)) - size;
      // extra bits of the symbol, by arithmetic (a table lookup here would be a second LDS round trip on the chain): offset codes
      // are their own count; length codes below A have none, up to C they have max(1, (code - A) / 2), from there on code - D
      // (LL: A 16, C 25, D 19; ML: A 32, C 43, D 36 -- RFC 8878 3.1.1.3.2.1.1)
      const int tA = (int)sym - (int)cA;
      const uint32_t half = (uint32_t)(tA >> 1) > 1u ? (uint32_t)(tA >> 1) : 1u;
      const uint32_t lenb = tA < 0 ? 0u : (sym < cC ? half : sym - cD);
      const uint32_t xb = (r == 0 ? sym : lenb) & 31u;
      // the widths of the quad's six fields: they follow each other downwards -- offset, match length, literal length extra
      // bits, then LL, ML, OF state bits
      const uint32_t b0 = ZQ_BCAST(xb, 0), b1 = ZQ_BCAST(xb, 1), b2 = ZQ_BCAST(xb, 2);
      const uint32_t n0 = ZQ_BCAST(nb, 0), n1 = ZQ_BCAST(nb, 1), n2 = ZQ_BCAST(nb, 2);
      const uint32_t e3 = b0 + b1 + b2;
      const uint32_t xend = b0 + (b1 & mx1) + (b2 & mx2);
      const uint32_t send = e3 + n2 + (n1 & ms1) + (n0 & ms0);
      const uint32_t total = e3 + n0 + n1 + n2;
      // behind the last sequence no state is read: what must be left then is what its extra bits leave (the lanes read on)
      if (i + 1 == nseq) Pfin = P - (int)e3;
      const int Pk = P;
      P -= (int)total;
      uint32_t xv = (uint32_t)(x >> (((uint32_t)t8 - xend) & 63u)) & mask(xb);
      uint32_t sv = (uint32_t)(x >> (((uint32_t)t8 - send) & 63u)) & mask(nb);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(total > 57u) != 0ull, 0)) {
        // some quad of the wavefront has more bits than the window surely holds (offsets / lengths near the format's limits):
        // a window per field, for every lane (the branch is taken by the whole wavefront or not at all)
        int u8;
        const uint64_t x1 = window_at(Pk - (int)(xend - xb), u8);
        xv = (uint32_t)(x1 >> ((uint32_t)u8 - xb)) & mask(xb);
        const uint64_t x2 = window_at(Pk - (int)(send - nb), u8);
        sv = (uint32_t)(x2 >> ((uint32_t)u8 - nb)) & mask(nb);
      }
      // an offset value is arithmetic; a length keeps {code, extra bits (16 at most)}: its base value is looked up in put_vals (a
      // lookup per step -- ds_bpermute, lane s holding code s -- stood in the in-order LDS queue in front of the next cell read)
      val[k] = r == 0 ? (1u << (sym & 31u)) + xv : (sym << 16 | xv);
      s = next + sv;
    }
  }
  if (maxn) put_vals((maxn - 1u) & ~(uint32_t)(ZQ_STEPS - 1));
  if (!st && Pfin != 0) st = Pfin < 0 ? 23 : 24;  // the stream ran dry / every bit must be used
  if (has && r == 0) status_out[c] = (uint32_t)st;
}

extern "C" __global__ void __launch_bounds__(64) zstd_seq_quads_kernel(const ZBlock* __restrict__ blocks, uint32_t n_chains, const uint16_t* ztab_,
                                                                      const ZSeqHdr* zhdr_, uint32_t* status_out_) {
  __shared__ __attribute__((aligned(16))) uint16_t tabs[16 * ZL_CELLS];
  __shared__ __attribute__((aligned(16))) uint8_t rings[16 * ZQ_RING_BYTES];
  __shared__ __attribute__((aligned(16))) uint8_t xpose[48 * ZQ_XROW + 512];  // (+ the base values of the length codes)
  zstd_seq_quads_body(blocks, n_chains, ztab_, zhdr_, status_out_, tabs, rings, xpose);
}
