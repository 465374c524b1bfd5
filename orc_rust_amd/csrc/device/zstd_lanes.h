// zstd_lanes.h -- Zstandard sequences, ONE LANE PER COMPRESSED BLOCK (table scale).
//
// The FSE state chain of a block is serial (RFC 8878 3.1.1.3.2.1.1): one table lookup and one variable-length bit read
// per sequence, each depending on the one before.  zstd_entropy.h gives a block a whole wavefront -- right when a call
// holds few blocks (a stripe: the call lasts as long as its longest chain), wasteful when it holds tens of thousands (a
// table: 8 of 64 lanes busy, the stage is bound by the instructions it issues per sequence).  Here every lane of a
// wavefront follows a chain of its own, with ordinary per-lane code: the same instructions advance 64 blocks.
//
//   * zstd_entropy_kernel (tables mode) builds the three FSE tables of every block as before and leaves them in memory as
//     2-byte cells {symbol : 6, next-state number : 10}; the number of state bits and the base of the next state follow
//     from the number (nb = log - floor(log2 n), next = (n << nb) - size), a symbol's extra bits and base value from two
//     64-entry tables held in a vector register (ds_bpermute).  A block's tables are 2560 bytes: LL 512 + ML 512 + OF 256 cells.
//   * zstd_seq_lanes_kernel: a wavefront loads the tables of its NCH chains into LDS (NCH x 2560 bytes: what bounds the chains
//     in flight per CU is LDS, 64 of them), then every lane decodes its block's sequences: three 2-byte LDS reads, the six bit
//     fields of a sequence cut out of a 64-bit window of the backward bit stream that the lane keeps in registers (128 bits
//     + the next 8 bytes on their way: no memory access sits on the chain), 12 bytes stored per sequence.
// Blocks are dealt out in the order of their sequence counts (the host sorts them), so the chains of a wavefront end together.
// Output and status words are those of zstd_entropy_kernel; lz_exec_kernel runs behind this kernel.
#pragma once

template <int NCH>
__device__ __forceinline__ void zstd_seq_lanes(uint16_t* tabs, const ZBlock* blocks, uint32_t n_chains, const uint16_t* ztab_, const ZSeqHdr* zhdr_,
                                               uint32_t* status_out_) {
  const uint32_t lane = threadIdx.x;
  const uint32_t first = blockIdx.x * NCH;
  if (first >= n_chains) return;
  const uint32_t nch = n_chains - first < (uint32_t)NCH ? n_chains - first : (uint32_t)NCH;
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
  const bool has = lane < nch;
  const uint32_t c = first + (has ? lane : 0u);  // (idle lanes follow lane 0's chain and store nothing)
  const ZBlock* B = glob(blocks + c);
  const ZSeqHdr H = zhdr[c];
  const uint8_t* q = as_global(B->src) + H.bit_off;
  const uint32_t end = B->content_end;
  uint32_t* so = (uint32_t*)as_global((void*)B->seq_out);
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
  if (st) {  // nothing to follow: an empty stream of zeros keeps the lane harmless
    qn = 8;
    lastb = 1;
  }
  uint32_t nseq = has && !st ? B->nseq : 0u;
  const uint32_t llog = H.logs & 0xffu, olog = (H.logs >> 8) & 0xffu, mlog = (H.logs >> 16) & 0xffu;
  const uint32_t lsize = 1u << llog, osize = 1u << olog, msize = 1u << mlog;
  // a symbol's extra bits and base value: lane s holds symbol s
  const uint32_t llpack = lane < 36 ? (uint32_t)Z_LL_BITS[lane] | Z_LL_BASE[lane] << 8 : 0u;
  const uint32_t mlpack = lane < 53 ? (uint32_t)Z_ML_BITS[lane] | Z_ML_BASE[lane] << 8 : 0u;

  // the bit stream is read from its last set bit downwards.  P = unread bits; lo = stream bits [wb64 - 64, wb64),
  // hi = [wb64, wb64 + 64) (kept as hi << 1), nx = the word below lo, on its way; always 0 <= P - wb64 <= 63.
  int P = (int)(qn - 1) * 8 + (31 - __builtin_clz(lastb));
  int wb64 = 8 * (int)qn - 64;
  uint64_t hi1 = zl_word(q, wb64) << 1;
  uint64_t lo = zl_word(q, wb64 - 64);
  // (nx is kept as loaded, with the shift that zeroes what lies before the stream: nothing waits for the load before the
  // refill after this one)
  uint64_t nx;
  uint32_t nxs;
  auto fetch = [&](int wb) {
    const int bo = wb >> 3;
    nx = ld_u64(q + (bo < 0 ? 0 : bo));
    nxs = bo < 0 ? (uint32_t)(-bo) * 8u : 0u;
  };
  fetch(wb64 - 128);
  auto refill = [&]() {
    if (P < wb64) {
      hi1 = lo << 1;
      lo = nxs >= 64u ? 0ull : nx << nxs;
      wb64 -= 64;
      fetch(wb64 - 128);
    }
  };
  auto window = [&]() -> uint64_t {  // stream bits [P - 64, P), bit 63 = the next unread bit
    const uint32_t s = (uint32_t)(P - wb64);
    return (lo >> s) | (hi1 << (63u - s));
  };
  // field of `cnt` bits (< 32) that ends `e` bits below the top of the window x (neg = -e: shifts take the low 6 bits)
  auto field = [](uint64_t x, uint32_t neg, uint32_t cnt) -> uint32_t { return (uint32_t)(x >> (neg & 63u)) & ~(~0u << cnt); };

  const uint32_t tb = (has ? lane : 0u) * ZL_CELLS;
  uint32_t sL, sO, sM;
  {
    // initial states: LL, OF, ML
    const uint64_t x = window();
    const uint32_t n1 = 0u - llog, n2 = n1 - olog, n3 = n2 - mlog;
    sL = field(x, n1, llog);
    sO = field(x, n2, olog);
    sM = field(x, n3, mlog);
    P -= (int)(llog + olog + mlog);
    if (P < 0 && !st) {
      st = 22;
      nseq = 0;
    }
    refill();
  }
  uint32_t maxn = nseq;
  for (int o = 32; o; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)maxn, o);
    maxn = t > maxn ? t : maxn;
  }
  maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
  int Pfin = nseq ? 1 : 0;  // unread bits behind the last sequence (must be none)

  for (uint32_t i = 0; i < maxn; i++) {
    const uint32_t cl = tabs[tb + ZL_LL + sL], cm = tabs[tb + ZL_ML + sM], co = tabs[tb + ZL_OF + sO];
    const uint32_t symL = cl & 63u, symM = cm & 63u, symO = co & 63u;
    const uint32_t nsL = cl >> 6, nsM = cm >> 6, nsO = co >> 6;
    const uint32_t pkL = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(symL << 2), (int)llpack);
    const uint32_t pkM = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(symM << 2), (int)mlpack);
    // (a lane without a chain, or behind the end of its chain, goes on decoding whatever it finds: its loads stay inside its
    // stream, its states inside its tables, it stores nothing)
    uint32_t nbL = llog - (31u - (uint32_t)__builtin_clz(nsL));
    uint32_t nbM = mlog - (31u - (uint32_t)__builtin_clz(nsM));
    uint32_t nbO = olog - (31u - (uint32_t)__builtin_clz(nsO));
    const uint32_t nextL = (nsL << (nbL & 31u)) - lsize, nextM = (nsM << (nbM & 31u)) - msize, nextO = (nsO << (nbO & 31u)) - osize;
    const bool last = i + 1 == nseq;
    if (last) nbL = nbM = nbO = 0;  // no state update behind the last sequence
    const uint32_t ofb = symO & 31u, mlb = pkM & 31u, llb = pkL & 31u;
    // the fields follow each other downwards: offset, match length, literal length extra bits, then LL, ML, OF state bits
    const uint32_t n1 = 0u - ofb, n2 = n1 - mlb, n3 = n2 - llb, n4 = n3 - nbL, n5 = n4 - nbM, n6 = n5 - nbO;
    const uint32_t total = 0u - n6;
    uint32_t ofx, mlx, llx, v4, v5, v6;
    if (__builtin_expect(total > 64u, 0)) {
      // more than 64 bits in one sequence (offsets / lengths near the format's limits): the extra bits (at most 63), then the states
      uint64_t x = window();
      ofx = field(x, n1, ofb);
      mlx = field(x, n2, mlb);
      llx = field(x, n3, llb);
      P -= (int)(0u - n3);
      refill();
      x = window();
      const uint32_t m4 = 0u - nbL, m5 = m4 - nbM, m6 = m5 - nbO;
      v4 = field(x, m4, nbL);
      v5 = field(x, m5, nbM);
      v6 = field(x, m6, nbO);
      P -= (int)(0u - m6);
    } else {
      const uint64_t x = window();
      ofx = field(x, n1, ofb);
      mlx = field(x, n2, mlb);
      llx = field(x, n3, llb);
      v4 = field(x, n4, nbL);
      v5 = field(x, n5, nbM);
      v6 = field(x, n6, nbO);
      P -= (int)total;
    }
    refill();
    if (i < nseq) {
      so[0] = (1u << ofb) + ofx;
      so[1] = (pkM >> 8) + mlx;
      so[2] = (pkL >> 8) + llx;
    }
    so += 3;
    if (last) Pfin = P;
    sL = (nextL + v4) & 511u;
    sM = (nextM + v5) & 511u;
    sO = (nextO + v6) & 255u;
  }
  if (!st && Pfin != 0) st = Pfin < 0 ? 23 : 24;  // the stream ran dry / every bit must be used
  if (has) status_out[c] = (uint32_t)st;
}

#define ZL_KERNEL(NAME, NCH)                                                                                                                  \
  extern "C" __global__ void __launch_bounds__(64) NAME(const ZBlock* __restrict__ blocks, uint32_t n_chains, const uint16_t* ztab,          \
                                                        const ZSeqHdr* zhdr, uint32_t* status_out) {                                          \
    __shared__ __attribute__((aligned(16))) uint16_t tabs[NCH * ZL_CELLS];                                                                    \
    zstd_seq_lanes<NCH>(tabs, blocks, n_chains, ztab, zhdr, status_out);                                                                      \
  }
ZL_KERNEL(zstd_seq_lanes64_kernel, 64)
ZL_KERNEL(zstd_seq_lanes16_kernel, 16)
