// rle_scan.hip -- pass 1 of the RLE decoders: find run boundaries in parallel.
//
// Replaces the serial header walk of the reference (rle_v2/mod.rs:112-146 decode_batch called
// run after run from rle.rs:68-107; rle_v1.rs:143-159; byte.rs:228-247).  Run boundaries are
// data dependent, so the stream is cut into RLE_BLK-byte blocks and one LANE walks one block:
//
//   guess     every block looks for a run header INSIDE ITS OWN BYTES without knowing the chain:
//             a "full run" header (RLE v2: length field 512, not SHORT_REPEAT; byte RLE / RLE v1:
//             a 128-literal group) that is followed by two (three) more headers of the same
//             kind is accepted -- a random payload position passes with probability < 2^-30.
//             Such a block is STRONG.  Blocks that find none fall back to offset 0 (WEAK): in
//             short-run regions a wrong entry re-synchronises with the true chain within a few
//             hops, so the guess is usually harmless there;
//   fill      a strong block whose last run covers whole following blocks marks them as
//             pass-through (their entry = distance to the next header);
//   relax     a weak block takes exit[b-1] as its entry and re-walks if that differs from what it
//             used (chaotic relaxation; the fixed point is unique).  Strong blocks keep their
//             entry, so garbage from a weak neighbour cannot destroy them;
//   verify    every block compares its entry with exit[b-1]; the first mismatch per stream is
//             recorded and rle_repair walks the true chain from there.  The result is exact
//             whatever the heuristics did.
//
// Traffic: header bytes of every run, plus one pass over the stream by the candidate search
// (skipped when the stride guess from the stream's first run is already verified).
#include "rle_kernels.h"
#include "rle_parse.h"

#define RLE_MEND_MIN 64u  // inconsistent blocks a stream must have for the parallel mending passes (rle_mend_kernel)
#define RLE_EXACT_MIN 32u  // inconsistent blocks the mending passes may leave before the stream takes the exact parallel walk (rle_exact_*)

// Phase timing for development builds (-DORC_PROF): per-phase sum / max of wavefront wall-clock ticks (10 ns).
#ifdef ORC_PROF
__device__ unsigned long long g_prof[176];  // [0, 32): walk kernels, [32, 48): counters, [48, 80): decompressors, [80, 112): lz_exec, [112, 144): zstd_entropy, [144, 176): counters
struct Prof {
  unsigned long long t;
  unsigned long long acc[16];
};
#define PROF_BEGIN()            \
  Prof prof;                    \
  for (int i_ = 0; i_ < 16; i_++) prof.acc[i_] = 0; \
  prof.t = wall_clock64()
#define PROF_MARK(i)                         \
  do {                                       \
    unsigned long long t_ = wall_clock64();  \
    prof.acc[(i) & 15] += t_ - prof.t;       \
    prof.t = t_;                             \
  } while (0)
#define PROF_END()                                              \
  do {                                                          \
    if ((threadIdx.x & 63) == 0) {                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++)         \
        if (prof.acc[i_]) {                                     \
          atomicAdd(&g_prof[2 * i_], prof.acc[i_]);             \
          atomicMax(&g_prof[2 * i_ + 1], prof.acc[i_]);         \
        }                                                       \
    }                                                           \
  } while (0)
#define PROF_END_AT(base)                                       \
  do {                                                          \
    if ((threadIdx.x & 63) == 0) {                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++)         \
        if (prof.acc[i_]) {                                     \
          atomicAdd(&g_prof[(base) + 2 * i_], prof.acc[i_]);    \
          atomicMax(&g_prof[(base) + 2 * i_ + 1], prof.acc[i_]);\
        }                                                       \
    }                                                           \
  } while (0)
#define PROF_COUNT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[144 + (i)], (unsigned long long)(v)); } while (0)
#define PROF_ARG , prof
#define PROF_PARM , Prof& prof
#else
#define PROF_ARG
#define PROF_PARM
#define PROF_MARK(i)
#define PROF_BEGIN()
#define PROF_END()
#define PROF_END_AT(base)
#define PROF_COUNT(i, v)
#endif

__device__ __forceinline__ void wave_sync_scan() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ const RleJob* find_job_by_block(const RleJob* jobs, int njobs, uint32_t b) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= b) lo = mid;
    else hi = mid - 1;
  }
  return &jobs[lo];
}

// Walks are speculative until verified: a header that fails to parse is most likely not a header
// at all (the chain started at a wrong byte), so the walk re-synchronises one byte further instead
// of giving up.  On the TRUE chain a failing run ends the stream for the decoder anyway (the
// expansion reports the error at that run and ignores what follows), so this costs nothing there.
__device__ __forceinline__ uint32_t hop_bytes(const RunHdr& h) { return h.err ? 1u : h.size; }

template <int CODEC>
__device__ __forceinline__ void walk_block(const uint8_t* data, uint64_t len, uint32_t lb, uint32_t entry, bool is_signed,
                                           int nbits, uint32_t* exit_out, uint32_t* nvals_out) {
  uint64_t bend = (uint64_t)(lb + 1) * RLE_BLK;
  uint64_t end = bend < len ? bend : len;
  uint64_t pos = (uint64_t)lb * RLE_BLK + entry;
  uint32_t nv = 0;
  while (pos < end) {
    uint32_t sz, n, err;
    hop_parse<CODEC>(data + pos, len - pos, is_signed, nbits, sz, n, err);
    nv += n;
    pos += err ? 1u : sz;
  }
  *exit_out = pos > bend ? (uint32_t)(pos - bend) : 0u;
  *nvals_out = nv;
}

__device__ __forceinline__ void walk_dispatch(const RleJob* j, const uint8_t* data, uint64_t len, uint32_t lb, uint32_t entry,
                                              uint32_t* ex, uint32_t* nv) {
  if (j->codec == CODEC_RLE2) walk_block<CODEC_RLE2>(data, len, lb, entry, j->is_signed, j->nbits, ex, nv);
  else if (j->codec == CODEC_RLE1) walk_block<CODEC_RLE1>(data, len, lb, entry, j->is_signed, j->nbits, ex, nv);
  else walk_block<CODEC_BYTE>(data, len, lb, entry, false, 8, ex, nv);
}

// Verified run starts (ROW_INDEX positions, orcgpu_stream::entries): one lane per entry follows the run headers from its
// position to the next entry's (the stream's end for the last) and notes, for every block a run carries it into, where that
// block's first header is.  A stream's entries start with its first byte.  Entries that are not in order, or a chain that
// does not arrive exactly at the next entry, mark the job: the guess round then ignores its hints.
template <int CODEC>
__device__ __forceinline__ bool hint_chain(const uint8_t* data, uint64_t len, uint64_t p, uint64_t end, bool is_signed, int nbits, uint32_t* hint, uint32_t max_hops) {
  while (p < end) {
    if (!max_hops--) return true;  // (the last entry of a stream: what lies far behind it is left to the ordinary walk)
    uint32_t hsize, hn, herr;
    hop_parse<CODEC>(data + p, len - p, is_signed, nbits, hsize, hn, herr);
    if (herr || !hsize) return false;  // (a run that does not parse: the ordinary walk deals with the stream)
    const uint64_t q = p + hsize;
    for (uint64_t b = p / RLE_BLK + 1; b <= q / RLE_BLK && b * RLE_BLK < len; b++) hint[b] = (uint32_t)(q - b * RLE_BLK);
    p = q;
  }
  return p == end || end >= len;
}
extern "C" __global__ void __launch_bounds__(64) rle_hint_kernel(RleJob* jobs, const RleHint* hints, uint32_t n_hints, RleBlocks blk,
                                                                  const uint64_t* scalars, const uint32_t* chunk_start) {
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  if (t >= n_hints) return;
  const RleHint h = hints[t];
  RleJob* j = jobs + h.job;
  const uint64_t len = scalars[j->len_idx];
  auto position = [&](const RleHint& e) -> int64_t {
    return (int64_t)(e.chunk == 0xffffffffu ? 0u : chunk_start[e.chunk]) + (int64_t)e.byte - (int64_t)j->hint_skip;
  };
  const int64_t p0 = position(h);
  int64_t p1 = (int64_t)len;
  if (t + 1 < j->hint0 + j->n_hints) p1 = position(hints[t + 1]);
  if (p1 > (int64_t)len) p1 = (int64_t)len;
  if (p0 < 0 || p0 > p1) {
    atomicOr(&j->hint_bad, 1u);
    return;
  }
  uint32_t* hint = blk.hint + j->block0;
  if (t == j->hint0) {
    if (p0 != 0) {
      atomicOr(&j->hint_bad, 1u);
      return;
    }
    hint[0] = 0;
  }
  const uint8_t* data = as_global(j->data);
  // A stream's last entry has no next one to arrive at: its lane follows a thousand runs (a stream may reach far beyond the
  // rows asked for: 40 000 short runs were 20 ms for one lane)
  const uint32_t max_hops = t + 1 < j->hint0 + j->n_hints ? 0xffffffffu : 1024u;
  bool ok;
  if (j->codec == CODEC_RLE2) ok = hint_chain<CODEC_RLE2>(data, len, (uint64_t)p0, (uint64_t)p1, j->is_signed, j->nbits, hint, max_hops);
  else if (j->codec == CODEC_RLE1) ok = hint_chain<CODEC_RLE1>(data, len, (uint64_t)p0, (uint64_t)p1, j->is_signed, j->nbits, hint, max_hops);
  else ok = hint_chain<CODEC_BYTE>(data, len, (uint64_t)p0, (uint64_t)p1, false, 8, hint, max_hops);
  if (!ok) atomicOr(&j->hint_bad, 1u);
}

// A "full run" header at p that is followed by `hops` more headers of the same kind.
// `bm` (optional): prefilter bitmaps of the 64 blocks from byte `wbase` on -- a header position whose
// bit is clear cannot start a full run (rejects without touching memory), and a set bit stands in
// for the last hop's parse.
template <int CODEC>
__device__ __forceinline__ bool plausible_header(const uint8_t* data, uint64_t len, uint64_t p, bool is_signed, int nbits, uint32_t* size_out = nullptr,
                                                 const unsigned long long (*bm)[8] = nullptr, uint64_t wbase = 0) {
  // lean parse (size / count only); the run type comes from the header byte itself
  uint32_t hsize, hn, herr;
  hop_parse<CODEC>(data + p, len - p, is_signed, nbits, hsize, hn, herr);
  if (herr) return false;
  if (size_out) *size_out = hsize;
  const uint32_t hb = data[p];
  if (CODEC == CODEC_RLE2) {
    if (hn != 512 || (hb >> 6) == RT_SR) return false;
  } else {
    if (hn != 128 || hb < 0x80) return false;  // 128 literals
  }
  const int hops = CODEC == CODEC_RLE2 ? 2 : 3;
  auto forward = [&]() -> bool {
    uint64_t q = p + hsize;
    for (int i = 0; i < hops; i++) {
      if (q == len) return true;
      if (q > len) return false;
      if (bm && q >= wbase && q - wbase < 64ull * RLE_BLK) {
        uint64_t o = q - wbase;
        bool set = (bm[o >> 9][(o >> 6) & 7] >> (o & 63)) & 1;
        if (!set) return false;
        if (i == hops - 1) return true;
      }
      uint32_t gsize, gn, gerr;
      hop_parse<CODEC>(data + q, len - q, is_signed, nbits, gsize, gn, gerr);
      const uint32_t gb = data[q];
      if (gerr || gn != hn || (CODEC == CODEC_RLE2 ? (gb >> 6) != (hb >> 6) : gb < 0x80)) return false;
      q += gsize;
    }
    return true;
  };
  if (forward()) return true;
  // The runs that follow may be of another kind -- a writer that flushes its encoder at row-group boundaries ends every group with
  // a short run -- while the runs BEFORE are full ones: two DIRECT runs with this header's two bytes (which fix the size) ending
  // exactly where this one starts are the same evidence, looked at backwards.
  if (CODEC == CODEC_RLE2 && (hb >> 6) == RT_DIRECT && p >= 2ull * hsize && p + 1 < len) {
    const uint32_t h2 = ld_u32(data + p) & 0xffffu;
    return (ld_u32(data + p - hsize) & 0xffffu) == h2 && (ld_u32(data + p - 2ull * hsize) & 0xffffu) == h2;
  }
  return false;
}

// Prefilter: bit i of the result = byte i of the 8-byte word `w` may start a full-run header
// (`nx` = the same word shifted by one byte, i.e. the bytes that follow).
template <int CODEC>
__device__ __forceinline__ uint32_t prefilter8(uint64_t w, uint64_t nx) {
  uint64_t m;
  if (CODEC == CODEC_RLE2) {
    // low bit set (length bit 8), type != SHORT_REPEAT, and the next byte == 0xFF (length low byte)
    uint64_t n = ~nx;
    uint64_t ff = (n - 0x0101010101010101ull) & ~n & 0x8080808080808080ull;
    m = ff & ((w & 0x0101010101010101ull) << 7) & ((w | (w << 1)) & 0x8080808080808080ull);
  } else {
    uint64_t x = w ^ 0x8080808080808080ull;  // header byte 0x80 = 128 literals
    m = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
  }
  return (uint32_t)(((m >> 7) * 0x0102040810204080ull) >> 56);  // gather bit 7 of every byte
}

// Candidate search for the 64 blocks of one wavefront (lane <-> block).  Lanes whose stride guess
// already verified skip it.  For every block that needs it, all 64 lanes read the block with
// coalesced 8-byte loads and leave a 512-bit prefilter bitmap in LDS; the owning lane then
// verifies the candidates in order.  Returns the verified entry (or RLE_BLK).
template <int CODEC>
__device__ __forceinline__ uint32_t wave_find_candidates(const uint8_t* data, uint64_t len, uint32_t lb, bool need, bool is_signed, int nbits,
                                                         unsigned long long (*bm)[8], uint32_t lane PROF_PARM) {
  unsigned long long todo = __ballot(need);
  uint32_t lb0 = lb - lane;  // first block of the wave (wave-uniform)
  const uint64_t wbase = (uint64_t)lb0 * RLE_BLK;
  // dense: most blocks of the window need a search -> read the whole 32 KiB window with 16-byte loads
  // (1 KiB = two blocks per instruction, eight instructions in flight)
  const bool dense = __popcll(todo) >= 24;
  if (dense) {
    // neighbouring wavefronts start in different quarters of their windows: in lockstep they would
    // all hit the same memory channels at the same time
    const int rot = (int)((lb0 >> 6) & 3) * 8;
    for (int it = 0; it < 32; it += 8) {
      const int i0 = (it + rot) & 31;
      uint64_t lo[8], hi[8];
      uint32_t edge[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        uint64_t p = wbase + (uint64_t)(i0 + u) * 1024 + lane * 16;
        // branch-free (all eight loads stay in flight): out-of-range lanes read the stream's first
        // bytes instead and discard them; the staged stream carries ORC_PAD readable bytes past len
        const bool in = p < len;
        uint64_t v[2];
        __builtin_memcpy(v, data + (in ? p : 0), 16);
        lo[u] = in ? v[0] : 0, hi[u] = in ? v[1] : 0;
        const bool ein = lane == 63 && p + 16 < len;
        const uint32_t eb = data[ein ? p + 16 : 0];
        edge[u] = ein ? eb : 0u;  // the byte after the instruction's 1 KiB
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        uint64_t p = wbase + (uint64_t)(i0 + u) * 1024 + lane * 16;
        uint32_t nxt = __shfl_down((uint32_t)(lo[u] & 0xff), 1);
        if (lane == 63) nxt = edge[u];
        uint64_t nlo = (lo[u] >> 8) | (hi[u] << 56), nhi = (hi[u] >> 8) | ((uint64_t)nxt << 56);
        uint32_t bits = 0;
        if (p < len) {
          bits = prefilter8<CODEC>(lo[u], nlo) | (prefilter8<CODEC>(hi[u], nhi) << 8);
          uint64_t rem = len - p;
          if (rem < 16) bits &= (1u << rem) - 1;
        }
        reinterpret_cast<uint16_t*>(bm[2 * (i0 + u) + (lane >> 5)])[lane & 31] = (uint16_t)bits;
      }
    }
    todo = 0;
  }
  while (todo) {
    // four blocks per trip: all loads are issued before any of them is consumed
    uint32_t kk[4];
    uint64_t w[4], nx[4], pp[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      kk[u] = todo ? (uint32_t)__builtin_ctzll(todo) : 64u;
      if (todo) todo &= todo - 1;
      pp[u] = (uint64_t)(lb0 + kk[u]) * RLE_BLK + lane * 8;
      bool in = kk[u] < 64 && pp[u] < len;
      w[u] = in ? ld_u64(data + pp[u]) : 0;
      nx[u] = in ? ld_u64(data + pp[u] + 1) : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (kk[u] < 64) {
        uint32_t bits = 0;
        if (pp[u] < len) {
          bits = prefilter8<CODEC>(w[u], nx[u]);
          uint64_t rem = len - pp[u];
          if (rem < 8) bits &= (1u << rem) - 1;
        }
        reinterpret_cast<uint8_t*>(bm[kk[u]])[lane] = (uint8_t)bits;
      }
    }
  }
  wave_sync_scan();
  PROF_MARK(1);
  uint32_t found = RLE_BLK;
#ifdef ORC_PROF
  int g_tries = 0;
#endif
  if (need) {
    const uint64_t b0 = (uint64_t)lb * RLE_BLK;
    int tries = 0;
#ifdef ORC_PROF
    g_tries = 0;
#endif
    for (int wi = 0; wi < 8 && found == RLE_BLK && tries < 6; wi++) {
      unsigned long long m = bm[lane][wi];
      while (m && tries < 6) {
        uint32_t i = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        uint64_t c = b0 + wi * 64 + i;
        tries++;
#ifdef ORC_PROF
        g_tries++;
#endif
        if (plausible_header<CODEC>(data, len, c, is_signed, nbits, nullptr, dense ? bm : nullptr, wbase)) {
          found = (uint32_t)(c - b0);
          break;
        }
      }
    }
  }
  wave_sync_scan();
  PROF_MARK(2);
#ifdef ORC_PROF
  {
    uint32_t t = need ? (uint32_t)g_tries : 0u, s = t;
    for (int off = 32; off; off >>= 1) {
      uint32_t o = __shfl_xor(t, off);
      t = o > t ? o : t;
      s += __shfl_xor(s, off);
    }
    if (lane == 0) {
      atomicAdd(&g_prof[32], t);
      atomicMax(&g_prof[33], t);
      atomicAdd(&g_prof[34], s);
      atomicAdd(&g_prof[36], 1);
    }
  }
#endif
  return found;
}

// Does the run chain that starts at `from` have a header exactly at `target` (> from)?
template <int CODEC>
__device__ __forceinline__ bool chain_hits(const uint8_t* data, uint64_t len, uint64_t from, uint64_t target, bool is_signed, int nbits) {
  uint64_t p = from;
  while (p < target && p < len) {
    RunHdr h;
    run_parse<CODEC, false>(data + p, len - p, is_signed, nbits, h);
    p += hop_bytes(h);
  }
  return p == target;
}

__device__ __forceinline__ bool rle_hopeless(const RleJob* j, uint64_t len) {
  const uint64_t nb = (len + RLE_BLK - 1) / RLE_BLK;
  return j->stat_bad >= RLE_MEND_MIN && (uint64_t)j->stat_bad * 8 >= nb;
}
// mode 0: guess; 1: relaxation round; 2: verify only; 3: strong blocks fill their pass-through blocks.
// Every wavefront covers 64 consecutive blocks of ONE stream (block ranges are RLE_TILE aligned).
// (Workgroups of ONE wavefront -- the wavefronts never meet --: a lane's walk starts while the other column lane's execution kernel,
// single-wavefront workgroups that fill every CU's LDS, still runs; a 256-thread workgroup then waits until four slots and 16 KiB
// of LDS are free on one CU at once, which that kernel never leaves: round 5's first walk launch spent 1.4 - 4.9 ms in the queue.)
extern "C" __global__ void __launch_bounds__(64) rle_walk_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars,
                                                                  uint32_t total_blocks, int mode) {
  __shared__ unsigned long long bitmaps[1][64][8];
  uint32_t b = blockIdx.x * 64u + threadIdx.x;
  uint32_t lane = threadIdx.x & 63;
  uint32_t bw = b - lane;  // first block of this wavefront
  if (bw >= total_blocks) return;
  PROF_BEGIN();
  RleJob* j = const_cast<RleJob*>(find_job_by_block(jobs, njobs, bw));
  uint32_t lb = b - j->block0;
  uint64_t len = scalars[j->len_idx];
  bool in_job = lb < j->nblocks;
  bool live = in_job && ((uint64_t)lb * RLE_BLK < len || lb == 0);
  const uint8_t* data = as_global(j->data);
  if (mode == 0) {
    if ((b & 31) == 0) blk.badmap[b >> 5] = 0;  // the verify round marks blocks here
    uint32_t want = 0, strong = 0;
    bool need = false;
    const uint32_t hinted = live && blk.hint && j->n_hints && !j->hint_bad ? blk.hint[b] : 0xffffffffu;
    if (live) {
      if (lb == 0) {
        strong = 1;
      } else if (hinted != 0xffffffffu) {
        want = hinted;  // the run chain from a verified start gives the block's first header (rle_hint_kernel)
        strong = 1;
      } else {
        // stride guess from the stream's first run: exact for streams of equal-sized runs
        RunHdr h;
        if (j->codec == CODEC_RLE2) run_parse<CODEC_RLE2, false>(data, len, j->is_signed, j->nbits, h);
        else if (j->codec == CODEC_RLE1) run_parse<CODEC_RLE1, false>(data, len, j->is_signed, j->nbits, h);
        else run_parse<CODEC_BYTE, false>(data, len, false, 8, h);
        uint32_t s0 = h.size ? h.size : 1;
        uint32_t r = (uint32_t)(((uint64_t)lb * RLE_BLK) % s0);
        uint32_t sg = r ? s0 - r : 0;
        uint64_t gp = (uint64_t)lb * RLE_BLK + sg;  // guessed position of the first header at or after the block start
        // Accept the guess only if the run BEFORE it is a verified full run of exactly s0 bytes that
        // starts before this block: then [gp - s0, gp) holds no other header and gp is the block's
        // first header (sg < RLE_BLK) or the block lies inside that run (sg >= RLE_BLK).
        bool ok = false;
        if (gp >= s0 && gp <= len) {
          // cheap screen first: the two header bytes at gp - s0 (and at gp) must equal the stream's
          // first two bytes.  For sub-encodings whose size is a function of those bytes alone
          // (DIRECT, byte RLE) that already proves a run of exactly s0 bytes; the others still get
          // the chained check.
          const uint32_t head2 = ld_u32(data) & 0xffffu;
          const uint32_t a2 = ld_u32(data + gp - s0) & 0xffffu;
          const uint32_t b2 = gp < len ? (ld_u32(data + gp) & 0xffffu) : head2;
          const uint32_t mask2 = j->codec == CODEC_RLE2 ? 0xffffu : 0xffu;
          if (((a2 ^ head2) & mask2) == 0 && ((b2 ^ head2) & mask2) == 0) {
            // (only for runs of some length: two matching header bytes at the predicted places are
            // evidence for a 3 KiB stride, not for a 2-byte one -- 0xff 0xff is everywhere in a bitmap)
            const bool self_sized = ((j->codec == CODEC_RLE2 && h.type == RT_DIRECT) || j->codec == CODEC_BYTE) && s0 >= 64;
            if (self_sized && !h.err) {
              ok = true;
            } else {
              uint32_t psize = 0;
              uint64_t pp = gp - s0;
              if (j->codec == CODEC_RLE2) ok = plausible_header<CODEC_RLE2>(data, len, pp, j->is_signed, j->nbits, &psize);
              else if (j->codec == CODEC_RLE1) ok = plausible_header<CODEC_RLE1>(data, len, pp, j->is_signed, j->nbits, &psize);
              else ok = plausible_header<CODEC_BYTE>(data, len, pp, false, 8, &psize);
              ok = ok && psize == s0;
            }
          }
        }
        if (ok) {
          want = sg;  // >= RLE_BLK: this block lies inside that run (pass-through)
          strong = 1;
        } else {
          need = true;
        }
      }
    }
    if (mode == 0) PROF_MARK(0);
    if (__ballot(need)) {
      uint32_t cand;
      unsigned long long(*bm)[8] = bitmaps[0];
      if (j->codec == CODEC_RLE2) cand = wave_find_candidates<CODEC_RLE2>(data, len, lb, need, j->is_signed, j->nbits, bm, lane PROF_ARG);
      else if (j->codec == CODEC_RLE1) cand = wave_find_candidates<CODEC_RLE1>(data, len, lb, need, j->is_signed, j->nbits, bm, lane PROF_ARG);
      else cand = wave_find_candidates<CODEC_BYTE>(data, len, lb, need, false, 8, bm, lane PROF_ARG);
      if (need && cand < RLE_BLK) {
        want = cand;
        strong = 1;
        need = false;
      }
    }
    PROF_MARK(3);
    if (!in_job) {
      blk.nvals[b] = 0;  // tile padding behind the job's last block: the scans read it
      return;
    }
    if (!live) {
      blk.entry[b] = RLE_BLK;
      blk.exit_[b] = 0;
      blk.nvals[b] = 0;
      blk.flags[b] = 0;
      return;
    }
    uint32_t ex, nv;
    if (want >= RLE_BLK) {
      ex = want - RLE_BLK;
      nv = 0;
    } else {
      walk_dispatch(j, data, len, lb, want, &ex, &nv);
    }
    blk.entry[b] = want;
    blk.exit_[b] = ex;
    blk.nvals[b] = nv;
    blk.flags[b] = (uint8_t)strong;
    PROF_MARK(4);
    PROF_END();
    return;
  }
  if (mode == 1) {
    // Relaxation: entry[b] must equal exit[b-1].  Up to four sweeps per launch run inside the
    // wavefront (the predecessor's exit comes from the neighbouring lane; lane 0 reads the previous
    // wavefront's last block from memory, so chains across wavefronts need another launch).
    uint32_t e = RLE_BLK, ex = 0, nv = 0, fl = 0;
    if (live) {
      e = blk.entry[b];
      ex = blk.exit_[b];
      fl = blk.flags[b];
    }
    uint32_t pex0 = 0, pfl0 = 0;
    bool panc0 = false;
    if (live && lb > 0 && lane == 0) {
      pex0 = blk.exit_[b - 1];
      pfl0 = blk.flags[b - 1];
      panc0 = lb == 1 || blk.entry[b - 1] == blk.exit_[b - 2];
    }
    bool dirty = false;
    for (int sweep = 0; sweep < 4; sweep++) {
      uint32_t pex = __shfl_up(ex, 1), pfl = __shfl_up(fl, 1);
      if (lane == 0) pex = pex0, pfl = pfl0;
      // "anchored": the block's entry agrees with its own predecessor's exit
      bool anchored = live && (lb == 0 || pex == e);
      bool panc = __shfl_up((int)anchored, 1);
      if (lane == 0) panc = panc0;
      bool change = false;
      if (live && lb > 0 && pex != e) {
        // A strong block ignores a weak predecessor (its exit may be garbage).  Two strong neighbours
        // that disagree: one of them holds a false candidate (it passed verification by hopping onto
        // the true chain).  An anchored predecessor is trusted; otherwise this block keeps its entry
        // only if its own chain reaches the predecessor's header exactly (then the predecessor
        // skipped runs, not us).
        bool keep = false;
        if (fl) {
          if (!pfl) {
            // ... unless the weak predecessor's exit lies in front of this block's verified header and its
            // chain arrives exactly there: then shorter runs precede the full run the search found
            keep = true;
            if (pex < e && e < RLE_BLK) {
              uint64_t from = (uint64_t)lb * RLE_BLK + pex, target = (uint64_t)lb * RLE_BLK + e;
              bool hits;
              if (j->codec == CODEC_RLE2) hits = chain_hits<CODEC_RLE2>(data, len, from, target, j->is_signed, j->nbits);
              else if (j->codec == CODEC_RLE1) hits = chain_hits<CODEC_RLE1>(data, len, from, target, j->is_signed, j->nbits);
              else hits = chain_hits<CODEC_BYTE>(data, len, from, target, false, 8);
              keep = !hits;
            }
          } else if (!panc && e < pex && e < RLE_BLK) {
            uint64_t from = (uint64_t)lb * RLE_BLK + e, target = (uint64_t)lb * RLE_BLK + pex;
            if (j->codec == CODEC_RLE2) keep = chain_hits<CODEC_RLE2>(data, len, from, target, j->is_signed, j->nbits);
            else if (j->codec == CODEC_RLE1) keep = chain_hits<CODEC_RLE1>(data, len, from, target, j->is_signed, j->nbits);
            else keep = chain_hits<CODEC_BYTE>(data, len, from, target, false, 8);
          }
        }
        if (!keep) {
          if (pex >= RLE_BLK) {
            ex = pex - RLE_BLK;
            nv = 0;
          } else {
            walk_dispatch(j, data, len, lb, pex, &ex, &nv);
          }
          e = pex;
          change = dirty = true;
        }
      }
      if (!__ballot(change)) break;
    }
    if (dirty) {
      blk.entry[b] = e;
      blk.exit_[b] = ex;
      blk.nvals[b] = nv;
    }
    return;
  }
  if (!live) return;
  if (mode == 3) {
    // a strong block whose last run extends over whole following blocks owns them
    if (!blk.flags[b]) return;
    uint32_t ex = blk.exit_[b];
    uint32_t nbk = j->nblocks;
    for (uint32_t pb = lb + 1; ex >= RLE_BLK && pb < nbk; pb++, ex -= RLE_BLK) {
      if ((uint64_t)pb * RLE_BLK >= len) break;
      blk.entry[b - lb + pb] = ex;
      blk.exit_[b - lb + pb] = ex - RLE_BLK;
      blk.nvals[b - lb + pb] = 0;
      blk.flags[b - lb + pb] = 1;
    }
    return;
  }
  // mode 2: verify; mode 5: verify again behind rle_mend_kernel (only the streams it worked on)
  if (mode == 5) {
    if (j->stat_bad < RLE_MEND_MIN || rle_hopeless(j, len)) return;
    // the bitmap words are written afresh (rle_mend_kernel reads them as a snapshot and leaves them alone)
    const uint32_t want5 = lb == 0 ? 0u : blk.exit_[b - 1];
    const bool bad = want5 != blk.entry[b];
    const unsigned long long m = __ballot(bad);
    blk.badmap[b >> 5] = (uint32_t)(lane < 32 ? m : m >> 32);  // (every lane of the half stores the same word)
    if (bad) atomicMin(&j->first_bad, lb);
    if (m && lane == 0) atomicAdd(&j->bad_left, (uint32_t)__builtin_popcountll(m));  // what this mending pass left (rle_exact_*)
    return;
  }
  uint32_t want = lb == 0 ? 0u : blk.exit_[b - 1];
  if (want == blk.entry[b]) return;
  atomicMin(&j->first_bad, lb);
  atomicAdd(&j->stat_bad, 1u);
  atomicOr(&blk.badmap[b >> 5], 1u << (b & 31));
}

// ---- short-run streams: exact intra-wave propagation out of LDS --------------------------------------
// Blocks without a verified header ("weak") are typical of streams of short runs (SHORT_REPEAT,
// small DIRECT groups, byte-RLE bitmaps).  One wavefront takes a span of 64 blocks plus a 32-block
// warm-up in front of it into LDS (48 KiB) and iterates entry[l] = exit[l-1] across lanes until
// nothing changes: inside the span the result is exactly what a serial walk would give, and the
// warm-up makes the span's first entry right with overwhelming probability (a chain started at a
// wrong byte merges with the true chain after a few run lengths).  Hops cost an LDS access instead
// of a global-memory round trip.  Spans without weak blocks return at once.
#define RLE_WARM 32u
template <int CODEC>
__device__ __forceinline__ void hop_lds(const uint8_t* buf, const uint8_t* data, uint64_t gstart, uint64_t gend, uint64_t len, uint64_t pos,
                                        bool is_signed, int nbits, uint32_t& sz, uint32_t& n, uint32_t& err) {
  // bytes a header parse may look at: 24 (RLE v2 window), 8 (byte RLE), a whole 128-varint literal group (RLE v1)
  constexpr uint64_t LOOK = CODEC == CODEC_RLE1 ? 1320 : 24;
  if (pos + LOOK <= gend) hop_parse<CODEC>(buf + (pos - gstart), len - pos, is_signed, nbits, sz, n, err);
  else hop_parse<CODEC>(data + pos, len - pos, is_signed, nbits, sz, n, err);
}

// Walk block lbv from `entry` (headers come from LDS); value counts are recomputed by count_lds()
// once the entries have converged.
template <int CODEC>
__device__ __forceinline__ void walk_lds(const uint8_t* buf, const uint8_t* data, uint64_t gstart, uint64_t gend, uint64_t len, uint32_t lbv,
                                         uint32_t entry, bool is_signed, int nbits, uint32_t* exit_io) {
  const uint64_t bend = (uint64_t)(lbv + 1) * RLE_BLK;
  const uint64_t end = bend < len ? bend : len;
  uint64_t pos = (uint64_t)lbv * RLE_BLK + entry;
  while (pos < end) {
    uint32_t sz, n, err;
    hop_lds<CODEC>(buf, data, gstart, gend, len, pos, is_signed, nbits, sz, n, err);
    pos += err ? 1u : sz;
  }
  *exit_io = pos > bend ? (uint32_t)(pos - bend) : 0u;
}

template <int CODEC>
__device__ __forceinline__ uint32_t count_lds(const uint8_t* buf, const uint8_t* data, uint64_t gstart, uint64_t gend, uint64_t len, uint32_t lbv,
                                              uint32_t entry, bool is_signed, int nbits) {
  const uint64_t bend = (uint64_t)(lbv + 1) * RLE_BLK;
  const uint64_t end = bend < len ? bend : len;
  uint64_t pos = (uint64_t)lbv * RLE_BLK + entry;
  uint32_t nv = 0;
  while (pos < end) {
    uint32_t sz, n, err;
    hop_lds<CODEC>(buf, data, gstart, gend, len, pos, is_signed, nbits, sz, n, err);
    nv += n;
    pos += err ? 1u : sz;
  }
  return nv;
}

// Exit and value count of one block for EVERY entry at once: every byte position of the block is parsed as if a run started
// there (next[p] = where that run ends, cnt[p] = its values; an unparsable position hops one byte like the walks do), then
// pointer doubling inside the block makes next[p] the first position at or behind the block's end that the chain from p
// reaches, and cnt[p] the values of the runs on the way.  Cost: 8 parses + at most 9 x 8 table updates per lane, whatever
// the data -- no serial chain through the block, and no dependence on how quickly wrong chains merge with the true one.
// (Tried in round 4, not kept: the doubling in registers -- eight sub-blocks of 64 positions, hops by ds_bpermute, composed
// from the last sub-block backwards, only the finished table written to LDS: 168 cross-lane operations instead of ~430 LDS
// accesses and no barrier per round, exact on the whole suite -- and 12 % SLOWER (C3's walk 0.61 -> 0.69 ms): the rounds are
// not what a block costs.  Nor is the warm-up in front of a span: 32 -> 8 blocks made the repairs behind it cost more.)
template <int CODEC>
__device__ __forceinline__ void block_exit_table(const uint8_t* data, uint64_t len, uint32_t lbv, bool is_signed, int nbits, uint32_t (*tab)[RLE_BLK],
                                                 uint32_t lane, int& cur) {
  const uint64_t bstart = (uint64_t)lbv * RLE_BLK;
  const uint32_t limit = len - bstart < RLE_BLK ? (uint32_t)(len - bstart) : RLE_BLK;  // positions below it lie inside the stream
  cur = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t p = lane + 64u * i;
    uint32_t nx = 0xffffu, c = 0;
    if (p < limit) {
      uint32_t sz, n, err;
      const uint64_t pos = bstart + p;
      if (CODEC == CODEC_RLE2 && len - pos >= 32) {
        // 32 or more bytes left: whatever the lean parse rejects is an error for the full parse too (the walks hop one
        // byte then), so the table needs nothing else -- and most byte positions of a block are no run header at all.
        // (Windows come straight from memory: neighbouring lanes read neighbouring bytes, the block sits in L2.  Staging the block
        // in LDS first -- one coalesced load, the windows as three unaligned LDS reads -- was 5 % slower.)
        const Win24 win = ld_win24(data + pos);
        err = !(rle2_lean(win, nbits, sz, n) && sz <= len - pos);
      } else {
        hop_parse<CODEC>(data + pos, len - pos, is_signed, nbits, sz, n, err);
      }
      const uint32_t to = p + (err ? 1u : sz);
      nx = to < 0xffffu ? to : 0xfffeu;
      c = err ? 0u : n;
    }
    tab[0][p] = (nx & 0xffffu) | (c << 16);  // one word per position: where the chain from it is (low half), the values on the way (16 bits, wrapping)
  }
  wave_sync_scan();
  for (int round = 0; round < 10; round++) {
    const uint32_t* t0 = tab[cur];
    uint32_t* t1 = tab[cur ^ 1];
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t p = lane + 64u * i;
      uint32_t v = t0[p];
      const uint32_t q = v & 0xffffu;
      if (q < limit) {
        const uint32_t w = t0[q];
        v = (w & 0xffffu) | ((v & 0xffff0000u) + (w & 0xffff0000u));
        changed = true;
      }
      t1[p] = v;
    }
    wave_sync_scan();
    cur ^= 1;
    if (!__ballot(changed)) break;
  }
}

#ifndef RLE_SHORT_WAVES
#define RLE_SHORT_WAVES 8  // (4: the same for C3, 5 % slower on the lineitem stripes; 16: 50 % slower.  Per-wavefront clocks of the kernel on C3
                           // (-DORC_PROF): parsing the positions 35 %, the doubling rounds 12 %, the chain's lookups 12 %, waiting at the two barriers of a
                           // round 39 % -- with the chain's three stores per block moved out from between them: the same)
#endif
#ifndef RLE_SHORT_MIN_WEAK
#define RLE_SHORT_MIN_WEAK 16  // unverified blocks (of 64) that make a span a short-run span; fewer: the relaxation rounds and rle_mend_kernel
#endif
// One span by a workgroup of RLE_SHORT_WAVES wavefronts: wavefront w takes the blocks k0 + w, + RLE_SHORT_WAVES, ...: it builds the
// block's all-entries table (block_exit_table: the expensive, entry-independent part), waits for the exit of the block before
// -- published in LDS by the wavefront that has it --, looks its own exit up and publishes it.  No barrier inside a span: with
// one around every round of tables the wavefronts spent 39 % of the kernel waiting for the slowest of them (-DORC_PROF); here a
// slow wavefront delays the lookups behind it, not the tables the others build meanwhile.
template <int CODEC>
__device__ __forceinline__ void short_span(RleJob* j, const RleBlocks& blk, uint64_t len, uint32_t lb0, uint32_t b0g, uint32_t (*tabs)[2][RLE_BLK],
                                           volatile uint32_t* exits, volatile uint32_t* ready, uint32_t tid, unsigned long long live_m,
                                           unsigned long long weak_m, uint32_t carry PROF_PARM) {
  const uint8_t* data = as_global(j->data);
  const bool is_signed = j->is_signed;
  const int nbits = j->nbits;
  const uint32_t lane = tid & 63, wv = tid >> 6;
  const uint32_t wstart = lb0 >= RLE_WARM ? lb0 - RLE_WARM : 0u;
  const uint32_t nwarm = lb0 - wstart;
  for (uint32_t i = tid; i < RLE_WARM + 64; i += 64 * RLE_SHORT_WAVES) ready[i] = 0;
  __syncthreads();
  // The warm-up blocks in front of the span come first (entry 0 at the first one: a chain started at a wrong byte merges
  // with the true chain after a few run lengths), then the span's own.  A weak block enters where its predecessor's last
  // run ended, a strong one at its verified header.
  const bool first_weak = weak_m & 1;
  // `carry` (not ~0): where the last run of the block in front of the span ends, as the span before -- walked by this workgroup a
  // moment ago -- found it: the true chain's entry into this span's first block, no warm-up needed.  (A workgroup takes up to eight
  // consecutive spans of a stream: seven of its eight warm-ups -- a third of the tables it built at table scale -- were for
  // entries it already knew.)
  const bool carried = carry != 0xffffffffu;
  const uint32_t k0 = (nwarm > 0 && first_weak && !carried) ? 0u : nwarm;
  for (uint32_t k = k0 + wv; k < nwarm + 64; k += RLE_SHORT_WAVES) {
    const bool in_span = k >= nwarm;
    const uint32_t sl = k - nwarm;
    const bool live = !in_span || ((live_m >> sl) & 1);  // (behind the stream's last block: nothing but the word to those who wait)
    const bool bweak = !in_span || ((weak_m >> sl) & 1);
    uint32_t e = 0, ex = 0, nv = 0;
    if (live) {
      const uint32_t lbv = wstart + k;
      const uint64_t bstart = (uint64_t)lbv * RLE_BLK;
      const uint32_t limit = bstart >= len ? 0u : (len - bstart < RLE_BLK ? (uint32_t)(len - bstart) : RLE_BLK);
      const uint32_t e_given = bweak ? 0u : blk.entry[b0g + sl];  // (on its way while the table is built)
      int cur = 0;
      if (limit) block_exit_table<CODEC>(data, len, lbv, is_signed, nbits, tabs[wv], lane, cur);
      if (!bweak) {
        e = e_given;
      } else if (k == k0 && carried && nwarm > 0) {
        e = carry;
      } else if (k != k0) {  // (k0: the first warm-up block, or the stream's first block: entry 0)
        uint32_t spins = 0;
        while (!ready[k - 1] && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);  // (the bound: whatever happens, the kernel ends -- and the verify rounds see the rest)
        e = exits[k - 1];  // where the previous block's last run ended (relative to this block's start)
      }
      if (e >= RLE_BLK) {
        ex = e - RLE_BLK;
      } else if (e < limit) {
        const uint32_t v = tabs[wv][cur][e];
        const uint32_t to = v & 0xffffu;
        nv = v >> 16;
        ex = to > RLE_BLK ? to - RLE_BLK : 0u;
      }
    }
    if (lane == 0) {
      exits[k] = ex;
      ready[k] = 1;
      if (live && in_span && bweak) {
        blk.entry[b0g + sl] = e;
        blk.exit_[b0g + sl] = ex;
        blk.nvals[b0g + sl] = nv;
      }
    }
  }
  __syncthreads();  // (the next span takes the flags back)
  PROF_MARK(7);
}

#ifndef RLE_SHORT_MIN_WAVES
#define RLE_SHORT_MIN_WAVES 8  // (the kernel came to 66 registers: 7 wavefronts per SIMD, three workgroups per CU where its LDS allows four.  Held to 64 -- no spills --: C3's walk 0.46 -> 0.40 ms)
#endif
extern "C" __global__ void __launch_bounds__(64 * RLE_SHORT_WAVES, RLE_SHORT_MIN_WAVES) rle_walk_short_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars,
                                                                                          uint32_t total_blocks, uint32_t spans_per_wg) {
  __shared__ uint32_t tabs[RLE_SHORT_WAVES][2][RLE_BLK];
  __shared__ uint32_t exits[RLE_WARM + 64], ready[RLE_WARM + 64];
  __shared__ unsigned long long masks[8][2];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  // up to eight spans per workgroup (the host picks a power of two; block ranges of a job are
  // RLE_TILE aligned, so they belong to one stream): their flags are fetched together, spans with
  // few weak blocks cost nothing more.  Big batches of long-run streams have thousands of spans
  // with nothing to do -- fewer, larger workgroups keep their dispatch cheap; small batches of
  // short-run streams get one span per workgroup so that the spans run side by side.
  uint32_t bw8 = blockIdx.x * 64u * spans_per_wg;
  if (bw8 >= total_blocks) return;
  PROF_BEGIN();
  RleJob* j = const_cast<RleJob*>(find_job_by_block(jobs, njobs, bw8));
  uint64_t len = scalars[j->len_idx];
  if (tid < 64) {
#pragma unroll
    for (int s = 0; s < 8; s++) {
      uint32_t lb = bw8 - j->block0 + s * 64 + lane;
      const bool live = (uint32_t)s < spans_per_wg && lb < j->nblocks && ((uint64_t)lb * RLE_BLK < len || lb == 0);
      const bool weak = live && !blk.flags[bw8 + s * 64 + lane];
      const unsigned long long lm = __ballot(live), wm = __ballot(weak);
      if (lane == 0) {
        masks[s][0] = lm;
        masks[s][1] = wm;
      }
    }
  }
  __syncthreads();
  uint32_t carry = 0xffffffffu;  // the exit of the last block of the span before, when this workgroup has just walked it
  for (int s = 0; s < 8; s++) {
    const unsigned long long live_m = masks[s][0], weak_m = masks[s][1];
    // isolated weak blocks inside long-run streams are left to the relaxation rounds
    if (__builtin_popcountll(weak_m) < RLE_SHORT_MIN_WEAK) {
      carry = 0xffffffffu;
      continue;
    }
    uint32_t bw = bw8 + s * 64, lb0 = bw - j->block0;
    if (j->codec == CODEC_RLE2) short_span<CODEC_RLE2>(j, blk, len, lb0, bw, tabs, exits, ready, tid, live_m, weak_m, carry PROF_ARG);
    else if (j->codec == CODEC_RLE1) short_span<CODEC_RLE1>(j, blk, len, lb0, bw, tabs, exits, ready, tid, live_m, weak_m, carry PROF_ARG);
    else short_span<CODEC_BYTE>(j, blk, len, lb0, bw, tabs, exits, ready, tid, live_m, weak_m, carry PROF_ARG);
    // (behind short_span's closing barrier: the exits are final.  The span's last block must lie inside the stream)
    carry = (live_m >> 63) & 1 ? exits[(lb0 >= RLE_WARM ? RLE_WARM : lb0) + 63] : 0xffffffffu;
#ifdef ORC_PROF
    if (tid == 0) atomicAdd(&g_prof[42], 1ull);
#endif
    PROF_MARK(8);
  }
  PROF_END();
}

// Repair of whatever the relaxation rounds left inconsistent (long runs of irregular size never
// re-synchronise from a wrong guess).  One wavefront per stream walks the true chain from the
// first inconsistent block:
//   * runs of at least RLE_BLK bytes: STRIDE SPECULATION -- lane i of 4 x 64 candidates parses a
//     header at pos + i*s (s = size of the run at pos); a ballot finds the first candidate whose
//     size differs, every candidate before it is a proven header, so up to 256 runs are resolved
//     per memory round trip.  Each proven header owns its block (one header per block because
//     s >= RLE_BLK) and the pass-through blocks up to the next header;
//   * shorter runs: lane 0 walks the block serially (the relaxation rounds normally handle these).
// The walk stops early when it lands on a block whose stored entry is already right and no later
// block is inconsistent.
template <int CODEC>
__device__ __forceinline__ void repair_chain(RleJob* j, const RleBlocks& blk, uint64_t len, uint32_t nb, uint32_t lane) {
  const uint8_t* data = as_global(j->data);
  const bool is_signed = j->is_signed;
  const int nbits = j->nbits;
  const uint32_t b0 = j->block0;
  uint32_t long_streak = 0;
  uint32_t fill_from = j->first_bad;  // blocks below are final; [fill_from, block(pos)) are pass-through
  uint64_t pos = (uint64_t)fill_from * RLE_BLK + (fill_from == 0 ? 0u : blk.exit_[b0 + fill_from - 1]);
  for (;;) {
    uint32_t lb = pos < len ? (uint32_t)(pos / RLE_BLK) : nb;
    if (lb > nb) lb = nb;
    // pass-through blocks between the last finished block and the block of the next header
    for (uint32_t pb = fill_from + lane; pb < lb; pb += 64) {
      uint64_t e = pos - (uint64_t)pb * RLE_BLK;  // >= RLE_BLK except in the stream's last block
      blk.entry[b0 + pb] = (uint32_t)e;
      blk.exit_[b0 + pb] = e >= RLE_BLK ? (uint32_t)(e - RLE_BLK) : 0u;
      blk.nvals[b0 + pb] = 0;
    }
    if (lb >= nb) return;
    fill_from = lb;
    uint32_t want = (uint32_t)(pos - (uint64_t)lb * RLE_BLK);
    if (blk.entry[b0 + lb] == want) {
      // consistent here: the verify round left a bitmap of inconsistent blocks; find the next one
      // after lb (64 words = 2048 blocks per step).  Blocks this kernel repaired lie below lb.
      uint32_t next = nb;
      {
        uint32_t g0 = b0 + lb + 1, g1 = b0 + nb;  // global block range to search
        for (uint32_t wbase = g0 >> 5; wbase <= ((g1 - 1) >> 5) && g0 < g1; wbase += 64) {
          uint32_t wi = wbase + lane;
          uint32_t word = 0;
          if (wi <= ((g1 - 1) >> 5)) {
            word = blk.badmap[wi];
            if (wi == (g0 >> 5)) word &= ~0u << (g0 & 31);
          }
          unsigned long long m = __ballot(word != 0);
          if (m) {
            uint32_t l = (uint32_t)__builtin_ctzll(m);
            uint32_t wsel = __shfl(word, l);
            uint32_t g = ((wbase + l) << 5) + (uint32_t)__builtin_ctz(wsel);
            if (g < g1) next = g - b0;
            break;
          }
        }
      }
      if (next >= nb) return;
      fill_from = next;
      pos = (uint64_t)next * RLE_BLK + blk.exit_[b0 + next - 1];
      continue;
    }
    RunHdr h0;
    run_parse<CODEC, false>(data + pos, len - pos, is_signed, nbits, h0);
    const uint32_t s = h0.size;
    if (s >= RLE_BLK && !h0.err) {
      // ---- long run: isolated damage is repaired one run at a time; after 8 long runs in a row
      //      the walker switches to stride speculation over 256 candidates per memory round trip
      uint32_t first_fail = 1;
      uint32_t nv[4] = {h0.n, 0, 0, 0};
      if (long_streak >= 8) {
        first_fail = 256;
        for (int k = 0; k < 4; k++) {
          uint64_t c = pos + (uint64_t)(k * 64 + lane) * s;
          bool ok = false;
          nv[k] = 0;
          if (c < len) {
            RunHdr h;
            run_parse<CODEC, false>(data + c, len - c, is_signed, nbits, h);
            ok = !h.err && h.size == s;
            nv[k] = h.n;
          }
          unsigned long long bad = __ballot(!ok);
          if (bad && first_fail == 256) first_fail = k * 64 + (uint32_t)__builtin_ctzll(bad);
        }
        if (first_fail < 16) long_streak = 0;
      } else {
        long_streak++;
      }
      // candidates [0, first_fail) are proven headers of size s >= RLE_BLK: one header per block
      for (int k = 0; k < 4; k++) {
        uint32_t i = k * 64 + lane;
        if (i < first_fail) {
          uint64_t c = pos + (uint64_t)i * s;
          uint32_t hb = (uint32_t)(c / RLE_BLK);
          uint64_t nextpos = c + s;
          if (hb < nb) {
            uint64_t bend = (uint64_t)(hb + 1) * RLE_BLK;
            blk.entry[b0 + hb] = (uint32_t)(c - (uint64_t)hb * RLE_BLK);
            blk.nvals[b0 + hb] = nv[k];
            blk.exit_[b0 + hb] = nextpos > bend ? (uint32_t)(nextpos - bend) : 0u;
          }
          for (uint32_t pb = hb + 1; (uint64_t)(pb + 1) * RLE_BLK <= nextpos && pb < nb; pb++) {
            uint32_t e = (uint32_t)(nextpos - (uint64_t)pb * RLE_BLK);
            blk.entry[b0 + pb] = e;
            blk.exit_[b0 + pb] = e - RLE_BLK;
            blk.nvals[b0 + pb] = 0;
          }
        }
      }
      if (lane == 0) j->stat_repaired += first_fail;
      pos += (uint64_t)first_fail * s;
      uint64_t fb = pos / RLE_BLK;
      fill_from = fb < nb ? (uint32_t)fb : nb;
    } else {
      // ---- short (or failing) run: lane 0 walks the whole block ----
      uint32_t ex = 0, nvv = 0;
      if (lane == 0) {
        walk_block<CODEC>(data, len, lb, want, is_signed, nbits, &ex, &nvv);
        blk.entry[b0 + lb] = want;
        blk.exit_[b0 + lb] = ex;
        blk.nvals[b0 + lb] = nvv;
        j->stat_repaired += 1;
      }
      ex = __shfl(ex, 0);
      pos = (uint64_t)(lb + 1) * RLE_BLK + ex;
      fill_from = lb + 1;
      long_streak = 0;
    }
  }
}

// Inconsistent blocks mark short damaged stretches.  The typical cause: a writer flushes its encoder at every row-group boundary
// (10 000 rows), so every twentieth run of an otherwise regular stream is short, the stride of the run headers breaks there, and
// the few blocks around the boundary are guessed wrong while everything between boundaries is found and verified.  The stretches
// are mended all at once, one LANE each (rle_mend_kernel): from where the block before the stretch says the chain comes in, block
// after block, until the positions agree with what is recorded, the stretch's strong end or another lane's stretch is reached, or
// RLE_MEND_STEPS blocks have been walked.  No block is written by two lanes, so a block's entry, exit and value count always
// belong together -- the verify rounds compare exits with entries and rely on that.  A stretch whose start was itself wrong shows
// up in the verify round that follows (mode 5) and is taken again; what is left after the passes goes to the serial
// rle_repair_kernel, which is exact whatever happened before.  Streams with only a few inconsistent blocks skip all this
// (RLE_MEND_MIN): the serial kernel is quicker for them.
// (More passes do mend what two leave -- the adversarial stream of bench.py's c2-adv, whose spans' warm-ups rarely meet the true
// chain: 2 passes + the serial repair 23 ms, 12 passes 7.6, 24 passes 6.1 -- but every pass is two launches on every call; and the
// passes looped inside ONE workgroup per stream, a launch that ends at once for all other streams, took 37 ms: a single workgroup
// going over 55 000 blocks is bound by its own memory latency.  Such streams are what the ROW_INDEX positions are for: 0.56 ms.)
#define RLE_MEND_STEPS 64
#define RLE_MEND_NEAR 48u
extern "C" __global__ void __launch_bounds__(256) rle_mend_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars, uint32_t total_blocks) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t bw = b - lane;
  if (bw >= total_blocks) return;
  RleJob* j = const_cast<RleJob*>(find_job_by_block(jobs, njobs, bw));
  if (j->stat_bad < RLE_MEND_MIN) return;  // (the whole wavefront: block ranges of a job are tile aligned)
  const uint32_t lb = b - j->block0;
  const uint64_t len = scalars[j->len_idx];
  if (rle_hopeless(j, len)) {
    // an eighth of the stream's blocks inconsistent: no regular stream with a few damaged stretches -- the exact walk takes it
    if (lb == 0) j->bad_left = j->stat_bad;
    return;
  }
  if (lb >= j->nblocks || !((uint64_t)lb * RLE_BLK < len || lb == 0)) return;
  const uint32_t b0 = j->block0;
  auto flagged = [&](uint32_t pb) { return ((blk.badmap[(b0 + pb) >> 5] >> ((b0 + pb) & 31)) & 1u) != 0; };
  // A damaged stretch shows as several inconsistent blocks: its first block, seams between pieces of wrong chains inside it, and
  // at its end a STRONG block (a header the search verified: true, inconsistent only with the wrong block before it).  One lane
  // mends a stretch: the inconsistent block whose nearest inconsistent neighbour to the left (within RLE_MEND_NEAR blocks) is
  // strong, or that has none.
  if (!flagged(lb)) return;
  for (uint32_t k = 1; k <= RLE_MEND_NEAR && k <= lb; k++)
    if (flagged(lb - k)) {
      if (!blk.flags[b0 + lb - k]) return;  // part of the stretch that block belongs to
      break;
    }
  const uint8_t* data = as_global(j->data);
  uint32_t nb = (uint32_t)((len + RLE_BLK - 1) / RLE_BLK);
  if (nb > j->nblocks) nb = j->nblocks;
  uint64_t pos = (uint64_t)lb * RLE_BLK + (lb == 0 ? 0u : blk.exit_[b - 1]);
  uint32_t fill_from = lb, done = 0, last_flag = lb;
  // what to do at block pb: 0 go on (rewrite it), 1 stop before it: the stretch's strong end, or a block that HEADS a stretch of
  // its own by the test above -- its nearest inconsistent neighbour to the left (last_flag: every block between is visited, so it
  // is the nearest) is strong or more than RLE_MEND_NEAR blocks away.  Exactly the blocks another lane starts at: ownership of a
  // block is exclusive, a block's entry / exit / value count are always one lane's.
  auto ends_here = [&](uint32_t pb) -> bool {
    if (pb == lb || !flagged(pb)) return false;
    if (blk.flags[b0 + pb] || pb - last_flag > RLE_MEND_NEAR || blk.flags[b0 + last_flag]) return true;
    last_flag = pb;
    return false;
  };
  for (uint32_t it = 0; it < RLE_MEND_STEPS; it++) {
    uint32_t c = pos < len ? (uint32_t)(pos / RLE_BLK) : nb;
    if (c > nb) c = nb;
    bool met = false;
    for (uint32_t pb = fill_from; pb < c; pb++) {  // blocks the run before reaches over
      if (ends_here(pb)) {
        met = true;
        break;
      }
      const uint64_t e = pos - (uint64_t)pb * RLE_BLK;
      blk.entry[b0 + pb] = (uint32_t)e;
      blk.exit_[b0 + pb] = e >= RLE_BLK ? (uint32_t)(e - RLE_BLK) : 0u;
      blk.nvals[b0 + pb] = 0;
    }
    if (met || c >= nb || ends_here(c)) break;
    const uint32_t want = (uint32_t)(pos - (uint64_t)c * RLE_BLK);
    if (c != lb && !flagged(c) && blk.entry[b0 + c] == want) break;  // back on a recorded chain
    uint32_t ex, nv;
    walk_dispatch(j, data, len, c, want, &ex, &nv);
    blk.entry[b0 + c] = want;
    blk.exit_[b0 + c] = ex;
    blk.nvals[b0 + c] = nv;
    pos = (uint64_t)(c + 1) * RLE_BLK + ex;
    fill_from = c + 1;
    done++;
  }
  atomicAdd(&j->stat_repaired, done);
  if (lb == j->first_bad) {
    j->first_bad = 0xffffffffu;  // the verify round behind this kernel finds the first one again (the first inconsistent block of a stream always heads a stretch)
    j->bad_left = 0;             // ... and counts what is left
  }
}

extern "C" __global__ void __launch_bounds__(64) rle_repair_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars) {
  RleJob* j = &jobs[blockIdx.x];
  uint32_t lane = threadIdx.x;
  if (j->first_bad == 0xffffffffu) return;
  uint64_t len = scalars[j->len_idx];
  uint32_t nb = (uint32_t)((len + RLE_BLK - 1) / RLE_BLK);
  if (nb > j->nblocks) nb = j->nblocks;
  if (j->codec == CODEC_RLE2) repair_chain<CODEC_RLE2>(j, blk, len, nb, lane);
  else if (j->codec == CODEC_RLE1) repair_chain<CODEC_RLE1>(j, blk, len, nb, lane);
  else repair_chain<CODEC_BYTE>(j, blk, len, nb, lane);
}

// ---- streams the heuristics cannot settle: the EXACT walk, in parallel ------------------------------------------------------
// Long runs of ever-changing size (bench.py's c2-adv; a file written without ROW_INDEX positions): no stride to prove, no
// candidate that verifies, and a chain started at a wrong byte does not meet the true one within a warm-up -- what the mending
// passes leave used to go to rle_repair_kernel, one dependent memory access per run (23 ms per 24 M rows).  The exact answer
// without a serial chain: a SPAN of 256 blocks is a function from where a chain enters it to where that chain leaves it, and
// functions compose.
//   rle_exact_map_kernel    one workgroup per span: for EVERY entry e in [0, RLE_EXACT_E) -- a run is at most 512 eight-byte
//                           values, a patch list and a header long: no chain enters a span further in -- the exit of the chain
//                           from e, all 4608 chains at once, through the all-entries tables of the span's blocks (block_exit_table)
//   rle_exact_tile_kernel   the maps of a tile's four spans composed (a tile = RLE_TILE blocks never holds two streams)
//   rle_exact_chain_kernel  one wavefront per stream: the true chain from byte 0 through the tiles' maps (a lookup per tile:
//                           191 for 100 MB), then -- tiles side by side -- through the spans' maps: every span's true entry
//   rle_exact_fill_kernel   one workgroup per span: its blocks' entry / exit / value count from the span's entry
// Streams take part whose last mending pass left RLE_EXACT_MIN or more inconsistent blocks; all four kernels end at once for the
// others.  The stream's blocks are then exact by construction: rle_repair_kernel has nothing to do (first_bad = none).
#define RLE_EXACT_E 4608u     // bytes from a span's start within which a chain may enter it (RLE v2: 4 + 512 * 8 + 31 * 8 patch bytes)
#define RLE_EXACT_SPAN 256u   // blocks per span (RLE_TILE / 4)
#ifndef RLE_EXACT_WG
#define RLE_EXACT_WG 512
#endif
__device__ __forceinline__ bool exact_span(const RleJob* jobs, int njobs, const uint64_t* scalars, uint32_t g, uint32_t total_blocks, RleJob*& j,
                                           uint32_t& lb0, uint64_t& len) {
  const uint32_t bw = g * RLE_EXACT_SPAN;
  if (bw >= total_blocks) return false;
  j = const_cast<RleJob*>(find_job_by_block(jobs, njobs, bw));
  if (j->bad_left < RLE_EXACT_MIN) return false;
  lb0 = bw - j->block0;
  len = scalars[j->len_idx];
  return lb0 < j->nblocks && (uint64_t)lb0 * RLE_BLK < len;
}
template <int CODEC>
__device__ __forceinline__ void exact_tables(const RleJob* j, uint64_t len, uint32_t lbv, uint32_t (*tab)[RLE_BLK], int* which, uint32_t lane) {
  const bool live = lbv < j->nblocks && (uint64_t)lbv * RLE_BLK < len;
  int c = -1;
  if (live) block_exit_table<CODEC>(as_global(j->data), len, lbv, j->is_signed, j->nbits, tab, lane, c);
  if (lane == 0) *which = c;
}
__device__ __forceinline__ void exact_tables_any(const RleJob* j, uint64_t len, uint32_t lbv, uint32_t (*tab)[RLE_BLK], int* which, uint32_t lane) {
  if (j->codec == CODEC_RLE2) exact_tables<CODEC_RLE2>(j, len, lbv, tab, which, lane);
  else if (j->codec == CODEC_RLE1) exact_tables<CODEC_RLE1>(j, len, lbv, tab, which, lane);
  else exact_tables<CODEC_BYTE>(j, len, lbv, tab, which, lane);
}
extern "C" __global__ void __launch_bounds__(RLE_EXACT_WG) rle_exact_map_kernel(RleJob* jobs, int njobs, const uint64_t* scalars, uint32_t total_blocks, uint16_t* fmap) {
  __shared__ uint32_t tabs[8][2][RLE_BLK];
  __shared__ int which[8];
  __shared__ uint32_t cur[RLE_EXACT_E];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t n_spans = (total_blocks + RLE_EXACT_SPAN - 1) / RLE_EXACT_SPAN;
  for (uint32_t g = blockIdx.x; g < n_spans; g += gridDim.x) {
    RleJob* j;
    uint32_t lb0;
    uint64_t len;
    if (!exact_span(jobs, njobs, scalars, g, total_blocks, j, lb0, len)) continue;  // (the whole workgroup)
    __syncthreads();
    for (uint32_t e = tid; e < RLE_EXACT_E; e += RLE_EXACT_WG) cur[e] = e;
    for (uint32_t k0 = 0; k0 < RLE_EXACT_SPAN; k0 += 8) {
      exact_tables_any(j, len, lb0 + k0 + wv, tabs[wv], &which[wv], lane);
      __syncthreads();
      for (uint32_t e = tid; e < RLE_EXACT_E; e += RLE_EXACT_WG) {
        uint32_t p = cur[e];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
          if ((p >> 9) == k0 + q) {
            const int c = which[q];
            // (behind the stream's end: the chain is over; its exit is of no interest to anybody)
            p = c < 0 ? RLE_EXACT_SPAN * RLE_BLK : (k0 + q) * RLE_BLK + (tabs[q][c][p & (RLE_BLK - 1)] & 0xffffu);
          }
        cur[e] = p;
      }
      __syncthreads();
    }
    for (uint32_t e = tid; e < RLE_EXACT_E; e += RLE_EXACT_WG) {
      const uint32_t p = cur[e];
      const uint32_t x = p >= RLE_EXACT_SPAN * RLE_BLK ? p - RLE_EXACT_SPAN * RLE_BLK : 0u;
      fmap[(size_t)g * RLE_EXACT_E + e] = (uint16_t)(x < RLE_EXACT_E ? x : RLE_EXACT_E - 1);
    }
  }
}
extern "C" __global__ void __launch_bounds__(RLE_EXACT_WG) rle_exact_tile_kernel(RleJob* jobs, int njobs, const uint64_t* scalars, uint32_t total_blocks, const uint16_t* fmap,
                                                                                 uint16_t* gmap) {
  const uint32_t n_tiles = total_blocks / RLE_TILE;
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    RleJob* j;
    uint32_t lb0;
    uint64_t len;
    if (!exact_span(jobs, njobs, scalars, t * 4, total_blocks, j, lb0, len)) continue;
    for (uint32_t e = threadIdx.x; e < RLE_EXACT_E; e += RLE_EXACT_WG) {
      uint32_t x = e;
      for (uint32_t s = 0; s < 4; s++) {
        const uint32_t lbs = lb0 + s * RLE_EXACT_SPAN;
        if (lbs < j->nblocks && (uint64_t)lbs * RLE_BLK < len) x = fmap[(size_t)(t * 4 + s) * RLE_EXACT_E + x];
      }
      gmap[(size_t)t * RLE_EXACT_E + e] = (uint16_t)x;
    }
  }
}
extern "C" __global__ void __launch_bounds__(64) rle_exact_chain_kernel(RleJob* jobs, int njobs, const uint64_t* scalars, const uint16_t* fmap, const uint16_t* gmap,
                                                                        uint32_t* tile_entry, uint32_t* span_entry) {
  RleJob* j = &jobs[blockIdx.x];
  if (j->bad_left < RLE_EXACT_MIN) return;
  const uint64_t len = scalars[j->len_idx];
  uint32_t nb = (uint32_t)((len + RLE_BLK - 1) / RLE_BLK);
  if (nb > j->nblocks) nb = j->nblocks;
  const uint32_t t0 = j->block0 / RLE_TILE, nt = (nb + RLE_TILE - 1) / RLE_TILE;
  const uint32_t lane = threadIdx.x;
  if (lane == 0) {
    uint32_t x = 0;  // (a stream starts with a run header)
    for (uint32_t t = 0; t < nt; t++) {
      tile_entry[t0 + t] = x;
      x = gmap[(size_t)(t0 + t) * RLE_EXACT_E + x];
    }
    __threadfence();
  }
  wave_sync_scan();
  for (uint32_t t = lane; t < nt; t += 64) {
    uint32_t x = __hip_atomic_load(&tile_entry[t0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t s = 0; s < 4; s++) {
      const uint32_t g = (t0 + t) * 4 + s, lbs = (t * 4 + s) * RLE_EXACT_SPAN;
      span_entry[g] = x;
      if (lbs < nb) x = fmap[(size_t)g * RLE_EXACT_E + x];
    }
  }
  if (lane == 0) j->first_bad = 0xffffffffu;  // exact from here on: nothing for rle_repair_kernel
}
extern "C" __global__ void __launch_bounds__(RLE_EXACT_WG) rle_exact_fill_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars, uint32_t total_blocks,
                                                                                 const uint32_t* span_entry) {
  __shared__ uint32_t tabs[8][2][RLE_BLK];
  __shared__ int which[8];
  __shared__ uint32_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t n_spans = (total_blocks + RLE_EXACT_SPAN - 1) / RLE_EXACT_SPAN;
  for (uint32_t g = blockIdx.x; g < n_spans; g += gridDim.x) {
    RleJob* j;
    uint32_t lb0;
    uint64_t len;
    if (!exact_span(jobs, njobs, scalars, g, total_blocks, j, lb0, len)) continue;
    __syncthreads();
    if (tid == 0) carry = span_entry[g];
    for (uint32_t k0 = 0; k0 < RLE_EXACT_SPAN; k0 += 8) {
      exact_tables_any(j, len, lb0 + k0 + wv, tabs[wv], &which[wv], lane);
      __syncthreads();
      if (tid == 0) {
        uint32_t e = carry;  // where the chain stands, relative to the start of block k0
        for (uint32_t q = 0; q < 8; q++) {
          const int c = which[q];
          if (c < 0) break;  // behind the stream's end
          const uint32_t b = j->block0 + lb0 + k0 + q;
          uint32_t ex = 0, nv = 0;
          if (e >= RLE_BLK) {
            ex = e - RLE_BLK;  // a run reaches over the block
          } else {
            const uint32_t v = tabs[q][c][e];
            const uint32_t to = v & 0xffffu;
            nv = v >> 16;
            ex = to > RLE_BLK ? to - RLE_BLK : 0u;
          }
          blk.entry[b] = e;
          blk.exit_[b] = ex;
          blk.nvals[b] = nv;
          e = ex;
        }
        carry = e;
      }
      __syncthreads();
    }
  }
}

// Exclusive scan of nvals inside each RLE_TILE-block tile (one workgroup per tile).
extern "C" __global__ void __launch_bounds__(256) rle_tile_scan_kernel(RleBlocks blk, uint32_t* tile_sum) {
  __shared__ uint32_t wsum[4];
  uint32_t tile = blockIdx.x;
  uint32_t base = tile * RLE_TILE + threadIdx.x * 4;
  uint4 v = *reinterpret_cast<const uint4*>(blk.nvals + base);
  uint32_t s = v.x + v.y + v.z + v.w;
  uint32_t incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
  uint32_t excl = wbase + incl - s;
  uint4 o;
  o.x = excl;
  o.y = excl + v.x;
  o.z = o.y + v.y;
  o.w = o.z + v.z;
  *reinterpret_cast<uint4*>(blk.voff + base) = o;
  if (threadIdx.x == 255) tile_sum[tile] = excl + s;
}

// Per job: exclusive scan of its tile sums -> tile_base, total -> scalars[total_idx]; group -> job table.
extern "C" __global__ void __launch_bounds__(256) rle_job_scan_kernel(const RleJob* jobs, RleBlocks blk, const uint32_t* tile_sum,
                                                                       uint64_t* scalars) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  const RleJob* j = &jobs[blockIdx.x];
  uint32_t t0 = j->block0 / RLE_TILE;
  uint32_t nt = (j->nblocks + RLE_TILE - 1) / RLE_TILE;
  // group -> job table of the expansion (saves every expansion wavefront a search over the jobs)
  for (uint32_t g = threadIdx.x; g < j->ngroups; g += 256) blk.group_job[j->group_tab0 + g] = j->class_index;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t s = 0; s < nt; s += 256) {
    uint32_t i = s + threadIdx.x;
    uint64_t v = i < nt ? tile_sum[t0 + i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    if (i < nt) blk.tile_base[t0 + i] = (uint32_t)(wbase + incl - v);
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) scalars[j->total_idx] = carry_s;
}
