// rle_scan.hip -- pass 1 of the RLE decoders: find run boundaries in parallel.
//
// Replaces the serial header walk of the reference (rle_v2/mod.rs:112-146 decode_batch called
// run after run from rle.rs:68-107; rle_v1.rs:143-159; byte.rs:228-247).  Run boundaries are
// data dependent, so the stream is cut into RLE_BLK-byte blocks and one LANE walks one block:
//
//   round 0   every block guesses where its first header is (a multiple of the stream's first
//             run size -- exact for the long regular runs ORC writers emit) and walks its runs;
//   round r   block b takes exit[b-1] as its entry and re-walks only if that differs from the
//             entry it used.  Short/irregular runs re-synchronise within a few hops, so the
//             iteration converges in 2-4 rounds (chaotic relaxation: reading a neighbour's old
//             or new exit are both fine, the fixed point is unique);
//   verify    records the first inconsistent block per stream; rle_repair then fixes what is
//             left with a serial walk (irregular long runs), so the result is always exact.
//
// Traffic: only header bytes are read (payload is skipped); per block 16 B of state.
#include "rle_kernels.h"
#include "rle_parse.h"

__device__ __forceinline__ const RleJob* find_job_by_block(const RleJob* jobs, int njobs, uint32_t b) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= b) lo = mid;
    else hi = mid - 1;
  }
  return &jobs[lo];
}

template <int CODEC>
__device__ __forceinline__ void walk_block(const uint8_t* data, uint64_t len, uint32_t lb, uint32_t entry, bool is_signed,
                                           int nbits, uint32_t* exit_out, uint32_t* nvals_out) {
  uint64_t bend = (uint64_t)(lb + 1) * RLE_BLK;
  uint64_t end = bend < len ? bend : len;
  uint64_t pos = (uint64_t)lb * RLE_BLK + entry;
  uint32_t nv = 0;
  while (pos < end) {
    RunHdr h;
    run_parse<CODEC, false>(data + pos, len - pos, is_signed, nbits, h);
    nv += h.n;
    pos += h.size;
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

// mode 0: first round (stride guess); 1: relaxation round; 2: verify only
extern "C" __global__ void __launch_bounds__(256) rle_walk_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars,
                                                                   uint32_t total_blocks, int mode) {
  uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b >= total_blocks) return;
  RleJob* j = const_cast<RleJob*>(find_job_by_block(jobs, njobs, b));
  uint32_t lb = b - j->block0;
  if (lb >= j->nblocks) return;
  uint64_t len = scalars[j->len_idx];
  if ((uint64_t)lb * RLE_BLK >= len && lb != 0) {
    if (mode == 0) {
      blk.entry[b] = RLE_BLK;
      blk.exit_[b] = 0;
      blk.nvals[b] = 0;
    }
    return;
  }
  const uint8_t* data = j->data;
  uint32_t want;
  if (lb == 0) {
    want = 0;
  } else if (mode == 0) {
    // stride guess from the stream's first run
    RunHdr h;
    if (j->codec == CODEC_RLE2) run_parse<CODEC_RLE2, false>(data, len, j->is_signed, j->nbits, h);
    else if (j->codec == CODEC_RLE1) run_parse<CODEC_RLE1, false>(data, len, j->is_signed, j->nbits, h);
    else run_parse<CODEC_BYTE, false>(data, len, false, 8, h);
    uint32_t s0 = h.size ? h.size : 1;
    uint32_t r = (uint32_t)(((uint64_t)lb * RLE_BLK) % s0);
    want = r ? s0 - r : 0;
  } else {
    want = blk.exit_[b - 1];
  }
  if (mode != 0 && want == blk.entry[b]) return;
  if (mode == 2) {
    atomicMin(&j->first_bad, lb);
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
}

// Serial repair of whatever the relaxation rounds left inconsistent: one wavefront per job,
// lane 0 walks, all lanes help re-verifying 64 blocks at a time after a re-synchronisation.
extern "C" __global__ void __launch_bounds__(64) rle_repair_kernel(RleJob* jobs, int njobs, RleBlocks blk, const uint64_t* scalars) {
  RleJob* j = &jobs[blockIdx.x];
  uint32_t lane = threadIdx.x;
  uint32_t bad = j->first_bad;
  if (bad == 0xffffffffu) return;
  uint64_t len = scalars[j->len_idx];
  uint32_t nb = (uint32_t)((len + RLE_BLK - 1) / RLE_BLK);
  if (nb > j->nblocks) nb = j->nblocks;
  const uint8_t* data = j->data;
  // Everything this kernel READS from the block arrays was written by earlier launches: the
  // exit of a block repaired here is carried in a register, never re-read.
  uint32_t lb = bad;
  uint32_t b0 = j->block0;
  uint32_t prev_exit = lb == 0 ? 0u : blk.exit_[b0 + lb - 1];
  while (lb < nb) {
    uint32_t b = b0 + lb;
    uint32_t want = lb == 0 ? 0u : prev_exit;
    if (want != blk.entry[b]) {
      uint32_t ex = 0, nv = 0;
      if (lane == 0) {
        if (want >= RLE_BLK) {
          ex = want - RLE_BLK;
        } else {
          walk_dispatch(j, data, len, lb, want, &ex, &nv);
        }
        blk.entry[b] = want;
        blk.exit_[b] = ex;
        blk.nvals[b] = nv;
      }
      prev_exit = __shfl(ex, 0);
      lb++;
      continue;
    }
    // consistent here: look for the next inconsistent block, 64 at a time
    uint32_t next = nb;
    for (uint32_t s = lb + 1; s < nb; s += 64) {
      uint32_t c = s + lane;
      bool mism = false;
      if (c < nb) mism = blk.exit_[b0 + c - 1] != blk.entry[b0 + c];
      unsigned long long m = __ballot(mism);
      if (m) {
        next = s + (uint32_t)__builtin_ctzll(m);
        break;
      }
    }
    lb = next;
    if (lb < nb) prev_exit = blk.exit_[b0 + lb - 1];
  }
  if (lane == 0) j->first_bad = 0xffffffffu;
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

// Per job: exclusive scan of its tile sums -> tile_base, total -> scalars[total_idx].
extern "C" __global__ void __launch_bounds__(256) rle_job_scan_kernel(const RleJob* jobs, RleBlocks blk, const uint32_t* tile_sum,
                                                                       uint64_t* scalars) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  const RleJob* j = &jobs[blockIdx.x];
  uint32_t t0 = j->block0 / RLE_TILE;
  uint32_t nt = (j->nblocks + RLE_TILE - 1) / RLE_TILE;
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
