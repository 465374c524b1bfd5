// rle_kernels.h -- device data structures shared by the RLE kernels and the host planner.
#pragma once
#include <stdint.h>

// A stream is cut into fixed blocks of RLE_BLK bytes.  One lane owns one block in the block
// walk; one wavefront owns `group_size` consecutive blocks in the expansion.
#define RLE_BLK 512u
#define RLE_TILE 1024u          // blocks per scan tile; every job's block range is tile aligned
#define RLE_NO_ERR 0xffffffffffffffffull

struct RleJob {
  const uint8_t* data;     // plain stream bytes in HBM (>= ORC_PAD bytes of slack behind them)
  void* out;               // dense output, `out_bytes` per value
  uint32_t len_idx;        // scalars[len_idx]    = stream length in bytes
  uint32_t needed_idx;     // scalars[needed_idx] = number of values the column consumes
  uint32_t total_idx;      // scalars[total_idx]  <- values present in the stream (written by the scan)
  uint32_t block0;         // first global block (multiple of RLE_TILE)
  uint32_t nblocks;        // upper bound of blocks (>= 1)
  uint32_t group0;         // first global expansion group
  uint32_t ngroups;
  uint32_t group_size;     // blocks per wavefront in the expansion (1..64)
  uint32_t group_tab0;     // index of this job's first group in RleBlocks::group_job
  uint32_t class_index;    // index of the job inside its class (what group_job holds)
  uint8_t codec;           // CODEC_*
  uint8_t is_signed;
  uint8_t nbits;           // width of the reference's NInt (8 for byte RLE)
  uint8_t out_bytes;       // 1, 2, 4 or 8
  uint32_t first_bad;      // first block whose entry disagrees with its predecessor (verify round)
  uint32_t stat_bad;       // verify round: number of inconsistent blocks (diagnostics)
  uint32_t stat_repaired;  // repair kernel: blocks rewritten (diagnostics)
  uint32_t skip;           // a stream entered at a row group (orcgpu_stream::skip_values): values of its first run that belong to
                           // the rows before; they are decoded in front of the column's values (the consumers start behind them)
  uint32_t bad_left;       // inconsistent blocks the last mending pass left (rle_walk_kernel mode 5): RLE_EXACT_MIN or more: the exact parallel walk
  // verified run starts (orcgpu_stream::entries): this job's slice of the call's RleHint table; hint_bad is set by
  // rle_hint_kernel when the entries do not lie on one run chain (they are then ignored)
  uint32_t hint0, n_hints, hint_bad, hint_skip;
  // a stream that is expected to hold ONE value throughout (the SECONDARY stream of a Decimal column: every value's scale, which every
  // writer makes the column's): scalars[uniform_idx] is set by rle2_uniform_kernel when the stream's bytes say so -- the expansion then
  // skips the job and the consumer takes `uniform_value` for every value; 0: no such check
  uint32_t uniform_idx, uniform_value;
  unsigned long long err;  // min over (first value index of the failing run << 8 | ORC_E_*)
  unsigned long long err_pos;  // min over (stream position of the failing run << 8 | ORC_E_*): its code is the first failure's
};

// One verified run start: the chunk of the stream it lies in (index within the call's chunk table; ~0: the stream is not
// compressed) and the byte inside that chunk's plain bytes
struct RleHint {
  uint32_t job, chunk, byte, pad;
};

struct RleBlocks {
  uint32_t* entry;      // offset of the first run header in the block (>= RLE_BLK: none), as used by the last walk
  uint32_t* exit_;      // offset into the NEXT block of the first header after this block's runs
  uint32_t* nvals;      // values produced by runs that START in the block
  uint32_t* voff;       // exclusive prefix of nvals inside the block's scan tile
  uint32_t* tile_base;  // per tile: values before the tile (within the job)
  uint8_t* flags;       // 1 = strong: entry verified by the candidate search (or filled by a strong owner)
  uint32_t* badmap;     // 1 bit per block: inconsistent at the verify round (zeroed every call)
  uint32_t* group_job;  // per expansion group: class-relative job index (one table per job class, back to back)
  uint32_t* hint;       // per block: its entry as the verified run starts give it (~0: none); null when no stream of the call has any
};
