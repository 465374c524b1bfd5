// string_kernels.hip -- String / Binary / Decimal finishers.
//
//   GenericByteArrayDecoder::next_byte_batch      array_decoder/string.rs:111-153
//     lengths (unsigned RLE, spaced, 0 at nulls) -> per-batch int32 offsets restarting at 0,
//     sum > i32::MAX -> OffsetOverflow, values = the next `sum` bytes of DATA, Utf8 validated
//   DictionaryStringArrayDecoder::next_batch      array_decoder/string.rs:204-224
//     keys (unsigned RLE, spaced) bounds-checked (DictionaryArray::try_new), cast to Utf8 =
//     per-row gather of the dictionary entry, null rows get an empty string
//   UnboundedVarintStreamDecoder + fix_i128_scale encoding/decimal.rs:28-52, array_decoder/decimal.rs:138-166
#include "rle_parse.h"

__device__ __forceinline__ void report_err64(unsigned long long* err, uint64_t idx, uint32_t code) {
  atomicMin(err, ((unsigned long long)idx << 8) | code);
}

// ---- direct strings: decoded lengths -> per-row int32 lengths -----------------------------------------
// dense: int64 lengths of the non-null rows.  GenericByteArrayDecoder::next_byte_batch (string.rs) first sums the
// batch's lengths as they are (i64, negative ones included) and raises OffsetOverflow when the sum exceeds
// i32::MAX; only then do negative lengths surface as an Arrow error.  A length outside 0..=i32::MAX is stored as 0
// and added to the batch's correction term `corr[b]`, so that batch_offsets_kernel sees the exact signed sum.
__device__ __forceinline__ void string_lens_body(const int64_t* dense, const unsigned long long* vbits, const uint32_t* rank,
                                                                      int32_t* lens, uint64_t n_rows, uint32_t batch, unsigned long long* corr,
                                                                      unsigned long long* err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  uint64_t d = i;
  bool valid = true;
  if (vbits) {
    unsigned long long word = vbits[i >> 6];
    uint32_t bit = i & 63;
    valid = (word >> bit) & 1;
    d = (uint64_t)rank[i >> 6] + __builtin_popcountll(word & ((1ull << bit) - 1));
  }
  int32_t len = 0;
  if (valid) {
    int64_t v = dense[d];
    if (v < 0 || v > 0x7fffffffll) {
      atomicAdd(&corr[i / batch], (unsigned long long)v);
      // (reported at the batch's first row, where the batch-level checks report too: the lower code wins a tie, which is
      //  the reference's order -- OffsetOverflow, negative length, values past the DATA stream)
      if (v < 0) report_err64(err, i / batch * batch, ORC_E_ARROW);
    } else {
      len = (int32_t)v;
    }
  }
  lens[i] = len;
}

// ---- per-batch exclusive scan of the lengths -> offsets (restart at 0 per batch) -------------------
// One workgroup per batch.  offsets layout: batch b at b * (batch + 1).
// corr: per-batch sum of the lengths string_lens_kernel stored as 0 (see there).
__device__ __forceinline__ void batch_offsets_body(const int32_t* lens, uint64_t n_rows, uint32_t batch, int32_t* offsets,
                                                                        unsigned long long* chartot, const unsigned long long* corr,
                                                                        unsigned long long* err, uint32_t ovf_code) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  uint64_t b = blockIdx.x;
  uint64_t row0 = b * batch;
  uint64_t rows = n_rows - row0 < batch ? n_rows - row0 : batch;
  int32_t* out = offsets + b * ((uint64_t)batch + 1);
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t s = 0; s < rows; s += 256) {
    uint64_t i = s + threadIdx.x;
    uint64_t v = i < rows ? (uint64_t)(uint32_t)lens[row0 + i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    if (i < rows) out[i] = (int32_t)(wbase + incl - v);
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[rows] = (int32_t)carry_s;
    chartot[b] = carry_s;
    const long long total = (long long)(carry_s + corr[b]);  // i64 sum as the reference forms it (wrapping)
    if (total > 0x7fffffffll) report_err64(err, row0, ovf_code);
    // huge lengths that wrap the sum below zero: `total as usize` bytes are then asked of DATA, which runs dry
    else if (total < 0) report_err64(err, row0, ORC_E_ARROW | ORC_E_EOF);
  }
}

// exclusive scan of the per-batch totals (single workgroup) -> charbase[b]; grand total -> *total
__device__ __forceinline__ void batch_base_body(const unsigned long long* chartot, unsigned long long* charbase, uint32_t n_batches,
                                                                     uint64_t* total) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t s = 0; s < n_batches; s += 256) {
    uint32_t i = s + threadIdx.x;
    uint64_t v = i < n_batches ? chartot[i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    if (i < n_batches) charbase[i] = wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

// ---- dictionary-encoded strings, all columns of a call in one launch ------------------------------------------
// DictionaryStringArrayDecoder::next_batch (array_decoder/string.rs:204-224) per batch: keys
// (bounds-checked) -> lengths -> int32 offsets restarting at 0 -> gather of the entries.
struct DictJob {
  const int32_t* dense;             // decoded keys of the non-null rows (-1: not a possible key, see store_val)
  const unsigned long long* vbits;  // stripe-wide validity words (null: no PRESENT stream)
  const uint32_t* rank;             // non-null rows before each 64-row word
  const int32_t* doff;              // dictionary offsets (dict_n + 1)
  const uint8_t* dbytes;            // dictionary bytes
  int32_t* offsets;                 // out: batch b at b * (batch + 1)
  unsigned long long* chartot;      // per-batch byte totals, then (at + n_batches) their exclusive scan
  unsigned long long* err;
  const unsigned long long* dict_err;  // error word of the dictionary itself (bad lengths / short blob): nothing of it may be touched then
  uint8_t* out_chars;               // value bytes of the column (set for the gather launch)
  uint64_t* total_out;              // scalar receiving the column's total value bytes
  uint64_t n_rows;
  uint32_t batch, n_batches;
  uint32_t dict_n_idx;              // scalar holding the dictionary size
  uint32_t pad;
};
__device__ __forceinline__ DictJob dict_job(const DictJob* jobs, uint32_t i) {
  DictJob j = jobs[i];  // every pointer in it is device global memory: see as_global()
  j.dense = glob(j.dense);
  j.vbits = glob(j.vbits);
  j.rank = glob(j.rank);
  j.doff = glob(j.doff);
  j.dbytes = glob(j.dbytes);
  j.offsets = glob(j.offsets);
  j.chartot = glob(j.chartot);
  j.err = glob(j.err);
  j.dict_err = glob(j.dict_err);
  j.out_chars = glob(j.out_chars);
  j.total_out = glob(j.total_out);
  return j;
}
#ifndef DICT_TILE
#define DICT_TILE 4096u   // rows per tile (a batch takes several): 256 threads x DICT_PER consecutive rows each
#endif
#define DICT_PER (DICT_TILE / 256u)
#define DICT_DOFF_LDS 2048u

// One workgroup per (batch, column): keys -> lengths -> offsets.  Rows are read coalesced into an
// LDS tile (padded: thread t then scans entries 32t..32t+31 without bank conflicts), one block scan
// of the 256 partial sums, offsets written back coalesced.
extern "C" __global__ void __launch_bounds__(256) dict_rows_kernel(const DictJob* jobs, const uint64_t* scalars) {
  __shared__ uint32_t lens[DICT_TILE + 256];
  __shared__ int32_t doffc[DICT_DOFF_LDS + 1];
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t tbase[256];
  const DictJob j = dict_job(jobs, blockIdx.y);
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  if (b >= j.n_batches) return;
  // a dictionary that failed its own checks (negative / overflowing lengths, blob shorter than their sum) fails the
  // column at construction; its offsets are not to be trusted: every row gets an empty string, nothing is gathered
  const bool dict_ok = *j.dict_err == RLE_NO_ERR;
  const uint64_t dict_n = dict_ok ? scalars[j.dict_n_idx] : 0;
  const bool cached = dict_n <= DICT_DOFF_LDS;
  if (cached)
    for (uint32_t i = tid; i <= dict_n; i += 256) doffc[i] = j.doff[i];
  __syncthreads();
  const uint64_t row0 = (uint64_t)b * j.batch;
  const uint64_t rows = j.n_rows - row0 < j.batch ? j.n_rows - row0 : j.batch;
  int32_t* out = j.offsets + (uint64_t)b * ((uint64_t)j.batch + 1);
  uint64_t carry = 0;
  for (uint64_t t0 = 0; t0 < rows; t0 += DICT_TILE) {
    const uint32_t tn = rows - t0 < DICT_TILE ? (uint32_t)(rows - t0) : DICT_TILE;
    // 1. keys and lengths of the tile: eight rows per trip, the loads of all eight issued together
    //    (validity word + rank, then the key) -- the chain of dependent loads is what this step costs
    for (uint32_t k0 = tid; k0 < DICT_TILE; k0 += 256 * 8) {
      unsigned long long word[8];
      uint32_t rk[8];
      int32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t k = k0 + u * 256;
        const uint64_t i = row0 + t0 + (k < tn ? k : 0);
        word[u] = j.vbits ? j.vbits[i >> 6] : ~0ull;
        rk[u] = j.vbits ? j.rank[i >> 6] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t k = k0 + u * 256;
        const uint64_t i = row0 + t0 + (k < tn ? k : 0);
        const uint32_t bit = i & 63;
        const bool valid = k < tn && ((word[u] >> bit) & 1);
        const uint64_t di = j.vbits ? (uint64_t)rk[u] + __builtin_popcountll(word[u] & ((1ull << bit) - 1)) : i;
        v[u] = j.dense[valid ? di : 0];
        if (!valid) v[u] = -1;  // null row (or behind the tile): no key
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t k = k0 + u * 256;
        const uint64_t i = row0 + t0 + k;
        const bool valid = k < tn && ((word[u] >> (i & 63)) & 1);
        uint32_t len = 0;
        if (valid) {
          if (!dict_ok) {
            // (the dictionary error is what the column reports)
          } else if (v[u] < 0 || (uint64_t)v[u] >= dict_n) {
            report_err64(j.err, i, ORC_E_ARROW);
          } else {
            len = cached ? (uint32_t)(doffc[v[u] + 1] - doffc[v[u]]) : (uint32_t)(j.doff[v[u] + 1] - j.doff[v[u]]);
          }
        }
        lens[k + k / DICT_PER] = len;
      }
    }
    __syncthreads();
    // 2. thread-local exclusive scan of DICT_PER consecutive entries
    uint64_t run = 0;
    {
      const uint32_t base = tid * (DICT_PER + 1);
      for (uint32_t m = 0; m < DICT_PER; m++) {
        const uint32_t l = lens[base + m];
        lens[base + m] = (uint32_t)run;  // < 2^32: checked against i32::MAX per batch below
        run += l;
      }
    }
    // 3. block scan of the 256 sums
    uint64_t incl = run;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry;
    for (uint32_t w = 0; w < (tid >> 6); w++) wbase += wsum[w];
    tbase[tid] = wbase + incl - run;
    const uint64_t tile_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    // 4. offsets, coalesced
    for (uint32_t k = tid; k < tn; k += 256) out[t0 + k] = (int32_t)(tbase[k / DICT_PER] + lens[k + k / DICT_PER]);
    carry += tile_total;
    __syncthreads();
  }
  if (tid == 0) {
    out[rows] = (int32_t)carry;
    j.chartot[b] = carry;
    if (carry > 0x7fffffffull) report_err64(j.err, row0, ORC_E_ARROW);
  }
}

// One workgroup per column: exclusive scan of the per-batch totals -> chartot[n_batches + b]; grand total.
extern "C" __global__ void __launch_bounds__(256) dict_base_kernel(const DictJob* jobs) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  const DictJob j = dict_job(jobs, blockIdx.x);
  unsigned long long* charbase = j.chartot + j.n_batches;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t s = 0; s < j.n_batches; s += 256) {
    uint32_t i = s + threadIdx.x;
    uint64_t v = i < j.n_batches ? j.chartot[i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    if (i < j.n_batches) charbase[i] = wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *j.total_out = carry_s;
}

// One workgroup per (batch, column): the batch's value bytes are assembled in LDS (every thread
// copies the entries of its rows to their offsets) and written to HBM as one coalesced run; batches
// whose bytes do not fit go row by row straight to memory.
#ifndef DICT_CHARS_LDS
#define DICT_CHARS_LDS 16384u
#endif
#define DICT_BYTES_LDS 8192u
extern "C" __global__ void __launch_bounds__(256) dict_gather2_kernel(const DictJob* jobs, const uint64_t* scalars) {
  __shared__ __attribute__((aligned(16))) uint8_t chars[DICT_CHARS_LDS + 16];
  __shared__ int32_t doffc[DICT_DOFF_LDS + 1];
  __shared__ uint8_t dbc[DICT_BYTES_LDS];
  const DictJob j = dict_job(jobs, blockIdx.y);
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  if (b >= j.n_batches) return;
  const uint64_t total = j.chartot[b];
  if (!total || !j.out_chars || *j.dict_err != RLE_NO_ERR) return;
  const uint64_t row0 = (uint64_t)b * j.batch;
  const uint32_t rows = (uint32_t)(j.n_rows - row0 < j.batch ? j.n_rows - row0 : j.batch);
  const int32_t* off = j.offsets + (uint64_t)b * ((uint64_t)j.batch + 1);
  uint8_t* out = j.out_chars + j.chartot[j.n_batches + b];
  // small dictionaries are copied to LDS once per workgroup (offsets and bytes)
  const uint64_t dict_n = scalars[j.dict_n_idx];
  const uint32_t dict_bytes = dict_n <= DICT_DOFF_LDS ? (uint32_t)j.doff[dict_n] : 0xffffffffu;
  const bool dcached = dict_n <= DICT_DOFF_LDS && dict_bytes <= DICT_BYTES_LDS;
  __shared__ uint32_t maxlen_s;
  if (tid == 0) maxlen_s = 0;
  __syncthreads();
  if (dcached) {
    uint32_t ml = 0;
    for (uint32_t i = tid; i <= dict_n; i += 256) {
      const int32_t o = j.doff[i];
      doffc[i] = o;
      if (i < dict_n) {
        const uint32_t l = (uint32_t)(j.doff[i + 1] - o);
        ml = l > ml ? l : ml;
      }
    }
    for (uint32_t i = tid; i < dict_bytes; i += 256) dbc[i] = j.dbytes[i];
    atomicMax(&maxlen_s, ml);
    __syncthreads();
  }
  // The batch's bytes are assembled in LDS a TILE of rows at a time (as many rows as are certain to fit: the longest entry
  // bounds a row) and written out as coalesced runs; a dictionary too big for LDS goes row by row straight to memory.
  const uint32_t maxlen = maxlen_s;
  const bool staged = dcached && maxlen > 0 && maxlen <= DICT_CHARS_LDS / 1024;  // (longer entries: few rows per tile, the direct path is quicker)
  if (!staged) {
    for (uint32_t k0 = tid; k0 < rows; k0 += 256 * 4) {
      // four rows per trip: offsets, validity words and ranks of all four are requested before any is used, then the keys,
      // then the dictionary offsets -- four chains of dependent loads side by side
      uint32_t o[4], e[4], so[4];
      int32_t key[4];
      unsigned long long word[4];
      uint32_t rk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + u * 256 < rows ? k0 + u * 256 : k0;
        o[u] = (uint32_t)off[k];
        e[u] = (uint32_t)off[k + 1];
        word[u] = j.vbits ? j.vbits[(row0 + k) >> 6] : ~0ull;
        rk[u] = j.vbits ? j.rank[(row0 + k) >> 6] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = k0 + u * 256 < rows ? k0 + u * 256 : k0;
        const uint64_t i = row0 + k;
        const uint64_t di = j.vbits ? (uint64_t)rk[u] + __builtin_popcountll(word[u] & ((1ull << (i & 63)) - 1)) : i;
        key[u] = e[u] != o[u] ? j.dense[di] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) so[u] = dcached ? (uint32_t)doffc[key[u]] : (uint32_t)j.doff[key[u]];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k0 + u * 256 >= rows) continue;
        const uint32_t len = e[u] - o[u];
        if (!len) continue;
        uint8_t* d = out + o[u];
        if (dcached) {
          const uint8_t* src = dbc + so[u];
          for (uint32_t m = 0; m < len; m++) d[m] = src[m];
        } else {
          const uint8_t* src = j.dbytes + so[u];
          uint32_t m = 0;
          for (; m + 8 <= len; m += 8) {
            uint64_t v = ld_u64(src + m);
            __builtin_memcpy(d + m, &v, 8);
          }
          for (; m < len; m++) d[m] = src[m];
        }
      }
    }
    return;
  }
  const uint32_t tile_rows = DICT_CHARS_LDS / maxlen;  // >= 1024
  for (uint32_t r0 = 0; r0 < rows; r0 += tile_rows) {
    const uint32_t r1 = r0 + tile_rows < rows ? r0 + tile_rows : rows;
    const uint32_t base = (uint32_t)off[r0], n = (uint32_t)off[r1] - base;
    if (n) {
      for (uint32_t k0 = r0 + tid; k0 < r1; k0 += 256 * 4) {
        // four rows per trip: offsets and keys of all four are requested before any is used
        uint32_t o[4], e[4];
        int32_t key[4];
        unsigned long long word[4];
        uint32_t rk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t k = k0 + u * 256 < r1 ? k0 + u * 256 : k0;
          o[u] = (uint32_t)off[k];
          e[u] = (uint32_t)off[k + 1];
          word[u] = j.vbits ? j.vbits[(row0 + k) >> 6] : ~0ull;
          rk[u] = j.vbits ? j.rank[(row0 + k) >> 6] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          // the row's key: a row with bytes is a non-null row with a key inside the dictionary (dict_rows_kernel)
          const uint32_t k = k0 + u * 256 < r1 ? k0 + u * 256 : k0;
          const uint64_t i = row0 + k;
          const uint64_t di = j.vbits ? (uint64_t)rk[u] + __builtin_popcountll(word[u] & ((1ull << (i & 63)) - 1)) : i;
          key[u] = e[u] != o[u] ? j.dense[di] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (k0 + u * 256 >= r1) continue;
          const uint32_t len = e[u] - o[u];
          if (!len) continue;
          uint8_t* d = chars + (o[u] - base);
          const uint8_t* src = dbc + doffc[key[u]];
          for (uint32_t m = 0; m < len; m++) d[m] = src[m];
        }
      }
      __syncthreads();
      // LDS -> HBM: bytes up to the first 16-byte boundary of the destination one by one, then 16 at a time
      uint8_t* o8 = out + base;
      uint32_t head = (uint32_t)((16 - ((uintptr_t)o8 & 15)) & 15);
      if (head > n) head = n;
      if (tid < head) o8[tid] = chars[tid];
      const uint32_t body = (n - head) / 16;
      for (uint32_t q = tid; q < body; q += 256) {
        uint64_t v[2];
        __builtin_memcpy(v, chars + head + q * 16, 16);
        __builtin_memcpy(o8 + head + (uint64_t)q * 16, v, 16);
      }
      const uint32_t done = head + body * 16;
      if (tid < n - done) o8[done + tid] = chars[done + tid];
      __syncthreads();
    }
  }
}

// ---- dictionary lengths -> dictionary offsets (single workgroup; the dictionary is loaded once per stripe)
__device__ __forceinline__ void dict_offsets_body(const int64_t* dlens, const uint64_t* scalars, uint32_t dict_n_idx,
                                                                       uint32_t data_len_idx, int32_t* dict_off, uint64_t* dict_bytes_out,
                                                                       unsigned long long* err) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  __shared__ int bad_s;
  uint64_t n = scalars[dict_n_idx];
  if (threadIdx.x == 0) {
    carry_s = 0;
    bad_s = 0;
  }
  __syncthreads();
  for (uint64_t s = 0; s < n; s += 256) {
    uint64_t i = s + threadIdx.x;
    int64_t l = i < n ? dlens[i] : 0;
    if (l < 0) {
      atomicOr(&bad_s, 1);
      l = 0;
    }
    uint64_t v = (uint64_t)l;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wbase = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += wsum[w];
    uint64_t e = wbase + incl - v;
    if (i < n) dict_off[i] = (int32_t)(e > 0x7fffffffull ? 0x7fffffff : e);
    __syncthreads();
    if (threadIdx.x == 255) carry_s = wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint64_t total = carry_s;
    dict_off[n] = (int32_t)(total > 0x7fffffffull ? 0x7fffffff : total);
    // construction errors fail the whole stripe decoder (new_string_decoder `?`, string.rs:70-72): index 0
    const bool failed = total > 0x7fffffffull || bad_s || total > scalars[data_len_idx];
    if (total > 0x7fffffffull) report_err64(err, 0, ORC_E_OFFSET_OVERFLOW);
    else if (bad_s) report_err64(err, 0, ORC_E_ARROW);
    else if (total > scalars[data_len_idx]) report_err64(err, 0, ORC_E_ARROW | ORC_E_EOF);  // offsets past the values buffer (try_new)
    // what the UTF-8 checks look at: nothing when the dictionary failed already (they come last in try_new)
    *dict_bytes_out = failed ? 0 : total;
  }
}

// ---- UTF-8 validation (StringArray::try_new): every byte checks its own role ----------------------
// err receives min(byte position << 8 | ORC_E_ARROW).  `n_idx` = scalar with the number of bytes to check.
__device__ __forceinline__ int utf8_lead_len(uint8_t c) {
  if (c < 0x80) return 1;
  if (c >= 0xc2 && c <= 0xdf) return 2;
  if (c >= 0xe0 && c <= 0xef) return 3;
  if (c >= 0xf0 && c <= 0xf4) return 4;
  return 0;  // continuation (0x80..0xbf) or invalid (0xc0, 0xc1, 0xf5..0xff)
}
__device__ __forceinline__ void utf8_validate_body(const uint8_t* s, const uint64_t* scalars, uint32_t n_idx, uint32_t len_idx,
                                                                        unsigned long long* err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t n = scalars[n_idx];
  if (n > scalars[len_idx]) n = scalars[len_idx];  // only bytes the stream really holds (a cut stream leaves stale ones behind)
  if (i >= n) return;
  uint8_t c = s[i];
  if (c < 0x80) return;
  bool bad = false;
  if ((c & 0xc0) == 0x80) {
    // continuation byte: some lead within the previous 3 bytes must cover it
    bool covered = false;
    for (int k = 1; k <= 3 && (uint64_t)k <= i; k++) {
      uint8_t p = s[i - k];
      if ((p & 0xc0) == 0x80) continue;
      covered = utf8_lead_len(p) > k;
      break;
    }
    bad = !covered;
  } else {
    int L = utf8_lead_len(c);
    if (L == 0 || i + L > n) {
      bad = true;
    } else {
      uint8_t c1 = s[i + 1];
      bad = (c1 & 0xc0) != 0x80;
      if (L >= 3) bad |= (s[i + 2] & 0xc0) != 0x80;
      if (L == 4) bad |= (s[i + 3] & 0xc0) != 0x80;
      if (c == 0xe0) bad |= c1 < 0xa0;
      if (c == 0xed) bad |= c1 > 0x9f;
      if (c == 0xf0) bad |= c1 < 0x90;
      if (c == 0xf4) bad |= c1 > 0x8f;
    }
  }
  if (bad) report_err64(err, i, ORC_E_ARROW);
}

// StringArray::try_new per batch: after the UTF-8 check of the batch's bytes, every row offset that lies INSIDE the
// batch's bytes must sit on a character boundary (an empty row at the very end of the batch is not looked at).
// The bytes of all batches sit back to back and utf8_validate_kernel checks them as one text, so one more case is
// settled here: a batch that starts inside a character means the batch before it ends inside one -- that one is
// invalid on its own and fails first.
// charbase / chartot: per-batch byte base and size (nullptr for the dictionary: one "batch" of n_rows offsets).
// Only bytes that exist are looked at (len_idx: the stream's real length); a batch that runs past them is reported
// by string_data_check_kernel / dict_offsets_kernel.
__device__ __forceinline__ void utf8_boundaries_body(const uint8_t* s, const int32_t* offsets, const unsigned long long* charbase,
                                                                          const unsigned long long* chartot, uint64_t n_rows, uint32_t batch,
                                                                          const uint64_t* scalars, uint32_t n_idx, uint32_t len_idx,
                                                                          unsigned long long* err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  const uint64_t b = i / batch;
  const uint64_t base = charbase ? charbase[b] : 0;
  const uint64_t p = base + (uint64_t)(uint32_t)offsets[b * ((uint64_t)batch + 1) + (i - b * batch)];
  const uint64_t have = scalars[len_idx];
  uint64_t n = scalars[n_idx];
  if (n > have) n = have;
  if (p >= n || (s[p] & 0xc0) != 0x80) return;
  if (charbase && i > 0 && i == b * batch) {
    // first row of a batch on a continuation byte that belongs to a character begun before it
    for (int k = 1; k <= 3 && (uint64_t)k <= p; k++) {
      uint8_t q = s[p - k];
      if ((q & 0xc0) == 0x80) continue;
      if (utf8_lead_len(q) > k) report_err64(err, i - 1, ORC_E_ARROW);
      break;
    }
  }
  const uint64_t end = chartot ? base + chartot[b] : n;
  if (end > have) return;
  if (p < end) report_err64(err, i, ORC_E_ARROW);
}

// Direct strings: the bytes consumed by all batches must exist in DATA (try_new: offsets past the buffer)
__device__ __forceinline__ void string_data_check_body(const unsigned long long* chartot, const unsigned long long* charbase, uint32_t n_batches,
                                                    const uint64_t* scalars, uint32_t data_len_idx, uint32_t batch, unsigned long long* err) {
  uint32_t b = blockIdx.x * 64 + threadIdx.x;
  if (b >= n_batches) return;
  if (charbase[b] + chartot[b] > scalars[data_len_idx]) report_err64(err, (uint64_t)b * batch, ORC_E_ARROW | ORC_E_EOF);
}

// ---- Decimal: zigzag varints -> i128 ----------------------------------------------------------------
// pass 1: per 64-byte word a bitmask of terminator bytes (top bit clear) + popcount
__device__ __forceinline__ void varint_terms_body(const uint8_t* s, const uint64_t* scalars, uint32_t len_idx, uint64_t n_words,
                                                                       unsigned long long* tmask, uint32_t* tpop) {
  uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  uint64_t len = scalars[len_idx];
  unsigned long long m = 0;
  uint64_t base = w * 64;
  for (int k = 0; k < 8; k++) {
    uint64_t p = base + 8 * k;
    if (p >= len) break;
    uint64_t v = ~ld_u64(s + p) & 0x8080808080808080ull;
    // gather bit 7 of each byte into 8 consecutive bits
    uint64_t bits = (v * 0x0002040810204081ull) >> 56;
    uint64_t rem = len - p;
    if (rem < 8) bits &= (1ull << rem) - 1;
    m |= bits << (8 * k);
  }
  tmask[w] = m;
  tpop[w] = (uint32_t)__builtin_popcountll(m);
}

// pass 2: one thread per stream byte; terminators decode their varint into dense[k]
__device__ __forceinline__ void varint_decode128_body(const uint8_t* s, const uint64_t* scalars, uint32_t len_idx, uint32_t needed_idx,
                                                                           const unsigned long long* tmask, const uint32_t* trank, __int128* dense,
                                                                           uint64_t n_upper, unsigned long long* err) {
  uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t len = scalars[len_idx];
  if (len > n_upper) len = n_upper;
  if (p >= len) return;
  unsigned long long m = tmask[p >> 6];
  uint32_t bit = p & 63;
  if (!((m >> bit) & 1)) {
    // an unterminated tail: the stream ends inside a varint
    if (p == len - 1) {
      uint64_t k = (uint64_t)trank[p >> 6] + __builtin_popcountll(m & ((1ull << bit) - 1));
      uint64_t needed = scalars[needed_idx];
      if (k < needed) {
        uint64_t run = 1;
        while (run <= 20 && run <= p && (s[p - run] & 0x80)) run++;
        report_err64(err, k, run >= 20 ? ORC_E_VARINT : ORC_E_IO);
      }
    }
    return;
  }
  uint64_t k = (uint64_t)trank[p >> 6] + __builtin_popcountll(m & ((1ull << bit) - 1));
  uint64_t needed = scalars[needed_idx];
  if (k >= needed) return;
  uint64_t start = p;
  while (start > 0 && p - start < 20 && (s[start - 1] & 0x80)) start--;
  uint32_t nb = (uint32_t)(p - start + 1);
  if (nb > 19) {  // byte index 19 has offset 133 >= 128: checked_shl fails (VarintTooLarge)
    report_err64(err, k, ORC_E_VARINT);
    return;
  }
  unsigned __int128 u = 0;
  for (uint32_t i = 0; i < nb; i++) u |= (unsigned __int128)(s[start + i] & 0x7f) << (7 * i);
  unsigned __int128 z = (u >> 1) ^ (unsigned __int128)(-(__int128)(u & 1));
  dense[k] = (__int128)z;
  // "not enough values": the last terminator knows how many values exist
  if (p == len - 1 && k + 1 < needed) report_err64(err, k + 1, ORC_E_IO);
}
// empty DATA stream with values needed
__device__ __forceinline__ void varint_empty_check_body(const uint64_t* scalars, uint32_t len_idx, uint32_t needed_idx, unsigned long long* err) {
  if (threadIdx.x == 0 && scalars[len_idx] == 0 && scalars[needed_idx] > 0) report_err64(err, 0, ORC_E_IO);
}

// Decimal finish: null spacing + per-value scale repair (array_decoder/decimal.rs:138-166; release-build wrapping)
__device__ __forceinline__ void decimal_finish_body(const __int128* dense, const int32_t* scales, const unsigned long long* vbits,
                                                                         const uint32_t* rank, __int128* out, uint64_t n_rows, uint32_t fixed_scale) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  uint64_t d = i;
  bool valid = true;
  if (vbits) {
    unsigned long long word = vbits[i >> 6];
    uint32_t bit = i & 63;
    valid = (word >> bit) & 1;
    d = (uint64_t)rank[i >> 6] + __builtin_popcountll(word & ((1ull << bit) - 1));
  }
  __int128 v = 0;
  if (valid) {
    v = dense[d];
    uint32_t vs = (uint32_t)scales[d];
    if (vs != fixed_scale) {
      uint32_t k = fixed_scale < vs ? vs - fixed_scale : fixed_scale - vs;
      unsigned __int128 f = 1;
      for (uint32_t t = 0; t < k && t < 200; t++) f *= 10;
      if (fixed_scale < vs) {
        __int128 sf = (__int128)f;
        if (sf != 0) v = v / sf;
      } else {
        v = (__int128)((unsigned __int128)v * f);
      }
    }
  }
  out[i] = v;
}
