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
  const void* dense;                // decoded keys of the non-null rows, key_bytes wide (all ones: not a possible key, see store_val)
  const unsigned long long* vbits;  // stripe-wide validity words (null: no PRESENT stream)
  const uint32_t* rank;             // non-null rows before each 64-row word
  const int32_t* doff;              // dictionary offsets (dict_n + 1)
  const uint8_t* dbytes;            // dictionary bytes
  int32_t* offsets;                 // out: batch b at b * (batch + 1)
  unsigned long long* chartot;      // per-batch byte totals, then (at + n_batches) their exclusive scan
  unsigned long long* err;
  const unsigned long long* dict_err;  // error word of the dictionary itself (bad lengths / short blob): nothing of it may be touched then
  uint8_t* out_chars;               // value bytes of the column (set for the second launch)
  uint64_t* total_out;              // scalar receiving the column's total value bytes
  uint64_t n_rows;
  uint32_t batch, n_batches;
  uint32_t dict_n_idx;              // scalar holding the dictionary size
  uint32_t key_bytes;               // 1 (dictionaries of up to 255 entries), 2 (up to 65 535) or 4
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
// key of dense value di; -1 for what the expansion marked as impossible (negative or wider than the key type)
__device__ __forceinline__ int32_t dict_key(const DictJob& j, uint64_t di) {
  if (j.key_bytes == 1) {
    const uint32_t k = static_cast<const uint8_t*>(j.dense)[di];
    return k == 0xffu ? -1 : (int32_t)k;
  }
  if (j.key_bytes == 2) {
    const uint32_t k = static_cast<const uint16_t*>(j.dense)[di];
    return k == 0xffffu ? -1 : (int32_t)k;
  }
  return static_cast<const int32_t*>(j.dense)[di];
}
#define DICT_PER 8u                 // consecutive rows per thread
#define DICT_TILE (256u * DICT_PER) // rows per tile (a batch takes several)
#define DICT_DOFF_LDS 2048u         // dictionaries of up to this many entries ...
#define DICT_BYTES_LDS 8192u        // ... and this many bytes are worked on from LDS
#define DICT_CHARS_LDS 16384u       // value bytes assembled in LDS per round
#define DICT_STAGE_MAXLEN 64u       // longer entries go row by row straight to memory

// The dictionary's offsets (and, when asked, bytes) in LDS if they fit; returns whether they do.  dict_n = 0 for a dictionary that
// failed its own checks (negative / overflowing lengths, blob shorter than their sum): the column fails at construction, its
// offsets are not to be trusted -- every row gets an empty string, nothing is gathered.
__device__ __forceinline__ bool dict_cache(const DictJob& j, uint64_t dict_n, uint16_t* doffc, uint8_t* dbc, uint32_t* maxlen_s, uint32_t tid) {
  if (dict_n > DICT_DOFF_LDS) return false;
  const uint32_t dict_bytes = (uint32_t)j.doff[dict_n];
  if (dict_bytes > DICT_BYTES_LDS) return false;
  uint32_t ml = 0;
  for (uint32_t i = tid; i <= dict_n; i += 256) {
    const int32_t o = j.doff[i];
    doffc[i] = (uint16_t)o;
    if (i < dict_n) {
      const uint32_t l = (uint32_t)(j.doff[i + 1] - o);
      ml = l > ml ? l : ml;
    }
  }
  if (dbc)
    for (uint32_t i = tid * 8; i < dict_bytes; i += 256 * 8) {
      const uint64_t v = ld_u64(j.dbytes + i);  // (the dictionary stream has ORC_PAD bytes of slack behind it)
      __builtin_memcpy(dbc + i, &v, 8);
    }
  if (maxlen_s) atomicMax(maxlen_s, ml);
  return true;
}

// Rows i .. i + cnt - 1 (cnt <= 8) of the column: their validity bits (bit r = row i + r) and the dense index of the first valid one.
__device__ __forceinline__ uint32_t dict_valid8(const DictJob& j, uint64_t i, uint32_t cnt, uint64_t* d0) {
  uint32_t bits = 0xffu;
  *d0 = i;
  if (j.vbits) {
    const uint32_t sh = i & 63;
    const unsigned long long w0 = j.vbits[i >> 6];
    const unsigned long long w1 = j.vbits[(i + 7) >> 6];  // (at most the word behind the last one: the bitmap has slack, `cnt` masks it)
    bits = (uint32_t)((w0 >> sh) | (sh ? w1 << (64 - sh) : 0ull)) & 0xffu;
    *d0 = (uint64_t)j.rank[i >> 6] + __builtin_popcountll(w0 & ((1ull << sh) - 1));
  }
  return bits & (cnt >= 8 ? 0xffu : (1u << cnt) - 1u);
}

// Lengths (and dictionary positions) of 8 consecutive rows: keys of the valid ones are consecutive dense values -- for one-byte
// keys a single 8-byte load.  A key that is no key (out of bounds, or marked impossible by the expansion) yields length 0 and
// sets bit r of *bad.
template <bool CACHED, bool KEYS = false>
__device__ __forceinline__ void dict_rows8(const DictJob& j, uint32_t bits, uint64_t d0, uint64_t dict_n, const uint16_t* doffc, uint32_t* len, uint32_t* so,
                                           uint32_t* bad) {
  uint64_t packed = 0;
  if (j.key_bytes == 1 && bits) packed = ld_u64(static_cast<const uint8_t*>(j.dense) + d0);  // (the key buffer has slack behind it)
  uint32_t idx = 0;
  *bad = 0;
#pragma unroll
  for (uint32_t r = 0; r < 8; r++) {
    len[r] = 0;
    so[r] = 0;
    if (!((bits >> r) & 1)) continue;
    int32_t key;
    if (j.key_bytes == 1) {
      const uint32_t k = (uint32_t)(packed >> (8 * idx)) & 0xffu;
      key = k == 0xffu ? -1 : (int32_t)k;
    } else {
      key = dict_key(j, d0 + idx);
    }
    idx++;
    if (key < 0 || (uint64_t)key >= dict_n) {
      *bad |= 1u << r;
      continue;
    }
    const uint32_t o = CACHED ? (uint32_t)doffc[key] : (uint32_t)j.doff[key];
    so[r] = KEYS ? (uint32_t)key : o;  // (KEYS: the entry's index: dict_emit_small8 reads padded entries)
    len[r] = (CACHED ? (uint32_t)doffc[key + 1] : (uint32_t)j.doff[key + 1]) - o;
  }
}

// Pass 1, one workgroup per (batch, column): keys bounds-checked (DictionaryArray::try_new) -> the batch's value byte total.
// Nothing per row is written: the second pass (behind the host's sizing of the character arena) forms the offsets again
// from the keys -- one byte each for small dictionaries -- instead of reading four bytes per row back.
extern "C" __global__ void __launch_bounds__(256) dict_totals_kernel(const DictJob* jobs, const uint64_t* scalars) {
  __shared__ uint16_t doffc[DICT_DOFF_LDS + 2];
  __shared__ uint64_t wsum[4];
  const DictJob j = dict_job(jobs, blockIdx.y);
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  if (b >= j.n_batches) return;
  const bool dict_ok = *j.dict_err == RLE_NO_ERR;
  const uint64_t dict_n = dict_ok ? scalars[j.dict_n_idx] : 0;
  const bool cached = dict_cache(j, dict_n, doffc, nullptr, nullptr, tid);
  __syncthreads();
  const uint64_t row0 = (uint64_t)b * j.batch;
  const uint64_t rows = j.n_rows - row0 < j.batch ? j.n_rows - row0 : j.batch;
  uint64_t sum = 0;
  for (uint64_t k = (uint64_t)tid * DICT_PER; k < rows; k += DICT_TILE) {
    const uint32_t cnt = rows - k < DICT_PER ? (uint32_t)(rows - k) : DICT_PER;
    uint64_t d0;
    const uint32_t bits = dict_valid8(j, row0 + k, cnt, &d0);
    uint32_t len[8], so[8], bad;
    if (cached) dict_rows8<true>(j, bits, d0, dict_n, doffc, len, so, &bad);
    else dict_rows8<false>(j, bits, d0, dict_n, doffc, len, so, &bad);
    if (bad && dict_ok) report_err64(j.err, row0 + k + (uint32_t)__builtin_ctz(bad), ORC_E_ARROW);  // (a failed dictionary is what the column reports)
#pragma unroll
    for (uint32_t r = 0; r < 8; r++) sum += len[r];
  }
  for (int o = 32; o; o >>= 1) sum += __shfl_down(sum, o);
  if ((tid & 63) == 0) wsum[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) {
    const uint64_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    j.chartot[b] = total;
    if (total > 0x7fffffffull) report_err64(j.err, row0, ORC_E_ARROW);
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

// Pass 2, one workgroup per (batch, column): keys -> lengths -> int32 offsets restarting at 0 (string.rs:139-140) AND the
// batch's value bytes, in one go.  A tile of 2048 rows at a time, thread t working on rows 8t .. 8t+7: their keys are eight
// consecutive dense values (one load), lengths and their running sums stay in registers, one block scan of the 256 sums places
// the thread, which writes its 8 offsets as two 16-byte stores.  The tile's value bytes are assembled in LDS from the LDS copy
// of the dictionary -- a thread's rows are one contiguous run of the output --, DICT_CHARS_LDS bytes per round (a row that
// straddles two rounds is copied in parts), and leave as 16-byte coalesced stores.  Dictionaries too big for LDS, or with
// entries longer than DICT_STAGE_MAXLEN, go row by row straight from memory to memory.
// Dictionaries of SHORT entries (eight bytes at most, 1024 entries at most: flags, ship modes, the 7-entry dictionary of BASELINE's
// C3): the LDS copy of the dictionary is kept PADDED, one aligned 8-byte word per entry, zeros behind its bytes.  A thread's eight
// rows are one contiguous run of the tile's value bytes: it streams them through a 64-bit accumulator -- entry word shifted in,
// a full word OR-ed into the (zeroed) LDS stage whenever eight bytes are together -- so a row costs one aligned LDS read, a dozen
// vector instructions and at most one LDS atomic, where the byte-by-byte copy cost ten LDS operations and their loop for a row of
// five bytes (C3: 0.42 -> 0.3x ms per 100 M rows; unaligned 8-byte LDS copies were slower than the bytes).  A word may be shared by
// two threads (a run starts and ends anywhere): every word goes in by ds_or.
__device__ __forceinline__ void dict_emit_small8(const DictJob& j, uint64_t dict_n, uint8_t* chars, const uint16_t* doffc, const uint64_t* dpad, uint64_t* wsum, uint32_t b,
                                                 uint32_t tid) {
  const uint64_t row0 = (uint64_t)b * j.batch;
  const uint64_t rows = j.n_rows - row0 < j.batch ? j.n_rows - row0 : j.batch;
  int32_t* out = j.offsets + (uint64_t)b * ((uint64_t)j.batch + 1);
  uint8_t* cout = j.out_chars ? j.out_chars + j.chartot[j.n_batches + b] : nullptr;
  unsigned long long* const stage = reinterpret_cast<unsigned long long*>(chars);
  uint64_t carry = 0;
  for (uint64_t t0 = 0; t0 < rows; t0 += DICT_TILE) {
    const uint64_t k = t0 + (uint64_t)tid * DICT_PER;
    const uint32_t cnt = k < rows ? (rows - k < DICT_PER ? (uint32_t)(rows - k) : DICT_PER) : 0u;
    uint32_t len[8], key[8], bad;
    uint64_t d0 = 0;
    const uint32_t bits = cnt ? dict_valid8(j, row0 + k, cnt, &d0) : 0u;
    dict_rows8<true, true>(j, bits, d0, dict_n, doffc, len, key, &bad);
    uint32_t ex[8];
    uint32_t run = 0;
#pragma unroll
    for (uint32_t r = 0; r < 8; r++) {
      ex[r] = run;
      run += len[r];
    }
    uint32_t incl = run;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if ((int)(tid & 63) >= o) incl += t;
    }
    __syncthreads();  // (wsum and the stage of the previous tile are no longer read)
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    // the stage, zeroed: 2048 rows of eight bytes at most
    for (uint32_t q = tid; q < DICT_CHARS_LDS / 16; q += 256) reinterpret_cast<uint4*>(chars)[q] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    uint32_t base = incl - run;
    for (uint32_t w = 0; w < (tid >> 6); w++) base += (uint32_t)wsum[w];
    const uint32_t tile_total = (uint32_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    if (cnt == DICT_PER) {
      int32_t o8[8];
#pragma unroll
      for (uint32_t r = 0; r < 8; r++) o8[r] = (int32_t)(carry + base + ex[r]);
      __builtin_memcpy(out + k, o8, 32);
    } else {
      for (uint32_t r = 0; r < cnt; r++) out[k + r] = (int32_t)(carry + base + ex[r]);
    }
    if (cout && tile_total) {
      // this thread's bytes: stage bytes [base, base + run).  acc holds the bytes of stage word `word` gathered so far (zeros elsewhere)
      uint32_t word = base >> 3, fill = base & 7u;
      uint64_t acc = 0;
#pragma unroll
      for (uint32_t r = 0; r < 8; r++) {
        const uint64_t v = len[r] ? dpad[key[r]] : 0ull;  // (the entry's bytes, zeros behind them)
        acc |= v << (8u * fill);
        const uint32_t nf = fill + len[r];
        if (nf >= 8u) {
          atomicOr(&stage[word], (unsigned long long)acc);
          word++;
          acc = fill ? v >> (8u * (8u - fill)) : 0ull;  // what did not fit (fill = 0: it all did)
        }
        fill = nf & 7u;
      }
      if (fill) atomicOr(&stage[word], (unsigned long long)acc);
    }
    __syncthreads();
    if (cout && tile_total) {
      // LDS -> HBM: bytes up to the first 16-byte boundary of the destination one by one, then 16 at a time
      uint8_t* p8 = cout + carry;
      const uint32_t n = tile_total;
      uint32_t head = (uint32_t)((16 - ((uintptr_t)p8 & 15)) & 15);
      if (head > n) head = n;
      if (tid < head) p8[tid] = chars[tid];
      const uint32_t body = (n - head) / 16;
      for (uint32_t q = tid; q < body; q += 256) {
        uint64_t v[2];
        __builtin_memcpy(v, chars + head + q * 16, 16);
        __builtin_memcpy(p8 + head + (uint64_t)q * 16, v, 16);
      }
      const uint32_t done = head + body * 16;
      if (tid < n - done) p8[done + tid] = chars[done + tid];
    }
    carry += tile_total;
  }
  if (tid == 0) out[rows] = (int32_t)carry;
}

template <bool CACHED>
__device__ __forceinline__ void dict_emit_body(const DictJob& j, uint64_t dict_n, uint8_t* chars, const uint16_t* doffc, const uint8_t* dbc, uint64_t* wsum,
                                               bool staged, uint32_t b, uint32_t tid) {
  const uint64_t row0 = (uint64_t)b * j.batch;
  const uint64_t rows = j.n_rows - row0 < j.batch ? j.n_rows - row0 : j.batch;
  int32_t* out = j.offsets + (uint64_t)b * ((uint64_t)j.batch + 1);
  uint8_t* cout = j.out_chars ? j.out_chars + j.chartot[j.n_batches + b] : nullptr;  // (null: the column has no value bytes at all)
  uint64_t carry = 0;
  for (uint64_t t0 = 0; t0 < rows; t0 += DICT_TILE) {
    const uint64_t k = t0 + (uint64_t)tid * DICT_PER;   // first row of this thread in the batch
    const uint32_t cnt = k < rows ? (rows - k < DICT_PER ? (uint32_t)(rows - k) : DICT_PER) : 0u;
    uint32_t len[8], so[8], bad;
    uint64_t d0 = 0;
    const uint32_t bits = cnt ? dict_valid8(j, row0 + k, cnt, &d0) : 0u;
    dict_rows8<CACHED>(j, bits, d0, dict_n, doffc, len, so, &bad);  // (pass 1 reported the keys out of bounds: length 0 here)
    // exclusive running sums of the thread's lengths, block scan of the 256 thread totals
    uint32_t ex[8];
    uint64_t run = 0;
#pragma unroll
    for (uint32_t r = 0; r < 8; r++) {
      ex[r] = (uint32_t)run;  // < 2^32: a batch total above i32::MAX was reported by pass 1
      run += len[r];
    }
    uint64_t incl = run;
    for (int o = 1; o < 64; o <<= 1) {
      uint64_t t = __shfl_up(incl, o);
      if ((int)(tid & 63) >= o) incl += t;
    }
    __syncthreads();  // (wsum and chars of the previous tile are no longer read)
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint64_t base = incl - run;
    for (uint32_t w = 0; w < (tid >> 6); w++) base += wsum[w];
    const uint64_t tile_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // offsets: 8 consecutive int32 per thread
    if (cnt == DICT_PER) {
      int32_t o8[8];
#pragma unroll
      for (uint32_t r = 0; r < 8; r++) o8[r] = (int32_t)(carry + base + ex[r]);
      __builtin_memcpy(out + k, o8, 32);
    } else {
      for (uint32_t r = 0; r < cnt; r++) out[k + r] = (int32_t)(carry + base + ex[r]);
    }
    // value bytes
    if (cout && tile_total) {
      uint8_t* o8 = cout + carry;
      if (staged) {
        for (uint64_t rb = 0; rb < tile_total; rb += DICT_CHARS_LDS) {
          const uint32_t n = tile_total - rb < DICT_CHARS_LDS ? (uint32_t)(tile_total - rb) : DICT_CHARS_LDS;
          if (rb) __syncthreads();  // (the previous round has left the stage)
          if (base + run > rb && base < rb + n) {
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) {
              const uint64_t o = base + ex[r], e = o + len[r];
              const uint64_t lo = o > rb ? o : rb, hi = e < rb + n ? e : rb + n;
              if (lo < hi) {
                const uint8_t* src = dbc + so[r] + (uint32_t)(lo - o);
                uint8_t* d = chars + (uint32_t)(lo - rb);
                // (a byte at a time: eight bytes a step -- one unaligned 8-byte LDS read, one to four stores -- was SLOWER, 0.48 against
                // 0.42 ms per 100 M rows: unaligned LDS accesses take several passes)
                for (uint32_t q = 0; q < (uint32_t)(hi - lo); q++) d[q] = src[q];
              }
            }
          }
          __syncthreads();
          // LDS -> HBM: bytes up to the first 16-byte boundary of the destination one by one, then 16 at a time
          uint8_t* p8 = o8 + rb;
          uint32_t head = (uint32_t)((16 - ((uintptr_t)p8 & 15)) & 15);
          if (head > n) head = n;
          if (tid < head) p8[tid] = chars[tid];
          const uint32_t body = (n - head) / 16;
          for (uint32_t q = tid; q < body; q += 256) {
            uint64_t v[2];
            __builtin_memcpy(v, chars + head + q * 16, 16);
            __builtin_memcpy(p8 + head + (uint64_t)q * 16, v, 16);
          }
          const uint32_t done = head + body * 16;
          if (tid < n - done) p8[done + tid] = chars[done + tid];
        }
      } else {
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
          if (!len[r]) continue;
          const uint8_t* src = j.dbytes + so[r];
          uint8_t* d = o8 + base + ex[r];
          uint32_t m = 0;
          for (; m + 8 <= len[r]; m += 8) {
            const uint64_t v = ld_u64(src + m);
            __builtin_memcpy(d + m, &v, 8);
          }
          for (; m < len[r]; m++) d[m] = src[m];
        }
      }
    }
    carry += tile_total;
  }
  if (tid == 0) out[rows] = (int32_t)carry;
}

// Between the passes: the value bytes of a result's dictionary columns are placed one behind the other in its character arena,
// here on the device -- the host, which used to wait for the totals of pass 1 to do it, places them the same way when the call
// is over, and sends pass 2 again in the rare case the arena (sized by the result's earlier use) proved too small: then the
// columns from the first misfit on get no value bytes from this launch (out_chars null), only their offsets.
struct DictPlace {
  uint8_t* base;   // the result's character arena
  uint64_t cap;
  uint32_t first, count;  // its jobs in the table
};
extern "C" __global__ void __launch_bounds__(64) dict_place_kernel(DictJob* jobs, const DictPlace* places, uint32_t n_places, uint64_t pad, uint64_t align) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_places) return;
  const DictPlace pl = places[i];
  uint64_t need = 0;
  bool over = false;
  for (uint32_t k = pl.first; k < pl.first + pl.count; k++) {
    const uint64_t total = *glob(jobs[k].total_out);
    const uint64_t off = (need + align - 1) & ~(align - 1);
    need = off + total + pad;
    over = over || need + align > pl.cap;
    jobs[k].out_chars = (!over && total) ? pl.base + off : nullptr;
  }
}

extern "C" __global__ void __launch_bounds__(256) dict_emit_kernel(const DictJob* jobs, const uint64_t* scalars) {
  __shared__ __attribute__((aligned(16))) uint8_t chars[DICT_CHARS_LDS + 16];
  __shared__ uint16_t doffc[DICT_DOFF_LDS + 2];
  __shared__ __attribute__((aligned(8))) uint8_t dbc[DICT_BYTES_LDS + 8];
  __shared__ uint64_t wsum[4];
  __shared__ uint32_t maxlen_s;
  const DictJob j = dict_job(jobs, blockIdx.y);
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  if (b >= j.n_batches) return;
  const bool dict_ok = *j.dict_err == RLE_NO_ERR;
  const uint64_t dict_n = dict_ok ? scalars[j.dict_n_idx] : 0;
  if (tid == 0) maxlen_s = 0;
  __syncthreads();
  const bool cached = dict_cache(j, dict_n, doffc, dbc, &maxlen_s, tid);
  __syncthreads();
  if (cached && maxlen_s <= 8 && dict_n <= DICT_BYTES_LDS / 8) {
    // short entries: the dictionary copy re-laid as one padded word per entry (read out first: it is rewritten in place)
    uint64_t pad[DICT_BYTES_LDS / 8 / 256];
#pragma unroll
    for (uint32_t q = 0; q < DICT_BYTES_LDS / 8 / 256; q++) {
      const uint32_t e = tid + 256 * q;
      pad[q] = 0;
      if (e < dict_n) {
        const uint32_t o = doffc[e], l = (uint32_t)doffc[e + 1] - o;
        uint64_t v;
        __builtin_memcpy(&v, dbc + o, 8);  // (eight bytes of slack behind the copy)
        pad[q] = l >= 8 ? v : v & ((1ull << (8 * l)) - 1);
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < DICT_BYTES_LDS / 8 / 256; q++) {
      const uint32_t e = tid + 256 * q;
      if (e < dict_n) reinterpret_cast<uint64_t*>(dbc)[e] = pad[q];
    }
    __syncthreads();
    dict_emit_small8(j, dict_n, chars, doffc, reinterpret_cast<const uint64_t*>(dbc), wsum, b, tid);
    return;
  }
  if (cached) dict_emit_body<true>(j, dict_n, chars, doffc, dbc, wsum, maxlen_s <= DICT_STAGE_MAXLEN, b, tid);
  else dict_emit_body<false>(j, dict_n, chars, doffc, dbc, wsum, false, b, tid);
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
// (16 bytes per thread: a vector without a byte >= 0x80 -- ASCII text, the common case -- is done with its one load)
__device__ __forceinline__ void utf8_validate_byte(const uint8_t* s, uint64_t i, uint64_t n, unsigned long long* err);
// `nonascii` (may be null): set when the text holds a byte of 0x80 or more -- text without one has no row offset inside a character,
// utf8_boundaries_body then has nothing to look for (a read of one byte per row at 75 M places of lineitem's l_comment otherwise)
__device__ __forceinline__ void utf8_validate_body(const uint8_t* s, const uint64_t* scalars, uint32_t n_idx, uint32_t len_idx,
                                                                        unsigned long long* err, uint64_t* nonascii) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  uint64_t n = scalars[n_idx];
  if (n > scalars[len_idx]) n = scalars[len_idx];  // only bytes the stream really holds (a cut stream leaves stale ones behind)
  if (i0 >= n) return;
  if (i0 + 16 <= n) {
    uint64_t v[2];
    __builtin_memcpy(v, s + i0, 16);
    if (((v[0] | v[1]) & 0x8080808080808080ull) == 0) return;
  }
  const uint64_t i1 = i0 + 16 < n ? i0 + 16 : n;
  for (uint64_t i = i0; i < i1; i++)
    if (s[i] >= 0x80) {
      if (nonascii) *nonascii = 1;
      utf8_validate_byte(s, i, n, err);
    }
}
// A direct string column's DATA stream is its Arrow value buffer: copied there AND checked in one pass (16 bytes per thread, the
// vector the copy has in its registers: ASCII text is done with it; round 4 read the 1.9 GB of lineitem's l_comment a second time)
__device__ __forceinline__ void copy_validate_body(const uint8_t* src, uint8_t* dst, uint64_t n_copy, const uint64_t* scalars, uint32_t n_idx, uint32_t len_idx,
                                                   unsigned long long* err, uint64_t* nonascii) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i0 >= n_copy) return;
  uint64_t n = scalars[n_idx];
  if (n > scalars[len_idx]) n = scalars[len_idx];  // only bytes the stream really holds are text
  if (i0 + 16 <= n_copy) {
    uint64_t v[2];
    __builtin_memcpy(v, src + i0, 16);
    __builtin_memcpy(dst + i0, v, 16);
    if (i0 + 16 <= n && ((v[0] | v[1]) & 0x8080808080808080ull) == 0) return;
  } else {
    for (uint64_t k = i0; k < n_copy; k++) dst[k] = src[k];
  }
  const uint64_t i1 = i0 + 16 < n ? i0 + 16 : n;
  for (uint64_t i = i0; i < i1; i++)
    if (src[i] >= 0x80) {
      if (nonascii) *nonascii = 1;
      utf8_validate_byte(src, i, n, err);
    }
}
__device__ __forceinline__ void utf8_validate_byte(const uint8_t* s, uint64_t i, uint64_t n, unsigned long long* err) {
  const uint8_t c = s[i];
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
                                                                          unsigned long long* err, const uint64_t* nonascii) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  if (nonascii && !*nonascii) return;  // (pure ASCII: no offset can lie inside a character -- the validation pass has seen every byte)
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

// A value of scale `vs` at the column's scale (array_decoder/decimal.rs:138-166; release-build wrapping)
__device__ __forceinline__ __int128 decimal_rescale(__int128 v, uint32_t vs, uint32_t fixed_scale) {
  if (vs == fixed_scale) return v;
  const uint32_t k = fixed_scale < vs ? vs - fixed_scale : fixed_scale - vs;
  unsigned __int128 f = 1;
  for (uint32_t t = 0; t < k && t < 200; t++) f *= 10;
  if (fixed_scale < vs) {
    const __int128 sf = (__int128)f;
    return sf != 0 ? v / sf : v;
  }
  return (__int128)((unsigned __int128)v * f);
}

// pass 2: one thread per EIGHT stream bytes (one thread per byte made 64 stream bytes a wavefront's whole work: the kernel then
// lasts as long as its waves' chains of dependent loads, not as its bytes -- 7.4 ms for 600 MB of lineitem's decimals); the
// thread's terminators, in order, decode their varints into dense[k], k counted up from the rank of the thread's first byte.
// `scales` (a column without nulls: value k IS row k): the value is brought to the column's scale here and `dense` is the
// column's Arrow buffer -- decimal_finish_body, a pass over 36 bytes per value that only re-reads what this one has just
// written, is not run.
__device__ __forceinline__ void varint_decode128_body(const uint8_t* s, const uint64_t* scalars, uint32_t len_idx, uint32_t needed_idx,
                                                                           const unsigned long long* tmask, const uint32_t* trank, __int128* dense,
                                                                           uint64_t n_upper, unsigned long long* err, const int32_t* scales_, uint32_t fixed_scale,
                                                                           const uint64_t* uniform) {
  // (`uniform`: the flag of rle2_uniform_kernel -- every value's scale is the column's: the scales were not expanded)
  const int32_t* scales = uniform && *uniform ? nullptr : scales_;
  // The values of a workgroup's 2 KiB of stream are one contiguous run of the output: they are gathered in LDS and leave as
  // consecutive 16-byte stores.  (Stored from the thread that decodes them -- its two to four values at a lane stride of 32 to 64
  // bytes, one value per instruction -- every instruction wrote a quarter to a half of each 64-byte line it touched: 2.5 x the
  // output in HBM writes by the counters.)
  __shared__ __attribute__((aligned(16))) __int128 stage[2048];  // (2048 stream bytes end 2048 values at the very most)
  __shared__ uint32_t stage_n;
  if (threadIdx.x == 0) stage_n = 0;
  __syncthreads();
  const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  uint64_t len = scalars[len_idx];
  if (len > n_upper) len = n_upper;
  const uint64_t needed = scalars[needed_idx];
  const uint64_t wg0 = (uint64_t)blockIdx.x * 2048;  // (a multiple of 64: the rank word itself)
  const uint64_t k0 = wg0 < len ? (uint64_t)trank[wg0 >> 6] : 0;
  if (p0 < len) {
  const unsigned long long m = tmask[p0 >> 6];
  const uint32_t sh = (uint32_t)p0 & 63;
  uint32_t mb = (uint32_t)(m >> sh) & 0xffu;  // (bits at and behind `len` are clear: varint_terms_body)
  uint64_t k = (uint64_t)trank[p0 >> 6] + __builtin_popcountll(m & ((1ull << sh) - 1));
  if (len - p0 <= 8 && !((mb >> (len - 1 - p0)) & 1)) {
    // an unterminated tail: the stream ends inside a varint
    const uint64_t p = len - 1, kt = k + __builtin_popcount(mb);
    if (kt < needed) {
      uint64_t run = 1;
      while (run <= 20 && run <= p && (s[p - run] & 0x80)) run++;
      report_err64(err, kt, run >= 20 ? ORC_E_VARINT : ORC_E_IO);
    }
  }
  if (mb && k < needed) {
  const uint64_t own = ld_u64(s + p0), prev = p0 ? ld_u64(s + p0 - 8) : 0;
  uint64_t kend = k;
  for (; mb && k < needed; mb &= mb - 1, k++) {
    kend = k + 1;
    const uint32_t j = (uint32_t)__builtin_ctz(mb);
    const uint64_t p = p0 + j;
    if (p >= 7) {
      // the common case, a varint of at most 7 bytes, out of the 8 bytes that end at p: the bytes before p that carry a
      // continuation flag, counted from p - 1 down, are the varint's (six or fewer: else the general path below)
      const uint64_t x = j == 7 ? own : (own << (8 * (7 - j))) | (prev >> (8 * (j + 1)));
      const uint64_t open = ~(x << 8) & 0x8080808080808000ull;  // flag CLEAR in bytes p - 1 (top) ... p - 7
      const uint32_t cont = open ? (uint32_t)__builtin_clzll(open) >> 3 : 7u;
      if (cont < 7) {
        const uint64_t v = x >> (8 * (7 - cont));  // the varint's first byte in byte 0
        uint64_t u = v & 0x7f;
        u |= (v >> 1) & (0x7full << 7);
        u |= (v >> 2) & (0x7full << 14);
        u |= (v >> 3) & (0x7full << 21);
        u |= (v >> 4) & (0x7full << 28);
        u |= (v >> 5) & (0x7full << 35);
        u |= (v >> 6) & (0x7full << 42);
        // (bytes behind the terminator do not exist in v: the shift brought zeros in; the terminator's own flag is clear)
        const int64_t z = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
        stage[k - k0] = scales ? decimal_rescale((__int128)z, (uint32_t)scales[k], fixed_scale) : (__int128)z;
        if (p == len - 1 && k + 1 < needed) report_err64(err, k + 1, ORC_E_IO);
        continue;
      }
    }
    uint64_t start = p;
    while (start > 0 && p - start < 20 && (s[start - 1] & 0x80)) start--;
    uint32_t nb = (uint32_t)(p - start + 1);
    if (nb > 19) {  // byte index 19 has offset 133 >= 128: checked_shl fails (VarintTooLarge)
      report_err64(err, k, ORC_E_VARINT);
      stage[k - k0] = 0;  // (never read by a consumer: the column fails at this value)
      continue;
    }
    unsigned __int128 u = 0;
    for (uint32_t i = 0; i < nb; i++) u |= (unsigned __int128)(s[start + i] & 0x7f) << (7 * i);
    unsigned __int128 z = (u >> 1) ^ (unsigned __int128)(-(__int128)(u & 1));
    stage[k - k0] = scales ? decimal_rescale((__int128)z, (uint32_t)scales[k], fixed_scale) : (__int128)z;
    // "not enough values": the last terminator knows how many values exist
    if (p == len - 1 && k + 1 < needed) report_err64(err, k + 1, ORC_E_IO);
  }
  atomicMax(&stage_n, (uint32_t)(kend - k0));
  }
  }
  __syncthreads();
  const uint32_t nv = stage_n;
  for (uint32_t i = threadIdx.x; i < nv; i += 256) dense[k0 + i] = stage[i];
}
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
  if (valid) v = decimal_rescale(dense[d], (uint32_t)scales[d], fixed_scale);
  out[i] = v;
}
