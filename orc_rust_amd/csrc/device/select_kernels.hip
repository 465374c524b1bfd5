// select_kernels.hip -- per-batch buffers of a ROW SELECTION over a decoded stripe.
//
// The reference steps a stripe's decoders through a RowSelection (skip / select runs): every select step of at most
// batch_size rows becomes a RecordBatch (array_decoder/mod.rs:302-365).  Here the stripe is decoded whole, in uniform
// batches; a selected batch is then a row range [start, start + len) of it.  Fixed-width values and string bytes of such
// a range are contiguous in the stripe-wide buffers and are used in place; what has to be rebuilt per selected batch is
// what the decoder keeps per batch: the validity bitmap and its null count, Boolean value bits, and string offsets
// (restarting at 0, string.rs:139-140).  One launch for all columns: blockIdx.x = selected batch, blockIdx.y = column.
#pragma once
#include <stdint.h>

struct SelBatch {
  uint64_t start;  // first row of the selected batch within the stripe
  uint32_t len;    // rows (<= batch)
  uint32_t pad;
};

struct SelJob {
  const unsigned long long* src_validity;  // per uniform batch: words_per_batch words (null: the column has no PRESENT stream)
  const unsigned long long* src_bool;      // Boolean columns: value bits, same layout (else null)
  const int32_t* src_offsets;              // strings: per uniform batch batch + 1 offsets restarting at 0 (else null)
  const unsigned long long* src_char_base; // strings: byte position of every uniform batch's first value byte (device copy)
  unsigned long long* out_validity;        // per selected batch: words_per_batch words
  unsigned long long* out_bool;
  int32_t* out_offsets;                    // per selected batch: batch + 1
  unsigned long long* out_nulls;           // per selected batch (zeroed by the host)
  unsigned long long* out_char_start;      // per selected batch: byte position of its first value byte in the stripe's bytes
  unsigned long long* out_char_total;      // per selected batch: its value bytes
  uint32_t batch, words_per_batch;
};

extern "C" __global__ void __launch_bounds__(256) select_build_kernel(const SelJob* jobs, const SelBatch* batches) {
  const SelJob j = jobs[blockIdx.y];
  const SelBatch sb = batches[blockIdx.x];
  const uint32_t B = j.batch, W = j.words_per_batch;
  const uint64_t ob = blockIdx.x;
  auto abs_pos = [&](uint64_t row, uint32_t plus) -> unsigned long long {  // byte position of row's value start (+1: its end)
    const uint64_t u = row / B, l = row % B;
    return j.src_char_base[u] + (unsigned long long)(uint32_t)j.src_offsets[u * (B + 1) + l + plus];
  };
  const unsigned long long abs0 = (j.src_offsets && sb.len) ? abs_pos(sb.start, 0) : 0ull;
  for (uint32_t i0 = 0; i0 < sb.len; i0 += 256) {
    const uint32_t i = i0 + threadIdx.x;
    const bool live = i < sb.len;
    const uint64_t row = sb.start + (live ? i : 0);
    const uint64_t u = row / B, l = row % B;
    if (j.src_validity || j.src_bool) {
      const bool valid = !j.src_validity || ((j.src_validity[u * W + (l >> 6)] >> (l & 63)) & 1);
      const unsigned long long vm = __ballot(live && valid);
      const unsigned long long lm = __ballot(live);
      if (j.src_validity) {
        if ((threadIdx.x & 63) == 0 && lm) {
          j.out_validity[ob * W + (i >> 6)] = vm;
          const uint32_t nulls = (uint32_t)__builtin_popcountll(lm & ~vm);
          if (nulls) atomicAdd(&j.out_nulls[ob], (unsigned long long)nulls);
        }
      }
      if (j.src_bool) {
        const bool bit = (j.src_bool[u * W + (l >> 6)] >> (l & 63)) & 1;
        const unsigned long long bm = __ballot(live && bit);
        if ((threadIdx.x & 63) == 0 && lm) j.out_bool[ob * W + (i >> 6)] = bm;
      }
    }
    if (j.src_offsets && live) {
      j.out_offsets[ob * (B + 1) + i] = (int32_t)(abs_pos(row, 0) - abs0);
      if (i + 1 == sb.len) {
        const unsigned long long end = abs_pos(row, 1);
        j.out_offsets[ob * (B + 1) + i + 1] = (int32_t)(end - abs0);
        j.out_char_start[ob] = abs0;
        j.out_char_total[ob] = end - abs0;
      }
    }
  }
  if (j.src_offsets && sb.len == 0 && threadIdx.x == 0) {
    j.out_offsets[ob * (B + 1)] = 0;
    j.out_char_start[ob] = 0;
    j.out_char_total[ob] = 0;
  }
}
