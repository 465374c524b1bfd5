/*
 * abi_stripe.c -- include/orcgpu.h used from PLAIN C (C99), the way a cgo / Rust-FFI / JNI binding sees it: nothing but the
 * header and liborcgpu.so.  Stages one hand-made stripe (byte vectors of the reference's own unit tests), decodes it,
 * reads the batches back, applies a row selection, exports a batch through the Arrow C Data Interface, then reads a
 * fixture FILE through the orcgpu_reader_* front end.  Exit code 0 = every check passed.
 *
 *     gcc -std=c99 -Wall -Wextra -Werror -pedantic -I include tests/c_abi/abi_stripe.c -L orc_rust_amd/csrc -lorcgpu
 *     ./a.out tests/golden/data/test.orc          (compiled and run by tests/test_gpu_c_abi.py)
 *
 * Vectors: SHORT_REPEAT `0a 27 10` = 5 x 10000 (src/encoding/integer/rle_v2/short_repeat.rs tests); DIRECT
 * `5e 03 5c a1 ab 1e de ad be ef` = 23713, 43806, 57005, 48879 (rle_v2/direct.rs tests); a direct string column
 * (LENGTH run + DATA bytes, array_decoder/string.rs:111-153) and a PRESENT stream (boolean.rs:101-113).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orcgpu.h"

/* Arrow C Data Interface (the specification's own definitions) */
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
      failures++;                                                          \
    }                                                                      \
  } while (0)

int main(int argc, char** argv) {
  orcgpu_ctx* ctx = orcgpu_open(0, NULL);
  if (!ctx) {
    fprintf(stderr, "orcgpu_open failed: no usable HIP device (the library has no CPU path)\n");
    return 2;
  }
  printf("%s\n", orcgpu_version());

  /* ---- one stripe, 9 rows: column 1 Long (signed RLE v2: SHORT_REPEAT x5, then DIRECT x4 of the unsigned vector's bytes
   * read as zigzag), column 2 String direct with one null ---- */
  static const uint8_t long_data[] = {0x0a, 0x27, 0x10, 0x5e, 0x03, 0x5c, 0xa1, 0xab, 0x1e, 0xde, 0xad, 0xbe, 0xef};
  /* zigzag of the unsigned test values: 10000 -> 5000; 23713 -> -11857; 43806 -> 21903; 57005 -> -28503; 48879 -> -24440 */
  static const int64_t long_expect[9] = {5000, 5000, 5000, 5000, 5000, -11857, 21903, -28503, -24440};
  /* PRESENT: 9 rows, row 4 null: bits 1111 0111 1 -> bytes f7 80; byte RLE literal run of 2 (header -2 = 0xfe) */
  static const uint8_t str_present[] = {0xfe, 0xf7, 0x80};
  /* LENGTH (unsigned RLE v2 DIRECT, 8 values of width 4 bits: 1 2 3 4 5 6 7 8): header 0x46 0x07 (width code 3 = 4 bits, len 8) */
  static const uint8_t str_length[] = {0x46, 0x07, 0x12, 0x34, 0x56, 0x78};
  static const char str_data[] = "abbcccddddeeeeeffffffggggggghhhhhhhh";
  orcgpu_stream streams[4];
  memset(streams, 0, sizeof(streams));
  streams[0].column_id = 1; streams[0].kind = ORCGPU_S_DATA; streams[0].ptr = long_data; streams[0].len = sizeof(long_data);
  streams[1].column_id = 2; streams[1].kind = ORCGPU_S_PRESENT; streams[1].ptr = str_present; streams[1].len = sizeof(str_present);
  streams[2].column_id = 2; streams[2].kind = ORCGPU_S_LENGTH; streams[2].ptr = str_length; streams[2].len = sizeof(str_length);
  streams[3].column_id = 2; streams[3].kind = ORCGPU_S_DATA; streams[3].ptr = (const uint8_t*)str_data; streams[3].len = 36;
  orcgpu_column cols[2];
  memset(cols, 0, sizeof(cols));
  cols[0].column_id = 1; cols[0].orc_type = ORCGPU_T_LONG; cols[0].encoding = ORCGPU_ENC_DIRECT_V2;
  cols[1].column_id = 2; cols[1].orc_type = ORCGPU_T_STRING; cols[1].encoding = ORCGPU_ENC_DIRECT_V2;
  orcgpu_stripe_desc desc;
  memset(&desc, 0, sizeof(desc));
  desc.n_rows = 9;
  desc.compression = ORCGPU_COMP_NONE;
  desc.batch_size = 4; /* three batches: 4 + 4 + 1 */
  desc.n_streams = 4; desc.streams = streams;
  desc.n_columns = 2; desc.columns = cols;

  orcgpu_staged* staged = NULL;
  orcgpu_result* res = NULL;
  int rc = orcgpu_stage_stripe(ctx, &desc, &staged);
  if (rc) fprintf(stderr, "stage: %d %s\n", rc, orcgpu_last_error(ctx));
  CHECK(rc == ORCGPU_OK && staged != NULL);
  CHECK(orcgpu_staged_bytes(staged) == sizeof(long_data) + sizeof(str_present) + sizeof(str_length) + 36);
  rc = orcgpu_decode_staged(ctx, &staged, 1, &res);
  if (rc) fprintf(stderr, "decode: %d %s\n", rc, orcgpu_last_error(ctx));
  CHECK(rc == ORCGPU_OK && res != NULL);
  uint32_t eb = 0, ec = 0;
  CHECK(orcgpu_result_status(res, &eb, &ec) == ORCGPU_OK);
  CHECK(orcgpu_result_rows(res) == 9 && orcgpu_result_batches(res) == 3);

  {
    int64_t got[9];
    size_t at = 0;
    for (uint32_t b = 0; b < 3; b++) {
      orcgpu_batch_view v;
      CHECK(orcgpu_result_batch_view(res, b, 0, &v) == ORCGPU_OK);
      CHECK(v.length == (b < 2 ? 4u : 1u) && v.null_count == 0 && v.validity == NULL && v.values_bytes == 8 * v.length);
      CHECK(orcgpu_result_copy_batch(ctx, res, b, 0, got + at, NULL, NULL) == ORCGPU_OK);
      at += (size_t)v.length;
    }
    CHECK(memcmp(got, long_expect, sizeof(got)) == 0);
    /* strings, batch 1 = rows 4..7: null, "ddddd"?  no: rows 0..3 take lengths 1 2 3 4; row 4 is null; rows 5..7 take 5 6 7 */
    orcgpu_batch_view v;
    CHECK(orcgpu_result_batch_view(res, 1, 1, &v) == ORCGPU_OK);
    CHECK(v.length == 4 && v.null_count == 1 && v.validity != NULL && v.offsets != NULL && v.values_bytes == 18);
    char chars[64];
    int32_t offs[5];
    uint8_t valid[1];
    CHECK(orcgpu_result_copy_batch(ctx, res, 1, 1, chars, offs, valid) == ORCGPU_OK);
    CHECK(offs[0] == 0 && offs[1] == 0 && offs[2] == 5 && offs[3] == 11 && offs[4] == 18);
    CHECK((valid[0] & 0x0f) == 0x0e);
    CHECK(memcmp(chars, "eeeeeffffffggggggg", 18) == 0);
  }

  /* ---- row selection: skip 2, select 3, skip 1, select 3 -> batches of rows [2,5) and [6,9) (mod.rs:302-365) ---- */
  {
    orcgpu_row_selector sel[4] = {{2, 1}, {3, 0}, {1, 1}, {3, 0}};
    uint64_t starts[8];
    uint32_t lens[8], n_out = 0, n_rest = 0;
    orcgpu_row_selector rest[8];
    CHECK(orcgpu_selection_batches(sel, 4, 9, 4, starts, lens, 8, &n_out, rest, 8, &n_rest) == ORCGPU_OK);
    CHECK(n_out == 2 && starts[0] == 2 && lens[0] == 3 && starts[1] == 6 && lens[1] == 3 && n_rest == 0);
    CHECK(orcgpu_result_select(ctx, res, sel, 4) == ORCGPU_OK);
    CHECK(orcgpu_result_batches(res) == 2);
    int64_t got[3];
    CHECK(orcgpu_result_copy_batch(ctx, res, 1, 0, got, NULL, NULL) == ORCGPU_OK);
    CHECK(got[0] == long_expect[6] && got[1] == long_expect[7] && got[2] == long_expect[8]);
    /* ... and the same batch as an Arrow struct array in host memory */
    struct ArrowArray a;
    struct ArrowSchema s;
    CHECK(orcgpu_result_export_batch(ctx, res, 0, &a, &s) == ORCGPU_OK);
    CHECK(a.length == 3 && a.n_children == 2 && s.n_children == 2 && strcmp(s.format, "+s") == 0);
    CHECK(strcmp(s.children[0]->format, "l") == 0 && strcmp(s.children[1]->format, "u") == 0);
    CHECK(a.children[0]->length == 3 && ((const int64_t*)a.children[0]->buffers[1])[2] == long_expect[4]);
    CHECK(a.children[1]->null_count == 1);
    {
      const int32_t* o = (const int32_t*)a.children[1]->buffers[1];
      const char* c = (const char*)a.children[1]->buffers[2];
      CHECK(o[0] == 0 && o[1] == 3 && o[2] == 7 && o[3] == 7 && memcmp(c, "cccdddd", 7) == 0); /* rows 2, 3, 4 (null) */
    }
    a.release(&a);
    s.release(&s);
    CHECK(a.release == NULL && s.release == NULL);
  }
  orcgpu_result_free(res);
  orcgpu_staged_free(staged);

  /* ---- a writer time zone through the ABI (host only) ---- */
  {
    int64_t when[2] = {0, 1700000000};
    int32_t offs[2] = {1, 1};
    int64_t epoch = 0;
    CHECK(orcgpu_timezone_offsets("UTC", when, 2, offs, &epoch) == ORCGPU_OK);
    CHECK(offs[0] == 0 && offs[1] == 0 && epoch == 1420070400);
  }

  /* ---- the file front end: every batch of a fixture file ---- */
  if (argc > 1) {
    orcgpu_reader* rd = NULL;
    rc = orcgpu_reader_open_file(ctx, argv[1], &rd);
    if (rc) fprintf(stderr, "open %s: %d %s\n", argv[1], rc, orcgpu_last_error(ctx));
    CHECK(rc == ORCGPU_OK && rd != NULL);
    if (rd) {
      const char* want[2] = {"int64", "utf8"};
      if (argc > 3) { want[0] = argv[2]; want[1] = argv[3]; }
      CHECK(orcgpu_reader_set_projection(rd, want, 2) == ORCGPU_OK);
      CHECK(orcgpu_reader_set_batch_size(rd, 2) == ORCGPU_OK);
      uint64_t total = orcgpu_reader_total_rows(rd), seen = 0;
      CHECK(orcgpu_reader_column_count(rd) == 2);
      for (;;) {
        struct ArrowArray a;
        struct ArrowSchema s;
        rc = orcgpu_reader_next_batch(rd, &a, &s);
        if (rc == ORCGPU_END_OF_FILE) break;
        CHECK(rc == ORCGPU_OK);
        if (rc) break;
        CHECK(a.n_children == 2 && a.length <= 2);
        seen += (uint64_t)a.length;
        a.release(&a);
        s.release(&s);
      }
      CHECK(seen == total && total > 0);
      printf("%s: %llu rows in batches of 2\n", argv[1], (unsigned long long)seen);
      orcgpu_reader_close(rd);
    }
  }
  orcgpu_close(ctx);
  if (failures) {
    fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  printf("c abi ok\n");
  return 0;
}
