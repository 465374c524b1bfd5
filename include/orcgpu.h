/*
 * orcgpu.h -- C ABI of the MI355X-native ORC stripe decoder (liborcgpu.so).
 *
 * Drop-in boundary.  In datafusion-contrib/orc-rust the stripe -> Arrow hot path sits behind a
 * Rust trait-object seam, not an FFI:
 *     NaiveStripeDecoder::new_with_selection(stripe, schema_ref, batch_size, row_selection)
 *         src/array_decoder/mod.rs:570-594, installed at src/arrow_reader.rs:310-316
 *     trait ArrayBatchDecoder::next_batch(batch_size, parent_present) -> ArrayRef
 *         src/array_decoder/mod.rs:61-85, built by array_decoder_factory  :390-511
 * The inputs at that seam are `Stripe { columns, stream_map: HashMap<(u32, Kind), Bytes>,
 * compression, number_of_rows, tz }` (src/stripe.rs:119-125, :311-316).  This header is what a
 * Rust shim implementing `Iterator<Item = Result<RecordBatch>>` binds instead of building the
 * CPU decoders (see INTEGRATION.md for the `extern "C"` block and the arrow::ffi import).
 *
 * Everything below is plain C: pointers, sizes and POD structs; no C++/torch types.
 * Threading: a context is thread-compatible (one thread at a time, may migrate), like
 * `ArrayBatchDecoder: Send` (array_decoder/mod.rs:61) -- with one exception that pipelines need: orcgpu_stage_stripe may run
 * in one thread while another one decodes / fetches / frees on the same context.
 */
#ifndef ORCGPU_H
#define ORCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: 1:1 with the OrcError variants the path raises (src/error.rs:31-174) ---- */
enum {
  ORCGPU_OK = 0,
  ORCGPU_IO_ERROR = 1,          /* IoError (short read / unexpected end of stream)          */
  ORCGPU_OUT_OF_SPEC = 2,       /* OutOfSpec { msg }                                        */
  ORCGPU_VARINT_TOO_LARGE = 3,  /* VarintTooLarge                                           */
  ORCGPU_DECODE_TIMESTAMP = 4,  /* DecodeTimestamp { seconds, nanoseconds, to_time_unit }   */
  ORCGPU_OFFSET_OVERFLOW = 5,   /* OffsetOverflow { total_length, max_size, batch_size }    */
  ORCGPU_MISMATCHED_SCHEMA = 6, /* MismatchedSchema { orc_type, arrow_type }                */
  ORCGPU_UNSUPPORTED = 7,       /* UnsupportedTypeVariant { msg }                           */
  ORCGPU_ARROW = 8,             /* Arrow (UTF-8 validation, dictionary key, offsets)        */
  ORCGPU_BUILD_DECODER = 9,     /* Build{Zstd,Snappy,Lz4}Decoder / inflate failure          */
  ORCGPU_UNEXPECTED = 10,       /* Unexpected { msg }                                       */
  ORCGPU_HIP_ERROR = 100,       /* HIP runtime failure (no device, out of memory, ...)      */
  ORCGPU_INVALID_ARGUMENT = 101,
  ORCGPU_END_OF_FILE = 110      /* not an error: orcgpu_reader_next_batch has handed out every batch (Iterator::next -> None) */
};

/* proto CompressionKind (format/orc_proto.proto:383-390) */
enum { ORCGPU_COMP_NONE = 0, ORCGPU_COMP_ZLIB = 1, ORCGPU_COMP_SNAPPY = 2, ORCGPU_COMP_LZO = 3, ORCGPU_COMP_LZ4 = 4, ORCGPU_COMP_ZSTD = 5 };

/* proto Type.Kind (format/orc_proto.proto:199-228; src/schema.rs:241-320) */
enum {
  ORCGPU_T_BOOLEAN = 0, ORCGPU_T_BYTE = 1, ORCGPU_T_SHORT = 2, ORCGPU_T_INT = 3, ORCGPU_T_LONG = 4, ORCGPU_T_FLOAT = 5,
  ORCGPU_T_DOUBLE = 6, ORCGPU_T_STRING = 7, ORCGPU_T_BINARY = 8, ORCGPU_T_TIMESTAMP = 9, ORCGPU_T_LIST = 10,
  ORCGPU_T_MAP = 11, ORCGPU_T_STRUCT = 12, ORCGPU_T_UNION = 13, ORCGPU_T_DECIMAL = 14, ORCGPU_T_DATE = 15,
  ORCGPU_T_VARCHAR = 16, ORCGPU_T_CHAR = 17, ORCGPU_T_TIMESTAMP_INSTANT = 18
};

/* proto Stream.Kind (format/orc_proto.proto:125-143) */
enum { ORCGPU_S_PRESENT = 0, ORCGPU_S_DATA = 1, ORCGPU_S_LENGTH = 2, ORCGPU_S_DICTIONARY_DATA = 3, ORCGPU_S_DICTIONARY_COUNT = 4,
       ORCGPU_S_SECONDARY = 5, ORCGPU_S_ROW_INDEX = 6, ORCGPU_S_BLOOM_FILTER = 7, ORCGPU_S_BLOOM_FILTER_UTF8 = 8 };

/* proto ColumnEncoding.Kind (format/orc_proto.proto:150-155; src/column.rs:47-59) */
enum { ORCGPU_ENC_DIRECT = 0, ORCGPU_ENC_DICTIONARY = 1, ORCGPU_ENC_DIRECT_V2 = 2, ORCGPU_ENC_DICTIONARY_V2 = 3 };

/* Arrow target of a column = the `hinted_arrow_type` of array_decoder_factory (mod.rs:390-394).
 * 0 picks the default mapping of src/schema.rs:503-579. */
enum {
  ORCGPU_ARROW_DEFAULT = 0,
  /* the default mapping at another time unit (with_timestamp_precision): Timestamp -> Timestamp(unit), TimestampInstant -> Timestamp(unit, "UTC") */
  ORCGPU_ARROW_TIMESTAMP_S = 1, ORCGPU_ARROW_TIMESTAMP_MS = 2, ORCGPU_ARROW_TIMESTAMP_US = 3, ORCGPU_ARROW_TIMESTAMP_NS = 4,
  /* an explicit Arrow type (with_schema): the pairs array_decoder_factory accepts decode, every other pair is
   * ORCGPU_MISMATCHED_SCHEMA (mod.rs:390-511, timestamp.rs:149-232) */
  ORCGPU_ARROW_BOOLEAN = 10, ORCGPU_ARROW_INT8 = 11, ORCGPU_ARROW_INT16 = 12, ORCGPU_ARROW_INT32 = 13, ORCGPU_ARROW_INT64 = 14,
  ORCGPU_ARROW_FLOAT32 = 15, ORCGPU_ARROW_FLOAT64 = 16, ORCGPU_ARROW_UTF8 = 17, ORCGPU_ARROW_BINARY = 18, ORCGPU_ARROW_DATE32 = 19,
  ORCGPU_ARROW_DECIMAL128 = 20,  /* precision / scale in arrow_precision / arrow_scale; (38, 9) is also the wide target of timestamps */
  ORCGPU_ARROW_TIMESTAMP_S_UTC = 21, ORCGPU_ARROW_TIMESTAMP_MS_UTC = 22, ORCGPU_ARROW_TIMESTAMP_US_UTC = 23, ORCGPU_ARROW_TIMESTAMP_NS_UTC = 24,
  ORCGPU_ARROW_LARGE_UTF8 = 25, ORCGPU_ARROW_LARGE_BINARY = 26,  /* (the encoder's inputs only: i64 offsets) */
  ORCGPU_ARROW_TIMESTAMP_S_NOTZ = 31, ORCGPU_ARROW_TIMESTAMP_MS_NOTZ = 32, ORCGPU_ARROW_TIMESTAMP_US_NOTZ = 33, ORCGPU_ARROW_TIMESTAMP_NS_NOTZ = 34,
  ORCGPU_ARROW_TIMESTAMP_OTHER_TZ = 35,  /* Timestamp(_, Some(tz)) with tz != "UTC": UnsupportedTypeVariant for TimestampInstant */
  /* nested targets (array_decoder/mod.rs:464-505): the hint of a Struct / List / Map / Union column; its children carry theirs */
  ORCGPU_ARROW_STRUCT = 40, ORCGPU_ARROW_LIST = 41, ORCGPU_ARROW_MAP = 42, ORCGPU_ARROW_UNION = 43,
  ORCGPU_ARROW_MAP_SORTED = 44,  /* Map with keys_sorted: UnsupportedTypeVariant "Sorted map" */
  ORCGPU_ARROW_OTHER = 99        /* any Arrow type the path has no decoder for */
};

typedef struct orcgpu_ctx orcgpu_ctx;          /* one GPU, one HIP stream, reusable workspace */
typedef struct orcgpu_staged orcgpu_staged;    /* stripe streams resident in HBM              */
typedef struct orcgpu_result orcgpu_result;    /* decoded Arrow buffers resident in HBM       */

typedef struct {
  uint64_t workspace_bytes;  /* initial HBM workspace (0 = grow on demand)  */
  uint32_t flags;            /* reserved, 0                                 */
} orcgpu_opts;

/* One entry of Stripe.stream_map (src/stripe.rs:311-316): raw, possibly compressed bytes. */
/* Where a row group starts in a stream: what a RowIndexEntry holds for it (row_index.rs:42-50; orcgpu_index_entry below deals
 * an entry's positions out to the column's streams). */
typedef struct {
  uint64_t chunk_offset;  /* where, in the stream's bytes, the entry's chunk header is (uncompressed file: the byte itself) */
  uint32_t skip_bytes;    /* bytes into the decompressed chunk (uncompressed: 0)                                         */
  uint32_t skip_values;   /* run-length streams: values (bit streams: bytes) of the run there already consumed            */
  uint32_t skip_bits;     /* bit streams: bits of that byte consumed                                                     */
} orcgpu_stream_entry;

typedef struct {
  uint32_t column_id;
  int32_t kind;        /* ORCGPU_S_* */
  const uint8_t* ptr;  /* HOST pointer (pageable or pinned) */
  uint64_t len;
  /* Entry point (0, 0: the stream is given from its start).  A stream may be given from a row group on: `ptr` then starts at
   * the chunk header (compressed file) or at the byte (uncompressed file) a ROW_INDEX position names, and the decoder of the
   * stream starts `skip_bytes` into the (decompressed) bytes and drops the first `skip_values` values it decodes there -- the
   * three numbers a RowIndexEntry holds per stream (row_index.rs:42-50; orcgpu_index_entry splits an entry's positions by
   * stream).  Dictionary streams (DICTIONARY_DATA, the LENGTH stream of a dictionary) are always given whole.  The bytes may
   * reach beyond what the rows need: what lies behind the values the stripe's rows consume is not looked at. */
  uint32_t skip_bytes;
  uint32_t skip_values; /* PRESENT / Boolean DATA: in bytes of the bit stream; the bits of the byte there already consumed: skip_bits below */
  /* Optional (NULL, 0: none): where the stripe's later row groups start in this stream, in stream order, chunk_offset counted
   * from `ptr` -- the ROW_INDEX positions of the stream.  Run-length streams use them as verified run starts: one lane per
   * row group follows the run headers from its entry to the next (some tens of dependent steps instead of the whole stream's),
   * which gives every 512-byte block of the stream its first header without the search the decoder otherwise makes for
   * them.  Only a hint: entries that do not lie on the stream's run chain are noticed and ignored. */
  const orcgpu_stream_entry* entries;
  uint32_t n_entries;
  uint32_t skip_bits;   /* PRESENT / Boolean DATA entered at a row group: bits (0..7) of the first byte that belong to the rows before */
} orcgpu_stream;

/* One projected leaf column: what Column / DataType / ColumnEncoding carry (src/column.rs:24-59). */
typedef struct {
  uint32_t column_id;
  int32_t orc_type;          /* ORCGPU_T_*                                   */
  int32_t encoding;          /* ORCGPU_ENC_*  (rle version = column.rs:52-59) */
  uint32_t dictionary_size;  /* column.rs:40-45                              */
  uint32_t precision, scale; /* DECIMAL                                      */
  int32_t arrow_target;      /* ORCGPU_ARROW_*                               */
  uint32_t arrow_precision, arrow_scale; /* ORCGPU_ARROW_DECIMAL128           */
  uint32_t parent;           /* 0: a root column; else 1 + the index in `columns` of the STRUCT this column is a field of
                              * (Column::children, column.rs; parents come before their children).  A STRUCT column is
                              * given with orc_type ORCGPU_T_STRUCT: it decodes to a validity bitmap, and its fields' validity
                              * is theirs merged with it (array_decoder/struct_decoder.rs:58-78, mod.rs:216-252) */
} orcgpu_column;

typedef struct {
  uint64_t n_rows;           /* Stripe.number_of_rows                                              */
  int32_t compression;       /* ORCGPU_COMP_* (Compression::from_proto, src/compression.rs:52-83)  */
  uint64_t block_size;       /* max_decompressed_block_size, 0 = 262144 (compression.rs:31).  What a chunk of a conforming writer
                              * expands to at most, and what lz4_flex enforces (compression.rs:185-195).  flate2, lzokay and the
                              * zstd crate grow their output instead: a chunk of theirs that does not fit is decoded again with
                              * room for max(block_size, 4 MiB), beyond which it is ORCGPU_BUILD_DECODER */
  int64_t ts_base_seconds;   /* ORC epoch in the writer timezone (array_decoder/timestamp.rs:133-147); 0 = 1420070400 */
  uint32_t batch_size;       /* rows per RecordBatch, 0 = 8192 (arrow_reader.rs:37)                */
  uint32_t n_streams;
  const orcgpu_stream* streams;
  uint32_t n_columns;
  const orcgpu_column* columns;
  const char* writer_timezone; /* StripeFooter.writer_timezone (stripe.rs:167-171) or NULL.  When given it names a zone of the
                              * system's tz database ($TZDIR, /usr/share/zoneinfo; ORCGPU_UNSUPPORTED if it is not there):
                              * ts_base_seconds is then taken from it and TIMESTAMP columns are re-labelled from that zone
                              * to UTC on the device (array_decoder/timestamp.rs:236-291; values chrono cannot hold become
                              * nulls, as there).  TIMESTAMP_INSTANT columns ignore it. */
} orcgpu_stripe_desc;

/* ---- context ---------------------------------------------------------------------------------- */
/* Returns NULL on failure (no HIP device: the library never falls back to a CPU path). */
orcgpu_ctx* orcgpu_open(int device, const orcgpu_opts* opts);
void orcgpu_close(orcgpu_ctx* ctx);
const char* orcgpu_last_error(const orcgpu_ctx* ctx);
const char* orcgpu_version(void);
/* The binary interface a caller was built against: bumped whenever a struct of this header changes size or layout, or an entry
 * point changes its meaning.  3 (round 6): orcgpu_lane_stats gained literals_kernel_ms; 2 (round 5): orcgpu_reader_next_batch
 * ends with ORCGPU_END_OF_FILE (110) instead of 1, orcgpu_stream gained skip_bits.  A binding checks orcgpu_abi_version() ==
 * ORCGPU_ABI_VERSION once, after loading the library. */
#define ORCGPU_ABI_VERSION 3
int orcgpu_abi_version(void);

/* ---- staging: host stream bytes -> HBM --------------------------------------------------------- */
/* Copies every stream of the stripe into one HBM arena (taken from a pool) and scans the 3-byte chunk headers
 * (compression.rs:113-123, :244-267; Zstandard: also the frame / block headers) on the host while the bytes are at hand.
 * The copy is a pipeline: 16 MiB pieces go through two pinned buffers, filled by a few host threads while the previous
 * piece is on its way (hipMemcpyAsync on a copy stream).  The call does NOT wait for the last piece: decodes wait for
 * it on the device, so staging stripe k + 1 overlaps decoding stripe k (the analogue of the reference's
 * Stripe::new reading ahead, stripe.rs:127-182 / async_arrow_reader.rs:165-280).  The caller's buffers may be
 * reused as soon as the call returns.  The returned handle keeps its own copy of the descriptor. */
int orcgpu_stage_stripe(orcgpu_ctx* ctx, const orcgpu_stripe_desc* desc, orcgpu_staged** out);
void orcgpu_staged_free(orcgpu_staged* s);
uint64_t orcgpu_staged_bytes(const orcgpu_staged* s); /* sum of Stream.length staged in HBM */

/* ---- decode: staged stripes -> Arrow buffers in HBM -------------------------------------------- */
/* Decodes n staged stripes with shared kernel launches.  Device work only; ends with ONE small
 * device-to-host copy (error words, null counts, string byte totals).  results[i] is allocated
 * unless it is passed non-NULL from an earlier decode of a same-shaped stripe (buffer reuse). */
int orcgpu_decode_staged(orcgpu_ctx* ctx, orcgpu_staged* const* stripes, uint32_t n, orcgpu_result** results);
/* Convenience: stage + decode one stripe. */
int orcgpu_stripe_decode(orcgpu_ctx* ctx, const orcgpu_stripe_desc* desc, orcgpu_result** out);
void orcgpu_result_free(orcgpu_result* r);

/* ---- result inspection ------------------------------------------------------------------------- */
/* Decode status of the stripe: ORCGPU_OK, or the error of the FIRST failing batch; *batch gets the
 * index of that batch (batches before it are valid, as with the reference's iterator, which
 * yields Ok batches until the failing one: arrow_reader.rs:333-346). */
int orcgpu_result_status(const orcgpu_result* r, uint32_t* batch, uint32_t* column);
uint64_t orcgpu_result_rows(const orcgpu_result* r);
uint32_t orcgpu_result_batches(const orcgpu_result* r);
uint64_t orcgpu_result_arrow_bytes(const orcgpu_result* r); /* value + offset + validity bytes emitted */

/* Raw view of one column of one batch.  Pointers are DEVICE pointers unless copied with
 * orcgpu_result_copy_batch; layouts follow the Arrow columnar format:
 *   values    fixed width data / Boolean bitmap (LSB first) / string bytes of this batch
 *   offsets   length+1 int32, first = 0 (restart per batch, array_decoder/string.rs:139-140)
 *   validity  LSB-first bitmap, NULL when null_count == 0 (array_decoder/mod.rs:247-251) */
typedef struct {
  uint64_t length;
  uint64_t null_count;
  const void* validity;
  const void* values;
  uint64_t values_bytes;
  const int32_t* offsets;
} orcgpu_batch_view;
int orcgpu_result_batch_view(const orcgpu_result* r, uint32_t batch, uint32_t column, orcgpu_batch_view* out);
/* Copies one column-batch to host memory supplied by the caller (sizes from the view). */
int orcgpu_result_copy_batch(orcgpu_ctx* ctx, const orcgpu_result* r, uint32_t batch, uint32_t column, void* values,
                             int32_t* offsets, void* validity);

/* ---- row selection --------------------------------------------------------------------------------------- */
/* One run of a RowSelection (src/row_selection.rs:32-52): row_count rows skipped or selected. */
typedef struct {
  uint64_t row_count;
  int32_t skip;  /* nonzero: skip, zero: select */
} orcgpu_row_selector;
/* Applies the stripe's part of a RowSelection to a decoded stripe -- the `row_selection` argument of
 * NaiveStripeDecoder::new_with_selection (src/array_decoder/mod.rs:570-594).  Afterwards the result's batches are the
 * RecordBatches next_with_row_selection yields (mod.rs:302-365): one per select step of at most batch_size rows, in
 * order; orcgpu_result_batches / _batch_view / _copy_batch / _export_batch speak in those batches.  The stripe was
 * decoded whole (the reference's skip_values decodes and discards too, rle_v2/mod.rs:148-175); values and string bytes
 * of a selected batch are used in place, its validity bitmap, Boolean bits and offsets are rebuilt on the device.
 * Selectors are normalised like `From<Vec<RowSelector>>` (row_selection.rs:466-482). */
int orcgpu_result_select(orcgpu_ctx* ctx, orcgpu_result* r, const orcgpu_row_selector* selectors, uint32_t n);
/* Host only: one stream's share of a RowIndexEntry's positions (row_index.rs:42-50, :204-226; the reference keeps the list
 * and never seeks with it).  `positions` is the entry of ONE row group of ONE column as the ROW_INDEX stream holds it;
 * the streams of the column take their numbers from it in the order PRESENT, DATA, then LENGTH or SECONDARY:
 * a compressed file gives every stream {offset of a chunk header in the stream, bytes into the decompressed chunk}, an
 * uncompressed one {byte offset}; run-length streams add {values of the run at that byte already consumed}, bit streams
 * (PRESENT, Boolean DATA) after that {bits of the current byte consumed}; dictionary streams have no positions.
 * Describe the column as it is in the stripe: `has_present` = it has a PRESENT stream, `compressed` = the file has a
 * compression codec.  Returns the numbers for stream `kind`, or ORCGPU_OUT_OF_SPEC when the entry has not exactly the
 * positions such a column needs (writers that drop an all-ones PRESENT stream drop its positions too). */
int orcgpu_index_entry(const orcgpu_column* column, int has_present, int compressed, const uint64_t* positions, uint32_t n_positions,
                       int32_t kind, orcgpu_stream_entry* out);
/* Host only: the UTC offsets (seconds east of Greenwich) the library uses for a writer time zone at n instants (seconds since
 * the UNIX epoch), and the ORC epoch in that zone (2015-01-01T00:00:00 there, timestamp.rs:133-147; orc_epoch may be NULL).
 * ORCGPU_UNSUPPORTED when the tz database does not have the zone. */
int orcgpu_timezone_offsets(const char* name, const int64_t* instants, uint32_t n, int32_t* offsets, int64_t* orc_epoch);
/* Host only: the stepping by itself.  Splits the first stripe_rows rows off the selection (RowSelection::split_off,
 * row_selection.rs:278-314), runs next_with_row_selection over them and reports the batches as row ranges of the stripe
 * (starts / lens, at most cap of them; *n_out = how many there are) and what is left of the selection (rest). */
int orcgpu_selection_batches(const orcgpu_row_selector* selectors, uint32_t n, uint64_t stripe_rows, uint32_t batch_size, uint64_t* starts,
                             uint32_t* lens, uint32_t cap, uint32_t* n_out, orcgpu_row_selector* rest, uint32_t rest_cap, uint32_t* n_rest);

/* Brings every Arrow buffer of the result to the host at once: one hipMemcpyAsync per result arena into pinned memory
 * (the counterpart of the reference handing out freshly allocated host buffers, array_decoder/mod.rs:100-120), one
 * synchronisation.  orcgpu_result_export_batch calls it on demand; the exported batches are views into that copy and
 * keep it alive after orcgpu_result_free. */
int orcgpu_result_fetch(orcgpu_ctx* ctx, orcgpu_result* r);
/* The same without the wait: the copies are enqueued on the context's device-to-host stream (behind the work enqueued on the
 * decode stream so far) and the call returns; the next stripe can be staged and decoded meanwhile.  orcgpu_result_fetch /
 * _export_batch wait for them; decoding into the result again, selecting on it or freeing it waits too.  With
 * orcgpu_stage_stripe (which does not wait for its copies either) this is what lets a caller keep three stripes in flight:
 * staging k + 1, decoding k, copying k - 1 back -- the analogue of the reference's async reader fetching the next stripe
 * while the current one is consumed (async_arrow_reader.rs:165-280). */
int orcgpu_result_fetch_async(orcgpu_ctx* ctx, orcgpu_result* r);

/* ---- Arrow C Data Interface export (https://arrow.apache.org/docs/format/CDataInterface.html) ---- */
struct ArrowSchema;
struct ArrowArray;
/* Exports batch `batch` as a struct array (one child per projected column) into HOST memory owned
 * by the release callbacks; a Rust caller imports it with arrow::ffi::from_ffi, Python with
 * pyarrow.RecordBatch._import_from_c.  Field names are "c<column_id>"; the caller renames. */
int orcgpu_result_export_batch(orcgpu_ctx* ctx, const orcgpu_result* r, uint32_t batch, struct ArrowArray* out_array,
                               struct ArrowSchema* out_schema);

/* ---- file reader: host-side mirror of ArrowReaderBuilder / ArrowReader ---------------------------------- */
/* The container layer around the hot path, restated in C++ inside liborcgpu.so because it produces
 * the kernel inputs: ChunkReader (src/reader/mod.rs:27-76), read_metadata (src/reader/metadata.rs:
 * 180-247), Stripe::new (src/stripe.rs:127-182), ProjectionMask::named_roots (src/projection.rs:52-69),
 * ArrowReaderBuilder::{try_new, with_batch_size, with_projection, with_file_byte_range,
 * with_timestamp_precision, build} (src/arrow_reader.rs:70-231) and ArrowReader::{next,
 * total_row_count} (:243-346).  Setters must be called before the first next_batch ("build"). */
typedef struct orcgpu_reader orcgpu_reader;
int orcgpu_reader_open_file(orcgpu_ctx* ctx, const char* path, orcgpu_reader** out);                 /* try_new(File)  */
int orcgpu_reader_open_bytes(orcgpu_ctx* ctx, const uint8_t* data, uint64_t len, orcgpu_reader** out); /* try_new(Bytes) */
void orcgpu_reader_close(orcgpu_reader* r);
int orcgpu_reader_set_batch_size(orcgpu_reader* r, uint32_t batch_size);                             /* with_batch_size */
int orcgpu_reader_set_projection(orcgpu_reader* r, const char* const* root_names, uint32_t n);       /* named_roots     */
int orcgpu_reader_set_projection_roots(orcgpu_reader* r, const uint32_t* root_indices, uint32_t n);  /* ProjectionMask::roots (projection.rs:37): indices of root columns */
/* with_schema (arrow_reader.rs:80): the Arrow schema to decode into, as an Arrow C Data Interface struct ("+s") whose
 * children are matched with the projected root columns by position (NaiveStripeDecoder::new zips columns and fields,
 * array_decoder/mod.rs:577-582).  Field names become the batches' column names; a type pair the reference has no decoder
 * for makes the first orcgpu_reader_next_batch return ORCGPU_MISMATCHED_SCHEMA.  The schema is read, not consumed. */
int orcgpu_reader_set_schema(orcgpu_reader* r, const struct ArrowSchema* schema);
int orcgpu_reader_set_byte_range(orcgpu_reader* r, uint64_t start, uint64_t end);                    /* with_file_byte_range */
/* `world` processes, one per GPU, read ONE file together; this reader is number `rank` of them.  The units of the
 * path are independent (stripe.rs:154-165: a stripe's columns are decoded one by one), so no data is exchanged -- each reader
 * reads, stages and decodes only what is its own; what the ranks do with their batches (an all-gather of row counts,
 * concatenation by column or by stripe) is the caller's.  The reference's own split hook is a byte range per reader
 * (arrow_reader.rs:86-89, :358-372).
 *   ORCGPU_SHARD_STRIPES  stripe k (of those the byte range leaves) belongs to rank k % world: whole stripes, every projected
 *                         column.  A row selection is stepped through every stripe, read or not, so the ranks' batches put
 *                         together in stripe order are the batches of a single reader.
 *   ORCGPU_SHARD_COLUMNS  the projected root columns are dealt out by their estimated Arrow bytes per row (orcgpu_shard_columns:
 *                         largest first, to the least loaded rank): every rank reads every stripe, its own columns only; its
 *                         batches have the same rows as a single reader's, the ranks' columns put side by side are its columns.
 * world = 1 (the default) reads everything. */
enum { ORCGPU_SHARD_STRIPES = 0, ORCGPU_SHARD_COLUMNS = 1 };
int orcgpu_reader_set_shard(orcgpu_reader* r, uint32_t rank, uint32_t world, int mode);
/* The column deal alone (host only, no GPU): rank_of[i] for n columns of the given weights.  Longest-processing-time rule:
 * columns in order of falling weight (ties: the earlier column first), each to the rank with the least weight so far (ties:
 * the lower rank).  What orcgpu_reader_set_shard(.., ORCGPU_SHARD_COLUMNS) uses, with orcgpu_reader_column_weight's weights. */
int orcgpu_shard_columns(const double* weights, uint32_t n, uint32_t world, uint32_t* rank_of);
/* Estimated Arrow bytes per row of root column `root` of the file (its index among the root columns): fixed widths, 16 for
 * Decimal128, 4 + 16 for strings and binaries, the sum of its fields for a Struct, offsets + two elements for Lists and Maps. */
double orcgpu_reader_column_weight(const orcgpu_reader* r, uint32_t root);
int orcgpu_reader_set_timestamp_precision(orcgpu_reader* r, int arrow_target);                       /* ORCGPU_ARROW_TIMESTAMP_* */
/* with_row_selection (arrow_reader.rs:113): a selection over the rows of the FILE; every stripe takes its share with
 * RowSelection::split_off, and once no rows are left in the selection later stripes are read whole (arrow_reader.rs:296-308). */
int orcgpu_reader_set_row_selection(orcgpu_reader* r, const orcgpu_row_selector* selectors, uint32_t n);
/* Row groups under a row selection (default: on).  The reference decodes a stripe from its first row and discards what a
 * selection skips (skip_values, rle_v2/mod.rs:148-175).  This reader reads the stripe's ROW_INDEX streams (row_index.rs:204-226)
 * and reads, stages and decodes only the row groups (rowIndexStride rows, stripe.rs:300) that hold selected rows -- every
 * stream from the entry point its index names, consecutive row groups together; a stripe without selected rows is not read
 * at all.  The batches are the ones the whole decode yields.  Stripes without usable indexes (no ROW_INDEX streams, an entry
 * that does not fit its column, bit streams entered in mid-byte, List / Map columns) are decoded whole.  What differs, on
 * damaged files only: an error inside row groups that are not read is not met.  `on` = 0 turns it off. */
int orcgpu_reader_set_row_group_pruning(orcgpu_reader* r, int on);
/* How many row groups the reader has read and how many the stripes it went through hold (whole stripes count all of theirs). */
int orcgpu_reader_row_groups(const orcgpu_reader* r, uint64_t* read, uint64_t* total);
/* ---- predicate pushdown onto row groups (src/predicate.rs, src/row_group_filter.rs, arrow_reader.rs:173, :258-308) ---- */
/* A Predicate as a list of nodes in pre-order: a node, then its children one after the other (AND / OR: n_children of them,
 * NOT: one; comparisons and null tests: none). */
enum { ORCGPU_PRED_EQ = 0, ORCGPU_PRED_NE = 1, ORCGPU_PRED_LT = 2, ORCGPU_PRED_LE = 3, ORCGPU_PRED_GT = 4, ORCGPU_PRED_GE = 5, /* ComparisonOp */
       ORCGPU_PRED_IS_NULL = 6, ORCGPU_PRED_IS_NOT_NULL = 7, ORCGPU_PRED_AND = 8, ORCGPU_PRED_OR = 9, ORCGPU_PRED_NOT = 10 };
/* PredicateValue (predicate.rs:29-47) */
enum { ORCGPU_PV_BOOLEAN = 0, ORCGPU_PV_INT8 = 1, ORCGPU_PV_INT16 = 2, ORCGPU_PV_INT32 = 3, ORCGPU_PV_INT64 = 4, ORCGPU_PV_FLOAT32 = 5,
       ORCGPU_PV_FLOAT64 = 6, ORCGPU_PV_UTF8 = 7 };
typedef struct {
  int32_t op;            /* ORCGPU_PRED_*                                                                 */
  uint32_t n_children;   /* AND / OR                                                                      */
  const char* column;    /* comparisons and null tests: the name of a projected root column               */
  int32_t value_type;    /* comparisons: ORCGPU_PV_*                                                       */
  int32_t value_is_null; /* ... Some / None                                                                */
  int64_t i;             /* Boolean (0 / 1) and the integer kinds                                          */
  double f;              /* Float32 / Float64                                                              */
  const char* s;         /* Utf8: bytes, not necessarily terminated                                        */
  uint64_t s_len;
} orcgpu_predicate_node;
/* What the reader knows about one column of one stripe: the plain (decompressed) bytes of its ROW_INDEX stream, a
 * proto RowIndex, and of its BLOOM_FILTER (else BLOOM_FILTER_UTF8) stream, a proto BloomFilterIndex (NULL, 0: none). */
typedef struct {
  const char* name;
  const uint8_t* row_index;
  uint64_t row_index_len;
  const uint8_t* bloom_index;
  uint64_t bloom_index_len;
} orcgpu_column_index;
/* Host only: evaluate_predicate (row_group_filter.rs:50-165) -- which row groups of a stripe might hold rows that satisfy the
 * predicate, judged by the row groups' statistics (min / max, null counts: :167-330, :470-620) and, for equality, their Bloom
 * filters (:332-371, bloom_filter.rs).  keep[g] = 1: row group g must be read; *n_groups = ceil(stripe_rows / rows_per_group)
 * (keep must have room for that many).  Anything but ORCGPU_OK means the reference's evaluation fails (a column that is not
 * there or has no index, a value of the wrong type, a row group without typed statistics): its reader then reads every row. */
int orcgpu_predicate_row_groups(const orcgpu_predicate_node* nodes, uint32_t n_nodes, const orcgpu_column_index* columns, uint32_t n_columns,
                                uint64_t stripe_rows, uint64_t rows_per_group, uint8_t* keep, uint32_t* n_groups);
/* with_predicate (arrow_reader.rs:173): per stripe the predicate is evaluated against the stripe's row indexes, the row groups
 * it keeps become a RowSelection (RowSelection::from_row_group_filter, row_selection.rs:348-392) and the stripe is decoded under
 * it -- here: only those row groups are read (orcgpu_reader_set_row_group_pruning).  A stripe whose evaluation fails is read
 * whole.  The nodes (and their strings) are copied.  With a row selection as well, a row is read when both select it (the
 * reference's own combination, arrow_reader.rs:296-308, panics unless the row selection selects every row of the stripe). */
int orcgpu_reader_set_predicate(orcgpu_reader* r, const orcgpu_predicate_node* nodes, uint32_t n_nodes);
/* Read-ahead: how many decoded stripes the reader may be ahead of the caller (default 2 -- every result set in flight costs a pinned host copy of a stripe --, at most 8; 0 = none: every stripe is
 * read, staged, decoded and copied back inside the orcgpu_reader_next_batch call that needs it).  With read-ahead two threads
 * of the reader work beside the caller: one reads and stages the stripes to come, one decodes the stripes staged so far
 * (several per call when decoding is the slower side) and starts their copies back, while the caller consumes the batches of
 * an earlier stripe (async_arrow_reader.rs:165-280: StreamState::Reading beside the decoding of the current stripe).  The
 * batches are the same either way.  The context must not be used for other calls while such a reader is open. */
int orcgpu_reader_set_prefetch(orcgpu_reader* r, uint32_t stripes);
uint64_t orcgpu_reader_total_rows(const orcgpu_reader* r);                                           /* total_row_count */
uint32_t orcgpu_reader_stripe_count(const orcgpu_reader* r);
uint32_t orcgpu_reader_column_count(orcgpu_reader* r);                                               /* projected flat columns */
const char* orcgpu_reader_column_name(orcgpu_reader* r, uint32_t i);
/* ArrowReader::next: 0 = one RecordBatch exported (struct array, host memory), ORCGPU_END_OF_FILE = no more batches,
 * anything else = the OrcError status of the failing batch (the iterator then ends). */
int orcgpu_reader_next_batch(orcgpu_reader* r, struct ArrowArray* out_array, struct ArrowSchema* out_schema);

/* ---- GPU encode: the reference's value encoders, byte for byte ------------------------------ */
/* Replace RleV2Encoder<N, S>::{write_slice, take_inner} (src/encoding/integer/rle_v2/mod.rs:255-531), ByteRleEncoder
 * (src/encoding/byte.rs:38-197) and BooleanEncoder (src/encoding/boolean.rs:119-170); the seam is PrimitiveValueEncoder
 * (src/encoding/mod.rs:36-50); a column's encoders are fed by src/writer/column.rs and flushed per stripe by
 * src/writer/stripe.rs:109-165.  The bytes are the reference encoder's own: the same runs (its greedy state machine's cuts, found
 * in parallel: device/rle_encode.hip), the same sub-encoding per run (SHORT_REPEAT / DIRECT / PATCHED_BASE / DELTA by
 * determine_variable_run_encoding's rules), the same bit widths.  tests/test_gpu_encode.py compares them with the restated
 * reference encoder (oracle/oo_encode.c).  (On two inputs the reference panics -- oo_encode.c's header --: such runs are DIRECT.)
 *
 * values / out: host memory, or device memory with ORCGPU_ENC_ON_DEVICE in flags (then nothing crosses PCIe but the size).
 * out = NULL only reports the size in *out_len; out_cap too small: ORCGPU_INVALID_ARGUMENT with the size in *out_len.
 * One call = one stream of a stripe (fewer than 2^32 - 1024 values).  Scratch memory is kept in the context between calls. */
#define ORCGPU_ENC_ON_DEVICE 1u
/* n values of int_bytes (2 / 4 / 8: the reference's N = i16 / i32 / i64) each; is_signed: SignedEncoding (zigzag), else UnsignedEncoding */
int orcgpu_encode_rle2(orcgpu_ctx* ctx, const void* values, uint64_t n, int int_bytes, int is_signed, uint32_t flags, uint8_t* out, uint64_t out_cap,
                       uint64_t* out_len);
/* = orcgpu_encode_rle2(ctx, values, n, 8, is_signed, 0, ...) */
int orcgpu_encode_rle2_i64(orcgpu_ctx* ctx, const int64_t* values, uint64_t n, int is_signed, uint8_t* out, uint64_t out_cap, uint64_t* out_len);
/* ByteRleEncoder over n bytes (Int8 columns) */
int orcgpu_encode_byte_rle(orcgpu_ctx* ctx, const void* values, uint64_t n, uint32_t flags, uint8_t* out, uint64_t out_cap, uint64_t* out_len);
/* BooleanEncoder over an Arrow bitmap of n_bits bits (least significant bit first): PRESENT streams, Boolean DATA */
int orcgpu_encode_boolean(orcgpu_ctx* ctx, const void* bits, uint64_t n_bits, uint32_t flags, uint8_t* out, uint64_t out_cap, uint64_t* out_len);

/* ColumnStripeEncoder::{encode_array, finish} for one Arrow array = one column of a stripe (src/writer/column.rs:103-165 primitive,
 * :196-258 Boolean, :304-391 strings / binaries; the types of src/writer/stripe.rs:175-185): only the valid rows' values reach the
 * encoders, a PRESENT stream is written when the array has a validity bitmap (array.nulls() is Some, whatever it holds).  The
 * streams come back in the order finish() returns them -- DATA, [LENGTH,] [PRESENT] -- in device memory of the context (valid until
 * its next encode call; orcgpu_encode_fetch copies one to the host). */
typedef struct orcgpu_enc_column {
  int32_t arrow_type;      /* ORCGPU_ARROW_BOOLEAN, _INT8 .. _INT64, _FLOAT32, _FLOAT64, _UTF8, _BINARY, _LARGE_UTF8, _LARGE_BINARY */
  uint32_t flags;          /* ORCGPU_ENC_ON_DEVICE: the three buffers are device memory */
  uint64_t n_rows;
  const uint8_t* validity; /* the validity bitmap, or NULL */
  const void* values;      /* fixed-width values; Boolean: the value bitmap; strings: value_data */
  const void* offsets;     /* strings: n_rows + 1 offsets, i32 (i64 for the LARGE_ types) */
} orcgpu_enc_column;
typedef struct orcgpu_enc_stream {
  int32_t kind;            /* ORC stream kind: 0 PRESENT, 1 DATA, 2 LENGTH */
  uint32_t pad;
  const uint8_t* data;     /* device memory */
  uint64_t len;
} orcgpu_enc_stream;
int orcgpu_encode_column(orcgpu_ctx* ctx, const orcgpu_enc_column* col, orcgpu_enc_stream streams[3], uint32_t* n_streams);
int orcgpu_encode_fetch(orcgpu_ctx* ctx, const orcgpu_enc_stream* stream, uint8_t* out);

/* ---- timing hooks used by bench.py (HIP events on the context's own stream) ---------------------- */
/* Milliseconds the device spent in the last orcgpu_decode_staged call, whole call and the RLE
 * expansion kernels alone (the dominant kernel), measured with hipEvents on the ctx stream. */
int orcgpu_last_timing(const orcgpu_ctx* ctx, float* total_ms, float* expand_ms, uint32_t* expand_launches);
/* The same call split into the phases of the pipeline (HIP events between them, same stream):
 *   0 block decompression (compression.rs:142-195): the block decoders' kernels
 *   1 the chunks' plain bytes strung together (Decompressor::read across chunks), run-boundary walk + output position scans
 *   2 PRESENT streams -> validity / ranks               3 RLE expansion (the three *_expand kernels)
 *   4 finishers (null spacing, strings, decimals, timestamps) + the summary copy
 *   5 (part of 0) the first stage of the block decompressors alone: Zstandard entropy decoding / Snappy and LZ4 token parsing;
 *     0 minus 5 = the LZ77 execution kernels (and DEFLATE, which is one kernel)
 *   6 (part of 5) Zstandard at table scale, where the sequences are decoded one lane per block: what runs in front of that
 *     kernel (the FSE table construction); 5 minus 6 = the sequences kernel (the literals kernel runs beside it).  0 otherwise
 * ms[0..n) receives the first n of them. */
#define ORCGPU_N_PHASES 7
int orcgpu_last_phase_ms(const orcgpu_ctx* ctx, float* ms, uint32_t n);

/* A decode call may run its columns as up to four COLUMN LANES side by side (each lane: a HIP stream of its own with the whole
 * pipeline over its share of the columns; the planner decides, ORCGPU_LANES = 1..4 forces a count -- read at every call).
 * Every lane launches its own kernels over its own bytes: a kernel's roofline figure must be priced launch by launch, with the
 * bytes THAT launch works for.  orcgpu_last_lane_stats reports lane `lane` of the last orcgpu_decode_staged call on `ctx`:
 *   n_lanes        lanes that call ran (the same in every lane's record)
 *   stream_bytes   staged (compressed) stream bytes of the columns the lane decoded      } SURVEY 8(d)'s algorithmic bytes
 *   arrow_bytes    Arrow bytes it left in the results                                     } of the lane's launches
 *   start_ms       host milliseconds from the call's entry to the lane's first launch (planning, sorting, table set-up)
 *   total_ms       first launch to the end of the lane's last kernel (HIP events on the lane's stream)
 *   phase_ms[]     the phases of orcgpu_last_phase_ms, for this lane
 *   seq_kernel_ms  Zstandard at table scale: zstd_seq_quads_kernel alone (an event in front of it and one behind it, in front of
 *                  the wait for the literals kernel that runs beside it); 0 when the call had no such launch
 *   exec_kernel_ms the LZ77 execution kernel(s) of the lane (lz_exec_wave_kernel / lz_exec_kernel / lz_exec_tokens_kernel)
 *   walk_short_kernel_ms, dict_emit_kernel_ms   rle_walk_short_kernel / dict_emit_kernel alone (events around the launch; 0: none)
 *   literals_kernel_ms   Zstandard at table scale: zstd_literals_kernel alone, on the stream it runs on beside the sequences kernel
 *                  (an event in front of it, behind the wait for the FSE tables, and one behind it); 0 when the call had no such launch
 * Returns ORCGPU_INVALID_ARGUMENT for a lane the last call did not run. */
typedef struct orcgpu_lane_stats {
  uint32_t lane, n_lanes;
  uint64_t stream_bytes, arrow_bytes;
  float start_ms, total_ms;
  float phase_ms[ORCGPU_N_PHASES];
  float seq_kernel_ms, exec_kernel_ms;
  float walk_short_kernel_ms, dict_emit_kernel_ms;
  float literals_kernel_ms;
} orcgpu_lane_stats;
int orcgpu_last_lane_stats(const orcgpu_ctx* ctx, uint32_t lane, orcgpu_lane_stats* out);

#ifdef __cplusplus
}
#endif
#endif
