"""ArrowReaderBuilder / ArrowReader: the reference's reader API (src/arrow_reader.rs:39-347) over the
C++ host layer inside liborcgpu.so.  Pure plumbing: every call forwards to the C ABI
(`orcgpu_reader_*`), batches arrive through the Arrow C Data Interface.

    reader = ArrowReaderBuilder.try_new("file.orc").with_batch_size(8192).with_projection(["a", "b"]).build()
    for batch in reader: ...            # pyarrow.RecordBatch, like `impl Iterator<Item = Result<RecordBatch>>`
"""
import ctypes as C

from . import capi

DEFAULT_BATCH_SIZE = 8192  # arrow_reader.rs:37


class ArrowReaderBuilder:
    def __init__(self, ctx, handle, keep=None):
        self._ctx, self._h, self._keep = ctx, handle, keep

    @classmethod
    def try_new(cls, source, ctx=None):
        """source: a path (`File`) or a bytes object (`Bytes`), the two ChunkReader impls of reader/mod.rs:48-76."""
        ctx = ctx or capi.Context(0)
        out = C.c_void_p()
        if isinstance(source, (bytes, bytearray)):
            data = bytes(source)
            ctx._check(ctx.L.orcgpu_reader_open_bytes(ctx.h, data, len(data), C.byref(out)))
            return cls(ctx, out.value, data)
        ctx._check(ctx.L.orcgpu_reader_open_file(ctx.h, str(source).encode(), C.byref(out)))
        return cls(ctx, out.value)

    def with_batch_size(self, n):
        self._ctx._check(self._ctx.L.orcgpu_reader_set_batch_size(self._h, n))
        return self

    def with_projection(self, root_names):
        """ProjectionMask::named_roots (projection.rs:52-69)."""
        arr = (C.c_char_p * len(root_names))(*[n.encode() for n in root_names])
        self._ctx._check(self._ctx.L.orcgpu_reader_set_projection(self._h, arr, len(root_names)))
        return self

    def with_projection_roots(self, indices):
        """ProjectionMask::roots (projection.rs:37): root columns by index."""
        arr = (C.c_uint32 * max(1, len(indices)))(*indices)
        self._ctx._check(self._ctx.L.orcgpu_reader_set_projection_roots(self._h, arr, len(indices)))
        return self

    def with_schema(self, schema):
        """with_schema (arrow_reader.rs:80): a pyarrow.Schema, handed over through the Arrow C Data Interface."""
        buf = (C.c_uint8 * 72)()   # struct ArrowSchema
        schema._export_to_c(C.addressof(buf))
        try:
            self._ctx._check(self._ctx.L.orcgpu_reader_set_schema(self._h, C.addressof(buf)))
        finally:
            release = C.cast(C.addressof(buf) + 56, C.POINTER(C.CFUNCTYPE(None, C.c_void_p)))[0]  # ArrowSchema::release
            if release:
                release(C.addressof(buf))
        return self

    def with_shard(self, rank, world, mode="stripes"):
        """This reader as number `rank` of `world` readers of the same file, one per GPU (orcgpu_reader_set_shard):
        mode "stripes": stripe k belongs to rank k % world; "columns": the projected root columns dealt out by estimated Arrow bytes."""
        self._ctx._check(self._ctx.L.orcgpu_reader_set_shard(self._h, rank, world, {"stripes": 0, "columns": 1}[mode]))
        return self

    def with_file_byte_range(self, start, end):
        self._ctx._check(self._ctx.L.orcgpu_reader_set_byte_range(self._h, start, end))
        return self

    def with_timestamp_precision(self, unit):
        """unit: 'us' or 'ns' (TimestampPrecision, schema.rs:31-38); 's'/'ms' as with_schema overrides."""
        self._ctx._check(self._ctx.L.orcgpu_reader_set_timestamp_precision(self._h, {"s": 1, "ms": 2, "us": 3, "ns": 4}[unit]))
        return self

    def with_row_selection(self, selectors):
        """RowSelection over the file's rows (arrow_reader.rs:113): selectors = [(row_count, skip)], i.e.
        RowSelector::select(n) = (n, False), RowSelector::skip(n) = (n, True)."""
        self._ctx._check(self._ctx.L.orcgpu_reader_set_row_selection(self._h, capi.selector_array(selectors), len(selectors)))
        return self

    def with_predicate(self, predicate):
        """with_predicate (arrow_reader.rs:173): a orc_rust_amd.predicate.Predicate evaluated against every stripe's row-group
        statistics and Bloom filters; only the row groups it keeps are read."""
        nodes, keep = predicate.flatten()
        self._ctx._check(self._ctx.L.orcgpu_reader_set_predicate(self._h, nodes, len(nodes)))
        return self

    def with_row_group_pruning(self, on):
        """Under a row selection, read only the row groups that hold selected rows (orcgpu_reader_set_row_group_pruning;
        default on).  The batches are the same either way."""
        self._ctx._check(self._ctx.L.orcgpu_reader_set_row_group_pruning(self._h, 1 if on else 0))
        return self

    def with_prefetch(self, stripes):
        """Read-ahead of the reader (orcgpu_reader_set_prefetch): decoded stripes it may be ahead of the consumer; 0 = none.
        The counterpart of choosing ArrowStreamReader (async_arrow_reader.rs) over ArrowReader: the batches are the same."""
        self._ctx._check(self._ctx.L.orcgpu_reader_set_prefetch(self._h, stripes))
        return self

    def total_row_count(self):
        return self._ctx.L.orcgpu_reader_total_rows(self._h)

    def stripe_count(self):
        return self._ctx.L.orcgpu_reader_stripe_count(self._h)

    def build(self):
        r = ArrowReader(self._ctx, self._h, self._keep)
        self._h = None
        return r

    def __del__(self):
        if getattr(self, "_h", None):
            self._ctx.L.orcgpu_reader_close(self._h)


class ArrowReader:
    def __init__(self, ctx, handle, keep=None):
        self._ctx, self._h, self._keep = ctx, handle, keep

    def total_row_count(self):
        return self._ctx.L.orcgpu_reader_total_rows(self._h)

    def row_groups(self):
        """(row groups read so far, row groups of the stripes gone through so far): orcgpu_reader_row_groups."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ctx._check(self._ctx.L.orcgpu_reader_row_groups(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def column_names(self):
        n = self._ctx.L.orcgpu_reader_column_count(self._h)
        return [self._ctx.L.orcgpu_reader_column_name(self._h, i).decode() for i in range(n)]

    def __iter__(self):
        return self

    def __next__(self):
        import pyarrow as pa
        a = (C.c_uint8 * 80)()
        s = (C.c_uint8 * 72)()
        rc = self._ctx.L.orcgpu_reader_next_batch(self._h, C.addressof(a), C.addressof(s))
        if rc == 110:  # ORCGPU_END_OF_FILE
            raise StopIteration
        self._ctx._check(rc)
        return pa.RecordBatch._import_from_c(C.addressof(a), C.addressof(s))

    def close(self):
        if self._h:
            self._ctx.L.orcgpu_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
