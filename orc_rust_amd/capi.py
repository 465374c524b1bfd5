"""ctypes binding of the C ABI in include/orcgpu.h (liborcgpu.so).

This is plumbing only: every function here forwards to the HIP library.  There is no CPU
fallback -- if the library is missing, or no MI355X is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK = 0
ERR_NAMES = {
    0: "Ok", 1: "IoError", 2: "OutOfSpec", 3: "VarintTooLarge", 4: "DecodeTimestamp", 5: "OffsetOverflow", 6: "MismatchedSchema",
    7: "UnsupportedTypeVariant", 8: "Arrow", 9: "BuildDecoder", 10: "Unexpected", 100: "HipError", 101: "InvalidArgument",
}
COMP = {"none": 0, "zlib": 1, "snappy": 2, "lzo": 3, "lz4": 4, "zstd": 5}
EXPORTS = [
    "orcgpu_open", "orcgpu_close", "orcgpu_last_error", "orcgpu_version", "orcgpu_abi_version", "orcgpu_stage_stripe", "orcgpu_staged_free",
    "orcgpu_staged_bytes", "orcgpu_decode_staged", "orcgpu_stripe_decode", "orcgpu_result_free", "orcgpu_result_status",
    "orcgpu_result_rows", "orcgpu_result_batches", "orcgpu_result_arrow_bytes", "orcgpu_result_batch_view",
    "orcgpu_result_copy_batch", "orcgpu_result_fetch", "orcgpu_result_fetch_async", "orcgpu_result_select", "orcgpu_selection_batches", "orcgpu_timezone_offsets", "orcgpu_result_export_batch", "orcgpu_last_timing", "orcgpu_last_phase_ms", "orcgpu_last_lane_stats", "orcgpu_encode_rle2_i64", "orcgpu_encode_rle2", "orcgpu_encode_byte_rle", "orcgpu_encode_boolean", "orcgpu_encode_column", "orcgpu_encode_fetch",
    "orcgpu_reader_open_file", "orcgpu_reader_open_bytes", "orcgpu_reader_close", "orcgpu_reader_set_batch_size",
    "orcgpu_reader_set_projection", "orcgpu_reader_set_projection_roots", "orcgpu_reader_set_schema", "orcgpu_reader_set_byte_range", "orcgpu_reader_set_shard", "orcgpu_shard_columns", "orcgpu_reader_column_weight", "orcgpu_reader_set_timestamp_precision", "orcgpu_reader_set_row_selection", "orcgpu_reader_set_prefetch",
    "orcgpu_reader_set_row_group_pruning", "orcgpu_reader_row_groups", "orcgpu_index_entry", "orcgpu_reader_set_predicate",
    "orcgpu_predicate_row_groups",
    "orcgpu_reader_total_rows", "orcgpu_reader_stripe_count", "orcgpu_reader_column_count", "orcgpu_reader_column_name",
    "orcgpu_reader_next_batch",
]


class OrcGpuError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class StreamEntry(C.Structure):
    _fields_ = [("chunk_offset", C.c_uint64), ("skip_bytes", C.c_uint32), ("skip_values", C.c_uint32), ("skip_bits", C.c_uint32)]


class Stream(C.Structure):
    _fields_ = [("column_id", C.c_uint32), ("kind", C.c_int32), ("ptr", C.c_void_p), ("len", C.c_uint64),
                ("skip_bytes", C.c_uint32), ("skip_values", C.c_uint32), ("entries", C.c_void_p), ("n_entries", C.c_uint32), ("skip_bits", C.c_uint32)]


class Column(C.Structure):
    _fields_ = [("column_id", C.c_uint32), ("orc_type", C.c_int32), ("encoding", C.c_int32), ("dictionary_size", C.c_uint32),
                ("precision", C.c_uint32), ("scale", C.c_uint32), ("arrow_target", C.c_int32), ("arrow_precision", C.c_uint32),
                ("arrow_scale", C.c_uint32), ("parent", C.c_uint32)]


class StripeDesc(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("compression", C.c_int32), ("block_size", C.c_uint64), ("ts_base_seconds", C.c_int64),
                ("batch_size", C.c_uint32), ("n_streams", C.c_uint32), ("streams", C.POINTER(Stream)), ("n_columns", C.c_uint32),
                ("columns", C.POINTER(Column)), ("writer_timezone", C.c_char_p)]


class LaneStats(C.Structure):
    _fields_ = [("lane", C.c_uint32), ("n_lanes", C.c_uint32), ("stream_bytes", C.c_uint64), ("arrow_bytes", C.c_uint64),
                ("start_ms", C.c_float), ("total_ms", C.c_float), ("phase_ms", C.c_float * 7), ("seq_kernel_ms", C.c_float),
                ("exec_kernel_ms", C.c_float), ("walk_short_kernel_ms", C.c_float), ("dict_emit_kernel_ms", C.c_float),
                ("literals_kernel_ms", C.c_float)]


class RowSelector(C.Structure):
    _fields_ = [("row_count", C.c_uint64), ("skip", C.c_int32)]


def selector_array(selectors):
    """[(row_count, skip)] -> ctypes array of orcgpu_row_selector"""
    arr = (RowSelector * max(1, len(selectors)))()
    for i, (n, skip) in enumerate(selectors):
        arr[i].row_count = n
        arr[i].skip = 1 if skip else 0
    return arr


def selection_batches(selectors, stripe_rows, batch_size=8192):
    """Host only: ([(start, len)] the selection yields on a stripe of stripe_rows rows, [(row_count, skip)] left for the next stripe)."""
    L = load()
    cap = len(selectors) + stripe_rows // max(1, batch_size) + 8
    starts, lens = (C.c_uint64 * cap)(), (C.c_uint32 * cap)()
    rest = (RowSelector * (len(selectors) + 2))()
    n_out, n_rest = C.c_uint32(), C.c_uint32()
    rc = L.orcgpu_selection_batches(selector_array(selectors), len(selectors), stripe_rows, batch_size, starts, lens, cap, C.byref(n_out), rest,
                                    len(selectors) + 2, C.byref(n_rest))
    assert rc == 0 and n_out.value <= cap
    return [(starts[i], lens[i]) for i in range(n_out.value)], [(rest[i].row_count, bool(rest[i].skip)) for i in range(n_rest.value)]


def timezone_offsets(name, instants):
    """Host only: (UTC offsets in seconds of zone `name` at the given UNIX instants, the ORC epoch in that zone)."""
    L = load()
    a = np.ascontiguousarray(instants, dtype=np.int64)
    out = np.zeros(len(a), dtype=np.int32)
    epoch = C.c_int64()
    rc = L.orcgpu_timezone_offsets(name.encode(), a.ctypes.data_as(C.c_void_p), len(a), out.ctypes.data_as(C.c_void_p), C.byref(epoch))
    if rc:
        raise OrcGpuError(rc, "time zone %r" % name)
    return out, epoch.value


class EncColumn(C.Structure):
    _fields_ = [("arrow_type", C.c_int32), ("flags", C.c_uint32), ("n_rows", C.c_uint64), ("validity", C.c_void_p), ("values", C.c_void_p),
                ("offsets", C.c_void_p)]


class EncStream(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad", C.c_uint32), ("data", C.c_void_p), ("len", C.c_uint64)]


ENC_ON_DEVICE = 1
ARROW = {"bool": 10, "int8": 11, "int16": 12, "int32": 13, "int64": 14, "float32": 15, "float64": 16, "utf8": 17, "binary": 18,
         "large_utf8": 25, "large_binary": 26}


class BatchView(C.Structure):
    _fields_ = [("length", C.c_uint64), ("null_count", C.c_uint64), ("validity", C.c_void_p), ("values", C.c_void_p),
                ("values_bytes", C.c_uint64), ("offsets", C.c_void_p)]


_lib = None
ABI_VERSION = 3  # include/orcgpu.h: ORCGPU_ABI_VERSION


def lib_path():
    return _build.SO


def load():
    """Loads liborcgpu.so (building it if the sources are newer).  Raises if it cannot be built or loaded."""
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("ORCGPU_LIB") or _build.build()
    # the host's setting to make, before the process's first HIP call (INTEGRATION.md "Runtime setting"): the decoder's streams side by side
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "TZDIR" not in os.environ and not os.path.isdir("/usr/share/zoneinfo"):
        # writer time zones are resolved by the library through the system's tz database; without one, use the tzdata package's
        try:
            import tzdata
            os.environ["TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
        except ImportError:
            pass
    L = C.CDLL(so)
    L.orcgpu_open.restype = C.c_void_p
    L.orcgpu_open.argtypes = [C.c_int, C.c_void_p]
    L.orcgpu_close.argtypes = [C.c_void_p]
    L.orcgpu_last_error.restype = C.c_char_p
    L.orcgpu_last_error.argtypes = [C.c_void_p]
    L.orcgpu_version.restype = C.c_char_p
    if L.orcgpu_abi_version() != ABI_VERSION:
        raise RuntimeError("liborcgpu.so speaks ABI %d, this binding %d (include/orcgpu.h: ORCGPU_ABI_VERSION)" % (L.orcgpu_abi_version(), ABI_VERSION))
    L.orcgpu_stage_stripe.argtypes = [C.c_void_p, C.POINTER(StripeDesc), C.POINTER(C.c_void_p)]
    L.orcgpu_staged_free.argtypes = [C.c_void_p]
    L.orcgpu_staged_bytes.restype = C.c_uint64
    L.orcgpu_staged_bytes.argtypes = [C.c_void_p]
    L.orcgpu_decode_staged.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p)]
    L.orcgpu_stripe_decode.argtypes = [C.c_void_p, C.POINTER(StripeDesc), C.POINTER(C.c_void_p)]
    L.orcgpu_result_free.argtypes = [C.c_void_p]
    L.orcgpu_result_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orcgpu_result_rows.restype = C.c_uint64
    L.orcgpu_result_rows.argtypes = [C.c_void_p]
    L.orcgpu_result_batches.restype = C.c_uint32
    L.orcgpu_result_batches.argtypes = [C.c_void_p]
    L.orcgpu_result_arrow_bytes.restype = C.c_uint64
    L.orcgpu_result_arrow_bytes.argtypes = [C.c_void_p]
    L.orcgpu_result_batch_view.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(BatchView)]
    L.orcgpu_result_copy_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orcgpu_result_fetch.argtypes = [C.c_void_p, C.c_void_p]
    L.orcgpu_result_fetch_async.argtypes = [C.c_void_p, C.c_void_p]
    L.orcgpu_reader_set_prefetch.argtypes = [C.c_void_p, C.c_uint32]
    L.orcgpu_reader_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    L.orcgpu_shard_columns.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.orcgpu_reader_column_weight.argtypes = [C.c_void_p, C.c_uint32]
    L.orcgpu_reader_column_weight.restype = C.c_double
    from .predicate import ColumnIndex, PredicateNode
    L.orcgpu_reader_set_predicate.argtypes = [C.c_void_p, C.POINTER(PredicateNode), C.c_uint32]
    L.orcgpu_predicate_row_groups.argtypes = [C.POINTER(PredicateNode), C.c_uint32, C.POINTER(ColumnIndex), C.c_uint32, C.c_uint64, C.c_uint64,
                                              C.c_void_p, C.POINTER(C.c_uint32)]
    L.orcgpu_reader_set_row_group_pruning.argtypes = [C.c_void_p, C.c_int]
    L.orcgpu_reader_row_groups.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orcgpu_index_entry.argtypes = [C.POINTER(Column), C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_uint32, C.c_int32, C.POINTER(StreamEntry)]
    L.orcgpu_result_select.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RowSelector), C.c_uint32]
    L.orcgpu_selection_batches.argtypes = [C.POINTER(RowSelector), C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                           C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(RowSelector), C.c_uint32, C.POINTER(C.c_uint32)]
    L.orcgpu_reader_set_row_selection.argtypes = [C.c_void_p, C.POINTER(RowSelector), C.c_uint32]
    L.orcgpu_result_export_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.orcgpu_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.orcgpu_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32]
    L.orcgpu_last_lane_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(LaneStats)]
    L.orcgpu_encode_rle2_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orcgpu_encode_rle2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orcgpu_encode_byte_rle.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orcgpu_encode_boolean.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orcgpu_encode_column.argtypes = [C.c_void_p, C.POINTER(EncColumn), C.POINTER(EncStream), C.POINTER(C.c_uint32)]
    L.orcgpu_encode_fetch.argtypes = [C.c_void_p, C.POINTER(EncStream), C.c_void_p]
    L.orcgpu_reader_open_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.orcgpu_reader_open_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.orcgpu_reader_close.argtypes = [C.c_void_p]
    L.orcgpu_reader_set_batch_size.argtypes = [C.c_void_p, C.c_uint32]
    L.orcgpu_reader_set_projection.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32]
    L.orcgpu_reader_set_projection_roots.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    L.orcgpu_reader_set_schema.argtypes = [C.c_void_p, C.c_void_p]
    L.orcgpu_reader_set_byte_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.orcgpu_reader_set_timestamp_precision.argtypes = [C.c_void_p, C.c_int]
    L.orcgpu_reader_total_rows.restype = C.c_uint64
    L.orcgpu_reader_total_rows.argtypes = [C.c_void_p]
    L.orcgpu_reader_stripe_count.restype = C.c_uint32
    L.orcgpu_reader_stripe_count.argtypes = [C.c_void_p]
    L.orcgpu_reader_column_count.restype = C.c_uint32
    L.orcgpu_reader_column_count.argtypes = [C.c_void_p]
    L.orcgpu_reader_column_name.restype = C.c_char_p
    L.orcgpu_reader_column_name.argtypes = [C.c_void_p, C.c_uint32]
    L.orcgpu_reader_next_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


class Context:
    """orcgpu_ctx: one GPU, one HIP stream.  Raises when no HIP device is usable."""

    def __init__(self, device=0):
        self.L = load()
        self.h = self.L.orcgpu_open(device, None)
        if not self.h:
            raise OrcGpuError(100, "orcgpu_open(%d) failed: no usable HIP device (this library has no CPU path)" % device)

    def close(self):
        if self.h:
            self.L.orcgpu_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        return self.L.orcgpu_last_error(self.h).decode("utf-8", "replace")  # (a message may quote bytes of a damaged file)

    def _check(self, rc):
        if rc:
            raise OrcGpuError(rc, self.error())

    def stage(self, n_rows, streams, columns, compression="none", block_size=262144, batch_size=8192, ts_base=0, writer_timezone=None):
        """streams: [(column_id, kind, bytes)] or [(column_id, kind, bytes, skip_bytes, skip_values)] (entry points; a sixth element: the positions of the later row groups as an array [groups, 2] = (byte, values) for an
        uncompressed or [groups, 3] = (chunk header offset, bytes into the chunk, values) for a compressed stream), columns: [dict(column_id, orc_type, encoding, dictionary_size, precision, scale, arrow_target)]"""
        keep = [bytes(s[2]) if not isinstance(s[2], (bytes, np.ndarray)) else s[2] for s in streams]
        sarr = (Stream * max(1, len(streams)))()
        for i, (s, buf) in enumerate(zip(streams, keep)):
            cid, kind = s[0], s[1]
            sarr[i].column_id = cid
            sarr[i].kind = kind
            if len(s) > 3:
                sarr[i].skip_bytes, sarr[i].skip_values = s[3], s[4]
            if len(s) > 5 and s[5] is not None and len(s[5]):
                pos = np.asarray(s[5], dtype=np.uint64)
                earr = (StreamEntry * len(pos))()
                ev = np.frombuffer(earr, dtype=np.dtype([("o", "<u8"), ("b", "<u4"), ("v", "<u4"), ("t", "<u4"), ("p", "<u4")]))
                ev["o"] = pos[:, 0]
                ev["b"] = pos[:, 1] if pos.shape[1] == 3 else 0
                ev["v"] = pos[:, -1]
                keep.append(earr)
                sarr[i].entries = C.addressof(earr)
                sarr[i].n_entries = len(pos)
            if isinstance(buf, np.ndarray):
                sarr[i].ptr = buf.ctypes.data
                sarr[i].len = buf.nbytes
            else:
                sarr[i].ptr = C.cast(C.c_char_p(buf), C.c_void_p)
                sarr[i].len = len(buf)
        carr = (Column * max(1, len(columns)))()
        for i, c in enumerate(columns):
            carr[i].column_id = c["column_id"]
            carr[i].orc_type = c["orc_type"]
            carr[i].encoding = c.get("encoding", 2)
            carr[i].dictionary_size = c.get("dictionary_size", 0)
            carr[i].precision = c.get("precision", 0)
            carr[i].scale = c.get("scale", 0)
            carr[i].arrow_target = c.get("arrow_target", 0)
            carr[i].arrow_precision = c.get("arrow_precision", 0)
            carr[i].arrow_scale = c.get("arrow_scale", 0)
            carr[i].parent = c.get("parent", 0)  # 1 + index of the Struct column this one is a field of (0: a root column)
        d = StripeDesc(n_rows, COMP[compression] if isinstance(compression, str) else compression, block_size, ts_base, batch_size,
                       len(streams), sarr, len(columns), carr, writer_timezone.encode() if writer_timezone else None)
        out = C.c_void_p()
        self._check(self.L.orcgpu_stage_stripe(self.h, C.byref(d), C.byref(out)))
        return Staged(self, out.value)

    def _encode(self, fn, *args):
        """size first, then the bytes (the two-call protocol of the orcgpu_encode_* entry points)"""
        n = C.c_uint64(0)
        self._check(fn(self.h, *args, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=np.uint8)
        m = C.c_uint64(0)
        self._check(fn(self.h, *args, out.ctypes.data, out.size, C.byref(m)))
        assert m.value == n.value
        return out[:n.value].tobytes()

    def encode_rle2(self, values, int_bytes=8, signed=True):
        """RleV2Encoder<N, S> (rle_v2/mod.rs:255-531) over host values of N = int_bytes"""
        v = np.ascontiguousarray(values, dtype={2: np.int16, 4: np.int32, 8: np.int64}[int_bytes])
        return self._encode(self.L.orcgpu_encode_rle2, v.ctypes.data, v.size, int_bytes, 1 if signed else 0, 0)

    def encode_byte_rle(self, values):
        v = np.ascontiguousarray(values).view(np.uint8)
        return self._encode(self.L.orcgpu_encode_byte_rle, v.ctypes.data, v.size, 0)

    def encode_boolean(self, bits_lsb, n_bits):
        v = np.ascontiguousarray(bits_lsb, dtype=np.uint8)
        assert v.size * 8 >= n_bits
        return self._encode(self.L.orcgpu_encode_boolean, v.ctypes.data, n_bits, 0)

    def encode_column(self, arrow_type, n_rows, values, validity=None, offsets=None):
        """ColumnStripeEncoder::{encode_array, finish} (writer/column.rs): [(ORC stream kind, bytes)] in finish()'s order"""
        keep = [np.ascontiguousarray(values)]
        col = EncColumn(ARROW[arrow_type], 0, n_rows, None, keep[0].ctypes.data if keep[0].size else None, None)
        if keep[0].size == 0:
            keep.append(np.zeros(1, dtype=np.uint8))
            col.values = keep[-1].ctypes.data
        if validity is not None:
            keep.append(np.ascontiguousarray(validity, dtype=np.uint8))
            if keep[-1].size == 0:
                keep[-1] = np.zeros(1, dtype=np.uint8)
            col.validity = keep[-1].ctypes.data
        if offsets is not None:
            keep.append(np.ascontiguousarray(offsets))
            col.offsets = keep[-1].ctypes.data
        streams = (EncStream * 3)()
        n = C.c_uint32(0)
        self._check(self.L.orcgpu_encode_column(self.h, C.byref(col), streams, C.byref(n)))
        out = []
        for i in range(n.value):
            buf = np.zeros(max(1, streams[i].len), dtype=np.uint8)
            self._check(self.L.orcgpu_encode_fetch(self.h, C.byref(streams[i]), buf.ctypes.data))
            out.append((streams[i].kind, buf[:streams[i].len].tobytes()))
        return out

    def decode(self, staged_list, results=None):
        n = len(staged_list)
        sarr = (C.c_void_p * n)(*[s.h for s in staged_list])
        rarr = (C.c_void_p * n)(*[(r.h if r is not None else None) for r in (results or [None] * n)])
        self._check(self.L.orcgpu_decode_staged(self.h, sarr, n, rarr))
        # (results[i] given: its buffers were decoded into again; None: a new result)
        return [results[i] if results and results[i] is not None else Result(self, rarr[i]) for i in range(n)]

    def timing(self):
        a, b, c = C.c_float(), C.c_float(), C.c_uint32()
        self.L.orcgpu_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


    PHASES = ("decompress", "walk", "present", "expand", "finish")  # + "decompress_stage1": the part of "decompress" spent in its first stage
    # + "decompress_tables": the part of that in front of the Zstandard sequences kernel (one lane per block; 0 otherwise)

    def phase_ms(self):
        """Device milliseconds of the last decode call per pipeline phase (orcgpu_last_phase_ms)."""
        a = (C.c_float * 7)()
        self.L.orcgpu_last_phase_ms(self.h, a, 7)
        return dict(zip(self.PHASES + ("decompress_stage1", "decompress_tables"), [float(x) for x in a]))

    def lane_stats(self):
        """Every column lane of the last decode call (orcgpu_last_lane_stats): [{lane, n_lanes, stream_bytes, arrow_bytes,
        start_ms, total_ms, phase_ms{}, seq_kernel_ms, exec_kernel_ms}] -- a lane's launches and the bytes they worked for."""
        out = []
        k = 0
        while True:
            st = LaneStats()
            if self.L.orcgpu_last_lane_stats(self.h, k, C.byref(st)) != OK:
                break
            out.append({"lane": st.lane, "n_lanes": st.n_lanes, "stream_bytes": int(st.stream_bytes), "arrow_bytes": int(st.arrow_bytes),
                        "start_ms": float(st.start_ms), "total_ms": float(st.total_ms),
                        "phase_ms": dict(zip(self.PHASES + ("decompress_stage1", "decompress_tables"), [float(x) for x in st.phase_ms])),
                        "seq_kernel_ms": float(st.seq_kernel_ms), "exec_kernel_ms": float(st.exec_kernel_ms),
                        "walk_short_kernel_ms": float(st.walk_short_kernel_ms), "dict_emit_kernel_ms": float(st.dict_emit_kernel_ms),
                        "literals_kernel_ms": float(st.literals_kernel_ms)})
            k += 1
            if k >= st.n_lanes:
                break
        return out


class Staged:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def nbytes(self):
        return self.ctx.L.orcgpu_staged_bytes(self.h)

    def free(self):
        if self.h:
            self.ctx.L.orcgpu_staged_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Result:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def free(self):
        if self.h:
            self.ctx.L.orcgpu_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def status(self):
        b, c = C.c_uint32(), C.c_uint32()
        st = self.ctx.L.orcgpu_result_status(self.h, C.byref(b), C.byref(c))
        return st, b.value, c.value

    @property
    def rows(self):
        return self.ctx.L.orcgpu_result_rows(self.h)

    @property
    def n_batches(self):
        return self.ctx.L.orcgpu_result_batches(self.h)

    @property
    def arrow_bytes(self):
        return self.ctx.L.orcgpu_result_arrow_bytes(self.h)

    def view(self, batch, column):
        v = BatchView()
        self.ctx._check(self.ctx.L.orcgpu_result_batch_view(self.h, batch, column, C.byref(v)))
        return v

    def batch(self, batch, column):
        """Host copy of one column-batch: dict(length, null_count, validity, values, offsets) -- the same
        layout the CPU oracle reports, so parity tests compare bytes."""
        v = self.view(batch, column)
        n = v.length
        values = np.zeros(max(1, v.values_bytes), dtype=np.uint8)
        offsets = np.zeros(n + 1, dtype=np.int32) if v.offsets else None
        validity = np.zeros((n + 7) // 8, dtype=np.uint8) if v.validity else None
        self.ctx._check(self.ctx.L.orcgpu_result_copy_batch(
            self.ctx.h, self.h, batch, column, values.ctypes.data, offsets.ctypes.data if offsets is not None else None,
            validity.ctypes.data if validity is not None else None))
        return {"status": 0, "length": n, "null_count": v.null_count, "validity": validity.tobytes() if validity is not None else None,
                "values": values[:v.values_bytes].tobytes(), "offsets": offsets}

    def select(self, selectors):
        """Row selection over the decoded stripe: selectors = [(row_count, skip)] (orcgpu_result_select)."""
        self.ctx._check(self.ctx.L.orcgpu_result_select(self.ctx.h, self.h, selector_array(selectors), len(selectors)))

    def fetch(self):
        """One device-to-host copy of all the result's Arrow buffers (pinned memory); export_batch then makes views."""
        self.ctx._check(self.ctx.L.orcgpu_result_fetch(self.ctx.h, self.h))

    def fetch_async(self):
        """Starts that copy and returns (orcgpu_result_fetch_async); fetch() / export_batch() wait for it."""
        self.ctx._check(self.ctx.L.orcgpu_result_fetch_async(self.ctx.h, self.h))

    def export_batch(self, batch):
        """Arrow C Data Interface export -> pyarrow.RecordBatch (zero-copy import of host buffers)."""
        import pyarrow as pa
        a = (C.c_uint8 * 80)()   # struct ArrowArray  (10 x 8 bytes)
        s = (C.c_uint8 * 72)()   # struct ArrowSchema (9 x 8 bytes)
        pa_ptr, ps_ptr = C.addressof(a), C.addressof(s)
        self.ctx._check(self.ctx.L.orcgpu_result_export_batch(self.ctx.h, self.h, batch, pa_ptr, ps_ptr))
        return pa.RecordBatch._import_from_c(pa_ptr, ps_ptr)
