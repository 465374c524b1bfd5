"""ctypes binding of the CPU ORACLE (oracle/liborc_oracle.so).  TEST INFRASTRUCTURE ONLY.

The oracle restates orc-rust's hot path in plain C (see oracle/orc_oracle.h); the product
package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ORC_ORACLE_SO: another build of the same sources -- oracle/liborc_oracle_asan.so, the AddressSanitizer / UBSan build that
# tests/test_host_sanitized.py runs the known-answer and codec tests against (LD_PRELOAD=libasan.so)
_SO = os.environ.get("ORC_ORACLE_SO") or os.path.join(ROOT, "oracle", "liborc_oracle.so")

OK, IO_ERROR, OUT_OF_SPEC, VARINT_TOO_LARGE, DECODE_TIMESTAMP, OFFSET_OVERFLOW = 0, 1, 2, 3, 4, 5
MISMATCHED_SCHEMA, UNSUPPORTED, ARROW, BUILD_DECODER, UNEXPECTED = 6, 7, 8, 9, 10
COMP = {"none": 0, "zlib": 1, "snappy": 2, "lzo": 3, "lz4": 4, "zstd": 5}


def build():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("oo_codecs.c", "oo_encoding.c", "oo_column.c", "oo_encode.c", "orc_oracle.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.basename(_SO)])
    return _SO


class Stream(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ptr", C.c_void_p), ("len", C.c_uint64)]


class ColumnDesc(C.Structure):
    _fields_ = [
        ("orc_type", C.c_int32), ("encoding", C.c_int32), ("dictionary_size", C.c_uint32),
        ("precision", C.c_uint32), ("scale", C.c_uint32), ("ts_unit", C.c_int32), ("ts_base", C.c_int64),
        ("compression", C.c_int32), ("block_size", C.c_uint64), ("n_streams", C.c_uint32),
        ("streams", C.POINTER(Stream)),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("length", C.c_uint64), ("null_count", C.c_uint64), ("validity", C.c_void_p),
        ("values", C.c_void_p), ("values_len", C.c_uint64), ("offsets", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        for name in ("oo_inflate_raw", "oo_snappy_raw", "oo_lz4_block", "oo_zstd_frame", "oo_lzo1x"):
            f = getattr(L, name)
            f.restype = C.c_long
            f.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oo_reader_new.restype = C.c_void_p
        L.oo_reader_new.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_size_t]
        L.oo_reader_free.argtypes = [C.c_void_p]
        L.oo_int_rle_new.restype = C.c_void_p
        L.oo_int_rle_new.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oo_int_rle_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oo_int_rle_free.argtypes = [C.c_void_p]
        L.oo_byte_rle_new.restype = C.c_void_p
        L.oo_byte_rle_new.argtypes = [C.c_void_p]
        L.oo_byte_rle_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oo_byte_rle_free.argtypes = [C.c_void_p]
        L.oo_bool_new.restype = C.c_void_p
        L.oo_bool_new.argtypes = [C.c_void_p]
        L.oo_bool_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oo_bool_free.argtypes = [C.c_void_p]
        L.oo_varint128_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oo_read_varint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.oo_decode_timestamp.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
        L.oo_fix_i128_scale.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
        L.oo_decode_chunk_header.restype = C.c_uint32
        L.oo_decode_chunk_header.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.oo_stream_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.oo_free.argtypes = [C.c_void_p]
        L.oo_column_new.restype = C.c_void_p
        L.oo_column_new.argtypes = [C.POINTER(ColumnDesc), C.POINTER(C.c_int)]
        L.oo_column_next_batch.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Batch)]
        L.oo_column_next_batch_under.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(Batch)]
        L.oo_column_free.argtypes = [C.c_void_p]
        L.oo_timestamps_to_utc.restype = C.c_uint64
        L.oo_timestamps_to_utc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int64, C.c_void_p]
        L.oo_timestamp_decimals_to_utc.restype = None
        L.oo_timestamp_decimals_to_utc.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int64]
        L.oo_enc_rle2.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
        L.oo_enc_rle2_variable_run.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.oo_enc_byte_rle.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.oo_enc_boolean.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.oo_enc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _enc_result(st, out, n):
    assert st == 0, st
    res = C.string_at(out, n.value) if n.value else b""
    lib().oo_enc_free(out)
    return res


def enc_rle2(values, int_bytes=8, signed=True, with_stats=False):
    """The reference's RleV2Encoder<N, S> over `values` (N = int_bytes wide): its bytes [, runs per sub-encoding + reference panics]."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    out, n = C.c_void_p(), C.c_uint64()
    stats = np.zeros(5, dtype=np.uint64)
    st = lib().oo_enc_rle2(v.ctypes.data, len(v), int_bytes, int(signed), C.byref(out), C.byref(n), stats.ctypes.data)
    res = _enc_result(st, out, n)
    return (res, stats) if with_stats else res


def enc_rle2_variable_run(literals, int_bytes=8, signed=True):
    v = np.ascontiguousarray(literals, dtype=np.int64)
    out, n = C.c_void_p(), C.c_uint64()
    st = lib().oo_enc_rle2_variable_run(v.ctypes.data, len(v), int_bytes, int(signed), C.byref(out), C.byref(n))
    return _enc_result(st, out, n)


def enc_byte_rle(values):
    v = np.ascontiguousarray(values).view(np.uint8)
    out, n = C.c_void_p(), C.c_uint64()
    st = lib().oo_enc_byte_rle(v.ctypes.data, len(v), C.byref(out), C.byref(n))
    return _enc_result(st, out, n)


def enc_boolean(bits_lsb, n_bits):
    """bits_lsb: an Arrow bitmap (numpy uint8, least significant bit first) of n_bits bits"""
    v = np.ascontiguousarray(bits_lsb, dtype=np.uint8)
    assert len(v) * 8 >= n_bits
    out, n = C.c_void_p(), C.c_uint64()
    st = lib().oo_enc_boolean(v.ctypes.data, n_bits, C.byref(out), C.byref(n))
    return _enc_result(st, out, n)


def codec(name, data, cap):
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = getattr(lib(), {"zlib": "oo_inflate_raw", "snappy": "oo_snappy_raw", "lz4": "oo_lz4_block", "zstd": "oo_zstd_frame", "lzo": "oo_lzo1x"}[name])(
        bytes(data), len(data), out.ctypes.data, cap)
    if n < 0:
        return None
    return out[:n].tobytes()


def stream_decompress(data, compression="none", block_size=262144):
    out = C.c_void_p()
    n = C.c_size_t()
    st = lib().oo_stream_decompress(bytes(data), len(data), COMP[compression], block_size, C.byref(out), C.byref(n))
    res = C.string_at(out, n.value)
    lib().oo_free(out)
    return st, res


def int_rle(data, n, version=2, signed=True, nbits=64, compression="none", block_size=262144, chunks=None):
    """Decode n values; `chunks` optionally splits the request into several decode() calls.
    Returns (status, values decoded by the successful calls)."""
    L = lib()
    data = bytes(data)
    r = L.oo_reader_new(data, len(data), COMP[compression], block_size)
    d = L.oo_int_rle_new(r, version, int(signed), nbits)
    out = []
    st = 0
    for c in (chunks or [n]):
        buf = np.zeros(max(c, 1), dtype=np.int64)
        st = L.oo_int_rle_decode(d, buf.ctypes.data, c)
        if st:
            break
        out.append(buf[:c].copy())
    L.oo_int_rle_free(d)
    L.oo_reader_free(r)
    return st, (np.concatenate(out) if out else np.zeros(0, dtype=np.int64))


def byte_rle(data, n, compression="none", block_size=262144):
    L = lib()
    data = bytes(data)
    r = L.oo_reader_new(data, len(data), COMP[compression], block_size)
    d = L.oo_byte_rle_new(r)
    buf = np.zeros(max(n, 1), dtype=np.int8)
    st = L.oo_byte_rle_decode(d, buf.ctypes.data, n)
    L.oo_byte_rle_free(d)
    L.oo_reader_free(r)
    return st, buf[:n]


def boolean(data, n, compression="none", block_size=262144):
    L = lib()
    data = bytes(data)
    r = L.oo_reader_new(data, len(data), COMP[compression], block_size)
    d = L.oo_bool_new(r)
    buf = np.zeros(max(n, 1), dtype=np.uint8)
    st = L.oo_bool_decode(d, buf.ctypes.data, n)
    L.oo_bool_free(d)
    L.oo_reader_free(r)
    return st, buf[:n]


def varint(data, nbits=64, signed=False):
    L = lib()
    data = bytes(data)
    r = L.oo_reader_new(data, len(data), 0, 0)
    v = C.c_int64()
    st = L.oo_read_varint(r, nbits, int(signed), C.byref(v))
    L.oo_reader_free(r)
    return st, v.value


def varint128(data, n):
    L = lib()
    data = bytes(data)
    r = L.oo_reader_new(data, len(data), 0, 0)
    buf = np.zeros(2 * max(n, 1), dtype=np.uint64)
    st = L.oo_varint128_decode(r, buf.ctypes.data, n)
    L.oo_reader_free(r)
    vals = []
    for i in range(n):
        v = int(buf[2 * i]) | (int(buf[2 * i + 1]) << 64)
        if v >= 1 << 127:
            v -= 1 << 128
        vals.append(v)
    return st, vals


def decode_timestamp(base, seconds, nanos, unit=3):
    v = C.c_int64()
    st = lib().oo_decode_timestamp(base, seconds, nanos, unit, C.byref(v))
    return st, v.value


def fix_scale(value, fixed_scale, varying_scale):
    inp = np.array([value & ((1 << 64) - 1), (value >> 64) & ((1 << 64) - 1)], dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    lib().oo_fix_i128_scale(inp.ctypes.data, fixed_scale, varying_scale, out.ctypes.data)
    v = int(out[0]) | (int(out[1]) << 64)
    return v - (1 << 128) if v >= 1 << 127 else v


_WIDTH = {1: 1, 2: 2, 3: 4, 15: 4, 5: 4, 4: 8, 6: 8, 9: 8, 18: 8, 14: 16}


def timestamps_to_utc(res, n, unit, tz):
    """TimestampOffsetArrayDecoder over one decoded batch `res` (oo_timestamps_to_utc)."""
    at = np.ascontiguousarray(tz[0], dtype=np.int64)
    offs = np.ascontiguousarray(tz[1], dtype=np.int32)
    vals = np.frombuffer(res["values"], dtype=np.int64).copy()
    vout = np.zeros((n + 7) // 8, dtype=np.uint8)
    vin = res["validity"]
    nulls = lib().oo_timestamps_to_utc(vals.ctypes.data, vin, n, unit, at.ctypes.data, offs.ctypes.data, len(at), int(tz[2]), int(tz[3]), vout.ctypes.data)
    out = dict(res)
    out["values"] = vals.tobytes()
    out["null_count"] = nulls
    out["validity"] = vout.tobytes() if nulls else None
    return out


class Column:
    """Batch-by-batch oracle column decoder (oo_column_*).  `streams` maps Stream.Kind -> bytes."""

    def __init__(self, orc_type, encoding, streams, dictionary_size=0, precision=0, scale=0, ts_unit=3,
                 ts_base=1420070400, compression="none", block_size=262144, tz=None):
        """tz: (at int64[], offs int32[], offs0, fold_at) -- the writer's zone (tests/tz_table.py); TIMESTAMP batches are then re-labelled to UTC."""
        L = lib()
        self.tz = tz if orc_type == 9 else None
        self.ts_unit = ts_unit
        self._keep = [bytes(v) for v in streams.values()]
        arr = (Stream * max(len(streams), 1))()
        for i, (k, v) in enumerate(zip(streams.keys(), self._keep)):
            arr[i].kind = k
            arr[i].ptr = C.cast(C.c_char_p(v), C.c_void_p)
            arr[i].len = len(v)
        self._arr = arr
        d = ColumnDesc(orc_type, encoding, dictionary_size, precision, scale, ts_unit, ts_base, COMP[compression],
                       block_size, len(streams), arr)
        st = C.c_int()
        self.orc_type = orc_type
        self.h = L.oo_column_new(C.byref(d), C.byref(st))
        self.status = st.value

    def next_batch(self, n, parent_present=None):
        """parent_present: one byte per row (numpy uint8 / bool; 0: the Struct or Union arm above is null there), None: no parent."""
        b = Batch()
        if parent_present is None:
            lib().oo_column_next_batch(self.h, n, C.byref(b))
        else:
            pp = np.ascontiguousarray(parent_present, dtype=np.uint8)
            assert pp.size == n
            lib().oo_column_next_batch_under(self.h, n, pp.ctypes.data, C.byref(b))
        res = {"status": b.status, "length": b.length, "null_count": b.null_count, "validity": None, "values": None, "offsets": None}
        if b.status:
            return res
        if b.validity:
            res["validity"] = C.string_at(b.validity, (n + 7) // 8)
        if b.values or b.values_len == 0:
            res["values"] = C.string_at(b.values, b.values_len) if b.values_len else b""
        if b.offsets:
            res["offsets"] = np.frombuffer(C.string_at(b.offsets, 4 * (n + 1)), dtype=np.int32).copy()
        if self.tz is not None and n:
            res = timestamps_to_utc(res, n, self.ts_unit, self.tz)
        return res

    def close(self):
        if self.h:
            lib().oo_column_free(self.h)
            self.h = None

    def __del__(self):
        self.close()
