"""Synthetic ORC stream generator (ctypes binding of gen/orcgen.c, built with gcc in-tree).

Host-side test/benchmark tooling: produces encoded RLE v2 / v1 / byte-RLE / boolean / varint
streams and ORC-framed Snappy / LZ4 chunks for the configs of BASELINE.md.  Not on the decode
path."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "orcgen.c"), os.path.join(HERE, "tpchgen.c")]
SO = os.path.join(HERE, "liborcgen.so")
_lib = None


def _fresh():
    return os.path.exists(SO) and all(os.path.getmtime(s) <= os.path.getmtime(SO) for s in SRCS)


def build(force=False):
    """Compiles liborcgen.so in-tree.  Safe with several ranks starting at once: one builds under a file
    lock into a temporary name that replaces the library atomically (same scheme as orc_rust_amd/build.py)."""
    import fcntl
    if not force and _fresh():
        return SO
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not _fresh():
                tmp = SO + ".tmp.%d" % os.getpid()
                subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-o", tmp] + SRCS)
                os.replace(tmp, SO)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        pp, ps = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
        L.orcgen_rle2.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, pp, ps, C.c_void_p]
        L.orcgen_rle2_segments.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, pp, ps, C.c_void_p]
        L.orcgen_rle2_indexed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_void_p, pp, ps]
        L.orcgen_rle2_marked.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, pp, ps]
        L.orcgen_byte_rle_positions.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orcgen_rle1.argtypes = [C.c_void_p, C.c_size_t, C.c_int, pp, ps]
        L.orcgen_byte_rle.argtypes = [C.c_void_p, C.c_size_t, pp, ps]
        L.orcgen_bool.argtypes = [C.c_void_p, C.c_size_t, pp, ps]
        L.orcgen_varint128.argtypes = [C.c_void_p, C.c_size_t, pp, ps]
        L.orcgen_compress_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, pp, ps]
        L.orcgen_free.argtypes = [C.c_void_p]
        L.orcgen_splitmix64.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
        L.orcgen_varint64.argtypes = [C.c_void_p, C.c_size_t, pp, ps]
        L.orcgen_lineitem_segment.restype = C.c_uint64
        L.orcgen_lineitem_segment.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        L.orcgen_lineitem_warm.argtypes = [C.c_uint64]
        _lib = L
    return _lib


def _take(out, n):
    buf = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[:n.value].copy()
    lib().orcgen_free(out)
    return buf


def rle2(values, signed=True, aligned=True, stats=False):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out, n = C.c_void_p(), C.c_size_t()
    st = np.zeros(4, dtype=np.uint64)
    lib().orcgen_rle2(v.ctypes.data, v.size, int(signed), int(aligned), C.byref(out), C.byref(n), st.ctypes.data)
    buf = _take(out, n)
    return (buf, dict(zip(("short_repeat", "direct", "patched_base", "delta"), st.tolist()))) if stats else buf


def rle2_segments(values, seg_lens, signed=True, aligned=False, stats=False):
    """RLE v2 with the encoder flushed behind every segment of seg_lens[k] values (forced run boundaries)."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    sl = np.ascontiguousarray(seg_lens, dtype=np.uint32)
    out, n = C.c_void_p(), C.c_size_t()
    st = np.zeros(4, dtype=np.uint64)
    lib().orcgen_rle2_segments(v.ctypes.data, v.size, sl.ctypes.data, sl.size, int(signed), int(aligned), C.byref(out), C.byref(n), st.ctypes.data)
    buf = _take(out, n)
    return (buf, dict(zip(("short_repeat", "direct", "patched_base", "delta"), st.tolist()))) if stats else buf


def rle2_indexed(values, stride, seg_lens=None, signed=True, aligned=False):
    """RLE v2 (flushed behind every segment when seg_lens is given) and, as a writer records them for its ROW_INDEX stream,
    the positions of every stride-th value: an array [groups, 2] of (bytes written, values the encoder held) -- the stream's
    share of the RowIndexEntry of an uncompressed file (row_index.rs:42-50)."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    sl = np.ascontiguousarray(seg_lens if seg_lens is not None else [], dtype=np.uint32)
    pos = np.zeros(((v.size + stride - 1) // stride, 2), dtype=np.uint64)
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_rle2_indexed(v.ctypes.data, v.size, sl.ctypes.data, sl.size, int(signed), int(aligned), stride, pos.ctypes.data, C.byref(out), C.byref(n))
    return _take(out, n), pos


def rle2_marked(values, marks, signed=True, aligned=False):
    """RLE v2 and the positions [marks, 2] = (bytes written, values the encoder held) of the values `marks` (ascending indices): a
    column with nulls, whose row groups start at the count of non-null rows before them."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    mk = np.ascontiguousarray(marks, dtype=np.uint64)
    pos = np.zeros((mk.size, 2), dtype=np.uint64)
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_rle2_marked(v.ctypes.data, v.size, None, 0, int(signed), int(aligned), 0, mk.ctypes.data, mk.size, pos.ctypes.data, C.byref(out), C.byref(n))
    return _take(out, n), pos


def byte_rle_positions(stream, marks):
    """Positions [marks, 2] = (offset of the group's header, values of the group in front) of the values `marks` of a byte-RLE stream."""
    s = np.frombuffer(bytes(stream), dtype=np.uint8)
    mk = np.ascontiguousarray(marks, dtype=np.uint64)
    pos = np.zeros((mk.size, 2), dtype=np.uint64)
    lib().orcgen_byte_rle_positions(s.ctypes.data, s.size, mk.ctypes.data, mk.size, pos.ctypes.data)
    return pos


def rle1(values, signed=True):
    v = np.ascontiguousarray(values, dtype=np.int64)
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_rle1(v.ctypes.data, v.size, int(signed), C.byref(out), C.byref(n))
    return _take(out, n)


def byte_rle(values):
    v = np.ascontiguousarray(values).view(np.uint8)
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_byte_rle(v.ctypes.data, v.size, C.byref(out), C.byref(n))
    return _take(out, n)


def boolean(values):
    v = np.ascontiguousarray(values, dtype=np.uint8)
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_bool(v.ctypes.data, v.size, C.byref(out), C.byref(n))
    return _take(out, n)


def varint128(values):
    """values: iterable of Python ints (signed, |v| < 2**127)."""
    arr = np.zeros(2 * len(values), dtype=np.uint64)
    for i, x in enumerate(values):
        u = x & ((1 << 128) - 1)
        arr[2 * i] = u & ((1 << 64) - 1)
        arr[2 * i + 1] = u >> 64
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_varint128(arr.ctypes.data, len(values), C.byref(out), C.byref(n))
    return _take(out, n)


def varint64(values):
    """zigzag varints of an int64 array (Decimal DATA of precision <= 18)."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    out, n = C.c_void_p(), C.c_size_t()
    lib().orcgen_varint64(v.ctypes.data, v.size, C.byref(out), C.byref(n))
    return _take(out, n)


def compress_stream(data, kind, block_size=262144):
    """ORC chunk framing around greedy Snappy ('snappy') / LZ4 ('lz4') blocks, zlib raw deflate
    ('zlib', Python's zlib) or Zstandard frames ('zstd', pyarrow.Codec when available)."""
    d = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8))
    if kind in ("snappy", "lz4"):
        out, n = C.c_void_p(), C.c_size_t()
        lib().orcgen_compress_stream(d.ctypes.data, d.size, 2 if kind == "snappy" else 4, block_size, C.byref(out), C.byref(n))
        return _take(out, n)
    raw = d.tobytes()
    parts = []
    for p in range(0, len(raw), block_size):
        blk = raw[p:p + block_size]
        if kind == "zlib":
            import zlib
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(blk) + c.flush()
        elif kind == "zstd":
            import pyarrow as pa
            comp = pa.Codec("zstd", compression_level=3).compress(blk, asbytes=True)
        else:
            raise ValueError(kind)
        if len(comp) < len(blk):
            h = len(comp) << 1
            parts.append(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + comp)
        else:
            h = (len(blk) << 1) | 1
            parts.append(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + blk)
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def splitmix64(seed, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().orcgen_splitmix64(seed, out.ctypes.data, n)
    return out
