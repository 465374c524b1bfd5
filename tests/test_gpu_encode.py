"""GPU ORC encode: the reference's value encoders on the device (device/rle_encode.hip) -- RleV2Encoder<N, S>
(rle_v2/mod.rs:255-531), ByteRleEncoder (byte.rs:38-197), BooleanEncoder (boolean.rs:119-170) and a column's streams as
ColumnStripeEncoder::{encode_array, finish} hands them to the stripe writer (writer/column.rs).  The bar is the reference's own
BYTES: every stream is compared with the restated reference encoder (oracle/oo_encode.c, pinned on the reference's writer vectors
by test_oracle_encode.py); what the device DECODER and the oracle's decoder make of the stream is checked as well."""
import ctypes as C

import numpy as np
import pytest

import gpu_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
LONG, DATA, DIRECT_V2 = 4, 1, 2
PRESENT, LENGTH = 0, 2


def shapes(rng, n, nbits=64):
    top = 1 << (nbits - 2)
    yield "random", rng.integers(-top, top, n)
    yield "small", rng.integers(0, 100, n)
    yield "runs", np.repeat(rng.integers(-1000, 1000, n // 5 + 1), rng.integers(1, 14, n // 5 + 1))[:n]
    yield "long_runs", np.repeat(rng.integers(-5, 5, n // 300 + 1), rng.integers(1, 1400, n // 300 + 1))[:n]
    yield "pairs_and_triples", np.repeat(rng.integers(0, 4, n // 2 + 1), rng.integers(1, 4, n // 2 + 1))[:n]
    yield "ascending", np.cumsum(rng.integers(0, 50, n))
    yield "descending", -np.cumsum(rng.integers(0, 5, n))
    yield "steps", np.arange(n) * 7 - 300
    v = rng.integers(0, 200, n)
    v[rng.integers(0, n, max(1, n // 40))] = rng.integers(1 << (nbits // 2 - 2), 1 << (nbits - 4), max(1, n // 40))
    yield "outliers", v
    v = rng.integers(0, 200, n)
    v[::300] = 1 << (nbits - 6)
    yield "sparse_outliers", v
    v = rng.integers(0, 16, n)
    v[::509] = 1 << (nbits - 9)
    yield "outliers_511_apart", v
    lim = np.iinfo({16: np.int16, 32: np.int32, 64: np.int64}[nbits])
    yield "extremes", rng.choice(np.array([lim.min, lim.max, 0, -1, 1], dtype=np.int64), n)
    yield "near_limits", rng.choice(np.array([lim.min, lim.min + 1, lim.min + 5], dtype=np.int64), n)
    yield "mixed", np.concatenate([np.repeat(7, 600), rng.integers(0, 9, 50), np.arange(40), np.repeat(-3, 2), rng.integers(-9, 9, 700),
                                   np.repeat(1 << 20, 1100), rng.integers(0, 3, 30)])[:n]


def as_n(v, int_bytes, signed):
    v = np.asarray(v, dtype=np.int64)
    if int_bytes < 8:
        v = v.astype({2: np.int16, 4: np.int32}[int_bytes]).astype(np.int64)
    if not signed:  # lengths, dictionary keys: never negative
        v = np.abs(np.maximum(v, -(2**62)))
        if int_bytes < 8:
            v = np.minimum(v, (1 << (int_bytes * 8 - 1)) - 1)
    return v


@pytest.mark.parametrize("int_bytes", [2, 4, 8])
@pytest.mark.parametrize("signed", [True, False])
def test_rle2_bytes_are_the_reference_encoders(int_bytes, signed):
    c = G.ctx()
    rng = np.random.default_rng(100 + int_bytes)
    kinds = np.zeros(5, dtype=np.uint64)
    for n in (1, 2, 3, 4, 5, 11, 511, 512, 513, 1023, 1025, 16_384, 16_385, 70_001):
        for name, v in shapes(rng, n, int_bytes * 8):
            v = as_n(v, int_bytes, signed)
            want, stats = O.enc_rle2(v, int_bytes, signed, with_stats=True)
            got = c.encode_rle2(v, int_bytes, signed)
            if stats[4] == 0:
                assert got == want, (name, n, int_bytes, signed, len(got), len(want))
            # (where the reference panics there are no bytes of its to match: the stream must still hold the values)
            st, back = O.int_rle(got, len(v), version=2, signed=signed, nbits=int_bytes * 8)
            assert st == 0 and np.array_equal(back, v), (name, n)
            kinds += stats
    assert all(kinds[:4] > 0), kinds  # SHORT_REPEAT, DIRECT, PATCHED_BASE and DELTA were all written


def test_rle2_chain_across_many_tiles_and_levels():
    """2 M values (4 000 tiles: three levels of exit maps): random literals (a run every 512, entries that never converge), short runs
    (the chain re-synchronises all the time), and runs of 600 (tiles entered in the middle of a run)"""
    c = G.ctx()
    rng = np.random.default_rng(9)
    n = 2_000_003
    for name, v in (("random", rng.integers(-2**40, 2**40, n)),
                    ("short_runs", np.repeat(rng.integers(0, 50, n // 4 + 1), rng.integers(1, 9, n // 4 + 1))[:n]),
                    ("runs_of_600", np.repeat(rng.integers(0, 50, n // 600 + 1), 600)[:n]),
                    ("literals_then_triples", np.where(np.arange(n) % 1021 < 3, 5, rng.integers(100, 1 << 20, n)))):
        v = np.asarray(v, dtype=np.int64)
        assert c.encode_rle2(v, 8, True) == O.enc_rle2(v, 8, True), name


def test_the_device_decoder_reads_the_encoders_streams():
    rng = np.random.default_rng(3)
    for name, v in shapes(rng, 70_001):
        v = np.asarray(v, dtype=np.int64)
        stream = G.ctx().encode_rle2(v, 8, True)
        col = {"column_id": 1, "orc_type": LONG, "encoding": DIRECT_V2}
        streams = [(1, DATA, stream)]
        res = G.gpu_decode(v.size, [col], streams)
        try:
            assert res.status()[0] == 0, (name, res.status())
            G.assert_column_parity(res, 0, col, streams, v.size, 8192, what=name)
            got = np.concatenate([np.frombuffer(res.batch(b, 0)["values"], dtype=np.int64) for b in range(res.n_batches)])
            assert np.array_equal(got, v), name
        finally:
            res.free()


def test_byte_rle_and_boolean_bytes_are_the_reference_encoders():
    c = G.ctx()
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4, 127, 128, 129, 130, 131, 132, 260, 5000, 300_007):
        for v in (rng.integers(0, 256, n), np.repeat(rng.integers(0, 256, n // 4 + 1), rng.integers(1, 9, n // 4 + 1))[:n],
                  np.repeat(rng.integers(0, 3, n // 100 + 1), rng.integers(1, 400, n // 100 + 1))[:n], np.zeros(n),
                  np.where(np.arange(n) % 127 < 3, 9, rng.integers(10, 250, n))):
            v = np.asarray(v, dtype=np.uint8)
            assert c.encode_byte_rle(v) == O.enc_byte_rle(v), n
            bits = np.packbits((v & 1).astype(np.uint8), bitorder="little")
            assert c.encode_boolean(bits, len(v)) == O.enc_boolean(bits, len(v)), n
    # spare bits of the last byte are not the encoder's business: set, they must not reach the stream
    bits = np.full(3, 0xff, dtype=np.uint8)
    assert c.encode_boolean(bits, 17) == O.enc_boolean(bits, 17)


def test_size_query_and_buffers_too_small():
    c = G.ctx()
    keys = np.random.default_rng(8).integers(0, 7, 100_000).astype(np.int64)
    stream = c.encode_rle2(keys, 8, False)
    n = C.c_uint64(0)
    small = np.zeros(4, dtype=np.uint8)
    rc = c.L.orcgpu_encode_rle2_i64(c.h, keys.ctypes.data, keys.size, 0, small.ctypes.data, small.size, C.byref(n))
    assert rc == 101 and n.value == len(stream)                   # too small: INVALID_ARGUMENT, the size reported
    assert c.L.orcgpu_encode_rle2_i64(c.h, None, 0, 1, None, 0, C.byref(n)) == 0 and n.value == 0
    assert c.L.orcgpu_encode_rle2(c.h, keys.ctypes.data, keys.size, 3, 1, 0, None, 0, C.byref(n)) == 101  # N of three bytes


def expected_column(arrow_type, values, validity_bits, offsets=None):
    """the streams the reference's column encoders finish() with (writer/column.rs), from the oracle's value encoders"""
    valid = np.ones(len(validity_bits) if validity_bits is not None else (len(offsets) - 1 if offsets is not None else len(values)), dtype=bool) \
        if validity_bits is None else validity_bits.astype(bool)
    out = []
    if arrow_type == "bool":
        v = np.asarray(values, dtype=np.uint8)[valid]
        out.append((DATA, O.enc_boolean(np.packbits(v, bitorder="little"), len(v))))
    elif arrow_type == "int8":
        out.append((DATA, O.enc_byte_rle(np.asarray(values, dtype=np.int8)[valid])))
    elif arrow_type in ("int16", "int32", "int64"):
        nb = {"int16": 2, "int32": 4, "int64": 8}[arrow_type]
        out.append((DATA, O.enc_rle2(np.asarray(values)[valid].astype(np.int64), nb, True)))
    elif arrow_type in ("float32", "float64"):
        out.append((DATA, np.asarray(values)[valid].tobytes()))
    else:
        nb = 8 if arrow_type.startswith("large") else 4
        data = bytes(values)
        lens = np.diff(np.asarray(offsets, dtype=np.int64))
        out.append((DATA, b"".join(data[offsets[i]:offsets[i + 1]] for i in np.nonzero(valid)[0])))
        out.append((LENGTH, O.enc_rle2(lens[valid], nb, False)))
    # (GenericBinaryColumnEncoder::encode_array returns at once for an EMPTY array, writer/column.rs:304-307 -- before it looks at the
    # null buffer: a string / binary column of no rows finishes without a PRESENT stream; the primitive and Boolean encoders do not)
    strings_empty = arrow_type not in ("bool", "int8", "int16", "int32", "int64", "float32", "float64") and len(valid) == 0
    if validity_bits is not None and not strings_empty:
        out.append((PRESENT, O.enc_boolean(np.packbits(validity_bits.astype(np.uint8), bitorder="little"), len(validity_bits))))
    return out


@pytest.mark.parametrize("nulls", [None, 0.0, 0.3, 1.0])
def test_column_streams_are_the_reference_column_encoders(nulls):
    c = G.ctx()
    rng = np.random.default_rng(21)
    for n in (0, 1, 7, 64, 65, 1000, 100_003):
        vb = None if nulls is None else (rng.random(n) >= nulls)
        vbits = None if vb is None else np.packbits(vb.astype(np.uint8), bitorder="little")
        for t, vals in (("int8", rng.integers(-128, 128, n).astype(np.int8)), ("int16", np.repeat(rng.integers(-300, 300, n // 3 + 1), 3)[:n].astype(np.int16)),
                        ("int32", np.cumsum(rng.integers(0, 9, n)).astype(np.int32)), ("int64", rng.integers(-2**50, 2**50, n).astype(np.int64)),
                        ("float32", rng.random(n).astype(np.float32)), ("float64", rng.random(n).astype(np.float64))):
            got = c.encode_column(t, n, vals, vbits)
            assert got == expected_column(t, vals, vb), (t, n, nulls)
        b = (rng.random(n) < 0.4).astype(np.uint8)
        got = c.encode_column("bool", n, np.packbits(b, bitorder="little") if n else np.zeros(1, np.uint8), vbits)
        assert got == expected_column("bool", b, vb), ("bool", n, nulls)
        for t in ("utf8", "large_binary"):
            lens = rng.integers(0, 40, n)
            skip = 5  # a sliced array: the first offset is not zero
            offs = (np.concatenate([[0], np.cumsum(lens)]) + skip).astype(np.int64 if t.startswith("large") else np.int32)
            data = rng.integers(97, 123, int(offs[-1])).astype(np.uint8)
            got = c.encode_column(t, n, data, vbits, offs)
            assert got == expected_column(t, data.tobytes(), vb, offs), (t, n, nulls)


def test_column_with_device_buffers():
    """ORCGPU_ENC_ON_DEVICE: values that never leave the GPU (here: the Arrow buffers a decode has just produced) are encoded where they lie"""
    from orc_rust_amd import capi
    c = G.ctx()
    rng = np.random.default_rng(4)
    v = np.repeat(rng.integers(0, 1 << 33, 40_000), rng.integers(1, 6, 40_000)).astype(np.int64)
    stream = c.encode_rle2(v, 8, True)
    col = {"column_id": 1, "orc_type": LONG, "encoding": DIRECT_V2}
    res = G.gpu_decode(v.size, [col], [(1, DATA, stream)], batch_size=v.size)
    try:
        view = capi.BatchView()
        assert c.L.orcgpu_result_batch_view(res.h, 0, 0, C.byref(view)) == 0
        ecol = capi.EncColumn(capi.ARROW["int64"], capi.ENC_ON_DEVICE, v.size, None, view.values, None)
        streams = (capi.EncStream * 3)()
        ns = C.c_uint32(0)
        assert c.L.orcgpu_encode_column(c.h, C.byref(ecol), streams, C.byref(ns)) == 0 and ns.value == 1
        buf = np.zeros(streams[0].len, dtype=np.uint8)
        assert c.L.orcgpu_encode_fetch(c.h, C.byref(streams[0]), buf.ctypes.data) == 0
        assert buf.tobytes() == stream
    finally:
        res.free()


def test_encode_argument_errors():
    from orc_rust_amd import capi
    c = G.ctx()
    n = C.c_uint64(7)
    v = np.arange(10, dtype=np.int64)
    assert c.L.orcgpu_encode_rle2(c.h, None, 10, 8, 1, 0, None, 0, C.byref(n)) == 101          # values missing
    assert c.L.orcgpu_encode_rle2(c.h, v.ctypes.data, 10, 8, 1, 0, None, 0, None) == 101       # nowhere to put the size
    assert c.L.orcgpu_encode_byte_rle(c.h, None, 0, 0, None, 0, C.byref(n)) == 0 and n.value == 0
    assert c.L.orcgpu_encode_boolean(c.h, None, 0, 0, None, 0, C.byref(n)) == 0 and n.value == 0
    streams = (capi.EncStream * 3)()
    ns = C.c_uint32(9)
    col = capi.EncColumn(99, 0, 10, None, v.ctypes.data, None)                                  # an Arrow type the reference's writer does not take
    assert c.L.orcgpu_encode_column(c.h, C.byref(col), streams, C.byref(ns)) == 7 and ns.value == 0   # UnsupportedTypeVariant
    col = capi.EncColumn(capi.ARROW["utf8"], 0, 10, None, v.ctypes.data, None)                  # strings without offsets
    assert c.L.orcgpu_encode_column(c.h, C.byref(col), streams, C.byref(ns)) == 101
    offs = np.array([5, 3], dtype=np.int32)                                                     # offsets that go backwards
    col = capi.EncColumn(capi.ARROW["utf8"], 0, 1, None, v.ctypes.data, offs.ctypes.data)
    assert c.L.orcgpu_encode_column(c.h, C.byref(col), streams, C.byref(ns)) == 101
    # ... in the MIDDLE of the array (first and last offset are in order: only the device sees it) -- a negative length cast to u32
    # used to send the copy kernel far outside the values buffer
    data = np.frombuffer(b"abcdefghijklmnop", dtype=np.uint8)
    offs = np.array([0, 9, 4, 12, 16], dtype=np.int32)
    col = capi.EncColumn(capi.ARROW["utf8"], 0, 4, None, data.ctypes.data, offs.ctypes.data)
    assert c.L.orcgpu_encode_column(c.h, C.byref(col), streams, C.byref(ns)) == 101
    valid = np.array([0b1011], dtype=np.uint8)
    col = capi.EncColumn(capi.ARROW["utf8"], 0, 4, valid.ctypes.data, data.ctypes.data, offs.ctypes.data)
    assert c.L.orcgpu_encode_column(c.h, C.byref(col), streams, C.byref(ns)) == 101
    # a string array of NO rows never gets a PRESENT stream, bitmap or not (writer/column.rs:304-307: encode_array returns at once)
    out = c.encode_column("utf8", 0, np.zeros(0, dtype=np.uint8), validity=np.zeros(1, dtype=np.uint8), offsets=np.zeros(1, dtype=np.int32))
    assert [k for k, _ in out] == [1, 2] and all(len(b) == 0 for _, b in out)
    # the context still encodes afterwards
    assert c.encode_rle2(v, 8, True) == O.enc_rle2(v, 8, True)
