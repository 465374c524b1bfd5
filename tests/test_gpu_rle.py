"""GPU parity, integer RLE columns (the headline path): the HIP decoder behind the C ABI must
reproduce the oracle bit for bit on seeded synthetic streams that exercise every RLE v2
sub-encoding, RLE v1, null spacing, narrow integer types, batch boundaries and error cases."""
import numpy as np
import pytest

import gpu_util as G
import oracle_lib as O
from orc_rust_amd import gen
from test_gen_roundtrip import patterns

pytestmark = pytest.mark.gpu

LONG, INT, SHORT, DATE, BYTE, BOOLEAN, FLOAT, DOUBLE, TIMESTAMP = 4, 3, 2, 15, 1, 0, 5, 6, 9
PRESENT, DATA, SECONDARY = 0, 1, 5


def col(cid, typ, enc=2, **kw):
    d = {"column_id": cid, "orc_type": typ, "encoding": enc}
    d.update(kw)
    return d


@pytest.mark.parametrize("n", [1, 7, 512, 513, 5000, 70000])
def test_long_rlev2_patterns(n):
    rng = np.random.default_rng(n)
    cols, streams, names = [], [], []
    for i, (name, vals) in enumerate(patterns(rng, n)):
        v = np.asarray(vals, dtype=np.int64)
        for aligned in (True, False):
            cid = len(cols) + 1
            cols.append(col(cid, LONG))
            streams.append((cid, DATA, gen.rle2(v, signed=True, aligned=aligned)))
            names.append((name, aligned))
    for batch in (8192, 1000):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        for ci, c in enumerate(cols):
            G.assert_column_parity(res, ci, c, streams, n, batch, what=(names[ci], n, batch))
        res.free()


@pytest.mark.parametrize("typ,bits", [(INT, 32), (SHORT, 16), (DATE, 32)])
def test_narrow_ints(typ, bits):
    n = 20000
    rng = np.random.default_rng(bits)
    lim = 1 << (bits - 1)
    cols, streams = [], []
    pats = [
        rng.integers(-lim, lim, n), np.arange(n) % lim, np.repeat(rng.integers(-lim, lim, n // 4 + 1), 4)[:n],
        np.clip(np.cumsum(rng.integers(0, 3, n)), -lim, lim - 1),
    ]
    outl = rng.integers(0, 100, n)
    outl[rng.choice(n, n // 30, replace=False)] = lim - 1
    pats.append(outl)
    for v in pats:
        cid = len(cols) + 1
        cols.append(col(cid, typ))
        streams.append((cid, DATA, gen.rle2(np.asarray(v, dtype=np.int64), signed=True)))
    res = G.gpu_decode(n, cols, streams)
    for ci, c in enumerate(cols):
        G.assert_column_parity(res, ci, c, streams, n, 8192, what=(typ, ci))


@pytest.mark.parametrize("null_frac", [0.0, 0.1, 0.5, 0.97, 1.0])
def test_nulls_spacing(null_frac):
    n = 50000
    rng = np.random.default_rng(int(null_frac * 100))
    present = (rng.random(n) >= null_frac).astype(np.uint8)
    k = int(present.sum())
    cols, streams = [], []
    for typ, vals in ((LONG, rng.integers(-(1 << 40), 1 << 40, k)), (INT, rng.integers(-1000, 1000, k)), (SHORT, np.arange(k) % 3000),
                      (LONG, np.arange(k) * 5)):
        cid = len(cols) + 1
        cols.append(col(cid, typ))
        streams.append((cid, PRESENT, gen.boolean(present)))
        streams.append((cid, DATA, gen.rle2(np.asarray(vals, dtype=np.int64), signed=True)))
    for batch in (8192, 777):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        for ci, c in enumerate(cols):
            G.assert_column_parity(res, ci, c, streams, n, batch, what=("nulls", null_frac, ci, batch))


@pytest.mark.parametrize("cut", [0, 1, 37, 500, 1023, 1024, 1500, 3000])
def test_present_stream_that_fails_is_swallowed(cut):
    """derive_present_vec (array_decoder/mod.rs:228-251) maps a PRESENT decode error to "no PRESENT
    stream": the failing batch and every later one are decoded with all rows valid."""
    n = 30000
    rng = np.random.default_rng(cut)
    present = (rng.random(n) >= 0.3).astype(np.uint8)
    pstream = gen.boolean(present)
    pcut = pstream[: min(cut, len(pstream))]
    # enough DATA values for any outcome (all rows valid from some batch on)
    vals = rng.integers(-1000, 1000, n)
    cols = [col(1, LONG), col(2, INT)]
    streams = [(1, PRESENT, pcut), (1, DATA, gen.rle2(vals, signed=True)), (2, PRESENT, pcut), (2, DATA, gen.rle2(vals % 100, signed=True))]
    for batch in (8192, 1000):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        for ci, c in enumerate(cols):
            G.assert_column_parity(res, ci, c, streams, n, batch, what=("present-cut", cut, ci, batch))
        res.free()


def test_rlev1_columns():
    n = 30000
    rng = np.random.default_rng(11)
    cols, streams = [], []
    for name, vals in patterns(rng, n):
        cid = len(cols) + 1
        cols.append(col(cid, LONG, enc=0))
        streams.append((cid, DATA, gen.rle1(np.asarray(vals, dtype=np.int64), signed=True)))
    res = G.gpu_decode(n, cols, streams)
    for ci, c in enumerate(cols):
        G.assert_column_parity(res, ci, c, streams, n, 8192, what=("rle1", ci))


def test_byte_boolean_float_columns():
    n = 40000
    rng = np.random.default_rng(5)
    present = (rng.random(n) >= 0.2).astype(np.uint8)
    k = int(present.sum())
    cols, streams = [], []
    b = np.repeat(rng.integers(-128, 128, n, dtype=np.int64), rng.integers(1, 6, n))[:n].astype(np.int8)
    cols.append(col(1, BYTE))
    streams.append((1, DATA, gen.byte_rle(b)))
    cols.append(col(2, BYTE))
    streams.append((2, PRESENT, gen.boolean(present)))
    streams.append((2, DATA, gen.byte_rle(b[:k])))
    bits = (rng.random(n) < 0.3).astype(np.uint8)
    cols.append(col(3, BOOLEAN))
    streams.append((3, DATA, gen.boolean(bits)))
    cols.append(col(4, BOOLEAN))
    streams.append((4, PRESENT, gen.boolean(present)))
    streams.append((4, DATA, gen.boolean(bits[:k])))
    f = rng.standard_normal(n).astype(np.float32)
    d = rng.standard_normal(n)
    cols.append(col(5, FLOAT))
    streams.append((5, DATA, f.view(np.uint8)))
    cols.append(col(6, DOUBLE))
    streams.append((6, PRESENT, gen.boolean(present)))
    streams.append((6, DATA, d[:k].view(np.uint8)))
    for batch in (8192, 1001):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        for ci, c in enumerate(cols):
            G.assert_column_parity(res, ci, c, streams, n, batch, what=("misc", ci, batch))


def test_timestamps():
    n = 30000
    rng = np.random.default_rng(9)
    present = (rng.random(n) >= 0.1).astype(np.uint8)
    k = int(present.sum())
    secs = rng.integers(-2_000_000_000, 2_000_000_000, k)
    nanos_us = rng.integers(0, 1_000_000, k) * 1000
    enc = np.where(nanos_us == 0, 0, (nanos_us // 1000 << 3) | 2)  # trailing-zero code 2 = x1000
    cols = [col(1, TIMESTAMP)]
    streams = [(1, PRESENT, gen.boolean(present)), (1, DATA, gen.rle2(secs, signed=True)), (1, SECONDARY, gen.rle2(enc, signed=False))]
    res = G.gpu_decode(n, cols, streams)
    G.assert_column_parity(res, 0, cols[0], streams, n, 8192, what="timestamp")
    # out-of-range seconds -> DecodeTimestamp error in the batch that holds the value
    secs2 = secs.copy()
    secs2[k // 2] = 1 << 40
    streams2 = [(1, PRESENT, gen.boolean(present)), (1, DATA, gen.rle2(secs2, signed=True)), (1, SECONDARY, gen.rle2(enc, signed=False))]
    res = G.gpu_decode(n, cols, streams2)
    assert res.status()[0] == O.DECODE_TIMESTAMP
    G.assert_column_parity(res, 0, cols[0], streams2, n, 8192, what="timestamp-overflow")


def test_error_cases():
    n = 3000
    rng = np.random.default_rng(2)
    v = rng.integers(0, 1 << 30, n)
    good = gen.rle2(v, signed=True)
    cases = {
        "truncated": good[: len(good) // 2],
        "empty": good[:0],
        "one_short": good[:-1],
        "delta_overflow": np.frombuffer(bytes([0xC0, 0x09]) + b"\xfc\xff\xff\xff\xff\xff\xff\xff\xff\x01" + b"\x02", dtype=np.uint8),
        "sr_too_wide_for_int": np.frombuffer(bytes([0x3A, 1, 2, 3, 4, 5, 6, 7, 8]), dtype=np.uint8),
    }
    for name, data in cases.items():
        for typ in (LONG, INT):
            cols = [col(1, typ)]
            streams = [(1, DATA, data)]
            for batch in (8192, 500):
                res = G.gpu_decode(n, cols, streams, batch_size=batch)
                G.assert_column_parity(res, 0, cols[0], streams, n, batch, what=(name, typ, batch))


def _zz_varint(v):
    u = (v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1
    out = bytearray()
    while True:
        b = u & 0x7F
        u >>= 7
        if u:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


@pytest.mark.parametrize("where", [None, 3, 150, 199])
@pytest.mark.parametrize("step", [3, -3, 0])
def test_fixed_delta_full_runs(where, step):
    """200 fixed-delta runs of 512 values (closed-form path of the expansion); optionally one run whose
    progression leaves the i64 range part-way (the reference fails on the first overflowing add)."""
    runs, n = 200, 200 * 512
    data = bytearray()
    for r in range(runs):
        base = r * 10_000 - 1_000_000
        if where is not None and r == where and step != 0:
            base = ((1 << 63) - 1 - 700) if step > 0 else (-(1 << 63) + 700)
        data += bytes([0xC1, 0xFF]) + _zz_varint(base) + _zz_varint(step)
    data = np.frombuffer(bytes(data), dtype=np.uint8)
    cols = [col(1, LONG)]
    streams = [(1, DATA, data)]
    for batch in (8192, 1000):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        G.assert_column_parity(res, 0, cols[0], streams, n, batch, what=("fixed-delta", where, step, batch))
        res.free()


@pytest.mark.parametrize("enc", ["rle2", "rle1", "byte", "bool"])
def test_truncation_sweep(enc):
    """Every way a stream can end early: the reference yields the complete batches and fails the one
    that runs dry (IoError / OutOfSpec by sub-encoding).  Sixty cut points per pattern, two batch sizes."""
    n = 2500
    rng = np.random.default_rng(7)
    pats = {name: np.asarray(v, dtype=np.int64) for name, v in patterns(rng, n)}
    cases = []
    if enc == "rle2":
        for name in list(pats)[:6]:
            cases.append((LONG, 2, gen.rle2(pats[name], signed=True)))
    elif enc == "rle1":
        for name in list(pats)[:4]:
            cases.append((LONG, 0, gen.rle1(pats[name], signed=True)))
    elif enc == "byte":
        b = np.repeat(rng.integers(-128, 128, n, dtype=np.int64), rng.integers(1, 6, n))[:n].astype(np.int8)
        cases.append((BYTE, 0, gen.byte_rle(b)))
    else:
        cases.append((BOOLEAN, 0, gen.boolean((rng.random(n) < 0.3).astype(np.uint8))))
    for typ, e, full in cases:
        cuts = sorted(set([0, 1, 2, 3, len(full) - 1] + [int(x) for x in rng.integers(0, len(full), 55)]))
        for cut in cuts:
            data = full[:cut]
            c = col(1, typ, enc=e)
            streams = [(1, DATA, data)]
            for batch in (1024, 700):
                res = G.gpu_decode(n, [c], streams, batch_size=batch)
                G.assert_column_parity(res, 0, c, streams, n, batch, what=(enc, typ, cut, len(full), batch))
                res.free()


@pytest.mark.parametrize("what", ["long+nulls", "boolean+nulls", "string-direct", "string-dict", "decimal", "timestamp"])
def test_truncation_sweep_columns_with_several_streams(what):
    """One stream of a multi-stream column cut short at a time (nulls present): the failing batch and the
    error kind must be the reference's (IoError / OutOfSpec / Arrow / construction-time failures)."""
    STRING, LENGTH, DICT, DECIMAL = 7, 2, 3, 14
    n = 2600
    rng = np.random.default_rng(len(what))
    present = (rng.random(n) >= 0.2).astype(np.uint8)
    k = int(present.sum())
    P = gen.boolean(present)
    if what == "long+nulls":
        c = col(1, LONG)
        streams = {PRESENT: P, DATA: gen.rle2(rng.integers(-10**6, 10**6, k), signed=True)}
    elif what == "boolean+nulls":
        c = col(1, BOOLEAN, enc=0)
        streams = {PRESENT: P, DATA: gen.boolean((rng.random(k) < 0.4).astype(np.uint8))}
    elif what == "string-direct":
        words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"", "h\u00e9llo".encode()]
        idx = rng.integers(0, len(words), k)
        c = col(1, STRING)
        streams = {PRESENT: P, LENGTH: gen.rle2(np.array([len(words[i]) for i in idx], dtype=np.int64), signed=False),
                   DATA: np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8)}
    elif what == "string-dict":
        dwords = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK"]
        c = col(1, STRING, enc=3, dictionary_size=len(dwords))
        streams = {PRESENT: P, DATA: gen.rle2(rng.integers(0, len(dwords), k), signed=False),
                   LENGTH: gen.rle2(np.array([len(w) for w in dwords], dtype=np.int64), signed=False),
                   DICT: np.frombuffer(b"".join(dwords), dtype=np.uint8)}
    elif what == "decimal":
        c = col(1, DECIMAL, precision=38, scale=3)
        streams = {PRESENT: P, DATA: gen.varint128([int(x) for x in rng.integers(-10**15, 10**15, k)]),
                   SECONDARY: gen.rle2(rng.integers(0, 6, k), signed=True)}
    else:
        secs = rng.integers(-2_000_000_000, 2_000_000_000, k)
        nanos = rng.integers(0, 1_000_000, k) * 1000
        c = col(1, TIMESTAMP)
        streams = {PRESENT: P, DATA: gen.rle2(secs, signed=True), SECONDARY: gen.rle2(np.where(nanos == 0, 0, (nanos // 1000 << 3) | 2), signed=False)}
    for kind, full in streams.items():
        cuts = sorted(set([0, 1, max(0, len(full) - 1)] + [int(x) for x in rng.integers(0, max(1, len(full)), 14)]))
        for cut in cuts:
            s = [(1, kk, (vv[:cut] if kk == kind else vv)) for kk, vv in streams.items()]
            for batch in (1024, 700):
                res = G.gpu_decode(n, [c], s, batch_size=batch)
                G.assert_column_parity(res, 0, c, s, n, batch, what=(what, "stream", kind, "cut", cut, len(full), batch))
                res.free()


@pytest.mark.parametrize("seed", [11, 12, 13])
@pytest.mark.parametrize("enc,typ", [("rle2", LONG), ("rle2", INT), ("rle2", SHORT), ("rle1", LONG), ("rle1", INT), ("rle1", SHORT)])
def test_corruption_sweep(enc, typ, seed):
    """One byte of a valid stream replaced by a random value, 40 positions per pattern: whatever the
    damage makes of the stream (other values, OutOfSpec, VarintTooLarge, IoError at the end), batches,
    values and the failing batch / error kind must be the reference's."""
    n = 2500
    rng = np.random.default_rng(seed)
    bits = {LONG: 40, INT: 31, SHORT: 15}[typ]
    lim = 1 << (bits - 1)
    pats = [rng.integers(-lim, lim, n), np.arange(n) % lim, np.repeat(rng.integers(-lim, lim, n // 5 + 1), 5)[:n],
            np.clip(np.cumsum(rng.integers(0, 200, n)), -lim, lim - 1)]
    outl = rng.integers(0, 100, n)
    outl[rng.choice(n, n // 30, replace=False)] = lim - 1
    pats.append(outl)
    for v in pats:
        v = np.asarray(v, dtype=np.int64)
        full = gen.rle2(v, signed=True) if enc == "rle2" else gen.rle1(v, signed=True)
        for pos in sorted(set(int(x) for x in rng.integers(0, len(full), 60))):
            data = full.copy()
            data[pos] = int(rng.integers(0, 256))
            c = col(1, typ, enc=2 if enc == "rle2" else 0)
            streams = [(1, DATA, data)]
            res = G.gpu_decode(n, [c], streams, batch_size=1024)
            G.assert_column_parity(res, 0, c, streams, n, 1024, what=(enc, typ, "byte", pos, int(data[pos]), "was", int(full[pos])))
            res.free()


@pytest.mark.parametrize("batch", [1, 63, 20000, 100000])
def test_unusual_batch_sizes(batch):
    """Batches smaller than a validity word and larger than the kernels' tiles (dictionary rows are
    scanned in tiles of 8192): same bytes as the oracle."""
    STRING, LENGTH, DICT = 7, 2, 3
    n = 70000 if batch > 1 else 300
    rng = np.random.default_rng(batch)
    present = (rng.random(n) >= 0.2).astype(np.uint8)
    k = int(present.sum())
    dwords = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK", b""]
    cols = [col(1, STRING, enc=3, dictionary_size=len(dwords)), col(2, LONG), col(3, STRING), col(4, BOOLEAN, enc=0)]
    idx = rng.integers(0, len(dwords), k)
    streams = [(1, PRESENT, gen.boolean(present)), (1, DATA, gen.rle2(rng.integers(0, len(dwords), k), signed=False)),
               (1, LENGTH, gen.rle2(np.array([len(w) for w in dwords], dtype=np.int64), signed=False)),
               (1, DICT, np.frombuffer(b"".join(dwords), dtype=np.uint8)),
               (2, PRESENT, gen.boolean(present)), (2, DATA, gen.rle2(rng.integers(-10**9, 10**9, k), signed=True)),
               (3, PRESENT, gen.boolean(present)), (3, LENGTH, gen.rle2(np.array([len(dwords[i]) for i in idx], dtype=np.int64), signed=False)),
               (3, DATA, np.frombuffer(b"".join(dwords[i] for i in idx), dtype=np.uint8)),
               (4, PRESENT, gen.boolean(present)), (4, DATA, gen.boolean((rng.random(k) < 0.5).astype(np.uint8)))]
    res = G.gpu_decode(n, cols, streams, batch_size=batch)
    for ci, c in enumerate(cols):
        G.assert_column_parity(res, ci, c, streams, n, batch, what=("batch", batch, ci))
    res.free()


def test_strings_direct_and_dictionary():
    STRING, BINARY, LENGTH, DICT = 7, 8, 2, 3
    n = 30000
    rng = np.random.default_rng(21)
    present = (rng.random(n) >= 0.15).astype(np.uint8)
    k = int(present.sum())
    words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK", "héllo".encode(), "日本語".encode(), b"", "\U0001f600".encode()]
    cols, streams = [], []
    # direct strings with nulls
    idx = rng.integers(0, len(words), k)
    lens = np.array([len(words[i]) for i in idx], dtype=np.int64)
    blob = b"".join(words[i] for i in idx)
    cols.append(col(1, STRING))
    streams += [(1, PRESENT, gen.boolean(present)), (1, LENGTH, gen.rle2(lens, signed=False)), (1, DATA, np.frombuffer(blob, dtype=np.uint8))]
    # dictionary strings with nulls (keys unsigned RLE)
    dwords = sorted(set(words))
    dlens = np.array([len(w) for w in dwords], dtype=np.int64)
    keys = rng.integers(0, len(dwords), k)
    cols.append(col(2, STRING, enc=3, dictionary_size=len(dwords)))
    streams += [(2, PRESENT, gen.boolean(present)), (2, DATA, gen.rle2(keys, signed=False)), (2, LENGTH, gen.rle2(dlens, signed=False)),
                (2, DICT, np.frombuffer(b"".join(dwords), dtype=np.uint8))]
    # dictionary without nulls, binary direct
    keys2 = rng.integers(0, len(dwords), n)
    cols.append(col(3, STRING, enc=3, dictionary_size=len(dwords)))
    streams += [(3, DATA, gen.rle2(keys2, signed=False)), (3, LENGTH, gen.rle2(dlens, signed=False)), (3, DICT, np.frombuffer(b"".join(dwords), dtype=np.uint8))]
    raw = rng.integers(0, 256, 5 * n, dtype=np.uint8)
    cols.append(col(4, BINARY))
    streams += [(4, LENGTH, gen.rle2(np.full(n, 5), signed=False)), (4, DATA, raw)]
    for batch in (8192, 999):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        for ci, c in enumerate(cols):
            G.assert_column_parity(res, ci, c, streams, n, batch, what=("strings", ci, batch))
    # error cases: key out of range, invalid UTF-8, truncated DATA
    bad_keys = keys2.copy()
    bad_keys[n // 2] = len(dwords)
    c = [col(1, STRING, enc=3, dictionary_size=len(dwords))]
    s = [(1, DATA, gen.rle2(bad_keys, signed=False)), (1, LENGTH, gen.rle2(dlens, signed=False)), (1, DICT, np.frombuffer(b"".join(dwords), dtype=np.uint8))]
    res = G.gpu_decode(n, c, s)
    assert res.status()[0] == O.ARROW
    G.assert_column_parity(res, 0, c[0], s, n, 8192, what="bad key")
    blob2 = bytearray(blob)
    blob2[len(blob2) // 2] = 0xFF
    c = [col(1, STRING)]
    s = [(1, PRESENT, gen.boolean(present)), (1, LENGTH, gen.rle2(lens, signed=False)), (1, DATA, np.frombuffer(bytes(blob2), dtype=np.uint8))]
    res = G.gpu_decode(n, c, s)
    G.assert_column_parity(res, 0, c[0], s, n, 8192, what="bad utf8")
    s = [(1, PRESENT, gen.boolean(present)), (1, LENGTH, gen.rle2(lens, signed=False)), (1, DATA, np.frombuffer(blob[: len(blob) // 3], dtype=np.uint8))]
    res = G.gpu_decode(n, c, s)
    G.assert_column_parity(res, 0, c[0], s, n, 8192, what="short data")


def test_decimals():
    DECIMAL = 14
    n = 25000
    rng = np.random.default_rng(33)
    present = (rng.random(n) >= 0.1).astype(np.uint8)
    k = int(present.sum())
    vals = [int(x) for x in rng.integers(-10**15, 10**15, k)]
    vals[0] = 10**30
    vals[1] = -(10**36)
    scales = rng.integers(0, 6, k)
    cols = [col(1, DECIMAL, precision=38, scale=3)]
    streams = [(1, PRESENT, gen.boolean(present)), (1, DATA, gen.varint128(vals)), (1, SECONDARY, gen.rle2(scales, signed=True))]
    for batch in (8192, 1234):
        res = G.gpu_decode(n, cols, streams, batch_size=batch)
        G.assert_column_parity(res, 0, cols[0], streams, n, batch, what=("decimal", batch))
    data = gen.varint128(vals)
    streams2 = [(1, PRESENT, gen.boolean(present)), (1, DATA, data[: len(data) // 2]), (1, SECONDARY, gen.rle2(scales, signed=True))]
    res = G.gpu_decode(n, cols, streams2)
    G.assert_column_parity(res, 0, cols[0], streams2, n, 8192, what="decimal truncated")


@pytest.mark.parametrize("precision, scale", [(0, 0), (39, 2), (5, 6), (38, 39), (256, 0), (266, 2), (10, 255)])
def test_decimal_types_arrow_refuses(precision, scale):
    """array_decoder/decimal.rs:96-100: a precision outside 1..=38 (taken `as u8`), a scale above 38 or above the precision is an
    ArrowError of the first batch -- behind a stream failure of that batch --; with nulls and without (the oracle restates it,
    tests/test_oracle_kat.py::test_decimal_types_arrow_refuses)."""
    DECIMAL = 14
    n = 20000
    rng = np.random.default_rng(precision * 100 + scale)
    for nulls in (False, True):
        present = (rng.random(n) >= 0.2).astype(np.uint8) if nulls else np.ones(n, dtype=np.uint8)
        k = int(present.sum())
        vals = [int(x) for x in rng.integers(-10**9, 10**9, k)]
        cols = [col(1, DECIMAL, precision=precision, scale=scale)]
        streams = ([(1, PRESENT, gen.boolean(present))] if nulls else []) + [(1, DATA, gen.varint128(vals)), (1, SECONDARY, gen.rle2(np.full(k, scale & 0x7f, dtype=np.int64), signed=True))]
        res = G.gpu_decode(n, cols, streams)
        G.assert_column_parity(res, 0, cols[0], streams, n, 8192, what=("decimal type", precision, scale, nulls))
        cut = [(c, kd, b if kd != DATA else b[: len(b) // 3]) for c, kd, b in streams]
        res = G.gpu_decode(n, cols, cut)
        G.assert_column_parity(res, 0, cols[0], cut, n, 8192, what=("decimal type, DATA cut", precision, scale, nulls))


@pytest.mark.parametrize("compression", ["none", "zstd"])
def test_decimal_scales_that_are_the_columns_scale_throughout(compression):
    """A Decimal column without nulls whose SECONDARY stream holds the column's scale and nothing else -- what every writer makes --
    is decoded without expanding that stream (rle2_uniform_kernel: the stream must BE k runs of 512 and a shorter last one, and hold
    a value per row).  Row counts around the run length (the last run as a DELTA run, as a SHORT_REPEAT, as one or two literal
    values: not the pattern -- the expansion takes those), a scale that is NOT the column's, one odd value in the middle, a stream
    that ends too early: every one of them must come out as the oracle has it."""
    DECIMAL = 14
    rng = np.random.default_rng(77)

    def case(n, scales, col_scale=2, cut=None, what=""):
        vals = [int(x) for x in rng.integers(-10**12, 10**12, n)]
        sec = gen.rle2(np.asarray(scales, dtype=np.int64), signed=True)
        if cut is not None:
            sec = sec[:cut]
        cols = [col(1, DECIMAL, precision=20, scale=col_scale)]
        raw = [(1, DATA, gen.varint128(vals)), (1, SECONDARY, sec)]
        streams = raw if compression == "none" else [(c, k, gen.compress_stream(b, compression, 4096)) for c, k, b in raw]
        kw = {} if compression == "none" else {"compression": compression, "block_size": 4096}
        res = G.gpu_decode(n, cols, streams, **kw)
        G.assert_column_parity(res, 0, cols[0], streams, n, 8192, what=("uniform scales", what, n, compression), **kw)

    for n in (1, 2, 3, 9, 10, 11, 511, 512, 513, 514, 515, 522, 523, 1024, 1025, 20000, 20480):
        case(n, np.full(n, 2), what="the column's scale")
    case(20000, np.full(20000, 3), what="another scale throughout")
    odd = np.full(20000, 2)
    odd[12345] = 4
    case(20000, odd, what="one odd value")
    case(20000, np.full(20000, 2), col_scale=0, what="scale 0 column, scales 2")
    case(20000, np.full(20000, 0), col_scale=0, what="scale 0 throughout")
    full = gen.rle2(np.full(20000, 2, dtype=np.int64), signed=True)
    case(20000, np.full(20000, 2), cut=len(full) - 4, what="a run short")
    case(20000, np.full(20000, 2), cut=len(full) - 1, what="a byte short")


@pytest.mark.parametrize("kind", ["snappy", "lz4", "zlib", "zstd"])
@pytest.mark.parametrize("block", [64, 4096, 262144])
def test_compressed_streams(kind, block):
    """Chunk framing + block codecs: runs and varints straddle chunk boundaries (32..64-byte chunks
    as in the reference's fixtures, scripts/write.py:82-96), original and compressed chunks mix."""
    STRING, LENGTH = 7, 2
    n = 40000
    rng = np.random.default_rng(block)
    present = (rng.random(n) >= 0.1).astype(np.uint8)
    k = int(present.sum())
    cols, streams = [], []
    for i, vals in enumerate((rng.integers(0, 1 << 40, k), np.arange(k) * 7, np.repeat(rng.integers(0, 50, k // 6 + 1), 6)[:k],
                              rng.integers(0, 7, k))):
        cid = i + 1
        cols.append(col(cid, LONG))
        streams.append((cid, PRESENT, gen.compress_stream(gen.boolean(present), kind, block)))
        streams.append((cid, DATA, gen.compress_stream(gen.rle2(np.asarray(vals, dtype=np.int64), signed=True), kind, block)))
    words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK"]
    idx = rng.integers(0, len(words), n)
    blob = b"".join(words[i] for i in idx)
    lens = np.array([len(words[i]) for i in idx], dtype=np.int64)
    cols.append(col(9, STRING))
    streams.append((9, LENGTH, gen.compress_stream(gen.rle2(lens, signed=False), kind, block)))
    streams.append((9, DATA, gen.compress_stream(np.frombuffer(blob, dtype=np.uint8), kind, block)))
    res = G.gpu_decode(n, cols, streams, compression=kind, block_size=block)
    assert res.status()[0] == 0, res.status()
    for ci, c in enumerate(cols):
        G.assert_column_parity(res, ci, c, streams, n, 8192, compression=kind, block_size=block, what=(kind, block, ci))
