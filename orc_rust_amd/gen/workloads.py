"""Synthetic workloads of BASELINE.md (configs C2..C5) as stripe stream sets for the C ABI.

Host-side tooling for bench.py, the tests and profiles/: every function returns, per stripe,
`(n_rows, columns, streams, expect)` where `columns` / `streams` are what `capi.Context.stage`
takes (the inputs of the reference's seam: Stripe.columns + Stripe.stream_map, stripe.rs:119-125)
and `expect` holds, per column, the Arrow buffers the generated VALUES imply (values bytes, string
lengths) -- the full-size check of bench.py compares the decoded buffers with them.  Nothing here
is on the decode path.

C4 "lineitem": schema of the reference's scripts/convert_tpch.py:46-63; value domains of TPC-H
4.2.3 (gen/tpchgen.c); encodings as an ORC writer with dictionary encoding enabled chooses them:
RLE v2 everywhere (DIRECT_V2 / DICTIONARY_V2), the four low-cardinality strings dictionary
encoded (sorted dictionary), l_comment direct, Decimal128(15,2) as zigzag varints + a SECONDARY
scale stream, no PRESENT streams (lineitem has no nulls), Zstandard level 3 in 256 KiB chunks.
Stripes hold LINEITEM_STRIPE_ROWS rows: what the ORC C++ writer (PyArrow 25) produces for this
table with stripe_size = 64 MiB (measured in the build container: tests/golden/make_lineitem.py).
"""
import ctypes as C

import numpy as np

from . import rle2_segments, boolean, compress_stream, lib, rle2, splitmix64, varint64

BOOLEAN, BYTE, SHORT, INT, LONG, FLOAT, DOUBLE, STRING, BINARY, TIMESTAMP = range(10)
DECIMAL, DATE = 14, 15
PRESENT, DATA, LENGTH, DICTIONARY_DATA, SECONDARY = 0, 1, 2, 3, 5
DIRECT_V2, DICTIONARY_V2 = 2, 3

LINEITEM_SF1_ROWS = 6_001_215
LINEITEM_STRIPE_ROWS = 2_189_312  # rows per stripe the ORC C++ writer produced (2 138 x 1024); see the module docstring

DICTS = {
    "l_returnflag": [b"A", b"N", b"R"],
    "l_linestatus": [b"F", b"O"],
    "l_shipinstruct": [b"COLLECT COD", b"DELIVER IN PERSON", b"NONE", b"TAKE BACK RETURN"],
    "l_shipmode": [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK"],
}
# (name, ORC type, how it is generated / encoded)
LINEITEM = [
    ("l_orderkey", LONG, "i64"), ("l_partkey", LONG, "i64"), ("l_suppkey", LONG, "i64"), ("l_linenumber", INT, "i32"),
    ("l_quantity", DECIMAL, "dec"), ("l_extendedprice", DECIMAL, "dec"), ("l_discount", DECIMAL, "dec"), ("l_tax", DECIMAL, "dec"),
    ("l_returnflag", STRING, "dict"), ("l_linestatus", STRING, "dict"), ("l_shipdate", DATE, "i32"), ("l_commitdate", DATE, "i32"),
    ("l_receiptdate", DATE, "i32"), ("l_shipinstruct", STRING, "dict"), ("l_shipmode", STRING, "dict"), ("l_comment", STRING, "direct"),
]
# Arrow bytes per row of each column (values + offsets; chars of l_comment average 26.5): used to balance column shards
LINEITEM_ARROW_BYTES_PER_ROW = [8, 8, 8, 4, 16, 16, 16, 16, 5, 5, 4, 4, 4, 4 + 12, 4 + 4.3, 4 + 26.5]


class _Cols(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("orderkey", "partkey", "suppkey", "quantity", "extendedprice", "discount", "tax", "linenumber",
                                          "shipdate", "commitdate", "receiptdate", "comment_len", "returnflag", "linestatus",
                                          "shipinstruct", "shipmode", "comment")]


def lineitem_table(rows=LINEITEM_SF1_ROWS, seed=7, scale_factor=1, names=None):
    """Column arrays of `rows` lineitem rows (dict name -> ndarray; l_comment -> (lengths int32, bytes uint8)).
    `names` limits the columns that are materialised (column shards generate only their own)."""
    want = set(names) if names is not None else {n for n, _, _ in LINEITEM}
    out, st = {}, _Cols()
    i64 = ("orderkey", "partkey", "suppkey", "quantity", "extendedprice", "discount", "tax")
    i32 = ("linenumber", "shipdate", "commitdate", "receiptdate")
    u8 = ("returnflag", "linestatus", "shipinstruct", "shipmode")
    for group, dt in ((i64, np.int64), (i32, np.int32), (u8, np.uint8)):
        for f in group:
            if "l_" + f in want:
                out["l_" + f] = np.zeros(rows, dtype=dt)
                setattr(st, f, out["l_" + f].ctypes.data)
    if "l_comment" in want:
        clen = np.zeros(rows, dtype=np.int32)
        cbytes = np.zeros(44 * rows + 64, dtype=np.uint8)
        st.comment_len, st.comment = clen.ctypes.data, cbytes.ctypes.data
    nb = lib().orcgen_lineitem(seed, rows, scale_factor, C.byref(st))
    if "l_comment" in want:
        out["l_comment"] = (clen, cbytes[:nb].copy())
    return out


def _dict_arrow(words, keys):
    """Arrow value bytes + lengths of a dictionary column: what cast(DictionaryArray, Utf8) materialises."""
    lens = np.array([len(w) for w in words], dtype=np.int32)
    width = int(lens.max())
    mat = np.zeros((len(words), width), dtype=np.uint8)
    for i, w in enumerate(words):
        mat[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
    klens = lens[keys]
    mask = np.arange(width, dtype=np.int32)[None, :] < klens[:, None]
    return mat[keys][mask], klens


def _comp(buf, compression, block_size):
    return buf if compression == "none" else compress_stream(buf, compression, block_size)


def lineitem_stripe(table, lo, hi, compression="zstd", block_size=262144, column_ids=None, want_expect=True):
    """Stripe of rows [lo, hi) of `table`.  Column ids are 1..16 in schema order (0 is the root struct)."""
    cols, streams, expect = [], [], {}
    n = hi - lo
    for cid, (name, typ, how) in enumerate(LINEITEM, start=1):
        if (column_ids is not None and cid not in column_ids) or name not in table:
            continue
        if how in ("i64", "i32"):
            v = table[name][lo:hi]
            cols.append({"column_id": cid, "orc_type": typ, "encoding": DIRECT_V2, "name": name})
            streams.append((cid, DATA, _comp(rle2(v.astype(np.int64), signed=True), compression, block_size)))
            if want_expect:
                expect[cid] = {"values": v.tobytes()}
        elif how == "dec":
            v = table[name][lo:hi]
            cols.append({"column_id": cid, "orc_type": typ, "encoding": DIRECT_V2, "precision": 15, "scale": 2, "name": name})
            streams.append((cid, DATA, _comp(varint64(v), compression, block_size)))
            streams.append((cid, SECONDARY, _comp(rle2(np.full(n, 2, dtype=np.int64), signed=True), compression, block_size)))
            if want_expect:
                wide = np.zeros((n, 2), dtype=np.int64)
                wide[:, 0] = v
                wide[:, 1] = v >> 63
                expect[cid] = {"values": wide.tobytes()}
        elif how == "dict":
            words = DICTS[name]
            keys = table[name][lo:hi]
            cols.append({"column_id": cid, "orc_type": typ, "encoding": DICTIONARY_V2, "dictionary_size": len(words), "name": name})
            streams.append((cid, DATA, _comp(rle2(keys.astype(np.int64), signed=False), compression, block_size)))
            streams.append((cid, LENGTH, _comp(rle2(np.array([len(w) for w in words], dtype=np.int64), signed=False), compression, block_size)))
            streams.append((cid, DICTIONARY_DATA, _comp(np.frombuffer(b"".join(words), dtype=np.uint8), compression, block_size)))
            if want_expect:
                vb, kl = _dict_arrow(words, keys)
                expect[cid] = {"values": vb.tobytes(), "lengths": kl}
        else:  # direct string
            clen, cbytes = table[name]
            start = int(clen[:lo].sum(dtype=np.int64))
            lens = clen[lo:hi]
            blob = cbytes[start:start + int(lens.sum(dtype=np.int64))]
            cols.append({"column_id": cid, "orc_type": typ, "encoding": DIRECT_V2, "name": name})
            streams.append((cid, LENGTH, _comp(rle2(lens.astype(np.int64), signed=False), compression, block_size)))
            streams.append((cid, DATA, _comp(blob, compression, block_size)))
            if want_expect:
                expect[cid] = {"values": blob.tobytes(), "lengths": lens}
    return n, cols, streams, expect


def lineitem_stripes(rows=LINEITEM_SF1_ROWS, stripe_rows=LINEITEM_STRIPE_ROWS, compression="zstd", seed=7, column_ids=None, want_expect=True):
    names = None if column_ids is None else [LINEITEM[c - 1][0] for c in column_ids]
    table = lineitem_table(rows, seed, names=names)
    out = []
    for lo in range(0, rows, stripe_rows):
        out.append(lineitem_stripe(table, lo, min(rows, lo + stripe_rows), compression, column_ids=column_ids, want_expect=want_expect))
    return out


# ---- C5: RLE v2 PATCHED_BASE seconds + Timestamp(ns), LZ4 ------------------------------------
ORC_EPOCH = 1420070400  # 2015-01-01T00:00:00Z, writer timezone GMT (array_decoder/timestamp.rs:51)


def c5_values(n, stripe_no):
    """seconds since the ORC epoch: i.i.d. U[0, 65535] with exactly 20 positions of every 512-value run replaced by
    U[2^24, 2^30) (3.9 %: below the chooser's 5 % cut-off and the 31-entry patch list limit => every DATA run is
    PATCHED_BASE); nanoseconds: multiples of 1000 (microsecond precision => trailing-zero code 2 => DIRECT)."""
    secs = (splitmix64(5000 + stripe_no, n) & np.uint64(0xFFFF)).astype(np.int64)
    runs = (n + 511) // 512
    # 20 distinct positions per run: one in each of 20 strata of 25 values (positions 0..499 of the run)
    r = splitmix64(5500 + stripe_no, runs * 20).reshape(runs, 20)
    pos = (r % np.uint64(25)).astype(np.int64) + (np.arange(20, dtype=np.int64) * 25)[None, :]
    idx = (pos + (np.arange(runs, dtype=np.int64) * 512)[:, None]).ravel()
    idx = idx[idx < n]
    out = splitmix64(6000 + stripe_no, idx.size)
    secs[idx] = ((out % np.uint64((1 << 30) - (1 << 24))) + np.uint64(1 << 24)).astype(np.int64)
    micros = (splitmix64(6500 + stripe_no, n) % np.uint64(1_000_000)).astype(np.int64)
    return secs, micros


def c5_stripe(n, stripe_no, compression="lz4", block_size=262144, want_expect=True):
    secs, micros = c5_values(n, stripe_no)
    # encoding/timestamp.rs:121-132 reads nanos as (value >> 3) * 10^(zeros + 1) with zeros = value & 7; a writer stores
    # microsecond-precision nanos n = m * 1000 as (m << 3) | 2 (m not divisible by 10), or with more zeros removed
    nanos_enc = np.zeros(n, dtype=np.int64)
    m = micros.copy()
    zeros = np.full(n, 2, dtype=np.int64)   # three trailing zeros of m * 1000 -> code 2
    for _ in range(5):
        more = (m % 10 == 0) & (m != 0) & (zeros < 7)
        m = np.where(more, m // 10, m)
        zeros = np.where(more, zeros + 1, zeros)
    nanos_enc = np.where(micros == 0, 0, (m << 3) | zeros)
    data, stats = rle2(secs, signed=True, stats=True)
    streams = [(1, DATA, _comp(data, compression, block_size)), (1, SECONDARY, _comp(rle2(nanos_enc, signed=False), compression, block_size))]
    cols = [{"column_id": 1, "orc_type": TIMESTAMP, "encoding": DIRECT_V2, "name": "ts"}]
    expect = {}
    if want_expect:
        expect[1] = {"values": ((secs + ORC_EPOCH) * 1_000_000_000 + micros * 1000).tobytes()}
    return n, cols, streams, expect, stats


# ---- C3: dictionary Utf8 + PRESENT -------------------------------------------------------------
def c3_stripe(n, stripe_no, compression="snappy", block_size=262144, want_expect=True):
    words = DICTS["l_shipmode"]
    present = (splitmix64(4 + stripe_no, n) % np.uint64(10) != 0).astype(np.uint8)
    k = int(present.sum())
    keys = (splitmix64(3 + stripe_no, k) % np.uint64(7)).astype(np.int64)
    streams = [(1, PRESENT, _comp(boolean(present), compression, block_size)), (1, DATA, _comp(rle2(keys, signed=False), compression, block_size)),
               (1, LENGTH, _comp(rle2(np.array([len(w) for w in words], dtype=np.int64), signed=False), compression, block_size)),
               (1, DICTIONARY_DATA, _comp(np.frombuffer(b"".join(words), dtype=np.uint8), compression, block_size))]
    cols = [{"column_id": 1, "orc_type": STRING, "encoding": DICTIONARY_V2, "dictionary_size": len(words), "name": "shipmode"}]
    expect = {}
    if want_expect:
        vb, kl = _dict_arrow(words, keys)
        lens = np.zeros(n, dtype=np.int32)
        lens[present.astype(bool)] = kl
        expect[1] = {"values": vb.tobytes(), "lengths": lens, "present": present}
    return n, cols, streams, expect


# ---- C2: RLE v2 DIRECT / DELTA Int64 -------------------------------------------------------------
def c2_stripe(n, stripe_no, kind, row0=0, base=0):
    if kind == "direct":
        vals = (splitmix64(1 + stripe_no, n) & np.uint64((1 << 40) - 1)).astype(np.int64)
    elif kind == "arange":
        vals = np.arange(row0, row0 + n, dtype=np.int64)
    else:
        vals = np.cumsum((splitmix64(2 + stripe_no, n) % np.uint64(255)).astype(np.int64) + 1) + base
    stream, stats = rle2(vals, signed=True, aligned=True, stats=True)
    cols = [{"column_id": 1, "orc_type": LONG, "encoding": DIRECT_V2, "name": "v"}]
    return n, cols, [(1, DATA, stream)], {1: {"values": vals.tobytes()}}, stats


def c2_rowgroup_stripe(n, stripe_no, stride=10000):
    """C2's DIRECT shape as a real writer lays it out with a row index: the RLE encoder is flushed at every row-group boundary
    (`stride` rows), so every 20th run is short (10 000 = 19 x 512 + 272) and the run stride of the stream keeps breaking."""
    vals = (splitmix64(1 + stripe_no, n) & np.uint64((1 << 40) - 1)).astype(np.int64)
    seg = np.full(n // stride + 1, stride, dtype=np.uint32)
    stream, stats = rle2_segments(vals, seg, signed=True, aligned=True, stats=True)
    cols = [{"column_id": 1, "orc_type": LONG, "encoding": DIRECT_V2, "name": "v"}]
    return n, cols, [(1, DATA, stream)], {1: {"values": vals.tobytes()}}, stats


def c2_adversarial_stripe(n, stripe_no):
    """Int64 column built to defeat the stride guesses of the run walk (rle_scan.hip): run lengths drawn from 200..511, the
    width changing from run to run (unaligned widths 3..40 bits), every third run with outliers (PATCHED_BASE), every
    fifth a varying-delta run."""
    seg = (splitmix64(91 + stripe_no, n // 200 + 2) % np.uint64(312)).astype(np.uint32) + 200
    ends = np.cumsum(seg, dtype=np.int64)
    k = int(np.searchsorted(ends, n)) + 1
    seg = seg[:k]
    r = splitmix64(92 + stripe_no, n)
    run_of = np.repeat(np.arange(k, dtype=np.int64), seg)[:n]
    width = (splitmix64(93 + stripe_no, k) % np.uint64(38)).astype(np.int64) + 3
    vals = (r & ((np.uint64(1) << width[run_of].astype(np.uint64)) - np.uint64(1))).astype(np.int64)
    kind = run_of % 15
    # outliers: 12 per run of the "patched" runs, 20 bits above the run's width
    pos_in_run = np.arange(n, dtype=np.int64) - np.concatenate(([0], ends[:k - 1]))[run_of]
    out_mask = (kind % 3 == 1) & (pos_in_run % 37 == 5)
    vals[out_mask] |= np.int64(1) << (width[run_of[out_mask]] + 18)
    # varying deltas: a random walk with small steps
    dl = kind % 5 == 2
    steps = (r % np.uint64(200)).astype(np.int64) - 100
    walk = np.cumsum(np.where(dl, steps, 0))
    vals = np.where(dl, walk + (np.int64(1) << 33), vals)
    stream, stats = rle2_segments(vals, seg, signed=True, aligned=False, stats=True)
    cols = [{"column_id": 1, "orc_type": LONG, "encoding": DIRECT_V2, "name": "v"}]
    return n, cols, [(1, DATA, stream)], {1: {"values": vals.tobytes()}}, stats


def check_result(res, cols, expect, batch_size=8192):
    """Full-size check: every decoded Arrow buffer of the stripe equals what the generated values imply
    (value bytes of all batches concatenated; string offsets of every batch = running sums of the lengths from 0;
    validity = the generated PRESENT bits).  Raises AssertionError with the column on mismatch."""
    st = res.status()
    assert st[0] == 0, ("decode status", st)
    nb = res.n_batches
    for ci, c in enumerate(cols):
        e = expect[c["column_id"]]
        parts = [res.batch(b, ci) for b in range(nb)]
        got = b"".join(p["values"] for p in parts)
        if c["orc_type"] == BOOLEAN:
            raise NotImplementedError
        assert got == e["values"], (c.get("name"), "values differ", len(got), len(e["values"]))
        if "lengths" in e:
            lens = e["lengths"]
            for b, p in enumerate(parts):
                ob = np.zeros(p["length"] + 1, dtype=np.int64)
                np.cumsum(lens[b * batch_size:b * batch_size + p["length"]], out=ob[1:])
                assert np.array_equal(p["offsets"], ob.astype(np.int32)), (c.get("name"), "offsets differ in batch", b)
        if "present" in e:
            pres = e["present"]
            for b, p in enumerate(parts):
                seg = pres[b * batch_size:b * batch_size + p["length"]]
                nulls = int(seg.size - seg.sum())
                assert p["null_count"] == nulls, (c.get("name"), "null count", b)
                if nulls:
                    assert p["validity"] == np.packbits(seg, bitorder="little").tobytes(), (c.get("name"), "validity", b)
                else:
                    assert p["validity"] is None
        else:
            assert all(p["null_count"] == 0 and p["validity"] is None for p in parts), (c.get("name"), "unexpected nulls")
