"""Row selection at the seam (orcgpu_result_select / orcgpu_reader_set_row_selection) against the reference's stepping
(tests/selection_model.py = array_decoder/mod.rs:302-365) applied to the CPU oracle's decode of the same streams: every
selected batch must hold exactly the rows the reference's decoders would have produced for that step.  Patterns follow the
reference's tests/row_selection/main.rs, plus ranges that cross the decoder's internal batch boundaries."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_util as A
import gpu_util as G
import oracle_lib as O
import selection_model as M
from orc_rust_amd import capi, gen

pytestmark = pytest.mark.gpu
S, K = (lambda n: (n, False)), (lambda n: (n, True))
LONG, STRING, BOOLEAN, DECIMAL, DATE = 4, 7, 0, 14, 15
PRESENT, DATA, LENGTH, DICT, SECONDARY = 0, 1, 2, 3, 5


def make_stripe(n, seed, nulls=True):
    rng = np.random.default_rng(seed)
    cols, streams = [], []

    def add(typ, mk, enc=2, **kw):
        cid = len(cols) + 1
        present = (rng.random(n) >= (0.2 if nulls else 0.0)).astype(np.uint8)
        k = int(present.sum())
        cols.append(dict(column_id=cid, orc_type=typ, encoding=enc, **kw))
        if nulls:
            streams.append((cid, PRESENT, gen.boolean(present)))
        for kind, data in mk(k):
            streams.append((cid, kind, data))
    words = [b"", b"a", b"bc", b"\xc3\xa9t\xc3\xa9", b"longer value here"]
    add(LONG, lambda k: [(DATA, gen.rle2(rng.integers(-10**9, 10**9, k), signed=True))])
    def direct(k):
        idx = rng.integers(0, 5, k)
        return [(LENGTH, gen.rle2(np.array([len(words[i]) for i in idx], dtype=np.int64), signed=False)),
                (DATA, np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8))]
    add(STRING, direct)
    add(STRING, lambda k: [(DATA, gen.rle2(rng.integers(0, 5, k), signed=False)), (LENGTH, gen.rle2(np.array([len(w) for w in words], dtype=np.int64), signed=False)),
                           (DICT, np.frombuffer(b"".join(words), dtype=np.uint8))], enc=3, dictionary_size=5)
    add(BOOLEAN, lambda k: [(DATA, gen.boolean(rng.integers(0, 2, k).astype(np.uint8)))], enc=0)
    add(DECIMAL, lambda k: [(DATA, gen.varint64(rng.integers(-10**12, 10**12, k))), (SECONDARY, gen.rle2(np.full(k, 3), signed=True))], precision=18, scale=3)
    add(DATE, lambda k: [(DATA, gen.rle2(rng.integers(0, 20000, k), signed=True))])
    return cols, streams


def oracle_column_array(col, streams, n):
    oc = G.oracle_column(col, streams)
    b = oc.next_batch(n)
    assert b["status"] == O.OK
    arr = A.to_arrow(col["orc_type"], b, col.get("precision", 0), col.get("scale", 0))
    oc.close()
    return arr


def check_selection(n, cols, streams, selectors, batch_size=8192, compression="none"):
    res = G.gpu_decode(n, cols, streams, compression=compression, batch_size=batch_size)
    assert res.status()[0] == 0
    res.select(selectors)
    want = M.stripe_batches(M.normalise(selectors), n, batch_size)
    assert res.n_batches == len(want), (selectors, res.n_batches, len(want))
    full = [oracle_column_array(c, [(cid, k, b) for cid, k, b in streams], n) for c in cols] if compression == "none" else None
    if full is None:
        plain = [(cid, k, np.frombuffer(O.stream_decompress(bytes(b), compression)[1], dtype=np.uint8)) for cid, k, b in streams]
        full = [oracle_column_array(c, plain, n) for c in cols]
    for bi, (start, ln) in enumerate(want):
        for ci, c in enumerate(cols):
            g = res.batch(bi, ci)
            assert g["length"] == ln
            got = A.to_arrow(c["orc_type"], g, c.get("precision", 0), c.get("scale", 0))
            exp = full[ci].slice(start, ln)
            assert got.equals(exp), (selectors, "batch", bi, "rows", start, ln, "column", ci)
            # per batch semantics of the reference: no validity buffer without nulls (mod.rs:247-251), offsets restart at 0
            assert (g["validity"] is None) == (exp.null_count == 0)
            if g["offsets"] is not None:
                assert g["offsets"][0] == 0
    # the Arrow C Data export speaks in selected batches too
    for bi, (start, ln) in enumerate(want[:3]):
        rb = res.export_batch(bi)
        assert rb.num_rows == ln
        for ci in range(len(cols)):
            assert rb.column(ci).equals(full[ci].slice(start, ln).cast(rb.column(ci).type)), (selectors, bi, ci)
    res.free()


def test_reference_patterns_small_stripe():  # tests/row_selection/main.rs:47-196 on a 5-row stripe
    cols, streams = make_stripe(5, 1)
    for sel in ([K(2), S(2), K(1)], [S(5)], [K(5)], [S(1), K(4)], [K(4), S(1)], [S(1), K(1), S(1), K(1), S(1)], []):
        check_selection(5, cols, streams, sel)


def test_large_file_pattern_and_batch_crossings():  # main.rs:306-329, and ranges that cross the internal 8192-row batches
    n = 30000
    cols, streams = make_stripe(n, 2)
    check_selection(n, cols, streams, [K(1000), S(500), K(8500)])
    check_selection(n, cols, streams, [K(8000), S(500), K(7000), S(1500), K(1), S(8192), K(100), S(3)])
    check_selection(n, cols, streams, [S(3), K(29990), S(100)])  # the last run asks for more rows than are left
    check_selection(n, cols, streams, [K(10), S(20000), K(70000)])  # select run longer than a batch (see selection_model)
    check_selection(n, cols, streams, [K(100), S(50), K(200), S(700)], batch_size=256)


def test_without_nulls_and_with_compression():  # main.rs:351-372
    cols, streams = make_stripe(12000, 3, nulls=False)
    check_selection(12000, cols, streams, [K(10), S(20), K(34), S(9000)])
    cols, streams = make_stripe(64, 4)
    comp = [(cid, k, gen.compress_stream(b, "zstd", 65536)) for cid, k, b in streams]
    check_selection(64, cols, comp, [K(10), S(20), K(34)], compression="zstd")


def test_reader_row_selection_over_stripes():
    """with_row_selection on a multi-stripe file: every stripe takes its share (split_off), the rest is read whole once the
    selection is used up (arrow_reader.rs:296-308).  Expectation: the committed PyArrow decode of the file, sliced."""
    import orcfile
    from orc_rust_amd.arrow_reader import ArrowReaderBuilder
    name = "TestOrcFile.testSeek.orc"  # several stripes
    f = orcfile.OrcFile(A.data_path(name))
    stripe_rows = [s.number_of_rows for s in f.stripes]
    assert len(stripe_rows) > 1
    expected = A.expected_table(name[:-4])
    total = sum(stripe_rows)
    sel = [K(100), S(50), K(stripe_rows[0]), S(3000), K(total)]
    flat = [c for c, _, t in f.flat_columns() if t.kind not in (9,)]
    reader = ArrowReaderBuilder.try_new(A.data_path(name), ctx=G.ctx()).with_projection(flat).with_row_selection(sel).build()
    got = list(reader)
    per_stripe = M.file_batches(sel, stripe_rows, 8192)
    want_ranges, base = [], 0
    for n, b in zip(stripe_rows, per_stripe):
        if b is None:
            b = [(s, min(8192, n - s)) for s in range(0, n, 8192)]
        want_ranges += [(base + s, ln) for s, ln in b]
        base += n
    assert [rb.num_rows for rb in got] == [ln for _, ln in want_ranges]
    for rb, (start, ln) in zip(got, want_ranges):
        for cname in flat:
            w = expected.column(cname).slice(start, ln).combine_chunks()
            gcol = rb.column(rb.schema.get_field_index(cname))
            assert gcol.equals(w.cast(gcol.type)), (cname, start, ln)
