"""with_predicate on the GPU reader (orcgpu_reader_set_predicate): the reference's integration tests restated
(tests/integration/main.rs:160-262 TestOrcFile.testPredicatePushdown / testWithoutIndex, :374-492 bloom_filter.orc with its
exact row counts), and files of the ORC C++ writer whose expected batches follow from the data itself: per row group the
column's true minimum / maximum / null counts, the reference's keep rules (src/row_group_filter.rs) applied to them in plain
Python, RowSelection::from_row_group_filter, and the stepping model of tests/selection_model.py."""
import datetime

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.orc as orc
import pytest

import arrow_util as A
import selection_model as M
from orc_rust_amd import capi
from orc_rust_amd.arrow_reader import ArrowReaderBuilder
from orc_rust_amd.predicate import Predicate as P, PredicateValue as V

pytestmark = pytest.mark.gpu

_CTX = None


def ctx():
    global _CTX
    if _CTX is None:
        _CTX = capi.Context()
    return _CTX


def read(path, predicate=None, names=None, batch_size=8192, prefetch=0, prune=True, selection=None):
    b = ArrowReaderBuilder.try_new(path, ctx()).with_batch_size(batch_size).with_prefetch(prefetch).with_row_group_pruning(prune)
    if names is not None:
        b = b.with_projection(names)
    if predicate is not None:
        b = b.with_predicate(predicate)
    if selection is not None:
        b = b.with_row_selection(selection)
    r = b.build()
    batches = list(r)
    groups = r.row_groups()
    r.close()
    return batches, groups


def count(path, predicate, **kw):
    return sum(b.num_rows for b in read(path, predicate, **kw)[0])


def test_bloom_filter_fixture_counts():
    """bloom_filter_predicate_prunes (tests/integration/main.rs:374-492): values inside the stripe's min / max that only the
    Bloom filters can rule out -- 0 rows --, values that are there -- all 204."""
    path = A.data_path("bloom_filter.orc")
    assert count(path, None) == 204

    def days(y, m, d):
        return (datetime.date(y, m, d) - datetime.date(1970, 1, 1)).days

    cases = [
        (P.eq("id", V.Int32(2)), 0), (P.eq("id", V.Int32(3)), 204),
        (P.eq("name", V.Utf8("beta")), 0), (P.eq("name", V.Utf8("alpha")), 204),
        (P.eq("score", V.Float64(2.0)), 0), (P.eq("score", V.Float64(1.0)), 204),
        (P.eq("event_date", V.Int32(days(2023, 1, 2))), 0), (P.eq("event_date", V.Int32(days(2023, 1, 1))), 204),
        (P.and_([P.eq("flag", V.Boolean(True)), P.eq("id", V.Int32(2))]), 0),
        (P.eq("data", V.Utf8("\x02")), 0), (P.eq("data", V.Utf8("\x01")), 204),
        (P.eq("dec", V.Utf8("2.22")), 0), (P.eq("dec", V.Utf8("1.11")), 204),
    ]
    for pred, want in cases:
        for prefetch in (0, 2):
            assert count(path, pred, prefetch=prefetch) == want, (pred.op, pred.column, want)
    # the rows that come are the file's
    got = pa.Table.from_batches(read(path, P.eq("id", V.Int32(3)))[0])
    assert got.equals(orc.ORCFile(path).read().cast(got.schema))


def group_facts(column, stripe_rows, stride):
    """Per row group of every stripe: (non-null count, has nulls, min, max) of the column's values there."""
    out, base = [], 0
    for n in stripe_rows:
        groups = []
        for g0 in range(0, n, stride):
            part = column.slice(base + g0, min(stride, n - g0))
            k = len(part) - part.null_count
            mm = pc.min_max(part).as_py() if k else {"min": None, "max": None}
            groups.append((k, part.null_count > 0, mm["min"], mm["max"]))
        out.append(groups)
        base += n
    return out


def keep_int(op, v, facts):
    """evaluate_integer_comparison (row_group_filter.rs:385-410) over true minima / maxima."""
    k, _, lo, hi = facts
    if k == 0:
        raise ValueError("no typed statistics")  # the evaluation fails: every row is read
    return {"eq": lo <= v <= hi, "ne": not (lo == v == hi), "lt": lo < v, "le": lo <= v, "gt": hi > v, "ge": hi >= v}[op]


def expected_rows(table, stripe_rows, stride, keeps, batch_size):
    """keeps: per stripe the row-group filter, or None (evaluation failed: select_all).  from_row_group_filter
    (row_selection.rs:348-392) + the stepping (selection_model) -> the table slices the reader yields."""
    out, base = [], 0
    for n, keep in zip(stripe_rows, keeps):
        sel = [(n, False)] if keep is None else M.normalise([(stride, not k) for k in keep])
        out += [table.slice(base + s, k) for s, k in M.stripe_batches(sel, n, batch_size)]
        base += n
    return out


def check(path, table, predicate, keeps, batch_size, names=None):
    f = orc.ORCFile(path)
    rows = [f.read_stripe(i).num_rows for i in range(f.nstripes)]
    want = expected_rows(table.select(names) if names else table, rows, f.row_index_stride, keeps, batch_size)
    g_read = g_total = 0
    for prefetch, prune in ((0, False), (3, True), (0, True)):
        got, (g_read, g_total) = read(path, predicate, names=names, batch_size=batch_size, prefetch=prefetch, prune=prune)
        assert len(got) == len(want), (len(got), len(want), prefetch, prune)
        for k, (a, w) in enumerate(zip(got, want)):
            assert pa.Table.from_batches([a]).cast(w.schema).equals(w), ("batch", k, prefetch, prune)
    return g_read, g_total


def test_predicate_pushdown_fixture():
    """TestOrcFile.testPredicatePushdown.orc (3500 rows, rowIndexStride 1000) under the predicates of
    tests/integration/main.rs:165-248 -- the reference only checks that they run; here the batches must be those the row-group
    statistics imply."""
    path = A.data_path("TestOrcFile.testPredicatePushdown.orc")
    table = orc.ORCFile(path).read()
    facts = group_facts(table.column("int1"), [3500], 1000)[0]
    for pred, rule in ((P.gt("int1", V.Int32(2000)), lambda g: keep_int("gt", 2000, g)),
                       (P.and_([P.gte("int1", V.Int32(1000)), P.lte("int1", V.Int32(5000))]), lambda g: keep_int("ge", 1000, g) and keep_int("le", 5000, g)),
                       (P.eq("int1", V.Int32(3000)), lambda g: keep_int("eq", 3000, g)),
                       (P.not_(P.or_([P.lt("int1", V.Int32(1500)), P.gt("int1", V.Int32(2500))])), lambda g: keep_int("ge", 1500, g) and keep_int("le", 2500, g)),
                       (P.is_null("int1"), lambda g: g[1]), (P.is_not_null("int1"), lambda g: g[0] > 0)):
        keep = [rule(g) for g in facts]
        for batch_size in (8192, 300):
            check(path, table, pred, [keep], batch_size)
    # a column that is not there, a value of the wrong type: the evaluation fails, every row is read (arrow_reader.rs:282-291)
    assert count(path, P.gt("nope", V.Int32(1))) == 3500
    assert count(path, P.gt("int1", V.Utf8("x"))) == 3500
    # a column outside the projection has no row index among the stripe's columns: the same
    assert count(path, P.gt("int1", V.Int32(3400)), names=["string1"]) == 3500
    # test_predicate_pushdown_without_index (:250-262)
    plain = A.data_path("TestOrcFile.testWithoutIndex.orc")
    assert count(plain, P.gt("int1", V.Int32(1000))) == orc.ORCFile(plain).nrows


def test_predicates_over_written_files(tmp_path):
    """Files of the ORC C++ writer: sorted keys (so that statistics cut), several stripes, nulls, strings, doubles; only the row
    groups the predicate keeps are read."""
    n = 200_000
    rng = np.random.default_rng(5)
    key = np.sort(rng.integers(0, 1_000_000, n)).astype(np.int64)
    table = pa.table({
        "key": pa.array(key),
        "half": pa.array(key // 2, mask=rng.random(n) < 0.2),
        "word": pa.array(["w%07d" % k for k in key]),
        "x": pa.array(key.astype(np.float64) / 7.0),
        "payload": pa.array(rng.integers(0, 1 << 40, n)),
    })
    path = str(tmp_path / "sorted.orc")
    orc.write_table(table, path, compression="zstd", row_index_stride=1000, stripe_size=1 << 20)
    f = orc.ORCFile(path)
    rows = [f.read_stripe(i).num_rows for i in range(f.nstripes)]
    assert len(rows) >= 3
    kf = group_facts(table.column("key"), rows, 1000)
    lo, hi = int(key[n // 3]), int(key[n // 3 + 2500])
    pred = P.and_([P.gte("key", V.Int64(lo)), P.lt("key", V.Int64(hi))])
    keeps = [[keep_int("ge", lo, g) and keep_int("lt", hi, g) for g in groups] for groups in kf]
    g_read, g_total = check(path, table, pred, keeps, 8192)
    assert g_read <= 5 and g_total == sum((r + 999) // 1000 for r in rows), (g_read, g_total)
    # a kept run longer than a batch is read to the stripe's end (the stepping of mod.rs:337-347, selection_model)
    check(path, table, pred, keeps, 500, names=["key", "payload"])
    # strings and doubles
    w = "w%07d" % key[n // 2]
    sf = group_facts(table.column("word"), rows, 1000)
    keeps = [[g[2] <= w <= g[3] for g in groups] for groups in sf]
    g_read, _ = check(path, table, P.eq("word", V.Utf8(w)), keeps, 8192, names=["key", "word"])
    assert g_read <= 2
    xv = float(key[n - 10]) / 7.0
    xf = group_facts(table.column("x"), rows, 1000)
    keeps = [[g[3] > xv for g in groups] for groups in xf]
    check(path, table, P.gt("x", V.Float64(xv)), keeps, 1000, names=["x"])
    # nulls
    hf = group_facts(table.column("half"), rows, 1000)
    check(path, table, P.not_(P.is_null("half")), [[g[0] > 0 for g in groups] for groups in hf], 1000, names=["half"])
    # with a row selection as well: a row is read when both select it
    sel = [(n // 3 + 100, True), (200, False), (n - n // 3 - 300, True)]
    got, _ = read(path, pred, names=["key"], batch_size=1000, selection=sel)
    want = table.column("key").slice(n // 3 + 100, 200)
    assert pa.concat_arrays([b.column(0) for b in got]).equals(want.combine_chunks())


@pytest.mark.parametrize("compression", ["uncompressed", "zstd", "snappy"])
@pytest.mark.parametrize("prefetch", [0, 2])
def test_stripes_kept_whole_take_their_row_index_positions(tmp_path, compression, prefetch):
    """A predicate that keeps every row group: the stripes are decoded whole, as without it -- same batches, no selection pass --,
    and the ROW_INDEX positions read for it go with the wide streams as run starts for the walk (whole_stripe_entries; only a
    hint: checked on the device).  Wide values in runs of irregular length and width, nulls, a stride that splits runs."""
    rng = np.random.default_rng(5)
    n = 230_000
    wide = rng.integers(-(1 << 62), 1 << 62, n)
    steps = np.cumsum(rng.integers(0, 1 << 20, n)) * rng.choice([1, 1, 1, 1 << 20], n)
    mixed = np.where((np.arange(n) // 700) % 3 == 0, wide, steps)
    nulls = rng.random(n) < 0.07
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)), "wide": pa.array(wide), "mixed": pa.array(mixed, mask=nulls),
                  "f": pa.array(rng.random(n))})
    path = str(tmp_path / "wide.orc")
    orc.write_table(t, path, compression=compression, stripe_size=1 << 20, row_index_stride=10000)
    plain, _ = read(path, None, prefetch=prefetch)
    kept, groups = read(path, P.gte("k", V.Int64(0)), prefetch=prefetch)
    assert groups[0] == groups[1] and groups[1] >= 23
    assert [b.num_rows for b in kept] == [b.num_rows for b in plain]
    assert pa.Table.from_batches(kept).equals(pa.Table.from_batches(plain))
    want = orc.ORCFile(path).read()
    assert pa.Table.from_batches(kept).cast(want.schema).equals(want)  # (the cast: the reader marks columns without nulls "not null")
