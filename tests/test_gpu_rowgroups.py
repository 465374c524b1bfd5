"""Row groups under a row selection (orcgpu_reader_set_row_group_pruning): the reader reads a stripe's ROW_INDEX streams
(row_index.rs:204-226) and reads, stages and decodes only the row groups that hold selected rows, every stream from the
entry point its index names (orcgpu_stream::skip_bytes / skip_values).  The reference decodes the stripe from its first row
and discards (skip_values, rle_v2/mod.rs:148-175): the RecordBatches must be the same.  Expectations: the same reader with
pruning off (whole-stripe decode, itself pinned by test_gpu_selection / test_gpu_reader), and independently the table the
ORC C++ writer was given, sliced by the model of the reference's stepping (tests/selection_model.py)."""
import decimal
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.orc as orc
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import selection_model as M
from orc_rust_amd import capi
from orc_rust_amd.arrow_reader import ArrowReaderBuilder

pytestmark = pytest.mark.gpu

if not os.path.isdir("/usr/share/zoneinfo") and "TZDIR" not in os.environ:
    # the ORC C++ writer looks its time zone ("GMT") up in the tz database: slim images only have Python's tzdata package
    try:
        import tzdata
        os.environ["TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
    except ImportError:
        pass

_CTX = None


def ctx():
    global _CTX
    if _CTX is None:
        _CTX = capi.Context()
    return _CTX


def K(n):
    return (n, False)


def S(n):
    return (n, True)


def make_table(n, seed, nulls=True):
    """Every flat kind the index has positions for: run-length integers (long runs, short runs, patched values), byte and
    bit streams, raw floats, direct and dictionary strings, binary, decimal, timestamp, date; with and without nulls."""
    rng = np.random.default_rng(seed)

    def mask(p):
        return rng.random(n) < p if nulls else None

    seq = np.cumsum(rng.integers(1, 50, n)).astype(np.int64)
    big = rng.integers(-(1 << 40), 1 << 40, n)
    rep = np.repeat(rng.integers(0, 1000, n // 7 + 1), 7)[:n].astype(np.int32)
    small = rng.integers(0, 200, n).astype(np.int64)
    small[rng.integers(0, n, n // 40)] = 1 << 33  # patched base
    words = np.array(["w%05d" % i for i in range(5000)])
    cols = {
        "seq": pa.array(seq, mask=mask(0.02)),
        "big": pa.array(big, mask=mask(0.3)),
        "rep": pa.array(rep),
        "patched": pa.array(small, mask=mask(0.01)),
        "i16": pa.array(rng.integers(-30000, 30000, n).astype(np.int16), mask=mask(0.1)),
        "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8)),
        "flag": pa.array(rng.random(n) < 0.3),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=mask(0.05)),
        "f64": pa.array(rng.standard_normal(n), mask=mask(0.5)),
        "direct": pa.array(["row %d %s" % (i, "x" * int(k)) for i, k in enumerate(rng.integers(0, 20, n))], mask=mask(0.1)),
        "dict": pa.array(words[rng.integers(0, 60, n)], mask=mask(0.2)),
        "mode": pa.array(np.array(["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"])[rng.integers(0, 7, n)]),
        "bin": pa.array([bytes(rng.integers(0, 256, int(k)).astype(np.uint8)) for k in rng.integers(0, 9, n)], type=pa.binary(), mask=mask(0.1)),
        "dec": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, n)], type=pa.decimal128(15, 2), mask=mask(0.1)),
        "ts": pa.array(rng.integers(0, 2 * 10**18, n), type=pa.timestamp("ns"), mask=mask(0.1)),
        "day": pa.array(rng.integers(0, 20000, n).astype(np.int32), type=pa.date32(), mask=mask(0.1)),
    }
    return pa.table(cols)


def write(tmp_path, table, name, **kw):
    path = str(tmp_path / name)
    orc.write_table(table, path, **kw)
    return path


def read(path, names, selection, prune, batch_size=1000, prefetch=0):
    r = (ArrowReaderBuilder.try_new(path, ctx()).with_projection(names).with_batch_size(batch_size).with_prefetch(prefetch)
         .with_row_selection(selection).with_row_group_pruning(prune).build())
    batches = list(r)
    groups = r.row_groups()
    r.close()
    return batches, groups


def stripe_rows(path):
    f = orc.ORCFile(path)
    return [f.read_stripe(i).num_rows for i in range(f.nstripes)]


def expected_batches(table, path, selection, batch_size):
    """The table sliced as the reference's stepping would yield it (selection_model)."""
    out, base = [], 0
    for n, ranges in zip(stripe_rows(path), M.file_batches(selection, stripe_rows(path), batch_size)):
        if ranges is None:
            ranges = [(s, min(batch_size, n - s)) for s in range(0, n, batch_size)]
        out += [table.slice(base + s, k) for s, k in ranges]
        base += n
    return out


def check(table, path, selection, batch_size=1000, prefetch=0, expect_pruned=True, names=None):
    names = names or table.schema.names
    pruned, (g_read, g_total) = read(path, names, selection, True, batch_size, prefetch)
    whole, (w_read, w_total) = read(path, names, selection, False, batch_size, prefetch)
    assert len(pruned) == len(whole), (len(pruned), len(whole))
    for k, (a, b) in enumerate(zip(pruned, whole)):
        assert a.schema.equals(b.schema) and a.num_rows == b.num_rows, k
        assert a.equals(b), ("batch", k, [n for n in names if not a.column(n).equals(b.column(n))])
    want = expected_batches(table.select(names), path, selection, batch_size)
    assert len(want) == len(pruned)
    for k, (a, w) in enumerate(zip(pruned, want)):
        got = pa.Table.from_batches([a])
        assert got.cast(w.schema).equals(w), ("batch", k)
    assert w_read == w_total
    if expect_pruned:
        assert g_read < g_total, (g_read, g_total)
    return g_read, g_total


SELECTIONS = [
    [S(12_345), K(10)],                                        # a few rows in one row group
    [S(999), K(2), S(20_000), K(1500), S(3), K(1)],            # runs that cross row-group boundaries
    [K(1), S(30_000), K(1), S(30_000), K(1)],                  # single rows far apart: three pieces of one stripe
    [S(41_000), K(900), S(100), K(900)],                       # neighbouring groups share a piece
    [S(5_000), K(2_500)],                                      # a select run longer than a batch: read to the stripe's end (selection_model)
]


@pytest.mark.parametrize("comp,block", [("uncompressed", 65536), ("zstd", 65536), ("zstd", 131072), ("snappy", 65536), ("zlib", 65536), ("lz4", 65536)])
def test_pruned_reads_match_whole_reads(tmp_path, comp, block):
    """One stripe of 70 row groups (stride 1000); 64 KiB compression blocks (the smallest the ORC C++ writer takes) put most
    entry points in the middle of chunks and make runs straddle them."""
    n = 70_000
    table = make_table(n, seed=len(comp) + block)
    path = write(tmp_path, table, "t.orc", compression=comp, compression_block_size=block, row_index_stride=1000, stripe_size=64 << 20,
                 dictionary_key_size_threshold=0.5)
    assert len(stripe_rows(path)) == 1
    for sel in SELECTIONS:
        check(table, path, sel)


def test_pruning_over_stripes_and_read_ahead(tmp_path):
    """Several stripes; the selection is split over them, stripes without selected rows are not read, stripes behind the
    selection's end are read whole (arrow_reader.rs:296-308); serial and read-ahead readers alike."""
    n = 120_000
    table = make_table(n, seed=7, nulls=False)
    path = write(tmp_path, table, "m.orc", compression="zstd", compression_block_size=65536, row_index_stride=2000, stripe_size=1 << 20,
                 dictionary_key_size_threshold=0.5)
    rows = stripe_rows(path)
    assert len(rows) >= 4, rows
    sel = [S(rows[0] + 4321), K(300), S(rows[1] + rows[2] - 4621 + 10), K(7)]  # nothing in stripe 0, a run in 1, none in 2, a few rows in 3
    sel_total = sum(k for k, _ in sel)
    for prefetch in (0, 3):
        g_read, g_total = check(table, path, sel + [S(sum(rows) - sel_total)], batch_size=400, prefetch=prefetch)
        assert g_read <= 4, (g_read, g_total)
    # a selection that ends early: the stripes behind it come whole
    check(table, path, [S(100), K(50), S(rows[0] - 150)], batch_size=4096, expect_pruned=True)


def test_one_percent_of_the_row_groups(tmp_path):
    """a selection of 1 % of a stripe's row groups reads about that share of it."""
    n = 400_000
    rng = np.random.default_rng(3)
    table = pa.table({"k": pa.array(np.cumsum(rng.integers(1, 9, n)).astype(np.int64)), "v": pa.array(rng.integers(0, 1 << 45, n)),
                      "s": pa.array(np.array(["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"])[rng.integers(0, 7, n)]),
                      "c": pa.array(["comment %d" % i for i in range(n)])})
    path = write(tmp_path, table, "p.orc", compression="zstd", row_index_stride=1000, stripe_size=256 << 20)
    assert len(stripe_rows(path)) == 1
    sel = [S(37_100), K(800), S(200_000), K(900), S(100_000), K(1000), S(n - 37_100 - 800 - 200_000 - 900 - 100_000 - 1000)]
    g_read, g_total = check(table, path, sel, batch_size=8192)
    assert g_total == 400 and g_read <= 5, (g_read, g_total)


def test_structs_and_unusable_indexes(tmp_path):
    """Fields of Structs take their entry points like root columns; bit streams entered in mid-byte (Boolean values behind nulls,
    the PRESENT stream of a field of a Struct with nulls: orcgpu_stream::skip_bits, round 5) are pruned like the others (a file
    without indexes: test_reference_fixtures_with_and_without_index)."""
    n = 30_000
    rng = np.random.default_rng(11)
    inner = pa.StructArray.from_arrays([pa.array(rng.integers(0, 1 << 30, n)), pa.array(["s%d" % (i % 97) for i in range(n)])], names=["a", "b"])
    table = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "st": inner})
    path = write(tmp_path, table, "s.orc", compression="zlib", compression_block_size=65536, row_index_stride=1000)
    sel = [S(17_500), K(600), S(n - 18_100)]
    check(table, path, sel)
    # Boolean values with nulls: the DATA bit stream is entered in mid-byte at most row groups
    flags = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "f": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.37)})
    path = write(tmp_path, flags, "b.orc", compression="uncompressed", row_index_stride=1000)
    check(flags, path, sel)
    check(flags, path, [S(12_345), K(3), S(9_000), K(2_000), S(n - 12_345 - 3 - 9_000 - 2_000)], batch_size=333)
    check(flags, path, sel, names=["id"])
    # a Struct with nulls: its fields' PRESENT streams count the Struct's non-null rows
    st = pa.StructArray.from_arrays([pa.array(rng.integers(0, 99, n), mask=rng.random(n) < 0.2)], names=["a"], mask=pa.array(rng.random(n) < 0.3))
    nul = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "st": st})
    path = write(tmp_path, nul, "n.orc", compression="snappy", row_index_stride=1000)
    check(nul, path, sel)
    # ... two levels of Structs with nulls, a Boolean field with nulls inside
    inner2 = pa.StructArray.from_arrays([pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.4), pa.array(rng.integers(0, 9, n), mask=rng.random(n) < 0.1)],
                                        names=["f", "v"], mask=pa.array(rng.random(n) < 0.25))
    outer = pa.StructArray.from_arrays([inner2, pa.array(["s%d" % (i % 13) for i in range(n)], mask=rng.random(n) < 0.3)], names=["in", "s"],
                                       mask=pa.array(rng.random(n) < 0.2))
    deep = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "o": outer})
    for comp in ("uncompressed", "zstd"):
        path = write(tmp_path, deep, "d_%s.orc" % comp, compression=comp, compression_block_size=65536, row_index_stride=1000)
        check(deep, path, sel)
        check(deep, path, [S(999), K(2), S(20_000), K(1500), S(3), K(1), S(n - 999 - 2 - 20_000 - 1500 - 3 - 1)], batch_size=512)


def test_lists_and_maps_are_pruned_like_flat_columns(tmp_path):
    """Stripes with List / Map columns (round 5): the LENGTH stream is entered at the row group's position like any RLE stream, the
    elements' streams at the positions their own index entries hold for the same row group (list.rs:89, map.rs:106 step a
    selection through the nested decoders); how many elements a piece holds follows from its lengths."""
    n = 60_000
    rng = np.random.default_rng(23)
    def lists(values, max_len, null_frac):
        lens = rng.integers(0, max_len + 1, n)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        return pa.ListArray.from_arrays(pa.array(offs), values(int(offs[-1])), mask=pa.array(rng.random(n) < null_frac) if null_frac else None)
    ints = lists(lambda m: pa.array(rng.integers(-2**40, 2**40, m), mask=rng.random(m) < 0.1), 6, 0.15)
    strs = lists(lambda m: pa.array(["e%d" % (i % 311) for i in range(m)]), 4, 0.0)
    nested = lists(lambda m: pa.ListArray.from_arrays(pa.array(np.arange(m + 1, dtype=np.int32) * 2), pa.array(rng.integers(0, 9, 2 * m).astype(np.int16))), 3, 0.05)
    mlens = rng.integers(0, 4, n)
    moffs = np.concatenate([[0], np.cumsum(mlens)]).astype(np.int32)
    m = int(moffs[-1])
    maps = pa.MapArray.from_arrays(pa.array(moffs), pa.array(["k%d" % (i % 17) for i in range(m)]), pa.array(rng.integers(0, 1 << 20, m), mask=rng.random(m) < 0.2))
    table = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "ints": ints, "strs": strs, "nested": nested, "maps": maps})
    sel = [S(17_500), K(600), S(20_000), K(1), S(999), K(1200), S(n - 17_500 - 600 - 20_000 - 1 - 999 - 1200)]
    for comp, block in (("uncompressed", 65536), ("zstd", 65536), ("snappy", 65536)):
        path = write(tmp_path, table, "l_%s.orc" % comp, compression=comp, compression_block_size=block, row_index_stride=1000, stripe_size=64 << 20)
        assert len(stripe_rows(path)) == 1
        # (elements with nulls -- `ints`, the values of `maps` --: their PRESENT streams are entered in mid-byte at most row groups)
        g_read, g_total = check(table, path, sel, batch_size=700)
        assert g_read < g_total == 60, (g_read, g_total)
        g_read, g_total = check(table, path, sel, batch_size=700, names=["id", "strs", "nested"])
        assert g_read < g_total == 60, (g_read, g_total)
        g_read, g_total = check(table, path, [S(59_990), K(10)], names=["nested"])
        assert g_read == 1, (g_read, g_total)
    # a Map without null values, Lists with null rows but whole elements
    maps2 = pa.MapArray.from_arrays(pa.array(moffs), pa.array(["k%d" % (i % 17) for i in range(m)]), pa.array(rng.integers(0, 1 << 20, m)))
    ints2 = lists(lambda k: pa.array(rng.integers(-2**40, 2**40, k)), 6, 0.15)
    t2 = pa.table({"id": table["id"], "maps": maps2, "ints": ints2})
    path = write(tmp_path, t2, "m2.orc", compression="zstd", compression_block_size=65536, row_index_stride=1000, stripe_size=64 << 20)
    g_read, g_total = check(t2, path, sel, batch_size=700)
    assert g_read < g_total, (g_read, g_total)


def test_unions_are_pruned_like_structs(tmp_path):
    """Stripes with Union columns (round 6; union.rs:69-136): the tag stream is entered at the row group's position like any byte-RLE
    stream, every arm's child at the positions the child's own index entries hold for the same row group (the writer records how
    far each column's streams have got at every boundary: for an arm's child that is the rows whose tag named it so far); how many
    values of an arm a piece holds follows from the piece's tags."""
    n = 40_000
    rng = np.random.default_rng(31)
    ids = rng.integers(0, 3, n).astype(np.int8)
    arms = [pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.1),
            pa.array(["u%d" % (i % 211) for i in range(n)], mask=rng.random(n) < 0.2),
            pa.array(rng.random(n) < 0.5)]
    un = pa.UnionArray.from_sparse(pa.array(ids), arms, ["l", "s", "b"])
    table = pa.table({"id": pa.array(np.arange(n, dtype=np.int64)), "un": un, "tail": pa.array(["t%d" % (i % 7) for i in range(n)])})
    sel = [S(17_500), K(600), S(10_000), K(1), S(999), K(1200), S(n - 17_500 - 600 - 10_000 - 1 - 999 - 1200)]
    for comp in ("uncompressed", "zstd", "snappy"):
        path = write(tmp_path, table, "u_%s.orc" % comp, compression=comp, compression_block_size=65536, row_index_stride=1000, stripe_size=64 << 20)
        assert len(stripe_rows(path)) == 1
        pruned, (g_read, g_total) = read(path, table.schema.names, sel, True, 700)
        whole, _ = read(path, table.schema.names, sel, False, 700)
        assert g_read < g_total == 40, (g_read, g_total)
        assert len(pruned) == len(whole)
        for k, (a, b) in enumerate(zip(pruned, whole)):
            assert a.num_rows == b.num_rows and a.equals(b), ("batch", k)
        # ... and against the table the writer was given, row by row (an arm's slots under another arm's tag are nulls here, whatever
        # the writer's arrays held there: compared as Python values)
        want = expected_batches(table, path, sel, 700)
        assert len(want) == len(pruned)
        for a, w in zip(pruned, want):
            assert a.column("id").to_pylist() == w.column("id").to_pylist()
            assert a.column("un").to_pylist() == w.column("un").to_pylist()
            assert a.column("tail").to_pylist() == w.column("tail").to_pylist()
        # a few rows of the last row group; the Union alone
        pruned, (g_read, g_total) = read(path, ["un"], [S(n - 10), K(10)], True, 1000)
        assert g_read == 1 and pruned[0].column("un").to_pylist() == table.column("un").slice(n - 10, 10).to_pylist()


def test_reference_fixtures_with_and_without_index():
    """TestOrcFile.testSeek.orc / testWithoutIndex.orc (the reference's fixtures): selections over their flat columns."""
    import arrow_util as A
    for name in ("TestOrcFile.testSeek.orc", "TestOrcFile.testWithoutIndex.orc"):
        path = A.data_path(name)
        f = orc.ORCFile(path)
        flat = [fld.name for fld in f.schema if not pa.types.is_nested(fld.type)]
        total = f.nrows
        sel = [S(1234), K(77), S(total // 2), K(300), S(total - 1234 - 77 - total // 2 - 300)]
        a, (g_read, g_total) = read(path, flat, sel, True, batch_size=1000)
        b, _ = read(path, flat, sel, False, batch_size=1000)
        assert len(a) == len(b) and all(x.equals(y) for x, y in zip(a, b)), name
        assert sum(x.num_rows for x in a) == 377
        if "WithoutIndex" not in name:
            assert g_read < g_total, (name, g_read, g_total)


def test_entry_points_at_the_stripe_boundary():
    """orcgpu_stream::skip_bytes / skip_values by hand: an RLE v2 stream entered in the middle of a run, an uncompressed and a
    compressed one."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gpu_util as G
    from orc_rust_amd import gen
    vals = (np.arange(3000, dtype=np.int64) * 37) % 1001 - 500
    stream = gen.rle2(vals, signed=True)
    # run boundaries of the encoder: 512-value runs.  Enter at value 1300 = run 2 (starts at value 1024), 276 values in
    runs = []
    pos, v = 0, 0
    raw = bytes(stream)
    while pos < len(raw):
        h = raw[pos]
        assert h >> 6 == 1, "the test expects DIRECT runs"
        width = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 26, 28, 30, 32, 40, 48, 56, 64][(h >> 1) & 31]
        length = (((h & 1) << 8) | raw[pos + 1]) + 1
        runs.append((pos, v, length))
        pos += 2 + (width * length + 7) // 8
        v += length
    pos2, v2, _ = runs[2]
    cols = [{"column_id": 1, "orc_type": 4, "encoding": 2}]
    res = G.gpu_decode(3000 - 1300, cols, [(1, 1, raw[pos2:], 0, 1300 - v2)], batch_size=500)
    assert res.status()[0] == 0
    got = np.concatenate([np.frombuffer(res.batch(b, 0)["values"], dtype=np.int64) for b in range(res.n_batches)])
    assert np.array_equal(got, vals[1300:])
    res.free()
    # compressed: the chunk holds the whole stream; the entry point is {chunk 0, pos2 bytes in, 276 values}
    comp = gen.compress_stream(raw, "zstd", 262144)
    res = G.gpu_decode(3000 - 1300, cols, [(1, 1, comp, pos2, 1300 - v2)], compression="zstd", batch_size=500)
    assert res.status()[0] == 0
    got = np.concatenate([np.frombuffer(res.batch(b, 0)["values"], dtype=np.int64) for b in range(res.n_batches)])
    assert np.array_equal(got, vals[1300:])
    res.free()
    # an entry point no index can name is refused
    with pytest.raises(capi.OrcGpuError):
        G.gpu_decode(10, cols, [(1, 1, raw, 0, 5000)])


def _chunk_positions(comp, plain_positions, block_size):
    """(byte, values) of the plain stream -> (chunk header offset, bytes into the chunk, values) of the framed one."""
    raw = bytes(comp)
    heads, at = [], 0
    while at < len(raw):
        heads.append(at)
        at += 3 + ((raw[at] | (raw[at + 1] << 8) | (raw[at + 2] << 16)) >> 1)
    return np.array([[heads[int(b) // block_size], int(b) % block_size, int(v)] for b, v in plain_positions], dtype=np.uint64)


def test_verified_run_starts():
    """orcgpu_stream::entries: the ROW_INDEX positions of a run-length stream as verified run starts of the walk.  Streams
    whose runs change length and width all the time (the walk's worst case without them) and streams flushed at every row
    group, uncompressed and compressed; entries that do not lie on the run chain are ignored.  The values are the same."""
    import gpu_util as G
    from orc_rust_amd import gen
    from orc_rust_amd.gen import workloads as W
    for make in (W.c2_adversarial_stripe, W.c2_rowgroup_stripe):
        n, cols, streams, expect, _ = make(600_000, 3, index=True)
        plain = [s[:3] for s in streams]
        res = G.gpu_decode(n, cols, streams)
        W.check_result(res, cols, expect)
        G.assert_column_parity(res, 0, cols[0], plain, n, 8192, what=(make.__name__, "entries"))
        res.free()
        cid, kind, stream, _, _, pos = streams[0]
        for comp, block in (("zstd", 65536), ("snappy", 4096), ("zlib", 262144)):
            framed = gen.compress_stream(stream, comp, block)
            res = G.gpu_decode(n, cols, [(cid, kind, framed, 0, 0, _chunk_positions(framed, pos, block))], compression=comp, block_size=block)
            W.check_result(res, cols, expect)
            res.free()
        # off the chain (one byte late), out of order, beyond the stream: hints only
        late = pos.copy()
        late[:, 0] += 1
        for bad in (late, pos[::-1].copy(), pos + np.uint64(len(bytes(stream)))):
            res = G.gpu_decode(n, cols, [(cid, kind, stream, 0, 0, bad)])
            W.check_result(res, cols, expect)
            res.free()


@pytest.mark.parametrize("compression,block", [("none", 262144), ("zstd", 8192), ("snappy", 1024), ("lz4", 4096), ("zlib", 65536)])
def test_entry_points_into_every_kind_of_run(compression, block):
    """A signed RLE v2 stream of every sub-encoding (SHORT_REPEAT, DIRECT, PATCHED_BASE, fixed and varying DELTA, long and short
    runs) entered at the positions a writer would record for row groups of 1000, 777 and 10 000 values: the column decoded from
    an entry point on == the tail of the column decoded whole."""
    import gpu_util as G
    from orc_rust_amd import gen
    rng = np.random.default_rng(17)
    parts = []
    for k in range(60):
        kind = k % 6
        m = int(rng.integers(3, 1500))
        if kind == 0:
            parts.append(np.full(m % 11 + 3, rng.integers(-1000, 1000)))
        elif kind == 1:
            parts.append(rng.integers(-(1 << int(rng.integers(2, 50))), 1 << int(rng.integers(2, 50)), m))
        elif kind == 2:
            v = rng.integers(0, 100, m)
            v[rng.integers(0, m, max(1, m // 40))] = 1 << 40
            parts.append(v)
        elif kind == 3:
            parts.append(np.arange(m) * int(rng.integers(-9, 9)) + int(rng.integers(-10**6, 10**6)))
        elif kind == 4:
            parts.append(np.cumsum(rng.integers(1, 300, m)))
        else:
            parts.append(np.repeat(rng.integers(0, 5, m // 4 + 1), 4)[:m])
    vals = np.concatenate(parts).astype(np.int64)
    n = len(vals)
    cols = [{"column_id": 1, "orc_type": 4, "encoding": 2}]
    for stride in (1000, 777, 10000):
        stream, pos = gen.rle2_indexed(vals, stride, signed=True)
        raw = bytes(stream)
        if compression == "none":
            framed, cpos = raw, None
        else:
            framed = bytes(gen.compress_stream(raw, compression, block))
            cpos = _chunk_positions(framed, pos, block)
        for g in sorted(set([1, len(pos) // 3, len(pos) // 2, len(pos) - 1])):
            if g <= 0 or g >= len(pos):
                continue
            if cpos is None:
                piece = (1, 1, raw[int(pos[g, 0]):], 0, int(pos[g, 1]))
            else:
                piece = (1, 1, framed[int(cpos[g, 0]):], int(cpos[g, 1]), int(cpos[g, 2]))
            rows = n - g * stride
            res = G.gpu_decode(rows, cols, [piece], compression=compression, block_size=block, batch_size=1000)
            assert res.status()[0] == 0, (stride, g, res.status())
            got = np.concatenate([np.frombuffer(res.batch(b, 0)["values"], dtype=np.int64) for b in range(res.n_batches)])
            assert np.array_equal(got, vals[g * stride:]), (compression, stride, g)
            res.free()


def test_damaged_index_streams_do_not_bring_the_reader_down(tmp_path):
    """The index section of every stripe with random bytes flipped (ROW_INDEX protobufs, their chunk framing, their compressed
    bytes): under a selection and under a predicate the reader either reads (possibly other rows: the positions are what they
    are), falls back to the whole decode, or reports an error -- it does not crash or run away."""
    from orc_rust_amd.predicate import Predicate as P, PredicateValue as V
    n = 60_000
    table = make_table(n, seed=21)
    for comp in ("zstd", "uncompressed"):
        path = write(tmp_path, table, "d_%s.orc" % comp, compression=comp, compression_block_size=65536, row_index_stride=1000, stripe_size=1 << 20)
        good = open(path, "rb").read()
        rng = np.random.default_rng(99)
        sel = [S(12_345), K(10), S(20_000), K(900), S(n - 12_345 - 10 - 20_000 - 900)]
        pred = P.and_([P.gte("seq", V.Int64(1000)), P.lt("seq", V.Int64(50_000))])
        outcomes = {"ok": 0, "error": 0}
        for trial in range(40):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 12))):
                # half of them in the first stripe's index section (it starts behind the "ORC" magic), the rest anywhere in the stripes
                # (index sections and data alike); the file's tail (footers) is left alone
                p = int(rng.integers(3, 12_000)) if rng.random() < 0.5 else int(rng.integers(3, max(4, len(bad) * 7 // 10)))
                bad[p] ^= int(rng.integers(1, 256))
            for kind in ("selection", "predicate"):
                try:
                    b = ArrowReaderBuilder.try_new(bytes(bad), ctx()).with_batch_size(1000).with_prefetch(int(trial % 2) * 2)
                    b = b.with_row_selection(sel) if kind == "selection" else b.with_predicate(pred)
                    r = b.build()
                    rows = sum(x.num_rows for x in r)
                    r.close()
                    assert rows <= n
                    outcomes["ok"] += 1
                except capi.OrcGpuError:
                    outcomes["error"] += 1
        assert outcomes["ok"] + outcomes["error"] == 80
    # the context still decodes
    check(table, path, [S(100), K(50), S(n - 150)], expect_pruned=True)


@pytest.mark.parametrize("compression", ["none", "snappy", "zstd"])
def test_verified_run_starts_of_a_column_with_nulls(compression):
    """C3's shape (dictionary keys in short runs, PRESENT with 10 % nulls) with the positions of both run-length streams: the
    DATA stream's row groups start at the count of non-null rows before them, the PRESENT stream's at whole bytes."""
    import gpu_util as G
    from orc_rust_amd.gen import workloads as W
    n, cols, streams, expect = W.c3_stripe(300_000, 2, compression, index=True)
    res = G.gpu_decode(n, cols, streams, compression=compression)
    W.check_result(res, cols, expect)
    G.assert_column_parity(res, 0, cols[0], [s[:3] for s in streams], n, 8192, compression=compression, what=("c3 entries", compression))
    res.free()
