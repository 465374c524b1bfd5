"""ArrowReaderBuilder / ArrowReader over the GPU path: reads the reference's fixture files the way
tests/basic/main.rs and tests/integration/main.rs do (open -> iterate RecordBatches -> compare with
the expected table), plus the builder knobs (batch size, projection, byte-range stripe filter)."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.orc as orc
import pytest

import arrow_util as A
import orcfile
from orc_rust_amd import ArrowReaderBuilder, capi

pytestmark = pytest.mark.gpu

UTC_ZONES = (None, "UTC", "GMT", "Etc/UTC", "Etc/GMT")
_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def flat_names(path):
    f = orcfile.OrcFile(path)
    return [n for n, c, t in f.flat_columns()], f


FILES = ["test.orc", "alltypes.none.orc", "alltypes.snappy.orc", "alltypes.zlib.orc", "alltypes.zstd.orc", "alltypes.lz4.orc", "alltypes.lzo.orc", "TestVectorOrcFile.testLzo.orc",
         "string_long_long.orc", "string_dict_gzip.orc", "long_bool_gzip.orc", "patched_int.orc", "test_bigint.orc",
         "TestOrcFile.testSnappy.orc", "TestOrcFile.testWithoutIndex.orc", "TestVectorOrcFile.testLz4.orc",
         "TestVectorOrcFile.testZstd.0.12.orc", "decimal.orc", "nulls-at-end-snappy.orc", "TestOrcFile.testSeek.orc",
         "TestOrcFile.test1.orc", "orc_split_elim_new.orc", "TestOrcFile.testDate1900.orc", "over1k_bloom.orc", "demo-12-zlib.orc"]


@pytest.mark.parametrize("name", FILES)
def test_read_file_matches_expectation(name):
    path = A.data_path(name)
    names, f = flat_names(path)
    expected = A.expected_table(name[:-4])
    reader = ArrowReaderBuilder.try_new(path, ctx()).with_projection(names).build()
    assert reader.total_row_count() == f.number_of_rows
    assert reader.column_names() == names
    batches = list(reader)
    assert sum(b.num_rows for b in batches) == f.number_of_rows
    assert all(b.num_rows <= 8192 for b in batches)
    for ci, cname in enumerate(names):
        got = pa.chunked_array([b.column(ci) for b in batches]) if batches else pa.chunked_array([], type=expected.column(cname).type)
        want = expected.column(cname)
        if got.type != want.type:
            want = want.cast(got.type)
        if f.types[dict((n, c) for n, c, t in f.flat_columns())[cname]].kind == 17:
            import pyarrow.compute as pc
            got, want = pc.utf8_rtrim_whitespace(got), pc.utf8_rtrim_whitespace(want)
        assert got.equals(want), (name, cname)


@pytest.mark.parametrize("prefetch", [0, 2])
def test_empty_projection_gives_row_counts_only(prefetch):
    """No column projected (array_decoder/mod.rs:534-549): the reference still yields one RecordBatch per batch of every stripe,
    without columns, whose row count is min(batch size, rows left in the stripe)."""
    path = A.data_path("TestOrcFile.testSeek.orc")  # 7 stripes, 32768 rows
    _, f = flat_names(path)
    r = ArrowReaderBuilder.try_new(path, ctx()).with_projection([]).with_batch_size(1000).with_prefetch(prefetch).build()
    assert r.column_names() == []
    batches = list(r)
    assert all(b.num_columns == 0 for b in batches)
    want = []
    for st in f.stripes:
        left = st.number_of_rows
        while left:
            want.append(min(1000, left))
            left -= want[-1]
    assert [b.num_rows for b in batches] == want and sum(want) == 32768
    # ... and under a row selection: the selected rows are counted (mod.rs:302-365)
    r = ArrowReaderBuilder.try_new(path, ctx()).with_projection([]).with_batch_size(1000).with_prefetch(prefetch)
    r = r.with_row_selection([(100, True), (2500, False), (30168, True)]).build()
    got = [b.num_rows for b in r]
    import selection_model as M
    model = M.file_batches([(100, True), (2500, False), (30168, True)], [st.number_of_rows for st in f.stripes], 1000)
    want = []
    for st, per in zip(f.stripes, model):
        if per is None:  # (a selection with no rows left no longer applies: the stripe is read whole, arrow_reader.rs:296-308)
            left = st.number_of_rows
            while left:
                want.append(min(1000, left))
                left -= want[-1]
        else:
            want += [ln for _, ln in per]
    assert got == want


def test_builder_knobs():
    path = A.data_path("TestOrcFile.testSeek.orc")  # 7 stripes, 32768 rows
    names, f = flat_names(path)
    expected = A.expected_table("TestOrcFile.testSeek")
    # batch size
    r = ArrowReaderBuilder.try_new(path, ctx()).with_projection(["int1", "string1"]).with_batch_size(1000).build()
    batches = list(r)
    assert all(b.num_rows <= 1000 for b in batches) and sum(b.num_rows for b in batches) == 32768
    assert batches[0].schema.names == ["int1", "string1"]
    assert pa.chunked_array([b.column(0) for b in batches]).equals(expected.column("int1"))
    # bytes source (the `Bytes` ChunkReader)
    data = open(path, "rb").read()
    r = ArrowReaderBuilder.try_new(data, ctx()).with_projection(["long1"]).build()
    assert pa.chunked_array([b.column(0) for b in r]).equals(expected.column("long1"))
    # byte range keeps the stripes whose offset lies inside (arrow_reader.rs:358-372)
    first, second = f.stripes[0], f.stripes[1]
    r = ArrowReaderBuilder.try_new(path, ctx()).with_projection(["int1"]).with_file_byte_range(second.offset, second.offset + 1).build()
    got = pa.chunked_array([b.column(0) for b in r])
    assert len(got) == second.number_of_rows
    assert got.equals(expected.column("int1").slice(first.number_of_rows, second.number_of_rows))
    # the whole file, nested columns (Struct, List, Map) included
    batches = list(ArrowReaderBuilder.try_new(path, ctx()).build())
    assert batches[0].schema.names == expected.schema.names
    for i, cname in enumerate(expected.schema.names):
        got = pa.chunked_array([b.column(i) for b in batches]).combine_chunks()
        want = expected.column(cname).combine_chunks()
        if got.type != want.type:
            want = want.cast(got.type)
        assert got.equals(want), cname


def test_timestamp_precision_and_errors():
    path = A.data_path("pyarrow_timestamps.orc")
    r = ArrowReaderBuilder.try_new(path, ctx()).with_timestamp_precision("us").build()
    b = next(iter(r))
    assert b.schema.field(0).type == pa.timestamp("us") and b.schema.field(1).type == pa.timestamp("us", tz="UTC")
    exp = A.expected_table("pyarrow_timestamps")
    assert b.column(0).equals(exp.column(0).cast(pa.timestamp("us")).chunk(0))
    # overflowing_timestamps.orc: out of the ns range -> DecodeTimestamp (tests/basic/main.rs:546-566)
    path = A.data_path("overflowing_timestamps.orc")
    r = ArrowReaderBuilder.try_new(path, ctx()).build()
    with pytest.raises(capi.OrcGpuError) as e:
        list(r)
    assert e.value.code == 4
    # ...but readable at microsecond precision
    r = ArrowReaderBuilder.try_new(path, ctx()).with_timestamp_precision("us").build()
    assert sum(b.num_rows for b in r) == 3


# ---- read-ahead (orcgpu_reader_set_prefetch): the same batches, three stripes in flight ---------------------------------------
def _all_batches(path_or_bytes, names, prefetch, batch_size=8192, selection=None):
    b = ArrowReaderBuilder.try_new(path_or_bytes, ctx()).with_projection(names).with_batch_size(batch_size).with_prefetch(prefetch)
    if selection is not None:
        b = b.with_row_selection(selection)
    return list(b.build())


def _same_batches(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.schema.equals(y.schema) and x.num_rows == y.num_rows and x.equals(y)


@pytest.mark.parametrize("prefetch", [1, 2, 4])
def test_reading_ahead_yields_the_batches_of_the_serial_reader(prefetch):
    """TestOrcFile.testSeek.orc: 7 stripes.  prefetch 0 reads, stages, decodes and copies back inside next_batch; with
    read-ahead a worker stages stripe k + 1, decodes k and starts its copy back while k - 1 is consumed
    (async_arrow_reader.rs:165-280) -- the RecordBatches are the same, one by one."""
    path = A.data_path("TestOrcFile.testSeek.orc")
    names = [n for n in flat_names(path)[0]]
    serial = _all_batches(path, names, 0, 1000)
    ahead = _all_batches(path, names, prefetch, 1000)
    _same_batches(serial, ahead)
    assert sum(b.num_rows for b in ahead) == 32768
    # ... and under a row selection that is split over the stripes
    sel = [(100, True), (5000, False), (9000, True), (3, False), (12000, True), (4000, False)]
    _same_batches(_all_batches(path, names, 0, 777, sel), _all_batches(path, names, prefetch, 777, sel))


def _lineitem_orc(tmp_path, rows, stripe_bytes):
    """A multi-stripe lineitem file written by the ORC C++ writer (PyArrow), Zstandard, dictionary encoding on."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_lineitem
    import pyarrow.orc as orc
    from orc_rust_amd.gen import workloads as W
    t = make_lineitem.arrow_table(W.lineitem_table(rows), rows)
    path = str(tmp_path / "lineitem.orc")
    orc.write_table(t, path, compression="zstd", compression_block_size=65536, dictionary_key_size_threshold=0.8, stripe_size=stripe_bytes)
    return path, t


def test_reading_ahead_over_a_multi_stripe_lineitem_file(tmp_path):
    path, table = _lineitem_orc(tmp_path, 400_000, 1 << 20)  # 10 stripes
    names = table.schema.names
    b0 = ArrowReaderBuilder.try_new(path, ctx())
    n_stripes = b0.stripe_count()
    assert n_stripes >= 8, n_stripes
    serial = _all_batches(path, names, 0)
    ahead = _all_batches(path, names, 2)
    _same_batches(serial, ahead)
    got = pa.Table.from_batches(ahead)
    want = table.cast(got.schema) if got.schema != table.schema else table
    assert got.equals(want)


def test_a_reading_ahead_reader_can_be_dropped_half_way(tmp_path):
    path = A.data_path("TestOrcFile.testSeek.orc")
    names = [n for n in flat_names(path)[0]]
    for _ in range(3):
        r = ArrowReaderBuilder.try_new(path, ctx()).with_projection(names).with_batch_size(500).with_prefetch(3).build()
        first = next(iter(r))
        assert first.num_rows == 500
        r.close()  # the worker is in the middle of later stripes: it is stopped, its results are freed
    # the context is usable afterwards
    assert sum(b.num_rows for b in _all_batches(path, names, 2)) == 32768


def test_errors_arrive_in_order_when_reading_ahead():
    # overflowing_timestamps.orc: DecodeTimestamp in its only stripe, serial and ahead alike
    path = A.data_path("overflowing_timestamps.orc")
    for prefetch in (0, 2):
        r = ArrowReaderBuilder.try_new(path, ctx()).with_prefetch(prefetch).build()
        with pytest.raises(capi.OrcGpuError) as e:
            list(r)
        assert e.value.code == 4


def test_an_io_error_ends_the_iterator_with_that_error(tmp_path):
    """The failing batch of a stripe may fail with ANY OrcError -- IoError is status 1 --: the end of the file has a code of its own
    (ORCGPU_END_OF_FILE), so a stream that runs dry is reported, not taken for the end (arrow_reader.rs:333-346: the iterator
    yields the error, then ends).  Batches in front of the failing one are the file's."""
    import orcfile
    n = 30_000
    t = pa.table({"a": pa.array(np.arange(n, dtype=np.int64) * 977 % 100003), "b": pa.array(np.arange(n, dtype=np.int32))})
    src = str(tmp_path / "src.orc")
    orc.write_table(t, src, compression="uncompressed", stripe_size=1 << 26)
    f = orcfile.OrcFile(src)
    cid = dict((nm, c) for nm, c, _ in f.root_columns())["a"]
    buf = bytearray(f.buf)
    s = f.stripes[0]
    off = s.offset
    for kind, col, length in s.stream_list:
        if (col, kind) == (cid, 1):
            buf[off + length - 2000:off + length] = bytes([0x7f]) * 2000  # a DIRECT header that promises more than the stream holds
        off += length
    bad = str(tmp_path / "bad.orc")
    open(bad, "wb").write(bytes(buf))
    oc = orcfile.OrcFile(bad).oracle_column(orcfile.OrcFile(bad).stripes[0], cid)
    ok = 0
    while True:
        b = oc.next_batch(8192)
        if b["status"]:
            break
        ok += 1
    assert b["status"] == 1 and 0 < ok < 4
    for prefetch in (0, 2):
        got = []
        with pytest.raises(capi.OrcGpuError) as ei:
            for rb in ArrowReaderBuilder.try_new(bad, ctx()).with_batch_size(8192).with_prefetch(prefetch).build():
                got.append(rb)
        assert ei.value.code == 1 and len(got) == ok, (prefetch, len(got), ok)
        assert pa.Table.from_batches(got).column("a").to_pylist() == t.column("a").to_pylist()[:ok * 8192]
