"""ArrowReaderBuilder / ArrowReader over the GPU path: reads the reference's fixture files the way
tests/basic/main.rs and tests/integration/main.rs do (open -> iterate RecordBatches -> compare with
the expected table), plus the builder knobs (batch size, projection, byte-range stripe filter)."""
import os

import pyarrow as pa
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


FILES = ["test.orc", "alltypes.none.orc", "alltypes.snappy.orc", "alltypes.zlib.orc", "alltypes.zstd.orc", "alltypes.lz4.orc",
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
    # a nested column is UnsupportedTypeVariant on this path
    with pytest.raises(capi.OrcGpuError) as e:
        list(ArrowReaderBuilder.try_new(path, ctx()).build())
    assert e.value.code == 7


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
