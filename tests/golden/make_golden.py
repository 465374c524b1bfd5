"""Generates the committed golden fixtures.  Runs ONLY in the build container, where
/root/reference and PyArrow (bundling Apache ORC C++) are present:

    python tests/golden/make_golden.py

* copies the reference's own fixture DATA files (tests/basic/data, tests/integration/data) into
  tests/golden/data/  -- data, not source;
* writes the expected decode of each file as Arrow IPC (feather v2, zstd) into
  tests/golden/expected/, produced by `pyarrow.orc` -- the same independent oracle the
  reference's integration suite is pinned to (scripts/generate_arrow.py:17-36,
  tests/integration/main.rs:35-70);
* copies the reference's committed expected_arrow feathers for the integration files it tests,
  so that the pin is the reference's own expectation wherever one exists.
Nothing here is read at test time except the two output directories.
"""
import os
import shutil
import sys

os.environ.setdefault("TZDIR", "/usr/local/lib/python3.10/dist-packages/tzdata/zoneinfo")
import pyarrow as pa  # noqa: E402
import pyarrow.feather as feather  # noqa: E402
import pyarrow.orc as orc  # noqa: E402
import pyarrow.parquet as pq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests"

BASIC = [
    "alltypes.none.orc", "alltypes.snappy.orc", "alltypes.zlib.orc", "alltypes.zstd.orc", "alltypes.lz4.orc",
    "long_bool.orc", "long_bool_gzip.orc", "string_dict.orc", "string_dict_gzip.orc", "string_long.orc",
    "string_long_long.orc", "string_long_long_gzip.orc", "test.orc", "test_bigint.orc", "patched_int.orc",
    "pyorc_rlev2_patchedbase.orc", "pyarrow_timestamps.orc", "overflowing_timestamps.orc", "demo-12-zlib.orc", "alltypes.lzo.orc", "nested_struct.orc", "nested_array.orc", "nested_map.orc", "nested_array_float.orc", "nested_array_struct.orc", "nested_map_struct.orc",
]
BIG = {"demo-12-zlib", "demo-11-zlib"}
INTEGRATION = [
    "TestOrcFile.columnProjection.orc", "TestOrcFile.testSnappy.orc", "TestOrcFile.testWithoutIndex.orc",
    "TestOrcFile.testPredicatePushdown.orc", "TestOrcFile.testMemoryManagementV11.orc", "TestOrcFile.testMemoryManagementV12.orc",
    "TestOrcFile.testStripeLevelStats.orc", "TestOrcFile.testStringAndBinaryStatistics.orc", "TestOrcFile.testSeek.orc",
    "TestOrcFile.test1.orc", "TestOrcFile.testDate1900.orc", "TestStringDictionary.testRowIndex.orc",
    "TestVectorOrcFile.testLz4.orc", "TestVectorOrcFile.testLzo.orc", "TestVectorOrcFile.testZstd.0.12.orc", "decimal.orc",
    "nulls-at-end-snappy.orc", "orc_index_int_string.orc", "orc_split_elim_new.orc",
    "orc_split_elim_cpp.orc", "over1k_bloom.orc", "bloom_filter.orc", "demo-11-zlib.orc",
    "TestOrcFile.testUnionAndTimestamp.orc", "TestOrcFile.testSargSkipPickupGroupWithoutIndexCPlusPlus.orc", "TestOrcFile.testSargSkipPickupGroupWithoutIndexJava.orc",
    # round 6: the rest of what the reference ships and its active tests read (tests/integration/main.rs:111-114 emptyFile: zero
    # stripes through the reader; tests/basic/main.rs:747-795 timestamps_0001: a year-0000 timestamp, no PyArrow expectation --
    # the reference's own expected value is restated in tests/test_gpu_reader.py)
    "TestOrcFile.emptyFile.orc", "TestOrcFile.metaData.orc", "timestamps_0001.orc", "decimal64_v2.orc", "decimal64_v2_cplusplus.orc",
    "orc_no_format.orc", "complextypes_iceberg.orc", "bad_bloom_filter_1.6.0.orc", "bad_bloom_filter_1.6.11.orc",
]
# Containers that are odd or broken on purpose: kept apart (tests/golden/edge/) so that the per-file parity tests, which list
# tests/golden/data/, do not pick them up.  zero.orc: 0 bytes; version1999.orc: a future format version, no rows, no types worth
# the name; the two files whose root type is not a Struct (the reference reads them as files without columns: schema.rs:154-162);
# tests/integration/data/corrupt/*: a string dictionary without its blob / its length stream, negative dictionary entry lengths,
# a stripe footer whose column encodings do not fit the schema.
EDGE = ["zero.orc", "version1999.orc", "TestOrcFile.testWithoutCompressionBlockSize.orc", "TestOrcFile.testTimestamp.orc",
        "corrupt/missing_blob_stream_in_string_dict.orc", "corrupt/missing_length_stream_in_string_dict.orc",
        "corrupt/negative_dict_entry_lengths.orc", "corrupt/stripe_footer_bad_column_encodings.orc"]


def main():
    data = os.path.join(HERE, "data")
    exp = os.path.join(HERE, "expected")
    os.makedirs(data, exist_ok=True)
    os.makedirs(exp, exist_ok=True)
    for sub, names in (("basic", BASIC), ("integration", INTEGRATION)):
        for n in names:
            src = os.path.join(REF, sub, "data", n)
            dst = os.path.join(data, n)
            shutil.copyfile(src, dst)
            stem = n[:-4]
            ref_feather = os.path.join(REF, "integration", "data", "expected_arrow", stem + ".feather")
            out = os.path.join(exp, stem + ".feather")
            try:
                table = orc.ORCFile(dst).read()
            except Exception as e:  # e.g. overflowing_timestamps: out of the ns range by design
                print("no pyarrow expectation for", n, "->", str(e)[:60])
                if sub == "integration" and os.path.exists(ref_feather):
                    shutil.copyfile(ref_feather, out)  # the reference's own committed expectation
                continue
            if stem in BIG:
                # 1.9 M rows: keep the expectation as dictionary-encoded parquet; demo-11 (ORC 0.11,
                # RLE v1, 385 stripes) holds the same table as demo-12 and shares its expectation.
                big = os.path.join(exp, "demo-12-zlib.parquet")
                if stem == "demo-12-zlib":
                    pq.write_table(table, big, compression="zstd", use_dictionary=True)
                else:
                    assert pq.read_table(big).equals(table), "demo-11 and demo-12 differ"
                continue
            if sub == "integration" and os.path.exists(ref_feather):
                ref_table = feather.read_table(ref_feather)
                same = ref_table.equals(table)
                print(n, "reference feather", "==" if same else "!=", "pyarrow re-read")
                shutil.copyfile(ref_feather, out)
            else:
                feather.write_feather(table, out, compression="zstd")
    edge = os.path.join(HERE, "edge")
    os.makedirs(edge, exist_ok=True)
    for n in EDGE:
        shutil.copyfile(os.path.join(REF, "integration", "data", n), os.path.join(edge, os.path.basename(n)))
    # a Decimal column whose precision reads 0 (what the container fuzz of tests/test_gpu_containers.py once produced): one bit of
    # alltypes.zlib.orc's footer flipped
    broken = bytearray(open(os.path.join(REF, "basic", "data", "alltypes.zlib.orc"), "rb").read())
    broken[1339] ^= 0x20
    open(os.path.join(edge, "decimal_precision_0.orc"), "wb").write(bytes(broken))
    print("done")


if __name__ == "__main__":
    sys.exit(main())
