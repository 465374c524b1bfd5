"""GPU parity at FILE level: every flat root column of every golden fixture (the reference's own
tests/basic/data + tests/integration/data files: all compressions, RLE v1 and v2, dictionary and
direct strings, decimals, timestamps, nulls, 32-byte compression chunks) goes through the C ABI
and must equal the oracle batch by batch, byte by byte (and thereby the committed PyArrow /
reference expectations the oracle is pinned to in test_oracle_files.py)."""
import pytest

import arrow_util as A
import gpu_util as G
import orcfile

pytestmark = pytest.mark.gpu


def stripe_inputs(f, s):
    cols, streams = [], []
    for name, cid, typ in f.flat_columns():
        enc, dsz = s.encodings[cid] if cid < len(s.encodings) else (0, 0)
        cols.append({"column_id": cid, "orc_type": typ.kind, "encoding": enc, "dictionary_size": dsz, "precision": typ.precision,
                     "scale": typ.scale, "name": name})
        for k, v in f.column_streams(s, cid).items():
            streams.append((cid, k, v))
    return cols, streams


@pytest.mark.parametrize("name", A.golden_files())
def test_gpu_matches_oracle_on_fixture(name):
    f = orcfile.OrcFile(A.data_path(name))
    if not f.flat_columns() or not f.stripes:
        pytest.skip("no flat columns / no stripes")
    big = f.number_of_rows > 500_000
    for si, s in enumerate(f.stripes):
        if not s.number_of_rows:
            continue
        cols, streams = stripe_inputs(f, s)
        for batch in ((8192,) if big else (8192, 1000)):
            res = G.gpu_decode(s.number_of_rows, cols, streams, compression=f.compression_name, block_size=f.block_size, batch_size=batch,
                               writer_timezone=s.writer_timezone)
            fails = A.REFERENCE_FAILS.get(name, ())
            if fails:
                # several columns fail (in their first batch): the result names the first of them, as the reader would (arrow_reader.rs:333-346)
                G.assert_stripe_parity(res, cols, streams, s.number_of_rows, batch, compression=f.compression_name, block_size=f.block_size, what=(name, si, batch))
            for ci, c in enumerate(cols):
                if c["name"] in fails:
                    continue
                G.assert_column_parity(res, ci, c, streams, s.number_of_rows, batch, compression=f.compression_name, block_size=f.block_size,
                                       what=(name, si, c["name"], batch), writer_timezone=s.writer_timezone)
            res.free()
        if big and si >= 3:
            break


@pytest.mark.parametrize("name", ["test.orc", "alltypes.none.orc", "TestOrcFile.testSnappy.orc", "TestVectorOrcFile.testLz4.orc", "decimal.orc",
                                  "TestOrcFile.testDate1900.orc", "orc_split_elim_new.orc"])
def test_arrow_c_data_export_equals_expectation(name):
    """Arrow C Data Interface export -> pyarrow: logical equality with the committed expectation."""
    import pyarrow as pa
    f = orcfile.OrcFile(A.data_path(name))
    expected = A.expected_table(name[:-4])
    batches = []
    names = None
    for s in f.stripes:
        cols, streams = stripe_inputs(f, s)
        names = [c["name"] for c in cols]
        res = G.gpu_decode(s.number_of_rows, cols, streams, compression=f.compression_name, block_size=f.block_size, writer_timezone=s.writer_timezone)
        assert res.status()[0] == 0
        for b in range(res.n_batches):
            batches.append(res.export_batch(b))
        res.free()
    for ci, cname in enumerate(names):
        got = pa.chunked_array([rb.column(ci) for rb in batches])
        want = expected.column(cname)
        if got.type != want.type:
            want = want.cast(got.type)
        assert got.equals(want), (name, cname)
