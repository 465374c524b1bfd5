"""Pins the NESTED part of the CPU oracle (tests/oracle_nested.py: Struct / List / Map / Union over the C primitives): the
reference's own nested fixtures against the feathers its integration suite is pinned to (scripts/generate_arrow.py ->
tests/golden/expected), the Apache ORC files with Struct-of-List-of-Struct / Map / Union columns against PyArrow, and tables
with nulls and empties at every level written by the ORC C++ writer."""
import numpy as np
import pyarrow as pa
import pyarrow.orc as orc
import pytest

import arrow_util as A
import oracle_nested as N
import orcfile


def column_of(f, name, batch_size=8192):
    cid = dict((n, c) for n, c, _ in f.root_columns())[name]
    chunks = N.read_column(f, cid, batch_size)
    return pa.chunked_array(chunks) if chunks else None


def same(got, want):
    """values and nulls, whatever the field names / nullability flags of the two type trees"""
    got, want = got.combine_chunks(), want.combine_chunks()
    return got.to_pylist() == want.to_pylist()


@pytest.mark.parametrize("stem", ["nested_struct", "nested_array", "nested_array_float", "nested_array_struct", "nested_map", "nested_map_struct"])
def test_the_references_nested_fixtures(stem):
    f = orcfile.OrcFile(A.data_path(stem + ".orc"))
    want = A.expected_table(stem)
    for name, cid, typ in f.root_columns():
        if typ.kind not in (N.STRUCT, N.LIST, N.MAP, N.UNION):
            continue
        for bs in (8192, 3):
            got = column_of(f, name, bs)
            assert same(got, want.column(name)), (stem, name, bs)


@pytest.mark.parametrize("name", ["TestOrcFile.testSeek.orc", "TestOrcFile.testUnionAndTimestamp.orc"])
def test_apache_files_with_nested_columns(name):
    f = orcfile.OrcFile(A.data_path(name))
    checked = 0
    for cname, cid, typ in f.root_columns():
        if typ.kind not in (N.STRUCT, N.LIST, N.MAP, N.UNION):
            continue
        want = orc.ORCFile(A.data_path(name)).read(columns=[cname])  # (column by column: the file's Timestamp column needs tzdata)
        got = column_of(f, cname, 1000)
        w = want.column(cname).combine_chunks()
        g = got.combine_chunks()
        if typ.kind == N.UNION:
            # the ORC C++ reader leaves default values in the arms a tag does not name; the reference nulls them (union.rs:95-108):
            # compare what a reader of the Union sees
            assert g.to_pylist() == w.to_pylist(), (name, cname)
        else:
            assert same(got, want.column(cname)), (name, cname)
        checked += 1
    assert checked


def test_written_tables_with_nulls_at_every_level(tmp_path):
    rng = np.random.default_rng(5)
    n = 30_000

    def maybe(v, p=0.1):
        return None if rng.random() < p else v

    rows = []
    for i in range(n):
        inner = maybe({"a": maybe(int(rng.integers(-50, 50))), "s": maybe("x" * int(rng.integers(0, 4)))})
        lst = maybe([maybe(int(rng.integers(0, 9))) for _ in range(int(rng.integers(0, 4)))])
        mp = maybe([(str(int(k)), maybe(float(k))) for k in rng.integers(0, 99, int(rng.integers(0, 3)))])
        rows.append({"st": maybe({"in": inner, "l": lst, "k": maybe(i)}), "m": mp, "ll": maybe([maybe([1, 2][: int(rng.integers(0, 3))]) for _ in range(int(rng.integers(0, 3)))])})
    typ = pa.struct([("st", pa.struct([("in", pa.struct([("a", pa.int64()), ("s", pa.string())])), ("l", pa.list_(pa.int32())), ("k", pa.int64())])),
                     ("m", pa.map_(pa.string(), pa.float64())), ("ll", pa.list_(pa.list_(pa.int16())))])
    arr = pa.array(rows, type=typ)
    t = pa.table({"st": arr.field("st"), "m": arr.field("m"), "ll": arr.field("ll")})
    path = str(tmp_path / "n.orc")
    orc.write_table(t, path, compression="zlib", stripe_size=1 << 16)
    f = orcfile.OrcFile(path)
    want = orc.ORCFile(path).read()
    for name in ("st", "m", "ll"):
        for bs in (8192, 777):
            assert same(column_of(f, name, bs), want.column(name)), (name, bs)


def test_a_failing_present_stream_of_a_struct_field_is_swallowed():
    """derive_present_vec maps an error to None (mod.rs:247-251): the FIELD of a Struct whose PRESENT stream fails is decoded as
    if every row of the batch were present -- even rows in which the Struct itself is null lose their null in the field's own
    validity (the Struct's validity still says so)."""
    f = orcfile.OrcFile(A.data_path("nested_struct.orc"))
    s = f.stripes[0]
    root = dict((n, c) for n, c, _ in f.root_columns())
    cid = next(c for n, c in root.items() if f.types[c].kind == N.STRUCT)
    field = f.types[cid].subtypes[0]
    assert (field, N.PRESENT) in s.streams
    good = N.build(f, s, cid).next_batch(s.number_of_rows, None)
    s.streams[(field, N.PRESENT)] = b""  # the stream runs dry at once: the decoder fails, derive_present_vec says "no nulls"
    try:
        bad = N.build(f, s, cid).next_batch(s.number_of_rows, None)
        # the Struct's validity is what it was; the field has no nulls of its own any more
        assert bad.is_valid().to_pylist() == good.is_valid().to_pylist()
        assert bad.field(0).null_count == 0
    except N.OracleError as e:
        # (a field with fewer values than rows runs dry instead: then the batch fails -- also what the reference does)
        assert e.status in (1, 2)
