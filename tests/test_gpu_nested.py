"""Struct columns on the device (array_decoder/struct_decoder.rs:58-78; parent / child PRESENT merge, mod.rs:216-252): a field's
PRESENT stream has one bit per row in which the Struct is present; the field's Arrow validity is those bits dealt out to the
Struct's valid rows.  Files are read through the reader front end (Arrow C Data export builds the nested arrays) and compared
with PyArrow (= Apache ORC C++, the oracle the reference's integration suite is pinned to): the reference's nested_struct.orc,
and files the ORC C++ writer makes here from tables with nulls at every level."""
import numpy as np
import pyarrow as pa
import pyarrow.orc as orc
import pytest

import arrow_util as A
from orc_rust_amd import ArrowReaderBuilder, capi

pytestmark = pytest.mark.gpu
_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def read_all(path, names=None, batch_size=8192, prefetch=2, selection=None):
    b = ArrowReaderBuilder.try_new(path, ctx()).with_batch_size(batch_size).with_prefetch(prefetch)
    if names is not None:
        b = b.with_projection(names)
    if selection is not None:
        b = b.with_row_selection(selection)
    return list(b.build())


def table_of(batches):
    """the batches as one table (a root column is `not null` in batches without nulls -- RecordBatch::try_from_iter -- so the
    batch schemas differ in that flag: the columns are put together, not the batches)"""
    names = batches[0].schema.names
    return pa.table({n: pa.chunked_array([b.column(i) for b in batches]) for i, n in enumerate(names)})


def test_the_references_nested_struct_file():
    path = A.data_path("nested_struct.orc")
    want = A.expected_table("nested_struct")
    batches = read_all(path)
    got = table_of(batches)
    assert got.schema.names == ["nest"] and got.schema.field("nest").type == want.schema.field("nest").type
    assert got.column("nest").to_pylist() == want.column("nest").to_pylist()
    assert got.column("nest").to_pylist() == [{"a": 1.0, "b": True}, {"a": 3.0, "b": None}, {"a": None, "b": None}, None, {"a": -3.0, "b": None}]


def struct_table(n, seed):
    rng = np.random.default_rng(seed)
    def maybe(vals, p):
        mask = rng.random(n) < p
        return pa.array(vals, mask=mask)
    inner = pa.StructArray.from_arrays(
        [maybe(rng.normal(size=n), 0.2), maybe(rng.random(n) < 0.5, 0.3), maybe(rng.integers(-1000, 1000, n).astype(np.int16), 0.0)],
        names=["d", "e", "h"], mask=pa.array(rng.random(n) < 0.15))
    words = np.array(["", "AIR", "REG AIR", "TRUCK", "a much longer string value than the others", "ü–€"])
    outer = pa.StructArray.from_arrays(
        [maybe(rng.integers(-2**40, 2**40, n), 0.1), maybe(words[rng.integers(0, len(words), n)], 0.25), inner,
         maybe(rng.integers(0, 3650, n).astype("datetime64[D]"), 0.5)],
        names=["a", "b", "c", "g"], mask=pa.array(rng.random(n) < 0.3))
    all_there = pa.StructArray.from_arrays([pa.array(rng.integers(0, 100, n)), maybe(rng.random(n), 0.4)], names=["x", "y"])  # a Struct without PRESENT
    return pa.table({"id": pa.array(np.arange(n)), "s": outer, "t": all_there, "plain": maybe(rng.integers(0, 9, n).astype(np.int32), 0.05)})


@pytest.mark.parametrize("compression", ["zstd", "uncompressed", "zlib"])
@pytest.mark.parametrize("batch_size", [8192, 1000])
def test_structs_with_nulls_at_every_level(tmp_path, compression, batch_size):
    n = 60_000
    t = struct_table(n, 11)
    path = str(tmp_path / "structs.orc")
    orc.write_table(t, path, compression=compression, stripe_size=1 << 16)
    f = orc.ORCFile(path)
    assert f.nstripes >= 2
    want = f.read()
    for prefetch in (0, 2):
        batches = read_all(path, batch_size=batch_size, prefetch=prefetch)
        assert all(b.num_rows <= batch_size for b in batches)
        got = table_of(batches)
        assert got.schema.names == want.schema.names
        for name in want.schema.names:
            g, w = got.column(name).combine_chunks(), want.column(name).combine_chunks()
            assert g.type == w.type, (name, g.type, w.type)
            assert g.equals(w), name
    # a projection that leaves the Structs out, and one that takes only a Struct
    assert table_of(read_all(path, ["plain", "id"])).column("plain").combine_chunks().equals(want.column("plain").combine_chunks())
    only = table_of(read_all(path, ["s"]))
    assert only.schema.names == ["s"] and only.column("s").combine_chunks().equals(want.column("s").combine_chunks())


def test_row_selection_over_struct_columns(tmp_path):
    n = 30_000
    t = struct_table(n, 5)
    path = str(tmp_path / "structs.orc")
    orc.write_table(t, path, compression="zstd", stripe_size=1 << 16)
    want = orc.ORCFile(path).read()
    sel = [(100, True), (5000, False), (9000, True), (3, False), (12000, True), (3000, False), (897, True)]  # (all 30 000 rows: stripes behind a selection's end are read whole, arrow_reader.rs:296-308)
    # (batch size above the longest select run: a select run longer than a batch is not stepped through by the reference,
    #  mod.rs:338-360 -- tests/selection_model.py has that; here the plain reading of the selection is what is compared)
    got = table_of(read_all(path, batch_size=8192, selection=sel))
    keep = np.zeros(n, dtype=bool)
    at = 0
    for cnt, skip in sel:
        if not skip:
            keep[at:at + cnt] = True
        at += cnt
    exp = want.filter(pa.array(keep))
    for name in want.schema.names:
        assert got.column(name).combine_chunks().equals(exp.column(name).combine_chunks()), name


@pytest.mark.parametrize("deep", [6, 14])
def test_structs_nested_six_deep(tmp_path, deep):
    """struct_decoder.rs builds field decoders recursively, to any depth; here the PRESENT phase runs once per depth the call has
    (six levels, and fourteen: deeper than the eight the reader used to follow).  Nulls at every level, a string and a List at the
    bottom."""
    n = 20_000 if deep == 6 else 6_000
    rng = np.random.default_rng(21)
    def level(depth):
        if depth == deep:
            return {"v": int(rng.integers(-1000, 1000)) if rng.random() > 0.1 else None, "s": "x" * int(rng.integers(0, 5)),
                    "l": None if rng.random() < 0.1 else [int(rng.integers(0, 9)) for _ in range(int(rng.integers(0, 3)))]}
        return None if rng.random() < (0.12 if deep == 6 else 0.05) else {"k%d" % depth: level(depth + 1), "n%d" % depth: depth if rng.random() > 0.2 else None}
    typ = pa.struct([("v", pa.int64()), ("s", pa.string()), ("l", pa.list_(pa.int32()))])
    for depth in range(deep - 1, 0, -1):
        typ = pa.struct([("k%d" % depth, typ), ("n%d" % depth, pa.int32())])
    t = pa.table({"id": pa.array(np.arange(n)), "deep": pa.array([level(1) for _ in range(n)], type=typ)})
    path = str(tmp_path / "deep.orc")
    orc.write_table(t, path, compression="zstd", stripe_size=1 << 17)
    want = orc.ORCFile(path).read()
    for batch_size in (8192, 777):
        got = table_of(read_all(path, batch_size=batch_size))
        assert got.column("deep").combine_chunks().type == want.column("deep").combine_chunks().type
        assert got.column("deep").combine_chunks().to_pylist() == want.column("deep").combine_chunks().to_pylist()


# ---- Lists and Maps (list.rs:63-87, map.rs:74-104): offsets on the stripe's rows, the elements in a pass of their own -------
@pytest.mark.parametrize("name", ["nested_array", "nested_map", "nested_array_float", "nested_array_struct", "nested_map_struct"])
def test_the_references_nested_list_and_map_files(name):
    path = A.data_path(name + ".orc")
    want = A.expected_table(name)
    got = table_of(read_all(path))
    assert got.schema.names == want.schema.names
    for col in want.schema.names:
        g, w = got.column(col).combine_chunks(), want.column(col).combine_chunks()
        assert g.type == w.type, (col, g.type, w.type)
        assert g.to_pylist() == w.to_pylist(), col


def list_table(n, seed):
    rng = np.random.default_rng(seed)
    def lists(elem_fn, p_null_list, p_empty, max_len=6):
        out = []
        for _ in range(n):
            u = rng.random()
            if u < p_null_list:
                out.append(None)
            elif u < p_null_list + p_empty:
                out.append([])
            else:
                out.append([elem_fn() for _ in range(int(rng.integers(1, max_len)))])
        return out
    words = ["", "AIR", "REG AIR", "TRUCK", "a much longer string value than the others", "ü–€"]
    ints = lists(lambda: None if rng.random() < 0.1 else int(rng.integers(-10**9, 10**9)), 0.1, 0.1)
    strs = lists(lambda: None if rng.random() < 0.2 else words[int(rng.integers(0, len(words)))], 0.05, 0.2)
    structs = lists(lambda: None if rng.random() < 0.1 else {"a": float(rng.normal()), "b": None if rng.random() < 0.3 else bool(rng.random() < 0.5)}, 0.1, 0.1)
    lol = lists(lambda: None if rng.random() < 0.1 else [int(x) for x in rng.integers(0, 100, int(rng.integers(0, 4)))], 0.1, 0.1, 4)
    maps = []
    for _ in range(n):
        u = rng.random()
        maps.append(None if u < 0.1 else [("k%d" % j, None if rng.random() < 0.2 else float(rng.random())) for j in range(int(rng.integers(0, 4)))])
    in_struct = pa.StructArray.from_arrays([pa.array(ints, type=pa.list_(pa.int64())), pa.array(np.arange(n, dtype=np.int32))], names=["l", "i"],
                                           mask=pa.array(rng.random(n) < 0.2))
    return pa.table({
        "id": pa.array(np.arange(n)),
        "ints": pa.array(ints, type=pa.list_(pa.int64())),
        "strs": pa.array(strs, type=pa.list_(pa.string())),
        "structs": pa.array(structs, type=pa.list_(pa.struct([("a", pa.float64()), ("b", pa.bool_())]))),
        "lol": pa.array(lol, type=pa.list_(pa.list_(pa.int32()))),
        "maps": pa.array(maps, type=pa.map_(pa.string(), pa.float64())),
        "in_struct": in_struct,
    })


@pytest.mark.parametrize("compression", ["zstd", "uncompressed"])
@pytest.mark.parametrize("batch_size", [8192, 1000])
def test_lists_and_maps_with_nulls_and_empties(tmp_path, compression, batch_size):
    n = 30_000
    t = list_table(n, 3)
    path = str(tmp_path / "lists.orc")
    orc.write_table(t, path, compression=compression, stripe_size=1 << 17)
    f = orc.ORCFile(path)
    assert f.nstripes >= 2
    want = f.read()
    for prefetch in (0, 2):
        got = table_of(read_all(path, batch_size=batch_size, prefetch=prefetch))
        assert got.schema.names == want.schema.names
        for name in want.schema.names:
            g, w = got.column(name).combine_chunks(), want.column(name).combine_chunks()
            assert g.type == w.type, (name, g.type, w.type)
            assert g.equals(w), name
    # only a nested column
    only = table_of(read_all(path, ["lol"], batch_size=batch_size))
    assert only.column("lol").combine_chunks().equals(want.column("lol").combine_chunks())
    flat = table_of(read_all(path, ["id"], selection=[(10, True), (100, False), (n - 110, True)]))
    assert flat.column("id").to_pylist() == list(range(10, 110))


def selected_rows(sel, n):
    keep = np.zeros(n, dtype=bool)
    at = 0
    for cnt, skip in sel:
        if not skip:
            keep[at:at + cnt] = True
        at += cnt
    return keep


def test_the_references_row_selections_over_nested_files():
    """tests/row_selection/main.rs:237-304 (and their async twins :559-593): select / skip runs over nested_struct.orc and
    nested_array.orc -- Lists are stepped like every other decoder (list.rs:89) --, and the same over nested_map.orc."""
    for name, sel in (("nested_struct", [(2, False), (2, True), (1, False)]), ("nested_array", [(1, True), (2, False), (2, True)]),
                      ("nested_map", [(1, True), (2, False), (2, True)]), ("nested_array_struct", [(1, False), (1, True), (1, False)]),
                      ("nested_map_struct", [(1, True), (1, False)])):
        path = A.data_path(name + ".orc")
        want = A.expected_table(name)
        n = want.num_rows
        sel = [s for s in sel]
        rest = n - sum(c for c, _ in sel)
        if rest > 0:
            sel.append((rest, True))
        got = table_of(read_all(path, selection=sel))
        exp = want.filter(pa.array(selected_rows(sel, n)[:n]))
        assert got.num_rows == exp.num_rows, (name, got.num_rows, exp.num_rows)
        for col in want.schema.names:
            assert got.column(col).combine_chunks().to_pylist() == exp.column(col).combine_chunks().to_pylist(), (name, col)
    # the reference's own expectation, literally (main.rs:293-302)
    got = table_of(read_all(A.data_path("nested_array.orc"), selection=[(1, True), (2, False), (2, True)]))
    assert got.column("value").to_pylist() == [[5, None, 32, 4, 15], [16, None, 3, 4, 5, 6]]


@pytest.mark.parametrize("compression", ["zstd", "uncompressed"])
@pytest.mark.parametrize("prefetch", [0, 2])
def test_row_selection_over_lists_and_maps(tmp_path, compression, prefetch):
    """Select runs that start and end inside stripes and batches, over List / Map columns with null Lists, empty Lists, null
    elements, Lists of Lists, Lists of Structs, a List inside a Struct: the selected rows of the table PyArrow reads."""
    n = 30_000
    t = list_table(n, 9)
    path = str(tmp_path / "lists_sel.orc")
    orc.write_table(t, path, compression=compression, stripe_size=1 << 17)
    f = orc.ORCFile(path)
    assert f.nstripes >= 2
    want = f.read()
    sel = [(100, True), (5000, False), (9000, True), (3, False), (1, True), (1, False), (11995, True), (3000, False), (900, True)]
    assert sum(c for c, _ in sel) == n
    got = table_of(read_all(path, batch_size=8192, prefetch=prefetch, selection=sel))
    exp = want.filter(pa.array(selected_rows(sel, n)))
    assert got.num_rows == exp.num_rows
    for name in want.schema.names:
        g, w = got.column(name).combine_chunks(), exp.column(name).combine_chunks()
        assert g.type == w.type, (name, g.type, w.type)
        assert g.to_pylist() == w.to_pylist(), name


# ---- Unions (array_decoder/union.rs:69-136): byte-RLE tags, one sparse child per arm ---------------------------------------
def test_the_references_union_file():
    """TestOrcFile.testUnionAndTimestamp.orc (tests/integration/main.rs): a Union of Int and String with null rows (the Union has a
    PRESENT stream: null rows read as type id 0 with a null in the first arm), two stripes."""
    path = A.data_path("TestOrcFile.testUnionAndTimestamp.orc")
    want = A.expected_table("TestOrcFile.testUnionAndTimestamp")
    for batch_size, prefetch in ((8192, 0), (1000, 2), (77, 2)):
        got = table_of(read_all(path, batch_size=batch_size, prefetch=prefetch))
        assert got.schema.field("union").type == want.schema.field("union").type
        assert got.column("union").to_pylist() == want.column("union").to_pylist()
        u = got.column("union").combine_chunks()
        w = want.column("union").combine_chunks()
        assert u.type_codes.equals(w.type_codes)
        # an arm is null wherever it is not the one the type id names (union.rs:95-108) -- and the first arm where the Union itself
        # is null; the ORC C++ reader leaves default values in the rows of the other arms, which no consumer looks at
        for k in range(2):
            here = np.asarray(u.type_codes) == k
            assert u.field(k).filter(pa.array(here)).equals(w.field(k).filter(pa.array(here))), k
            assert u.field(k).filter(pa.array(~here)).null_count == int((~here).sum()), k
        assert got.column("time").equals(want.column("time")) and got.column("decimal").equals(want.column("decimal"))


def union_table(n, seed):
    rng = np.random.default_rng(seed)
    tags = rng.integers(0, 4, n).astype(np.int8)
    tags[: n // 10] = 2  # a long run of one tag
    words = np.array(["", "AIR", "REG AIR", "a longer string value", "ü–€"])
    kids = [pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.2), pa.array(words[rng.integers(0, len(words), n)], mask=rng.random(n) < 0.1),
            pa.array(rng.normal(size=n)), pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.3)]
    u = pa.UnionArray.from_sparse(pa.array(tags), kids)
    inside = pa.StructArray.from_arrays([pa.array(rng.integers(0, 99, n).astype(np.int32)), pa.UnionArray.from_sparse(pa.array((tags % 2).astype(np.int8)), kids[:2])],
                                        names=["k", "v"])  # (a Struct with nulls above a Union: PyArrow's writer aborts on it)
    return pa.table({"id": pa.array(np.arange(n)), "u": u, "s": inside})


@pytest.mark.parametrize("compression", ["zstd", "uncompressed"])
def test_unions_written_by_the_orc_cpp_writer(tmp_path, compression):
    """Four arms (Long, String, Double, Boolean) with nulls of their own; a Union as a field of a Struct."""
    n = 50_000
    t = union_table(n, 5)
    path = str(tmp_path / "unions.orc")
    orc.write_table(t, path, compression=compression, stripe_size=1 << 17)
    f = orc.ORCFile(path)
    want = f.read()
    for batch_size, prefetch in ((8192, 2), (999, 0)):
        got = table_of(read_all(path, batch_size=batch_size, prefetch=prefetch))
        assert got.schema.field("u").type == want.schema.field("u").type
        for name in ("id", "u", "s"):
            assert got.column(name).to_pylist() == want.column(name).to_pylist(), name
    # the Union alone (projection), and under a row selection
    sel = [(1234, True), (500, False), (n - 1734, True)]
    got = table_of(read_all(path, names=["u"], batch_size=8192, selection=sel))
    assert got.column("u").to_pylist() == want.column("u").slice(1234, 500).to_pylist()


# ---- against the nested CPU oracle (tests/oracle_nested.py: the reference's composite decoders restated), batch by batch ------
def _oracle_batches(path, name, batch_size):
    import oracle_nested as N
    import orcfile
    f = orcfile.OrcFile(path)
    cid = dict((n, c) for n, c, _ in f.root_columns())[name]
    return N.read_column(f, cid, batch_size)


@pytest.mark.parametrize("name", ["nested_struct", "nested_array", "nested_array_float", "nested_array_struct", "nested_map", "nested_map_struct"])
def test_the_references_nested_files_batch_by_batch_against_the_oracle(name):
    path = A.data_path(name + ".orc")
    for bs in (8192, 2):
        batches = read_all(path, batch_size=bs)
        for ci, col in enumerate(batches[0].schema.names):
            want = _oracle_batches(path, col, bs)
            assert len(want) == len(batches), (name, col, bs)
            for b, (g, w) in enumerate(zip(batches, want)):
                assert g.column(ci).to_pylist() == w.to_pylist(), (name, col, bs, b)


def test_written_nested_tables_batch_by_batch_against_the_oracle(tmp_path):
    rng = np.random.default_rng(9)
    n = 40_000

    def maybe(v, p=0.1):
        return None if rng.random() < p else v

    rows = []
    for i in range(n):
        inner = maybe({"a": maybe(int(rng.integers(-50, 50))), "s": maybe("x" * int(rng.integers(0, 4)))})
        rows.append({"st": maybe({"in": inner, "l": maybe([maybe(int(rng.integers(0, 9))) for _ in range(int(rng.integers(0, 4)))]), "k": maybe(i)}),
                     "m": maybe([(str(int(k)), maybe(float(k))) for k in rng.integers(0, 99, int(rng.integers(0, 3)))])})
    typ = pa.struct([("st", pa.struct([("in", pa.struct([("a", pa.int64()), ("s", pa.string())])), ("l", pa.list_(pa.int32())), ("k", pa.int64())])),
                     ("m", pa.map_(pa.string(), pa.float64()))])
    arr = pa.array(rows, type=typ)
    path = str(tmp_path / "n.orc")
    orc.write_table(pa.table({"st": arr.field("st"), "m": arr.field("m")}), path, compression="zstd", stripe_size=1 << 16)
    for bs in (8192, 1000):
        batches = read_all(path, batch_size=bs)
        for ci, col in enumerate(("st", "m")):
            want = _oracle_batches(path, col, bs)
            assert len(want) == len(batches)
            for b, (g, w) in enumerate(zip(batches, want)):
                assert g.column(ci).to_pylist() == w.to_pylist(), (col, bs, b)


def _patched_file(tmp_path, src, edits):
    """the file with some streams' bytes replaced by bytes of the same length (uncompressed files: a stream lies where the stripe
    footer says)"""
    import orcfile
    f = orcfile.OrcFile(src)
    assert f.compression_name == "none"
    buf = bytearray(f.buf)
    for s in f.stripes:
        off = s.offset
        for kind, col, length in s.stream_list:
            if (col, kind) in edits:
                new = edits[(col, kind)](bytes(buf[off:off + length]))
                assert len(new) == length
                buf[off:off + length] = new
            off += length
    p = str(tmp_path / "patched.orc")
    open(p, "wb").write(bytes(buf))
    return p


def test_the_two_documented_differences_on_corrupt_nested_input(tmp_path):
    """DESIGN 2 names two places where corrupt NESTED input is handled differently from the reference; the nested oracle says what
    the reference does, the test pins what this path does (and that valid input around it is untouched).
    (1) A Struct FIELD whose PRESENT stream fails: the reference drops the field's validity for the batch (derive_present_vec,
        mod.rs:247-251) and decodes n values; here the bits the stream did not deliver read as present.
    (2) A failure among the ELEMENTS of a List: the reference fails the batch whose elements run dry; here the elements of a
        stripe are one batch: the stripe fails in its first batch."""
    import oracle_nested as N
    import orcfile
    rng = np.random.default_rng(2)
    n = 20_000
    vals = pa.array(rng.integers(0, 1000, n), mask=rng.random(n) < 0.3)
    st = pa.StructArray.from_arrays([vals], names=["v"], mask=pa.array(rng.random(n) < 0.2))
    lst = pa.array([[int(x) for x in rng.integers(0, 9, int(rng.integers(0, 4)))] for _ in range(n)], type=pa.list_(pa.int64()))
    src = str(tmp_path / "src.orc")
    orc.write_table(pa.table({"st": st, "l": lst}), src, compression="uncompressed", stripe_size=1 << 26)
    f = orcfile.OrcFile(src)
    assert len(f.stripes) == 1
    root = dict((nm, c) for nm, c, _ in f.root_columns())
    field = f.types[root["st"]].subtypes[0]
    elem = f.types[root["l"]].subtypes[0]
    # (1) the field's PRESENT stream cut short by a byte-RLE header that promises more bytes than the stream has
    p1 = _patched_file(tmp_path, src, {(field, N.PRESENT): lambda b: b[:len(b) // 2] + bytes([0x81]) + b[len(b) // 2 + 1:][: len(b) - len(b) // 2 - 1]})
    want = _oracle_batches(p1, "st", 8192)  # (the oracle does not fail: the error is swallowed)
    got = read_all(p1, names=["st"], batch_size=8192)
    assert len(got) == len(want)
    g_all = pa.chunked_array([b.column(0) for b in got]).combine_chunks()
    w_all = pa.chunked_array(want).combine_chunks()
    assert g_all.is_valid().to_pylist() == w_all.is_valid().to_pylist()  # the Struct's own validity: the same
    same = [b for b, (g, w) in enumerate(zip(got, want)) if g.column(0).to_pylist() == w.to_pylist()]
    assert same and same[0] == 0  # every batch in front of the damage is the oracle's, bit for bit
    # (2) the elements' DATA stream cut short: the reference fails at the batch that runs dry, this path in the stripe's first batch
    # (600 bytes of 0x7f: longer than a run of these one-byte values, so a run HEADER lies in them -- DIRECT, 64 bits wide, 384 values:
    # the stream runs dry)
    p2 = _patched_file(tmp_path, src, {(elem, N.DATA): lambda b: b[:len(b) - 600] + bytes([0x7f] * 600)})
    ok_batches = 0
    try:
        f2 = orcfile.OrcFile(p2)
        node = N.build(f2, f2.stripes[0], root["l"])
        left = n
        while left > 0:
            node.next_batch(min(8192, left), None)
            left -= 8192
            ok_batches += 1
        oracle_fails = None
    except N.OracleError as e:
        oracle_fails = e.status
    assert oracle_fails == 1 and ok_batches == 2  # IoError, in the stripe's LAST batch: two batches come out of the reference
    got = []
    with pytest.raises(capi.OrcGpuError) as ei:
        for b in ArrowReaderBuilder.try_new(p2, ctx()).with_projection(["l"]).with_batch_size(8192).build():
            got.append(b)
    assert ei.value.code == oracle_fails and len(got) == 0  # the same error kind; in the stripe's FIRST batch (documented)
