"""The reader's shard spec (orcgpu_reader_set_shard): `world` readers of ONE file, each reading only its own
stripes or its own columns -- here all of them on the one GPU of the box.  Put together (by stripe / side by side) their
batches are the batches of a single reader: same rows, same values, same batch boundaries."""
import numpy as np
import pyarrow as pa
import pyarrow.orc as orc
import pytest

import arrow_util as A
from orc_rust_amd import ArrowReaderBuilder, capi
from orc_rust_amd.gen import workloads as W

pytestmark = pytest.mark.gpu
_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def batches_of(path, rank=0, world=1, mode="stripes", batch_size=1000, prefetch=0, selection=None, names=None):
    b = ArrowReaderBuilder.try_new(path, ctx()).with_batch_size(batch_size).with_prefetch(prefetch)
    if names is not None:
        b = b.with_projection(names)
    if world > 1:
        b = b.with_shard(rank, world, mode)
    if selection is not None:
        b = b.with_row_selection(selection)
    return list(b.build())


def col_lists(batches, i):
    return [b.column(i).to_pylist() for b in batches]


def lineitem_file(tmp_path, rows=260_000):
    """a multi-stripe file of the lineitem table's 16 columns (the decimals as their unscaled int64 values), written by the ORC C++ writer"""
    t = W.lineitem_table(rows)
    cols = {}
    for name, typ, how in W.LINEITEM:
        if how == "dict":
            words = W.DICTS[name]
            cols[name] = pa.array([words[k].decode() for k in t[name]])
        elif how == "direct":
            lens, blob = t[name]
            offs = np.concatenate([[0], np.cumsum(lens)])
            cols[name] = pa.array([bytes(blob[offs[i]:offs[i + 1]]).decode("latin-1") for i in range(rows)])
        else:
            cols[name] = pa.array(t[name])
    path = str(tmp_path / "lineitem.orc")
    orc.write_table(pa.table(cols), path, compression="zstd", stripe_size=1 << 20)
    return path


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("prefetch", [0, 2])
def test_stripe_shards_put_together_are_the_single_reader(world, prefetch):
    path = A.data_path("TestOrcFile.testSeek.orc")  # 7 stripes
    f = orc.ORCFile(path)
    names = ["int1", "string1", "long1", "double1"]
    whole = batches_of(path, names=names, prefetch=prefetch)
    # stripe k's batches come from rank k % world, in order
    per_stripe = [len(range(0, f.read_stripe(k).num_rows, 1000)) for k in range(f.nstripes)]
    shards = [batches_of(path, r, world, "stripes", names=names, prefetch=prefetch) for r in range(world)]
    assert sum(len(s) for s in shards) == len(whole)
    cursor = [0] * world
    together = []
    for k, nb in enumerate(per_stripe):
        r = k % world
        together += shards[r][cursor[r]:cursor[r] + nb]
        cursor[r] += nb
    assert [b.num_rows for b in together] == [b.num_rows for b in whole]
    for i in range(len(names)):
        assert col_lists(together, i) == col_lists(whole, i), names[i]
    # under a row selection: every rank steps the selection through every stripe, read or not
    sel = [(4000, True), (3000, False), (9000, True), (2, False), (10000, True), (6766, False)]
    whole = batches_of(path, names=names, selection=sel, prefetch=prefetch)
    shards = [batches_of(path, r, world, "stripes", names=names, selection=sel, prefetch=prefetch) for r in range(world)]
    rows = lambda bs: sorted(v for b in bs for v in b.column(2).to_pylist())  # long1 values as row identities
    assert sum(b.num_rows for s in shards for b in s) == sum(b.num_rows for b in whole)
    assert sorted(v for s in shards for v in rows(s)) == rows(whole)


@pytest.mark.parametrize("world", [2, 4])
def test_column_shards_side_by_side_are_the_single_reader(tmp_path, world):
    path = lineitem_file(tmp_path)
    assert orc.ORCFile(path).nstripes >= 2
    whole = batches_of(path, batch_size=8192, prefetch=2)
    names = whole[0].schema.names
    seen = []
    for r in range(world):
        mine = batches_of(path, r, world, "columns", batch_size=8192, prefetch=2)
        assert [b.num_rows for b in mine] == [b.num_rows for b in whole]          # the same rows, batch by batch
        for n in mine[0].schema.names:
            i, j = mine[0].schema.names.index(n), names.index(n)
            assert all(a.column(i).equals(b.column(j)) for a, b in zip(mine, whole)), n
            seen.append(n)
    assert sorted(seen) == sorted(names)                                           # every column on exactly one rank
    # the deal follows the weights the header states
    L = ctx().L
    import ctypes as C
    rd = C.c_void_p()
    assert L.orcgpu_reader_open_file(ctx().h, path.encode(), C.byref(rd)) == 0
    w = [L.orcgpu_reader_column_weight(rd, k) for k in range(len(names))]
    L.orcgpu_reader_close(rd)
    assert w[0] == 8 and w[3] == 4 and w[8] == 20 and w[15] == 20  # int64, int32, strings


def test_shard_arguments():
    path = A.data_path("TestOrcFile.testSeek.orc")
    with pytest.raises(capi.OrcGpuError):
        ArrowReaderBuilder.try_new(path, ctx()).with_shard(2, 2)
    with pytest.raises(capi.OrcGpuError):
        ArrowReaderBuilder.try_new(path, ctx()).with_shard(0, 0)
