"""The workload generators of orc_rust_amd/gen/workloads.py against the pinned
CPU oracle: the streams they emit must decode, batch by batch, to exactly the Arrow buffers their `expect`
dictionaries state -- bench.py's full-size check relies on those -- and must have the shapes BASELINE.md names
(C5: every DATA run PATCHED_BASE; C4: dictionary strings, Decimal128(15,2), no PRESENT streams)."""
import numpy as np

import oracle_lib as O
from orc_rust_amd.gen import workloads as W


def oracle_buffers(col, streams, n, compression, batch=8192):
    sd = {k: (b.tobytes() if isinstance(b, np.ndarray) else bytes(b)) for cid, k, b in streams if cid == col["column_id"]}
    oc = O.Column(col["orc_type"], col["encoding"], sd, dictionary_size=col.get("dictionary_size", 0), precision=col.get("precision", 0),
                  scale=col.get("scale", 0), compression=compression)
    assert oc.status == O.OK
    vals, lens, nulls, left = [], [], 0, n
    while left > 0:
        b = oc.next_batch(min(batch, left))
        assert b["status"] == O.OK, (col, b["status"])
        vals.append(b["values"])
        nulls += b["null_count"]
        if b["offsets"] is not None:
            assert b["offsets"][0] == 0
            lens.append(np.diff(b["offsets"]))
        left -= batch
    oc.close()
    return b"".join(vals), (np.concatenate(lens) if lens else None), nulls


def test_lineitem_stripe_streams_decode_to_the_generated_values():
    rows = 40_000
    table = W.lineitem_table(rows)
    assert table["l_orderkey"][0] == 1 and int(table["l_linenumber"].max()) <= 7 and int(table["l_quantity"].max()) == 5000
    for comp in ("zstd", "none"):
        n, cols, streams, expect = W.lineitem_stripe(table, 5_000, rows, comp)
        assert len(cols) == 16 and not any(k == W.PRESENT for _, k, _ in streams)
        for c in cols:
            got, lens, nulls = oracle_buffers(c, streams, n, comp)
            e = expect[c["column_id"]]
            assert got == e["values"], (comp, c["name"])
            assert nulls == 0
            if "lengths" in e:
                assert np.array_equal(lens, e["lengths"]), c["name"]


def test_lineitem_column_subset_matches_the_full_table():
    rows = 20_000
    full = W.lineitem_table(rows)
    part = W.lineitem_table(rows, names=["l_suppkey", "l_comment", "l_shipmode"])
    assert sorted(part) == ["l_comment", "l_shipmode", "l_suppkey"]
    assert np.array_equal(part["l_suppkey"], full["l_suppkey"]) and np.array_equal(part["l_shipmode"], full["l_shipmode"])
    assert np.array_equal(part["l_comment"][0], full["l_comment"][0]) and np.array_equal(part["l_comment"][1], full["l_comment"][1])
    n, cols, streams, _ = W.lineitem_stripe(part, 0, rows, "none", column_ids=[3, 15, 16])
    assert [c["column_id"] for c in cols] == [3, 15, 16]


def test_c5_is_all_patched_base_and_decodes():
    n = 100_000
    _, cols, streams, expect, stats = W.c5_stripe(n, 3)
    assert stats["patched_base"] == (n + 511) // 512 and stats["direct"] == stats["delta"] == stats["short_repeat"] == 0
    got, _, nulls = oracle_buffers(cols[0], streams, n, "lz4")
    assert got == expect[1]["values"] and nulls == 0
    secs, micros = W.c5_values(n, 3)
    full = (n // 512) * 512
    per_run = (secs[:full] >= (1 << 24)).reshape(-1, 512).sum(axis=1)
    assert int(per_run.min()) == 20 and int(per_run.max()) == 20


def test_c3_decodes():
    n = 50_000
    _, cols, streams, expect = W.c3_stripe(n, 1)
    got, lens, nulls = oracle_buffers(cols[0], streams, n, "snappy")
    assert got == expect[1]["values"] and np.array_equal(lens, expect[1]["lengths"])
    assert nulls == int(n - expect[1]["present"].sum())
