"""BASELINE.md's workload SHAPES on the HIP path (through the C ABI), batch by batch against the CPU oracle and against
the buffers the generated values imply:

  C4  TPC-H-shaped lineitem stripes, 16 columns, Zstandard (and uncompressed), two stripes in one decode call
  C5  Timestamp(ns) with an all-PATCHED_BASE seconds stream + DIRECT nanoseconds, LZ4
  C3  dictionary Utf8 + PRESENT, Snappy

at 1-2 M rows (the oracle finishes in seconds); bench.py runs the same generators at full size with the
generator-implied check (`workloads.check_result`)."""
import numpy as np
import pytest

import gpu_util as G
from orc_rust_amd.gen import workloads as W

pytestmark = pytest.mark.gpu


def decode_all(stripes, compression):
    c = G.ctx()
    staged = [c.stage(n, streams, cols, compression=compression) for n, cols, streams, *_ in stripes]
    results = c.decode(staged)
    for s in staged:
        s.free()
    return results


@pytest.mark.parametrize("compression", ["zstd", "none"])
def test_c4_lineitem_stripes(compression):
    rows = 1_300_000
    table = W.lineitem_table(rows)
    stripes = [W.lineitem_stripe(table, 0, 1_000_000, compression), W.lineitem_stripe(table, 1_000_000, rows, compression)]
    results = decode_all(stripes, compression)
    for (n, cols, streams, expect), res in zip(stripes, results):
        assert res.status()[0] == 0, res.status()
        assert res.rows == n and res.n_batches == (n + 8191) // 8192
        W.check_result(res, cols, expect)
        for ci, c in enumerate(cols):
            G.assert_column_parity(res, ci, c, streams, n, 8192, compression=compression, what=("C4", compression, c["name"]))
        res.free()


def test_c4_lineitem_column_shard_equals_the_full_decode():
    """A column shard (the multi-GPU partition of C4) decodes to the same buffers as the same columns of a full decode."""
    rows = 200_000
    table = W.lineitem_table(rows)
    full = W.lineitem_stripe(table, 0, rows, "zstd")
    mine = [2, 5, 9, 16]
    part = W.lineitem_stripe(table, 0, rows, "zstd", column_ids=mine)
    rf, rp = decode_all([full], "zstd")[0], decode_all([part], "zstd")[0]
    for pi, c in enumerate(part[1]):
        fi = [k for k, fc in enumerate(full[1]) if fc["column_id"] == c["column_id"]][0]
        for b in range(rf.n_batches):
            x, y = rf.batch(b, fi), rp.batch(b, pi)
            assert x["values"] == y["values"] and x["null_count"] == y["null_count"]
            assert (x["offsets"] is None) == (y["offsets"] is None) and (x["offsets"] is None or np.array_equal(x["offsets"], y["offsets"]))
    rf.free()
    rp.free()


def test_c5_patched_base_timestamps_lz4():
    stripes = [W.c5_stripe(1_048_576, 0), W.c5_stripe(700_001, 1)]
    assert all(s[4]["patched_base"] == (s[0] + 511) // 512 for s in stripes)
    results = decode_all(stripes, "lz4")
    for (n, cols, streams, expect, _), res in zip(stripes, results):
        W.check_result(res, cols, expect)
        G.assert_column_parity(res, 0, cols[0], streams, n, 8192, compression="lz4", what="C5")
        res.free()


@pytest.mark.parametrize("compression", ["snappy", "zstd"])
def test_c3_dictionary_utf8_with_present(compression):
    stripes = [W.c3_stripe(1_500_000, 0, compression), W.c3_stripe(300_000, 1, compression)]
    results = decode_all(stripes, compression)
    for (n, cols, streams, expect), res in zip(stripes, results):
        W.check_result(res, cols, expect)
        G.assert_column_parity(res, 0, cols[0], streams, n, 8192, compression=compression, what=("C3", compression))
        res.free()


def test_c2_direct_delta_stripes():
    """BASELINE config C2 as bench.py runs it (`--workload c2`): one DIRECT stripe (40-bit values, byte-aligned 512-value
    runs) and one DELTA stripe (8-bit deltas) in ONE call, > 1 M rows each, against the generator's values and the oracle."""
    stripes = [W.c2_stripe(1_200_000, 0, "direct"), W.c2_stripe(1_100_003, 1, "delta", base=7)]
    assert stripes[0][4]["direct"] == (1_200_000 + 511) // 512 and stripes[1][4]["delta"] == (1_100_003 + 511) // 512
    results = decode_all([s[:4] for s in stripes], "none")
    for (n, cols, streams, expect, _), res, kind in zip(stripes, results, ("direct", "delta")):
        assert res.status()[0] == 0, res.status()
        assert res.rows == n and res.n_batches == (n + 8191) // 8192
        W.check_result(res, cols, expect)
        G.assert_column_parity(res, 0, cols[0], streams, n, 8192, what=("C2", kind))
        res.free()


def test_c2_adversarial_walk():
    """Run lengths 200..511 with the width changing from run to run and PATCHED_BASE runs in between: no stride guess of the run
    walk holds, the verify + repair kernels carry the stream (rle_scan.hip).  Decoded values = generated values = oracle."""
    n, cols, streams, expect, stats = W.c2_adversarial_stripe(1_500_000, 3)
    assert stats["patched_base"] > 500 and stats["direct"] > 2000
    # (twice: a context that meets such a stream for the first time settles it with the serial repair, the calls behind it with the
    # exact parallel walk -- orcgpu_ctx::exact_on --; both are exact)
    for turn in range(3):
        res = decode_all([(n, cols, streams, expect)], "none")[0]
        assert res.status()[0] == 0, res.status()
        W.check_result(res, cols, expect)
        G.assert_column_parity(res, 0, cols[0], streams, n, 8192, what=("c2-adv", turn))
        res.free()


def test_c2_rowgroup_flushes():
    """DIRECT runs with the encoder flushed every 10 000 rows (what a writer with a row index produces): the run stride breaks at
    every row-group boundary; the damaged stretches are mended in parallel (rle_mend_kernel), the result is exact."""
    n, cols, streams, expect, stats = W.c2_rowgroup_stripe(2_000_000, 1)
    assert stats["direct"] == 200 * 20
    res = decode_all([(n, cols, streams, expect)], "none")[0]
    assert res.status()[0] == 0, res.status()
    W.check_result(res, cols, expect)
    G.assert_column_parity(res, 0, cols[0], streams, n, 8192, what="c2-rowgroup")
    res.free()


def test_zstd_chunks_put_off_by_the_streaming_pass_are_taken_again(monkeypatch):
    """The Zstandard execution kernel runs beside the entropy kernel; a workgroup that waits too long for sequences puts its chunk
    off and a second launch behind the entropy kernel takes it.  With no patience at all (ORCGPU_EXEC_PATIENCE=0) that happens to
    every chunk whose sequences are not there at once: the result must not change."""
    monkeypatch.setenv("ORCGPU_EXEC_PATIENCE", "0")
    rows = 400_000
    table = W.lineitem_table(rows)
    stripe = W.lineitem_stripe(table, 0, rows, "zstd")
    res = decode_all([stripe], "zstd")[0]
    n, cols, streams, expect = stripe
    assert res.status()[0] == 0, res.status()
    W.check_result(res, cols, expect)
    res.free()


def test_dictionary_columns_into_results_used_before():
    """A result decoded into again keeps its character arena: the value bytes of its dictionary columns are then placed on the
    device without the host waiting for their totals (dict_place_kernel), and pass 2 is sent again when the arena proves too
    small.  The same results take: small stripes (fresh: the host places), larger ones (too small: again), smaller ones (fit),
    lineitem stripes (four dictionary columns each, arenas from a one-column stripe: again), the same once more (fit)."""
    c = G.ctx()
    table = W.lineitem_table(260_000)

    def li(a, b):
        return W.lineitem_stripe(table, a, b, "snappy")

    rounds = [
        [W.c3_stripe(40_000, 0, "snappy"), W.c3_stripe(9_000, 1, "snappy")],
        [W.c3_stripe(700_000, 2, "snappy"), W.c3_stripe(350_000, 3, "snappy")],
        [W.c3_stripe(100_001, 4, "snappy"), W.c3_stripe(8_192, 5, "snappy")],
        [li(0, 150_000), li(150_000, 260_000)],
        [li(150_000, 260_000), li(0, 150_000)],
    ]
    results = None
    for no, stripes in enumerate(rounds):
        staged = [c.stage(n, streams, cols, compression="snappy") for n, cols, streams, _ in stripes]
        results = c.decode(staged, results)
        for s in staged:
            s.free()
        for (n, cols, streams, expect), res in zip(stripes, results):
            assert res.status()[0] == 0, (no, res.status())
            W.check_result(res, cols, expect)
            for ci, cc in enumerate(cols):
                if cc.get("encoding") in (W.DICTIONARY_V2,):
                    G.assert_column_parity(res, ci, cc, streams, n, 8192, compression="snappy", what=("reused result", no, cc.get("name")))
    for res in results:
        res.free()
