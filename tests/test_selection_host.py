"""The C++ restatement of the reference's row-selection stepping (orcgpu_selection_batches: host only, no GPU) against the
Python restatement in tests/selection_model.py, on the patterns of the reference's tests/row_selection/main.rs and on
random selections."""
import numpy as np

import selection_model as M
from orc_rust_amd import capi


def check(selectors, stripe_rows, batch_size):
    got, rest = capi.selection_batches(selectors, stripe_rows, batch_size)
    mine, want_rest = M.split_off(M.normalise(selectors), stripe_rows)
    assert got == M.stripe_batches(mine, stripe_rows, batch_size), (selectors, stripe_rows, batch_size)
    assert rest == want_rest, (selectors, stripe_rows)


def test_reference_patterns():  # tests/row_selection/main.rs:47-372
    S, K = (lambda n: (n, False)), (lambda n: (n, True))
    for sel, rows in [([K(2), S(2), K(1)], 5), ([S(5)], 5), ([K(5)], 5), ([S(1), K(4)], 5), ([K(4), S(1)], 5),
                      ([S(1), K(1), S(1), K(1), S(1)], 5), ([K(1), S(2), K(2)], 5), ([S(2), K(2), S(1)], 5),
                      ([K(1000), S(500), K(8500)], 10000), ([], 5), ([K(10), S(20), K(34)], 64)]:
        for batch in (8192, 3, 1):
            check(sel, rows, batch)
    assert capi.selection_batches([K(2), S(2), K(1)], 5)[0] == [(2, 2)]
    assert capi.selection_batches([K(1000), S(500), K(8500)], 10000)[0] == [(1000, 500)]


def test_select_run_longer_than_a_batch_keeps_reading():
    # mod.rs:337-347 compares the rows of ONE step with the selector's row_count and never shortens the selector: a select run
    # longer than batch_size yields batches to the end of the stripe
    assert capi.selection_batches([(10, True), (20000, False), (70000, True)], 100000, 8192)[0] == M.stripe_batches(
        [(10, True), (20000, False), (70000, True)], 100000, 8192)
    assert len(capi.selection_batches([(20000, False), (80000, True)], 100000, 8192)[0]) == 13  # 12 x 8192 + 1696 = 100000 rows


def test_random_selections_and_stripe_chaining():
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(0, 12))
        sel = [(int(rng.integers(0, 40)), bool(rng.integers(0, 2))) for _ in range(n)]
        rows = int(rng.integers(1, 120))
        batch = int(rng.choice([1, 7, 16, 8192]))
        check(sel, rows, batch)
    # a file of five stripes: every stripe takes its share (split_off), the rest goes on to the next one
    sel = [(30, True), (50, False), (100, True), (25, False)]
    want = M.file_batches(sel, [64] * 5, 16)
    rest = sel
    for k in range(5):
        if sum(x[0] for x in M.normalise(rest)) == 0:
            assert want[k] is None  # selection used up: the stripe is read whole (arrow_reader.rs:296-308)
            continue
        got, rest = capi.selection_batches(rest, 64, 16)
        assert got == want[k], k
    assert want[3] == [(0, 13)] and want[4] is None
