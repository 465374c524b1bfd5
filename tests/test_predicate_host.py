"""orcgpu_predicate_row_groups (host only) against the known answers of the reference's own unit tests
(src/row_group_filter.rs:673-1490: the statistics, predicates and expected keep / skip of every test there, restated as
data) and of src/bloom_filter.rs:230-300.  The inputs are real protobuf bytes (tests/predicate_model.py)."""
import ctypes as C

import numpy as np
import pytest

import predicate_model as PM
from orc_rust_amd import capi
from orc_rust_amd.predicate import ColumnIndex, Predicate as P, PredicateValue as V


def evaluate(pred, columns, stripe_rows, rows_per_group=10000):
    """columns: {name: (row index bytes, bloom index bytes or None)} -> list of bools, or the error code."""
    L = capi.load()
    nodes, keep_alive = pred.flatten()
    arr = (ColumnIndex * max(1, len(columns)))()
    bufs = []
    for k, (name, (ri, bl)) in enumerate(columns.items()):
        nb = name.encode()
        bufs += [nb, ri, bl]
        arr[k].name = nb
        arr[k].row_index = C.cast(C.c_char_p(ri), C.c_void_p)
        arr[k].row_index_len = len(ri)
        if bl:
            arr[k].bloom_index = C.cast(C.c_char_p(bl), C.c_void_p)
            arr[k].bloom_index_len = len(bl)
    n_groups = (stripe_rows + rows_per_group - 1) // rows_per_group
    keep = np.full(max(1, n_groups), 7, dtype=np.uint8)
    ng = C.c_uint32(0)
    rc = L.orcgpu_predicate_row_groups(nodes, len(nodes), arr, len(columns), stripe_rows, rows_per_group, keep.ctypes.data, C.byref(ng))
    if rc:
        return rc
    assert ng.value == n_groups
    return [bool(x) for x in keep[:n_groups]]


def ints(*groups):
    """RowIndex of an integer column: groups = (number_of_values, has_null, min, max) or None."""
    return PM.row_index([None if g is None else PM.column_statistics(g[0], g[1], integer=(g[2], g[3])) if g[0] else PM.column_statistics(0, g[1])
                         for g in groups])


AGE = {"age": (ints((5000, False, 18, 25), (5000, False, 26, 65)), None)}  # create_test_row_index (row_group_filter.rs:574-631)


def test_reference_unit_tests_integer_comparisons():
    i32 = V.Int32
    assert evaluate(P.gt("age", i32(20)), AGE, 20000) == [True, True]          # :673
    assert evaluate(P.gte("age", i32(30)), AGE, 20000) == [False, True]        # :691
    assert evaluate(P.lt("age", i32(30)), AGE, 20000) == [True, True]          # :709
    assert evaluate(P.lte("age", i32(20)), AGE, 20000) == [True, False]        # :727
    assert evaluate(P.eq("age", i32(20)), AGE, 20000) == [True, False]         # :745
    assert evaluate(P.ne("age", i32(20)), AGE, 20000) == [True, True]          # :763
    single = {"age": (ints((1000, False, 20, 20)), None)}
    assert evaluate(P.ne("age", i32(20)), single, 10000) == [False]            # :784
    # every integer width is an integer value (row_group_filter.rs:207-219)
    for mk in (V.Int8, V.Int16, V.Int32, V.Int64):
        assert evaluate(P.eq("age", mk(20)), AGE, 20000) == [True, False]


def test_reference_unit_tests_combinations_and_nulls():
    i32 = V.Int32
    assert evaluate(P.and_([P.gte("age", i32(20)), P.lte("age", i32(30))]), AGE, 20000) == [True, True]   # :883
    assert evaluate(P.or_([P.lt("age", i32(20)), P.gt("age", i32(30))]), AGE, 20000) == [True, True]     # :905
    nulls = {"age": (ints((5000, True, 18, 25), (5000, False, 26, 65)), None)}
    assert evaluate(P.is_null("age"), nulls, 20000) == [True, False]                                    # :928
    empties = {"age": (ints((5000, True, 18, 25), (0, True, 0, 0)), None)}
    assert evaluate(P.is_not_null("age"), empties, 20000) == [True, False]                              # :983
    assert evaluate(P.gt("nonexistent", i32(10)), AGE, 20000) == 10                                     # :1034 Err(Unexpected)
    assert evaluate(P.gt("age", i32(10)), {}, 20000) == 10                                              # :1048
    assert evaluate(P.not_(P.is_null("age")), {"age": (ints((5000, True, 18, 25)), None)}, 10000) == [True]   # :1065
    assert evaluate(P.not_(P.is_not_null("age")), nulls, 20000) == [True, False]                        # :1101
    assert evaluate(P.not_(P.gt("age", i32(50))), {"age": (ints((10000, False, 10, 20)), None)}, 10000) == [True]   # :1158
    two = {"age": (ints((10000, False, 0, 10), (10000, False, 20, 30)), None)}
    assert evaluate(P.not_(P.and_([P.gte("age", i32(15)), P.lte("age", i32(25))])), two, 20000) == [True, True]     # :1193
    span = {"age": (ints((10000, False, 0, 5), (10000, False, 5, 15)), None)}
    assert evaluate(P.not_(P.or_([P.lt("age", i32(10)), P.gt("age", i32(30))])), span, 20000) == [False, True]      # :1254
    assert evaluate(P.not_(P.not_(P.gt("age", i32(15)))), {"age": (ints((10000, False, 10, 20)), None)}, 10000) == [True]   # :1312
    # a row group without typed statistics fails the evaluation (statistics.rs:118: number_of_values == 0 -> None; :196)
    assert evaluate(P.gt("age", i32(10)), empties, 20000) == 10
    # ... and so does a value of the wrong type, or a NULL literal
    assert evaluate(P.gt("age", V.Utf8("x")), AGE, 20000) == 10
    assert evaluate(P.gt("age", i32(None)), AGE, 20000) == 10
    # an entry without statistics keeps its row group
    assert evaluate(P.gt("age", i32(100)), {"age": (ints(None, (5000, False, 18, 25)), None)}, 20000) == [True, False]


def test_reference_unit_tests_bloom_filters():
    only10 = PM.bloom_index([PM.bloom_with([PM.hash_long(10)], 3, 2)])
    assert evaluate(P.eq("age", V.Int32(20)), {"age": (PM.row_index([None]), only10)}, 10000) == [False]    # :821
    assert evaluate(P.eq("age", V.Int32(10)), {"age": (PM.row_index([None]), only10)}, 10000) == [True]
    assert evaluate(P.gt("age", V.Int32(20)), {"age": (PM.row_index([None]), only10)}, 10000) == [True]     # only equality asks the filter
    has50 = PM.bloom_index([PM.bloom_with([PM.hash_long(50)], 3, 2)])
    assert evaluate(P.eq("age", V.Int32(50)), {"age": (ints((1000, False, 100, 200)), has50)}, 10000) == [False]   # :845 statistics first
    # strings: Murmur3 of the bytes, the utf8bitset form (bloom_filter.rs:28-48), numHashFunctions 0 -> 3
    words = PM.bloom_index([PM.bloom_with([PM.murmur3_64(b"alpha"), PM.murmur3_64(b"gamma")], 3, 4)], utf8=True)
    names = {"name": (PM.row_index([PM.column_statistics(100, False, string={"minimum": "alpha", "maximum": "gamma"})]), words)}
    assert evaluate(P.eq("name", V.Utf8("alpha")), names, 100) == [True]
    assert evaluate(P.eq("name", V.Utf8("beta")), names, 100) == [False]
    assert evaluate(P.eq("name", V.Utf8("a much longer key than eight bytes")), names, 100) == [False]
    # doubles hash their bit pattern; booleans 0 / 1
    scores = {"score": (PM.row_index([PM.column_statistics(100, False, double=(1.0, 3.0))]),
                        PM.bloom_index([PM.bloom_with([PM.hash_long(int.from_bytes(np.float64(1.0).tobytes(), "little", signed=True))], 3, 2)]))}
    assert evaluate(P.eq("score", V.Float64(1.0)), scores, 100) == [True]
    assert evaluate(P.eq("score", V.Float64(2.0)), scores, 100) == [False]
    assert evaluate(P.eq("score", V.Float32(1.0)), scores, 100) == [True]
    # a filter count that does not match the entries: no filters (the reference asserts)
    assert evaluate(P.eq("age", V.Int32(20)), {"age": (PM.row_index([None, None]), only10)}, 20000) == [True, True]


def test_reference_unit_tests_strings_and_other_kinds():
    def s(lo, up, op, v, exact_min=True, exact_max=True):   # evaluate_string_comparison's cases (:1352-1490)
        st = {("minimum" if exact_min else "lower_bound"): lo, ("maximum" if exact_max else "upper_bound"): up}
        return evaluate(P.comparison("s", op, V.Utf8(v)), {"s": (PM.row_index([PM.column_statistics(10, False, string=st)]), None)}, 10)[0]
    from orc_rust_amd.predicate import EQ, NE, LT, GT, LE, GE
    assert s("a", "c", EQ, "b") and not s("a", "c", EQ, "d") and s("a", "c", EQ, "a") and s("a", "c", EQ, "c")
    assert not s("a", "c", EQ, "a", exact_min=False) and not s("a", "c", EQ, "c", exact_max=False)
    assert s("a", "c", LT, "b") and not s("d", "e", LT, "b") and not s("a", "c", LT, "a") and not s("a", "c", LT, "a", exact_min=False)
    assert s("a", "c", GT, "b") and not s("a", "b", GT, "c") and not s("a", "c", GT, "c") and not s("a", "c", GT, "c", exact_max=False)
    assert s("a", "c", NE, "b") and not s("a", "a", NE, "a")
    assert s("a", "c", LE, "a") and not s("a", "c", LE, "a", exact_min=False) and s("a", "c", GE, "c") and not s("a", "c", GE, "c", exact_max=False)
    # doubles (epsilon on equality), dates, timestamps (the UTC pair), decimals (compared as strings), booleans (true counts)
    d = {"x": (PM.row_index([PM.column_statistics(10, False, double=(1.5, 2.5))]), None)}
    assert evaluate(P.eq("x", V.Float64(2.5 + 1e-10)), d, 10) == [True] and evaluate(P.gt("x", V.Float64(2.5)), d, 10) == [False]
    assert evaluate(P.eq("x", V.Int32(2)), d, 10) == 10
    day = {"x": (PM.row_index([PM.column_statistics(10, False, date=(19000, 19010))]), None)}
    assert evaluate(P.eq("x", V.Int32(19005)), day, 10) == [True] and evaluate(P.lt("x", V.Int64(19000)), day, 10) == [False]
    ts = {"x": (PM.row_index([PM.column_statistics(10, False, timestamp_utc=(1000, 2000))]), None)}
    assert evaluate(P.gte("x", V.Int64(2000)), ts, 10) == [True] and evaluate(P.gte("x", V.Int64(2001)), ts, 10) == [False]
    assert evaluate(P.gte("x", V.Int32(5)), ts, 10) == 10
    dec = {"x": (PM.row_index([PM.column_statistics(10, False, decimal=("10.5", "9.5"))]), None)}
    assert evaluate(P.eq("x", V.Utf8("5")), dec, 10) == [True] and evaluate(P.lt("x", V.Utf8("1")), dec, 10) == [False]
    b = {"x": (PM.row_index([PM.column_statistics(10, False, bucket=10), PM.column_statistics(10, False, bucket=0)]), None)}
    assert evaluate(P.eq("x", V.Boolean(True)), b, 20, 10) == [True, False] and evaluate(P.eq("x", V.Boolean(False)), b, 20, 10) == [False, True]
    assert evaluate(P.ne("x", V.Boolean(True)), b, 20, 10) == [False, True] and evaluate(P.lt("x", V.Boolean(True)), b, 20, 10) == [True, True]
    # binary statistics say nothing
    assert evaluate(P.eq("x", V.Utf8("q")), {"x": (PM.row_index([PM.column_statistics(10, False, binary_sum=99)]), None)}, 10) == [True]
    # fewer entries than row groups: the groups behind them stay kept; no row groups at all: an empty filter
    assert evaluate(P.gt("age", V.Int32(100)), AGE, 40000) == [False, False, True, True]
    assert evaluate(P.and_([]), AGE, 0) == []
