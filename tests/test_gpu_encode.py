"""GPU ORC encode: orcgpu_encode_rle2_i64 -- Integer RLE v2 of an Int64 column, one wavefront per 512-value
run (device/rle_encode.hip; replaces RleV2Encoder::{write_slice, take_inner}, rle_v2/mod.rs:403-531).  The round trip is the
test: what the device encodes, the device DECODER (rle_expand.hip, through the C ABI) and the CPU oracle (the restatement of
the reference's decoders) read back value for value; and the stream uses the sub-encodings it claims to (headers parsed here)."""
import ctypes as C

import numpy as np
import pytest

import gpu_util as G

pytestmark = pytest.mark.gpu
LONG, DATA, DIRECT_V2 = 4, 1, 2


def encode(values, signed=True):
    v = np.ascontiguousarray(values, dtype=np.int64)
    c = G.ctx()
    n = C.c_uint64(0)
    rc = c.L.orcgpu_encode_rle2_i64(c.h, v.ctypes.data, v.size, 1 if signed else 0, None, 0, C.byref(n))  # the size first
    assert rc == 0, rc
    out = np.zeros(max(1, n.value), dtype=np.uint8)
    m = C.c_uint64(0)
    rc = c.L.orcgpu_encode_rle2_i64(c.h, v.ctypes.data, v.size, 1 if signed else 0, out.ctypes.data, out.size, C.byref(m))
    assert rc == 0 and m.value == n.value, (rc, m.value, n.value)
    return out[:n.value]


def run_kinds(stream):
    """sub-encoding and length of every run of an RLE v2 stream (headers only: rle_v2/mod.rs:112-146)"""
    widths = list(range(1, 25)) + [26, 28, 30, 32, 40, 48, 56, 64]
    b, p, out = bytes(stream), 0, []
    while p < len(b):
        kind = b[p] >> 6
        if kind == 0:
            w, ln = ((b[p] >> 3) & 7) + 1, (b[p] & 7) + 3
            out.append(("short_repeat", ln))
            p += 1 + w
        elif kind == 1:
            w, ln = widths[(b[p] >> 1) & 31], ((b[p] & 1) << 8 | b[p + 1]) + 1
            out.append(("direct", ln))
            p += 2 + (ln * w + 7) // 8
        elif kind == 3:
            code, ln = (b[p] >> 1) & 31, ((b[p] & 1) << 8 | b[p + 1]) + 1
            assert code == 0  # the encoder only writes fixed steps
            p += 2
            for _ in range(2):
                while b[p] & 0x80:
                    p += 1
                p += 1
            out.append(("delta", ln))
        else:
            raise AssertionError("PATCHED_BASE is never written")
    assert p == len(b)
    return out


def round_trip(values, signed=True, what=""):
    values = np.ascontiguousarray(values, dtype=np.int64)
    stream = encode(values, signed)
    kinds = run_kinds(stream)
    assert sum(n for _, n in kinds) == values.size, what
    col = {"column_id": 1, "orc_type": LONG, "encoding": DIRECT_V2}
    if not signed:
        # an unsigned stream (what dictionary keys and lengths are) is read back by the oracle's stream decoder
        import oracle_lib as O
        st, got = O.int_rle(bytes(stream), values.size, version=2, signed=False)
        assert st == 0 and np.array_equal(got, values), what
        return kinds
    streams = [(1, DATA, stream)]
    res = G.gpu_decode(values.size, [col], streams)
    try:
        assert res.status()[0] == 0, (what, res.status())
        G.assert_column_parity(res, 0, col, streams, values.size, 8192, what=what)  # device decoder == oracle, batch by batch
        got = np.concatenate([np.frombuffer(res.batch(b, 0)["values"], dtype=np.int64) for b in range(res.n_batches)]) if values.size else values
        assert np.array_equal(got, values), what
    finally:
        res.free()
    return kinds


@pytest.mark.parametrize("n", [1, 2, 3, 10, 11, 511, 512, 513, 1024, 70_001])
def test_random_values_of_every_width(n):
    rng = np.random.default_rng(n)
    for bits in (1, 2, 7, 8, 13, 24, 25, 31, 32, 33, 47, 56, 63):
        v = rng.integers(-(1 << (bits - 1)) if bits > 1 else -1, (1 << (bits - 1)) if bits > 1 else 1, n, dtype=np.int64)
        kinds = round_trip(v, what=("random", n, bits))
        assert all(k in ("direct", "delta", "short_repeat") for k, _ in kinds)
    extremes = rng.choice(np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, -1, 0, 1], dtype=np.int64), n)
    round_trip(extremes, what=("extremes", n))


def test_progressions_become_delta_runs_and_repeats_short_repeats():
    n = 5000
    kinds = round_trip(np.arange(n, dtype=np.int64) * 7 - 1234, what="rising")
    assert {k for k, _ in kinds} == {"delta"} and [ln for _, ln in kinds][:2] == [512, 512]
    kinds = round_trip(-(np.arange(n, dtype=np.int64) * 3) + 99, what="falling")
    assert {k for k, _ in kinds} == {"delta"}
    kinds = round_trip(np.full(n, -42, dtype=np.int64), what="constant")          # long repeats: a fixed step of zero
    assert {k for k, _ in kinds} == {"delta"}
    for ln in range(3, 11):                                                          # 3..10 equal values: SHORT_REPEAT
        kinds = round_trip(np.full(ln, 1 << 40, dtype=np.int64), what=("repeat", ln))
        assert kinds == [("short_repeat", ln)]
    # a progression whose step does not fit int64 is not a DELTA run
    big = np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, np.iinfo(np.int64).min], dtype=np.int64)
    assert round_trip(big, what="overflowing step") == [("direct", 3)]
    # mixed: runs of different kinds side by side
    rng = np.random.default_rng(5)
    v = np.concatenate([np.arange(512) * 2, rng.integers(0, 1000, 512), np.full(512, 9), rng.integers(-2**50, 2**50, 300), np.full(7, 3)]).astype(np.int64)
    kinds = round_trip(v, what="mixed")
    assert [k for k, _ in kinds] == ["delta", "direct", "delta", "direct"] or [k for k, _ in kinds][:3] == ["delta", "direct", "delta"]


def test_unsigned_streams_and_the_size_query():
    rng = np.random.default_rng(8)
    keys = rng.integers(0, 7, 100_000).astype(np.int64)          # dictionary keys: 3 bits per value
    stream = encode(keys, signed=False)
    assert len(stream) <= 100_000 * 3 // 8 + 2 * 196 + 16
    round_trip(keys, signed=False, what="keys")
    c = G.ctx()
    n = C.c_uint64(0)
    small = np.zeros(4, dtype=np.uint8)
    rc = c.L.orcgpu_encode_rle2_i64(c.h, keys.ctypes.data, keys.size, 0, small.ctypes.data, small.size, C.byref(n))
    assert rc == 101 and n.value == len(stream)                   # too small: INVALID_ARGUMENT, the size reported
    assert c.L.orcgpu_encode_rle2_i64(c.h, None, 0, 1, None, 0, C.byref(n)) == 0 and n.value == 0
