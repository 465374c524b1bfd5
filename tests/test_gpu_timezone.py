"""Writer time zones on the device: TIMESTAMP columns of a stripe whose footer names a zone are decoded against that zone's ORC
epoch and re-labelled to UTC (array_decoder/timestamp.rs:128-147, :236-291, :316-349), in every unit, with PRESENT streams,
and with values the conversion turns into nulls.  The oracle's zone table comes from Python's zoneinfo (tests/tz_table.py),
the library's from its own TZif reader."""
import ctypes as C

import numpy as np
import pytest

import gpu_util as G
import oracle_lib as O
import tz_table
from orc_rust_amd import gen

pytestmark = pytest.mark.gpu

ZONES = ["America/Los_Angeles", "Asia/Kolkata", "Australia/Lord_Howe", "Europe/London", "Asia/Tokyo", "GMT", "UTC"]


def ts_streams(rng, n, unit, nulls, extremes, base=1_420_070_400):
    per = [10**9, 10**6, 10**3, 1][unit]
    present = rng.random(n) > 0.2 if nulls else np.ones(n, dtype=bool)
    k = int(present.sum())
    # seconds relative to 2015: 1850 .. 2200 mostly, dense around the switch hours of 2021
    secs = rng.integers(-5_200_000_000, 5_800_000_000, k)
    secs[: k // 4] = rng.integers(194_000_000, 226_000_000, k // 4)
    if extremes and k > 64:
        idx = rng.choice(k, 48, replace=False)
        if unit == 3:
            # the edges of i64 nanoseconds: decodable, but adding a zone's offset leaves the range (timestamp_nanos_opt -> null)
            secs[idx[:16]] = 9_223_372_036 - base - rng.integers(1, 60_000, 16)
            secs[idx[16:32]] = -9_223_372_036 - base + rng.integers(1, 60_000, 16)
        else:
            # beyond chrono's years -262143 ..= 262142 (timestamp_micros(..).single() is None -> null), and far but inside
            secs[idx[:16]] = rng.integers(8_220_000_000_000, 9_000_000_000_000, 16)
            secs[idx[16:32]] = -rng.integers(8_340_000_000_000, 9_000_000_000_000, 16)
            secs[idx[32:]] = rng.integers(-2**42, 2**42, 16)
    nanos = (rng.integers(0, 10**9 // per, k) * per).astype(np.int64)
    # SECONDARY: nanoseconds with trailing zeros folded (timestamp.rs: decode_timestamp): value << 3 | (zeros - 1)
    enc = np.empty(k, dtype=np.int64)
    for i, v in enumerate(nanos):
        z = 0
        v = int(v)
        while v and v % 10 == 0 and z < 8:
            v //= 10
            z += 1
        if z >= 2:
            enc[i] = (v << 3) | (z - 1)
        else:
            enc[i] = int(nanos[i]) << 3
    streams = [(1, 1, gen.rle2(secs.astype(np.int64), signed=True)), (1, 5, gen.rle2(enc, signed=False))]
    if nulls:
        streams.append((1, 0, gen.boolean(present.astype(np.uint8))))
    return streams, present, secs, nanos


@pytest.mark.parametrize("zone", ZONES)
@pytest.mark.parametrize("unit", [0, 1, 2, 3])
@pytest.mark.parametrize("nulls", [False, True])
def test_timestamps_of_a_zone(zone, unit, nulls):
    rng = np.random.default_rng(unit * 100 + len(zone) + nulls)
    n = 20_000
    streams, _, _, _ = ts_streams(rng, n, unit, nulls, extremes=True, base=tz_table.orc_epoch(zone))
    col = {"column_id": 1, "orc_type": 9, "encoding": 2, "arrow_target": unit + 1, "name": "ts"}
    res = G.gpu_decode(n, [col], streams, batch_size=4096, writer_timezone=zone)
    G.assert_column_parity(res, 0, col, streams, n, 4096, ts_unit=unit, what=(zone, unit, nulls), writer_timezone=zone)
    if not nulls and zone not in ("GMT", "UTC"):
        # the extremes really are there: batches carry nulls although there is no PRESENT stream
        assert sum(res.batch(b, 0)["null_count"] for b in range(res.n_batches)) >= (32 if unit < 3 else 1)
    res.free()


def test_timestamp_instant_ignores_the_zone():
    rng = np.random.default_rng(5)
    n = 5000
    streams, _, _, _ = ts_streams(rng, n, 3, True, extremes=False)
    col = {"column_id": 1, "orc_type": 18, "encoding": 2, "name": "ts"}
    res = G.gpu_decode(n, [col], streams, writer_timezone="America/Los_Angeles")
    G.assert_column_parity(res, 0, col, streams, n, 8192, what="instant")  # oracle: no zone, UTC base
    res.free()


@pytest.mark.parametrize("zone", ["America/Los_Angeles", "Asia/Kolkata"])
def test_decimal128_target_of_a_zone(zone):
    rng = np.random.default_rng(11)
    n = 10_000
    streams, present, secs, nanos = ts_streams(rng, n, 3, True, extremes=False)
    col = {"column_id": 1, "orc_type": 9, "encoding": 2, "arrow_target": 20, "arrow_precision": 38, "arrow_scale": 9, "name": "ts"}
    res = G.gpu_decode(n, [col], streams, writer_timezone=zone)
    assert res.status()[0] == 0
    base = tz_table.orc_epoch(zone)
    at, offs, offs0, fold_at = tz_table.table(zone)
    want = []
    for s, ns in zip(secs.tolist(), nanos.tolist()):
        sse = s + base
        if sse < 0 and ns > 999_999:
            sse -= 1
        want.append(sse * 10**9 + ns)
    words = np.zeros(2 * len(want), dtype=np.uint64)
    for i, v in enumerate(want):
        words[2 * i] = v & (2**64 - 1)
        words[2 * i + 1] = (v >> 64) & (2**64 - 1)
    O.lib().oo_timestamp_decimals_to_utc(words.ctypes.data, len(want), at.ctypes.data, offs.ctypes.data, len(at), int(offs0), int(fold_at))
    dense = words.reshape(-1, 2)
    spaced = np.zeros((n, 2), dtype=np.uint64)
    spaced[present] = dense
    got = np.concatenate([np.frombuffer(res.batch(b, 0)["values"], dtype=np.uint64) for b in range(res.n_batches)]).reshape(-1, 2)
    assert np.array_equal(got, spaced)
    res.free()


def test_unknown_zone_is_refused():
    from orc_rust_amd import capi
    col = {"column_id": 1, "orc_type": 9, "encoding": 2, "name": "ts"}
    with pytest.raises(capi.OrcGpuError) as e:
        G.gpu_decode(1, [col], [(1, 1, gen.rle2(np.array([1], dtype=np.int64))), (1, 5, gen.rle2(np.array([0], dtype=np.int64), signed=False))],
                     writer_timezone="Mars/Olympus_Mons")
    assert e.value.code == 7
