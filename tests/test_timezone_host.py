"""Writer time zones on the host: the library's reading of the tz database (TZif + POSIX footer rule) against Python's zoneinfo.

Reference: array_decoder/timestamp.rs:128-147 (ORC epoch in the writer's zone), :236-291 (re-labelling to UTC)."""
import datetime as dt
import zoneinfo

import numpy as np
import pytest

from orc_rust_amd import capi

ZONES = ["America/Los_Angeles", "US/Pacific", "Europe/London", "Asia/Kolkata", "Australia/Lord_Howe", "America/St_Johns", "Asia/Kathmandu",
         "Africa/Casablanca", "Europe/Dublin", "America/Sao_Paulo", "Pacific/Apia", "Asia/Tehran", "Antarctica/Troll", "UTC", "GMT", "Etc/GMT+5",
         "Asia/Tokyo", "America/Argentina/Buenos_Aires", "Pacific/Chatham"]
UTC = dt.timezone.utc


def want_offsets(name, instants):
    z = zoneinfo.ZoneInfo(name)
    epoch = dt.datetime(1970, 1, 1, tzinfo=UTC)
    return np.array([int((epoch + dt.timedelta(seconds=int(t))).astimezone(z).utcoffset().total_seconds()) for t in instants], dtype=np.int32)


@pytest.mark.parametrize("name", ZONES)
def test_offsets_match_zoneinfo(name):
    rng = np.random.default_rng(hash(name) & 0xFFFF)
    # 1850 .. 2399, dense around today's rules and sparse elsewhere, plus the hours around each year's usual switch dates
    t = np.concatenate([rng.integers(-3786825600, 13569465600, 4000), rng.integers(0, 2524608000, 4000)])
    got, epoch = capi.timezone_offsets(name, t)
    want = want_offsets(name, t)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (name, t[bad[:5]], got[bad[:5]], want[bad[:5]])
    z = zoneinfo.ZoneInfo(name)
    assert epoch == int(dt.datetime(2015, 1, 1, tzinfo=z).timestamp())


@pytest.mark.parametrize("name", ["America/Los_Angeles", "Europe/London", "Australia/Lord_Howe", "Africa/Casablanca"])
def test_offsets_at_every_transition_second(name):
    """One second either side of every switch between 1900 and 2200: found by scanning zoneinfo day by day and bisecting."""
    z = zoneinfo.ZoneInfo(name)
    epoch = dt.datetime(1970, 1, 1, tzinfo=UTC)

    def off(t):
        return (epoch + dt.timedelta(seconds=int(t))).astimezone(z).utcoffset().total_seconds()
    edges = []
    day = 86400
    t0 = -2208988800
    prev = off(t0)
    for t in range(t0 + day, 7258118400, day):
        cur = off(t)
        if cur != prev:
            lo, hi = t - day, t
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if off(mid) == prev:
                    lo = mid
                else:
                    hi = mid
            edges += [lo - 1, lo, hi, hi + 1]
        prev = cur
    assert len(edges) > 100
    got, _ = capi.timezone_offsets(name, np.array(edges))
    assert np.array_equal(got, want_offsets(name, edges))


def test_unknown_zone_is_refused():
    with pytest.raises(capi.OrcGpuError) as e:
        capi.timezone_offsets("Mars/Olympus_Mons", [0])
    assert e.value.code == 7
    with pytest.raises(capi.OrcGpuError):
        capi.timezone_offsets("../../etc/passwd", [0])
