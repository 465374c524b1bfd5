"""Writer time zones on the host: the library's reading of the tz database (TZif + POSIX footer rule) against Python's zoneinfo.

Reference: array_decoder/timestamp.rs:128-147 (ORC epoch in the writer's zone), :236-291 (re-labelling to UTC)."""
import datetime as dt
import zoneinfo

import zlib

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
    rng = np.random.default_rng(zlib.crc32(name.encode()) & 0xFFFF)  # (not hash(): randomised per process)
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


@pytest.mark.parametrize("name", ["America/Los_Angeles", "Europe/London", "Australia/Lord_Howe", "America/Sao_Paulo", "Asia/Tokyo"])
def test_rule_holds_for_ever(name):
    """Behind the year 2400 (the end of the expanded table) the footer's rule still applies: chrono-tz and zoneinfo evaluate it
    without end; here instants fold back by whole 400-year Gregorian cycles (orcgpu_tz.inc: TzTable::fold_at)."""
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.integers(13569465600, 253402300799 - 86400, 6000),     # 2400 .. 9999 (datetime's range)
                        np.arange(13569465600 - 86400 * 400, 13569465600 + 86400 * 800, 86400 // 2 + 17)])  # either side of the fold
    got, _ = capi.timezone_offsets(name, t)
    want = want_offsets(name, t)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (name, t[bad[:5]], got[bad[:5]], want[bad[:5]])


def _tzif(footer, std_off=-18000):
    """A minimal version-2 TZif file: no transitions, one local time type, the POSIX rule in the footer."""
    import struct
    block = b"TZif2" + b"\0" * 15 + struct.pack(">6I", 0, 0, 0, 0, 1, 4) + struct.pack(">iBB", std_off, 0, 0) + b"EST\0"
    return block + block + b"\n" + footer.encode() + b"\n"  # (the 32-bit block, the 64-bit block, the footer)


@pytest.mark.parametrize("footer", ["EST5EDT,J60/2,J300/2", "EST5EDT,59/2:30,299/1", "EST5EDT4,M3.2.0,M11.1.0", "<+03>-3<+04>-4,J100,J200/0"])
def test_day_of_year_rule_forms(tmp_path, monkeypatch, footer):
    """The Jn (1..365, 29 February never counted) and n (0..365) forms of a POSIX rule, which no current zone uses but the format allows."""
    zdir = tmp_path / "zoneinfo"
    (zdir / "Test").mkdir(parents=True)
    path = zdir / "Test" / "Zone"
    path.write_bytes(_tzif(footer, std_off=-18000 if footer.startswith("EST") else 10800))
    monkeypatch.setenv("TZDIR", str(zdir))
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.integers(0, 4102444800, 6000), rng.integers(13569465600, 20000000000, 1000)])
    got, _ = capi.timezone_offsets("Test/Zone", t)
    epoch = dt.datetime(1970, 1, 1, tzinfo=UTC)
    if footer == "EST5EDT,59/2:30,299/1":
        # the zero-based form counts leap days (POSIX: day 0 is 1 January, 29 February can be named): day 59 is 1 March, or 29
        # February in a leap year.  (zoneinfo of Python 3.10 is one day early for this form -- a CPython bug fixed later --, so
        # the expectation is spelled out here: daylight time from day 59 02:30 EST to day 299 01:00 EDT.)
        def off(x):
            y = (epoch + dt.timedelta(seconds=int(x))).year
            start = dt.datetime(y, 1, 1, tzinfo=UTC) + dt.timedelta(days=59, hours=2, minutes=30) + dt.timedelta(hours=5)
            end = dt.datetime(y, 1, 1, tzinfo=UTC) + dt.timedelta(days=299, hours=1) + dt.timedelta(hours=4)
            return -14400 if start.timestamp() <= x < end.timestamp() else -18000
        want = np.array([off(x) for x in t], dtype=np.int32)
    else:
        with open(path, "rb") as f:
            z = zoneinfo.ZoneInfo.from_file(f)
        want = np.array([int((epoch + dt.timedelta(seconds=int(x))).astimezone(z).utcoffset().total_seconds()) for x in t], dtype=np.int32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (footer, t[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_unreadable_rule_is_refused_not_guessed(tmp_path, monkeypatch):
    zdir = tmp_path / "zoneinfo"
    zdir.mkdir()
    (zdir / "Odd").write_bytes(_tzif("EST5EDT,Q3.2.0,M11.1.0"))
    monkeypatch.setenv("TZDIR", str(zdir))
    with pytest.raises(capi.OrcGpuError) as e:
        capi.timezone_offsets("Odd", [0])
    assert e.value.code == 7


@pytest.mark.parametrize("name", ["UTC", "GMT", "Etc/UTC", "Etc/GMT", "Zulu", "Etc/Universal"])
def test_utc_family_needs_no_database(tmp_path, monkeypatch, name):
    """Java writers always name a zone, usually UTC: it must stage on a host without tzdata (slim containers)."""
    monkeypatch.setenv("TZDIR", str(tmp_path))  # an empty directory first in the search
    got, epoch = capi.timezone_offsets(name, [0, 1700000000, -5000000000])
    assert not got.any() and epoch == 1420070400
