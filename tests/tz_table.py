"""Transition tables of IANA zones for the ORACLE, derived from Python's zoneinfo alone (not from the library's TZif reader):
the zone is sampled day by day from 1800 to 2900 and every change of offset is bisected to the second.  A zone that still
switches in its last sampled years ends in a daylight-saving RULE (zoneinfo, like chrono-tz, applies it without end): the
oracle then looks instants behind 2900 up whole 400-year cycles earlier (fold_at; the years 2500..2900 are governed by the
rule alone, and the Gregorian calendar repeats after 400 years)."""
import datetime as dt
import functools
import zoneinfo

import numpy as np

_EPOCH = dt.datetime(1970, 1, 1, tzinfo=dt.timezone.utc)
UTC_ZONES = (None, "UTC", "GMT", "Etc/UTC", "Etc/GMT")


@functools.lru_cache(maxsize=None)
def table(name):
    """(at int64[], offs int32[], offs0, fold_at): offs[i] holds from UTC instant at[i] on, offs0 before at[0]; fold_at: see above
    (2 ** 63 - 1 for a zone whose last offset holds for ever)."""
    z = zoneinfo.ZoneInfo(name)

    def off(t):
        return int((_EPOCH + dt.timedelta(seconds=t)).astimezone(z).utcoffset().total_seconds())
    day = 86400
    t0 = -5364662400  # 1800-01-01
    t1 = 29348006400 + day  # 2900-01-02
    prev = offs0 = off(t0)
    at, offs = [], []
    for t in range(t0 + day, t1, day):
        cur = off(t)
        if cur != prev:
            lo, hi = t - day, t
            base = prev
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if off(mid) == base:
                    lo = mid
                else:
                    hi = mid
            at.append(hi)
            offs.append(off(hi))
            if offs[-1] != cur:  # two changes within the day: find the second one too
                lo2, hi2 = hi, t
                while hi2 - lo2 > 1:
                    mid = (lo2 + hi2) // 2
                    if off(mid) == offs[-1]:
                        lo2 = mid
                    else:
                        hi2 = mid
                at.append(hi2)
                offs.append(cur)
        prev = cur
    ruled = bool(at) and at[-1] > t1 - 400 * day
    return np.array(at, dtype=np.int64), np.array(offs, dtype=np.int32), offs0, (t1 if ruled else 2 ** 63 - 1)


def orc_epoch(name):
    """2015-01-01T00:00:00 in the zone, as seconds since the UNIX epoch (array_decoder/timestamp.rs:133-147)."""
    if name in UTC_ZONES:
        return 1420070400
    return int(dt.datetime(2015, 1, 1, tzinfo=zoneinfo.ZoneInfo(name)).timestamp())
