"""GPU parity, differential fuzz: seeded random stripes (fuzz_gen.py) decoded through the C ABI and compared with the
oracle batch by batch -- values, validity, offsets, and for failing input the index and the kind of the first
failing batch.  The corrupted half overwrites one byte of one stream; the reference panics on some of that input
(chunk framing, rejected blocks), DESIGN.md section 2 says what both sides report there."""
import numpy as np
import pytest

import os

import fuzz_gen as F
import gpu_util as G

pytestmark = pytest.mark.gpu

# ORCGPU_FUZZ_CAMPAIGN=k (k = 1, 2, ...): the same tests over OTHER seeds -- every range below shifted by k * 10 000 -- for a
# campaign beside the suite; 0 (the suite): the ranges as they stand, so that a failure comes back.
SHIFT = 10_000 * int(os.environ.get("ORCGPU_FUZZ_CAMPAIGN", "0"))


def seeds(lo, hi):
    return range(lo + SHIFT, hi + SHIFT)

# seeds that once exposed a bug, kept forever:
#  200036 FLOAT short DATA behind nulls (error batch through the non-null index)
#  200066 corrupted dictionary LENGTH stream (dictionary offsets must not be followed)
#  200020 / 200093 PRESENT chunk rejected by the codec (swallowed; must not turn the column's error into BuildDecoder)
#  200040 / 200059 / 200063 chunk length beyond the stream (framing -> IoError)
#  200814 direct string DATA with broken framing
#  202495 / 205516 negative and huge string lengths in one batch (OffsetOverflow first)
#  204688 / 304091 / 305092 / 402035 / 302098 UTF-8 damage at batch boundaries, empty rows on a continuation byte
#  302488 truncated DELTA run whose first step overflows
#  308998 RLE error before a later framing error keeps its kind
#  404398 DECIMAL with an empty DATA stream behind leading nulls
#  200189 rejected chunk in a direct string LENGTH stream (garbage lengths must not be followed; run after others)
REGRESSIONS = [200036, 200066, 200020, 200093, 200040, 200059, 200063, 200814, 202495, 205516, 204688, 304091, 305092, 402035, 302098,
               302488, 308998, 404398, 200189]


def run_case(seed, corrupt):
    n, comp, block, batch, cols, streams, _ = F.make_case(seed, corrupt)
    res = G.gpu_decode(n, cols, streams, compression=comp, block_size=block, batch_size=batch)
    try:
        for ci, cc in enumerate(cols):
            G.assert_column_parity(res, ci, cc, streams, n, batch, compression=comp, block_size=block, what=(seed, cc["orc_type"], comp, block, batch, n))
    finally:
        res.free()
    return True


def test_valid_stripes():
    for seed in seeds(7_000_000, 7_000_300):
        run_case(seed, False)


def test_corrupted_stripes_fail_like_the_oracle():
    for seed in REGRESSIONS:
        run_case(seed, True)
    for seed in seeds(7_100_000, 7_100_600):
        run_case(seed, True)


def test_corrupted_stripes_over_stale_buffers():
    """The same workspace serves every decode of a context: what a failed stream leaves unwritten holds bytes of
    earlier decodes.  Replaying one stretch of seeds in order once crashed on offsets read from such bytes."""
    for seed in range(200150, 200192):
        run_case(seed, True)


# (seed, corrupted bytes): multi-hit cases that exposed a bug
#  615166/8 run that fails to parse right before speculative garbage near the end of the stream (kind of the first failure)
#  603070/8, 606892/8 negative string length and values past a rejected DATA chunk in one batch
#  609996/8 dictionary blob cut short whose first byte is no character start (blob error before UTF-8)
#  500147/3, 500259/3, 501380/3, 600168/8 dictionary construction errors vs key errors of batch 0
#  604291/8, 612714/8 value-level failure of a fixed DELTA run followed by a run that fails to parse
MULTI = [(615166, 8), (603070, 8), (606892, 8), (609996, 8), (500147, 3), (500259, 3), (501380, 3), (600168, 8), (604291, 8), (612714, 8)]


def run_stripe_case(seed, hits):
    n, comp, block, batch, cols, streams, _ = F.make_case(seed, True, hits)
    res = G.gpu_decode(n, cols, streams, compression=comp, block_size=block, batch_size=batch)
    try:
        G.assert_stripe_parity(res, cols, streams, n, batch, compression=comp, block_size=block, what=(seed, hits, comp, block, batch, n))
    finally:
        res.free()


def test_stripes_with_several_corrupted_bytes():
    for seed, hits in MULTI:
        run_stripe_case(seed, hits)
    for seed in seeds(7_200_000, 7_200_300):
        run_stripe_case(seed, 3)
    for seed in seeds(7_300_000, 7_300_300):
        run_stripe_case(seed, 8)
    for seed in seeds(7_400_000, 7_400_300):
        run_stripe_case(seed, -2)  # one overwritten byte and two streams cut short


def test_wide_type_family():
    """Binary / Varchar / Char / TimestampInstant columns, decimals of other precisions and scales."""
    for seed in seeds(7_600_000, 7_600_250):
        n, comp, block, batch, cols, streams, _ = F.make_case(seed, False, 1, True)
        res = G.gpu_decode(n, cols, streams, compression=comp, block_size=block, batch_size=batch)
        G.assert_stripe_parity(res, cols, streams, n, batch, compression=comp, block_size=block, what=(seed, comp, block, batch, n))
        res.free()
    # 3309319/4: string lengths whose i64 sum wraps below zero
    for seed, hits in [(3309319, 4)] + [(s, 1 + s % 4) for s in seeds(7_700_000, 7_700_300)]:
        n, comp, block, batch, cols, streams, _ = F.make_case(seed, True, hits, True)
        res = G.gpu_decode(n, cols, streams, compression=comp, block_size=block, batch_size=batch)
        try:
            G.assert_stripe_parity(res, cols, streams, n, batch, compression=comp, block_size=block, what=(seed, hits, comp, block, batch, n))
        finally:
            res.free()


def test_failing_stripes_do_not_disturb_their_neighbours():
    """Stripes of one orcgpu_decode_staged call share every launch (one job table, one summary): valid and
    corrupted stripes mixed in one call must each come out as they do alone."""
    ctx = G.ctx()
    for base in range(7_500_000 + SHIFT, 7_500_120 + SHIFT, 12):
        cases = [F.make_case(base + k, k % 3 == 1, 3) for k in range(12)]
        staged = [ctx.stage(n, streams, cols, compression=comp, block_size=block, batch_size=batch) for n, comp, block, batch, cols, streams, _ in cases]
        results = ctx.decode(staged)
        for (n, comp, block, batch, cols, streams, _), res in zip(cases, results):
            G.assert_stripe_parity(res, cols, streams, n, batch, compression=comp, block_size=block, what=(base, comp, block, batch, n))
        for r in results:
            r.free()
        for s in staged:
            s.free()
