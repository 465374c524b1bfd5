"""GPU parity, Zstandard at table scale (device/zstd_lanes.h: sequences by four lanes per block, sixteen blocks per
wavefront; zstd_literals_kernel; lz_exec_wave_kernel).  The library takes that path by itself only for calls with tens of
millions of sequences (bench.py's table); ORCGPU_ZSTD_LANES=1 forces it, so the same inputs the wavefront-per-block
kernels are tested with go through it here: real encoder output over the ten data shapes at three block sizes (one
chain .. 600 chains per call, ragged last wavefront, blocks with fewer than eight sequences), lineitem stripes against
the oracle and the generator's expectations, and the differential fuzz (valid and corrupted Zstandard stripes: same
failing batch, same error kind as the oracle)."""
import numpy as np
import pytest

import fuzz_gen as F
import gpu_util as G
from orc_rust_amd.gen import workloads as W
from test_gpu_codecs import CODECS, DATA, DOUBLE, frame, shapes

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["beside", "in front"])
def lanes(request, monkeypatch):
    """The table-scale path forced on; the literals kernel beside the sequences kernel (second stream) or in front of it."""
    monkeypatch.setenv("ORCGPU_ZSTD_LANES", "1")
    monkeypatch.setenv("ORCGPU_ZSTD_LIT_ASIDE", "1" if request.param == "beside" else "0")
    return request.param


@pytest.mark.parametrize("block", [262144, 65536, 1000])
def test_real_encoder_over_data_shapes(lanes, block):
    for name, raw in shapes(block).items():
        c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
        st = [(1, DATA, frame(raw, CODECS["zstd"], block))]
        n = len(raw) // 8
        res = G.gpu_decode(n, [c], st, compression="zstd", block_size=block)
        assert res.status()[0] == 0, (block, name, res.status())
        G.assert_column_parity(res, 0, c, st, n, 8192, compression="zstd", block_size=block, what=("lanes", lanes, block, name))
        res.free()


def test_all_columns_of_one_call_share_the_wavefronts(lanes):
    """Every shape as a column of ONE stripe: chains of very different lengths side by side in a wavefront."""
    block = 4096
    cols, streams, rows = [], [], []
    for name, raw in shapes(7).items():
        cid = len(cols) + 1
        cols.append({"column_id": cid, "orc_type": DOUBLE, "encoding": 0})
        streams.append((cid, DATA, frame(raw, CODECS["zstd"], block)))
        rows.append(len(raw) // 8)
    n = min(r for r in rows if r)
    keep = [i for i, r in enumerate(rows) if r >= n and r]
    cols, streams = [cols[i] for i in keep], [streams[i] for i in keep]
    res = G.gpu_decode(n, cols, streams, compression="zstd", block_size=block)
    for ci, c in enumerate(cols):
        G.assert_column_parity(res, ci, c, streams, n, 8192, compression="zstd", block_size=block, what=("lanes one call", lanes, ci))
    res.free()


def test_lineitem_stripes(lanes):
    rows = 420_000
    table = W.lineitem_table(rows)
    stripes = [W.lineitem_stripe(table, 0, 300_000, "zstd"), W.lineitem_stripe(table, 300_000, rows, "zstd")]
    c = G.ctx()
    staged = [c.stage(n, streams, cols, compression="zstd") for n, cols, streams, _ in stripes]
    results = c.decode(staged)
    for s in staged:
        s.free()
    for (n, cols, streams, expect), res in zip(stripes, results):
        assert res.status()[0] == 0, res.status()
        W.check_result(res, cols, expect)
        for ci, cc in enumerate(cols):
            G.assert_column_parity(res, ci, cc, streams, n, 8192, compression="zstd", what=("lanes C4", lanes, cc["name"]))
        res.free()


def zstd_cases(first, count, corrupt):
    out, seed = [], first
    while len(out) < count:
        if F.make_case(seed, corrupt)[1] == "zstd":
            out.append(seed)
        seed += 1
    return out


def test_differential_fuzz(lanes):
    for corrupt, first in ((False, 7_000_000), (True, 7_100_000)):
        for seed in zstd_cases(first, 60, corrupt):
            n, comp, block, batch, cols, streams, _ = F.make_case(seed, corrupt)
            res = G.gpu_decode(n, cols, streams, compression=comp, block_size=block, batch_size=batch)
            try:
                for ci, cc in enumerate(cols):
                    G.assert_column_parity(res, ci, cc, streams, n, batch, compression=comp, block_size=block, what=("lanes", lanes, seed, corrupt))
            finally:
                res.free()


def test_corrupted_sequence_sections_fail_like_the_oracle(lanes):
    """One byte of the SEQUENCES section of a block overwritten (table descriptions, the bit stream's last byte, its middle):
    the chunk is rejected or decodes to the same bytes as in the oracle."""
    rng = np.random.default_rng(11)
    text = shapes(3)["text"][:160000]
    good = frame(text, CODECS["zstd"], 65536)
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    n = len(text) // 8
    for trial in range(40):
        bad = good.copy()
        # the sequences section ends the chunk: hit its tail (bit stream) or somewhere behind the literals
        hdr = int(bad[0]) | int(bad[1]) << 8 | int(bad[2]) << 16
        clen = hdr >> 1
        pos = 3 + (clen - 1 - int(rng.integers(0, 4)) if trial % 2 == 0 else int(rng.integers(clen // 2, clen)))
        bad[pos] = rng.integers(0, 256)
        st = [(1, DATA, bad)]
        res = G.gpu_decode(n, [c], st, compression="zstd", block_size=65536)
        try:
            G.assert_column_parity(res, 0, c, st, n, 8192, compression="zstd", block_size=65536, what=("lanes corrupt", lanes, trial, pos))
        finally:
            res.free()


# ---- the path bench.py's headline takes: several COLUMN LANES and the table-scale kernels together -------------------------
# (the planner takes it by itself only from 64 MiB staged / 40 M sequences on; ORCGPU_LANES is read at every call)
@pytest.mark.parametrize("n_lanes", [2, 3, 4])
def test_column_lanes_with_the_table_scale_kernels(monkeypatch, n_lanes):
    """Two lineitem stripes in ONE call, split over 2 / 3 / 4 column lanes, every lane through zstd_literals_kernel, the FSE table
    kernel, zstd_seq_quads_kernel (8-byte sequence records) and lz_exec_wave_kernel: every column of both stripes against the
    oracle, batch by batch, and against what the generator implies; the lanes' statistics add up to the call."""
    monkeypatch.setenv("ORCGPU_ZSTD_LANES", "1")
    monkeypatch.setenv("ORCGPU_LANES", str(n_lanes))
    rows = 260_000
    table = W.lineitem_table(rows)
    stripes = [W.lineitem_stripe(table, 0, 150_000, "zstd"), W.lineitem_stripe(table, 150_000, rows, "zstd")]
    c = G.ctx()
    staged = [c.stage(n, streams, cols, compression="zstd") for n, cols, streams, _ in stripes]
    results = c.decode(staged)
    stats = c.lane_stats()
    assert len(stats) == n_lanes and all(s["n_lanes"] == n_lanes for s in stats), stats
    assert sum(s["stream_bytes"] for s in stats) == sum(s.nbytes() for s in staged)
    assert sum(s["arrow_bytes"] for s in stats) == sum(r.arrow_bytes for r in results)
    assert all(s["seq_kernel_ms"] > 0 and s["exec_kernel_ms"] > 0 for s in stats), stats  # every lane ran the table-scale kernels
    for s in staged:
        s.free()
    for (n, cols, streams, expect), res in zip(stripes, results):
        assert res.status()[0] == 0, res.status()
        W.check_result(res, cols, expect)
        for ci, cc in enumerate(cols):
            G.assert_column_parity(res, ci, cc, streams, n, 8192, compression="zstd", what=("lanes x table scale", n_lanes, cc["name"]))
        res.free()


def test_lane_count_does_not_change_a_byte(monkeypatch):
    """The same stripe through 1, 2 and 3 lanes, table-scale kernels on and off: identical Arrow buffers."""
    rows = 120_000
    table = W.lineitem_table(rows)
    n, cols, streams, expect = W.lineitem_stripe(table, 0, rows, "zstd")
    ref = None
    for zl in ("0", "1"):
        for nl in ("1", "2", "3"):
            monkeypatch.setenv("ORCGPU_ZSTD_LANES", zl)
            monkeypatch.setenv("ORCGPU_LANES", nl)
            res = G.gpu_decode(n, cols, streams, compression="zstd")
            assert res.status()[0] == 0, (zl, nl, res.status())
            got = [[res.batch(b, ci) for ci in range(len(cols))] for b in range(res.n_batches)]
            res.free()
            if ref is None:
                ref = got
                continue
            for b, (rb, gb) in enumerate(zip(ref, got)):
                for ci, (r1, g1) in enumerate(zip(rb, gb)):
                    assert r1["values"] == g1["values"] and r1["validity"] == g1["validity"] and r1["null_count"] == g1["null_count"], (zl, nl, b, cols[ci]["name"])
                    assert (r1["offsets"] is None) == (g1["offsets"] is None) and (r1["offsets"] is None or np.array_equal(r1["offsets"], g1["offsets"])), (zl, nl, b, ci)


def test_offset_values_beyond_the_record_format_are_rejected(lanes):
    """zseq_pack keeps 29 bits of an offset value: a sequence whose offset code says 2^30 is a corrupt chunk for the oracle (the
    match reaches before the frame) and for the GPU path (the record holds 2^29 - 1: just as far out of reach)."""
    # a Compressed_Block by hand: raw literals "abcd", ONE sequence {ll 4, ml 3, offset code 30 with 30 zero extra bits}; all three
    # tables in RLE_Mode (one symbol each, no state bits): the bit stream is the extra bits alone + the closing 1 bit
    lit = b"abcd"
    seq = bytes([1, 0b01010100, 4, 30, 0])  # 1 sequence; modes LL=RLE, OF=RLE, ML=RLE; symbols LL 4, OF 30, ML 0
    bits = bytes([0, 0, 0, 0b01000000])     # 30 extra bits of the offset (zero), then the final-bit marker
    body = bytes([len(lit) << 3]) + lit + seq + bits
    bh = (len(body) << 3) | (2 << 1) | 1
    frame_bytes = b"\x28\xb5\x2f\xfd" + bytes([0x20, 7]) + bytes([bh & 255, (bh >> 8) & 255, bh >> 16]) + body  # single segment, content size 7
    chunk = bytes([(len(frame_bytes) << 1) & 255, (len(frame_bytes) >> 7) & 255, (len(frame_bytes) >> 15) & 255]) + frame_bytes
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    st = [(1, DATA, np.frombuffer(chunk, dtype=np.uint8))]
    res = G.gpu_decode(1, [c], st, compression="zstd", block_size=65536)
    try:
        assert res.status()[0] != 0
        G.assert_column_parity(res, 0, c, st, 1, 8192, compression="zstd", block_size=65536, what=("offset code 30", lanes))
    finally:
        res.free()


# ---- content checksums (RFC 8878 3.1.1: XXH64 of the frame's content, low 32 bits) ------------------------------------------
@pytest.mark.parametrize("table_scale", ["0", "1"])
def test_content_checksums_are_verified(monkeypatch, table_scale):
    """libzstd behind the reference's zstd crate verifies a frame's checksum when the header flags one (compression.rs:151-159):
    frames carrying the right checksum decode to the oracle's bytes; ONE chunk with a wrong checksum fails its batch like the
    oracle (BuildDecoder), and the batches before it are intact.  Both Zstandard paths (one wavefront / one lane per block)."""
    from test_oracle_codecs import with_checksum
    import pyarrow as pa
    monkeypatch.setenv("ORCGPU_ZSTD_LANES", table_scale)
    block = 65536
    cd = pa.Codec("zstd")
    for name, raw in shapes(5).items():
        raw = raw[:block * 5 + 1234]
        for bad_chunk in (None, 0, 3):
            def comp(blk, state={"k": 0}):
                k = state["k"]
                state["k"] += 1
                c = cd.compress(blk, asbytes=True)
                return with_checksum(c, blk, wrong=(bad_chunk is not None and k == bad_chunk))
            st = [(1, DATA, frame(raw, comp, block))]
            c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
            n = len(raw) // 8
            res = G.gpu_decode(n, [c], st, compression="zstd", block_size=block)
            try:
                if bad_chunk is None:
                    assert res.status()[0] == 0, (name, res.status())
                G.assert_column_parity(res, 0, c, st, n, 8192, compression="zstd", block_size=block, what=("checksum", table_scale, name, bad_chunk))
            finally:
                res.free()
