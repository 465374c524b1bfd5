"""Shared helpers of the GPU parity tests: drive the HIP path through the C ABI
(orc_rust_amd.capi -> liborcgpu.so) and compare it batch by batch, byte by byte, with the CPU
oracle on the same streams."""
import numpy as np

import oracle_lib as O
from orc_rust_amd import capi

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def gpu_decode(n_rows, columns, streams, compression="none", block_size=262144, batch_size=8192, ts_base=0, writer_timezone=None):
    """columns: [dict(column_id, orc_type, encoding, ...)], streams: [(column_id, kind, bytes)]"""
    c = ctx()
    staged = c.stage(n_rows, streams, columns, compression=compression, block_size=block_size, batch_size=batch_size, ts_base=ts_base,
                     writer_timezone=writer_timezone)
    res = c.decode([staged])[0]
    staged.free()
    return res


def oracle_column(col, streams, compression="none", block_size=262144, ts_unit=3, ts_base=1420070400, writer_timezone=None):
    sd = {k: (b.tobytes() if isinstance(b, np.ndarray) else bytes(b)) for cid, k, b in streams if cid == col["column_id"]}
    tz = None
    if writer_timezone and col["orc_type"] == 9:
        # Stripe::writer_tz -> base epoch in that zone + TimestampOffsetArrayDecoder (timestamp.rs:128-147, :236-291)
        import tz_table
        tz = tz_table.table(writer_timezone)
        ts_base = tz_table.orc_epoch(writer_timezone)
    return O.Column(col["orc_type"], col.get("encoding", 2), sd, dictionary_size=col.get("dictionary_size", 0),
                    precision=col.get("precision", 0), scale=col.get("scale", 0), ts_unit=ts_unit, ts_base=ts_base,
                    compression=compression, block_size=block_size, tz=tz)


def assert_column_parity(res, ci, col, streams, n_rows, batch_size, compression="none", block_size=262144, ts_unit=3,
                         ts_base=1420070400, what="", writer_timezone=None):
    """The reference yields Ok batches until the first failing one (arrow_reader.rs:333-346): the
    GPU result must agree on every Ok batch bit for bit and on the index + kind of the failure."""
    oc = oracle_column(col, streams, compression, block_size, ts_unit, ts_base, writer_timezone)
    gst, gbatch, gcol = res.status()
    left = n_rows
    b = 0
    if oc.status != O.OK:
        assert gst == oc.status, (what, "construction error", gst, oc.status)
        return
    while left > 0:
        n = min(batch_size, left)
        ob = oc.next_batch(n)
        if ob["status"] != O.OK:
            assert gst != 0 and gcol == ci and gbatch == b, (what, "oracle fails at batch", b, ob["status"], "gpu", gst, gbatch, gcol)
            assert gst == ob["status"], (what, "error kind", gst, ob["status"])
            return
        assert not (gst != 0 and gcol == ci and gbatch <= b), (what, "gpu fails at batch", gbatch, "code", gst, "oracle ok at", b)
        gb = res.batch(b, ci)
        assert gb["length"] == ob["length"], what
        assert gb["null_count"] == ob["null_count"], (what, b, gb["null_count"], ob["null_count"])
        assert (gb["validity"] is None) == (ob["validity"] is None), (what, b)
        if ob["validity"] is not None:
            assert gb["validity"] == ob["validity"], (what, "validity", b)
        if ob["offsets"] is not None:
            assert np.array_equal(gb["offsets"], ob["offsets"]), (what, "offsets", b)
        if gb["values"] != ob["values"]:
            ga = np.frombuffer(gb["values"], dtype=np.uint8)
            oa = np.frombuffer(ob["values"], dtype=np.uint8)
            m = min(ga.size, oa.size)
            bad = np.nonzero(ga[:m] != oa[:m])[0]
            raise AssertionError((what, "values differ in batch", b, "first byte", int(bad[0]) if bad.size else m, ga.size, oa.size))
        left -= n
        b += 1
    oc.close()


def assert_stripe_parity(res, cols, streams, n_rows, batch_size, compression="none", block_size=262144, what=""):
    """Stripe-level view for input on which several columns may fail: the reader ends at the first failing batch,
    and inside it at the first failing column (arrow_reader.rs:333-346 decodes a batch column by column).  The GPU
    result must name that (batch, column, kind), and agree bit for bit on every batch before it."""
    nb = (n_rows + batch_size - 1) // batch_size
    first = None  # (batch, column, status)
    for ci, cc in enumerate(cols):
        oc = oracle_column(cc, streams, compression, block_size)
        if oc.status != O.OK:
            first = min(first or (1 << 60, 0, 0), (0, ci, oc.status))
            continue
        left, b = n_rows, 0
        while left > 0:
            ob = oc.next_batch(min(batch_size, left))
            if ob["status"] != O.OK:
                first = min(first or (1 << 60, 0, 0), (b, ci, ob["status"]))
                break
            left -= batch_size
            b += 1
        oc.close()
    gst, gb, gc = res.status()
    if first is None:
        assert gst == 0, (what, "gpu fails, oracle does not", gst, gb, gc)
    else:
        assert (gb, gc, gst) == first, (what, "first failure (batch, column, kind): gpu", (gb, gc, gst), "oracle", first)
    for ci, cc in enumerate(cols):
        oc = oracle_column(cc, streams, compression, block_size)
        left = n_rows
        for b in range(nb if first is None else first[0]):
            ob = oc.next_batch(min(batch_size, left))
            left -= batch_size
            assert ob["status"] == O.OK
            g = res.batch(b, ci)
            assert g["null_count"] == ob["null_count"] and g["validity"] == ob["validity"], (what, "validity", ci, b)
            assert g["values"] == ob["values"], (what, "values", ci, b)
            if ob["offsets"] is not None:
                assert np.array_equal(g["offsets"], ob["offsets"]), (what, "offsets", ci, b)
        oc.close()
