"""Helpers shared by the parity tests: turn (values, offsets, validity) batch buffers -- the
layout both the CPU oracle and the HIP path emit -- into pyarrow arrays, and load the committed
golden expectations."""
import os

import numpy as np
import pyarrow as pa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# Columns the REFERENCE cannot decode (there is no expectation to pin on; oracle and GPU path must fail the way the reference
# does): decimal64_v2*.orc are written with ORC 2.0's "decimal64" encoding -- DATA is an RLE v2 stream of i64, there is no
# SECONDARY stream -- while decimal.rs:36-60 reads DATA as unbounded varints and the scales from a stream the stripe does not
# hold (an empty one, stripe.rs:322-336): the first batch ends in OutOfSpec.  Column `a` (Int64) decodes; so does a column of
# precision above 18 (`c` of decimal64_v2.orc), which keeps the classic encoding.
REFERENCE_FAILS = {"decimal64_v2.orc": {"b", "d", "e"}, "decimal64_v2_cplusplus.orc": {"b", "c", "d", "e"}}


def arrow_type(kind, precision=0, scale=0, ts_unit=3):
    unit = ["s", "ms", "us", "ns"][ts_unit]
    return {
        0: pa.bool_(), 1: pa.int8(), 2: pa.int16(), 3: pa.int32(), 4: pa.int64(), 5: pa.float32(), 6: pa.float64(),
        7: pa.string(), 8: pa.binary(), 9: pa.timestamp(unit), 14: pa.decimal128(precision or 38, scale), 15: pa.date32(),
        16: pa.string(), 17: pa.string(), 18: pa.timestamp(unit, tz="UTC"),
    }[kind]


def to_arrow(kind, batch, precision=0, scale=0, ts_unit=3):
    """batch: dict(length, null_count, validity: bytes|None, values: bytes, offsets: np.int32|None)."""
    n = int(batch["length"])
    typ = arrow_type(kind, precision, scale, ts_unit)
    validity = pa.py_buffer(batch["validity"]) if batch["validity"] is not None else None
    nulls = int(batch["null_count"])
    if kind in (7, 8, 16, 17):
        bufs = [validity, pa.py_buffer(np.ascontiguousarray(batch["offsets"], dtype=np.int32).tobytes()), pa.py_buffer(batch["values"])]
    else:
        bufs = [validity, pa.py_buffer(batch["values"])]
    return pa.Array.from_buffers(typ, n, bufs, null_count=nulls)


def expected_table(stem):
    import pyarrow.ipc as ipc
    import pyarrow.parquet as pq
    if stem in ("demo-12-zlib", "demo-11-zlib"):
        return pq.read_table(os.path.join(GOLDEN, "expected", "demo-12-zlib.parquet"))
    p = os.path.join(GOLDEN, "expected", stem + ".feather")
    if not os.path.exists(p):
        return None
    with pa.memory_map(p) as src:
        return ipc.open_file(src).read_all()


def data_path(name):
    return os.path.join(GOLDEN, "data", name)


def golden_files():
    return sorted(f for f in os.listdir(os.path.join(GOLDEN, "data")) if f.endswith(".orc"))
