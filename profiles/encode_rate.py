"""Rate of the device encoders on device-resident Int64 columns: orcgpu_encode_column with ORCGPU_ENC_ON_DEVICE,
whole call (three host synchronisations inside: run count, stream size, end), beside the restated reference encoder on one host
core (oracle/oo_encode.c) over a sample.  Usage: python profiles/encode_rate.py [rows]  -> one JSON object on stdout."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orc_rust_amd import capi  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48_000_000
    rng = np.random.default_rng(1)
    shapes = {
        "orderkey (runs of 1-7 ascending keys)": np.repeat(np.cumsum(rng.integers(1, 4, n // 4 + 1)), rng.integers(1, 8, n // 4 + 1))[:n],
        "quantity (random 1..50)": rng.integers(1, 51, n),
        "shipdate (random days in 7 years)": rng.integers(8000, 10556, n),
        "unique ascending keys": np.cumsum(rng.integers(1, 30, n)),
        "random 40-bit with 2 % outliers": np.where(rng.random(n) < 0.02, rng.integers(1 << 50, 1 << 60, n), rng.integers(0, 1 << 40, n)),
        "constant": np.full(n, 7),
    }
    ctx = capi.Context()
    out = {"rows": n, "unit": "GB/s of Int64 values in", "shapes": {}}
    for name, v in shapes.items():
        v = np.ascontiguousarray(v, dtype=np.int64)
        # the column on the device: decode a stream of it and take the Arrow buffer where it lies
        t0 = time.perf_counter()
        stream = ctx.encode_rle2(v, 8, True)
        host_call_ms = (time.perf_counter() - t0) * 1e3
        staged = ctx.stage(n, [(1, 1, stream)], [{"column_id": 1, "orc_type": 4, "encoding": 2}], batch_size=n)
        res = ctx.decode([staged])[0]
        staged.free()
        view = capi.BatchView()
        ctx._check(ctx.L.orcgpu_result_batch_view(res.h, 0, 0, C.byref(view)))
        col = capi.EncColumn(capi.ARROW["int64"], capi.ENC_ON_DEVICE, n, None, view.values, None)
        streams = (capi.EncStream * 3)()
        ns = C.c_uint32(0)
        times = []
        for _ in range(6):
            t0 = time.perf_counter()
            ctx._check(ctx.L.orcgpu_encode_column(ctx.h, C.byref(col), streams, C.byref(ns)))
            times.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.median(times[1:]))
        buf = np.zeros(streams[0].len, dtype=np.uint8)
        ctx._check(ctx.L.orcgpu_encode_fetch(ctx.h, C.byref(streams[0]), buf.ctypes.data))
        assert buf.tobytes() == stream
        res.free()
        m = min(n, 4_000_000)
        t0 = time.perf_counter()
        want = O.enc_rle2(v[:m], 8, True)
        cpu_s = time.perf_counter() - t0
        if m == n:
            assert want == stream
        out["shapes"][name] = {"device_ms": round(ms, 3), "device_gbps": round(n * 8 / ms / 1e6, 2), "stream_bytes": len(stream),
                               "host_in_host_out_ms": round(host_call_ms, 1),
                               "cpu_restated_reference_gbps_1_core": round(m * 8 / cpu_s / 1e9, 3), "cpu_sample_rows": m}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
