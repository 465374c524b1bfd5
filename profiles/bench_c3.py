"""Secondary measurement (not the driver's bench): config C3 of BASELINE.md -- dictionary Utf8 + PRESENT,
Snappy, l_shipmode-like 7-entry dictionary, 10 % nulls."""
import sys, time, json, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in profiles/)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from orc_rust_amd import capi, gen
rows = int(os.environ.get("ROWS", 100_000_000)); stripe_rows = 8_388_608
comp = os.environ.get("COMP", "snappy")
words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK"]
dlens = np.array([len(w) for w in words], dtype=np.int64)
dblob = np.frombuffer(b"".join(words), dtype=np.uint8)
ctx = capi.Context(0)
staged = []; t0 = time.time(); row = 0; s = 0
cols = [{"column_id": 1, "orc_type": 7, "encoding": 3, "dictionary_size": 7}]
while row < rows:
    n = min(stripe_rows, rows - row)
    present = (gen.splitmix64(4 + s, n) % np.uint64(10) != 0).astype(np.uint8)
    k = int(present.sum())
    keys = (gen.splitmix64(3 + s, k) % np.uint64(7)).astype(np.int64)
    c = (lambda b: gen.compress_stream(b, comp, 262144)) if comp != "none" else (lambda b: b)
    streams = [(1, 0, c(gen.boolean(present))), (1, 1, c(gen.rle2(keys, signed=False))), (1, 2, c(gen.rle2(dlens, signed=False))), (1, 3, c(dblob))]
    staged.append(ctx.stage(n, streams, cols, compression=comp))
    row += n; s += 1
print("gen %.1fs, staged bytes %d" % (time.time() - t0, sum(x.nbytes() for x in staged)), file=sys.stderr)
res = ctx.decode(staged)
assert all(r.status()[0] == 0 for r in res)
ab = sum(r.arrow_bytes for r in res)
for _ in range(2): ctx.decode(staged, res)
K = 10; t0 = time.perf_counter(); tot = 0
for _ in range(K):
    ctx.decode(staged, res); tot += ctx.timing()[0]
dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": "C3 dict-Utf8+PRESENT %s" % comp, "rows": rows, "ms_per_step": dt * 1e3, "device_ms": tot / K, "decoded_GBps": ab / dt / 1e9,
                  "mrows_per_s": rows / dt / 1e6, "arrow_bytes": ab, "stream_bytes": sum(x.nbytes() for x in staged)}))
