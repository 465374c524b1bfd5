"""Secondary measurement (not the driver's bench): a lineitem-shaped stripe set in the spirit of BASELINE
config C4 -- 16 flat columns of mixed types with nulls (Long / Int / Date / Double / Decimal / Timestamp /
dictionary and direct Utf8), ROWS rows in stripes of 1 Mi rows, codec COMP."""
import sys, time, json, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orc_rust_amd import capi, gen

rows = int(os.environ.get("ROWS", 8 * 1048576)); stripe_rows = 1048576
comp = os.environ.get("COMP", "none")
LONG, INT, DATE, DOUBLE, STRING, DECIMAL, TIMESTAMP = 4, 3, 15, 6, 7, 14, 9
PRESENT, DATA, LENGTH, DICT, SECONDARY = 0, 1, 2, 3, 5
words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK"]
c = (lambda b: gen.compress_stream(b, comp, 262144)) if comp != "none" else (lambda b: b)
ctx = capi.Context(0)
staged = []; t0 = time.time(); s = 0; row = 0
while row < rows:
    n = min(stripe_rows, rows - row)
    rng = np.random.default_rng(100 + s)
    cols, streams = [], []
    def add(typ, mk, enc=2, **kw):
        cid = len(cols) + 1
        present = (rng.random(n) >= 0.05).astype(np.uint8)
        k = int(present.sum())
        cols.append(dict(column_id=cid, orc_type=typ, encoding=enc, **kw))
        streams.append((cid, PRESENT, c(gen.boolean(present))))
        for kind, data in mk(k):
            streams.append((cid, kind, c(data)))
    add(LONG, lambda k: [(DATA, gen.rle2(np.arange(k) + row, signed=True))])                       # orderkey-like
    add(LONG, lambda k: [(DATA, gen.rle2(rng.integers(1, 200000, k), signed=True))])               # partkey
    add(LONG, lambda k: [(DATA, gen.rle2(rng.integers(1, 10000, k), signed=True))])                # suppkey
    add(INT, lambda k: [(DATA, gen.rle2(rng.integers(1, 8, k), signed=True))])                     # linenumber
    add(INT, lambda k: [(DATA, gen.rle2(rng.integers(1, 51, k), signed=True))])                    # quantity
    add(DOUBLE, lambda k: [(DATA, rng.random(k).view(np.uint8))], enc=0)                           # extendedprice
    add(DOUBLE, lambda k: [(DATA, (rng.integers(0, 11, k) / 100.0).view(np.uint8))], enc=0)        # discount
    add(DECIMAL, lambda k: [(DATA, gen.varint128([int(x) for x in rng.integers(0, 9, k)])), (SECONDARY, gen.rle2(np.full(k, 2), signed=True))],
        precision=12, scale=2)                                                                    # tax
    for _ in range(2):                                                                            # returnflag / linestatus / shipmode-like
        add(STRING, lambda k: [(DATA, gen.rle2(rng.integers(0, len(words), k), signed=False)),
                               (LENGTH, gen.rle2(np.array([len(w) for w in words], dtype=np.int64), signed=False)),
                               (DICT, np.frombuffer(b"".join(words), dtype=np.uint8))], enc=3, dictionary_size=len(words))
    for _ in range(3):                                                                            # ship / commit / receipt dates
        add(DATE, lambda k: [(DATA, gen.rle2(rng.integers(8000, 10600, k), signed=True))])
    add(TIMESTAMP, lambda k: [(DATA, gen.rle2(rng.integers(0, 10**8, k), signed=True)), (SECONDARY, gen.rle2(np.zeros(k, dtype=np.int64), signed=False))])
    def comment(k):
        idx = rng.integers(0, len(words), k)
        return [(LENGTH, gen.rle2(np.array([len(words[i]) for i in idx], dtype=np.int64), signed=False)),
                (DATA, np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8))]
    add(STRING, comment)                                                                          # comment (direct)
    add(LONG, lambda k: [(DATA, gen.rle2(np.repeat(rng.integers(0, 1000, k // 7 + 1), 7)[:k], signed=True))])
    staged.append(ctx.stage(n, streams, cols, compression=comp))
    row += n; s += 1
print("gen %.1fs, %d stripes x %d columns, staged bytes %d" % (time.time() - t0, len(staged), len(cols), sum(x.nbytes() for x in staged)), file=sys.stderr)
res = ctx.decode(staged)
assert all(r.status()[0] == 0 for r in res), [r.status() for r in res]
ab = sum(r.arrow_bytes for r in res)
for _ in range(2): ctx.decode(staged, res)
K = 10; t0 = time.perf_counter(); tot = 0
for _ in range(K):
    ctx.decode(staged, res); tot += ctx.timing()[0]
dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": "mixed 16-column stripes, %s" % comp, "rows": rows, "columns": len(cols), "stripes": len(staged), "ms_per_step": round(dt * 1e3, 3),
                  "device_ms": round(tot / K, 3), "decoded_GBps": round(ab / dt / 1e9, 1), "mrows_per_s": round(rows / dt / 1e6, 1), "arrow_bytes": ab,
                  "stream_bytes": sum(x.nbytes() for x in staged)}))
