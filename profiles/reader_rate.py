#!/usr/bin/env python3
"""What the file reader itself delivers: a TPC-H-shaped lineitem file written by the ORC C++ writer (PyArrow; Zstandard, 64 MiB
stripes) read through orcgpu_reader_next_batch, every exported batch released at once (no Arrow import on the Python side):
file bytes -> staged -> decoded -> copied back -> Arrow C Data batches.  Best of three passes (the first pins the host buffers).
    python3 profiles/reader_rate.py [rows]      prints one JSON object"""
import ctypes as C, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if not os.path.isdir("/usr/share/zoneinfo"):
    import tzdata; os.environ["TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
import pyarrow.orc as orc
import make_lineitem
from orc_rust_amd import capi
from orc_rust_amd.gen import workloads as W
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
table = make_lineitem.arrow_table(W.lineitem_table(rows), rows)
path = os.path.join(tempfile.mkdtemp(), "li.orc")
orc.write_table(table, path, compression="zstd", dictionary_key_size_threshold=0.8, stripe_size=64 << 20)
arrow_bytes = table.nbytes
ctx = capi.Context(0)
L = ctx.L
class ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]
class ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_void_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]
out = {"file": {"rows": rows, "bytes": os.path.getsize(path), "arrow_bytes": arrow_bytes}, "runs": []}
for bs in (8192, 65536):
    for prefetch in (0, 2):
        best = None
        for _ in range(3):
            h = C.c_void_p()
            assert L.orcgpu_reader_open_file(ctx.h, path.encode(), C.byref(h)) == 0
            L.orcgpu_reader_set_batch_size(h, bs); L.orcgpu_reader_set_prefetch(h, prefetch)
            t0 = time.perf_counter(); n = nb = 0
            while True:
                a, s = ArrowArray(), ArrowSchema()
                rc = L.orcgpu_reader_next_batch(h, C.byref(a), C.byref(s))
                if rc == 110: break  # ORCGPU_END_OF_FILE
                assert rc == 0, rc
                n += a.length; nb += 1
                a.release(C.addressof(a)); s.release(C.addressof(s))
            dt = time.perf_counter() - t0
            L.orcgpu_reader_close(h)
            best = dt if best is None else min(best, dt)
        out["runs"].append({"batch_size": bs, "prefetch": prefetch, "batches": nb, "rows": n, "ms": round(best * 1e3, 1), "arrow_GBps": round(arrow_bytes / best / 1e9, 1)})
print(json.dumps(out))
