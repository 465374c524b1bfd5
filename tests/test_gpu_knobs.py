"""Every ORCGPU_* environment switch the library still reads (round 6 pruned the development ones) has a parity test:
ORCGPU_LANES, ORCGPU_ZSTD_LANES, ORCGPU_ZSTD_LIT_ASIDE and ORCGPU_EXEC_PATIENCE in tests/test_gpu_zstd_lanes.py; here the
ones that are read once per process or per context -- ORCGPU_POISON (workspace and result arenas filled with a byte pattern first:
nothing may depend on what a buffer held before), ORCGPU_PINNED_POOL_MB=0 (no pinned host memory kept between results),
ORCGPU_STAGE_THREADS=0 (the caller stages alone), ORCGPU_NO_MMAP=1 (the file read with fread instead of mapped).  A child
process decodes a lineitem stripe (Zstandard, every column kind) and reads two fixture files through the reader under each
setting and prints a digest of every Arrow buffer: the digests must be those of the plain run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import gpu_util as G, arrow_util as A
from orc_rust_amd import ArrowReaderBuilder, capi
from orc_rust_amd.gen import workloads as W
h = hashlib.sha256()
table = W.lineitem_table(50_000)
n, cols, streams, expect = W.lineitem_stripe(table, 0, 50_000, "zstd")
for _ in range(2):  # (the second call decodes into pooled arenas: what POISON is about)
    res = G.gpu_decode(n, cols, streams, compression="zstd", batch_size=8192)
    assert res.status()[0] == 0
    for ci in range(len(cols)):
        for b in range(res.n_batches):
            g = res.batch(b, ci)
            h.update(g["values"]); h.update(b"|")
            if g["offsets"] is not None: h.update(g["offsets"].tobytes())
            if g["validity"] is not None: h.update(g["validity"])
            h.update(str((g["length"], g["null_count"])).encode())
    res.free()
ctx = capi.Context(0)
for name in ("TestOrcFile.testSeek.orc", "alltypes.zlib.orc"):
    for prefetch in (0, 2):
        for rb in ArrowReaderBuilder.try_new(A.data_path(name), ctx).with_prefetch(prefetch).with_batch_size(1000).build():
            for col in rb.columns:
                for buf in col.buffers():
                    if buf is not None: h.update(buf.to_pybytes()[:0])  # (buffer sizes may be padded: hash the values instead)
                h.update(str(col.to_pylist()).encode())
print("DIGEST", h.hexdigest())
''' % (ROOT, ROOT)


def digest(extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("ORCGPU_")}
    env.update(extra)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return [l for l in p.stdout.decode().splitlines() if l.startswith("DIGEST")][-1]


def test_the_process_wide_switches_do_not_change_a_byte():
    plain = digest({})
    for extra in ({"ORCGPU_POISON": "0xA5"}, {"ORCGPU_POISON": "0xFF", "ORCGPU_LANES": "3"}, {"ORCGPU_PINNED_POOL_MB": "0"},
                  {"ORCGPU_STAGE_THREADS": "0"}, {"ORCGPU_STAGE_THREADS": "3", "ORCGPU_NO_MMAP": "1"}):
        assert digest(extra) == plain, extra
