"""Cross-check of the lineitem workload generator against the ORC C++ writer (PyArrow), and the source of
the small committed lineitem fixture.  Runs ONLY in the build container (PyArrow bundles Apache ORC C++):

    python tests/golden/make_lineitem.py [rows_for_the_stripe_measurement]

* writes tests/golden/data/lineitem_8k.zstd.orc (9 000 rows of orc_rust_amd.gen.workloads.lineitem_table, schema of
  the reference's scripts/convert_tpch.py:46-63, Zstandard, dictionary encoding on, 64 KiB compression blocks;
  and its expected decode tests/golden/expected/lineitem_8k.zstd.feather (pyarrow.orc read);
* prints, for a larger table, the stripe layout the ORC C++ writer produces with stripe_size = 64 MiB and how its
  stream sizes compare with the generator's own encoders (the figures quoted in gen/workloads.py and DESIGN.md).
"""
import datetime
import decimal
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.feather as feather
import pyarrow.orc as orc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from orc_rust_amd.gen import workloads as W  # noqa: E402


def arrow_table(tab, n):
    cols = {}
    for name, typ, how in W.LINEITEM:
        if how == "i64":
            cols[name] = pa.array(tab[name][:n], type=pa.int64())
        elif how == "i32" and typ == W.DATE:
            cols[name] = pa.array(tab[name][:n], type=pa.int32()).cast(pa.date32())
        elif how == "i32":
            cols[name] = pa.array(tab[name][:n], type=pa.int32())
        elif how == "dec":
            # Decimal128(15,2) from unscaled int64: through the raw buffers (no Python Decimal objects)
            wide = np.zeros((n, 2), dtype=np.int64)
            wide[:, 0] = tab[name][:n]
            wide[:, 1] = tab[name][:n] >> 63
            cols[name] = pa.Array.from_buffers(pa.decimal128(15, 2), n, [None, pa.py_buffer(wide.tobytes())])
        elif how == "dict":
            words = W.DICTS[name]
            vb, kl = W._dict_arrow(words, tab[name][:n])
            off = np.zeros(n + 1, dtype=np.int32)
            np.cumsum(kl, out=off[1:])
            cols[name] = pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(off.tobytes()), pa.py_buffer(vb.tobytes())])
        else:
            clen, cb = tab[name]
            off = np.zeros(n + 1, dtype=np.int32)
            np.cumsum(clen[:n], out=off[1:])
            cols[name] = pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(off.tobytes()), pa.py_buffer(cb[:off[-1]].tobytes())])
    return pa.table(cols)


def main():
    n_small = 9000
    tab = W.lineitem_table(n_small)
    t = arrow_table(tab, n_small)
    path = os.path.join(HERE, "data", "lineitem_8k.zstd.orc")
    orc.write_table(t, path, compression="zstd", compression_block_size=65536, dictionary_key_size_threshold=0.8, stripe_size=64 << 20)
    back = orc.ORCFile(path).read()
    assert back.equals(t)
    feather.write_feather(back, os.path.join(HERE, "expected", "lineitem_8k.zstd.feather"), compression="zstd")
    print("fixture:", os.path.getsize(path), "bytes")
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
    tab = W.lineitem_table(rows)
    t = arrow_table(tab, rows)
    tmp = "/tmp/lineitem_check.orc"
    orc.write_table(t, tmp, compression="zstd", dictionary_key_size_threshold=0.8, stripe_size=64 << 20)
    f = orc.ORCFile(tmp)
    print("ORC C++ writer, stripe_size 64 MiB: file %d bytes (%.2f B/row), %d stripes, rows per stripe %s" % (
        os.path.getsize(tmp), os.path.getsize(tmp) / rows, f.nstripes, [f.read_stripe(i).num_rows for i in range(f.nstripes)]))
    n, cols, streams, _ = W.lineitem_stripe(tab, 0, rows, "zstd", want_expect=False)
    print("generator's encoders + zstd-3, same rows: %.2f B/row" % (sum(len(b) for _, _, b in streams) / rows))
    os.remove(tmp)


if __name__ == "__main__":
    main()
