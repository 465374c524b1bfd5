#!/usr/bin/env python3
"""What a row selection costs with and without row-group pruning (orcgpu_reader_set_row_group_pruning): a TPC-H-shaped
lineitem file written by the ORC C++ writer (PyArrow; Zstandard, 64 MiB stripes, rowIndexStride 10 000), read through the
file reader (a) whole, (b) under a selection of about 1 % of its row groups with the stripes decoded whole and the rows
discarded -- what the reference does (skip_values decodes and discards, rle_v2/mod.rs:148-175) --, (c) the same selection
with only the row groups that hold selected rows read, staged and decoded.  Prints one JSON object.
    python3 profiles/select_cost.py [rows]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if not os.path.isdir("/usr/share/zoneinfo") and "TZDIR" not in os.environ:
    import tzdata
    os.environ["TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")

import numpy as np
import pyarrow.orc as orc

import make_lineitem
from orc_rust_amd import capi
from orc_rust_amd.arrow_reader import ArrowReaderBuilder
from orc_rust_amd.gen import workloads as W


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
    t0 = time.time()
    table = make_lineitem.arrow_table(W.lineitem_table(rows), rows)
    path = os.path.join(tempfile.mkdtemp(), "lineitem.orc")
    orc.write_table(table, path, compression="zstd", dictionary_key_size_threshold=0.8, stripe_size=64 << 20, row_index_stride=10000)
    f = orc.ORCFile(path)
    stripe_rows = [f.read_stripe(i).num_rows for i in range(f.nstripes)]
    write_s = time.time() - t0
    ctx = capi.Context(0)
    stride = 10000
    n_groups = sum((n + stride - 1) // stride for n in stripe_rows)
    # ~1 % of the row groups, spread over the file: 300 rows out of the middle of every 100th group
    rng = np.random.default_rng(1)
    picks = sorted(rng.choice(rows // stride, max(1, n_groups // 100), replace=False).tolist())
    sel, at = [], 0
    for g in picks:
        start = g * stride + 4000
        sel += [(start - at, True), (300, False)]
        at = start + 300
    sel.append((rows - at, True))

    def run(selection, prune, prefetch):
        best, groups, n_out = None, None, 0
        for _ in range(3):
            b = ArrowReaderBuilder.try_new(path, ctx).with_prefetch(prefetch)
            if selection is not None:
                b = b.with_row_selection(selection).with_row_group_pruning(prune)
            r = b.build()
            t = time.time()
            n_out = sum(x.num_rows for x in r)
            dt = time.time() - t
            groups = r.row_groups()
            r.close()
            best = dt if best is None else min(best, dt)
        return {"seconds": round(best, 4), "rows_out": n_out, "row_groups_read": groups[0], "row_groups": groups[1]}

    out = {"file": {"rows": rows, "bytes": os.path.getsize(path), "stripes": len(stripe_rows), "row_groups": n_groups, "write_s": round(write_s, 1)},
           "selection": {"row_groups_with_selected_rows": len(picks), "rows": 300 * len(picks)}}
    for prefetch in (0, 4):
        tag = "serial" if prefetch == 0 else "read_ahead"
        whole = run(None, True, prefetch)
        discard = run(sel, False, prefetch)
        pruned = run(sel, True, prefetch)
        out[tag] = {"whole_file": whole, "selection_decode_and_discard": discard, "selection_pruned": pruned,
                    "pruned_over_whole": round(pruned["seconds"] / whole["seconds"], 4),
                    "pruned_over_discard": round(pruned["seconds"] / discard["seconds"], 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
