#!/usr/bin/env python3
"""bench.py -- decoded GB/s + Mrows/s into Arrow for the ORC stripe decode hot path.

Workload (BASELINE.json configs[1], "C2"): one RLEv2 Int64 column, 100 M rows, uncompressed,
12 stripes of 8 388 608 rows (the last one shorter), seeded synthetic data:
  * even stripes  DIRECT:  splitmix64(seed=1) & (2^40-1)  -> zigzag 41 bits -> aligned width 48,
                           512-value runs (3074 B in -> 4096 B out per run)
  * odd stripes   DELTA:   strictly increasing, deltas uniform in [1, 255] (seed=2) -> 8-bit
                           varying-delta runs of 512 values
i.e. the 50/50 DIRECT/DELTA mix of BASELINE.md.  A "step" decodes all stripes of the column
once: staged stream bytes are already resident in HBM, Arrow buffers are left in HBM.

    python bench.py --gpus N --steps K --warmup W     (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `value` = whole-job decoded GB/s (Arrow bytes out / time).
`roofline` prices the dominant kernel (rle2_expand_kernel) with HIP events on the decoder's own
stream; `cpu_baseline` is the CPU oracle (a port of the reference's algorithm, single thread)
on the same streams on this box's host cores.  Multi-GPU: stripes are independent, every rank
decodes its own copy of the workload (weak scaling), the only collective is the final RCCL
all-gather of per-rank row counts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_TOTAL = 100_000_000
STRIPE_ROWS = 8_388_608
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def build_workload(rows_total, stripe_rows, kind="mix"):
    from orc_rust_amd import gen
    stripes = []
    base = 0
    s = 0
    row = 0
    while row < rows_total:
        n = min(stripe_rows, rows_total - row)
        if (kind == "mix" and s % 2 == 0) or kind == "direct":
            vals = (gen.splitmix64(1 + s, n) & np.uint64((1 << 40) - 1)).astype(np.int64)
            skind = "direct"
        elif kind == "arange":
            vals = np.arange(row, row + n, dtype=np.int64)
            skind = "delta-fixed"
        else:
            deltas = (gen.splitmix64(2 + s, n) % np.uint64(255)).astype(np.int64) + 1
            vals = np.cumsum(deltas) + base
            skind = "delta"
        stream, stats = gen.rle2(vals, signed=True, aligned=True, stats=True)
        stripes.append({"n": n, "stream": stream, "kind": skind, "stats": stats, "first": vals[:4].copy(), "last": int(vals[-1]),
                        "xor": int(np.bitwise_xor.reduce(vals.view(np.uint64))), "sum": int(vals.view(np.uint64).sum(dtype=np.uint64))})
        row += n
        s += 1
    return stripes


def cpu_baseline(stripes, budget_s=20.0):
    """CPU oracle (oracle/: port of the reference's decoders, 1 thread) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    rows = 0
    arrow_bytes = 0
    t0 = time.perf_counter()
    used = 0
    for st in stripes:
        col = O.Column(4, 2, {1: st["stream"].tobytes()})
        left = st["n"]
        while left > 0:
            n = min(8192, left)
            b = col.next_batch(n)
            assert b["status"] == 0
            left -= n
        col.close()
        rows += st["n"]
        arrow_bytes += st["n"] * 8
        used += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(arrow_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "mrows_per_s": round(rows / dt / 1e6, 2),
            "sample": "%d of %d stripes (%d rows) of the same workload, batch 8192, oracle/liborc_oracle.so" % (used, len(stripes), rows)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS_TOTAL)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--kind", default="mix", choices=["mix", "direct", "delta", "arange"], help="mix = the headline 50/50 workload")
    ap.add_argument("--skip-check", action="store_true", help="profiling runs: skip the full-size parity properties (12 k small D2H copies)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # BENCH_BACKEND=gloo is a dry-run aid for boxes with fewer GPUs than ranks (ranks then share devices and
    # the two tiny collectives run on CPU tensors); the driver's runs use the default, RCCL ("nccl").
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)

    from orc_rust_amd import capi
    ctx = capi.Context(local_rank)
    stripes = build_workload(args.rows, STRIPE_ROWS, args.kind)
    cols = [{"column_id": 1, "orc_type": 4, "encoding": 2}]
    torch.cuda.synchronize()
    t_stage = time.perf_counter()
    staged = [ctx.stage(st["n"], [(1, 1, st["stream"])], cols) for st in stripes]
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t_stage  # host buffers -> HBM through the pinned bounce buffer (not part of `value`)
    stream_bytes = sum(s.nbytes() for s in staged)
    rows = sum(st["n"] for st in stripes)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    results = ctx.decode(staged)
    # parity properties at full size (size independent): first values, xor and wrapping sum per stripe
    for st, res in zip(stripes, results):
        assert res.status()[0] == 0, res.status()
        if args.skip_check:
            continue
        v0 = np.frombuffer(res.batch(0, 0)["values"], dtype=np.int64)
        assert np.array_equal(v0[:4], st["first"])
        nb = res.n_batches
        acc_x, acc_s = np.uint64(0), np.uint64(0)
        for b in range(nb):
            vb = np.frombuffer(res.batch(b, 0)["values"], dtype=np.uint64)
            acc_x ^= np.bitwise_xor.reduce(vb)
            acc_s = np.uint64((int(acc_s) + int(vb.sum(dtype=np.uint64))) & ((1 << 64) - 1))
        assert int(acc_x) == st["xor"] and int(acc_s) == st["sum"], "GPU decode differs from the generated values"
    arrow_bytes = sum(r.arrow_bytes for r in results)

    for _ in range(args.warmup):
        ctx.decode(staged, results)
    barrier()
    t0 = time.perf_counter()
    exp_ms = 0.0
    tot_ms = 0.0
    for _ in range(args.steps):
        ctx.decode(staged, results)
        t, e, _n = ctx.timing()
        exp_ms += e
        tot_ms += t
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the path's only exchange: per-rank row counts (so every rank knows the global row offsets)
        cnt = torch.tensor([rows], device=coll_dev, dtype=torch.int64)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        total_rows = int(sum(int(c.item()) for c in allc))
    else:
        total_rows = rows
    ms_per_step = dt / args.steps * 1e3
    value = arrow_bytes * world / (dt / args.steps) / 1e9
    exp_avg_ms = exp_ms / args.steps
    algo_bytes = stream_bytes + arrow_bytes  # staged stream bytes in + Arrow bytes out
    achieved = algo_bytes / (exp_avg_ms * 1e-3) / 1e9 if exp_avg_ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("rle2_expand_kernel_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "decoded GB/s + Mrows/s into Arrow", "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64", "data": "synthetic",
        "config": {"workload": "C2: RLEv2 DIRECT(48-bit)/DELTA(8-bit) 50/50 Int64 column, %d rows, uncompressed, %d stripes" % (rows, len(stripes))
                   if args.kind == "mix" else "C2 variant '%s': Int64 column, %d rows, uncompressed, %d stripes" % (args.kind, rows, len(stripes)),
                   "rows_per_gpu": rows, "stripe_rows": STRIPE_ROWS, "batch_size": 8192, "parallelism": "stripe-shard x%d" % world},
        "mrows_per_s": round(total_rows / (dt / args.steps) / 1e6, 1),
        "stream_bytes_in": stream_bytes, "arrow_bytes_out": arrow_bytes,
        "device_ms_per_step": round(tot_ms / args.steps, 4),
        # staging the host stream buffers (pinned bounce buffer + hipMemcpyAsync) is outside the timed region;
        # the PCIe-inclusive rate is reported for DESIGN.md only
        "h2d_stage_ms": round(t_stage * 1e3, 3),
        "pcie_inclusive_GBps": round(arrow_bytes / (t_stage + dt / args.steps) / 1e9, 2),
        "roofline": {"bound": "hbm", "kernel": "rle2_expand_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes,
                     "kernel_ms": round(exp_avg_ms, 4)},
    }
    if rank == 0:
        if not args.no_cpu and world == 1:  # timed on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(stripes)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
