#!/usr/bin/env python3
"""bench.py -- decoded GB/s + Mrows/s into Arrow for the ORC stripe decode hot path.

Default workload = the one BASELINE.json's metric is quoted on, config C4: the stripes of a TPC-H-shaped `lineitem` table
(the 16 columns of the reference's scripts/convert_tpch.py:46-63: 3 x Int64, Int32, 4 x Decimal128(15,2), 3 x Date32,
4 dictionary Utf8, 1 direct Utf8), Zstandard level 3 in 256 KiB chunks, stripes of 2 189 312 rows (what the ORC C++ writer
cuts at stripe_size = 64 MiB for this table), seeded synthetic data (orc_rust_amd/gen/tpchgen.c: no dbgen in the image).
Size: ONE GPU'S SHARE of C4 -- scale factor 100 over 8 GPUs = SF 12.5 = 75 004 738 rows = 35 stripes per GPU; with
`--gpus N` the table is N shares (N = 8: the whole SF100 table, 600 037 902 rows), `--sf` overrides the per-GPU size,
`--scaling strong` keeps the table at its one-GPU size and shards it over the ranks instead.
A "step" decodes every stripe of the rank once: compressed stream bytes are already resident in HBM (staged through the C
ABI before the timed region), Arrow buffers are left in HBM.

    python bench.py --gpus N --steps K --warmup W [--workload lineitem|c2|c2-direct|c2-delta|c2-arange|c2-adv|c2-rowgroup|c3|c5]
    N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the ranks
    are there: WORLD_SIZE = N), or plainly as `python bench.py --gpus N`: bench.py then starts N rank processes itself
    (fresh children, before anything touches the GPU) and relays rank 0's line.

Prints ONE JSON line (rank 0).  `value` = whole-job decoded GB/s (Arrow bytes out of all ranks / time of the slowest
rank).  `roofline` prices the dominant phase of the pipeline (HIP events on the decoder's own stream, per phase:
orcgpu_last_phase_ms) against the HBM peak with the ALGORITHMIC bytes of SURVEY 8(d): staged stream bytes in + Arrow
bytes out.  `cpu_baseline` = the CPU oracle (a C port of the reference's algorithm; the Rust reference cannot be
built here) on a bounded sample of the same stripes on this box's host cores, 1 thread and all cores.

Checks before timing (lineitem): the first stripe of the rank is compared buffer by buffer with what the generator implies
(workloads.check_result); EVERY stripe is compared through weighted word sums of its whole Arrow buffers (values, offsets;
null counts must be zero): the worker process that generated a stripe reduces the expected buffers, torch reduces the
decoded ones on the device (workloads.buffer_sum / device_sum below) -- nothing of the product is used by the checker.

Multi-GPU (`--gpus N`): the path shards with no data-path collective, one process per GPU.  lineitem: the table is cut
into (stripe, column) UNITS spread over the ranks by their Arrow bytes, longest first (orc_rust_amd.shard.unit_shard):
every rank generates (only its columns of only its stripes: the generator draws by counter), stages and decodes its own
units.  c2 / c3 / c5: STRIPE shard (round robin).  The only exchange is one all-gather (RCCL) of {rows x columns decoded,
value bytes of the rank's string columns, Arrow bytes, stream bytes, units, error word, milliseconds per step}; rank 0
checks that every unit was decoded exactly once and that the row counts add up.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The decoder runs its column lanes, the literals kernel and the copies back on streams of their own; the HIP runtime maps streams
# onto 4 hardware queues by default and kernels that share a queue run one after the other (measured: the Zstandard literals
# kernel beside the sequences kernel only with more queues).  A setting of the runtime, read when it starts: INTEGRATION.md.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
C2_ROWS, C2_STRIPE_ROWS = 100_000_000, 8_388_608
C4_SF100_ROWS = 600_037_902  # BASELINE.json configs[3]: TPC-H SF100 lineitem, sharded over 8 GPUs
PHASE_KERNELS = {
    "decompress": "block decompression of every chunk of the call",
    "decompress_stage1": {"zstd": "zstd_entropy_kernel (FSE sequences + Huffman literals, one wavefront per block)",
                          "snappy": "lz_parse_kernel (token stage, one workgroup per chunk)", "lz4": "lz_parse_kernel (token stage, one workgroup per chunk)"},
    "decompress_sequences": "zstd_seq_quads_kernel (FSE sequences, four lanes per block; the Huffman literals kernel runs beside it)",
    "decompress_stage2_wave": "lz_exec_wave_kernel (LZ77 execution, one wavefront per chunk)",
    "decompress_stage2": {"zstd": "lz_exec_kernel (LZ77 execution, one workgroup per chunk)", "snappy": "lz_exec_tokens_kernel (LZ77 execution, one workgroup per chunk)",
                          "lz4": "lz_exec_tokens_kernel (LZ77 execution, one workgroup per chunk)", "zlib": "decompress_deflate_kernel (one wavefront per chunk)"},
    "walk": "decompress_finalize_kernel + rle_walk_kernel / rle_walk_short_kernel rounds + scans (run boundaries)",
    "present": "pres_*_kernel (PRESENT -> validity, ranks)",
    "expand": "rle2_expand_kernel (+ rle1 / byte expand)",
    "finish": "finishers: null spacing, strings, varint / decimal, timestamps",
}


def usable_cores():
    """CPU cores this process can actually use: hardware threads, cut down to the scheduler affinity and to the cgroup's CPU
    quota (a GPU box shows 256 hardware threads to a container whose quota is 16 CPUs: 256 busy processes then run at 1/16
    speed each, and "256 cores" would overstate the host by 16x)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
        except (OSError, ValueError):
            pass
    try:  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p_ > 0:
            n = min(n, max(1, int(q / p_ + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def host_workers(world):
    """Worker processes this rank may use for host-side set-up (generation, the CPU baseline): its share of the usable cores."""
    return max(1, usable_cores() // max(1, world))


def lineitem_total_rows(args, world):
    if args.sf:
        per = int(round(args.sf * 6_001_215))
    elif args.rows:
        per = args.rows
    else:  # one GPU's share of C4 (SF100 over 8 GPUs); N shares make exactly the SF100 table at N = 8
        return int(round(C4_SF100_ROWS / 8)) if args.scaling == "strong" else int(round(C4_SF100_ROWS * world / 8))
    return per if args.scaling == "strong" else per * world


def build_workload(args, rank, world):
    """Returns (stripes, compression, label, shard description, plan).  stripes: [(n_rows, cols, streams, expect, sums)] of THIS rank
    (expect: the generator's Arrow buffers, kept for the first stripe only at lineitem scale; sums: their reductions)."""
    from orc_rust_amd import shard
    from orc_rust_amd.gen import workloads as W
    wl = args.workload
    if wl == "lineitem":
        import multiprocessing as mp
        from orc_rust_amd import gen
        comp = args.compression or "zstd"
        rows = lineitem_total_rows(args, world)
        stripe_rows = [min(W.LINEITEM_STRIPE_ROWS, rows - lo) for lo in range(0, rows, W.LINEITEM_STRIPE_ROWS)]
        units, loads = shard.unit_shard(stripe_rows, W.LINEITEM_ARROW_BYTES_PER_ROW, world)
        mine = units[rank]
        if getattr(args, "columns", ""):
            keep = {int(c) - 1 for c in args.columns.split(",")}
            mine = [u for u in mine if u[1] in keep]
        desc = "all 16 columns of every stripe" if world == 1 else "(stripe, column) units x%d, LPT by Arrow bytes (not whole columns: l_comment alone is 18 %% of the bytes)" % world
        by_stripe = {}
        for s_, c_ in mine:
            by_stripe.setdefault(s_, []).append(c_ + 1)
        order = sorted(by_stripe)
        tasks = [(s_, stripe_rows[s_], comp, sorted(by_stripe[s_]), 7, k == 0) for k, s_ in enumerate(order)]
        # every stripe is generated, encoded and compressed by a worker process of its own (this rank's share of the host cores)
        nproc = min(len(tasks), host_workers(world))
        if any(16 in t[3] for t in tasks):
            gen.lib().orcgen_lineitem_warm(7)  # the comment text pool: built once here, inherited by the forked workers
        t0 = time.perf_counter()
        if nproc > 1:
            # (close + join, never terminate: under rocprofv3 a worker that gets SIGTERM can hang in the profiler's signal handler,
            # and the run with it)
            pool = mp.get_context("fork").Pool(nproc)
            try:
                stripes = pool.map(W.lineitem_unit_task, tasks, chunksize=1)
            finally:
                pool.close()
                pool.join()
        else:
            stripes = [W.lineitem_unit_task(t) for t in tasks]
        gen_s = time.perf_counter() - t0
        label = ("C4: TPC-H-shaped lineitem stripes, scale factor %.4g (%d rows, 16 columns: 3 Int64, Int32, 4 Decimal128(15,2), 3 Date32, "
                 "4 dictionary Utf8, 1 direct Utf8), %s, %d-row stripes%s" % (
                     rows / 6_001_215, rows, comp, W.LINEITEM_STRIPE_ROWS,
                     "" if world == 1 else "; %s scaling: %s" % (args.scaling, "the table is %d one-GPU shares" % world if args.scaling == "weak" else "one table over all ranks")))
        return stripes, comp, label, desc, {"units": mine, "n_stripes": len(stripe_rows), "n_columns": 16, "rows": rows, "load_estimate": loads,
                                            "gen_s": gen_s, "gen_procs": nproc}
    per = args.rows or C2_ROWS
    rows = per if args.scaling == "strong" else per * world
    n_stripes = (rows + C2_STRIPE_ROWS - 1) // C2_STRIPE_ROWS
    mine = shard.stripe_shard(n_stripes, rank, world)
    desc = "all stripes" if world == 1 else "stripe shard x%d (round robin)" % world
    stripes = []
    t0 = time.perf_counter()
    if wl.startswith("c2"):
        comp = "none"
        for s in range(n_stripes):
            n = min(C2_STRIPE_ROWS, rows - s * C2_STRIPE_ROWS)
            kind = {"c2": "direct" if s % 2 == 0 else "delta", "c2-direct": "direct", "c2-delta": "delta", "c2-arange": "arange", "c2-adv": "adv", "c2-rowgroup": "rg"}[wl]
            if s in mine:
                idx = not args.no_row_index  # the DATA stream's ROW_INDEX positions go with it (verified run starts for the walk)
                stripes.append(W.c2_adversarial_stripe(n, s, index=idx)[:4] if kind == "adv" else (W.c2_rowgroup_stripe(n, s, index=idx)[:4] if kind == "rg" else W.c2_stripe(n, s, kind, row0=s * C2_STRIPE_ROWS)[:4]))
        label = "C2%s: RLEv2 Int64 column, %d rows, uncompressed, %d stripes" % (
            " (DIRECT 48-bit / DELTA 8-bit alternating)" if wl == "c2" else (
                " DIRECT 48-bit with the encoder flushed every 10 000 rows (row-group boundaries of a real writer)" if wl == "c2-rowgroup" else " adversarial walk (run lengths 200..511, widths 3..58 bits changing per run, every third run PATCHED_BASE)" if wl == "c2-adv" else " variant " + wl[3:]),
            rows, n_stripes)
        if wl in ("c2-adv", "c2-rowgroup"):
            label += ", the stream's ROW_INDEX positions (one per 10 000 rows) " + ("withheld" if args.no_row_index else "given as verified run starts")
    elif wl == "c3":
        comp = args.compression or "snappy"
        for s in mine:
            stripes.append(W.c3_stripe(min(C2_STRIPE_ROWS, rows - s * C2_STRIPE_ROWS), s, comp, index=args.row_index))
        label = "C3: dictionary Utf8 (7 entries) + PRESENT (10 %% nulls), %d rows, %s, %d stripes" % (rows, comp, n_stripes)
        if args.row_index:
            label += ", the ROW_INDEX positions of the PRESENT and DATA streams (one per 10 000 rows) given as verified run starts"
    elif wl == "c5":
        comp = args.compression or "lz4"
        for s in mine:
            stripes.append(W.c5_stripe(min(C2_STRIPE_ROWS, rows - s * C2_STRIPE_ROWS), s, comp)[:4])
        label = "C5: Timestamp(ns), PATCHED_BASE seconds + DIRECT nanoseconds, %d rows, %s, %d stripes" % (rows, comp, n_stripes)
    else:
        raise SystemExit("unknown workload " + wl)
    stripes = [tuple(st) + (None,) for st in stripes]
    return stripes, comp, label, desc, {"units": [(s_, 0) for s_ in mine], "n_stripes": n_stripes, "n_columns": 1, "rows": rows,
                                        "load_estimate": [len(shard.stripe_shard(n_stripes, r, world)) for r in range(world)],
                                        "gen_s": time.perf_counter() - t0, "gen_procs": 1}


_TASKS = []  # (n_rows, column, {kind: bytes}, compression): filled before the worker processes are forked


def _oracle_stripe(index):
    """Decodes one (stripe, column) with the CPU oracle, batch by batch; returns (rows, arrow bytes, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n, col, sd, comp = _TASKS[index % len(_TASKS)]
    t0 = time.perf_counter()
    oc = O.Column(col["orc_type"], col.get("encoding", 2), sd, dictionary_size=col.get("dictionary_size", 0), precision=col.get("precision", 0),
                  scale=col.get("scale", 0), compression=comp)
    assert oc.status == 0
    left, out = n, 0
    while left > 0:
        b = oc.next_batch(min(8192, left))
        assert b["status"] == 0
        out += len(b["values"]) + (4 * (b["length"] + 1) if b["offsets"] is not None else 0) + (len(b["validity"]) if b["validity"] else 0)
        left -= 8192
    oc.close()
    return n, out, time.perf_counter() - t0


def cpu_baseline(stripes, comp, budget_s=12.0):
    """CPU oracle (oracle/: C port of the reference's decoders -- the Rust reference cannot be built in this image) on
    the host cores of this box: 1 thread (the reference decodes columns, batches and stripes sequentially) and all
    USABLE cores (usable_cores(): the cgroup's CPU quota, not the hardware thread count; one task per (stripe, column), at
    least four tasks per core), each on a bounded sample of the same stripes."""
    import multiprocessing as mp
    tasks = _TASKS
    del tasks[:]
    for n, cols, streams, _, _ in stripes:
        for c in cols:
            sd = {k: (b.tobytes() if isinstance(b, np.ndarray) else bytes(b)) for cid, k, b in streams if cid == c["column_id"]}
            tasks.append((n, {k: v for k, v in c.items()}, sd, comp))
    ncols = max(1, len(stripes[0][1]))
    # 1 thread: whole columns of the first stripe(s) until the budget is used
    t0 = time.perf_counter()
    rows1 = bytes1 = used = 0
    per_column = {}
    for t in range(len(tasks)):
        n, ab, sec = _oracle_stripe(t)
        bytes1 += ab
        used += 1
        pc = per_column.setdefault(tasks[t][1].get("name", "c%d" % tasks[t][1]["column_id"]), [0, 0.0])
        pc[0] += ab
        pc[1] += sec
        if used % ncols == 0:
            rows1 += n
            if time.perf_counter() - t0 > budget_s:
                break
    dt1 = time.perf_counter() - t0
    if used % ncols:
        rows1 += tasks[used - 1][0] * (used % ncols) / ncols
    cores = usable_cores()
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"value": round(bytes1 / dt1 / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port", "mrows_per_s": round(rows1 / dt1 / 1e6, 3),
           "sample": "%d of %d (stripe, column) tasks of the same workload (%.1f stripes), batch 8192, oracle/liborc_oracle.so; the Rust "
                     "reference itself cannot be built or timed here (no cargo/rustc)" % (used, len(tasks), used / ncols),
           "cpu_model": model, "host_cores": cores, "hardware_threads": os.cpu_count(),
           # where the single thread's time goes: per column, decompression + decode together (the oracle streams)
           "per_column": {k: {"ms": round(v[1] * 1e3, 1), "GBps": round(v[0] / v[1] / 1e9, 3) if v[1] > 0 else None} for k, v in per_column.items()}}
    if cores > 1:
        # all cores: at least four tasks per core (the task list is walked round and round if the rank holds fewer), as many
        # as ~budget_s of wall time allows at the measured single-thread rate
        per_task = dt1 / used
        ntask = int(max(4 * cores, min(16 * cores, budget_s * cores / per_task)))
        sel = list(range(ntask))
        sel.sort(key=lambda i: -sum(len(v) for v in tasks[i % len(tasks)][2].values()))  # longest first
        pool = mp.get_context("fork").Pool(cores)
        try:
            pool.map(_oracle_stripe, sel[:cores], chunksize=1)  # start the workers (library load) outside the timed region
            t0 = time.perf_counter()
            res = pool.map(_oracle_stripe, sel, chunksize=1)
            dtn = time.perf_counter() - t0
        finally:
            pool.close()
            pool.join()
        busy = sum(r[2] for r in res)
        out["all_cores"] = {"value": round(sum(r[1] for r in res) / dtn / 1e9, 4), "unit": "GB/s", "cores": cores,
                            "mrows_per_s": round(sum(r[0] for r in res) / ncols / dtn / 1e6, 3),
                            "sample": "%d (stripe, column) tasks (%d distinct) over %d processes, longest first" % (len(sel), min(len(sel), len(tasks)), cores),
                            "worker_busy_frac": round(busy / (dtn * cores), 3)}
    return out


class CpuBaselineHelper:
    """The CPU baseline forks worker processes, which this process may only do before it initialises the GPU -- but timed in front
    of the GPU work it made rank 0 late at the rendezvous of an N-rank job (the other ranks waited for it at the store).  So: a
    helper process is forked HERE, early (it shares the generated stripes copy-on-write and never touches the GPU), sleeps on a
    pipe, and times the baseline when rank 0 asks for it -- behind the timed region, while the other ranks wait at the last barrier."""

    def __init__(self, stripes, comp):
        self._go_r, self._go_w = os.pipe()
        self._out_r, self._out_w = os.pipe()
        sys.stdout.flush()
        sys.stderr.flush()
        self.pid = os.fork()
        if self.pid == 0:
            code = 1
            try:
                os.close(self._go_w)
                os.close(self._out_r)
                if os.read(self._go_r, 1) == b"g":
                    data = json.dumps(cpu_baseline(stripes, comp)).encode()
                    while data:
                        data = data[os.write(self._out_w, data):]
                code = 0
            finally:
                os._exit(code)
        os.close(self._go_r)
        os.close(self._out_w)

    def run(self):
        os.write(self._go_w, b"g")
        os.close(self._go_w)
        chunks = []
        while True:
            b = os.read(self._out_r, 1 << 16)
            if not b:
                break
            chunks.append(b)
        os.close(self._out_r)
        _, status = os.waitpid(self.pid, 0)
        if status != 0 or not chunks:
            raise RuntimeError("the CPU baseline helper failed (exit status %d)" % status)
        return json.loads(b"".join(chunks).decode())


class _DevBytes:
    """A device range as torch sees it (__cuda_array_interface__): the checker's only view of the result buffers."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_sum(torch, ptr, nbytes):
    """workloads.buffer_sum of nbytes device bytes at ptr, computed by torch on the device: (bytes, weighted word sum mod 2^64)."""
    if not nbytes:
        return 0, 0
    t = torch.as_tensor(_DevBytes(ptr, nbytes), device="cuda")
    n8 = nbytes // 8
    total = 0
    if n8:
        w = t[:n8 * 8].view(torch.int64)
        chunk = 1 << 26
        for lo in range(0, n8, chunk):
            hi = min(n8, lo + chunk)
            k = torch.arange(lo, hi, device="cuda", dtype=torch.int64) * 2 + 1
            total += int((w[lo:hi] * k).sum().item())
    if nbytes & 7:
        tail = bytes(t[n8 * 8:].cpu().numpy().tobytes()) + b"\0" * (8 - (nbytes & 7))
        total += int.from_bytes(tail, "little") * (2 * n8 + 1)
    return nbytes, total & ((1 << 64) - 1)


def check_sums(torch, res, cols, sums, what):
    """Every Arrow buffer of a decoded stripe against the sums of the expected ones: the values buffer of a column is one
    range over all batches (checked to be contiguous), offsets are (batch + 1) entries per batch at a fixed stride."""
    nb = res.n_batches
    for ci, c in enumerate(cols):
        e = sums[c["column_id"]]
        views = [res.view(b, ci) for b in range(nb)]
        assert all(v.null_count == 0 and not v.validity for v in views), (what, c.get("name"), "unexpected nulls")
        at = views[0].values
        for v in views:
            assert v.values == at, (what, c.get("name"), "values of the batches are not back to back")
            at += v.values_bytes
        got = device_sum(torch, views[0].values, at - views[0].values)
        assert got == tuple(e["values"]), (what, c.get("name"), "values differ", got, e["values"])
        if "offsets" in e:
            stride = (views[1].offsets - views[0].offsets) if nb > 1 else 0
            assert nb == 1 or all(views[b].offsets == views[0].offsets + b * stride for b in range(nb)), (what, c.get("name"), "offset stride")
            n_entries = (nb - 1) * (stride // 4) + views[-1].length + 1
            got = device_sum(torch, views[0].offsets, 4 * n_entries)
            assert got == tuple(e["offsets"]), (what, c.get("name"), "offsets differ", got, e["offsets"])


def pipelined_end_to_end(ctx, stripes, comp, group=4, passes=3):
    """Host buffers in, host (pinned) Arrow buffers out, with three things in flight at once -- what the read-ahead reader does
    (orcgpu_reader_set_prefetch; the reference: async_arrow_reader.rs:165-280): a second thread stages the stripes to come
    (orcgpu_stage_stripe: host copies into pinned pieces + H2D on the copy stream), this thread decodes `group` staged
    stripes per call (one stripe alone leaves most of the GPU idle: its longest Zstandard block chains bound the call) and
    starts their copies back (orcgpu_result_fetch_async, device-to-host stream), then waits for the copies of the group
    before.  Three sets of results are used in turn: from the fourth group on nothing is allocated.
    Returns (seconds of the last pass, Arrow bytes) -- never part of `value`."""
    import queue
    import threading
    ring = [None, None, None]
    dt = arrow = 0
    for _ in range(passes):  # the first passes allocate the arenas and the pinned host copies (three sets of results in turn)
        arrow = 0
        q = queue.Queue(maxsize=2 * group)
        failure = []

        def stager():
            try:
                for n_, cols_, streams_, _, _ in stripes:
                    q.put(ctx.stage(n_, streams_, cols_, compression=comp))
            except Exception as e:  # noqa: BLE001 -- handed to the main thread
                failure.append(e)
            q.put(None)
        t0 = time.perf_counter()
        th = threading.Thread(target=stager)
        th.start()
        g, prev, more = 0, None, True
        while more:
            batch = []
            while len(batch) < group:
                item = q.get()
                if item is None:
                    more = False
                    break
                batch.append(item)
            if not batch:
                break
            slot = ring[g % 3] or []
            res = ctx.decode(batch, [slot[i] if i < len(slot) else None for i in range(len(batch))])
            ring[g % 3] = res + slot[len(batch):]  # (a shorter last group leaves the rest of its set alone)
            for st in batch:
                st.free()
            for r in res:
                assert r.status()[0] == 0
                arrow += r.arrow_bytes
                r.fetch_async()
            if prev is not None:
                for r in prev:
                    r.fetch()  # the group before is complete on the host: a consumer would read it now
            prev = res
            g += 1
        if prev is not None:
            for r in prev:
                r.fetch()
        dt = time.perf_counter() - t0
        th.join()
        if failure:
            raise failure[0]
    for slot in ring:
        for r in slot or []:
            r.free()
    return dt, arrow


def spawn_ranks(args):
    """`bench.py --gpus N` started plainly (no WORLD_SIZE): start the N ranks as fresh child processes -- this process has not
    touched the GPU and never does --, hand them the rendezvous through the environment, relay rank 0's line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line = b""
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                break  # a rank failed: the others would wait for it at the next barrier for ever
            time.sleep(0.2)
        rc = max((p.poll() or 0) for p in procs if p.poll() is not None) if any(p.poll() for p in procs) else 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # exactly the processes started above
        line = procs[0].stdout.read()
        for p in procs:
            p.wait()
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    raise SystemExit(rc)


# The kernels a roofline line may name: ONE kernel each, bracketed by HIP events on the stream that launches it (a lane's stream).
# "walk" / "present" / "finish" are many kernels (and, beside another lane, waits for CU slots): phases, never the dominant kernel.
ROOF_KERNELS = {
    "seq": "zstd_seq_quads_kernel (Zstandard FSE sequences, four lanes per block; events around the kernel alone)",
    "lit": "zstd_literals_kernel (Zstandard Huffman literals, beside the sequences kernel on a stream of its own; events around the kernel alone)",
    "exec": {"zstd": "lz_exec_wave_kernel / lz_exec_kernel (LZ77 execution of Zstandard sequences)", "snappy": "lz_exec_tokens_kernel (LZ77 execution, one workgroup per chunk)",
             "lz4": "lz_exec_tokens_kernel (LZ77 execution, one workgroup per chunk)", "zlib": "lz_exec_kernel + decompress_deflate_kernel (DEFLATE execution)"},
    "stage1": {"zstd": "zstd_entropy_kernel (FSE sequences + Huffman literals, one wavefront per block)", "snappy": "lz_parse_kernel (token stage, one workgroup per chunk)",
               "lz4": "lz_parse_kernel (token stage, one workgroup per chunk)", "zlib": "inflate_parse_kernel (DEFLATE token stage)"},
    "expand": "rle2_expand_kernel (+ rle1 / byte expand: RLE expansion into the Arrow value buffers)",
    "walk_short": "rle_walk_short_kernel (run boundaries of short-run streams: every byte position parsed, pointer doubling)",
    "dict_emit": "dict_emit_kernel (dictionary keys -> Utf8 offsets + value bytes)",
}
PMC_NAMES = {"seq": "zstd_seq_quads_kernel", "lit": "zstd_literals_kernel", "exec": "lz_exec_wave_kernel", "stage1": "zstd_entropy_kernel", "expand": "rle2_expand_kernel",
             "walk_short": "rle_walk_short_kernel", "dict_emit": "dict_emit_kernel"}


def pmc_traffic(workload, comp, kernel_key, algo_bytes_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r*_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE in rocprofv3 passes of their own; PMC counters cannot be read inside this process).  The passes may have run at
    another table size: the kernel's measured bytes per ALGORITHMIC byte are applied to this run's algorithmic bytes per launch.
    FETCH_SIZE is taken raw (the gfx950 x2 correction applies to wide coalesced reads only: both figures are in the file)."""
    import glob
    from orc_rust_amd import build as _b
    tag = {"lineitem": "lineitem_" + str(comp), "c3": "c3_" + str(comp), "c2": "c2"}.get(workload)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            doc = json.load(open(f))
            if doc.get("source_digest") != _b.source_digest():
                # the passes measured other kernels than the ones running now: no figure is better than a stale one
                return None, "%s was collected for sources %s, these are %s: collect the PMC passes again (profiles/collect_r06.sh)" % (
                    os.path.basename(f), doc.get("source_digest"), _b.source_digest())
            w = doc["workloads"].get(tag)
            k = w["kernels"][PMC_NAMES[kernel_key]]
            per_algo = (k["FETCH_KB_run"] + k["WRITE_KB_run"]) * 1024.0 / w["decode_calls_in_run"] / w["algorithmic_bytes_per_step"]
            return int(per_algo * algo_bytes_per_launch), "%s: %s, (FETCH_SIZE raw + WRITE_SIZE) per algorithmic byte of the pass x this run's algorithmic bytes per launch" % (
                os.path.basename(f), PMC_NAMES[kernel_key])
        except (KeyError, TypeError, ValueError, ZeroDivisionError, OSError):
            continue
    return None, None


def roofline_of(lane_acc, comp, workload, step_algo_bytes):
    """The dominant KERNEL of the step, priced launch by launch: every column lane launches its own instance over its own columns;
    `achieved` = (sum over the lanes' launches of the algorithmic bytes THAT launch works for: the lane's staged stream bytes in +
    its Arrow bytes out) / (sum of those launches' durations) = algorithmic bytes per launch / average launch duration."""
    best = None
    for key in ("seq", "lit", "exec", "stage1", "expand", "walk_short", "dict_emit"):
        if key == "stage1" and any(a["seq"] > 0 for a in lane_acc.values()):
            continue  # (table scale: the first stage is the table kernel + the sequences kernel, priced as "seq")
        ls = [a for a in lane_acc.values() if a[key] > 0 and a["steps"]]
        if not ls:
            continue
        ms = sum(a[key] for a in ls)            # summed over lanes and steps
        by = sum(a["stream_bytes"] + a["arrow_bytes"] for a in ls)
        launches = sum(a["steps"] for a in ls)
        if best is None or ms / launches * len(ls) > best[1]:
            best = (key, ms / launches * len(ls), ms, by, launches, len(ls))
    if best is None:
        return {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 0.0, "traffic": None}
    key, _, ms, by, launches, nl = best
    name = ROOF_KERNELS[key]
    if isinstance(name, dict):
        name = name.get(comp, "block decompression kernel")
    achieved = by / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic(workload, comp, key, by / launches)
    return {"bound": "hbm", "kernel": name, "phase": key, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": src,
            "launches_per_step": nl, "algorithmic_bytes_per_launch": int(by / launches), "kernel_ms": round(ms / launches, 4),
            "kernel_ms_per_step_all_launches": round(ms / launches * nl, 4), "algorithmic_bytes_per_step": step_algo_bytes,
            "method": "sum over column lanes of (lane's stream bytes in + Arrow bytes out) / sum of the lanes' kernel spans (HIP events on each lane's stream)"}


def lanes_of(lane_acc):
    out = []
    for l in sorted(lane_acc):
        a = lane_acc[l]
        n = max(1, a["steps"])
        out.append({"lane": l, "stream_bytes": a["stream_bytes"] // n, "arrow_bytes": a["arrow_bytes"] // n, "host_ms_before_first_launch": round(a["start_ms"] / n, 3),
                    "device_ms": round(a["total_ms"] / n, 3), "zstd_tables_ms": round(a["tables"] / n, 3), "zstd_seq_quads_kernel_ms": round(a["seq"] / n, 3), "zstd_literals_kernel_ms": round(a["lit"] / n, 3),
                    "stage1_ms": round(a["stage1"] / n, 3), "exec_kernel_ms": round(a["exec"] / n, 3), "walk_ms": round(a["walk"] / n, 3),
                    "expand_ms": round(a["expand"] / n, 3), "finish_ms": round(a["finish"] / n, 3),
                    "rle_walk_short_kernel_ms": round(a["walk_short"] / n, 3), "dict_emit_kernel_ms": round(a["dict_emit"] / n, 3)})
    return out


def bench_demo12(args):
    """The reference's OWN benchmark workload (benches/arrow_reader.rs:42-67): the full read of tests/basic/data/demo-12-zlib.orc
    -- 1 920 800 rows, 9 columns, ONE stripe, ZLIB -- through the reader front (orcgpu_reader_open_file / next_batch: file bytes ->
    staged -> decoded -> copied back -> Arrow C Data batches of 8192 rows, every batch released at once).  A step = one whole read,
    open included.  The worst case for this design: a 46 KB file whose DEFLATE chunks are a handful of serial chains -- latency, not
    throughput.  Beside it: Apache ORC C++ through PyArrow (311 ms on the survey container) and the CPU oracle on one core."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    path = os.path.join(ROOT, "tests", "golden", "data", "demo-12-zlib.orc")
    if not os.path.isdir("/usr/share/zoneinfo") and "TZDIR" not in os.environ:
        import tzdata
        os.environ["TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
    import pyarrow as pa
    import pyarrow.orc as orc
    table = orc.ORCFile(path).read()
    rows, arrow_bytes, file_bytes = table.num_rows, table.nbytes, os.path.getsize(path)
    cpu = None
    if not args.no_cpu:
        best_pa = min(_timed(lambda: orc.ORCFile(path).read()) for _ in range(5))
        import orcfile
        import oracle_lib as O

        def oracle_read():
            f = orcfile.OrcFile(path)
            for s_ in f.stripes:
                for _, cid, _t in f.flat_columns():
                    col = f.oracle_column(s_, cid)
                    left = s_.number_of_rows
                    while left > 0:
                        b = col.next_batch(min(8192, left))
                        assert b["status"] == O.OK
                        left -= 8192
                    col.close()
        best_or = min(_timed(oracle_read) for _ in range(2))
        cpu = {"value": round(arrow_bytes / best_or / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port", "ms": round(best_or * 1e3, 1),
               "sample": "the whole file, every flat column batch by batch (8192), oracle/liborc_oracle.so, best of 2",
               "pyarrow_orc_cpp": {"ms": round(best_pa * 1e3, 1), "GBps": round(arrow_bytes / best_pa / 1e9, 4), "cores": 1,
                                   "what": "pyarrow.orc.ORCFile(path).read() = Apache ORC C++, best of 5 on this host (BASELINE.md: 311 ms on the survey container)"}}
    import torch
    torch.cuda.set_device(0)
    from orc_rust_amd import ArrowReaderBuilder, capi
    ctx = capi.Context(0)
    L = ctx.L
    # parity first: the reader's batches are the PyArrow table
    batches = list(ArrowReaderBuilder.try_new(path, ctx).build())
    got = pa.Table.from_batches(batches)
    assert got.num_rows == rows
    for name in table.schema.names:
        g, w = got.column(name), table.column(name)
        assert g.equals(w.cast(g.type) if g.type != w.type else w), name
    del batches, got

    class ArrowArray(C.Structure):
        _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                    ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]

    class ArrowSchema(C.Structure):
        _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_void_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                    ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]

    def read_once(prefetch):
        h = C.c_void_p()
        t0 = time.perf_counter()
        assert L.orcgpu_reader_open_file(ctx.h, path.encode(), C.byref(h)) == 0
        L.orcgpu_reader_set_prefetch(h, prefetch)
        n = 0
        while True:
            a, sc = ArrowArray(), ArrowSchema()
            rc = L.orcgpu_reader_next_batch(h, C.byref(a), C.byref(sc))
            if rc == 110:  # ORCGPU_END_OF_FILE
                break
            assert rc == 0, rc
            n += a.length
            a.release(C.addressof(a))
            sc.release(C.addressof(sc))
        dt = time.perf_counter() - t0
        L.orcgpu_reader_close(h)
        assert n == rows
        return dt
    modes = {}
    for prefetch in (0, 2):
        for _ in range(max(1, args.warmup)):
            read_once(prefetch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            read_once(prefetch)
        modes[prefetch] = (time.perf_counter() - t0) / args.steps
    dt = modes[0]
    algo = file_bytes + arrow_bytes
    out = {"metric": "decoded GB/s + Mrows/s into Arrow", "value": round(arrow_bytes / dt / 1e9, 3), "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64/u8", "data": "the reference's fixture file",
           "config": {"workload": "the reference's own benchmark (benches/arrow_reader.rs:42-67): full read of demo-12-zlib.orc, %d rows, 9 columns, 1 stripe, ZLIB, "
                                  "through orcgpu_reader_* (file -> Arrow C Data batches in pinned host memory: PCIe both ways INCLUDED)" % rows,
                      "rows": rows, "stripes": 1, "batch_size": 8192, "compression": "zlib", "parallelism": "one reader, serial (prefetch 0)"},
           "mrows_per_s": round(rows / dt / 1e6, 1), "file_bytes": file_bytes, "arrow_bytes_out": arrow_bytes,
           "ms_per_read_with_read_ahead": round(modes[2] * 1e3, 4),
           "roofline": {"bound": "hbm", "kernel": None, "achieved": round(algo / dt / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(algo / dt / 1e9 / HBM_PEAK_GBPS, 6),
                        "traffic": None, "note": "a whole-file read, host to host: latency of one stripe's launch chain and its DEFLATE chains, not a throughput figure"}}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out))
    sys.stdout.flush()


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--row-index", action="store_true",
                    help="c3: stage the PRESENT and DATA streams with their ROW_INDEX positions (BASELINE's C3 is measured without)")
    ap.add_argument("--no-row-index", action="store_true",
                    help="c2-adv / c2-rowgroup: stage the stream without its ROW_INDEX positions (a file written without indexes)")
    ap.add_argument("--workload", default="lineitem", choices=["lineitem", "c2", "c2-direct", "c2-delta", "c2-arange", "c2-adv", "c2-rowgroup", "c3", "c5", "demo12"],
                    help="lineitem (default) = the headline; the others are BASELINE.md's remaining configs, recorded under profiles/")
    ap.add_argument("--compression", default=None, choices=[None, "none", "zstd", "snappy", "lz4", "zlib"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (weak) / of the table (strong); 0 = the config's own size")
    ap.add_argument("--sf", type=float, default=0.0, help="lineitem: TPC-H scale factor per GPU (weak) / of the table (strong); default 12.5 = one GPU's share of C4's SF100")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU gets a one-GPU share, the table grows with N (N = 8: C4's SF100); strong: one fixed table sharded over the ranks")
    ap.add_argument("--columns", default="", help="lineitem, profiling runs: decode only these columns (1-based positions, e.g. 1,4,16) of every stripe")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--skip-check", action="store_true", help="profiling runs: skip the checks before timing")
    ap.add_argument("--no-e2e", action="store_true", help="profiling runs: skip the pipelined host-to-host measurement behind the timed region")
    args = ap.parse_args()

    if args.workload == "demo12":
        if args.gpus != 1:
            raise SystemExit("--workload demo12 is a one-file, one-stripe read: one GPU")
        return bench_demo12(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...` or plainly (no WORLD_SIZE) so that it starts "
                         "its own ranks" % (args.gpus, world, args.gpus, args.gpus))
    # The workload and the CPU baseline come first: both fork worker processes, which must happen before
    # this process initialises the GPU.
    stripes, comp, label, shard_desc, plan = build_workload(args, rank, world)
    cpu = None
    cpu_helper = None
    if rank == 0 and not args.no_cpu and stripes:
        # rank 0 only, on ITS share of the workload; timed behind the GPU work (CpuBaselineHelper: forked now, before the GPU is touched)
        cpu_helper = CpuBaselineHelper(stripes, comp)
    if world > 1:
        # one process per GPU on one host: every rank's staging helpers get the rank's share of the usable cores
        os.environ.setdefault("ORCGPU_STAGE_THREADS", str(max(0, min(6, host_workers(world) - 1))))
    import torch
    dist = None
    # BENCH_BACKEND=gloo is a dry-run aid for boxes with fewer GPUs than ranks (ranks then share devices and
    # the collective runs on CPU tensors); the driver's runs use the default, RCCL ("nccl").
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count():  # (none: orcgpu_open below reports it)
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)

    from orc_rust_amd import capi, shard
    from orc_rust_amd.gen import workloads as W
    ctx = capi.Context(local_rank)
    # host buffers -> HBM (not part of `value`).  Staged twice: the first pass also pays for the pinned pieces, the copy
    # threads and the arenas (hipHostMalloc / hipMalloc); the second one, timed, finds them in the context's pools -- the
    # state a reader is in from its second stripe on.
    for s_ in [ctx.stage(n, streams, cols, compression=comp) for n, cols, streams, _, _ in stripes]:
        s_.free()
    torch.cuda.synchronize()
    t_stage = time.perf_counter()
    staged = [ctx.stage(n, streams, cols, compression=comp) for n, cols, streams, _, _ in stripes]
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t_stage
    stream_bytes = sum(s.nbytes() for s in staged)
    rows = sum(s[0] for s in stripes)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    results = ctx.decode(staged) if staged else []
    err_word = 0
    t_check = time.perf_counter()
    for k, ((n, cols, streams, expect, sums), res) in enumerate(zip(stripes, results)):
        st = res.status()
        err_word |= st[0]
        assert st[0] == 0, st
        if args.skip_check:
            continue
        if expect:
            # buffer by buffer: every decoded Arrow buffer equals what the generated values imply
            W.check_result(res, cols, expect)
        if sums is not None:
            # every stripe: whole-buffer sums, the expected side reduced by the process that generated the stripe
            check_sums(torch, res, cols, sums, "stripe %d of rank %d" % (k, rank))
    t_check = time.perf_counter() - t_check
    arrow_bytes = sum(r.arrow_bytes for r in results)

    for _ in range(args.warmup):
        if staged:
            ctx.decode(staged, results)
    barrier()
    t0 = time.perf_counter()
    phase = {k: 0.0 for k in capi.Context.PHASES + ("decompress_stage1", "decompress_tables")}
    tot_ms = 0.0
    lane_acc = {}  # lane -> sums over the timed steps of its kernels' spans and of the bytes its launches worked for
    for _ in range(args.steps):
        if staged:
            ctx.decode(staged, results)
            tot_ms += ctx.timing()[0]
            for k, v in ctx.phase_ms().items():
                phase[k] += v
            for ls in ctx.lane_stats():
                a = lane_acc.setdefault(ls["lane"], {"steps": 0, "stream_bytes": 0, "arrow_bytes": 0, "start_ms": 0.0, "total_ms": 0.0, "seq": 0.0, "lit": 0.0, "exec": 0.0,
                                                     "stage1": 0.0, "tables": 0.0, "expand": 0.0, "walk": 0.0, "finish": 0.0, "walk_short": 0.0, "dict_emit": 0.0})
                a["steps"] += 1
                a["stream_bytes"] += ls["stream_bytes"]
                a["arrow_bytes"] += ls["arrow_bytes"]
                a["start_ms"] += ls["start_ms"]
                a["total_ms"] += ls["total_ms"]
                a["seq"] += ls["seq_kernel_ms"]
                a["lit"] += ls["literals_kernel_ms"]
                a["exec"] += ls["exec_kernel_ms"]
                a["stage1"] += ls["phase_ms"]["decompress_stage1"]
                a["tables"] += ls["phase_ms"]["decompress_tables"]
                a["expand"] += ls["phase_ms"]["expand"]
                a["walk"] += ls["phase_ms"]["walk"]
                a["finish"] += ls["phase_ms"]["finish"]
                a["walk_short"] += ls["walk_short_kernel_ms"]
                a["dict_emit"] += ls["dict_emit_kernel_ms"]
    torch.cuda.synchronize()
    my_dt = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    # the way back (outside the timed region, like staging): ONE pinned device-to-host copy per result arena; timed on
    # the second round, when the pinned host copies exist (the first one allocates them).  At table scale a sample of the
    # stripes is fetched (pinned host memory for every result of a 13 GB table is not what a reader holds either).
    fetch_n = min(len(results), 4)
    for r in results[:fetch_n]:
        r.fetch()
    if staged:
        ctx.decode(staged[:fetch_n], results[:fetch_n])
    t_fetch = time.perf_counter()
    for r in results[:fetch_n]:
        r.fetch()
    t_fetch = time.perf_counter() - t_fetch
    fetch_bytes = sum(r.arrow_bytes for r in results[:fetch_n])
    string_bytes = sum(sum((len(e["values"]) if "values" in e else 0) for cid, e in expect.items() if "lengths" in e) for _, _, _, expect, _ in stripes)
    unit_rows = sum(n * len(cols) for n, cols, _, _, _ in stripes)  # rows x columns this rank decoded
    my_ms = my_dt / args.steps * 1e3
    if dist is not None:
        tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the path's only exchange: {rows x columns, value bytes of this rank's string columns, Arrow bytes, stream bytes, units, error word, us per step}
        allc = shard.gather_counts([unit_rows, string_bytes, arrow_bytes, stream_bytes, len(plan["units"]), err_word, int(my_ms * 1e3)], dist, coll_dev)
        total_arrow = sum(c[2] for c in allc)
        total_stream = sum(c[3] for c in allc)
        assert all(c[5] == 0 for c in allc), "a rank reported a decode error"
        assert sum(c[4] for c in allc) == plan["n_stripes"] * plan["n_columns"], "every (stripe, column) unit must be decoded exactly once"
        assert sum(c[0] for c in allc) == plan["rows"] * plan["n_columns"], "decoded rows do not add up: %s" % allc
        total_rows = plan["rows"]
        tot_load = float(sum(plan["load_estimate"])) or 1.0
        per_rank = [{"rows_x_columns": c[0], "string_bytes": c[1], "arrow_bytes": c[2], "stream_bytes": c[3], "units": c[4],
                     "ms_per_step": c[6] / 1e3, "load_estimate_frac": round(plan["load_estimate"][r] / tot_load, 4)} for r, c in enumerate(allc)]
    else:
        total_rows, total_arrow, total_stream = rows, arrow_bytes, stream_bytes
        per_rank = None
    ms_per_step = dt / args.steps * 1e3
    value = total_arrow / (dt / args.steps) / 1e9
    phase = {k: v / args.steps for k, v in phase.items()}
    roof = roofline_of(lane_acc, comp, args.workload, stream_bytes + arrow_bytes)
    e2e_dt, e2e_bytes = pipelined_end_to_end(ctx, stripes, comp) if len(stripes) >= 2 and not args.no_e2e else (0.0, 0)
    h2d = stream_bytes / t_stage / 1e9 if t_stage > 0 else None
    d2h = fetch_bytes / t_fetch / 1e9 if t_fetch > 0 and fetch_bytes else None
    serial_s = (t_stage + dt / args.steps + (arrow_bytes / (d2h * 1e9) if d2h else 0.0))
    out = {
        "metric": "decoded GB/s + Mrows/s into Arrow, TPC-H lineitem stripe" if args.workload == "lineitem" else "decoded GB/s + Mrows/s into Arrow",
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "int64" if args.workload.startswith(("c2", "c5")) else ("u8" if args.workload == "c3" else "int64/i128/u8"),
        "data": "synthetic",
        "config": {"workload": label, "rows": total_rows, "stripes": plan["n_stripes"], "batch_size": 8192, "compression": comp,
                   "parallelism": shard_desc},
        "mrows_per_s": round(total_rows / (dt / args.steps) / 1e6, 1),
        "stream_bytes_in": total_stream, "arrow_bytes_out": total_arrow,
        "device_ms_per_step": round(tot_ms / args.steps, 4),
        "phase_ms": {k: round(v, 4) for k, v in phase.items()},
        # staging the host stream buffers is outside the timed region; the PCIe-inclusive rates are reported for DESIGN.md only
        "h2d_stage_ms": round(t_stage * 1e3, 3), "h2d_GBps": round(h2d, 2) if h2d else None,
        "d2h_fetch_sample": "%d of %d stripes" % (fetch_n, len(results)), "d2h_GBps": round(d2h, 2) if d2h else None,
        "pcie_inclusive_GBps": round(arrow_bytes / (t_stage + dt / args.steps) / 1e9, 2),
        "end_to_end_serial_GBps": round(arrow_bytes / serial_s / 1e9, 2),
        # host stream buffers in, pinned host Arrow buffers out: a staging thread, four stripes per decode call, copies back on
        # their own stream (stage k + 1 / decode k / copy back k - 1 overlap): measured, see pipelined_end_to_end
        "end_to_end_GBps": round(e2e_bytes / e2e_dt / 1e9, 2) if e2e_dt > 0 else None,
        "end_to_end_ms_per_stripe": round(e2e_dt / len(stripes) * 1e3, 3) if e2e_dt > 0 else None,
        "setup": {"generate_s": round(plan["gen_s"], 2), "generate_procs": plan["gen_procs"], "check_s": round(t_check, 2),
                  "checked": "skipped" if args.skip_check else ("first stripe buffer by buffer + every stripe by whole-buffer sums" if args.workload == "lineitem" else "every stripe buffer by buffer")},
        # device memory in use at the end of the run (staged streams, workspace, results, pinned mirrors do not count): what decides
        # which table sizes fit one GPU's 288 GB
        "hbm_bytes_in_use": (lambda fr_to: int(fr_to[1] - fr_to[0]))(torch.cuda.mem_get_info()),
        "roofline": dict(roof, whole_step_frac=round((stream_bytes + arrow_bytes) / (my_dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4)),
        "lanes": lanes_of(lane_acc),
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    if rank == 0:
        if cpu_helper is not None:
            # (N > 1: the other ranks are at the barrier below meanwhile, their cores idle -- the timed region is long over)
            cpu = cpu_helper.run()
            if world > 1:
                cpu["sample"] += "; rank 0's share of the %d-rank job" % world
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
