"""Experiment: stripes split in time (two contexts, each decoding half of the stripes with all columns, the second started late)
against the product's split by columns (one context, two column lanes in lock step)."""
import os, sys, time, threading, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (the repository: this script lives in profiles/)
sys.path.insert(0, ROOT)
import bench
sys.argv = [sys.argv[0]]
ap_args = argparse.Namespace(gpus=1, steps=10, warmup=3, row_index=False, no_row_index=False, workload="lineitem", compression=None, rows=0,
                             sf=float(os.environ.get("SF", "12.5")), scaling="weak", no_cpu=True, skip_check=True, no_e2e=True)
stripes, comp, label, shard_desc, plan = bench.build_workload(ap_args, 0, 1)
from orc_rust_amd import capi
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def sync(): hip.hipDeviceSynchronize()
print(label, len(stripes), flush=True)

def run(parts, lanes, delays, steps=8, warm=3):
    os.environ["ORCGPU_LANES"] = str(lanes)
    ctxs = [capi.Context(0) for _ in parts]
    staged = [[c.stage(n, streams, cols, compression=comp) for n, cols, streams, _, _ in part] for c, part in zip(ctxs, parts)]
    results = [c.decode(s) for c, s in zip(ctxs, staged)]
    sync()
    def one():
        th = []
        t0 = time.perf_counter()
        for k, (c, s, r) in enumerate(zip(ctxs, staged, results)):
            def work(c=c, s=s, r=r, d=delays[k]):
                if d: time.sleep(d * 1e-3)
                c.decode(s, r)
            t = threading.Thread(target=work); t.start(); th.append(t)
        for t in th: t.join()
        sync()
        return (time.perf_counter() - t0) * 1e3
    for _ in range(warm): one()
    ts = [one() for _ in range(steps)]
    for r_ in results:
        for r in r_: r.free()
    for s_ in staged:
        for s in s_: s.free()
    for c in ctxs: c.close()
    return sorted(ts)[len(ts) // 2], min(ts)

h = len(stripes) // 2
print("one ctx, 2 column lanes", run([stripes], 2, [0]), flush=True)
print("one ctx, 1 lane", run([stripes], 1, [0]), flush=True)
for d in (0, 4, 8, 12, 16, 20):
    print("two ctx (halves of the stripes), 1 lane each, delay", d, run([stripes[:h], stripes[h:]], 1, [0, d]), flush=True)
for d in (0, 8, 16):
    print("two ctx (halves), 2 lanes each, delay", d, run([stripes[:h], stripes[h:]], 2, [0, d]), flush=True)
q = len(stripes) // 4
for d in (0, 5, 10):
    print("four ctx (quarters), 1 lane each, delay step", d, run([stripes[:q], stripes[q:2*q], stripes[2*q:3*q], stripes[3*q:]], 1, [0, d, 2*d, 3*d]), flush=True)
