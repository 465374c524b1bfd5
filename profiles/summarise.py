#!/usr/bin/env python3
"""Turns gpurun_out/final/ (written by profiles/collect.sh on the GPU box) into the tracked files of profiles/."""
import csv, glob, json, collections, shutil
O, P = 'gpurun_out/final', 'profiles'
for name, dst in (('c2_stats', 'r01_c2_kernel_stats.csv'), ('c3_none', 'r01_c3_none_kernel_stats.csv'), ('c3_snappy', 'r01_c3_snappy_kernel_stats.csv')):
    shutil.copy(glob.glob(f'{O}/{name}/**/*kernel_stats.csv', recursive=True)[0], f'{P}/{dst}')
shutil.copy(f'{O}/bench_line.json', f'{P}/r01_bench_line.json')
json.dump({k: json.load(open(f'{O}/bench_{k}.json')) for k in ('direct', 'delta', 'arange')}, open(f'{P}/r01_c2_variants.json', 'w'), indent=1)
json.dump({c: json.loads(open(f'{O}/c3_line_{c}.json').read()) for c in ('none', 'snappy', 'lz4', 'zlib', 'zstd')}, open(f'{P}/r01_c3_lines.json', 'w'), indent=1)
json.dump({c: json.loads(open(f'{O}/mixed_line_{c}.json').read()) for c in ('none', 'zstd')}, open(f'{P}/r01_mixed_lines.json', 'w'), indent=1)
res = {}
for cnt, d in (('FETCH_SIZE', 'c2_fetch'), ('WRITE_SIZE', 'c2_write')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True)[0])):
        if r['Counter_Name'] == cnt:
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    res[cnt] = {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = res['FETCH_SIZE']['rle2_expand_kernel'], res['WRITE_SIZE']['rle2_expand_kernel']
bl = json.load(open(f'{O}/bench_line.json'))
json.dump({
    "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu --skip-check",
    "per_kernel_average_KB": res,
    "rle2_expand_kernel": {
        "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
        "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (guide: x2 for wide coalesced streams); this kernel mixes 16-byte and 8-byte per-lane loads, so both the raw and the x2 figure are given; WRITE_SIZE is exact for its 16-byte per-lane stores",
        "hbm_bytes_raw": (fe + wr) * 1024, "hbm_bytes_fetch_x2": (2 * fe + wr) * 1024},
    "rle2_expand_kernel_bytes_per_launch": int((2 * fe + wr) * 1024),
    "algorithmic_bytes_per_launch": bl["roofline"]["algorithmic_bytes_per_launch"]}, open(f'{P}/r01_pmc_traffic.json', 'w'), indent=1)
print(open(f'{P}/r01_bench_line.json').read()[:300])
