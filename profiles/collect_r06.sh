#!/bin/bash
# Runs on the GPU box (through gpurun): everything the r06_* files of profiles/ are built from.  Output under gpurun_out/r06c/.
#   gpurun --timeout 3000 -- 'bash profiles/collect_r06.sh'        then here:  python profiles/summarise_r06.py
# Every profiler pass runs under its own `timeout` (a pass that hangs must not take the rest with it) and is checked by the
# summariser: a missing pass is an error there, never a zero.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06c
rm -rf $O; mkdir -p $O
cd $R
# 1. the driver's line: default workload (lineitem, one GPU's share of SF100 = SF 12.5, Zstandard), CPU baseline included
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err
# 2. where GB/s saturates: scale factors 1 .. 50 of the same table, then C4's CONFIGURED size -- SF 100, the whole table on ONE GPU --
#    if the SF 50 run says it fits: device memory in use grows linearly with the table (hbm_bytes_in_use of the line), SF 100 is
#    attempted only when twice the SF 50 figure stays below 250 GB of the 288 (a box that runs out of memory is lost)
for sf in 1 2 4 8 12.5 25 50; do
  timeout 900 python bench.py --sf $sf --steps 5 --warmup 2 --no-cpu --no-e2e 2> $O/sf_$sf.err | tail -1 > $O/sf_$sf.json
done
python3 - $O <<'PY' > $O/sf100_decision.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1] + '/sf_50.json').read())
    need = 2 * d['hbm_bytes_in_use']
    print('fits' if need < 250e9 else 'too large', need)
except Exception as e:
    print('unknown', e)
PY
if grep -q '^fits' $O/sf100_decision.txt; then
  timeout 1500 python bench.py --sf 100 --steps 3 --warmup 1 --no-cpu --no-e2e --skip-check 2> $O/sf_100.err | tail -1 > $O/sf_100.json
fi
# 3. one bench line + one kernel table per workload / codec
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  ( cd $R && timeout 600 python bench.py "$@" --steps 10 --warmup 3 --no-cpu 2> $O/$tag.err | tail -1 > $O/line_$tag.json )
  ( cd $R && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$tag -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/prof_$tag.err )
  f=$(find $O/raw_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_$tag.csv
  rm -rf $O/raw_$tag
}
run lineitem_zstd --workload lineitem --compression zstd
for c in snappy lz4 zlib none; do run lineitem_$c --workload lineitem --compression $c --sf 4; done
run c2 --workload c2
run c2_adv --workload c2-adv --rows 24000000
run c2_adv_noindex --workload c2-adv --rows 24000000 --no-row-index
run c2_rowgroup --workload c2-rowgroup
run c2_rowgroup_noindex --workload c2-rowgroup --no-row-index
for c in none snappy zstd lz4 zlib; do run c3_$c --workload c3 --compression $c; done
for c in none snappy; do run c3_${c}_index --workload c3 --compression $c --row-index; done
run c5_lz4 --workload c5 --compression lz4
# 3a. the reference's own benchmark workload (benches/arrow_reader.rs:42-67): the whole-file read of demo-12-zlib.orc through the reader
( cd $R && timeout 600 python bench.py --workload demo12 2> $O/demo12.err | tail -1 > $O/line_demo12.json )
( cd $R && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_demo12 -- python3 bench.py --workload demo12 --steps 5 --warmup 2 --no-cpu > /dev/null 2> $O/prof_demo12.err )
f=$(find $O/raw_demo12 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_demo12.csv
rm -rf $O/raw_demo12
# 3a'. the post-decompression path alone at the headline's size (lineitem, uncompressed, SF 12.5)
run lineitem_none_sf12 --workload lineitem --compression none
# 3b. the timeline of one headline step (kernel, queue, start, end): what runs beside what
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw_trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/prof_trace.err )
python3 $R/profiles/timeline.py $O/raw_trace 30 > $O/timeline_lineitem_zstd.txt 2>&1
( cd /tmp && ORCGPU_LANES=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw_trace1 -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/prof_trace1.err )
python3 $R/profiles/timeline.py $O/raw_trace1 30 > $O/timeline_lineitem_zstd_one_lane.txt 2>&1
rm -rf $O/raw_trace1
rm -rf $O/raw_trace
# 3c. what a row selection / a predicate costs with row-group pruning; the reader's own rate
( cd $R && timeout 600 python profiles/select_cost.py 24000000 > $O/select_cost.json 2> $O/select_cost.err )
( cd $R && timeout 600 python profiles/reader_rate.py 24000000 > $O/reader_rate.json 2> $O/reader_rate.err )
# 3d. the device encoders (8(f)-4): rate on device-resident Int64 columns, and their kernel table
( cd $R && timeout 600 python profiles/encode_rate.py 48000000 > $O/encode_rate.json 2> $O/encode_rate.err )
( cd $R && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_encode -- python3 profiles/encode_rate.py 16000000 > /dev/null 2> $O/prof_encode.err )
f=$(find $O/raw_encode -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_encode.csv
rm -rf $O/raw_encode
# 4. HBM traffic of the headline (SF 4: the table-scale kernels, 11 stripes) and of C3 / C2: FETCH_SIZE and WRITE_SIZE in passes of
#    their own (no trace domain beside --kernel-trace), each bounded
pmc() {  # tag, counter, bench args...
  tag=$1; cnt=$2; shift; shift
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/raw_pmc -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/pmc_${tag}_$cnt.err )
  f=$(find $O/raw_pmc -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/pmc_${tag}_$cnt.csv
  rm -rf $O/raw_pmc
}
for cnt in FETCH_SIZE WRITE_SIZE; do
  pmc lineitem_zstd_sf12 $cnt --workload lineitem --compression zstd
  pmc lineitem_zstd $cnt --workload lineitem --compression zstd --sf 4
  pmc c3_none $cnt --workload c3 --compression none
  pmc c2 $cnt --workload c2
done
( cd $R && timeout 300 python bench.py --workload lineitem --compression zstd --sf 4 --steps 5 --warmup 2 --no-cpu --no-e2e 2> /dev/null | tail -1 > $O/line_lineitem_zstd_sf4.json )
# run-to-run spread of the headline: five more lines
for i in 1 2 3 4 5; do ( cd $R && timeout 300 python bench.py --no-cpu --no-e2e --skip-check 2> /dev/null | tail -1 > $O/repeat_$i.json ); done
# which sources these passes measured (bench.py refuses to price `roofline.traffic` with passes of other sources)
( cd $R && python3 -c "from orc_rust_amd import build as b; print(b.source_digest())" > $O/source_digest.txt )
# the counter files are large (one row per dispatch): keep per-kernel averages only
python3 - $O <<'PY'
import csv, sys, glob, json, collections, os
O = sys.argv[1]
out = {}
for f in sorted(glob.glob(O + '/pmc_*_*.csv')):
    tag = os.path.basename(f)[4:-4]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    out[tag] = {k: {"launches": len(v), "avg": sum(v) / len(v), "sum": sum(v)} for k, v in acc.items()}
    os.remove(f)
json.dump(out, open(O + '/pmc_per_kernel.json', 'w'), indent=1)
PY
ls -la $O | head -90
