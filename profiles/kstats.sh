#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 --kernel-trace --stats of one bench command, per-kernel table printed
# and kept as gpurun_out/<tag>/kernel_stats.csv.
#   bash profiles/kstats.sh <tag> <python script and args ...>       e.g.  bash profiles/kstats.sh li bench.py --steps 5 --warmup 2 --no-cpu --skip-check
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 "$@" > $O/run.json 2> $O/run.err )
f=$(find $O/raw -name '*kernel_stats.csv' | head -1)
cp "$f" $O/kernel_stats.csv
rm -rf $O/raw
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-44s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[:24]:
    print("%-44s %7s %12.1f %12.2f %7s" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -c 600 $O/run.json
