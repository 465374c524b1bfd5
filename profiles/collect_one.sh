#!/bin/bash
# Runs on the GPU box: one workload of profiles/collect_r05.sh again (bench line + kernel table) into gpurun_out/r05c/.
#   gpurun -- 'bash profiles/collect_one.sh <tag> <bench args ...>'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py "$@" --steps 10 --warmup 3 --no-cpu 2> $O/$tag.err | tail -1 > $O/line_$tag.json )
( cd $R && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$tag -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/prof_$tag.err )
f=$(find $O/raw_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_$tag.csv
rm -rf $O/raw_$tag
cut -c1-200 $O/line_$tag.json
