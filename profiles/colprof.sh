#!/bin/bash
# Per-column-group cost of the post-decompression path (lineitem, uncompressed) -- where the walk / expansion / finisher time goes.
#   gpurun --timeout 1200 -- 'bash profiles/colprof.sh [SF] [COMP]'   -> gpurun_out/colprof/
R=${GRAFT_REPO_ROOT:-$(pwd)}
SF=${1:-4}
COMP=${2:-none}
O=$R/gpurun_out/colprof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 1 2,3 4 5,6,7,8 9,10,14,15 11,12,13 16 all; do
  a="--columns $g"; [ $g = all ] && a=""
  tag=$(echo $g | tr , _)
  ( cd $R && timeout 300 python bench.py --workload lineitem --compression $COMP --sf $SF $a --steps 10 --warmup 3 --no-cpu --no-e2e --skip-check 2> $O/$tag.err | tail -1 > $O/line_$tag.json )
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 bench.py --workload lineitem --compression $COMP --sf $SF $a --steps 5 --warmup 2 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/prof_$tag.err )
  f=$(find $O/raw -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cut -d, -f1-4 "$f" | head -14 > $O/kstats_$tag.csv
  rm -rf $O/raw
done
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + '/line_*.json')):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), d['ms_per_step'], d['value'], {k: v for k, v in d['phase_ms'].items() if v > 0.01})
    except Exception as e:
        print(f, 'failed', e)
PY
