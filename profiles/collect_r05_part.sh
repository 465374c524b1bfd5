#!/bin/bash
# Re-runs single workloads of profiles/collect_r05.sh into the same gpurun_out/r05c/ (no clean-up): for the lines a late kernel
# change touches.  Usage on the GPU box: bash profiles/collect_r05_part.sh <tag> <bench args...> [-- <tag> <bench args...>]...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  ( cd $R && timeout 600 python bench.py "$@" --steps 10 --warmup 3 --no-cpu 2> $O/$tag.err | tail -1 > $O/line_$tag.json )
  ( cd $R && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$tag -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu --skip-check --no-e2e > /dev/null 2> $O/prof_$tag.err )
  f=$(find $O/raw_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_$tag.csv
  rm -rf $O/raw_$tag
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run "${args[@]}"
ls $O | wc -l
