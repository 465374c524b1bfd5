#!/bin/bash
# Runs on the GPU box: everything profiles/ is built from.  Output under gpurun_out/final/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err
for k in direct delta arange; do python bench.py --steps 10 --warmup 2 --no-cpu --kind $k > $O/bench_$k.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --skip-check > $O/c2_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c2_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --skip-check > $O/c2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c2_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --skip-check > $O/c2_write.log 2>&1
for c in none snappy; do
  export COMP=$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_$c -- python3 $R/profiles/bench_c3.py > $O/c3_$c.log 2>&1
done
unset COMP
cd $R
for c in none snappy lz4 zlib zstd; do COMP=$c python profiles/bench_c3.py 2>/dev/null | tail -1 > $O/c3_line_$c.json; done
ls -la $O
for c in none zstd; do COMP=$c python profiles/bench_mixed.py 2>/dev/null | tail -1 > $O/mixed_line_$c.json; done
ls $O | head -40
