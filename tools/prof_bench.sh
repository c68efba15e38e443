#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command; writes gpurun_out/prof_bench/{kernel_stats.csv,line.json}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o b -- python bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/line.json
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/raw
head -40 $OUT/kernel_stats.csv | cut -c1-200
