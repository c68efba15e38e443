#!/bin/bash
# Per-kernel time of one microbench case: tools/kstats.sh <case, e.g. K5_gae> <n> -> rocprofv3 --kernel-trace --stats summary (top rows)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$1_$2
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o mb -- python tools/microbench.py $2 --only $1 --no-variants > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("%-70s calls %6s  avg %10.1f us  total %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
