#!/bin/bash
# Derived counters of ONE kernel of the microbench: [ONLY=<microbench case, e.g. K2_reward>] tools/pmc_kernel.sh <kernel-name substring> <n> <counter> [<counter> ...]
# (one rocprofv3 --pmc pass per counter group of <= 3; gpurun forbids mixing --pmc with trace domains other than kernel-trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=$1; N=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel
rm -rf $OUT; mkdir -p $OUT
i=0
while [ $# -gt 0 ]; do
  grp="$1 ${2:-} ${3:-}"; shift; [ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o mb -- python tools/microbench.py $N ${ONLY:+--only $ONLY --no-variants} > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "$K" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-28s launches %4d  avg %.4g" % (k, n, s / max(n, 1)))
PY
rm -rf $OUT/p*/
