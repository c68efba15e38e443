#!/bin/bash
# HBM traffic of the kernels by PMC counters: separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md: TCC slots do not fit both; gpurun forbids mixing --pmc with trace domains other than kernel-trace).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_microbench
mkdir -p $OUT
N=${1:-65536}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o mb -- python tools/microbench.py $N > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in f:
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != c: continue
            k = r["Kernel_Name"][:60]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    res[c] = acc
keys = sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"]))
print("kernel,launches,FETCH_SIZE_avg,WRITE_SIZE_avg")
for k in keys:
    if "egp::" not in k and "k_dynamics" not in k and "k_gemm" not in k: continue
    f, nf = res["FETCH_SIZE"].get(k, [0, 1]); w, nw = res["WRITE_SIZE"].get(k, [0, 1])
    print("%s,%d,%.1f,%.1f" % (k, nf, f / max(nf, 1), w / max(nw, 1)))
PY
ls $OUT/FETCH_SIZE | head; find $OUT -name "*.csv" | head -5; rm -rf $OUT/*/mb_kernel_trace.csv
