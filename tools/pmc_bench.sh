#!/bin/bash
# PMC HBM traffic of the egp kernels inside the real bench workload (separate passes per counter).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench_r01
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o b -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-k1-events > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != c or "egp::" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"][:64] + "|grid=" + r.get("Grid_Size", "?")
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    res[c] = acc
print("kernel|grid,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg")
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    f, nf = res["FETCH_SIZE"].get(k, [0, 0]); w, nw = res["WRITE_SIZE"].get(k, [0, 0])
    print("%s,%d,%.2f,%.2f" % (k, max(nf, nw), f / max(nf, 1), w / max(nw, 1)))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
