#!/bin/bash
# state_reg step (ResNet-18 encoder + bi-LSTM + MLP): kernel stats of the channels_last variants and the matrix-core
# utilisation of the default one (bf16 encoder copy with float32 master weights, BASELINE config 4)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_statereg
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o s -- python tools/statereg_bench.py 256 channels_last > $OUT/bench.log 2>&1
cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/raw
timeout 900 rocprofv3 --pmc MfmaUtil VALUBusy --kernel-trace --output-format csv -d $OUT/pmc -o s -- python tools/statereg_bench.py 256 Bf16Shadow > $OUT/pmc.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"][:90]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
# the kernels that use the matrix cores first (by launches), then the 15 most launched of the rest (a fresh box's first steps run
# MIOpen's find phase: thousands of im2col / Tensile candidates with ~0 % utilisation -- they are not the steady-state step)
util = lambda v: v.get("MfmaUtil", [0, 1])[0] / max(1, v.get("MfmaUtil", [0, 1])[1])
on_mfma = sorted([kv for kv in acc.items() if util(kv[1]) >= 1.0], key=lambda kv: -kv[1].get("MfmaUtil", [0, 0])[1])
rest = sorted([kv for kv in acc.items() if util(kv[1]) < 1.0], key=lambda kv: -kv[1].get("MfmaUtil", [0, 0])[1])
rows = on_mfma + rest[:15]
with open("$OUT/mfma_util.csv", "w") as f:
    f.write("kernel,launches,MfmaUtil_avg_pct,VALUBusy_avg_pct\n")
    for k, v in rows:
        m, b = v.get("MfmaUtil", [0, 1]), v.get("VALUBusy", [0, 1])
        f.write('"%s",%d,%.2f,%.2f\n' % (k, m[1], m[0] / max(1, m[1]), b[0] / max(1, b[1])))
print(open("$OUT/mfma_util.csv").read()[:3000])
PY
rm -rf $OUT/pmc
tail -3 $OUT/bench.log; head -12 $OUT/kernel_stats.csv | cut -c1-150
