cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
for c in 0 2 3 5; do
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | tr ' ' '_')
    ONLY=$c timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/c${c}_$tag -o mb -- python tools/probes/gemm_bound_probe.py 3 > $OUT/c${c}_$tag.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    case = re.search(r"/c(\d)_", path).group(1)
    for r in csv.DictReader(open(path)):
        if "k_gemm_ws" in r["Kernel_Name"]:
            a = acc[(case, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("case %s %-16s launches %3d  avg %.6g" % (k[0], k[1], n, s / max(n, 1)))
PY
rm -rf $OUT/c*/
