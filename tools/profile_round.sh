#!/bin/bash
# Round profile set (run on the GPU box through gpurun): writes gpurun_out/profiles_round/
#   bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats over the default `python bench.py`
#   bench_line_under_rocprof.json   the JSON line that run printed
#   pmc_bench_fetch_write.csv   FETCH_SIZE / WRITE_SIZE per egp kernel (separate --pmc passes, kernel-trace only)
#   pmc_k1_traffic.json         HBM bytes per env-substep of the dominant kernel from those passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ROUND_TAG=${ROUND_TAG:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_round
rm -rf $OUT; mkdir -p $OUT
# (headline only: the extra legs / microbenchmarks of the default run launch the same kernels under other conditions
#  and would blur the per-kernel averages the bench line's roofline is checked against)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o b -- python bench.py --no-legs --no-kernels > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench_line_under_rocprof.json
cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/bench_kernel_stats.csv
rm -rf $OUT/raw
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o b -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-k1-events --no-legs --no-kernels > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
OUT = "$OUT"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(OUT + "/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != c or "egp::" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"][:64] + "|grid=" + r.get("Grid_Size", "?")
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    res[c] = acc
rows = ["kernel|grid,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg"]
k1 = None
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    f, nf = res["FETCH_SIZE"].get(k, [0, 0]); w, nw = res["WRITE_SIZE"].get(k, [0, 0])
    rows.append("%s,%d,%.2f,%.2f" % (k, max(nf, nw), f / max(nf, 1), w / max(nw, 1)))
    if "k_pd_server_tree58" in k and (k1 is None or max(nf, nw) > k1[1]):
        k1 = (k, max(nf, nw), f / max(nf, 1), w / max(nw, 1))
open(OUT + "/pmc_bench_fetch_write.csv", "w").write("\n".join(rows) + "\n")
if k1:
    grid = int(k1[0].split("grid=")[1])
    envs = grid // 256 * 4
    # env-steps of the profiled run (its own JSON line): every one of them is 15 stepped env-substeps of some launch
    steps = []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            line = [l for l in open(OUT + "/%s.log" % c) if l.startswith("{")][-1]
            steps.append(json.loads(line)["env_steps"])
        except Exception:
            pass
    env_steps = sum(steps) / len(steps) if steps else None
    d = {"kernel": "k_pd_server_tree58 (resident K1: one launch = 15 substeps of a %d-env group)" % envs,
         "source": "profiles/${ROUND_TAG}_pmc_bench_n1_fetch_write.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 0, %d launches)" % k1[1],
         "fetch_kib_per_launch": k1[2], "write_kib_per_launch": k1[3], "envs_per_launch": envs, "substeps_per_launch": 15,
         "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> bytes = (2*FETCH + WRITE) * 1024; the counters also see the state rows / torques in pinned host memory (zero-copy) and the go-word polls",
         "hbm_bytes_per_env_substep": (2 * k1[2] + k1[3]) * 1024 / (envs * 15),
         "launches": k1[1], "env_steps_of_profiled_run": env_steps, "commit": "${ROUND_COMMIT:-unknown}",
         "hbm_bytes_per_stepped_env_substep": (None if not env_steps else (2 * k1[2] + k1[3]) * 1024 * k1[1] / (env_steps * 15))}
    json.dump(d, open(OUT + "/pmc_k1_traffic.json", "w"), indent=1)
print("\n".join(rows))
PY
# matrix-core utilisation of the update's kernels (own PMC pass, kernel-trace only): the split-operand GEMM, the persistent
# LSTM recurrences and whatever library GEMM is left
timeout 900 rocprofv3 --pmc MfmaUtil VALUBusy --kernel-trace --output-format csv -d $OUT/mfma -o b -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-k1-events --no-legs --no-kernels > $OUT/mfma.log 2>&1
python - <<PY
import csv, glob, collections
OUT = "$OUT"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(OUT + "/mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if not ("k_gemm" in n or "k_lstm" in n or n.startswith("Cijk") or "k_policy" in n):
            continue
        a = acc[n[:110]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("MfmaUtil", [0, 0])[1])
with open(OUT + "/update_mfma_util.csv", "w") as f:
    f.write("kernel,launches,MfmaUtil_avg_pct,VALUBusy_avg_pct\n")
    for k, v in rows:
        m, b = v.get("MfmaUtil", [0, 1]), v.get("VALUBusy", [0, 1])
        f.write('"%s",%d,%.2f,%.2f\n' % (k, m[1], m[0] / max(1, m[1]), b[0] / max(1, b[1])))
print(open(OUT + "/update_mfma_util.csv").read()[:4000])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
head -12 $OUT/bench_kernel_stats.csv | cut -c1-160
cat $OUT/pmc_k1_traffic.json
