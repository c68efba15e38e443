#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun): everything under gpurun_out/profiles_round6/, to be copied to profiles/r06_*.
# ROUND_COMMIT=<git rev-parse --short HEAD of the tree that was sent> is stamped into every file's header / meta.
#   r06_meta.json                                                                     commit, date, what ran
#   r06_bench_n1_kernel_stats.csv / _kernel_classes.txt / _line_under_rocprof.json   rocprofv3 --kernel-trace --stats over bench.py (headline only)
#   r06_pmc_bench_n1_fetch_write.csv, pmc_k1_traffic.json                            FETCH_SIZE / WRITE_SIZE passes over bench.py --steps 1
#   r06_update_mfma_util.csv                                                          MfmaUtil / VALUBusy of the update's kernels and the policy step
#   r06_microbench.jsonl                                                              K1-K6 / K8 at 1 024 .. 1 048 576 envs, launches rotating over input sets (HIP events)
#   r06_pmc_microbench_{65536,1024}_fetch_write.csv                                   FETCH_SIZE / WRITE_SIZE of the same kernels
#   r06_pmc_{k1_grid58,k2_reward,k5_gae,k6_zfilter,k8_dynamics}_{65536,1048576}.txt      VALUBusy / occupancy / SALUBusy of K1, K2, K5, K6, K8
#   r06_statereg_mfma_util.csv, r06_statereg_kernel_stats.csv                         config 4 (256 x 224 x 224, bf16 encoder)
#   r06_phase_profile.txt                                                             one rollout and one update separately (torch profiler)
#   r06_update_epoch_trace.txt                                                        one epoch of the update call by call (HIP events; shapes, TB/s, TFLOP/s)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_round6
C=${ROUND_COMMIT:-unknown}
stamp() { { echo "# commit $C (tools/profile_round6.sh)"; cat "$1"; } > "$2"; }
if [ "${PART:-all}" = "2" ]; then      # PART=2: only the kernel PMC passes, config 4 and the phase profile (the bench passes ran in another call)
  mkdir -p $OUT
else
rm -rf $OUT; mkdir -p $OUT
echo "{\"commit\": \"$C\", \"date\": \"$(date -u +%FT%TZ)\", \"script\": \"tools/profile_round6.sh\"}" > $OUT/r06_meta.json
ROUND_TAG=r06 ROUND_COMMIT=$C bash tools/profile_round.sh > $OUT/profile_round.log 2>&1
P=$GRAFT_REPO_ROOT/gpurun_out/profiles_round
stamp $P/bench_kernel_stats.csv $OUT/r06_bench_n1_kernel_stats.csv
cp $P/bench_line_under_rocprof.json $OUT/r06_bench_n1_line_under_rocprof.json
stamp $P/pmc_bench_fetch_write.csv $OUT/r06_pmc_bench_n1_fetch_write.csv
cp $P/pmc_k1_traffic.json $OUT/pmc_k1_traffic.json
stamp $P/update_mfma_util.csv $OUT/r06_update_mfma_util.csv
{ echo "# commit $C"; python tools/classify_kernel_stats.py $P/bench_kernel_stats.csv 5; } > $OUT/r06_bench_n1_kernel_classes.txt 2>&1
if [ "${QUICK:-0}" = "1" ]; then ls -la $OUT; exit 0; fi      # QUICK=1: only the bench passes above (kernel stats, FETCH / WRITE, MfmaUtil)
python tools/microbench.py 1024 8192 65536 1048576 > $OUT/r06_microbench.jsonl 2> $OUT/microbench.err
fi
for n in 1048576 65536 1024; do
  [ "${PART:-all}" = "2" ] && break
  { echo "# commit $C"; bash tools/pmc_k1.sh $n; } > $OUT/r06_pmc_microbench_${n}_fetch_write.csv 2> $OUT/pmc_k1_$n.err
done
# (round 6: the kernel passes at 1 048 576 envs too -- the size at which K2-K8 no longer fit the 256 MiB Infinity Cache)
for n in 65536 1048576; do
for kv in k1_grid58:k_pd_torque_grid58:K1_pd_torque k2_reward:k_reward_quat_v3:K2_reward k5_gae:k_gae:K5_gae k6_zfilter:k_zf:K6_zfilter k8_dynamics:k_dynamics:K8_dynamics; do
  tag=${kv%%:*}; rest=${kv#*:}; kern=${rest%%:*}; only=${rest##*:}
  grep -q VALUBusy $OUT/r06_pmc_${tag}_$n.txt 2>/dev/null && continue        # (PART=2 after a call that ran out of time: keep what is there)
  { echo "# commit $C  kernel $kern at $n envs (tools/pmc_kernel.sh, microbench case $only alone, inputs rotating beyond the Infinity Cache)"; ONLY=$only bash tools/pmc_kernel.sh $kern $n VALUBusy MeanOccupancyPerCU SALUBusy; } > $OUT/r06_pmc_${tag}_$n.txt 2>&1
done
done
bash tools/prof_statereg.sh > $OUT/prof_statereg.log 2>&1
stamp $GRAFT_REPO_ROOT/gpurun_out/prof_statereg/mfma_util.csv $OUT/r06_statereg_mfma_util.csv
stamp $GRAFT_REPO_ROOT/gpurun_out/prof_statereg/kernel_stats.csv $OUT/r06_statereg_kernel_stats.csv
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_statereg/bench.log > $OUT/r06_statereg_bench_tail.txt
python tools/phase_profile.py --out $OUT/r06_phase_profile.txt > $OUT/phase_profile.log 2>&1
sed -i "1i # commit $C" $OUT/r06_phase_profile.txt
# one epoch of the update call by call (HIP events around every product and sweep: shapes, TB/s, float32-equivalent TFLOP/s)
{ echo "# commit $C (tools/epoch_trace.py --epoch 5: backward of epoch 5, then the forward of epoch 6; bytes = operands + result once, TF/s = 2MNK)"; python tools/epoch_trace.py --epoch 5; } > $OUT/r06_update_epoch_trace.txt 2> $OUT/epoch_trace.err
ls -la $OUT
