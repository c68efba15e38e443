#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun): everything under gpurun_out/profiles_round3/, to be copied to profiles/r03_*
#   r03_bench_n1_kernel_stats.csv / _kernel_classes.txt / _line_under_rocprof.json   rocprofv3 --kernel-trace --stats over bench.py (headline only)
#   r03_pmc_bench_n1_fetch_write.csv, pmc_k1_traffic.json                            FETCH_SIZE / WRITE_SIZE passes over bench.py --steps 1
#   r03_update_mfma_util.csv                                                          MfmaUtil / VALUBusy of the update's kernels
#   r03_microbench.jsonl                                                              K1-K6 / K8 with HBM-resident inputs (HIP events)
#   r03_pmc_microbench_{65536,1024}_fetch_write.csv                                   FETCH_SIZE / WRITE_SIZE of the same kernels (K1 = the lane-grid kernel)
#   r03_pmc_k1_grid58_65536.txt                                                       VALUBusy / occupancy counters of k_pd_torque_grid58
#   r03_phase_profile.txt                                                             one rollout and one update separately (torch profiler)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_round3
rm -rf $OUT; mkdir -p $OUT
ROUND_TAG=r03 bash tools/profile_round.sh > $OUT/profile_round.log 2>&1
P=$GRAFT_REPO_ROOT/gpurun_out/profiles_round
cp $P/bench_kernel_stats.csv $OUT/r03_bench_n1_kernel_stats.csv
cp $P/bench_line_under_rocprof.json $OUT/r03_bench_n1_line_under_rocprof.json
cp $P/pmc_bench_fetch_write.csv $OUT/r03_pmc_bench_n1_fetch_write.csv
cp $P/pmc_k1_traffic.json $OUT/pmc_k1_traffic.json
cp $P/update_mfma_util.csv $OUT/r03_update_mfma_util.csv
python tools/classify_kernel_stats.py $OUT/r03_bench_n1_kernel_stats.csv 5 > $OUT/r03_bench_n1_kernel_classes.txt 2>&1
python tools/microbench.py 1024 8192 65536 > $OUT/r03_microbench.jsonl 2> $OUT/microbench.err
for n in 65536 1024; do
  bash tools/pmc_k1.sh $n > $OUT/r03_pmc_microbench_${n}_fetch_write.csv 2> $OUT/pmc_k1_$n.err
done
bash tools/pmc_kernel.sh k_pd_torque_grid58 65536 VALUBusy MeanOccupancyPerCU SALUBusy FetchSize WriteSize > $OUT/r03_pmc_k1_grid58_65536.txt 2>&1
python tools/phase_profile.py --out $OUT/r03_phase_profile.txt > $OUT/phase_profile.log 2>&1
ls -la $OUT
