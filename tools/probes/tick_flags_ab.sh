#!/bin/bash
# Where the tick's flag slab lives (EGP_TICK_FLAGS = kernel | bar | upload), A/B inside one gpurun call:
#   1. the rollout parity tests under 'bar', 2. T_sample of the bench workload, alternating, 3. the tick kernels' durations in a
#   rollout under rocprofv3 (kernel trace only) for each mode.    tools/probes/tick_flags_ab.sh "kernel bar"
MODES=${1:-"kernel bar"}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "=== parity under EGP_TICK_FLAGS=bar"
EGP_TICK_FLAGS=bar timeout 900 python -m pytest tests/test_rollout_gpu.py -q -x -m gpu 2>&1 | grep -a -E "passed|failed|^E  |^FAILED" | tail -5
for r in 1 2 3; do
  for m in $MODES; do
    echo "== round $r EGP_TICK_FLAGS=$m"
    EGP_TICK_FLAGS=$m timeout 300 python tools/probes/sample_time.py 9 2>&1 | grep -a "T_sample" | tail -1
  done
done
cd /tmp && export TMPDIR=/tmp
for m in $MODES; do
  echo "=== kernel durations, EGP_TICK_FLAGS=$m"
  rm -rf /tmp/tf_$m
  EGP_TICK_FLAGS=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tf_$m -o p -- python $R/tools/probes/sample_time.py 5 > /tmp/tf_$m.log 2>&1
  grep -a "T_sample" /tmp/tf_$m.log | tail -1
  python - "$m" <<'P'
import csv, glob, sys
fs = glob.glob('/tmp/tf_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)
for r in csv.DictReader(open(fs[0])):
    n = r['Name']
    if any(k in n for k in ('k_policy', 'k_zf_partial', 'k_reward_quat', 'k_zf_apply', 'k_pd_server')):
        print('  %-62.62s calls %6s  avg %8.2f us' % (n, r['Calls'], float(r['AverageNs']) / 1e3))
P
done
# phase stamps of the policy step's prologue, plain and with the filter's apply pass (trace build: last, the box is discarded after)
cd $R
EGP_BUILD_DEFS=-DEGP_POLICY_TRACE=3 timeout 600 python -m egopose_amd.build --force > /tmp/trace_build.log 2>&1 || tail -3 /tmp/trace_build.log
timeout 120 python tools/probes/policy_trace.py 2>&1 | grep -a "deltas\|tile" | tail -2
EGP_TRACE_FILTER=1 timeout 120 python tools/probes/policy_trace.py 2>&1 | grep -a "deltas\|tile\|Error" | tail -3
