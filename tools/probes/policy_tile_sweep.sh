#!/bin/bash
# Policy-step tile sweep inside one gpurun call (library built with EGP_BUILD_DEFS=-DEGP_POLICY_TRACE=3 for the phase stamps):
# parity tests once per tile, phase stamps of one workgroup, the kernel's bracket time (engine idle / stepping), T_sample of the bench workload.
#   tools/probes/policy_tile_sweep.sh "8x4x2 4x4x4 ..."
for t in $1; do
  echo "=== EGP_POLICY_TILE=$t"
  EGP_POLICY_TILE=$t timeout 300 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "fused_policy or filter_apply_in_the_policy" 2>&1 | tail -1
  EGP_POLICY_TILE=$t timeout 120 python tools/probes/policy_trace.py 2>&1 | grep -a "deltas" | tail -1
  EGP_POLICY_TILE=$t timeout 120 python tools/contention_probe.py 2>&1 | grep -a "policy kernel"
  EGP_POLICY_TILE=$t timeout 200 python tools/probes/sample_time.py 9 2>&1 | grep -a "T_sample"
done
