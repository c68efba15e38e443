#!/bin/bash
# Policy-step tile sweep inside one gpurun call: parity tests once per tile, the kernel's bracket time (engine idle / stepping),
# and T_sample of the bench workload.  tools/probes/policy_tile_sweep.sh "8x4x4 8x4x2 ..."
for t in $1; do
  echo "=== EGP_POLICY_TILE=$t"
  EGP_POLICY_TILE=$t timeout 300 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "fused_policy or filter_apply_in_the_policy" 2>&1 | tail -2
  EGP_POLICY_TILE=$t timeout 120 python tools/contention_probe.py 2>&1 | grep -a "policy kernel"
  EGP_POLICY_TILE=$t timeout 200 python tools/probes/sample_time.py 9 2>&1 | grep -a "T_sample"
done
