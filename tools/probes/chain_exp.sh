#!/bin/bash
# timing ablations of the chained MLP launches, built on the GPU box: tools/probes/chain_exp.sh ["" "-DEGP_CHAIN_NO_MFMA" ...]
for v in "${@:-}"; do
  echo "=== variant [$v]"
  EGP_BUILD_DEFS="-DEGP_CHAIN_TRACE=3 $v" python -m egopose_amd.build --force > /dev/null 2>&1
  timeout 300 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | grep -a -E "passed|failed" | head -2
  timeout 300 python tools/probes/chain_trace.py 2>&1 | grep -a "ward\|rror" | tail -2
  timeout 300 python tools/probes/chain_time.py 2>&1 | grep -a "n_out 52\|rror"
done
