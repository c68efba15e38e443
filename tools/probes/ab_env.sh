#!/bin/bash
# A/B an environment switch inside one gpurun call: ab_env.sh VAR=a VAR=b -- command...   (three rounds, alternating)
A=$1; B=$2; shift 3
for r in 1 2 3 4; do
  for v in "$A" "$B"; do echo "== round $r $v"; env $v "$@" 2>&1 | grep -a "T_sample\|T_update" | tail -1; done
done
