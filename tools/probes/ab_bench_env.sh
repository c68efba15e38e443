#!/bin/bash
# A/B an environment switch under bench.py itself (headline only, 5 steps): ab_bench_env.sh VAR=a VAR=b
for r in 1 2 3; do for v in "$1" "$2"; do echo "== round $r $v"; env $v python bench.py --no-legs --no-cpu-baseline --no-kernels --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), [x[0] for x in d['per_iteration_ms_sample_update']], [x[1] for x in d['per_iteration_ms_sample_update']])"; done; done
