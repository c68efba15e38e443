for r in 1 2 3; do for v in "" "--no-k1-events"; do echo "== round $r events: ${v:-on}"; python bench.py --no-legs --no-cpu-baseline --no-kernels --steps 5 $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), [x[0] for x in d['per_iteration_ms_sample_update']])"; done; done
