for r in 1 2 3; do for v in 0 1; do echo "== round $r EGP_BAR_GO=$v"; EGP_BAR_GO=$v python bench.py --no-legs --no-cpu-baseline --no-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['per_iteration_ms_sample_update'], round(d['roofline']['avg_launch_us'],1))"; done; done
