timeout 900 python -m pytest tests/test_rollout_gpu.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -2
run() { # label, args...
  label=$1; shift
  env $ENVV timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['rollout_timing']; print('$label', round(d['value']), round(d['rollout_only_env_steps_per_s']), 'tsample', round(d['t_sample_s'],3), 'wait',t['wait'])
"
}
for i in 1 2 3; do ENVV="EGP_ZF_MERGE_IN_APPLY=1" run merged; ENVV="EGP_ZF_MERGE_IN_APPLY=0" run separate; done
