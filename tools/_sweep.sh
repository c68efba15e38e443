timeout 900 python -m pytest tests/test_rollout_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { # label, args...
  label=$1; shift
  env $ENVV timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['rollout_timing']; print('$label', round(d['value']), round(d['rollout_only_env_steps_per_s']), 'tsample', round(d['t_sample_s'],3), 'tupd', round(d['t_update_s'],3), 'policy', t['policy'], 'wait',t['wait'],'post', t['post'], 'reset', t['reset'])
"
}
ENVV="EGP_FAST_TICK=0" run slow; ENVV="EGP_FAST_TICK=1" run fast; ENVV="EGP_FAST_TICK=0" run slow; ENVV="EGP_FAST_TICK=1" run fast
