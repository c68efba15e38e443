run() { label=$1; shift
  env $ENVV timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label', round(d['value']), 'tsample', round(d['t_sample_s'],3), 'tupdate', round(d['t_update_s'],3))
"
}
for i in 1 2; do ENVV="EGP_TUNED_GEMMS=0" run fc_default --task egoforecast; ENVV="EGP_X=1" run fc_tuned --task egoforecast; done
ENVV="EGP_X=1" run mimic_tuned
