run() { label=$1; shift
  env $ENVV timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label', round(d['value']), 'tsample', round(d['t_sample_s'],3), 'tupdate', round(d['t_update_s'],3))
"
}
for i in 1 2; do ENVV="EGP_LSTM_MFMA=0" run fma; ENVV="EGP_LSTM_MFMA=1" run mfma; ENVV="EGP_UPDATE_OVERLAP=1" run mfma_overlap; done
ENVV="EGP_X=1" run forecast --task egoforecast
