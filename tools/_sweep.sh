run() { label=$1; shift
  env $ENVV timeout -s USR1 -k 5 120 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/err_$label.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label', round(d['value']), 'tsample', round(d['t_sample_s'],3), 'tupdate', round(d['t_update_s'],3))
"
  rc=${PIPESTATUS[0]}; if [ $rc != 0 ]; then echo "$label rc=$rc"; grep -v amdgpu.ids gpurun_out/err_$label.log | tail -30; fi
}
for i in 1 2 3; do ENVV="EGP_LSTM_GROUP=0" run sep$i; ENVV="EGP_LSTM_GROUP=1" run grp$i; done
ENVV="EGP_X=1" run forecast --task egoforecast
