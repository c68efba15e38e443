run() { label=$1; shift
  env $ENVV timeout -s USR1 -k 5 90 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/err_$label.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label', round(d['value']), 'tsample', round(d['t_sample_s'],3), 'tupdate', round(d['t_update_s'],3))
"
  echo "$label rc=${PIPESTATUS[0]}"; grep -v amdgpu.ids gpurun_out/err_$label.log | tail -60
}
for i in 1 2 3 4 5 6 7 8; do ENVV="EGP_X=1" run ov$i; done
