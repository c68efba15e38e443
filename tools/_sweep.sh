
run() { label=$1; shift
  env $ENVV timeout -s USR1 -k 5 120 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/err_$label.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['rollout_timing']; print('$label', round(d['value']), 'tsample', round(d['t_sample_s'],3), 'tupdate', round(d['t_update_s'],3), 'py', t['policy'], t['wait'], t['post'], t['reset'])
"
  rc=${PIPESTATUS[0]}; if [ $rc != 0 ]; then echo "$label rc=$rc"; grep -v amdgpu.ids gpurun_out/err_$label.log | tail -30; fi
}
for i in 1 2; do ENVV="EGP_X=1" run base$i; ENVV="EGP_ZF_MERGE_IN_APPLY=1" run zfm$i; ENVV="EGP_TICK_FLAGS=zerocopy" run zc$i; ENVV="EGP_ZF_MERGE_IN_APPLY=1 EGP_TICK_FLAGS=zerocopy" run both$i; done
