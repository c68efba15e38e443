# A/B of an environment switch in one lease: VAR=name VALS="0 1" REPS=3 bash tools/probes/ab_env.sh
for rep in $(seq 1 ${REPS:-3}); do for v in $VALS; do
env $VAR=$v python bench.py --steps 10 --warmup 4 --no-legs --no-cpu-baseline --no-kernels --no-host-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$VAR=$v', round(d['value']), c.get('t_sample_ms_median'), c.get('t_update_ms_median'), 'wait', c.get('rollout_wait_s'), 'load', c.get('host_loadavg_1m'))"
done; done
