# A/B of the state row's layout in one lease: (pad 0, skip 0) = 22 aligned lines per env-substep, (pad 1, skip 7) = 21
for rep in 1 2 3; do for cfg in "0 0" "1 7"; do set -- $cfg
EGP_ROW_PAD=$1 EGP_ROW_SKIP=$2 python bench.py --steps 10 --warmup 4 --no-legs --no-cpu-baseline --no-kernels --no-host-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('pad $1 skip $2', round(d['value']), c.get('t_sample_ms_median'), c.get('t_update_ms_median'), c.get('rollout_wait_s'), 'load', c.get('host_loadavg_1m'))"
done; done
