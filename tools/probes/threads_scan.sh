# rollout rate against the number of host physics threads (16-CPU cgroup quota), interleaved so that box drift shows
for rep in 1 2; do for t in ${THREADS:-14 12 10 8}; do
python bench.py --steps 10 --warmup 4 --threads $t --no-legs --no-cpu-baseline --no-kernels --no-host-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('threads $t', round(d['value']), c.get('t_sample_ms_median'), c.get('t_update_ms_median'), 'load', c.get('host_loadavg_1m'), 'throttled', c.get('host_cgroup_throttled_events'), c.get('host_cgroup_throttled_ms_all_threads'))"
done; done
