#!/bin/bash
# sample the GPU's shader clock while a command runs: clock_watch.sh <command...>
"$@" > /tmp/cw_cmd.log 2>&1 &
PID=$!
sleep ${CW_DELAY:-25}
for i in $(seq 1 40); do
  for f in /sys/class/drm/card*/device/pp_dpm_sclk; do grep '\*' $f | tr '\n' ' '; done
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1
  sleep 0.05
done | sort | uniq -c | sort -rn | head -8
wait $PID
grep "T_sample\|T_update" /tmp/cw_cmd.log
