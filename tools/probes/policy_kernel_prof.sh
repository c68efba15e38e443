#!/bin/bash
# kernel durations of the policy step (engine idle / stepping) under rocprofv3, per tile: tools/probes/policy_kernel_prof.sh "8x4x2 ..."
cd /tmp && export TMPDIR=/tmp
for t in $1; do
  echo "=== EGP_POLICY_TILE=$t"
  rm -rf /tmp/pp_$t
  EGP_POLICY_TILE=$t timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$t -o p -- python $GRAFT_REPO_ROOT/tools/contention_probe.py > /tmp/pp_$t.log 2>&1
  grep -a "policy kernel" /tmp/pp_$t.log || tail -5 /tmp/pp_$t.log
  f=$(find /tmp/pp_$t -name "*kernel_stats.csv" | head -1)
  grep -a "k_policy\|k_pd_server" "$f" | cut -c1-200
  python - "$t" <<'P'
import csv, glob, sys
t = sys.argv[1]
fs = glob.glob('/tmp/pp_%s/**/*kernel_trace.csv' % t, recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if 'k_policy' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
h = len(d) // 2
for tag, x in (('idle', sorted(d[10:h])), ('stepping', sorted(d[h + 10:]))):
    print(tag, 'n', len(x), 'median %.1f us  p10 %.1f  p90 %.1f' % (x[len(x) // 2], x[len(x) // 10], x[len(x) * 9 // 10]))
P
done
