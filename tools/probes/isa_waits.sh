#!/bin/bash
# Where a kernel waits for memory in its first stretch: the s_waitcnt vmcnt(..) of the ISA up to the first s_barrier, each with
# what was issued since the previous one (vector loads, scalar loads, stores, float64 ops, branches). A wait right behind a
# handful of loads, in a run of branches, is a dependent round trip.   tools/probes/isa_waits.sh [source.hip] [mangled-name regex]
SRC=${1:-egopose_amd/csrc/egp_policy.hip}
PAT=${2:-'^_ZN12_GLOBAL__N_120k_policy_gaussian_w4ILi4ELi2ELb1E'}
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -S --cuda-device-only $SRC -o /tmp/isa_waits.s 2>/dev/null
A=$(grep -n "$PAT" /tmp/isa_waits.s | grep ':$\|: *;' | head -1 | cut -d: -f1)
[ -z "$A" ] && { echo "no function matches $PAT"; exit 1; }
sed -n "$A,\$p" /tmp/isa_waits.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/isa_waits_f.s
B=$(grep -n "s_barrier" /tmp/isa_waits_f.s | head -1 | cut -d: -f1)
echo "first barrier at line ${B:-none} of $(wc -l < /tmp/isa_waits_f.s); scratch accesses: $(grep -c scratch_ /tmp/isa_waits_f.s)"
head -${B:-100000} /tmp/isa_waits_f.s | python3 -c "
import sys, re
loads = sl = st = f64 = br = 0
for i, l in enumerate(sys.stdin):
    l = l.strip()
    if l.startswith(('global_load', 'buffer_load')): loads += 1
    elif l.startswith(('s_load', 's_buffer_load')): sl += 1
    elif l.startswith('global_store'): st += 1
    elif re.match(r'v_\w+_f64', l): f64 += 1
    elif l.startswith('s_cbranch'): br += 1
    elif (l.startswith('s_waitcnt') and 'vmcnt' in l) or l.startswith('s_barrier'):
        print('%5d  [+%d vload %d sload %d store %d f64 %d br] %s' % (i, loads, sl, st, f64, br, l[:60])); loads = sl = st = f64 = br = 0
"
