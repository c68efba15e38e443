#!/bin/bash
# the per-tick GPU chain's links and gaps (tools/probes/chain_gaps.py) from a kernel trace of three rollouts
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/cg; "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/cg -o t -- python $R/tools/probes/sample_time.py 3 > /tmp/cg.log 2>&1
grep -a T_sample /tmp/cg.log | tail -1
python $R/tools/probes/chain_gaps.py /tmp/cg | tail -3; python $R/tools/probes/chain_links.py /tmp/cg
