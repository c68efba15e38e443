#!/bin/bash
# phase stamps of the policy step with the filter's apply pass in its prologue (trace build on the box; run last in a call)
cd ${GRAFT_REPO_ROOT:-/root/repo}
EGP_BUILD_DEFS=-DEGP_POLICY_TRACE=3 timeout 600 python -m egopose_amd.build --force > /tmp/trace_build.log 2>&1 || tail -3 /tmp/trace_build.log
timeout 120 python tools/probes/policy_trace.py 2>&1 | grep -a "deltas\|tile" | tail -2
EGP_TRACE_FILTER=1 timeout 120 python tools/probes/policy_trace.py 2>&1 | grep -a "deltas\|tile\|Error\|state wave" | tail -4
