"""ORACLE -- test infrastructure only.

A CPU (numpy / torch-CPU float64) restatement of the reference algorithm for the EgoPose PPO
rollout+update hot path, each function citing the reference file:line it follows. It is
pinned to golden vectors produced by importing the reference itself (tools/gen_golden.py ->
tests/golden/*.npz). Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product (``egopose_amd``) never does and fails
loudly when its HIP library is missing.

Physics (MuJoCo ``mj_step``) is an un-vendored, unpinned dependency of the reference: parity
for everything downstream of it is defined on *drained state* (see DESIGN.md), and the
physics itself is marked "parity unpinned".
"""
