#!/usr/bin/env python3
"""tests/golden/mujoco_dynamics.npz -- MuJoCo's own numbers for the quantities K8 / the physics boundary stand in for. Run on a
MACHINE THAT HAS MUJOCO (this build image does not): the fixture it writes is what turns SURVEY row f1 ("parity unpinned") into
a pinned row -- tests/test_dynamics.py consumes it when present and skips when not.

    python tools/gen_mujoco_golden.py --model /path/to/assets/mujoco_models/humanoid_1205_v1.xml [--n 64]

Bindings: the `mujoco` package (>= 2.1.2; the reference's MJCF uses coordinate="global", so convert it first with an older
release's `compile` tool) or `mujoco_py` (the reference's own, README.md:20-21). Records, for seeded random states:
    qpos (n, nq), qvel (n, nv)          inputs
    qM (n, nM)                           mjData.qM after mj_forward (legacy sparse; mj_fullM's input, humanoid_v1.py:133-135)
    qfrc_bias (n, nv)                    mjData.qfrc_bias (humanoid_v1.py:136)
    xpos (n, nbody - 1, 3)               mjData.xpos[1:] (humanoid_v1.py:98-111, get_body_com)
and a short controlled trajectory (ctrl (T, nu) -> qpos / qvel after each mj_step) for the plugin backend."""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden", "mujoco_dynamics.npz")


class _New:
    def __init__(self, path):
        import mujoco
        self.mj = mujoco
        self.m = mujoco.MjModel.from_xml_path(path)
        self.d = mujoco.MjData(self.m)
        self.version = "mujoco-%s" % mujoco.__version__

    def dims(self):
        m = self.m
        return m.nq, m.nv, m.nu, m.nbody, m.nM

    def forward(self, q, v):
        self.mj.mj_resetData(self.m, self.d)
        self.d.qpos[:], self.d.qvel[:] = q, v
        self.mj.mj_forward(self.m, self.d)
        return self.d.qM.copy(), self.d.qfrc_bias.copy(), self.d.xpos[1:].copy()

    def step(self, ctrl):
        self.d.ctrl[:] = ctrl
        self.mj.mj_step(self.m, self.d)
        return self.d.qpos.copy(), self.d.qvel.copy()


class _Old:
    def __init__(self, path):
        import mujoco_py
        self.m = mujoco_py.load_model_from_path(path)
        self.sim = mujoco_py.MjSim(self.m)
        self.version = "mujoco_py-%s" % getattr(mujoco_py, "__version__", "?")

    def dims(self):
        m = self.m
        return m.nq, m.nv, m.nu, m.nbody, m.nM

    def forward(self, q, v):
        self.sim.reset()
        st = self.sim.get_state()
        st.qpos[:], st.qvel[:] = q, v
        self.sim.set_state(st)
        self.sim.forward()
        d = self.sim.data
        return d.qM.copy(), d.qfrc_bias.copy(), d.body_xpos[1:].copy()

    def step(self, ctrl):
        self.sim.data.ctrl[:] = ctrl
        self.sim.step()
        return self.sim.data.qpos.copy(), self.sim.data.qvel.copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    try:
        B = _New(args.model)
    except ImportError:
        B = _Old(args.model)
    nq, nv, nu, nbody, nM = B.dims()
    rng = np.random.RandomState(1205)
    qpos = np.zeros((args.n, nq))
    qpos[:, :3] = rng.normal(size=(args.n, 3)) * [1.0, 1.0, 0.05] + [0, 0, 0.9]
    quat = rng.normal(size=(args.n, 4))
    qpos[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    qpos[:, 7:] = rng.normal(size=(args.n, nq - 7)) * 0.4
    qpos[0, 3:7], qpos[0, 7:] = [1, 0, 0, 0], 0.0                   # the zero pose
    qvel = rng.normal(size=(args.n, nv))
    qvel[0] = 0.0
    qM, bias, xpos = zip(*(B.forward(qpos[i], qvel[i]) for i in range(args.n)))
    # a controlled trajectory from state 1
    B.forward(qpos[1], qvel[1])
    ctrl = rng.normal(size=(args.steps, nu)) * 20.0
    traj = [B.step(c) for c in ctrl]
    np.savez_compressed(OUT, qpos=qpos, qvel=qvel, qM=np.stack(qM), qfrc_bias=np.stack(bias), xpos=np.stack(xpos), ctrl=ctrl,
                        traj_qpos=np.stack([t[0] for t in traj]), traj_qvel=np.stack([t[1] for t in traj]),
                        dims=np.array([nq, nv, nu, nbody - 1, nM]), version=np.array(B.version), model=np.array(os.path.basename(args.model)))
    print("wrote", OUT, B.version)


if __name__ == "__main__":
    main()
