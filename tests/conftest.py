import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a ROCm device the `gpu` tests are skipped (not errors), so a plain `pytest tests` on a CPU box shows
    real regressions only. The GPU box must not take this branch: there a missing device is a failure of the run."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no ROCm device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def skel():
    from egopose_amd.skeleton import load_skeleton
    return load_skeleton()


class VaryingInertiaBackend:
    """Test double behind egp_physics_register: the surrogate's dynamics, but the inertia it reports changes on
    every step (qM = qM0 * (1 + 0.03 * ((step + env) % 4))) and so does its epoch -- the shape of a MuJoCo adapter,
    whose M depends on qpos. `torques` logs every ctrl row the engine handed to step()."""

    def __init__(self, skel, n_env, fail_at=None):
        from egopose_amd.physics import SurrogatePhysics, CallbackPhysics
        self.fail_at = fail_at                      # (env, step count): step() raises there (failure-path tests)
        self.inner = SurrogatePhysics(skel, n_env)
        self.k = np.zeros(n_env, np.int64)
        self.resets = np.zeros(n_env, np.int64)
        self.torques = [[] for _ in range(n_env)]
        self.physics = CallbackPhysics(skel, n_env, self._reset, self._step, self._drain, self._epoch, name="varying-inertia")
        self.handle = self.physics.handle
        self.skel, self.n_env = skel, n_env

    def scale(self, env, k):
        return 1.0 + 0.03 * ((int(k) + int(env)) % 4)

    def _reset(self, env, qpos, qvel):
        self.inner.reset(env, qpos.copy(), qvel.copy())
        self.k[env] = 0
        self.resets[env] += 1

    def _step(self, env, ctrl):
        if self.fail_at is not None and (env, int(self.k[env])) == tuple(self.fail_at):
            raise RuntimeError("contact solver diverged (injected)")
        self.torques[env].append(ctrl.copy())
        self.inner.step(env, ctrl.copy())
        self.k[env] += 1

    def _drain(self, env, qpos, qvel, qM, bias, xpos):
        q, v, m, b, x = self.inner.drain(env, want_xpos=xpos is not None)
        qpos[:], qvel[:], bias[:] = q, v, b
        if qM is not None:
            qM[:] = m * self.scale(env, self.k[env])
        if xpos is not None:
            xpos[:] = x

    def _epoch(self, env):
        return int(self.resets[env]) * 100000 + int(self.k[env]) + 1

    def close(self):
        self.physics.close()
        self.inner.close()
