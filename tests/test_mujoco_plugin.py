"""The MuJoCo plugin (egopose_amd/csrc/egp_physics_mujoco.cpp), compiled against tests/mujoco_api -- a TEST DOUBLE of the few
MuJoCo entry points it calls, running this package's surrogate integrator behind MuJoCo's names -- and driven through the
physics boundary. What this covers: the plugin compiles, registers, and moves state correctly (set_state / step / drain, the
world-body offset of xpos, nM inertia entries, per-env mjData, inertia epochs, the diverged-step error, the model-table
cross-check). What it does NOT cover: MuJoCo's arithmetic -- nothing here pins physics (tests/golden/mujoco_dynamics.npz from
tools/gen_mujoco_golden.py on a MuJoCo-equipped machine does, and is still absent)."""
import ctypes as C
import os

import numpy as np
import pytest

import mujoco_api


@pytest.fixture(scope="module")
def plugin(tmp_path_factory, skel):
    root = str(tmp_path_factory.mktemp("fake_mujoco"))
    lib = mujoco_api.build(root)
    model = mujoco_api.write_fake_model(skel, os.path.join(root, "humanoid_fake.xml"))
    return lib, model


def _state(skel, rng, n):
    q = rng.normal(size=(n, skel.nq)) * 0.3
    q[:, 2] += 0.9
    q[:, 3:7] = rng.normal(size=(n, 4))
    q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    return q, rng.normal(size=(n, skel.nv))


def test_plugin_moves_state_like_the_backend_it_wraps(plugin, skel):
    """Three envs behind the plugin against three envs of the surrogate itself: identical drained state after set_state and after
    every step (the double IS the surrogate, so any difference is the plugin's plumbing: field mix-ups, the xpos offset, a
    shared mjData); the inertia epoch of an env moves with each of its steps and only its own."""
    from egopose_amd import _lib as L
    from egopose_amd.physics import MujocoPhysics, SurrogatePhysics
    lib, model = plugin
    mj = MujocoPhysics(skel, 3, model, plugin=lib)            # (runs the dof-tree / nM / timestep cross-check against the skeleton)
    assert mj.name == "mujoco-0"
    ref = SurrogatePhysics(skel, 3)
    rng = np.random.RandomState(12)
    q, v = _state(skel, rng, 3)
    for e in range(3):
        mj.reset(e, q[e], v[e]); ref.reset(e, q[e], v[e])
    for step in range(6):
        for e in (2, 0, 1):
            if step == 3 and e == 1:
                continue                                     # env 1 skips a step: per-env mjData, no lockstep assumption
            ctrl = rng.normal(size=skel.nu) * 30
            mj.step(e, ctrl); ref.step(e, ctrl)
        for e in range(3):
            a, b = mj.drain(e), ref.drain(e)
            for x, y, what in zip(a, b, ("qpos", "qvel", "qM", "qfrc_bias", "xpos")):
                np.testing.assert_array_equal(x, y, err_msg="%s of env %d after step %d" % (what, e, step))
            assert a[2].shape == (skel.nM,) and a[4].shape == (len(skel.body_names), 3)
    # a diverged step comes back as an error instead of being stepped on
    with pytest.raises(L.EgpError):
        mj.step(0, np.full(skel.nu, np.nan))
    mj.close(); ref.close()


def test_plugin_refuses_a_model_that_is_not_the_humanoid(plugin, skel, tmp_path):
    from egopose_amd import _lib as L
    from egopose_amd.physics import MujocoPhysics
    lib, model = plugin
    with pytest.raises(L.EgpError):
        MujocoPhysics(skel, 1, str(tmp_path / "missing.xml"), plugin=lib)


@pytest.mark.gpu
def test_engine_steps_through_the_plugin(plugin, skel):
    """One env-step of the rollout engine (resident K1 <-> host threads) with the plugin as the physics backend equals the same
    env-step on the surrogate backend bit for bit; the plugin reports a changing inertia on every step (MuJoCo's qM depends on
    qpos), so this is also the engine's host-fed inertia path."""
    import torch
    from conftest import load_golden
    from egopose_amd.hip import EgpContext
    from egopose_amd.physics import MujocoPhysics, RolloutEngine, SurrogatePhysics
    lib, model = plugin
    c = load_golden("config_subject_03.npz")
    N = 16
    rng = np.random.RandomState(5)
    q, v = _state(skel, rng, N)
    act = torch.as_tensor(rng.normal(size=(N, skel.nu)) * 0.2, device="cuda")
    out = []
    for kind in ("mujoco", "surrogate"):
        ctx = EgpContext(skel, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"])
        ph = MujocoPhysics(skel, N, model, plugin=lib) if kind == "mujoco" else SurrogatePhysics(skel, N)
        eng = RolloutEngine(ctx, ph, N, n_threads=2, n_groups=1)
        eng.reset(np.arange(N), q, v)
        for _ in range(2):
            eng.step_async(0, act)
            eng.wait(0)
        torch.cuda.synchronize()
        out.append((eng.qpos.cpu().numpy().copy(), eng.qvel.cpu().numpy().copy(), eng.ee_wpos.cpu().numpy().copy(), eng.head_z.copy()))
        eng.close(); ph.close(); ctx.close()
    for a, b, what in zip(out[0], out[1], ("qpos", "qvel", "ee_wpos", "head_z")):
        np.testing.assert_array_equal(a, b, err_msg=what)
