"""K8 (csrc/egp_dynamics.hip): forward kinematics, joint-space inertia (MuJoCo legacy sparse qM) and bias force on the
GPU. MuJoCo is not available, so the oracle (oracle/dynamics.py) is first pinned to first principles on the CPU
(kinetic energy and Newton-Euler from finite differences of the forward kinematics, the zero-pose inertia the
surrogate backend already uses), then the kernel is compared with it."""
import numpy as np
import pytest
import torch

from oracle import dynamics as D
from oracle import humanoid as H


@pytest.fixture(scope="module")
def ctx(skel):
    from conftest import load_golden
    from egopose_amd.hip import EgpContext
    c = load_golden("config_subject_03.npz")
    cx = EgpContext(skel, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"])
    yield cx
    cx.close()


def _rand_state(skel, rng, n, joint_scale=0.5, vel_scale=1.0):
    q = np.zeros((n, skel.nq))
    q[:, :3] = rng.normal(size=(n, 3))
    quat = rng.normal(size=(n, 4))
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    q[:, 7:] = rng.normal(size=(n, skel.nq - 7)) * joint_scale
    return q, rng.normal(size=(n, skel.nv)) * vel_scale


def test_oracle_dynamics_first_principles(skel):
    rng = np.random.RandomState(3)
    q0 = np.zeros(skel.nq)
    q0[2], q0[3] = 1.0, 1.0
    np.testing.assert_allclose(D.inertia_matrix(skel, q0), skel.zero_pose_inertia(), atol=1e-12)
    q, v = _rand_state(skel, rng, 3)
    for i in range(3):
        np.testing.assert_allclose(D.fk(skel, q[i])[1], skel.body_xpos(q[i]), atol=1e-12)
        M = D.inertia_matrix(skel, q[i])
        assert np.linalg.eigvalsh(M).min() > 0 and np.abs(M - M.T).max() == 0
        ke = 0.5 * v[i] @ M @ v[i] - 0.5 * skel.armature * (v[i, 6:] @ v[i, 6:])
        assert ke == pytest.approx(D.kinetic_energy_fd(skel, q[i], v[i]), rel=1e-7)
        # tree sparsity: dofs on different branches do not couple
        rows, cols = skel.sparse_index()
        mask = np.zeros_like(M, bool)
        mask[rows, cols] = mask[cols, rows] = True
        assert np.abs(M[~mask]).max() < 1e-12
        # the spatial-vector formulation (the kernel's) against the Jacobian sum and the finite-difference Newton-Euler
        Ms, Cs, _ = D.crba_rne_spatial(skel, q[i], v[i])
        np.testing.assert_allclose(Ms, M, atol=1e-11)
        C = D.bias_force(skel, q[i], v[i])
        np.testing.assert_allclose(Cs, C, rtol=0, atol=2e-5 * max(1.0, np.abs(C).max()))
    # at rest the bias is pure gravity: root force = total weight, straight up against g
    q1, _ = _rand_state(skel, rng, 1)
    C0 = D.bias_force(skel, q1[0], np.zeros(skel.nv))
    np.testing.assert_allclose(C0[:3], [0, 0, 9.81 * skel.body_mass.sum()], atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 64, 131])
def test_dynamics_kernel_matches_oracle(ctx, skel, n):
    rng = np.random.RandomState(n)
    q, v = _rand_state(skel, rng, n)
    dev = torch.device("cuda")
    out = ctx.dynamics(torch.as_tensor(q, device=dev), torch.as_tensor(v, device=dev), want_xpos=True)
    qM, bias, xpos = (out[k].cpu().numpy() for k in ("qM", "bias", "xpos"))
    assert qM.shape == (n, skel.nM) and bias.shape == (n, skel.nv) and xpos.shape == (n, len(skel.body_names), 3)
    for i in range(min(n, 12)):
        np.testing.assert_allclose(xpos[i], skel.body_xpos(q[i]), rtol=0, atol=1e-12)
        Ms, Cs, _ = D.crba_rne_spatial(skel, q[i], v[i])
        np.testing.assert_allclose(skel.full_from_sparse(qM[i]), Ms, rtol=0, atol=1e-10)
        np.testing.assert_allclose(bias[i], Cs, rtol=0, atol=1e-9 * max(1.0, np.abs(Cs).max()))
    # independent formulations on a couple of envs (Jacobian sum; finite-difference Newton-Euler)
    for i in range(min(n, 2)):
        np.testing.assert_allclose(skel.full_from_sparse(qM[i]), D.inertia_matrix(skel, q[i]), rtol=0, atol=1e-10)
        C = D.bias_force(skel, q[i], v[i])
        np.testing.assert_allclose(bias[i], C, rtol=0, atol=2e-5 * max(1.0, np.abs(C).max()))


@pytest.mark.gpu
def test_dynamics_feeds_stable_pd(ctx, skel):
    """K8 -> K1 on the device: torques from GPU-computed (qM, qfrc_bias) == the oracle's stable PD on the oracle's M, C."""
    from conftest import load_golden
    c = load_golden("config_subject_03.npz")
    rng = np.random.RandomState(8)
    n = 9
    q, v = _rand_state(skel, rng, n, joint_scale=0.3, vel_scale=0.5)
    a = rng.normal(size=(n, skel.nu)) * 0.2
    dev = torch.device("cuda")
    qd, vd = torch.as_tensor(q, device=dev), torch.as_tensor(v, device=dev)
    dyn = ctx.dynamics(qd, vd)
    tq = ctx.pd_torque(qd, vd, torch.as_tensor(a, device=dev), dyn["qM"], dyn["bias"]).cpu().numpy()
    for i in range(n):
        M, C, _ = D.crba_rne_spatial(skel, q[i], v[i])
        _, ref = H.pd_torque(q[i], v[i], a[i], M, C, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], skel.timestep)
        np.testing.assert_allclose(tq[i], ref[0], rtol=1e-8, atol=1e-8)
    # strided qM output (the engine's 912-double rows) and partial outputs
    rows = torch.zeros(n, 912, dtype=torch.float64, device=dev)
    only = ctx.dynamics(qd, vd, want_bias=False, qM_out=rows)
    assert "bias" not in only and torch.equal(rows[:, :skel.nM], dyn["qM"]) and float(rows[:, skel.nM:].abs().max()) == 0.0
    assert ctx.dynamics(qd[:0], vd[:0])["qM"].shape == (0, skel.nM)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["resident", "per_substep"])
def test_engine_device_dynamics_matches_host_loop(ctx, skel, mode, monkeypatch):
    """Engine with device_dynamics: the backend is only asked for qpos / qvel (drain always gets qM == NULL), K8 feeds K1 -- inside
    the resident kernel (one launch per env-step), or as a K8 + K1 launch pair per substep -- with the REFERENCE's timing
    (ego_pose/envs/humanoid_v1.py:130-144: compute_torque reads data.qM / data.qfrc_bias as the previous mj_step left them, i.e.
    evaluated at the state that step started from; fresh only after the reset's sim.forward(), envs/common/mujoco_env.py:97-101).
    3 env-steps with a reset of some envs in between == the host loop with the oracle's stable PD fed the oracle's M, C of the
    PREVIOUS substep's state."""
    if mode == "per_substep":
        monkeypatch.setenv("EGP_SERVER", "0")
    from conftest import load_golden, VaryingInertiaBackend
    from egopose_amd.physics import RolloutEngine, SurrogatePhysics
    c = load_golden("config_subject_03.npz")
    g = load_golden("body_quat_obs.npz")
    n = 13
    rng = np.random.RandomState(2)
    qpos0, qvel0 = g["qpos"][:n], g["qvel"][:n] * 0.2
    be = VaryingInertiaBackend(skel, n)
    got_qM = []
    orig_drain = be._drain

    def spy_drain(env, qpos, qvel, qM, bias, xpos):
        got_qM.append(qM is not None)
        orig_drain(env, qpos, qvel, qM, bias, xpos)
        bias[:] = 1e9                                  # whatever the backend reports as bias must be ignored

    be._drain = spy_drain
    eng = RolloutEngine(ctx, be, n, n_threads=2, n_groups=1, device_dynamics=True)
    assert eng.substeps_per_launch == (15 if mode == "resident" else 1)
    eng.reset(np.arange(n), qpos0, qvel0)
    acts = [rng.normal(size=(n, 52)) * 0.2 for _ in range(3)]
    re_ids = np.array([0, 6, 9])                      # reset between env-steps 2 and 3: their M, C are fresh again, the others' stay stale
    re_q, re_v = g["qpos"][n:n + 3], g["qvel"][n:n + 3] * 0.1
    for k, a in enumerate(acts):
        if k == 2:
            eng.reset(re_ids, re_q, re_v)
        ad = torch.as_tensor(a, device="cuda")
        torch.cuda.synchronize()
        eng.step_async(0, ad)
        eng.wait(0)
        torch.cuda.synchronize()
    assert not be.physics.errors and not any(got_qM)
    got_q = eng.qpos.cpu().numpy()
    logged = [np.array(t) for t in be.torques]
    eng.close()
    ref = SurrogatePhysics(skel, 1)
    fresh_err = 0.0
    for e in [0, 3, 6, 9, 12]:
        ref.reset(0, qpos0[e], qvel0[e])
        M, C, _ = D.crba_rne_spatial(skel, qpos0[e], qvel0[e])             # sim.forward() of the reset
        row = 0
        for k, a in enumerate(acts):
            if k == 2 and e in re_ids:
                j = int(np.where(re_ids == e)[0][0])
                ref.reset(0, re_q[j], re_v[j])
                M, C, _ = D.crba_rne_spatial(skel, re_q[j], re_v[j])
            for s in range(15):
                q, v, _, _, _ = ref.drain(0, want_xpos=False)
                _, tc = H.pd_torque(q, v, a[e], M, C, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], skel.timestep)
                np.testing.assert_allclose(logged[e][row], tc[0], rtol=1e-9, atol=1e-9, err_msg="env %d substep %d" % (e, row))
                M_now, C_now, _ = D.crba_rne_spatial(skel, q, v)          # what this substep's mj_step leaves behind
                if row > 0:
                    _, tf = H.pd_torque(q, v, a[e], M_now, C_now, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], skel.timestep)
                    fresh_err = max(fresh_err, float(np.abs(tf[0] - tc[0]).max()))
                M, C = M_now, C_now
                ref.step(0, tc[0])
                row += 1
        q, *_ = ref.drain(0, want_xpos=False)
        np.testing.assert_allclose(got_q[e], q, rtol=1e-9, atol=1e-9)
    assert fresh_err > 1e-6, "the test cannot tell stale from fresh M, C (%g)" % fresh_err
    ref.close()
    be.close()


# ---------------------------------------------------------------------------------------------- MuJoCo-pinned (when the fixture exists)
def _mujoco_fixture():
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "mujoco_dynamics.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/mujoco_dynamics.npz is written by tools/gen_mujoco_golden.py on a machine that has MuJoCo "
                    "(not in this image): row f1 stays pinned to first principles only")
    return np.load(path)


def test_oracle_dynamics_matches_mujoco(skel):
    """mjData.qM / qfrc_bias / xpos (what humanoid_v1.py:98-111,130-144 reads) against the oracle's restatement."""
    g = _mujoco_fixture()
    assert tuple(g["dims"]) == (skel.nq, skel.nv, skel.nu, len(skel.body_names), skel.nM)
    for i in range(min(16, g["qpos"].shape[0])):
        Ms, Cs, _ = D.crba_rne_spatial(skel, g["qpos"][i], g["qvel"][i])
        np.testing.assert_allclose(Ms, skel.full_from_sparse(g["qM"][i]), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(Cs, g["qfrc_bias"][i], rtol=1e-8, atol=1e-7)
        np.testing.assert_allclose(skel.body_xpos(g["qpos"][i]), g["xpos"][i], rtol=0, atol=1e-10)


@pytest.mark.gpu
def test_dynamics_kernel_matches_mujoco(ctx, skel):
    """K8 on the device against MuJoCo's own numbers: the comparison that pins SURVEY row f1."""
    g = _mujoco_fixture()
    dev = torch.device("cuda")
    out = ctx.dynamics(torch.as_tensor(g["qpos"], device=dev), torch.as_tensor(g["qvel"], device=dev), want_xpos=True)
    np.testing.assert_allclose(out["qM"].cpu().numpy(), g["qM"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["bias"].cpu().numpy(), g["qfrc_bias"], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(out["xpos"].cpu().numpy(), g["xpos"], rtol=0, atol=1e-10)


def test_mujoco_plugin_replays_the_recorded_trajectory(skel):
    """The compiled backend (csrc/egp_physics_mujoco.cpp) stepping the fixture's controls reproduces MuJoCo's own trajectory."""
    import os
    from egopose_amd.physics import MujocoPhysics
    g = _mujoco_fixture()
    model = os.environ.get("EGP_MUJOCO_MODEL")
    if not MujocoPhysics.available() or not model:
        pytest.skip("needs the plugin (python -m egopose_amd.build_mujoco) and EGP_MUJOCO_MODEL=<the MJCF the fixture was made from>")
    ph = MujocoPhysics(skel, 2, model)
    ph.reset(1, g["qpos"][1], g["qvel"][1])
    q, v, qM, bias, xpos = ph.drain(1)
    np.testing.assert_allclose(qM, g["qM"][1], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(bias, g["qfrc_bias"][1], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(xpos, g["xpos"][1], rtol=0, atol=1e-12)
    for t in range(g["ctrl"].shape[0]):
        ph.step(1, g["ctrl"][t])
        q, v, _, _, _ = ph.drain(1, want_xpos=False)
        np.testing.assert_allclose(q, g["traj_qpos"][t], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(v, g["traj_qvel"][t], rtol=1e-12, atol=1e-12)
    ph.close()
