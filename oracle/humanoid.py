"""ORACLE (test infrastructure, not product code): per-env humanoid arithmetic.

Batch-first numpy restatement of the non-MuJoCo arithmetic of
``ego_pose/envs/humanoid_v1.py`` plus the finite-difference helpers of ``utils/math.py``.
Pinned against tests/golden/{body_quat_obs,pd_torque,reward}.npz. MuJoCo itself
(``mj_step``, ``mj_fullM`` inputs, ``body_xpos``) is an un-vendored dependency: the
dense-from-sparse expansion below follows MuJoCo's published legacy ``qM`` layout and is
anchored on the reference call site humanoid_v1.py:133-135 -- physics parity is unpinned.

Reference lines restated:
  get_body_quat          ego_pose/envs/humanoid_v1.py:113-125
  get_full_obs           ego_pose/envs/humanoid_v1.py:73-96   (obs_coord='heading', root_deheading, obs_vel='full')
  get_ee_pos             ego_pose/envs/humanoid_v1.py:98-111
  compute_desired_accel  ego_pose/envs/humanoid_v1.py:130-144
  compute_torque         ego_pose/envs/humanoid_v1.py:146-156
  do_simulation (target + clip)  ego_pose/envs/humanoid_v1.py:167-172
  get_qvel_fd            utils/math.py:20-35
  get_angvel_fd          utils/math.py:38-44
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve

from . import quat as Q


def body_quat(qpos, body_qpos_start, body_ndof):
    """(B,59) -> (B,84). 1-DoF bodies put their single angle in the first Euler slot."""
    qpos = np.atleast_2d(np.asarray(qpos, float))
    B, nb = qpos.shape[0], len(body_ndof)
    out = np.empty((B, nb, 4))
    out[:, 0] = qpos[:, 3:7]
    e = np.zeros((B, nb - 1, 3))
    for b in range(1, nb):
        s, n = int(body_qpos_start[b]), int(body_ndof[b])
        e[:, b - 1, :n] = qpos[:, s:s + n]
    out[:, 1:] = Q.q_from_euler_sxyz(e[..., 0], e[..., 1], e[..., 2])
    return out.reshape(B, nb * 4)


def full_obs(qpos, qvel, obs_heading=False, root_deheading=True, obs_coord="heading", obs_vel="full", phase=None):
    """HumanoidEnv.get_full_obs (ego_pose/envs/humanoid_v1.py:73-96), batch-first. Defaults (every shipped config):
    (B,59),(B,58) -> (B,115) = [qpos[2:] with de-headed root quat, qvel with heading-frame root lin-vel].
    `phase` = (cur_t (B,), env_episode_len) under cfg.obs_phase: one more column min(cur_t / env_episode_len, 1) (:92-94)."""
    qpos = np.atleast_2d(np.asarray(qpos, float)).copy()
    qvel = np.atleast_2d(np.asarray(qvel, float)).copy()
    root_q = qpos[:, 3:7].copy()
    qvel[:, :3] = Q.transform_vec(qvel[:, :3], root_q, obs_coord)                     # :78
    parts = []
    if obs_heading:                                                                    # :81-82
        parts.append(np.asarray(Q.heading(root_q), float).reshape(-1, 1))
    if root_deheading:                                                                 # :83-84
        qpos[:, 3:7] = Q.de_heading(root_q)
    parts.append(qpos[:, 2:])
    if obs_vel == "root":                                                              # :86-89
        parts.append(qvel[:, :6])
    elif obs_vel == "full":
        parts.append(qvel)
    if phase is not None:                                                              # :91-94
        cur_t, ep_len = phase
        parts.append(np.minimum(np.asarray(cur_t, float).reshape(-1, 1) / ep_len, 1.0))
    return np.concatenate(parts, axis=1)


def ee_pos(qpos, ee_wpos, transform="heading"):
    """World end-effector positions (B,5,3) -> root-relative in the heading frame, flat (B,15)."""
    qpos = np.atleast_2d(np.asarray(qpos, float))
    w = np.asarray(ee_wpos, float).reshape(qpos.shape[0], -1, 3)
    if transform is None:
        return w.reshape(qpos.shape[0], -1)
    rel = w - qpos[:, None, :3]
    out = Q.transform_vec(rel, qpos[:, None, 3:7], transform)
    return out.reshape(qpos.shape[0], -1)


_SPARSE_INDEX = {}


def _sparse_index(dof_parentid, dof_Madr):
    """(rows, cols) of every qM entry: dof i's chain i, parent(i), ... starts at dof_Madr[i]."""
    key = (np.asarray(dof_parentid).tobytes(), np.asarray(dof_Madr).tobytes())
    if key not in _SPARSE_INDEX:
        rows, cols = [], []
        for i in range(len(dof_parentid)):
            adr, j = int(dof_Madr[i]), i
            while j >= 0:
                assert adr == len(rows), "dof_Madr is not the cumulative chain length"
                rows.append(i)
                cols.append(j)
                adr += 1
                j = int(dof_parentid[j])
        _SPARSE_INDEX[key] = (np.array(rows), np.array(cols))
    return _SPARSE_INDEX[key]


def full_from_sparse(qM, dof_parentid, dof_Madr):
    """mj_fullM: MuJoCo legacy sparse inertia (B,nM) -> dense symmetric (B,nv,nv).

    (The reference calls MuJoCo's C routine; the index walk is done once and cached so this port's
    per-call cost is a scatter, not a Python loop.)"""
    qM = np.atleast_2d(np.asarray(qM, float))
    rows, cols = _sparse_index(dof_parentid, dof_Madr)
    nv = len(dof_parentid)
    M = np.zeros((qM.shape[0], nv, nv))
    M[:, rows, cols] = qM
    M[:, cols, rows] = qM
    return M


def pd_torque(qpos, qvel, action, M, C, jkp, jkd, a_ref, a_scale, torque_lim, dt):
    """Stable-PD torque for one substep; returns (torque, clipped torque), both (B,52).

    (M + Kd dt) qacc = -C - Kp e_q - Kd e_v  by Cholesky (SciPy/LAPACK as the reference),
    tau = -kp e_q[6:] - kd (qvel + qacc dt)[6:].
    """
    qpos = np.atleast_2d(np.asarray(qpos, float))
    qvel = np.atleast_2d(np.asarray(qvel, float))
    action = np.atleast_2d(np.asarray(action, float))
    M = np.asarray(M, float).reshape(qpos.shape[0], qvel.shape[1], qvel.shape[1])
    C = np.atleast_2d(np.asarray(C, float))
    B, nv = qvel.shape
    kp = np.concatenate([np.zeros(6), jkp])
    kd = np.concatenate([np.zeros(6), jkd])
    ctrl = a_ref + action * a_scale
    e_q = np.concatenate([np.zeros((B, 6)), qpos[:, 7:] - ctrl], axis=1)
    rhs = -C - kp * e_q - kd * qvel
    qacc = np.empty((B, nv))
    for b in range(B):
        qacc[b] = cho_solve(cho_factor(M[b] + np.diag(kd) * dt), rhs[b])
    e_v = qvel + qacc * dt
    tau = -jkp * e_q[:, 6:] - jkd * e_v[:, 6:]
    return tau, np.clip(tau, -torque_lim, torque_lim)


def control_torque(action_type, qpos, qvel, action, M, C, jkp, jkd, a_ref, a_scale, torque_lim, dt):
    """One substep of do_simulation's control law (ego_pose/envs/humanoid_v1.py:167-172); returns (torque, clipped torque).

    ctrl = a_ref + action * a_scale; 'position': torque = compute_torque(ctrl) (stable PD); 'torque': torque = ctrl;
    then clip to +-torque_lim. Any other action_type leaves `torque` unbound in the reference (UnboundLocalError)."""
    if action_type == "position":
        return pd_torque(qpos, qvel, action, M, C, jkp, jkd, a_ref, a_scale, torque_lim, dt)
    if action_type == "torque":
        tau = a_ref + np.atleast_2d(np.asarray(action, float)) * a_scale
        return tau, np.clip(tau, -torque_lim, torque_lim)
    raise UnboundLocalError("action_type %r: local variable 'torque' referenced before assignment" % (action_type,))


def qvel_fd(cur_qpos, next_qpos, dt, transform=None):
    """Finite-difference generalized velocity (B,58): world lin-vel (or heading frame), root-frame ang-vel."""
    a = np.atleast_2d(np.asarray(cur_qpos, float))
    b = np.atleast_2d(np.asarray(next_qpos, float))
    v = (b[:, :3] - a[:, :3]) / dt
    qrel = Q.qmul(b[:, 3:7], Q.qinv(a[:, 3:7]))
    axis, angle = Q.rot_axis_angle(qrel)
    angle = np.where(angle > np.pi, angle - 2 * np.pi, np.where(angle < -np.pi, angle + 2 * np.pi, angle))
    rv = Q.transform_vec(axis * angle[:, None] / dt, a[:, 3:7], "root")
    if transform is not None:
        v = Q.transform_vec(v, a[:, 3:7], transform)
    return np.concatenate([v, rv, (b[:, 7:] - a[:, 7:]) / dt], axis=1)


def angvel_fd(prev_bquat, cur_bquat, dt):
    """(B,84),(B,84) -> (B,63) body angular velocities rotation_from_quaternion(q_t q_{t-1}^-1)/dt."""
    p = np.atleast_2d(np.asarray(prev_bquat, float))
    c = np.atleast_2d(np.asarray(cur_bquat, float))
    d = Q.multi_quat_diff(c, p).reshape(p.shape[0], -1, 4)
    return (Q.rot_vec(d) / dt).reshape(p.shape[0], -1)
