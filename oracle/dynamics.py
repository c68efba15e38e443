"""ORACLE (test infrastructure): rigid-body dynamics terms of the humanoid tree -- forward kinematics with body
rotations, the joint-space inertia M(q) and the bias force C(q, qvel) (Coriolis + centrifugal + gravity), i.e. what
MuJoCo exposes as data.xpos / data.qM / data.qfrc_bias and what the reference's stable PD consumes
(/root/reference/ego_pose/envs/humanoid_v1.py:130-144: `mj_fullM(model, M, data.qM)`, `data.qfrc_bias`).

**Parity unpinned against MuJoCo**: MuJoCo (mujoco-py, un-vendored and unpinned in the reference) is not in this image and the
reference holds no fixtures of qM / qfrc_bias. The oracle is therefore written in a formulation that shares nothing
with the HIP kernel (egp_dynamics: composite-rigid-body + recursive Newton-Euler in spatial vectors) and is itself
pinned to first principles by the tests:
  * M: sum over bodies of m JvT Jv + JwT I Jw with geometric Jacobians; checked against the kinetic energy obtained
    from finite differences of the forward kinematics alone;
  * C: Newton-Euler on body accelerations obtained by finite differences of the forward kinematics along the
    configuration path q(t) = q (+) qvel t (MuJoCo's mj_integratePos convention: root angular velocity in the
    body frame), projected with the same Jacobians.
Conventions (MuJoCo): qpos = [root pos(3), root quat wxyz(4), hinges]; qvel = [root linear velocity (world),
root angular velocity (BODY frame), hinge rates]; gravity (0, 0, -9.81); armature added to the hinge diagonal of M.
MJCF `coordinate="global"`: at the zero pose all body frames are axis aligned (skeleton.py: Skeleton.body_xpos).
"""
import numpy as np

GRAVITY = np.array([0.0, 0.0, -9.81])


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def axis_angle_mat(a, ang):
    a = np.asarray(a, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def fk(skel, qpos):
    """-> R (nb,3,3), p (nb,3) body frames; axis_w (nj,3), anchor_w (nj,3) of every hinge in the world."""
    nb = len(skel.body_names)
    R, p = np.zeros((nb, 3, 3)), np.zeros((nb, 3))
    axis_w, anchor_w = np.zeros((len(skel.joint_names), 3)), np.zeros((len(skel.joint_names), 3))
    R[0], p[0] = quat_to_mat(qpos[3:7]), qpos[:3]
    j = 0
    for b in range(1, nb):
        par = int(skel.body_parent[b])
        Rb = R[par].copy()
        pb = p[par] + R[par] @ (skel.body_pos[b] - skel.body_pos[par])
        for _ in range(int(skel.body_ndof[b])):
            a_loc = skel.joint_axis[j]
            anc_loc = skel.joint_anchor[j] - skel.body_pos[b]
            axis_w[j] = Rb @ a_loc
            anchor_w[j] = pb + Rb @ anc_loc
            Rb = Rb @ axis_angle_mat(a_loc, qpos[7 + j])
            pb = anchor_w[j] - Rb @ anc_loc
            j += 1
        R[b], p[b] = Rb, pb
    return R, p, axis_w, anchor_w


def body_world_inertials(skel, R, p):
    """World COM (nb,3) and world inertia about the COM (nb,3,3) of every body."""
    com = p + np.einsum("bij,bj->bi", R, skel.body_com - skel.body_pos)
    Iw = np.einsum("bij,bjk,blk->bil", R, skel.body_inertia, R)
    return com, Iw


def _body_dof_chain(skel):
    nb = len(skel.body_names)
    own, d = [list(range(6))], 6
    for b in range(1, nb):
        own.append(list(range(d, d + int(skel.body_ndof[b]))))
        d += int(skel.body_ndof[b])
    chains = []
    for b in range(nb):
        c, q = [], b
        while q >= 0:
            c = own[q] + c
            q = int(skel.body_parent[q])
        chains.append(c)
    return chains


def jacobians(skel, qpos):
    """Geometric Jacobians of every body's COM: Jv, Jw (nb,3,nv) with MuJoCo's qvel convention."""
    R, p, axis_w, anchor_w = fk(skel, qpos)
    com, Iw = body_world_inertials(skel, R, p)
    nb, nv = len(skel.body_names), skel.nv
    Jv, Jw = np.zeros((nb, 3, nv)), np.zeros((nb, 3, nv))
    for b, chain in enumerate(_body_dof_chain(skel)):
        for d in chain:
            if d < 3:
                Jv[b, d, d] = 1.0
            elif d < 6:
                a = R[0][:, d - 3]                       # root angular velocity is expressed in the body frame
                Jw[b, :, d] = a
                Jv[b, :, d] = np.cross(a, com[b] - p[0])
            else:
                a = axis_w[d - 6]
                Jw[b, :, d] = a
                Jv[b, :, d] = np.cross(a, com[b] - anchor_w[d - 6])
    return Jv, Jw, com, Iw


def inertia_matrix(skel, qpos):
    """Dense joint-space inertia M(q) (nv,nv), armature included."""
    Jv, Jw, _, Iw = jacobians(skel, qpos)
    M = np.zeros((skel.nv, skel.nv))
    for b in range(len(skel.body_names)):
        M += skel.body_mass[b] * Jv[b].T @ Jv[b] + Jw[b].T @ Iw[b] @ Jw[b]
    M[np.arange(6, skel.nv), np.arange(6, skel.nv)] += skel.armature
    return 0.5 * (M + M.T)


def integrate_pos(qpos, qvel, t):
    """mj_integratePos: q(t) = q (+) qvel * t (root rotation: q * exp(omega_body t))."""
    out = np.array(qpos, float, copy=True)
    out[:3] += qvel[:3] * t
    w = np.asarray(qvel[3:6], float) * t
    ang = np.linalg.norm(w)
    if ang > 0:
        ax = w / ang
        dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
    else:
        dq = np.array([1.0, 0, 0, 0])
    a, b = out[3:7], dq
    out[3:7] = [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]
    out[3:7] /= np.linalg.norm(out[3:7])
    out[7:] += qvel[6:] * t
    return out


def _rot_vec(Ra, Rb):
    """rotation vector of Rb Ra^T (small angles)."""
    D = Rb @ Ra.T
    return 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])


def body_kinematics_fd(skel, qpos, qvel, h=1e-4):
    """COM velocity / acceleration and angular velocity / acceleration of every body along q(t) = q (+) qvel t,
    by central differences of the forward kinematics (4th order for the velocities)."""
    def at(t):
        R, p, _, _ = fk(skel, integrate_pos(qpos, qvel, t))
        com, _ = body_world_inertials(skel, R, p)
        return R, com
    (Rm2, cm2), (Rm, cm), (R0, c0), (Rp, cp), (Rp2, cp2) = at(-2 * h), at(-h), at(0.0), at(h), at(2 * h)
    nb = len(skel.body_names)
    v = (-cp2 + 8 * cp - 8 * cm + cm2) / (12 * h)
    a = (-cp2 + 16 * cp - 30 * c0 + 16 * cm - cm2) / (12 * h * h)
    w = np.zeros((nb, 3))
    wdot = np.zeros((nb, 3))
    for b in range(nb):
        w_p = _rot_vec(R0[b], Rp2[b]) / (2 * h)      # mean angular velocity over [0, 2h]  ~ w(h)
        w_m = _rot_vec(Rm2[b], R0[b]) / (2 * h)      # ~ w(-h)
        w[b] = _rot_vec(Rm[b], Rp[b]) / (2 * h)
        wdot[b] = (w_p - w_m) / (2 * h)
    return v, a, w, wdot


def bias_force(skel, qpos, qvel, h=1e-4):
    """qfrc_bias = inverse dynamics at zero joint acceleration: sum_b JvT m (a_b - g) + JwT (I w' + w x I w)."""
    Jv, Jw, _, Iw = jacobians(skel, qpos)
    _, a, w, wdot = body_kinematics_fd(skel, qpos, qvel, h)
    tau = np.zeros(skel.nv)
    for b in range(len(skel.body_names)):
        f = skel.body_mass[b] * (a[b] - GRAVITY)
        n = Iw[b] @ wdot[b] + np.cross(w[b], Iw[b] @ w[b])
        tau += Jv[b].T @ f + Jw[b].T @ n
    return tau


def kinetic_energy_fd(skel, qpos, qvel, h=1e-5):
    v, _, w, _ = body_kinematics_fd(skel, qpos, qvel, h)
    _, _, _, Iw = jacobians(skel, qpos)
    return sum(0.5 * skel.body_mass[b] * v[b] @ v[b] + 0.5 * w[b] @ Iw[b] @ w[b] for b in range(len(skel.body_names)))


# ---------------------------------------------------------------------- the kernel's formulation, restated (to localise failures)
def _cross_m(a, b):      # spatial motion cross product  [w; v] x [w'; v']
    return np.r_[np.cross(a[:3], b[:3]), np.cross(a[:3], b[3:]) + np.cross(a[3:], b[:3])]


def _cross_f(a, f):      # spatial force cross product   [w; v] x* [n; f]
    return np.r_[np.cross(a[:3], f[:3]) + np.cross(a[3:], f[3:]), np.cross(a[:3], f[3:])]


def _inertia_apply(m, hvec, Ibar, s):
    """spatial inertia (mass, first moment h = m c, rotational inertia about the world origin) times a motion vector."""
    w, v = s[:3], s[3:]
    return np.r_[Ibar @ w + np.cross(hvec, v), m * v + np.cross(w, hvec)]


def crba_rne_spatial(skel, qpos, qvel):
    """Composite-rigid-body M (dense) and recursive Newton-Euler bias, all spatial vectors in world coordinates about
    the world origin -- the algorithm of csrc/egp_dynamics.hip."""
    R, p, axis_w, anchor_w = fk(skel, qpos)
    com, Iw = body_world_inertials(skel, R, p)
    nb, nv = len(skel.body_names), skel.nv
    S = np.zeros((nv, 6))
    for d in range(3):
        S[d, 3 + d] = 1.0
        a = R[0][:, d]
        S[3 + d] = np.r_[a, np.cross(p[0], a)]
    for j in range(nv - 6):
        S[6 + j] = np.r_[axis_w[j], np.cross(anchor_w[j], axis_w[j])]
    m = skel.body_mass
    hv = m[:, None] * com
    Ibar = np.array([Iw[b] + m[b] * (com[b] @ com[b] * np.eye(3) - np.outer(com[b], com[b])) for b in range(nb)])
    # composite inertias, leaves -> root
    mc, hc, Ic = m.copy(), hv.copy(), Ibar.copy()
    for b in range(nb - 1, 0, -1):
        par = int(skel.body_parent[b])
        mc[par] += mc[b]; hc[par] += hc[b]; Ic[par] += Ic[b]
    dof_body = np.r_[np.zeros(6, int), skel.joint_body]
    M = np.zeros((nv, nv))
    for j in range(nv):
        b = dof_body[j]
        F = _inertia_apply(mc[b], hc[b], Ic[b], S[j])
        i = j
        while i >= 0:
            M[i, j] = M[j, i] = S[i] @ F
            i = int(skel.dof_parentid[i])
    M[np.arange(6, nv), np.arange(6, nv)] += skel.armature
    # RNE with zero joint acceleration; gravity enters as the base acceleration -g
    vb, ab = np.zeros((nb, 6)), np.zeros((nb, 6))
    a0 = np.r_[np.zeros(3), -GRAVITY]
    own = [list(range(6))]
    d = 6
    for b in range(1, nb):
        own.append(list(range(d, d + int(skel.body_ndof[b]))))
        d += int(skel.body_ndof[b])
    for b in range(nb):
        if b == 0:
            # free joint: the rotational axes are fixed in the root itself, so dS/dt = v_root x S with the FULL root velocity
            v_run = sum(S[dd] * qvel[dd] for dd in range(6))
            a_run = a0 + sum(_cross_m(v_run, S[dd]) * qvel[dd] for dd in range(3, 6))
        else:
            par = int(skel.body_parent[b])
            v_run, a_run = vb[par].copy(), ab[par].copy()
            for dd in own[b]:               # a hinge axis is fixed in the frame that precedes the hinge
                a_run = a_run + _cross_m(v_run, S[dd]) * qvel[dd]
                v_run = v_run + S[dd] * qvel[dd]
        vb[b], ab[b] = v_run, a_run
    fb = np.zeros((nb, 6))
    for b in range(nb):
        Iv = _inertia_apply(m[b], hv[b], Ibar[b], vb[b])
        fb[b] = _inertia_apply(m[b], hv[b], Ibar[b], ab[b]) + _cross_f(vb[b], Iv)
    for b in range(nb - 1, 0, -1):
        fb[int(skel.body_parent[b])] += fb[b]
    tau = np.array([S[dd] @ fb[dof_body[dd]] for dd in range(nv)])
    return M, tau, dict(R=R, p=p, com=com)
