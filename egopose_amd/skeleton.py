"""Humanoid skeleton tree: the read-only model data shared by every env.

The reference reads these quantities from ``mujoco_py``'s compiled model
(``model.body_names``, ``body_jntadr``, ``jnt_qposadr``, ``actuator_names``:
/root/reference/utils/tools.py:55-68, /root/reference/ego_pose/envs/humanoid_v1.py:113-125).
MuJoCo is not available here, so the tree is parsed from the MJCF (when the file
is at hand) or loaded from the compact JSON asset ``assets/humanoid_1205_v1.json``
that ``tools/make_skeleton_asset.py`` derives from
/root/reference/assets/mujoco_models/humanoid_1205_v1.xml:22-192.

Everything the HIP kernels stage in LDS comes from here:
  * body -> (qpos start, ndof)          (K2/K4 body quaternions)
  * dof_parentid / dof_Madr             (K1: MuJoCo sparse inertia ``qM`` -> dense)
  * end-effector body ids               (K2 end-effector term)
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
DEFAULT_ASSET = os.path.join(ASSET_DIR, "humanoid_1205_v1.json")

# order fixed by /root/reference/ego_pose/envs/humanoid_v1.py:100
EE_NAMES = ("LeftFoot", "RightFoot", "LeftHand", "RightHand", "Head")


@dataclass
class Skeleton:
    body_names: List[str]                 # without the world body
    body_parent: np.ndarray               # (nb,) int, -1 for the root body
    body_pos: np.ndarray                  # (nb,3) global position at the zero pose
    body_qpos_start: np.ndarray           # (nb,) int   (root: 0)
    body_ndof: np.ndarray                 # (nb,) int   (root: 6 dofs / 7 qpos)
    joint_names: List[str]                # hinge joints, qpos order (52)
    joint_body: np.ndarray                # (nj,) body index of each hinge
    joint_axis: np.ndarray                # (nj,3) axis at zero pose (global)
    joint_anchor: np.ndarray              # (nj,3) anchor at zero pose (global)
    joint_range: np.ndarray               # (nj,2) radians
    actuator_names: List[str]
    timestep: float
    armature: float
    body_mass: np.ndarray                 # (nb,)
    body_com: np.ndarray                  # (nb,3) global COM at zero pose
    body_inertia: np.ndarray              # (nb,3,3) about COM, global axes
    # derived
    nq: int = 0
    nv: int = 0
    nu: int = 0
    dof_parentid: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    dof_Madr: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    nM: int = 0
    ee_body: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    # ------------------------------------------------------------------ derived tables
    def finalize(self) -> "Skeleton":
        nb = len(self.body_names)
        self.nq = 7 + len(self.joint_names)
        self.nv = 6 + len(self.joint_names)
        self.nu = len(self.actuator_names)
        # dof tree (MuJoCo convention: consecutive dofs of one body chain onto each
        # other; the first dof of a body hangs off the last dof of its parent body)
        body_last_dof = np.full(nb, -1, np.int32)
        parent = np.full(self.nv, -1, np.int32)
        for d in range(1, 6):
            parent[d] = d - 1
        body_last_dof[0] = 5
        dof = 6
        for b in range(1, nb):
            nd = int(self.body_ndof[b])
            for k in range(nd):
                parent[dof] = body_last_dof[self.body_parent[b]] if k == 0 else dof - 1
                dof += 1
            body_last_dof[b] = dof - 1 if nd > 0 else body_last_dof[self.body_parent[b]]
        assert dof == self.nv
        self.dof_parentid = parent
        madr = np.zeros(self.nv, np.int32)
        adr = 0
        for i in range(self.nv):
            madr[i] = adr
            j = i
            while j >= 0:
                adr += 1
                j = parent[j]
        self.dof_Madr = madr
        self.nM = adr
        self.ee_body = np.array([self.body_names.index(n) for n in EE_NAMES], np.int32)
        return self

    # body name -> (qpos start, qpos end), as utils/tools.py:55-68 returns it
    def body_qposaddr(self) -> dict:
        out = {}
        for b, name in enumerate(self.body_names):
            s = int(self.body_qpos_start[b])
            n = 7 if b == 0 else int(self.body_ndof[b])
            out[name] = (s, s + n)
        return out

    # ------------------------------------------------------------------ sparse inertia helpers
    def sparse_index(self):
        """(rows, cols) of every qM entry in MuJoCo's legacy sparse order."""
        rows = np.zeros(self.nM, np.int32)
        cols = np.zeros(self.nM, np.int32)
        for i in range(self.nv):
            adr = self.dof_Madr[i]
            j = i
            while j >= 0:
                rows[adr] = i
                cols[adr] = j
                adr += 1
                j = self.dof_parentid[j]
        return rows, cols

    def full_from_sparse(self, qM: np.ndarray) -> np.ndarray:
        """What ``mj_fullM`` does (/root/reference/ego_pose/envs/humanoid_v1.py:133-135)."""
        rows, cols = self.sparse_index()
        M = np.zeros((self.nv, self.nv))
        M[rows, cols] = qM
        M[cols, rows] = qM
        return M

    def sparse_from_full(self, M: np.ndarray) -> np.ndarray:
        rows, cols = self.sparse_index()
        return np.ascontiguousarray(M[rows, cols])

    # ------------------------------------------------------------------ zero-pose inertia (CRBA by Jacobians)
    def zero_pose_inertia(self) -> np.ndarray:
        """Joint-space inertia at the zero pose (dense, nv x nv), armature included.

        Used only to give the surrogate physics backend a physically shaped SPD
        matrix with the correct tree sparsity; not a MuJoCo result.
        """
        nb, nv = len(self.body_names), self.nv
        # ancestors-or-self dof list per body
        body_dofs = [[] for _ in range(nb)]
        dof = 6
        own = [list(range(6))] + [[] for _ in range(nb - 1)]
        for b in range(1, nb):
            own[b] = list(range(dof, dof + int(self.body_ndof[b])))
            dof += int(self.body_ndof[b])
        for b in range(nb):
            chain, p = [], b
            while p >= 0:
                chain = own[p] + chain
                p = int(self.body_parent[p])
            body_dofs[b] = chain
        M = np.zeros((nv, nv))
        root = self.body_pos[0]
        for b in range(nb):
            m, c, I = self.body_mass[b], self.body_com[b], self.body_inertia[b]
            Jv = np.zeros((3, nv))
            Jw = np.zeros((3, nv))
            for d in body_dofs[b]:
                if d < 3:
                    Jv[d, d] = 1.0
                elif d < 6:
                    a = np.zeros(3)
                    a[d - 3] = 1.0
                    Jw[:, d] = a
                    Jv[:, d] = np.cross(a, c - root)
                else:
                    a = self.joint_axis[d - 6]
                    Jw[:, d] = a
                    Jv[:, d] = np.cross(a, c - self.joint_anchor[d - 6])
            M += m * Jv.T @ Jv + Jw.T @ I @ Jw
        M[np.arange(6, nv), np.arange(6, nv)] += self.armature
        return 0.5 * (M + M.T)

    # ------------------------------------------------------------------ forward kinematics (host-side reference)
    def body_xpos(self, qpos: np.ndarray) -> np.ndarray:
        """World positions of the body frames for one qpos (nb,3).

        MJCF ``coordinate="global"`` semantics: at the zero pose every body frame is
        axis-aligned and sits at ``body_pos``; hinges of a body rotate it (and its
        subtree) about their anchor, applied in joint order x->y->z.
        """
        nb = len(self.body_names)
        R = [None] * nb
        p = [None] * nb
        R[0] = _quat_to_mat(qpos[3:7])
        p[0] = np.asarray(qpos[:3], float)
        jidx = 0
        for b in range(1, nb):
            par = int(self.body_parent[b])
            Rb = R[par].copy()
            pb = p[par] + R[par] @ (self.body_pos[b] - self.body_pos[par])
            for k in range(int(self.body_ndof[b])):
                a = self.joint_axis[jidx]
                anchor_local = self.joint_anchor[jidx] - self.body_pos[b]
                ang = qpos[7 + jidx]
                Rj = _axis_angle_mat(a, ang)
                # rotate the body about the anchor (expressed in the body's current frame)
                anchor_w = pb + Rb @ anchor_local
                Rb = Rb @ Rj
                pb = anchor_w - Rb @ anchor_local
                jidx += 1
            R[b], p[b] = Rb, pb
        return np.stack(p)

    # ------------------------------------------------------------------ (de)serialisation
    def to_json(self) -> dict:
        return {
            "body_names": self.body_names,
            "body_parent": self.body_parent.tolist(),
            "body_pos": self.body_pos.tolist(),
            "body_qpos_start": self.body_qpos_start.tolist(),
            "body_ndof": self.body_ndof.tolist(),
            "joint_names": self.joint_names,
            "joint_body": self.joint_body.tolist(),
            "joint_axis": self.joint_axis.tolist(),
            "joint_anchor": self.joint_anchor.tolist(),
            "joint_range": self.joint_range.tolist(),
            "actuator_names": self.actuator_names,
            "timestep": self.timestep,
            "armature": self.armature,
            "body_mass": self.body_mass.tolist(),
            "body_com": self.body_com.tolist(),
            "body_inertia": self.body_inertia.tolist(),
        }

    @staticmethod
    def from_json(d: dict) -> "Skeleton":
        return Skeleton(
            body_names=list(d["body_names"]),
            body_parent=np.array(d["body_parent"], np.int32),
            body_pos=np.array(d["body_pos"], float),
            body_qpos_start=np.array(d["body_qpos_start"], np.int32),
            body_ndof=np.array(d["body_ndof"], np.int32),
            joint_names=list(d["joint_names"]),
            joint_body=np.array(d["joint_body"], np.int32),
            joint_axis=np.array(d["joint_axis"], float),
            joint_anchor=np.array(d["joint_anchor"], float),
            joint_range=np.array(d["joint_range"], float),
            actuator_names=list(d["actuator_names"]),
            timestep=float(d["timestep"]),
            armature=float(d["armature"]),
            body_mass=np.array(d["body_mass"], float),
            body_com=np.array(d["body_com"], float),
            body_inertia=np.array(d["body_inertia"], float),
        ).finalize()


def _quat_to_mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _axis_angle_mat(a, ang):
    a = np.asarray(a, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


# ---------------------------------------------------------------------- MJCF parsing
def _floats(s):
    return np.array([float(x) for x in s.split()], float)


def _geom_mass_props(g, density=1000.0):
    """mass, com, inertia (global axes, about com) of a sphere / capsule / box geom."""
    t = g.get("type", "sphere")
    size = _floats(g.get("size"))
    if t == "sphere":
        r = size[0]
        m = density * 4.0 / 3.0 * np.pi * r ** 3
        return m, _floats(g.get("pos")), np.eye(3) * (0.4 * m * r * r)
    if t == "capsule":
        ft = _floats(g.get("fromto"))
        p0, p1 = ft[:3], ft[3:]
        r = size[0]
        h = np.linalg.norm(p1 - p0)
        axis = (p1 - p0) / h
        mc = density * np.pi * r * r * h
        ms = density * 4.0 / 3.0 * np.pi * r ** 3
        m = mc + ms
        # cylinder + two hemispheres (standard solid-capsule formulas)
        i_ax = 0.5 * mc * r * r + 0.4 * ms * r * r
        i_tr = mc * (h * h / 12.0 + r * r / 4.0) + ms * (0.4 * r * r + 0.375 * r * h + 0.25 * h * h)
        P = np.outer(axis, axis)
        return m, 0.5 * (p0 + p1), i_ax * P + i_tr * (np.eye(3) - P)
    if t == "box":
        hx, hy, hz = size
        m = density * 8 * hx * hy * hz
        I = np.diag([m / 3.0 * (hy * hy + hz * hz), m / 3.0 * (hx * hx + hz * hz), m / 3.0 * (hx * hx + hy * hy)])
        return m, _floats(g.get("pos")), I
    raise ValueError("unsupported geom type %s" % t)


def parse_mjcf(path: str) -> Skeleton:
    """Read the humanoid tree out of an MJCF file written with coordinate="global"."""
    import xml.etree.ElementTree as ET

    root = ET.parse(path).getroot()
    comp = root.find("compiler")
    assert comp is not None and comp.get("coordinate") == "global", "expects coordinate=global MJCF"
    deg = comp.get("angle", "degree") == "degree"
    jdef = root.find("default/joint")
    armature = float(jdef.get("armature", "0")) if jdef is not None else 0.0
    timestep = float(root.find("option").get("timestep"))

    names, parents, pos, qstart, ndof = [], [], [], [], []
    jn, jb, jax, janc, jrng = [], [], [], [], []
    mass, com, inertia = [], [], []
    qadr = [0]

    def visit(elem, parent_idx):
        idx = len(names)
        names.append(elem.get("name"))
        parents.append(parent_idx)
        pos.append(_floats(elem.get("pos")))
        joints = elem.findall("joint")
        if joints and joints[0].get("type") == "free":
            qstart.append(qadr[0])
            ndof.append(6)
            qadr[0] += 7
        else:
            qstart.append(qadr[0])
            ndof.append(len(joints))
            for j in joints:
                assert j.get("type", "hinge") == "hinge"
                jn.append(j.get("name"))
                jb.append(idx)
                jax.append(_floats(j.get("axis")))
                janc.append(_floats(j.get("pos")))
                r = _floats(j.get("range"))
                jrng.append(np.deg2rad(r) if deg else r)
                qadr[0] += 1
        m_tot, c_acc, parts = 0.0, np.zeros(3), []
        for g in elem.findall("geom"):
            m, c, I = _geom_mass_props(g)
            parts.append((m, c, I))
            m_tot += m
            c_acc += m * c
        c_tot = c_acc / m_tot
        I_tot = np.zeros((3, 3))
        for m, c, I in parts:
            d = c - c_tot
            I_tot += I + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        mass.append(m_tot)
        com.append(c_tot)
        inertia.append(I_tot)
        for child in elem.findall("body"):
            visit(child, idx)

    visit(root.find("worldbody").find("body"), -1)
    acts = [m.get("name") for m in root.find("actuator").findall("motor")]
    return Skeleton(
        body_names=names, body_parent=np.array(parents, np.int32), body_pos=np.array(pos),
        body_qpos_start=np.array(qstart, np.int32), body_ndof=np.array(ndof, np.int32),
        joint_names=jn, joint_body=np.array(jb, np.int32), joint_axis=np.array(jax),
        joint_anchor=np.array(janc), joint_range=np.array(jrng), actuator_names=acts,
        timestep=timestep, armature=armature, body_mass=np.array(mass), body_com=np.array(com),
        body_inertia=np.array(inertia)).finalize()


_CACHE = {}


def load_skeleton(path: Optional[str] = None) -> Skeleton:
    """MJCF path -> parse it; JSON path or None -> the packaged asset."""
    key = path or DEFAULT_ASSET
    if key in _CACHE:
        return _CACHE[key]
    if key.endswith(".xml"):
        sk = parse_mjcf(key)
    else:
        with open(key, "r") as f:
            sk = Skeleton.from_json(json.load(f))
    _CACHE[key] = sk
    return sk
