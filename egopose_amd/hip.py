"""Thin torch-tensor front end of the C-ABI: device memory + streams come from torch, the
arithmetic is the HIP library's. Every wrapper validates shapes/dtypes, launches on torch's
current stream and raises on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .skeleton import Skeleton

_DT = {torch.float64: "f64", torch.float32: "f32"}


def _stream():
    return L.current_stream()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _need(t: torch.Tensor, shape, dtype, name):
    if not t.is_cuda:
        raise ValueError("%s must be a CUDA (HIP) tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _np_i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _np_f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ip(a):
    return a.ctypes.data_as(L.c_int_p)


def _dp(a):
    return a.ctypes.data_as(L.c_dbl_p)


DEFAULT_REWARD = dict(w_p=0.5, w_v=0.1, w_e=0.2, w_rp=0.1, w_rv=0.1, k_p=2.0, k_v=0.005, k_e=20.0,
                      k_rh=300.0, k_rq=300.0, k_rl=5.0, k_ra=0.5, v_ord=2, decay=False)


QUAT_OPS = dict(mul=(0, 4, 4, 4), inv=(1, 4, 0, 4), from_euler_sxyz=(2, 3, 0, 4), heading_q=(3, 4, 0, 4), de_heading=(4, 4, 0, 4),
                transform_vec_root=(5, 3, 4, 3), transform_vec_heading=(6, 3, 4, 3), rotation=(7, 4, 0, 4), diff_half_angle=(8, 4, 4, 1))


def quat_op(name, a, b=None):
    """Batched quaternion algebra on the device (`egp_quat_op_*`, include/egopose_hip.h): name -> (op, width a, width b, width out)."""
    op, wa, wb, wo = QUAT_OPS[name]
    n = a.shape[0]
    _need(a, (n, wa), a.dtype, "a")
    if wb:
        if b is None:
            raise ValueError("%s needs two operands" % name)
        _need(b, (n, wb), a.dtype, "b")
    out = torch.empty((n, wo) if wo > 1 else (n,), dtype=a.dtype, device=a.device)
    fn = getattr(L.load(), "egp_quat_op_" + _DT[a.dtype])
    L.check(fn(op, _ptr(a), _ptr(b if wb else None), n, _ptr(out), _stream()), "egp_quat_op")
    return out


def obs_options_of(cfg):
    """The env switches of a Config (egomimic_config.py:99-105) as EgpContext's `obs_options`: the observation variants,
    the frame `obs_coord` (which the reward uses too, reward_function.py:19,23) and `action_type`."""
    if getattr(cfg, "obs_type", "full") != "full":
        raise NotImplementedError("obs_type %r: the reference's get_obs only knows 'full' (humanoid_v1.py:68-71)" % cfg.obs_type)
    return dict(obs_heading=bool(getattr(cfg, "obs_heading", False)), root_deheading=bool(getattr(cfg, "root_deheading", True)),
                obs_coord=getattr(cfg, "obs_coord", "heading"), obs_vel=getattr(cfg, "obs_vel", "full"),
                action_type=getattr(cfg, "action_type", "position"), obs_phase=bool(getattr(cfg, "obs_phase", False)))


class EgpContext:
    """Model constants + expert table resident in HBM (``egp_ctx``)."""

    def __init__(self, skel: Skeleton, jkp, jkd, a_ref, a_scale, torque_lim, b_diffw, reward_weights=None,
                 episode_len=200, frame_skip=15, device: int = 0, obs_options=None):
        """`obs_options`: dict with any of obs_heading / root_deheading / obs_coord / obs_vel (the config keys of
        egomimic_config.py:99-103; `obs_options_of(cfg)`); None = the defaults every shipped config uses."""
        self.lib = L.load()
        self.skel = skel
        self.device = int(device)
        self.frame_skip = int(frame_skip)
        self.episode_len = int(episode_len)
        self.nq, self.nv, self.nu, self.nbody, self.nM = skel.nq, skel.nv, skel.nu, len(skel.body_names), skel.nM
        oo = dict(obs_heading=False, root_deheading=True, obs_coord="heading", obs_vel="full", action_type="position", obs_phase=False)
        oo.update(obs_options or {})
        if oo["obs_coord"] not in ("heading", "root"):
            raise ValueError("obs_coord must be 'heading' or 'root', got %r" % (oo["obs_coord"],))      # transform_vec asserts
        if oo["action_type"] not in ("position", "torque"):
            # humanoid_v1.py:167-172 knows these two; anything else leaves `torque` unbound there (UnboundLocalError)
            raise ValueError("action_type must be 'position' or 'torque', got %r" % (oo["action_type"],))
        self.obs_options = oo
        self._obs_vel = {"full": 0, "root": 1}.get(oo["obs_vel"], 2)        # anything else: no velocity block (humanoid_v1.py:86-89)
        self.obs_phase = bool(oo["obs_phase"])       # humanoid_v1.py:92-94: a last column min(cur_t / env_episode_len, 1); kernels then need the rows' cur_t
        self.obs_dim = (1 if oo["obs_heading"] else 0) + self.nq - 2 + (self.nv, 6, 0)[self._obs_vel] + (1 if self.obs_phase else 0)
        self._keep = dict(
            bqs=_np_i32(skel.body_qpos_start), bnd=_np_i32(skel.body_ndof), dpar=_np_i32(skel.dof_parentid),
            madr=_np_i32(skel.dof_Madr), ee=_np_i32(skel.ee_body), jkp=_np_f64(jkp), jkd=_np_f64(jkd),
            a_ref=_np_f64(a_ref), a_scale=_np_f64(a_scale), tl=_np_f64(torque_lim), bw=_np_f64(b_diffw))
        for k in ("jkp", "jkd", "a_ref", "a_scale", "tl"):
            if self._keep[k].shape != (self.nu,):
                raise ValueError("%s must have %d entries" % (k, self.nu))
        if self._keep["bw"].shape != (self.nbody - 1,):
            raise ValueError("b_diffw must have %d entries" % (self.nbody - 1))
        self.reward_weights = None
        desc = self._desc(reward_weights)
        h = C.c_void_p()
        L.check(self.lib.egp_create(C.byref(desc), self.device, C.byref(h)), "egp_create")
        self.handle = h
        if int(self.lib.egp_obs_dim(h)) != self.obs_dim:
            raise RuntimeError("observation width: library %d, binding %d" % (self.lib.egp_obs_dim(h), self.obs_dim))
        self.n_frames = 0
        self.take_offset = None
        self.head_height_lb = None
        self._ws = {}

    def _desc(self, reward_weights):
        ws = dict(DEFAULT_REWARD)
        if reward_weights:
            ws.update(reward_weights)
        self.reward_weights = ws
        k = self._keep
        d = L.ModelDesc()
        d.nq, d.nv, d.nu, d.nbody, d.nM = self.nq, self.nv, self.nu, self.nbody, self.nM
        d.body_qpos_start, d.body_ndof, d.dof_parentid, d.dof_Madr, d.ee_body = (
            _ip(k["bqs"]), _ip(k["bnd"]), _ip(k["dpar"]), _ip(k["madr"]), _ip(k["ee"]))
        d.jkp, d.jkd, d.a_ref, d.a_scale, d.torque_lim, d.b_diffw = (
            _dp(k["jkp"]), _dp(k["jkd"]), _dp(k["a_ref"]), _dp(k["a_scale"]), _dp(k["tl"]), _dp(k["bw"]))
        d.sub_dt, d.frame_skip, d.episode_len = float(self.skel.timestep), self.frame_skip, self.episode_len
        for f in ("w_p", "w_v", "w_e", "w_rp", "w_rv", "k_p", "k_v", "k_e", "k_rh", "k_rq", "k_rl", "k_ra", "v_ord"):
            setattr(d, f, float(ws[f]))
        d.decay = 1 if ws.get("decay", False) else 0
        oo = self.obs_options
        d.obs_heading = 1 if oo["obs_heading"] else 0
        d.obs_keep_root_heading = 0 if oo["root_deheading"] else 1
        d.obs_coord_root = 1 if oo["obs_coord"] == "root" else 0
        d.obs_vel = self._obs_vel
        d.action_torque = 1 if oo["action_type"] == "torque" else 0
        d.obs_phase = 1 if oo["obs_phase"] else 0
        return d

    def close(self):
        if getattr(self, "handle", None):
            self.lib.egp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ configuration
    def set_reward_weights(self, reward_weights):
        desc = self._desc(reward_weights)
        L.check(self.lib.egp_set_reward_weights(self.handle, C.byref(desc)), "egp_set_reward_weights")

    def set_pd_variant(self, variant: int):
        L.check(self.lib.egp_set_pd_variant(self.handle, int(variant)), "egp_set_pd_variant")

    def upload_experts(self, takes):
        """takes: list of dicts with the expert keys of gen_expert.py:28-83 (float64 arrays)."""
        if not takes:
            raise ValueError("expert list is empty")
        lens = [int(t["qpos"].shape[0]) for t in takes]
        off = _np_i32(np.concatenate([[0], np.cumsum(lens)]))
        cat = {k: _np_f64(np.concatenate([np.asarray(t[k]).reshape(n, -1) for t, n in zip(takes, lens)], axis=0))
               for k in ("qpos", "qvel", "rlinv_local", "rangv", "rq_rmh", "ee_pos", "bquat", "bangvel")}
        lb = _np_f64([float(t["head_height_lb"]) for t in takes])
        tb = L.ExpertTable()
        tb.n_takes, tb.n_frames, tb.take_offset = len(takes), int(off[-1]), _ip(off)
        for k, v in cat.items():
            setattr(tb, k, _dp(v))
        tb.head_height_lb = _dp(lb)
        L.check(self.lib.egp_upload_experts(self.handle, C.byref(tb)), "egp_upload_experts")
        self.n_frames, self.take_offset, self.head_height_lb = int(off[-1]), off.astype(np.int64), lb

    # ------------------------------------------------------------------ kernels
    def _sfx(self, t):
        try:
            return _DT[t.dtype]
        except KeyError:
            raise ValueError("only float32/float64 tensors are supported, got %s" % t.dtype)

    def body_quat(self, qpos, out=None):
        n = qpos.shape[0]
        _need(qpos, (n, self.nq), qpos.dtype, "qpos")
        out = torch.empty(n, 4 * self.nbody, dtype=qpos.dtype, device=qpos.device) if out is None else out
        _need(out, (n, 4 * self.nbody), qpos.dtype, "out")
        fn = getattr(self.lib, "egp_body_quat_" + self._sfx(qpos))
        L.check(fn(self.handle, _ptr(qpos), n, _ptr(out), _stream()), "egp_body_quat")
        return out

    def obs(self, qpos, qvel, out=None, phase_t=None):
        n = qpos.shape[0]
        phase_t = self._phase_t(phase_t, n)
        _need(qpos, (n, self.nq), qpos.dtype, "qpos")
        _need(qvel, (n, self.nv), qpos.dtype, "qvel")
        out = torch.empty(n, self.obs_dim, dtype=qpos.dtype, device=qpos.device) if out is None else out
        _need(out, (n, self.obs_dim), qpos.dtype, "out")
        fn = getattr(self.lib, "egp_obs_" + self._sfx(qpos))
        L.check(fn(self.handle, _ptr(qpos), _ptr(qvel), _ptr(phase_t), n, _ptr(out), _stream()), "egp_obs")
        return out

    def pd_torque(self, qpos, qvel, action, qM, bias, want_raw=False):
        n, dt = qpos.shape[0], qpos.dtype
        _need(qpos, (n, self.nq), dt, "qpos")
        _need(qvel, (n, self.nv), dt, "qvel")
        _need(action, (n, self.nu), dt, "action")
        _need(qM, (n, self.nM), dt, "qM")
        _need(bias, (n, self.nv), dt, "qfrc_bias")
        tq = torch.empty(n, self.nu, dtype=dt, device=qpos.device)
        raw = torch.empty_like(tq) if want_raw else None
        fn = getattr(self.lib, "egp_pd_torque_" + self._sfx(qpos))
        L.check(fn(self.handle, _ptr(qpos), _ptr(qvel), _ptr(action), _ptr(qM), _ptr(bias), n, _ptr(tq), _ptr(raw),
                   _stream()), "egp_pd_torque")
        return (tq, raw) if want_raw else tq

    REWARD_KINDS = {"quat_v3": 0, "constant": 1, "pose_dist": 2}       # reward_function.py:78-80

    def reward_cinfo_dim(self, kind="quat_v3"):
        return 5 if kind == "quat_v3" else 1

    def reward_simple(self, kind, cur_qpos, frame, end, end_reward, active=None, reward_out=None, cinfo_out=None):
        """`constant` / `pose_dist` (egp_reward_simple_f64): float64, c_info (n, 1)."""
        n = cur_qpos.shape[0]
        _need(cur_qpos, (n, self.nq), torch.float64, "cur_qpos")
        for name, a in (("frame", frame), ("end", end)):
            _need(a, (n,), torch.int32, name)
        r = torch.empty(n, dtype=torch.float64, device=cur_qpos.device) if reward_out is None else reward_out
        ci = torch.empty(n, 1, dtype=torch.float64, device=cur_qpos.device) if cinfo_out is None else cinfo_out
        _need(r, (n,), torch.float64, "reward_out")
        _need(ci, (n, 1), torch.float64, "cinfo_out")
        L.check(self.lib.egp_reward_simple_f64(self.handle, self.REWARD_KINDS[kind], _ptr(cur_qpos), _ptr(frame), _ptr(end), _ptr(active),
                                               float(end_reward), n, _ptr(r), _ptr(ci), _stream()), "egp_reward_simple_f64")
        return r, ci

    def reward(self, cur_qpos, prev_qpos, ee_wpos, t, frame, end, end_reward, active=None, reward_out=None,
               cinfo_out=None, kind="quat_v3"):
        if kind != "quat_v3":
            return self.reward_simple(kind, cur_qpos, frame, end, end_reward, active, reward_out, cinfo_out)
        n, dt = cur_qpos.shape[0], cur_qpos.dtype
        _need(cur_qpos, (n, self.nq), dt, "cur_qpos")
        _need(prev_qpos, (n, self.nq), dt, "prev_qpos")
        _need(ee_wpos, (n, 15), dt, "ee_wpos")
        for name, a in (("t", t), ("frame", frame), ("end", end)):
            _need(a, (n,), torch.int32, name)
        if active is not None:
            _need(active, (n,), torch.int32, "active")
        r = torch.empty(n, dtype=dt, device=cur_qpos.device) if reward_out is None else reward_out
        ci = torch.empty(n, 5, dtype=dt, device=cur_qpos.device) if cinfo_out is None else cinfo_out
        _need(r, (n,), dt, "reward_out")
        _need(ci, (n, 5), dt, "cinfo_out")
        fn = getattr(self.lib, "egp_reward_quat_v3_" + self._sfx(cur_qpos))
        L.check(fn(self.handle, _ptr(cur_qpos), _ptr(prev_qpos), _ptr(ee_wpos), _ptr(t), _ptr(frame), _ptr(end),
                   _ptr(active), float(end_reward), n, _ptr(r), _ptr(ci), _stream()), "egp_reward_quat_v3")
        return r, ci

    # ------------------------------------------------------------------ K8: dynamics terms on the GPU
    def set_dynamics_model(self, skel=None, gravity=(0.0, 0.0, -9.81)):
        """Upload the inertial tree (masses, COMs, inertias, hinge axes / anchors at the zero pose) for `dynamics`."""
        sk = skel if skel is not None else self.skel
        keep = dict(parent=np.ascontiguousarray(sk.body_parent, np.int32), pos=np.ascontiguousarray(sk.body_pos, np.float64),
                    com=np.ascontiguousarray(sk.body_com, np.float64), inertia=np.ascontiguousarray(sk.body_inertia, np.float64),
                    mass=np.ascontiguousarray(sk.body_mass, np.float64), ndof=np.ascontiguousarray(sk.body_ndof, np.int32),
                    axis=np.ascontiguousarray(sk.joint_axis, np.float64), anchor=np.ascontiguousarray(sk.joint_anchor, np.float64))
        d = L.DynamicsDesc()
        d.nbody, d.njoint = len(sk.body_names), len(sk.joint_names)
        d.body_parent = keep["parent"].ctypes.data_as(L.c_int_p)
        d.body_pos = keep["pos"].ctypes.data_as(L.c_dbl_p)
        d.body_com = keep["com"].ctypes.data_as(L.c_dbl_p)
        d.body_inertia = keep["inertia"].ctypes.data_as(L.c_dbl_p)
        d.body_mass = keep["mass"].ctypes.data_as(L.c_dbl_p)
        d.body_ndof = keep["ndof"].ctypes.data_as(L.c_int_p)
        d.joint_axis = keep["axis"].ctypes.data_as(L.c_dbl_p)
        d.joint_anchor = keep["anchor"].ctypes.data_as(L.c_dbl_p)
        d.armature = float(sk.armature)
        for i in range(3):
            d.gravity[i] = float(gravity[i])
        L.check(self.lib.egp_set_dynamics_model(self.handle, C.byref(d)), "egp_set_dynamics_model")
        self._has_dynamics = True

    def dynamics(self, qpos, qvel, want_qM=True, want_bias=True, want_xpos=False, qM_out=None):
        """(qpos, qvel) float64 [n] -> dict(qM [n][nM] legacy sparse, bias [n][nv], xpos [n][nbody][3]) on the GPU."""
        if not getattr(self, "_has_dynamics", False):
            self.set_dynamics_model()
        n = qpos.shape[0]
        _need(qpos, (n, self.nq), torch.float64, "qpos")
        _need(qvel, (n, self.nv), torch.float64, "qvel")
        dev = qpos.device
        out = {}
        ld = self.nM
        if qM_out is not None:
            assert qM_out.dtype == torch.float64 and qM_out.shape[0] == n and qM_out.stride(1) == 1 and qM_out.shape[1] >= self.nM
            out["qM"], ld = qM_out, qM_out.stride(0)
        elif want_qM:
            out["qM"] = torch.empty(n, self.nM, dtype=torch.float64, device=dev)
        if want_bias:
            out["bias"] = torch.empty(n, self.nv, dtype=torch.float64, device=dev)
        if want_xpos:
            out["xpos"] = torch.empty(n, self.nbody, 3, dtype=torch.float64, device=dev)
        L.check(self.lib.egp_dynamics_f64(self.handle, _ptr(qpos), _ptr(qvel), n, _ptr(out.get("qM")), int(ld), _ptr(out.get("bias")),
                                          _ptr(out.get("xpos")), _stream()), "egp_dynamics_f64")
        return out

    def pose_features(self, cur_qpos, prev_qpos, ee_wpos, expert_convention=False):
        """-> dict(qvel, rlinv_local, rangv, rq_rmh, ee_pos, bquat, bangvel) device tensors (K7)."""
        n, dt = cur_qpos.shape[0], cur_qpos.dtype
        _need(cur_qpos, (n, self.nq), dt, "cur_qpos")
        _need(prev_qpos, (n, self.nq), dt, "prev_qpos")
        _need(ee_wpos, (n, 15), dt, "ee_wpos")
        mk = lambda d: torch.empty(n, d, dtype=dt, device=cur_qpos.device)
        out = dict(qvel=mk(self.nv), rlinv_local=mk(3), rangv=mk(3), rq_rmh=mk(4), ee_pos=mk(15),
                   bquat=mk(4 * self.nbody), bangvel=mk(3 * self.nbody))
        fn = getattr(self.lib, "egp_pose_features_" + self._sfx(cur_qpos))
        L.check(fn(self.handle, _ptr(cur_qpos), _ptr(prev_qpos), _ptr(ee_wpos), n, 1 if expert_convention else 0,
                   _ptr(out["qvel"]), _ptr(out["rlinv_local"]), _ptr(out["rangv"]), _ptr(out["rq_rmh"]), _ptr(out["ee_pos"]),
                   _ptr(out["bquat"]), _ptr(out["bangvel"]), _stream()), "egp_pose_features")
        return out

    def _workspace(self, key, nbytes, device):
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def zfilter(self, x, state_in, state_out=None, update=True, clip=5.0, active=None, out=None):
        """state = float64 [1 + 2*dim] = (count, mean, S). Returns y (and writes state_out when update)."""
        n, dim = x.shape
        _need(x, (n, dim), x.dtype, "x")
        _need(state_in, (1 + 2 * dim,), torch.float64, "state_in")
        if update:
            if state_out is None:
                raise ValueError("update=True needs state_out")
            _need(state_out, (1 + 2 * dim,), torch.float64, "state_out")
        if active is not None:
            _need(active, (n,), torch.int32, "active")
        y = torch.empty_like(x) if out is None else out
        _need(y, (n, dim), x.dtype, "out")
        ws = self._workspace("zf", self.lib.egp_zfilter_workspace_bytes(n, dim), x.device) if update else None
        fn = getattr(self.lib, "egp_zfilter_" + self._sfx(x))
        L.check(fn(_ptr(x), _ptr(active), n, dim, _ptr(state_in), _ptr(state_out if update else None), 1 if update else 0,
                   float(clip or 0.0), _ptr(y), _ptr(ws), _stream()), "egp_zfilter")
        return y

    def _phase_t(self, phase_t, n):
        """`phase_t`: int32 [n] cur_t of the rows, required when the model has obs_phase (ignored otherwise)."""
        if not self.obs_phase:
            return None
        if phase_t is None:
            raise ValueError("the model has obs_phase: pass phase_t (int32 cur_t of every row)")
        _need(phase_t, (n,), torch.int32, "phase_t")
        return phase_t

    def obs_zfilter(self, qpos, qvel, state_in, state_out, clip, out, out2=None, active=None, write_only_active=False, phase_t=None):
        """K3+K6 fused: filtered observations of (qpos, qvel) written to ``out`` (and ``out2``); state_in None = raw."""
        n, dt = qpos.shape[0], qpos.dtype
        phase_t = self._phase_t(phase_t, n)
        _need(qpos, (n, self.nq), dt, "qpos")
        _need(qvel, (n, self.nv), dt, "qvel")
        _need(out, (n, self.obs_dim), dt, "out")
        if out2 is not None:
            _need(out2, (n, self.obs_dim), dt, "out2")
        if active is not None:
            _need(active, (n,), torch.int32, "active")
        ws = None
        if state_in is not None:
            _need(state_in, (1 + 2 * self.obs_dim,), torch.float64, "state_in")
            _need(state_out, (1 + 2 * self.obs_dim,), torch.float64, "state_out")
            ws = self._workspace("zf", self.lib.egp_zfilter_workspace_bytes(n, self.obs_dim), qpos.device)
        fn = getattr(self.lib, "egp_obs_zfilter_" + self._sfx(qpos))
        L.check(fn(self.handle, _ptr(qpos), _ptr(qvel), _ptr(phase_t), _ptr(active), n, _ptr(state_in), _ptr(state_out), float(clip or 0.0),
                   _ptr(out), _ptr(out2), 1 if write_only_active else 0, _ptr(ws), _stream()), "egp_obs_zfilter")
        return out

    def obs_zfilter_stats(self, qpos, qvel, workspace, active=None, phase_t=None):
        """First launch of obs_zfilter on its own (`egp_obs_zfilter_stats_f64`): tile statistics into `workspace`."""
        n = qpos.shape[0]
        L.check(self.lib.egp_obs_zfilter_stats_f64(self.handle, _ptr(qpos), _ptr(qvel), _ptr(self._phase_t(phase_t, n)), _ptr(active), n, _ptr(workspace), _stream()),
                "egp_obs_zfilter_stats_f64")

    def obs_zfilter_apply(self, qpos, qvel, state_in, state_out, clip, out, out2, workspace, phase_t=None):
        """Second launch of obs_zfilter on its own (`egp_obs_zfilter_apply_f64`)."""
        n = qpos.shape[0]
        L.check(self.lib.egp_obs_zfilter_apply_f64(self.handle, _ptr(qpos), _ptr(qvel), _ptr(self._phase_t(phase_t, n)), n, _ptr(state_in), _ptr(state_out), float(clip or 0.0),
                                                   _ptr(out), _ptr(out2), _ptr(workspace), _stream()), "egp_obs_zfilter_apply_f64")
        return out

    def gae(self, rewards, masks, values, gamma, tau):
        """-> (adv_raw (n,), returns (n,), stats float64[3] = {n, mean, M2}) all on device."""
        n, dt = rewards.shape[0], rewards.dtype
        _need(rewards, (n,), dt, "rewards")
        _need(masks, (n,), dt, "masks")
        _need(values, (n,), dt, "values")
        adv = torch.empty_like(rewards)
        ret = torch.empty_like(rewards)
        stats = torch.empty(3, dtype=torch.float64, device=rewards.device)
        ws = self._workspace("gae", self.lib.egp_gae_workspace_bytes(n), rewards.device)
        fn = getattr(self.lib, "egp_gae_" + self._sfx(rewards))
        L.check(fn(_ptr(rewards), _ptr(masks), _ptr(values), n, float(gamma), float(tau), _ptr(adv), _ptr(ret),
                   _ptr(stats), _ptr(ws), _stream()), "egp_gae")
        return adv, ret, stats

    def gae_standardize(self, adv, stats):
        _need(adv, (adv.shape[0],), adv.dtype, "adv")
        _need(stats, (3,), torch.float64, "stats")
        fn = getattr(self.lib, "egp_gae_standardize_" + self._sfx(adv))
        L.check(fn(_ptr(adv), adv.shape[0], _ptr(stats), _stream()), "egp_gae_standardize")
        return adv
