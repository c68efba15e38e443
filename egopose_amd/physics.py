"""Python handles of the host physics boundary and the lockstep rollout engine (C-ABI: egp_physics_*,
egp_engine_*). The reference couples HumanoidEnv to mujoco_py's MjSim
(/root/reference/envs/common/mujoco_env.py:18-42); here physics is a pluggable host backend and the
only built-in one is the deterministic surrogate (not MuJoCo -- physics parity is unpinned).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _lib as L
from .skeleton import Skeleton


def _np_i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _np_f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class SurrogatePhysics:
    """egp_physics_create_surrogate: semi-implicit Euler on qacc = M0^-1 (tau - C)."""

    def __init__(self, skel: Skeleton, n_env: int, damping: float = 1.0, support_k: float = 2000.0,
                 support_c: float = 200.0):
        self.lib = L.load()
        self.skel = skel
        self.n_env = int(n_env)
        M0 = skel.zero_pose_inertia()
        self.qM0 = _np_f64(skel.sparse_from_full(M0))
        self.Minv0 = _np_f64(np.linalg.inv(M0))
        self._keep = dict(
            body_parent=_np_i32(skel.body_parent), body_pos=_np_f64(skel.body_pos.ravel()),
            body_ndof=_np_i32(skel.body_ndof), joint_axis=_np_f64(skel.joint_axis.ravel()),
            joint_anchor=_np_f64(skel.joint_anchor.ravel()))
        d = L.SurrogateDesc()
        d.nq, d.nv, d.nu, d.nbody, d.nM, d.njoint = skel.nq, skel.nv, skel.nu, len(skel.body_names), skel.nM, len(skel.joint_names)
        d.qM0 = self.qM0.ctypes.data_as(L.c_dbl_p)
        d.Minv0 = self.Minv0.ctypes.data_as(L.c_dbl_p)
        d.body_parent = self._keep["body_parent"].ctypes.data_as(L.c_int_p)
        d.body_pos = self._keep["body_pos"].ctypes.data_as(L.c_dbl_p)
        d.body_ndof = self._keep["body_ndof"].ctypes.data_as(L.c_int_p)
        d.joint_axis = self._keep["joint_axis"].ctypes.data_as(L.c_dbl_p)
        d.joint_anchor = self._keep["joint_anchor"].ctypes.data_as(L.c_dbl_p)
        d.sub_dt, d.damping, d.support_k, d.support_c = float(skel.timestep), float(damping), float(support_k), float(support_c)
        h = C.c_void_p()
        L.check(self.lib.egp_physics_create_surrogate(C.byref(d), self.n_env, C.byref(h)), "egp_physics_create_surrogate")
        self.handle = h

    @property
    def name(self):
        return self.lib.egp_physics_name(self.handle).decode()

    # single-env host access (CPU baseline / tests)
    def reset(self, env, qpos, qvel):
        qpos, qvel = _np_f64(qpos), _np_f64(qvel)
        L.check(self.lib.egp_physics_reset_host(self.handle, int(env), qpos.ctypes.data, qvel.ctypes.data), "physics reset")

    def step(self, env, ctrl):
        ctrl = _np_f64(ctrl)
        L.check(self.lib.egp_physics_step_host(self.handle, int(env), ctrl.ctypes.data), "physics step")

    def drain(self, env, want_xpos=True):
        sk = self.skel
        qpos, qvel = np.empty(sk.nq), np.empty(sk.nv)
        qM, bias = np.empty(sk.nM), np.empty(sk.nv)
        xpos = np.empty((len(sk.body_names), 3)) if want_xpos else None
        L.check(self.lib.egp_physics_drain_host(self.handle, int(env), qpos.ctypes.data, qvel.ctypes.data, qM.ctypes.data,
                                                bias.ctypes.data, xpos.ctypes.data if want_xpos else None), "physics drain")
        return qpos, qvel, qM, bias, xpos

    def close(self):
        if getattr(self, "handle", None):
            self.lib.egp_physics_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CallbackPhysics:
    """egp_physics_register with Python callables: the route for a backend that lives in Python (a `mujoco`-bindings
    adapter, a test double). The callables get NumPy views of the engine's host buffers and are invoked from the
    engine's worker threads (ctypes takes the GIL for each call), so this is for bring-up and tests, not speed.

        reset(env, qpos[nq], qvel[nv])      step(env, ctrl[nu])
        drain(env, qpos, qvel, qM or None, qfrc_bias, xpos or None)     fill the arrays in place
        inertia_epoch(env) -> int           optional, see egp_physics_vtable
    """

    def __init__(self, skel: Skeleton, n_env: int, reset, step, drain, inertia_epoch=None, name="python-callbacks"):
        self.lib = L.load()
        self.skel, self.n_env = skel, int(n_env)
        nq, nv, nu, nM, nb = skel.nq, skel.nv, skel.nu, skel.nM, len(skel.body_names)
        arr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)) if p else None
        self.errors = []

        def guard(fn):
            def call(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as exc:           # a raising callback must not unwind through C
                    self.errors.append(exc)
                    return 1
            return call

        self._reset = L.PHYS_RESET(lambda u, e, qp, qv: guard(reset)(e, arr(qp, nq), arr(qv, nv)))
        self._step = L.PHYS_STEP(lambda u, e, c: guard(step)(e, arr(c, nu)))
        self._drain = L.PHYS_DRAIN(lambda u, e, qp, qv, qM, b, xp: guard(drain)(
            e, arr(qp, nq), arr(qv, nv), arr(qM, nM), arr(b, nv), arr(xp, nb * 3).reshape(nb, 3) if xp else None))
        self._destroy = L.PHYS_DESTROY(lambda u: None)
        self._epoch = L.PHYS_EPOCH(lambda u, e: int(inertia_epoch(e))) if inertia_epoch is not None else L.PHYS_EPOCH()
        self._name = name.encode()
        vt = L.PhysicsVtable(None, self._reset, self._step, self._drain, self._destroy, self._name, self._epoch)
        self._vt = vt
        h = C.c_void_p()
        L.check(self.lib.egp_physics_register(C.byref(vt), self.n_env, C.byref(h)), "egp_physics_register")
        self.handle = h

    @property
    def name(self):
        return self.lib.egp_physics_name(self.handle).decode()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.egp_physics_destroy(self.handle)
            self.handle = None


class MujocoPhysics(SurrogatePhysics):
    """MuJoCo behind the physics boundary: the compiled plugin csrc/egp_physics_mujoco.cpp (one shared mjModel, one mjData per
    env, stepped by the engine's own threads -- no Python in the loop). The plugin is built separately
    (`MUJOCO_DIR=... python -m egopose_amd.build_mujoco`): MuJoCo is not in the build image. Same host accessors as the surrogate
    (reset / step / drain through `egp_physics_*_host`)."""

    PLUGIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libegopose_mujoco.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PLUGIN)

    def __init__(self, skel: Skeleton, n_env: int, mjcf_path: str, plugin: Optional[str] = None):
        """`plugin`: path of the compiled plugin (default: EGP_MUJOCO_PLUGIN, else egopose_amd/libegopose_mujoco.so)."""
        self.lib = L.load()
        plugin = plugin or os.environ.get("EGP_MUJOCO_PLUGIN") or self.PLUGIN
        if not os.path.exists(plugin):
            raise L.EgpError("the MuJoCo plugin %s is not built: MUJOCO_DIR=<mujoco tree> python -m egopose_amd.build_mujoco "
                             "(MuJoCo does not ship with this package)" % plugin)
        self.plugin = C.CDLL(plugin, mode=C.RTLD_GLOBAL)
        fn = self.plugin.egp_physics_create_mujoco
        fn.restype, fn.argtypes = C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p), C.c_char_p, C.c_int32]
        self.skel, self.n_env = skel, int(n_env)
        err = C.create_string_buffer(1024)
        h = C.c_void_p()
        rc = fn(os.fsencode(mjcf_path), self.n_env, C.byref(h), err, 1024)
        if rc != 0:
            raise L.EgpError("egp_physics_create_mujoco(%s) failed (%d): %s" % (mjcf_path, rc, err.value.decode("utf-8", "replace")))
        self.handle = h
        self._check_model(mjcf_path)

    def _check_model(self, mjcf_path):
        """The kernel context was built from egopose_amd.skeleton's reading of the MJCF: MuJoCo's own mjModel must agree on the
        dof tree and the sparse-inertia addressing (what mj_fullM walks), or K1 would expand qM wrongly."""
        fn = self.plugin.egp_mujoco_model_tables
        i32p = C.POINTER(C.c_int32)
        fn.restype = C.c_int
        fn.argtypes = [C.c_char_p] + [i32p] * 5 + [C.POINTER(C.c_double)] + [i32p] * 4 + [C.c_char_p, C.c_int32]
        sk = self.skel
        dims = [C.c_int32() for _ in range(5)]
        ts = C.c_double()
        par, madr = np.zeros(sk.nv, np.int32), np.zeros(sk.nv, np.int32)
        qadr, ndof = np.zeros(len(sk.body_names), np.int32), np.zeros(len(sk.body_names), np.int32)
        err = C.create_string_buffer(1024)
        rc = fn(os.fsencode(mjcf_path), *[C.byref(x) for x in dims], C.byref(ts), par.ctypes.data_as(i32p), madr.ctypes.data_as(i32p),
                qadr.ctypes.data_as(i32p), ndof.ctypes.data_as(i32p), err, 1024)
        if rc != 0:
            raise L.EgpError("egp_mujoco_model_tables failed: %s" % err.value.decode("utf-8", "replace"))
        got = tuple(x.value for x in dims)
        want = (sk.nq, sk.nv, sk.nu, len(sk.body_names), sk.nM)
        if got != want or not np.array_equal(par, sk.dof_parentid) or not np.array_equal(madr, sk.dof_Madr) \
                or not np.array_equal(qadr, sk.body_qpos_start) or abs(ts.value - sk.timestep) > 1e-15:
            raise L.EgpError("MuJoCo's model %s disagrees with the skeleton tables the kernels were built from "
                             "(dims %s vs %s)" % (mjcf_path, got, want))


def make_physics(skel, n_env, cfg=None):
    """The backend `EGP_PHYSICS` names: 'surrogate' (default; deterministic stand-in, physics parity unpinned) or 'mujoco' (the
    plugin over cfg.mujoco_model_file, humanoid_v1.py:15 -> mujoco_env.py:18-23)."""
    kind = os.environ.get("EGP_PHYSICS", "surrogate")
    if kind == "surrogate":
        return SurrogatePhysics(skel, n_env)
    if kind == "mujoco":
        path = getattr(cfg, "mujoco_model_file", None)
        if not path or not os.path.exists(path):
            raise IOError("File %s does not exist" % path)                  # envs/common/mujoco_env.py:19-20
        return MujocoPhysics(skel, n_env, path)
    raise ValueError("EGP_PHYSICS must be 'surrogate' or 'mujoco', got %r" % kind)


def available_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:                                              # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:                                          # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def _read(path):
    with open(path) as f:
        return f.read().strip()


def _parse_cpulist(text):
    out = set()
    for part in text.split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index, read=_read):
    """NUMA node the GPU hangs off (sysfs, through its PCI address), or None when that cannot be told."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(read("/sys/bus/pci/devices/%s/numa_node" % bdf))
        return node if node >= 0 else None
    except Exception:
        return None


def numa_physical_cpus(node, allowed=None, read=_read):
    """One hardware thread per physical core of NUMA `node`, restricted to `allowed` (default: the affinity mask).
    The physics threads spin: two of them on SMT siblings halve each other, and a thread on the other socket pays the
    inter-socket hop for every pinned row it trades with the GPU (measured: +3..7 % rollout rate, less run-to-run spread)."""
    try:
        cpus = _parse_cpulist(read("/sys/devices/system/node/node%d/cpulist" % node))
        if allowed is None:
            allowed = os.sched_getaffinity(0)
        cpus &= set(allowed)
        keep = set()
        for c in sorted(cpus):
            sib = _parse_cpulist(read("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c))
            if c == min(sib & cpus if sib & cpus else {c}):
                keep.add(c)
        return keep
    except Exception:
        return set()


def default_threads(share=1, device_index=None):
    """Engine threads spin inside the substep loop: never oversubscribe the CPU budget and leave room for the Python
    driver thread. ``share`` = processes splitting this host (ranks per node, rank r on GPU r). With a device index the
    budget is also bounded by the physical cores of the GPU's NUMA node divided by the ranks whose GPU sits on that node
    (the engine confines its threads to those cores, `RolloutEngine`)."""
    budget = available_cpus() // max(1, share)
    if device_index is not None and os.environ.get("EGP_PIN_NUMA", "1") != "0":
        node = gpu_numa_node(device_index)
        cores = numa_physical_cpus(node) if node is not None else set()
        if cores:
            same = sum(1 for r in range(max(1, share)) if gpu_numa_node(r) == node) or 1
            budget = min(budget, len(cores) // same)
    return max(1, min(budget - 2, 64))


def _pin_to_gpu_node(device_index, n_threads):
    """Narrow this thread's affinity to the physical cores of the GPU's NUMA node when they can hold n_threads + 1 threads (what
    `RolloutEngine` does around egp_engine_create). Returns (saved mask or None, sorted cores or None)."""
    if os.environ.get("EGP_PIN_NUMA", "1") == "0":
        return None, None
    node = gpu_numa_node(device_index)
    cores = numa_physical_cpus(node) if node is not None else set()
    if len(cores) < n_threads + 1:
        return None, None
    try:
        saved = os.sched_getaffinity(0)
        os.sched_setaffinity(0, cores)
        return saved, sorted(cores)
    except OSError:
        return None, None


def host_info(device_index=0):
    """What the box is, as far as sysfs / procfs tell: CPU model, frequency governor, the GPU's NUMA node, SMT, load average."""
    info = {"cpu_model": None, "cpu_governor": None, "gpu_numa_node": gpu_numa_node(device_index), "loadavg_1m": None, "smt": None}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        info["cpu_governor"] = _read("/sys/devices/system/cpu/cpu0/cpufreq/scaling_governor")
    except Exception:
        pass
    try:
        info["loadavg_1m"] = float(_read("/proc/loadavg").split()[0])
    except Exception:
        pass
    try:
        info["smt"] = _read("/sys/devices/system/cpu/smt/active") == "1"
    except Exception:
        pass
    return info


def host_probe(device_index=0, n_threads=0, millis=300):
    """egp_host_probe (include/egopose_hip.h) with the spinning threads confined as the engine's are: PCIe read rate of pinned state
    rows, go-word round trip, and whether spinning threads keep their cores. A dict of plain numbers."""
    lib = L.load()
    res = L.HostProbeResult()
    saved, cores = _pin_to_gpu_node(device_index, n_threads)
    try:
        L.check(lib.egp_host_probe(int(device_index), int(n_threads), int(millis), C.byref(res)), "egp_host_probe")
    finally:
        if saved is not None:
            os.sched_setaffinity(0, saved)
    out = {k: getattr(res, k) for k, _ in res._fields_}
    out["spin_cpus_pinned"] = len(cores) if cores else None
    return out


_RANKS_ON_HOST = None


def ranks_on_host():
    """Processes sharing this host's cores: torchrun's LOCAL_WORLD_SIZE; else what `dist.count_ranks_on_host()` found when the
    process group was set up (a collective every rank passes there -- this function only reads the cached value, it never
    communicates); a single-node WORLD_SIZE as the last resort; else 1."""
    v = os.environ.get("LOCAL_WORLD_SIZE")
    if v and v.isdigit() and int(v) > 0:
        return int(v)
    if _RANKS_ON_HOST is not None:
        return _RANKS_ON_HOST
    v = os.environ.get("WORLD_SIZE")
    if v and v.isdigit() and int(v) > 0:
        return int(v)
    return 1


class RolloutEngine:
    """egp_engine: n_env envs advance one env-step per step_async/wait pair."""

    def __init__(self, ctx, physics, n_env: int, n_threads: Optional[int] = None, n_groups: int = 1, device_dynamics: bool = False):
        import torch
        self.lib = L.load()
        self.ctx, self.physics = ctx, physics
        self.n_env = int(n_env)
        # (spinning threads: every rank of the host takes its share of the cores, not the whole machine)
        n_threads = default_threads(share=ranks_on_host(), device_index=ctx.device) if n_threads is None else int(n_threads)
        n_threads = max(int(n_groups), min(n_threads, self.n_env))
        if device_dynamics and not getattr(ctx, "_has_dynamics", False):
            ctx.set_dynamics_model()
        d = L.EngineDesc(self.n_env, n_threads, int(n_groups), 1 if device_dynamics else 0)
        self.device_dynamics = bool(device_dynamics)
        h = C.c_void_p()
        # the engine's threads inherit the creating thread's affinity mask: narrow it to the physical cores of the GPU's
        # NUMA node (when they can hold all of them) for the duration of the call only -- the caller's own mask, and with
        # it every thread pool it creates later (OpenMP, the oracle in the tests, the CPU baseline), stays as it was
        saved, self.pinned_cpus = _pin_to_gpu_node(ctx.device, n_threads)
        try:
            L.check(self.lib.egp_engine_create(ctx.handle, physics.handle, C.byref(d), C.byref(h)), "egp_engine_create")
        finally:
            if saved is not None:
                os.sched_setaffinity(0, saved)
        self.handle = h
        self.n_threads, self.n_groups = n_threads, int(n_groups)
        ptrs = [C.c_void_p() for _ in range(7)]
        L.check(self.lib.egp_engine_state(self.handle, *[C.byref(p) for p in ptrs]), "egp_engine_state")
        dev = torch.device("cuda", ctx.device)
        self.qpos = _wrap_device(ptrs[0].value, (self.n_env, ctx.nq), dev)
        self.qvel = _wrap_device(ptrs[1].value, (self.n_env, ctx.nv), dev)
        self.ee_wpos = _wrap_device(ptrs[2].value, (self.n_env, 15), dev)
        self.head_z = _wrap_host(ptrs[3].value, (self.n_env,))
        self.qpos_host = _wrap_host(ptrs[4].value, (self.n_env, ctx.nq))
        self.qvel_host = _wrap_host(ptrs[5].value, (self.n_env, ctx.nv))
        self.prev_qpos = _wrap_device(ptrs[6].value, (self.n_env, ctx.nq), dev)

    def group_range(self, g):
        a, b = C.c_int32(), C.c_int32()
        L.check(self.lib.egp_engine_group_range(self.handle, int(g), C.byref(a), C.byref(b)), "egp_engine_group_range")
        return a.value, b.value

    @property
    def substeps_per_launch(self):
        """frame_skip when the resident K1 serves a whole env-step per launch, else 1."""
        return int(self.lib.egp_engine_substeps_per_launch(self.handle))

    @property
    def envs_per_wave(self):
        """Envs a wavefront of the resident K1 serves in turn (1, 2 or 4; 0: per-substep form)."""
        return int(self.lib.egp_engine_envs_per_wave(self.handle, None))

    @property
    def resident_capacity(self):
        """Workgroups of the resident K1 the chip holds at once (the kernel's own residency probe at engine creation)."""
        cap = C.c_int32(0)
        self.lib.egp_engine_envs_per_wave(self.handle, C.byref(cap))
        return int(cap.value)

    def reset(self, env_ids, qpos, qvel):
        import torch
        ids = _np_i32(env_ids)
        qpos, qvel = _np_f64(qpos), _np_f64(qvel)
        if qpos.shape != (ids.shape[0], self.ctx.nq) or qvel.shape != (ids.shape[0], self.ctx.nv):
            raise ValueError("reset rows must be (n, nq) / (n, nv)")
        s = L.current_stream()
        L.check(self.lib.egp_engine_reset(self.handle, ids.ctypes.data, ids.shape[0], qpos.ctypes.data, qvel.ctypes.data, s),
                "egp_engine_reset")

    def step_async(self, group, action, active_host=None, ready_event=None):
        if not action.is_cuda or action.dtype.is_floating_point is False or tuple(action.shape) != (self.n_env, self.ctx.nu):
            raise ValueError("action must be a CUDA float64 tensor of shape (n_env, nu)")
        import torch
        if action.dtype != torch.float64 or not action.is_contiguous():
            raise ValueError("action must be contiguous float64")
        act = None if active_host is None else _np_i32(active_host)
        self._keep_active = act
        ev = C.c_void_p(ready_event.cuda_event) if ready_event is not None else C.c_void_p(0)
        L.check(self.lib.egp_engine_step_async(self.handle, int(group), C.c_void_p(action.data_ptr()),
                                               act.ctypes.data if act is not None else None, ev), "egp_engine_step_async")

    def wait(self, group):
        import torch
        s = L.current_stream()
        L.check(self.lib.egp_engine_wait(self.handle, int(group), s), "egp_engine_wait")

    def set_profile(self, on=True, every=1):
        """K1 timing by HIP events on the launch streams; ``every`` = N samples every Nth env-step."""
        L.check(self.lib.egp_engine_set_profile(self.handle, (max(1, int(every)) if on else 0)), "egp_engine_set_profile")

    def reset_timing(self):
        L.check(self.lib.egp_engine_reset_timing(self.handle), "egp_engine_reset_timing")

    def inertia_uploads(self):
        return int(self.lib.egp_engine_inertia_uploads(self.handle))

    def timing(self):
        p, w, k = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        L.check(self.lib.egp_engine_timing(self.handle, C.byref(p), C.byref(w), C.byref(k), C.byref(n)), "egp_engine_timing")
        return dict(phys_s=p.value, gpu_wait_s=w.value, k1_ms=k.value, k1_launches=n.value,
                    k1_env_substeps=int(self.lib.egp_engine_k1_env_substeps(self.handle)),
                    event_overhead_us=float(self.lib.egp_engine_event_overhead_ms(self.handle)) * 1e3)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.egp_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevArray:
    """__cuda_array_interface__ shim so torch can alias engine-owned HBM without copying."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _wrap_device(ptr, shape, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, shape), device=device)


def _wrap_host(ptr, shape):
    n = int(np.prod(shape))
    buf = (C.c_double * n).from_address(int(ptr))
    return np.frombuffer(buf, dtype=np.float64).reshape(shape)
