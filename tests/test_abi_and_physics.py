"""CPU-side checks: the C-ABI library loads and exports every symbol include/egopose_hip.h declares;
the host physics boundary (surrogate backend) behaves as the contract says. No GPU compute here."""
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "egopose_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(egp_[a-z0-9_]+)\s*\(", txt))
    return sorted(n for n in names)


def test_library_exports_every_declared_symbol():
    from egopose_amd import _lib as L
    lib = L.load()
    declared = _header_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), "missing export: %s" % name
        assert name in L.SIGNATURES, "ctypes signature missing for %s" % name
    assert b"gfx950" in lib.egp_version()


def test_ctypes_mirrors_have_the_library_s_struct_sizes():
    """every descriptor struct the Python side mirrors must have the size the library was compiled with -- a field
    added on one side only would otherwise show up as garbage arguments, not as an error"""
    from egopose_amd import _lib
    L = _lib.load()
    mirrors = {"egp_model_desc": _lib.ModelDesc, "egp_expert_table": _lib.ExpertTable,
               "egp_gemm_desc": _lib.GemmDesc, "egp_dynamics_desc": _lib.DynamicsDesc,
               "egp_mlp_layer": _lib.MlpLayer, "egp_physics_vtable": _lib.PhysicsVtable,
               "egp_surrogate_desc": _lib.SurrogateDesc, "egp_engine_desc": _lib.EngineDesc,
               "egp_rollout_tick": _lib.RolloutTick, "egp_ppo_loss_desc": _lib.PpoLossDesc,
               "egp_adam_segment": _lib.AdamSegment, "egp_host_probe_result": _lib.HostProbeResult}
    header = open(os.path.join(REPO, "include", "egopose_hip.h")).read()
    declared = set(re.findall(r"^\} (egp_[a-z_]+);", header, flags=re.M))
    assert declared == set(mirrors), "a struct in include/egopose_hip.h has no ctypes mirror (or the reverse)"
    import ctypes
    for name, cls in mirrors.items():
        assert L.egp_abi_sizeof(name.encode()) == ctypes.sizeof(cls), name
    assert L.egp_abi_sizeof(b"no_such_struct") == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from egopose_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.EgpError, match="no CPU fallback"):
        L.load()


def test_skeleton_matches_reference_layout(skel):
    # qpos map of humanoid_1205_v1
    assert (skel.nq, skel.nv, skel.nu, len(skel.body_names), skel.nM) == (59, 58, 52, 21, 910)
    addr = skel.body_qposaddr()
    assert addr["Hips"] == (0, 7) and addr["RightForeArm"] == (31, 32) and addr["LeftLeg"] == (55, 56)
    assert addr["LeftFoot"] == (56, 59)
    assert list(skel.ee_body) == [20, 17, 14, 10, 6]
    assert skel.actuator_names[24] == "RightForeArm_z" and len(skel.actuator_names) == 52
    g = load_golden("pd_torque.npz")
    np.testing.assert_array_equal(skel.full_from_sparse(g["qM"][0]), g["M"][0])


def test_surrogate_fk_matches_python_fk(skel):
    from egopose_amd.physics import SurrogatePhysics
    g = load_golden("body_quat_obs.npz")
    ph = SurrogatePhysics(skel, 4)
    assert ph.name.startswith("surrogate")
    for i in range(4):
        ph.reset(i, g["qpos"][i], g["qvel"][i])
        qpos, qvel, qM, bias, xpos = ph.drain(i)
        np.testing.assert_array_equal(qpos, g["qpos"][i])
        np.testing.assert_array_equal(qvel, g["qvel"][i])
        np.testing.assert_array_equal(qM, ph.qM0)
        np.testing.assert_allclose(xpos, skel.body_xpos(g["qpos"][i]), rtol=1e-12, atol=1e-12)
    ph.close()


def test_surrogate_step_is_deterministic_and_stale_bias(skel):
    from egopose_amd.physics import SurrogatePhysics
    g = load_golden("pd_torque.npz")
    ph = SurrogatePhysics(skel, 2)
    for e in range(2):
        ph.reset(e, g["qpos"][0], g["qvel"][0])
    ctrl = g["torque_clipped"][0]
    for _ in range(5):
        ph.step(0, ctrl)
        ph.step(1, ctrl)
    a, b = ph.drain(0), ph.drain(1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    # numpy restatement of one surrogate step: bias is evaluated BEFORE integration (mj_step order)
    ph.reset(0, g["qpos"][1], g["qvel"][1])
    q0, v0 = g["qpos"][1].copy(), g["qvel"][1].copy()
    C = np.zeros(58)
    C[2] = 200.0 * v0[2]
    C[6:] = 1.0 * v0[6:]
    f = np.concatenate([np.zeros(6), ctrl]) - C
    acc = ph.Minv0 @ f
    v1 = v0 + skel.timestep * acc
    ph.step(0, ctrl)
    qpos, qvel, qM, bias, _ = ph.drain(0, want_xpos=False)
    np.testing.assert_allclose(qvel, v1, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(bias, C, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(qpos[7:], q0[7:] + skel.timestep * v1[6:], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(qpos[3:7]), 1.0, atol=1e-14)
    with pytest.raises(ValueError):
        ph.step(5, ctrl)
    ph.close()


def test_surrogate_minimum_substep_cost_changes_time_only(skel, monkeypatch):
    """EGP_SURROGATE_SUBSTEP_US (bench.py's `simulator_cost_per_substep` leg): every step takes at least that long, the
    backend's name says so, and the numbers are the ones the plain surrogate produces."""
    import time
    from egopose_amd.physics import SurrogatePhysics
    g = load_golden("pd_torque.npz")
    ctrl = g["torque_clipped"][0]
    plain = SurrogatePhysics(skel, 1)
    monkeypatch.setenv("EGP_SURROGATE_SUBSTEP_US", "200")
    slow = SurrogatePhysics(skel, 1)
    monkeypatch.delenv("EGP_SURROGATE_SUBSTEP_US")
    assert plain.name == "surrogate-euler-M0" and slow.name == "surrogate-euler-M0+200us-per-substep"
    for ph in (plain, slow):
        ph.reset(0, g["qpos"][0], g["qvel"][0])
    t0 = time.perf_counter()
    for _ in range(50):
        slow.step(0, ctrl)
    assert time.perf_counter() - t0 >= 50 * 200e-6
    for _ in range(50):
        plain.step(0, ctrl)
    for x, y in zip(plain.drain(0), slow.drain(0)):
        np.testing.assert_array_equal(x, y)
    plain.close()
    slow.close()


def test_callback_backend_round_trip(skel):
    """egp_physics_register with Python callables: reset/step/drain reach the callables with views of the caller's
    buffers, the epoch callback is wired, and a raising callable turns into an error code instead of unwinding."""
    from conftest import VaryingInertiaBackend
    from egopose_amd import _lib as L
    g = load_golden("body_quat_obs.npz")
    be = VaryingInertiaBackend(skel, 2)
    lib = L.load()
    q0, v0 = np.ascontiguousarray(g["qpos"][0]), np.ascontiguousarray(g["qvel"][0] * 0.1)
    L.check(lib.egp_physics_reset_host(be.handle, 1, q0.ctypes.data, v0.ctypes.data), "reset")
    assert be.physics.name == "varying-inertia" and lib.egp_physics_n_env(be.handle) == 2
    ctrl = np.linspace(-1, 1, skel.nu)
    for k in range(3):
        L.check(lib.egp_physics_step_host(be.handle, 1, ctrl.ctypes.data), "step")
    assert len(be.torques[1]) == 3 and np.array_equal(be.torques[1][0], ctrl) and be.k[1] == 3
    qpos, qvel, qM, bias = np.empty(skel.nq), np.empty(skel.nv), np.empty(skel.nM), np.empty(skel.nv)
    L.check(lib.egp_physics_drain_host(be.handle, 1, qpos.ctypes.data, qvel.ctypes.data, qM.ctypes.data, bias.ctypes.data, None), "drain")
    rq, rv, rM, rb, _ = be.inner.drain(1, want_xpos=False)
    np.testing.assert_array_equal(qpos, rq)
    np.testing.assert_array_equal(qM, rM * be.scale(1, 3))
    # a callable that raises -> nonzero status, exception kept on the wrapper
    rc = lib.egp_physics_step_host(be.handle, 7, ctrl.ctypes.data)      # env 7 does not exist in the inner backend
    assert rc != 0
    be.close()
