"""Host-side logic on CPU: Config vs the reference's parsed arrays, the compat import surface the unmodified
driver needs, ZFilter host object vs golden, synthetic motion generator, and the oracle CPU sampler
(cpu_baseline leg) end to end on a tiny dataset."""
import os
import pickle
import subprocess
import sys
import types

import numpy as np
import pytest
import torch
import yaml

from conftest import REPO, load_golden


def _workspace(tmp_path, skel, n_takes=2, n_frames=90):
    """A tiny dataset in the reference's formats built with ORACLE math (no GPU here)."""
    from egopose_amd.config import _ASSET_CFG
    from egopose_amd.physics import SurrogatePhysics
    from egopose_amd.synthetic import synth_qpos_sequence
    from oracle import humanoid as H, quat as Q
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "config", "egomimic"))
    os.makedirs(os.path.join(root, "datasets", "meta"))
    os.makedirs(os.path.join(root, "datasets", "features"))
    import shutil
    shutil.copy(os.path.join(_ASSET_CFG, "subject_03.yml"), os.path.join(root, "config", "egomimic", "subject_03.yml"))
    rng = np.random.RandomState(3)
    ph = SurrogatePhysics(skel, 1)
    dt = skel.timestep * 15
    experts, feats, names = {}, {}, ["t%d" % i for i in range(n_takes)]
    for name in names:
        q = synth_qpos_sequence(skel, rng, n_frames)
        xpos = []
        for row in q:
            ph.reset(0, row, np.zeros(58))
            xpos.append(ph.drain(0)[4])
        xpos = np.stack(xpos)
        bq = H.body_quat(q, skel.body_qpos_start, skel.body_ndof)
        qv = H.qvel_fd(q[:-1], q[1:], dt)
        qv = np.vstack([qv[:1], qv])
        experts[name] = dict(
            qpos=q, qvel=qv, rlinv_local=np.vstack([Q.transform_vec(qv[1:2, :3], q[1:2, 3:7], "heading"), Q.transform_vec(qv[1:, :3], q[1:, 3:7], "heading")]),
            rangv=qv[:, 3:6].copy(), rq_rmh=Q.de_heading(q[:, 3:7]), ee_pos=H.ee_pos(q, xpos[:, skel.ee_body].reshape(len(q), 15)),
            bquat=bq, bangvel=np.vstack([H.angvel_fd(bq[:1], bq[1:2], dt), H.angvel_fd(bq[:-1], bq[1:], dt)]),
            head_pos=xpos[:, 6], len=len(q), height_lb=q[:, 2].min(), head_height_lb=xpos[:, 6, 2].min())
        feats[name] = rng.normal(size=(n_frames, 16))
    ph.close()
    yaml.safe_dump({"train": names, "test": names[:1]}, open(os.path.join(root, "datasets", "meta", "meta_subject_03.yml"), "w"))
    pickle.dump(experts, open(os.path.join(root, "datasets", "features", "expert_subject_03.p"), "wb"))
    pickle.dump((feats, {}), open(os.path.join(root, "datasets", "features", "cnn_feat_subject_03.p"), "wb"))
    return root


def test_config_matches_reference_parse(tmp_path, skel, monkeypatch):
    from egopose_amd.config import Config
    root = _workspace(tmp_path, skel)
    monkeypatch.chdir(root)
    cfg = Config("subject_03", create_dirs=True)
    g = load_golden("config_subject_03.npz")
    for k in ("jkp", "jkd", "a_ref", "a_scale", "torque_lim", "b_diffw"):
        np.testing.assert_array_equal(getattr(cfg, k), g[k])
    for k in ("gamma", "tau", "clip_epsilon", "log_std", "min_batch_size", "num_optim_epoch", "env_episode_len", "fr_margin",
              "policy_lr", "value_lr", "policy_v_hdim"):
        assert getattr(cfg, k) == g[k]
    assert bool(cfg.fix_std) == bool(g["fix_std"]) and list(cfg.policy_hsize) == list(g["policy_hsize"])
    cfg.update_adaptive_params(0)
    assert (cfg.adp_noise_rate, cfg.adp_log_std, cfg.adp_policy_lr) == (float(g["adp_noise_rate"]), float(g["adp_log_std"]), float(g["adp_policy_lr"]))
    ws = dict(zip([str(x) for x in g["reward_keys"]], g["reward_vals"]))
    assert {k: float(v) for k, v in cfg.reward_weights.items()} == {k: float(v) for k, v in ws.items()}
    assert cfg.model_dir == "results/egomimic/subject_03/models" and os.path.isdir(cfg.log_dir)
    assert cfg.expert_feat_file == "datasets/features/expert_subject_03.p" and cfg.takes["train"] == ["t0", "t1"]
    # piecewise-linear schedules (egomimic_config.py:124-131)
    cfg2 = Config("x", cfg_dict=dict(meta_id="meta_subject_03", mujoco_model="m", vis_model="v", adp_iter_cp=[0, 10, 30],
                                     adp_noise_rate_cp=[1.0, 0.5], adp_log_std_cp=[-2.0, -3.0, -4.0]))
    cfg2.update_adaptive_params(5)
    assert cfg2.adp_noise_rate == pytest.approx(0.75) and cfg2.adp_log_std == pytest.approx(-2.5)
    cfg2.update_adaptive_params(40)
    assert cfg2.adp_noise_rate == pytest.approx(0.5) and cfg2.adp_log_std == pytest.approx(-4.0)
    with pytest.raises(SystemExit):
        Config("does_not_exist")


REF_DRIVER = "/root/reference/ego_pose/ego_mimic.py"


def _driver_env():
    return dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.path.join(REPO, "egopose_amd", "compat"), HIP_VISIBLE_DEVICES="")


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference tree only exists in the build container")
def test_unmodified_reference_driver_runs_on_the_compat_packages(tmp_path, skel):
    """Row (b): `python <reference>/ego_pose/ego_mimic.py --cfg subject_03` executed AS IS (the file under /root/reference,
    not a restatement) with egopose_amd/compat on PYTHONPATH and a config whose max_iter_num is 0: the whole module-level
    set-up (imports, Config, env + experts, nets, optimizers, AgentEgo(dtype=float64, ...)) and an empty main_loop run
    through this package's classes. No MI355X in the build container, so the driver picks its CPU device and nothing
    is sampled; the iterations themselves are covered on the GPU by tests/test_dropin_gpu.py."""
    root = _workspace(tmp_path, skel)
    cfg_path = os.path.join(root, "config", "egomimic", "subject_03.yml")
    y = yaml.safe_load(open(cfg_path))
    y["max_iter_num"] = 0
    yaml.safe_dump(y, open(cfg_path, "w"))
    out = subprocess.run([sys.executable, REF_DRIVER, "--cfg", "subject_03", "--num-threads", "2"], cwd=root, env=_driver_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    log = open(os.path.join(root, "results/egomimic/subject_03/log/log.txt")).read()
    assert "training done!" in log
    # every module the driver imported came from this package's mirrors, none from the reference tree
    probe = ("import runpy, sys; sys.argv = [%r, '--cfg', 'subject_03']; runpy.run_path(%r, run_name='__main__'); "
             "mods = ['utils', 'core.policy_gaussian', 'core.critic', 'models.mlp', 'models.video_state_net', "
             "'ego_pose.envs.humanoid_v1', 'ego_pose.core.agent_ego', 'ego_pose.utils.egomimic_config', "
             "'ego_pose.core.reward_function']; "
             "print('ORIGINS', [sys.modules[m].__file__ for m in mods])" % (REF_DRIVER, REF_DRIVER))
    out = subprocess.run([sys.executable, "-c", probe], cwd=root, env=_driver_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    origins = eval(out.stdout.split("ORIGINS", 1)[1].strip().splitlines()[0])
    assert len(origins) == 9 and all(o.startswith(os.path.join(REPO, "egopose_amd", "compat")) for o in origins), origins


REF_FORECAST_DRIVER = "/root/reference/ego_pose/ego_forecast.py"


@pytest.mark.skipif(not os.path.exists(REF_FORECAST_DRIVER), reason="the reference tree only exists in the build container")
def test_unmodified_forecast_driver_runs_on_the_compat_packages(tmp_path, skel, monkeypatch):
    """Row f2 through row (b): `python <reference>/ego_pose/ego_forecast.py --cfg subject_03` executed AS IS on egopose_amd/compat
    with max_iter_num 0 -- its imports (models.video_forecast_net, ego_pose.utils.egoforecast_config + egomimic_config), the warm
    start from an ego_mimic checkpoint written in the reference's container (plain pickle.load, filter_state_dict of the first
    affine layer because policy_s_net is 'lstm', load_state_dict(strict=False); ego_forecast.py:60-68), the two VideoForecastNets,
    optimizers, AgentEgo and an empty main_loop. The iterations themselves run on the GPU (tests/test_rollout_gpu.py forecast tests)."""
    from egopose_amd.config import Config, _ASSET_CFG
    from egopose_amd.train import Trainer
    root = _workspace(tmp_path, skel, n_frames=150)
    monkeypatch.chdir(root)
    # the ego_mimic checkpoint the forecast driver starts from (iter 1: a checkpoint of this package, in the reference's format)
    tr = Trainer(Config("subject_03", create_dirs=True), torch.device("cpu"), torch.float64, num_envs=2, num_threads=1, num_groups=1)
    tr.save("results/egomimic/subject_03/models/iter_0001.p")
    tr.close()
    os.makedirs(os.path.join(root, "config", "egoforecast"))
    y = yaml.safe_load(open(os.path.join(os.path.dirname(_ASSET_CFG), "egoforecast", "subject_03.yml")))
    y["max_iter_num"], y["ego_mimic_iter"] = 0, 1
    yaml.safe_dump(y, open(os.path.join(root, "config", "egoforecast", "subject_03.yml"), "w"))
    out = subprocess.run([sys.executable, REF_FORECAST_DRIVER, "--cfg", "subject_03", "--num-threads", "2"], cwd=root, env=_driver_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    log = open(os.path.join(root, "results/egoforecast/subject_03/log/log.txt")).read()
    assert "loading model from ego mimic checkpoint: results/egomimic/subject_03/models/iter_0001.p" in log and "training done!" in log
    probe = ("import runpy, sys; sys.argv = [%r, '--cfg', 'subject_03']; ns = runpy.run_path(%r, run_name='__main__'); "
             "mods = ['models.video_forecast_net', 'ego_pose.utils.egoforecast_config', 'ego_pose.utils.egomimic_config', "
             "'ego_pose.core.agent_ego', 'ego_pose.envs.humanoid_v1']; "
             "print('ORIGINS', [sys.modules[m].__file__ for m in mods]); "
             "print('SHAPES', [ns['policy_vs_net'].out_dim, ns['policy_net'].net.affine_layers[0].in_features, "
             "type(ns['policy_vs_net']).__module__, ns['cfg'].env_episode_len, ns['cfg'].fr_margin])" % (REF_FORECAST_DRIVER, REF_FORECAST_DRIVER))
    out = subprocess.run([sys.executable, "-c", probe], cwd=root, env=_driver_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    origins = eval(out.stdout.split("ORIGINS", 1)[1].strip().splitlines()[0])
    assert len(origins) == 5 and all(o.startswith(os.path.join(REPO, "egopose_amd", "compat")) for o in origins), origins
    shapes = eval(out.stdout.split("SHAPES", 1)[1].strip().splitlines()[0])
    assert shapes[0] == shapes[1] == 128 + 128 and shapes[2] == "egopose_amd.nets" and shapes[3:] == [90, 30], shapes   # v_hdim + s_hdim (lstm state net)


def test_compat_packages_expose_the_driver_surface(tmp_path, skel):
    """The names ego_pose/ego_mimic.py:8-16 imports and the calls it makes on them (:29-99), resolved through
    egopose_amd/compat -- runs everywhere (the test above needs the reference tree). Sampling without an MI355X must
    refuse loudly."""
    root = _workspace(tmp_path, skel)
    code = r"""
import os
import utils as U
for name in ("torch", "np", "Logger", "create_logger", "ZFilter", "to_device", "to_cpu", "set_optimizer_lr"):
    assert hasattr(U, name), name
from utils import *
from core.policy_gaussian import PolicyGaussian
from core.critic import Value
from models.mlp import MLP
from models.video_state_net import VideoStateNet
from ego_pose.envs.humanoid_v1 import HumanoidEnv
from ego_pose.core.agent_ego import AgentEgo
from ego_pose.utils.egomimic_config import Config
from ego_pose.core.reward_function import reward_func
torch.set_default_dtype(torch.float64)
cfg = Config("subject_03", create_dirs=True)
env = HumanoidEnv(cfg)
env.seed(cfg.seed)
env.load_experts(cfg.takes["train"], cfg.expert_feat_file, cfg.cnn_feat_file)
assert len(env.model.actuator_names) == 52 and (env.observation_space.shape[0], env.action_space.shape[0]) == (115, 52)
vs = [VideoStateNet(env.cnn_feat[0].shape[-1], cfg.policy_v_hdim, cfg.fr_margin, cfg.policy_v_net, cfg.policy_v_net_param, cfg.causal) for _ in range(2)]
pol = PolicyGaussian(MLP(115 + cfg.policy_v_hdim, cfg.policy_hsize, cfg.policy_htype), 52, log_std=cfg.log_std, fix_std=cfg.fix_std)
val = Value(MLP(115 + cfg.value_v_hdim, cfg.value_hsize, cfg.value_htype))
pp, vp = list(pol.parameters()) + list(vs[0].parameters()), list(val.parameters()) + list(vs[1].parameters())
agent = AgentEgo(env=env, dtype=torch.float64, device=torch.device("cpu"), running_state=ZFilter((115,), clip=5),
                 custom_reward=reward_func[cfg.reward_id], mean_action=False, render=False, num_threads=2, policy_net=pol,
                 policy_vs_net=vs[0], value_net=val, value_vs_net=vs[1], optimizer_policy=torch.optim.Adam(pp, lr=cfg.policy_lr),
                 optimizer_value=torch.optim.Adam(vp, lr=cfg.value_lr), opt_num_epochs=cfg.num_optim_epoch, gamma=cfg.gamma,
                 tau=cfg.tau, clip_epsilon=cfg.clip_epsilon, policy_grad_clip=[(pp, 40)])
Logger(cfg.tb_dir).scalar_summary("total_reward", 0.5, 0)
with to_cpu(pol, val):
    sd = pol.state_dict()
assert "net.affine_layers.0.weight" in sd and "action_log_std" in sd
try:
    agent.sample(100)
except RuntimeError as e:
    assert "no CPU fallback" in str(e)
else:
    raise SystemExit("sampling on CPU must fail")
print("DRIVER_SURFACE_OK")
"""
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=_driver_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "DRIVER_SURFACE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert os.path.exists(os.path.join(root, "results/egomimic/subject_03/tb/scalars.jsonl"))


def test_zfilter_host_object_and_device_bridge():
    from egopose_amd.zfilter import ZFilter
    g = load_golden("zfilter.npz")
    zf = ZFilter((115,), clip=5)
    Y = np.stack([zf(x) for x in g["X"]])
    np.testing.assert_allclose(Y, g["Y"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(zf.rs.std, g["std"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(np.stack([zf(x, update=False) for x in g["X"][:16]]), g["Yfrozen"], rtol=1e-11, atol=1e-11)
    st = zf.to_device_state("cpu")
    assert st.shape == (231,) and st[0] == 300
    z2 = ZFilter((115,), clip=5)
    z2.from_device_state(st)
    np.testing.assert_array_equal(z2.rs.mean, zf.rs.mean)
    z3 = pickle.loads(pickle.dumps(zf))               # checkpoints carry the filter object
    np.testing.assert_array_equal(z3.rs.std, zf.rs.std)
    with pytest.raises(AssertionError):
        zf.rs.push(np.zeros(3))


def test_synthetic_motion_is_smooth_and_in_range(skel):
    from egopose_amd.synthetic import synth_qpos_sequence
    q = synth_qpos_sequence(skel, np.random.RandomState(0), 120)
    assert q.shape == (120, 59)
    np.testing.assert_allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0, atol=1e-12)
    assert (q[:, 7:] >= skel.joint_range[:, 0] - 1e-12).all() and (q[:, 7:] <= skel.joint_range[:, 1] + 1e-12).all()
    assert np.abs(np.diff(q[:, 7:], axis=0)).max() < 0.2 and (q[:, 32:35] == 0).all() and (q[:, 42:45] == 0).all()


def test_oracle_cpu_sampler_runs_the_reference_structure(tmp_path, skel):
    """bench.py's cpu_baseline leg: 2 forked workers, batch-1 float64 policy, numpy PD/reward, surrogate physics."""
    root = _workspace(tmp_path, skel, n_takes=2, n_frames=80)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    # shrink the episode so the test takes seconds: patch the YAML copy
    p = os.path.join(root, "config", "egomimic", "subject_03.yml")
    cfg = yaml.safe_load(open(p))
    cfg["env_episode_len"] = 8
    cfg["fr_margin"] = 2
    yaml.safe_dump(cfg, open(p, "w"))
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_env", "--dataset", root, "--threads", "2", "--steps", "30"],
                         cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["env_steps"] >= 30 and r["threads"] == 2 and r["episodes"] >= 4 and 0 < r["avg_c_reward"] < 5
    assert r["physics"].startswith("surrogate")


def test_eval_metrics_and_helpers_match_reference():
    """Product-side eval metrics (egopose_amd.metrics) == ego_pose/utils/metrics.py, align_human_state ==
    utils/tools.py:71-75, VideoRegNet(no_cnn) == models/video_reg_net.py on the reference's weights."""
    import torch
    from egopose_amd import metrics as M
    from egopose_amd.nets import VideoRegNet
    g = load_golden("metrics.npz")
    dt = float(g["dt"])
    np.testing.assert_allclose(M.get_joint_angles(g["traj"]), g["angles"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(M.get_joint_vels(g["traj"], dt), g["vels"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(M.get_joint_accels(g["vels"], dt), g["accels"], rtol=1e-12, atol=1e-12)
    assert M.get_mean_dist(M.get_joint_angles(g["traj"]), M.get_joint_angles(g["traj2"])) == pytest.approx(float(g["mean_dist"]), rel=1e-12)
    assert M.get_mean_abs(g["accels"]) == pytest.approx(float(g["mean_abs"]), rel=1e-12)
    res = {"traj_pred": {"a": g["traj2"], "b": g["traj"]}, "traj_orig": {"a": g["traj"], "b": g["traj"]}}
    out = M.compute_metrics(res, dt)
    assert out["pose_dist"] == pytest.approx(float(g["mean_dist"]) / 2, rel=1e-12) and out["per_take"]["b"][0] == 0.0
    assert out["accels"] > 0 and np.isfinite(out["vel_dist"])
    noisy = {"traj_pred": {"a": g["traj"].copy()}}
    M.remove_noisy_hands(noisy)
    assert (noisy["traj_pred"]["a"][:, 32:35] == 0).all() and (noisy["traj_pred"]["a"][:, 42:45] == 0).all()
    assert (noisy["traj_pred"]["a"][:, 35:42] == g["traj"][:, 35:42]).all()
    e = load_golden("eval_tools.npz")
    for i in range(e["qpos"].shape[0]):
        q, v = e["qpos"][i].copy(), e["qvel"][i].copy()
        M.align_human_state(q, v, e["ref_qpos"][i])
        np.testing.assert_allclose(q, e["out_qpos"][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(v, e["out_qvel"][i], rtol=1e-12, atol=1e-12)
    net = VideoRegNet(9, 32, 16, no_cnn=True, mlp_dim=(24, 12)).double()
    net.load_state_dict({k[3:]: torch.as_tensor(e[k]) for k in e.files if k.startswith("sn_") and k not in ("sn_x", "sn_y")})
    with torch.no_grad():
        y = net(torch.as_tensor(e["sn_x"])).numpy()
    np.testing.assert_allclose(y, e["sn_y"], rtol=1e-10, atol=1e-12)
    with pytest.raises(NotImplementedError):
        VideoRegNet(9, 32, 16, no_cnn=False, cnn_type="mobile")


def test_statereg_dataset_config_and_nets(tmp_path, monkeypatch):
    """state_reg data path == ego_pose/utils/statereg_dataset.py on the same files (normalised trajectories, chunking with
    overlap, take order, frame offsets); config loads; ResNet-18 has torchvision's parameter set; one CPU training step."""
    from egopose_amd import statereg as SR
    from egopose_amd.nets import ResNet18
    g = load_golden("statereg_dataset.npz")
    monkeypatch.chdir(tmp_path)
    names = ["tk_a", "tk_b", "tk_c"]
    for d in ("datasets/traj", "datasets/fpv_of", "datasets/meta"):
        os.makedirs(d)
    msync = {n: [int(v) for v in g["msync"][i]] for i, n in enumerate(names)}
    for i, n in enumerate(names):
        tr = g["traj_" + n]
        with open("datasets/traj/%s_traj.p" % n, "wb") as f:
            np.save(f, tr)
        os.makedirs("datasets/fpv_of/%s" % n)
        for k in range(tr.shape[0] + msync[n][0] + 2):
            np.save("datasets/fpv_of/%s/%05d.npy" % (n, k), np.full((2, 2, 2), float(k) + 1000 * i))
    yaml.safe_dump({"train": ["tk_a", "tk_b"], "test": ["tk_c"], "video_mocap_sync": msync, "capture": {"fps": 30}},
                   open("datasets/meta/meta_sr_test.yml", "w"))
    ds = SR.Dataset("meta_sr_test", "train", 16, "iter", False, 6, 100)
    assert ds.traj_dim == int(g["traj_dim"]) and ds.len == int(g["length"])
    np.testing.assert_allclose(ds.mean, g["mean"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ds.std, g["std"], rtol=1e-10, atol=1e-12)
    of_ids, norm, orig, lens = [], [], [], []
    for of, nt, ot in ds:
        of_ids.append(of[:, 0, 0, 0]); norm.append(nt); orig.append(ot); lens.append(len(of))
    assert lens == list(g["chunk_len"]) and len(lens) == int(g["n_chunks"])
    np.testing.assert_array_equal(np.concatenate(of_ids), g["of_ids"])
    # (columns with ~zero spread -- the zeroed wrist joints -- divide rounding noise by std + 1e-8)
    np.testing.assert_allclose(np.vstack(norm), g["norm"], rtol=1e-9, atol=1e-6)
    np.testing.assert_array_equal(np.vstack(orig), g["orig"])
    dt_ = SR.Dataset("meta_sr_test", "test", 16, "iter", False, 6, 100)
    dt_.set_mean_std(ds.mean, ds.std)
    np.testing.assert_allclose(np.vstack([nt for _, nt, _ in dt_]), g["test_norm"], rtol=1e-9, atol=1e-6)
    # the packaged config and the published parameter count of ResNet-18 without its classifier
    cfg = SR.StateRegConfig("subject_03")
    assert (cfg.fr_num, cfg.v_hdim, cfg.cnn_fdim, cfg.fr_margin, cfg.cnn_type) == (120, 128, 128, 10, "resnet")
    rn = ResNet18(128)
    assert sum(p.numel() for n_, p in rn.named_parameters() if not n_.startswith("fc.")) == 11176512
    assert {"conv1.weight", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight", "layer4.1.bn2.bias", "fc.weight"} <= set(dict(rn.named_parameters()))
    # integrating a regressor output that equals the ground truth reproduces height, joints and (to first order) the path
    traj = ds.orig_trajs[0][3:40]
    sp = ds.trajs[0][3:40]
    rec = SR.get_traj_from_state_pred(sp, traj[0, :2], SR.M._heading_q(traj[0, 3:7]), ds.dt, ds.traj_dim)
    np.testing.assert_allclose(rec[:, 2], traj[:, 2], atol=1e-12)
    np.testing.assert_allclose(rec[:, 7:], traj[:, 7:], atol=1e-12)
    assert np.abs(rec[:, :2] - traj[:, :2]).max() < 0.02 and np.abs(np.abs((rec[:, 3:7] * traj[:, 3:7]).sum(1)) - 1).max() < 1e-3
    # one optimisation step on the CPU with a tiny frame size
    cfg.fr_margin, cfg.mlp_dim, cfg.v_hdim, cfg.cnn_fdim = 3, [16, 8], 16, 8
    for n in names[:2]:
        for k in range(g["traj_" + n].shape[0] + msync[n][0] + 2):
            np.save("datasets/fpv_of/%s/%05d.npy" % (n, k), np.random.RandomState(k).normal(size=(32, 32, 2)))
    ds2 = SR.Dataset("meta_sr_test", "train", 16, "iter", False, 6, 100)
    tr = SR.StateRegTrainer(cfg, ds2, "cpu", torch.float32, frame_shape=(3, 32, 32))
    before = tr.net.cnn.resnet.conv1.weight.detach().clone()
    loss, n_s, _ = tr.train_epoch()
    assert np.isfinite(loss) and n_s > 0 and not torch.equal(before, tr.net.cnn.resnet.conv1.weight)
    res, meta = tr.test()
    assert set(res["traj_pred"]) == {"tk_a", "tk_b"} and res["traj_pred"]["tk_a"].shape[1] == 59 and meta["algo"] == "state_reg"
    tr.save(str(tmp_path / "inf.p"), inference=True)
    cp, m2 = pickle.load(open(tmp_path / "inf.p", "rb"))
    assert "cfg" in m2 and not any(k.startswith("cnn.") for k in cp["state_net_dict"])


def test_numa_core_selection_and_thread_budget(monkeypatch):
    """physics.numa_physical_cpus / default_threads on a fake two-socket, SMT-2 topology: one hardware thread per core of
    the GPU's node, inside the affinity mask; the thread budget is the node's cores split over the ranks on that node."""
    from egopose_amd import physics as P
    files = {"/sys/devices/system/node/node0/cpulist": "0-7,16-23", "/sys/devices/system/node/node1/cpulist": "8-15,24-31"}
    for c in range(32):
        sib = "%d,%d" % (c % 16, c % 16 + 16)
        files["/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c] = sib

    def read(path):
        return files[path]
    assert P._parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    assert P.numa_physical_cpus(0, allowed=range(32), read=read) == set(range(8))
    assert P.numa_physical_cpus(1, allowed=range(32), read=read) == set(range(8, 16))
    assert P.numa_physical_cpus(1, allowed=[9, 10, 25, 27], read=read) == {9, 10, 27}       # sibling 25 of 9 dropped, 27 alone kept
    assert P.numa_physical_cpus(3, allowed=range(32), read=read) == set()                     # unknown node
    # thread budget: 4 ranks, GPUs 0,1 on node 0 and 2,3 on node 1; 8 cores per node; no quota
    monkeypatch.setattr(P, "available_cpus", lambda: 32)
    monkeypatch.setattr(P, "gpu_numa_node", lambda dev, read=None: dev // 2)
    monkeypatch.setattr(P, "numa_physical_cpus", lambda node, allowed=None, read=None: set(range(8 * node, 8 * node + 8)))
    assert P.default_threads(share=4, device_index=2) == 8 // 2 - 2
    assert P.default_threads(share=1, device_index=0) == 8 - 2
    assert P.default_threads(share=4) == 32 // 4 - 2                                          # no device: the old rule
    monkeypatch.setenv("EGP_PIN_NUMA", "0")
    assert P.default_threads(share=4, device_index=2) == 32 // 4 - 2


def test_surrogate_matvec_variants_are_bit_identical():
    """EGP_SURROGATE_SIMD = plain / avx2 / avx512 (register-blocked Minv0 * f in the surrogate's step): the same fused
    multiply-adds in the same order, so the trajectories must agree to the last bit. One fresh process per variant (the
    choice is made once per process); variants the CPU lacks fall back and trivially agree."""
    import hashlib
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, numpy as np
from egopose_amd.skeleton import load_skeleton
from egopose_amd.physics import SurrogatePhysics
skel = load_skeleton()
rng = np.random.RandomState(0)
n = 5
ph = SurrogatePhysics(skel, n)
h = hashlib.sha256()
for e in range(n):
    q = rng.normal(size=59) * 0.2; q[3:7] = [1, 0, 0, 0]; q[2] = 0.9
    ph.reset(e, q, rng.normal(size=58) * 0.5)
for k in range(40):
    for e in range(n):
        ph.step(e, rng.normal(size=52) * 30)
        q, v, qM, bias, xpos = ph.drain(e)
        h.update(np.ascontiguousarray(q).tobytes()); h.update(np.ascontiguousarray(v).tobytes()); h.update(np.ascontiguousarray(xpos).tobytes())
print("HASH", h.hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for variant in ("plain", "avx2", "avx512"):
        env = dict(os.environ, EGP_SURROGATE_SIMD=variant, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out[variant] = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0]
    assert out["plain"] == out["avx2"] == out["avx512"], out


def test_running_state_pickles_under_the_reference_module_path(tmp_path):
    """Checkpoints name `utils.zfilter.ZFilter` (where the reference defines it), not egopose_amd.zfilter: the unmodified
    reference can load what Trainer.save wrote, and the alias modules do not outlive the call."""
    import pickletools
    from egopose_amd.zfilter import ZFilter, reference_pickle_names
    zf = ZFilter((5,), clip=5)
    for x in np.random.RandomState(0).normal(size=(7, 5)):
        zf(x)
    had = {k: sys.modules.get(k) for k in ("utils", "utils.zfilter")}
    with reference_pickle_names():
        blob = pickle.dumps({"running_state": zf})
    assert {k: sys.modules.get(k) for k in had} == had and ZFilter.__module__ == "egopose_amd.zfilter"
    ops = [str(arg) for _, arg, _ in pickletools.genops(blob) if arg is not None]
    assert any("utils.zfilter" in a for a in ops) and not any("egopose_amd" in a for a in ops)
    with reference_pickle_names():
        back = pickle.loads(blob)["running_state"]
    assert isinstance(back, ZFilter) and back.rs.n == 7
    np.testing.assert_array_equal(back.rs.std, zf.rs.std)
    # the thread-safe forms Trainer.save / load use: same bytes on the wire, nothing process-wide touched, usable from threads
    import io
    import threading
    from egopose_amd.zfilter import dump_reference_pickle, load_reference_pickle
    blobs, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(20):
                b = io.BytesIO()
                dump_reference_pickle({"running_state": zf, "i": i}, b)
                got = load_reference_pickle(io.BytesIO(b.getvalue()))
                assert isinstance(got["running_state"], ZFilter) and got["i"] == i and ZFilter.__module__ == "egopose_amd.zfilter"
            blobs[i] = b.getvalue()
        except Exception as e:           # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs and all(b is not None and b"utils.zfilter" in b and b"egopose_amd" not in b for b in blobs)
    assert isinstance(load_reference_pickle(io.BytesIO(blob))["running_state"], ZFilter)          # the context manager's bytes load too
    if os.path.exists("/root/reference/utils/zfilter.py"):       # build container: the reference's own class takes it
        code = ("import sys, pickle, numpy as np; sys.path.insert(0, '/root/reference/utils'); import importlib.util as iu; "
                "spec = iu.spec_from_file_location('utils.zfilter', '/root/reference/utils/zfilter.py'); m = iu.module_from_spec(spec); "
                "import types; sys.modules['utils'] = types.ModuleType('utils'); sys.modules['utils.zfilter'] = m; spec.loader.exec_module(m); "
                "rs = pickle.load(open(sys.argv[1], 'rb'))['running_state']; assert type(rs).__module__ == 'utils.zfilter'; "
                "print('REF_LOADED', rs.rs.n, float(rs(np.zeros(5), update=False).sum()))")
        path = str(tmp_path / "cp.p")
        open(path, "wb").write(blob)
        out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "REF_LOADED 7" in out.stdout, out.stdout + out.stderr
        assert float(out.stdout.split()[-1]) == pytest.approx(float(zf(np.zeros(5), update=False).sum()), rel=1e-12)


def test_global_step_budget_rearms_parked_slots_when_running_episodes_end_early():
    """rollout.global_budget (EGP_STEP_BUDGET=global): slots park while collected + in-flight steps cover the batch; when the
    episodes in flight then fail early, the decision at their end restarts them AND brings parked slots of the group back, so
    that the batch is covered within a bounded number of ticks instead of by one slot stepping alone. A lockstep simulation of
    two groups in which every episode after the first fails on its second step."""
    from egopose_amd.rollout import global_budget
    N, T_ep, min_batch = 32, 40, 32 * 12
    groups = [(0, 16), (16, 32)]
    steps_done, cur_t, active = np.zeros(N, np.int64), np.zeros(N, np.int64), np.ones(N, bool)
    ep_no = np.zeros(N, np.int64)
    # first episodes: group 0 fails at step 3 and parks (group 1's sixteen episodes would cover the batch), then group 1's episodes
    # fail one after the other at steps 5, 6, ... 20: each parks in turn, and the last one ends with a shortfall of > 1 episode
    fail_at = np.where(np.arange(N) < 16, 3, 5 + np.arange(N) - 16)
    ticks, parked_seen, rearmed = 0, False, 0
    while active.any():
        ticks += 1
        assert ticks <= 12 + 3 * T_ep, "the batch was left to too few slots"
        for a, b in groups:
            act = active[a:b].copy()
            cur_t[a:b] += act
            steps_done[a:b] += act
            done = act & ((cur_t[a:b] >= fail_at[a:b]) | (cur_t[a:b] >= T_ep))
            if done.any():
                ids = np.nonzero(done)[0] + a
                park, rearm = global_budget(steps_done, cur_t, active, ids, a, b, T_ep, min_batch)
                assert set(rearm.tolist()).isdisjoint(ids.tolist()) and all(not active[r] for r in rearm)
                if park:
                    active[ids] = False
                    parked_seen = True
                restart = rearm if park else np.concatenate([ids, rearm])
                active[restart] = True
                cur_t[restart] = 0
                ep_no[restart] += 1
                fail_at[restart] = 2                      # every later episode fails on its second step
                rearmed += len(rearm)
    assert steps_done.sum() >= min_batch and parked_seen
    # the re-arm rule itself: one slot ends, everything else is parked, 284 steps are missing -> that slot restarts and
    # ceil((284 - 40) / 40) = 7 parked slots of ITS group come back (none of the other group's)
    steps_done = np.full(N, 3, np.int64); steps_done[31] = 7          # 100 steps collected
    active = np.zeros(N, bool); active[31] = True
    park, rearm = global_budget(steps_done, np.zeros(N, np.int64), active, np.array([31]), 16, 32, T_ep, min_batch)
    assert not park and rearm.tolist() == list(range(16, 23))
    park, rearm = global_budget(steps_done + 10, np.zeros(N, np.int64), active, np.array([31]), 16, 32, T_ep, min_batch)
    assert park and len(rearm) == 0                                    # 420 steps collected: covered
