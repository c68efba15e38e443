"""state_reg: the video -> humanoid-state regressor that produces the CNN features the ego_mimic hot path consumes and
the state predictions its evaluation fail-safe re-seats on.

Mirrors /root/reference/ego_pose/state_reg.py (train / test / save_inf modes), ego_pose/utils/statereg_config.py and
ego_pose/utils/statereg_dataset.py (file layout `datasets/{meta,traj,fpv_of}`; trajectories as [de-headed qpos[2:],
heading-frame finite-difference qvel], normalised by the training set's mean / std; 'iter' and 'sample' iteration).
The network is `nets.VideoRegNet` (ResNet-18 per optical-flow frame -> bi-LSTM -> MLP -> Linear); on the MI355X the
convolutions run on MIOpen and the LSTM on the persistent HIP kernels. No custom kernel is specific to this path.
"""
from __future__ import annotations

import os
import pickle
import time

import numpy as np
import torch
import yaml

from . import metrics as M
from .config import recreate_dirs

_ASSET_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "config", "statereg")


class StateRegConfig:
    """Drop-in for ego_pose/utils/statereg_config.py:6-51 (same attributes and directory layout)."""

    def __init__(self, cfg_id, create_dirs=False, cfg_dict=None):
        self.id = cfg_id
        if cfg_dict is None:
            path = "config/statereg/%s.yml" % cfg_id
            if not os.path.exists(path):
                path = os.path.join(_ASSET_CFG, "%s.yml" % cfg_id)
                if not os.path.exists(path):
                    print("Config file doesn't exist: config/statereg/%s.yml" % cfg_id)
                    raise SystemExit(0)
            with open(path, "r") as f:
                cfg_dict = yaml.safe_load(f)
        cfg = cfg_dict
        self.base_dir = "results"
        self.cfg_dir = "%s/statereg/%s" % (self.base_dir, cfg_id)
        self.model_dir, self.result_dir = "%s/models" % self.cfg_dir, "%s/results" % self.cfg_dir
        self.log_dir, self.tb_dir = "%s/log" % self.cfg_dir, "%s/tb" % self.cfg_dir
        os.makedirs(self.model_dir, exist_ok=True)
        os.makedirs(self.result_dir, exist_ok=True)
        if create_dirs:
            recreate_dirs(self.log_dir, self.tb_dir)
        for key in ("meta_id", "seed", "fr_num", "v_hdim", "mlp_dim", "cnn_fdim", "lr", "num_epoch", "iter_method",
                    "save_model_interval", "fr_margin", "humanoid_model", "vis_model"):
            setattr(self, key, cfg[key])
        for key, default in (("v_net", "lstm"), ("v_net_param", None), ("shuffle", False), ("num_sample", 20000),
                             ("pose_only", False), ("causal", False), ("cnn_type", "resnet")):
            setattr(self, key, cfg.get(key, default))


def _de_heading(q):
    """utils/math.py:71-72 for a batch: remove the rotation about z."""
    h = M._heading_q(q)
    return M._qmul(M._qinv(h), q)


class Dataset:
    """ego_pose/utils/statereg_dataset.py:8-163."""

    def __init__(self, meta_id, mode, fr_num, iter_method="iter", shuffle=False, overlap=0, num_sample=20000, base_folder="datasets"):
        self.meta_id, self.mode, self.fr_num = meta_id, mode, fr_num
        self.iter_method, self.shuffle, self.overlap, self.num_sample = iter_method, shuffle, overlap, num_sample
        self.base_folder = base_folder
        self.of_folder = os.path.join(base_folder, "fpv_of")
        self.traj_folder = os.path.join(base_folder, "traj")
        with open("%s/meta/%s.yml" % (base_folder, meta_id), "r") as f:
            self.meta = yaml.safe_load(f)
        self.no_traj = self.meta.get("no_traj", False)
        self.msync = self.meta["video_mocap_sync"]
        self.dt = 1 / self.meta["capture"]["fps"]
        self.takes = self.meta["train"] + self.meta["test"] if mode == "all" else self.meta[mode]
        self.len = np.sum([self.msync[x][2] - self.msync[x][1] for x in self.takes])
        self.trajs = self.orig_trajs = self.norm_trajs = None
        self.mean = self.std = None
        if not self.no_traj:
            self.trajs, self.orig_trajs = [], []
            for take in self.takes:
                orig = np.load("%s/%s_traj.p" % (self.traj_folder, take))
                orig[:, 32:35] = 0.0                 # noisy hand pose of the capture
                orig[:, 42:45] = 0.0
                self.trajs.append(np.hstack((self.get_traj_pos(orig), self.get_traj_vel(orig))))
                self.orig_trajs.append(orig)
            if mode == "train":
                all_traj = np.vstack(self.trajs)
                self.mean, self.std = np.mean(all_traj, axis=0), np.std(all_traj, axis=0)
                self.norm_trajs = self.normalize_traj()
            self.traj_dim = self.trajs[0].shape[1]
        self.sample_count = self.take_indices = self.cur_ind = self.cur_tid = self.cur_fr = None
        self.fr_lb = self.fr_ub = self.im_offset = None

    # ------------------------------------------------------------------ trajectory features
    def get_traj_pos(self, orig_traj):
        pos = orig_traj[:, 2:].copy()
        pos[:, 1:5] = _de_heading(pos[:, 1:5])
        return pos

    def get_traj_vel(self, orig_traj):
        vel = M.get_qvel_fd(orig_traj[:-1], orig_traj[1:], self.dt, "heading")
        return np.vstack((vel, vel[-1:]))

    def set_mean_std(self, mean, std):
        self.mean, self.std = mean, std
        if not self.no_traj:
            self.norm_trajs = self.normalize_traj()

    def normalize_traj(self):
        return [(t - self.mean[None, :]) / (self.std[None, :] + 1e-8) for t in self.trajs]

    # ------------------------------------------------------------------ iteration
    def __iter__(self):
        if self.iter_method == "sample":
            self.sample_count = 0
        else:
            self.cur_ind = -1
            self.take_indices = np.arange(len(self.takes))
            if self.shuffle:
                np.random.shuffle(self.take_indices)
            self._next_take()
        return self

    def _next_take(self):
        self.cur_ind += 1
        if self.cur_ind < len(self.take_indices):
            self.cur_tid = self.take_indices[self.cur_ind]
            self.im_offset, self.fr_lb, self.fr_ub = self.msync[self.takes[self.cur_tid]]
            self.cur_fr = self.fr_lb

    def __next__(self):
        if self.iter_method == "sample":
            if self.sample_count >= self.num_sample:
                raise StopIteration
            self.sample_count += self.fr_num - self.overlap
            return self.sample()
        if self.cur_ind >= len(self.takes):
            raise StopIteration
        fr_start = self.cur_fr
        fr_end = self.cur_fr + self.fr_num if self.cur_fr + self.fr_num + 30 < self.fr_ub else self.fr_ub
        of = self.load_of(self.cur_tid, fr_start + self.im_offset, fr_end + self.im_offset)
        norm = None if self.no_traj else self.norm_trajs[self.cur_tid][fr_start: fr_end]
        orig = None if self.no_traj else self.orig_trajs[self.cur_tid][fr_start: fr_end]
        self.cur_fr = fr_end - self.overlap
        if fr_end == self.fr_ub:
            self._next_take()
        return of, norm, orig

    def sample(self):
        take_ind = np.random.randint(len(self.takes))
        im_offset, fr_lb, fr_ub = self.msync[self.takes[take_ind]]
        fr_start = np.random.randint(fr_lb, fr_ub - self.fr_num)
        fr_end = fr_start + self.fr_num
        of = self.load_of(take_ind, fr_start + im_offset, fr_end + im_offset)
        norm = None if self.no_traj else self.norm_trajs[take_ind][fr_start: fr_end]
        orig = None if self.no_traj else self.orig_trajs[take_ind][fr_start: fr_end]
        return of, norm, orig

    def load_of(self, take_ind, start, end):
        folder = "%s/%s" % (self.of_folder, self.takes[take_ind])
        return np.stack([np.load("%s/%05d.npy" % (folder, i)) for i in range(start, end)])


def of_to_frames(of_np, dtype, device):
    """(T, H, W, 2) optical flow -> (T, 1, 3, H, W): third channel zero (state_reg.py:70-71)."""
    of = torch.as_tensor(of_np, dtype=dtype, device=device)
    of = torch.cat((of, of.new_zeros(of.shape[:-1] + (1,))), dim=-1)
    return of.permute(0, 3, 1, 2).unsqueeze(1).contiguous()


def get_traj_from_state_pred(state_pred, init_pos, init_heading, dt, traj_dim):
    """state_reg.py:104-123: integrate the predicted heading-frame velocities into a world trajectory."""
    nv = (traj_dim + 1) // 2
    nq = nv + 1
    pos = np.array(init_pos, float, copy=True)
    heading = np.array(init_heading, float, copy=True)
    out = []
    for i in range(state_pred.shape[0]):
        qpos = np.concatenate((pos, state_pred[i, :nq - 2]))
        qvel = state_pred[i, nq - 2:]
        qpos[3:7] = M._qmul(heading, qpos[3:7])
        linv = M._rot_matrix(heading) @ qvel[:3]
        angv = M._rot_matrix(qpos[3:7]) @ qvel[3:6]
        pos = pos + linv[:2] * dt
        e = angv * dt
        ang = np.linalg.norm(e)
        ax = e / ang if ang >= 1e-12 else np.array([1.0, 0.0, 0.0])
        dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
        heading = M._heading_q(M._qmul(dq, qpos[3:7]))
        out.append(qpos)
    return np.vstack(out)


class StateRegTrainer:
    """The three modes of ego_pose/state_reg.py over one object."""

    def __init__(self, cfg, dataset, device, dtype=torch.float32, no_cnn=False, frame_shape=(3, 224, 224), autocast=None,
                 bf16_encoder=None):
        from .nets import VideoRegNet
        self.cfg, self.dataset, self.device, self.dtype = cfg, dataset, torch.device(device), dtype
        self.state_dim = (dataset.traj_dim - 1) // 2 + 6 if cfg.pose_only else dataset.traj_dim
        self.net = VideoRegNet(self.state_dim, cfg.v_hdim, cfg.cnn_fdim, no_cnn=no_cnn, frame_shape=frame_shape, cnn_type=cfg.cnn_type,
                               mlp_dim=cfg.mlp_dim, v_net_type=cfg.v_net, v_net_param=cfg.v_net_param, causal=cfg.causal)
        self.net.to(self.device, dtype)
        if self.device.type == "cuda" and not no_cnn:
            self.net.channels_last()       # NHWC encoder: MIOpen's faster layout on the MI355X (nets.VideoRegNet.channels_last)
        self.optimizer = torch.optim.Adam([p for p in self.net.parameters() if p.requires_grad], lr=cfg.lr)
        self.autocast = autocast           # torch.autocast dtype (measured 3.4x SLOWER than float32 here: kept for A/B runs)
        # BASELINE config 4: the ResNet-18 encoder in bf16 on the matrix cores, float32 master weights (nets.Bf16Shadow);
        # default on the GPU for float32 trainers, bf16_encoder=False keeps float32 convolutions
        if bf16_encoder is None:
            bf16_encoder = self.device.type == "cuda" and dtype == torch.float32 and not no_cnn and autocast is None
        self.bf16_encoder = bool(bf16_encoder)
        if self.bf16_encoder:
            self.net.bf16_encoder()

    def _forward(self, of_np):
        x = of_to_frames(of_np, self.dtype, self.device)
        if self.autocast is not None and self.device.type == "cuda":
            with torch.autocast("cuda", dtype=self.autocast):
                return self.net(x).float()
        return self.net(x)

    def train_epoch(self):
        m = self.cfg.fr_margin
        self.net.train()
        t0, n_sample, loss_sum = time.time(), 0, 0.0
        for of_np, traj_np, _ in self.dataset:
            num = traj_np.shape[0] - 2 * m
            gt = torch.as_tensor(traj_np[m:-m, :self.state_dim], dtype=torch.float32 if self.autocast else self.dtype, device=self.device)
            pred = self._forward(of_np)[m:-m]
            loss = (gt - pred).pow(2).sum(dim=1).mean()
            self.optimizer.zero_grad()
            loss.backward()
            self.net.encoder_grads_ready()
            self.optimizer.step()
            self.net.encoder_stepped()
            loss_sum += float(loss.detach()) * num
            n_sample += num
        return loss_sum / max(1, n_sample), n_sample, time.time() - t0

    @torch.no_grad()
    def test(self):
        """-> (results, meta) in the reference's layout: per take the integrated predicted trajectory and the mocap one."""
        ds, m = self.dataset, self.cfg.fr_margin
        self.net.eval()
        ds.iter_method, ds.shuffle = "iter", False
        res_pred, res_orig, preds, origs = {}, {}, [], []
        n_sample, loss_sum = 0, 0.0
        take = ds.takes[0]
        for of_np, traj_np, orig_np in ds:
            num = traj_np.shape[0] - 2 * m
            gt = torch.as_tensor(traj_np[m:-m, :self.state_dim], dtype=torch.float32 if self.autocast else self.dtype, device=self.device)
            pred = self._forward(of_np)[m:-m]
            loss_sum += float((gt - pred).pow(2).sum(dim=1).mean()) * num
            n_sample += num
            preds.append(pred.double().cpu().numpy() * ds.std[None, :self.state_dim] + ds.mean[None, :self.state_dim])
            origs.append(orig_np[m:-m])
            if ds.cur_ind >= len(ds.takes) or ds.takes[ds.cur_tid] != take:
                sp, orig = np.vstack(preds), np.vstack(origs)
                res_pred[take] = get_traj_from_state_pred(sp, orig[0, :2], M._heading_q(orig[0, 3:7]), ds.dt, ds.traj_dim)
                res_orig[take] = orig
                preds, origs = [], []
                if ds.cur_ind < len(ds.takes):
                    take = ds.takes[ds.cur_tid]
        meta = {"algo": "state_reg", "num_sample": n_sample, "epoch_loss": loss_sum / max(1, n_sample)}
        return {"traj_pred": res_pred, "traj_orig": res_orig}, meta

    @torch.no_grad()
    def cnn_features(self):
        """What ego_pose/data_process/gen_cnn_feature.py stores: per take the (frames, cnn_fdim) encoder output."""
        ds = self.dataset
        self.net.eval()
        out = {}
        for ti, take in enumerate(ds.takes):
            off, lb, ub = ds.msync[take]
            feats = []
            for s in range(lb, ub, 64):
                x = of_to_frames(ds.load_of(ti, s + off, min(ub, s + 64) + off), self.dtype, self.device)
                feats.append(self.net.get_cnn_feature(x).double().cpu().numpy())
            out[take] = np.vstack(feats)
        return out

    def save(self, path, inference=False):
        sd = {k: v.detach().cpu() for k, v in self.net.state_dict().items() if not (inference and k.startswith("cnn."))}
        meta = {"mean": self.dataset.mean, "std": self.dataset.std}
        if inference:
            meta["cfg"] = self.cfg
        with open(path, "wb") as f:
            pickle.dump(({"state_net_dict": sd}, meta), f)

    def load(self, path, strict=True):
        with open(path, "rb") as f:
            cp, meta = pickle.load(f)
        self.net.load_state_dict(cp["state_net_dict"], strict=strict)
        self.net.encoder_stepped()           # the bf16 encoder copy follows the loaded weights
        return meta


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="subject_03")
    ap.add_argument("--mode", default="train", choices=["train", "test", "save_inf"])
    ap.add_argument("--data", default=None)
    ap.add_argument("--gpu-index", type=int, default=0)
    ap.add_argument("--iter", type=int, default=0)
    ap.add_argument("--bf16", action="store_true", help="autocast the encoder / GEMMs to bfloat16 (fp32 master weights)")
    args = ap.parse_args(argv)
    data = args.data or (args.mode if args.mode in ("train", "test") else "train")
    cfg = StateRegConfig(args.cfg, create_dirs=(args.iter == 0 and args.mode == "train"))
    np.random.seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    ds = Dataset(cfg.meta_id, data, cfg.fr_num, cfg.iter_method, cfg.shuffle, 2 * cfg.fr_margin, cfg.num_sample)
    dev = torch.device("cuda", args.gpu_index) if torch.cuda.is_available() else torch.device("cpu")
    tr = StateRegTrainer(cfg, ds, dev, no_cnn=args.mode == "save_inf", autocast=torch.bfloat16 if args.bf16 else None)
    if args.iter > 0:
        meta = tr.load("%s/iter_%04d.p" % (cfg.model_dir, args.iter), strict=args.mode != "save_inf")
        if data != "train":
            ds.set_mean_std(meta["mean"], meta["std"])
    if args.mode == "train":
        for ep in range(args.iter, cfg.num_epoch):
            loss, n, dt = tr.train_epoch()
            print("epoch %4d    time %.2f     nsample %d   loss %.4f" % (ep, dt, n, loss))
            if cfg.save_model_interval > 0 and (ep + 1) % cfg.save_model_interval == 0:
                tr.save("%s/iter_%04d.p" % (cfg.model_dir, ep + 1))
    elif args.mode == "test":
        results, meta = tr.test()
        path = "%s/iter_%04d_%s.p" % (cfg.result_dir, args.iter, data)
        with open(path, "wb") as f:
            pickle.dump((results, meta), f)
        print("nsample %d   loss %.4f\nsaved results to %s" % (meta["num_sample"], meta["epoch_loss"], path))
    else:
        tr.save("%s/iter_%04d_inf.p" % (cfg.model_dir, args.iter), inference=True)


if __name__ == "__main__":
    main()
