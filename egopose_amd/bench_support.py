"""Shared set-up for bench.py, smoke() and the GPU integration tests: a synthetic dataset on disk in the
reference's formats, generated with the product's own kernels (K7 features + backend FK)."""
from __future__ import annotations

import os


def write_synthetic_dataset(root, cfg_id="subject_03", device_index=0, n_takes=8, n_frames=2000, seed=1, cnn_dim=128):
    from .hip import EgpContext
    from .physics import SurrogatePhysics
    from .presets import config_params
    from .skeleton import load_skeleton
    from .synthetic import make_dataset
    sk = load_skeleton()
    p = config_params(cfg_id)
    ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"],
                     episode_len=p["episode_len"], device=device_index)
    phys = SurrogatePhysics(sk, 1)
    try:
        return make_dataset(os.path.abspath(root), ctx, phys, cfg_id, n_takes=n_takes, n_frames=n_frames, cnn_dim=cnn_dim, seed=seed)
    finally:
        phys.close()
        ctx.close()


HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec


def _time_launches(fn, iters=50, warm=5, warm_s=0.03):
    """Seconds per call in the steady state: at least `warm` calls AND `warm_s` seconds of back-to-back launches before
    the timed ones (a few short launches after an idle stretch are timed at the clocks the GPU idles at, not the ones
    it sustains: K1 at 65 536 envs read 334 us after five warm-up calls in `bench.py` and 271 us after 30 ms of them)."""
    import time
    import torch
    t0 = time.perf_counter()
    done = 0
    while done < warm or time.perf_counter() - t0 < warm_s:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        done += warm
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def kernel_microbench(sizes=(1024, 65536), device_index=0, variants=False, iters=50, only=None, rotate=None):
    """K1-K6 / K8 with inputs resident in HBM (HIP events on the launch stream, float64): per (kernel, n) the time per
    call, the ALGORITHMIC bytes (SURVEY.md 8d / DESIGN.md section 4), achieved GB/s and the fraction of the 8 TB/s HBM
    peak. These are the numbers an HBM roofline can bind; the in-rollout K1 launch waits for host physics instead.
    `variants=True` also times the other K1 kernels (lane per row, dense order, generic LDS). Every launch works on the next of
    `rotating_sets` input / output sets, so that a size whose footprint fits the 256 MiB Infinity Cache is still read from HBM
    (`beyond_mall`: the sets together exceed the cache; round 5's figures at 65 536 envs re-ran one set and were cache-resident)."""
    import numpy as np
    import torch
    from .hip import EgpContext
    from .presets import subject_03_params
    from .skeleton import load_skeleton
    sk = load_skeleton()
    p = subject_03_params()
    ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"],
                     device=device_index)
    dev = torch.device("cuda", device_index)
    dt, W = torch.float64, 8
    rng = np.random.RandomState(0)
    qM0 = sk.sparse_from_full(sk.zero_pose_inertia())
    F = 4096                                   # a small expert table for the reward gathers
    take = dict(qpos=rng.normal(size=(F, 59)), qvel=rng.normal(size=(F, 58)), rlinv_local=rng.normal(size=(F, 3)),
                rangv=rng.normal(size=(F, 3)), rq_rmh=rng.normal(size=(F, 4)), ee_pos=rng.normal(size=(F, 15)),
                bquat=rng.normal(size=(F, 84)), bangvel=rng.normal(size=(F, 63)), head_height_lb=1.0)
    ctx.upload_experts([take])
    out = []
    MALL = 256 << 20                           # Infinity Cache (MI355X_MICROARCH.md): inputs re-read from it are not an HBM measurement
    try:
        for n in sizes:
            ns = n * 200 // 8
            # `rot` input / output sets, visited round-robin: a launch only finds its lines in the 256 MiB cache if the sets between
            # two visits of the same set are smaller than the cache. Sized on the smallest per-env footprint (K4: 1 144 B).
            rot = rotate if rotate is not None else max(1, min(16, -(-2 * MALL // (n * 1144))))

            def make_set():
                g = lambda *s: torch.randn(*s, dtype=dt, device=dev)
                d = dict(qpos=g(n, 59) * 0.3, qvel=g(n, 58), act=g(n, 52) * 0.3, C=g(n, 58), ee=g(n, 15), obs=g(n, 115))
                d["qpos"][:, 3:7] = torch.nn.functional.normalize(g(n, 4), dim=1)
                d["prev"] = d["qpos"] + g(n, 59) * 0.01
                d["t"] = torch.randint(1, 100, (n,), dtype=torch.int32, device=dev)
                d["frame"] = torch.randint(0, F, (n,), dtype=torch.int32, device=dev)
                d["end"] = torch.zeros(n, dtype=torch.int32, device=dev)
                d["st0"] = torch.zeros(231, dtype=dt, device=dev)
                d["st1"] = torch.empty_like(d["st0"])
                d["rew"], d["msk"], d["val"] = torch.rand(ns, dtype=dt, device=dev), torch.ones(ns, dtype=dt, device=dev), g(ns)
                return d
            sets = [make_set() for _ in range(rot)]
            big = {}                                           # the 7.3 kB-per-env inertia rows: allocated for the kernels that touch them only

            def need_big(key):
                if key not in big:
                    big[key] = [torch.as_tensor(qM0, device=dev).repeat(n, 1).contiguous() if key == "qM" else
                                torch.empty(n, sk.nM, dtype=dt, device=dev) for _ in range(max(1, min(rot, -(-2 * MALL // (n * 7280)))))]
                return big[key]
            turn = [0]

            def nxt():
                turn[0] += 1
                return sets[turn[0] % rot], turn[0]
            def k1():
                d, i = nxt(); qm = need_big("qM"); return ctx.pd_torque(d["qpos"], d["qvel"], d["act"], qm[i % len(qm)], d["C"])
            def k2():
                d, _ = nxt(); return ctx.reward(d["qpos"], d["prev"], d["ee"], d["t"], d["frame"], d["end"], 0.0)
            def k3():
                d, _ = nxt(); return ctx.obs(d["qpos"], d["qvel"])
            def k4():
                d, _ = nxt(); return ctx.body_quat(d["qpos"])
            def k6():
                d, _ = nxt(); return ctx.zfilter(d["obs"], d["st0"], d["st1"], update=True)
            def k5():
                d, _ = nxt(); return ctx.gae(d["rew"], d["msk"], d["val"], 0.95, 0.95)
            def k8():
                d, i = nxt(); qo = need_big("qM_dyn"); return ctx.dynamics(d["qpos"], d["qvel"], want_xpos=True, qM_out=qo[i % len(qo)])
            cases = [
                ("K1_pd_torque", k1, (910 + 58 + 52 + 58 + 52 + 52) * W, 1),
                ("K2_reward", k2, (59 + 59 + 15 + 166 + 6) * W, 1),
                ("K3_obs", k3, (59 + 58 + 115) * W, 1),
                ("K4_body_quat", k4, (59 + 84) * W, 1),
                ("K6_zfilter", k6, (115 + 115) * W, 1),
                ("K5_gae", k5, 5 * W, ns / n),
                ("K8_dynamics", k8, (59 + 58 + 910 + 58 + 63) * W, 1),
            ]
            for name, fn, bytes_per_unit, units_per_env in cases:
                if only and name not in only:          # (PMC passes profile one kernel at a time: tools/pmc_kernel.sh)
                    continue
                todo = [(0, "_tree58")] + ([(3, "_tree58_rows"), (2, "_reg58"), (1, "_lds")] if variants else []) if name == "K1_pd_torque" else [(None, "")]
                for variant, sfx in todo:
                    if variant is not None:
                        ctx.set_pd_variant(variant)
                    ab = bytes_per_unit * n * units_per_env
                    its = min(iters, 20) if (variant == 1 or ab > (1 << 30)) else iters
                    s = _time_launches(fn, iters=its)
                    sets_k = len(big.get("qM" if name == "K1_pd_torque" else "qM_dyn", [])) if name in ("K1_pd_torque", "K8_dynamics") else rot
                    out.append(dict(kernel=name + sfx, n=n, us=s * 1e6, alg_bytes=ab, GBps=ab / s / 1e9, frac_hbm=ab / s / HBM_PEAK,
                                    rotating_sets=sets_k, beyond_mall=bool(ab * sets_k > 1.5 * MALL)))
                if name == "K1_pd_torque":
                    ctx.set_pd_variant(0)
            del sets, big
            torch.cuda.empty_cache()
    finally:
        ctx.close()
    return out


def statereg_config4(device_index=0, frames=256, steps=8, warmup=5, lr=1e-4):
    """BASELINE config 4 ("state_reg cross_01: ResNet-18 VideoRegNet bf16 on MFMA, batch 256, 1xMI355X") as a timed leg:
    optimisation steps of VideoRegNet (ResNet-18 -> bi-LSTM -> MLP[300,200] -> 115, models/video_reg_net.py:10-59,
    ego_pose/state_reg.py:60-95, config/statereg/cross_01.yml: fr_num 120-frame clips are the reference's unit; 256
    frames per step is BASELINE's batch) on synthetic optical-flow clips of 224 x 224 frames resident in HBM. The encoder
    runs as a bf16 copy on the matrix cores with float32 master weights (nets.Bf16Shadow), the LSTM / MLP in float32.
    HIP events around exactly `steps` steps; loss read back after the timed region."""
    import torch
    from .nets import VideoRegNet
    dev = torch.device("cuda", device_index)
    gen = torch.Generator(device=dev).manual_seed(4)
    net = VideoRegNet(115, 128, 128, no_cnn=False).to(dev).channels_last().bf16_encoder()
    net.train()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=lr)
    x = torch.randn(frames, 1, 3, 224, 224, device=dev, generator=gen)
    gt = torch.randn(frames, 115, device=dev, generator=gen)

    def step():
        loss = (gt - net(x)).pow(2).sum(dim=1).mean()
        opt.zero_grad()
        loss.backward()
        net.encoder_grads_ready()
        opt.step()
        net.encoder_stepped()
        return loss

    first = None
    for _ in range(warmup):
        l = step()
        first = l if first is None else first
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize(dev)
    evs[0].record()
    for i in range(steps):
        l = step()
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    ms = evs[0].elapsed_time(evs[steps]) / steps           # (the quoted rate: all timed steps; the per-step list shows what a cold box does to it)
    n_par = sum(p.numel() for p in net.parameters())
    return {"frames_per_s": frames / (ms * 1e-3), "ms_per_step": ms, "frames_per_step": frames, "frame_shape": [3, 224, 224], "steps": steps,
            "ms_per_step_all": [round(t, 2) for t in per_step], "warmup": warmup, "dtype": "bf16 ResNet-18 encoder on MFMA (float32 master weights) + f32 bi-LSTM / MLP",
            "parameters": int(n_par), "loss_first": float(first), "loss_last": float(l), "data": "synthetic frames resident in HBM",
            "config": "state_reg: VideoRegNet(out 115, v_hdim 128, cnn_fdim 128, mlp [300, 200]), Adam lr %g, one clip of %d frames per step" % (lr, frames)}
