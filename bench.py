#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the ego_mimic PPO rollout+update loop on N MI355X of one node.

A "step" is one PPO iteration: the lockstep rollout of `--envs` env slots per GPU until every slot has met its
step quota (min_batch_size 50 000 per GPU, config/egomimic/subject_03.yml) followed by the full-batch update
(10 epochs). Workload = BASELINE.json configs[1]: "ego_mimic subject_03, 1024 parallel envs on 1xMI355X,
precomputed features, MLP policy/value" on a synthetic subject_03-shaped dataset and random-init nets.
Weak scaling: every rank owns 1024 slots; value = env-steps of all ranks / max-over-ranks wall time.

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     K1 (stable-PD torque, the dominant kernel): algorithmic bytes per launch / mean launch duration
               measured with HIP events on the launch streams inside the timed region, against 8 TB/s HBM
  cpu_baseline the oracle's restatement of the reference CPU sampler (2 forked workers, float64, OMP=1) on a
               bounded sample, same physics backend (rank 0, N=1 only)
"""
import argparse
import faulthandler
import json
import os
import subprocess
import sys
import tempfile
import time

import signal

faulthandler.register(signal.SIGUSR1, all_threads=True)       # `kill -USR1 <pid>`: where is a stuck run waiting?

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec
K1_BYTES_PER_ENV = (910 + 58 + 52 + 58 + 52 + 52) * 8   # qM + qfrc_bias + qpos[7:] + qvel + action + torque, float64


def cpu_baseline(dataset, steps, threads):
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, "-m", "oracle.cpu_env", "--dataset", dataset, "--threads", str(threads), "--steps", str(steps)]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        return {"value": None, "unit": "env-steps/s", "cores": threads, "kind": "port", "sample": "failed: " + out.stderr[-300:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    return {"value": r["env_steps_per_s"], "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "oracle CPU sampler (reference structure: %d forked workers, batch-1 float64 policy, numpy reward/PD, "
                      "OMP_NUM_THREADS=1), %d env-steps of the same synthetic subject_03 workload in %.1f s, physics=%s"
                      % (threads, r["env_steps"], r["seconds"], r["physics"])}


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this container's CPU cgroup, or None."""
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=1024, help="env slots per GPU")
    ap.add_argument("--threads", type=int, default=0, help="host physics threads per GPU (0 = auto)")
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--min-batch", type=int, default=0, help="env-steps per GPU per iteration (0 = config: 50000)")
    ap.add_argument("--cfg", default="subject_03")
    ap.add_argument("--task", choices=["egomimic", "egoforecast"], default="egomimic",
                    help="egoforecast = BASELINE config 5's nets (VideoForecastNet, 90-step episodes, decayed reward)")
    ap.add_argument("--cpu-steps", type=int, default=24000, help="env-steps of the CPU baseline sample (~15 s on 2 cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-k1-events", action="store_true")
    ap.add_argument("--k1-event-every", type=int, default=8, help="bracket K1 with HIP events on every Nth env-step")
    args = ap.parse_args()

    import numpy as np
    import torch
    from egopose_amd import dist as D
    rank, world, local = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local = local % max(1, torch.cuda.device_count())     # (several ranks may share a device in the gloo self-test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import Config, ForecastConfig
    from egopose_amd.train import Trainer

    root = tempfile.mkdtemp(prefix="egp_bench_r%d_" % rank)
    write_synthetic_dataset(root, args.cfg, device_index=local)
    os.chdir(root)
    cfg = (ForecastConfig if args.task == "egoforecast" else Config)(args.cfg, create_dirs=False)
    from egopose_amd.physics import available_cpus, default_threads
    cores = available_cpus()
    n_threads = args.threads or max(args.groups, default_threads(share=world, device_index=local))
    tr = Trainer(cfg, dev, torch.float32, num_envs=args.envs, num_threads=n_threads, num_groups=args.groups, seed_offset=rank)
    min_batch = (args.min_batch or cfg.min_batch_size) * world      # Agent.sample splits it evenly over ranks

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    it = 0
    for _ in range(args.warmup):
        tr.iteration(it, min_batch)
        it += 1
    eng = tr.agent._get_rollout().engine
    if not args.no_k1_events:
        eng.set_profile(True, every=args.k1_event_every)
    eng.reset_timing()
    thr0 = cgroup_throttle()
    barrier()
    t0 = time.time()
    steps_local, t_sample, t_update = 0, 0.0, 0.0
    for _ in range(args.steps):
        log, ts, tu, n = tr.iteration(it, min_batch)
        steps_local += n
        t_sample += ts
        t_update += tu
        it += 1
    barrier()
    elapsed = time.time() - t0
    thr1 = cgroup_throttle()
    tim = eng.timing()
    tot = torch.tensor([float(steps_local), elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        steps_t, el_t = tot[:1].clone(), tot[1:].clone()
        torch.distributed.all_reduce(steps_t)
        torch.distributed.all_reduce(el_t, op=torch.distributed.ReduceOp.MAX)
        total_steps, elapsed = float(steps_t.item()), float(el_t.item())
    else:
        total_steps = float(steps_local)
    ro = tr.agent._get_rollout()
    if rank == 0:
        res = {
            "metric": "env-steps/sec (whole node) ego_mimic PPO", "value": total_steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (rollout kernels K1-K6, physics state) + f32 (policy/value nets)", "data": "synthetic",
            "config": {"workload": ("ego_mimic %s, %d lockstep env slots per MI355X, precomputed (synthetic) CNN features, "
                                    "MLP policy/value + bi-LSTM video context, PPO 10 full-batch epochs" % (args.cfg, args.envs))
                       if args.task == "egomimic" else
                       ("ego_forecast %s, %d lockstep env slots per MI355X, precomputed (synthetic) CNN features, MLP policy/value + "
                        "causal video LSTM over the past %d frames + per-step state LSTM, %d-step episodes, decayed reward, "
                        "PPO 10 full-batch epochs" % (args.cfg, args.envs, cfg.fr_margin, cfg.env_episode_len)),
                       "envs_per_gpu": args.envs, "min_batch_per_gpu": min_batch // world, "physics": ro.sim.physics.name,
                       "host_threads_per_gpu": n_threads, "env_groups": args.groups, "host_cores_seen": cores,
                       "host_cpus_pinned": (len(eng.pinned_cpus) if eng.pinned_cpus else None),
                       "parallelism": "dp%d" % world},
            "env_steps": total_steps, "rollout_only_env_steps_per_s": steps_local / max(t_sample, 1e-9) * world,
            "t_sample_s": t_sample, "t_update_s": t_update,
            "host_cgroup_throttled": None if not (thr0 and thr1) else {"events": thr1[0] - thr0[0], "usec_all_threads": thr1[1] - thr0[1]}, "rollout_timing": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ro.timing.items()},
        }
        if tim["k1_launches"] > 0:
            avg_s = tim["k1_ms"] * 1e-3 / tim["k1_launches"]
            envs_per_launch = args.envs / float(args.groups) / max(1, eng.launches_per_substep)
            sub_per_launch = max(1, eng.substeps_per_launch)       # 15 when the resident K1 serves a whole env-step
            achieved = K1_BYTES_PER_ENV * envs_per_launch * sub_per_launch / avg_s
            traffic, traffic_src = None, None
            try:      # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
                pm = json.load(open(os.path.join(REPO, "profiles", "pmc_k1_traffic.json")))
                traffic = pm["hbm_bytes_per_env_substep"] * envs_per_launch * sub_per_launch
                traffic_src = pm["source"] + "; " + pm["correction"]
            except Exception:
                pass
            res["roofline"] = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_unit": "bytes per launch",
                               "traffic_source": traffic_src,
                               "kernel": "k_pd_server_tree58 (resident: one launch = 15 substeps, duration includes the waits for host physics)" if sub_per_launch > 1 else "k_pd_torque_tree58<double>",
                               "substeps_per_launch": sub_per_launch,
                               "avg_launch_us": avg_s * 1e6, "event_pair_overhead_us_subtracted": tim["event_overhead_us"], "launches_timed": tim["k1_launches"], "event_sampling": "every %d-th env-step of each group inside the timed region" % args.k1_event_every, "envs_per_launch": envs_per_launch,
                               "alg_bytes_per_env_substep": K1_BYTES_PER_ENV}
        else:
            res["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(root, args.cpu_steps, 2)
        print(json.dumps(res))
    tr.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
