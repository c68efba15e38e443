#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the ego_mimic PPO rollout+update loop on N MI355X of one node.

A "step" is one PPO iteration: the lockstep rollout of `--envs` env slots per GPU until every slot has met its
step quota (min_batch_size 50 000 per GPU, config/egomimic/subject_03.yml) followed by the full-batch update
(10 epochs). Workload = BASELINE.json configs[1]: "ego_mimic subject_03, 1024 parallel envs on 1xMI355X,
precomputed features, MLP policy/value" on a synthetic subject_03-shaped dataset and random-init nets.
Weak scaling: every rank owns 1024 slots; value = env-steps of all ranks / max-over-ranks wall time.

`--gpus N` with N > 1 and no torchrun environment (WORLD_SIZE unset) launches the N ranks itself -- one process per GPU
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, backend nccl (= RCCL) -- as the reference starts
all of its parallelism from one command (agents/agent.py:93-108). Under torchrun it is one of the ranks.

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     K1 (stable-PD torque, the dominant kernel): algorithmic bytes of the env-substeps the timed launches
               actually STEPPED (finished slots of the rollout's tail move nothing) / their duration, measured with HIP
               events on the launch streams inside the timed region, against 8 TB/s HBM; `frac_slot_based` is the same
               with every slot of a launch counted (round 1's figure)
  kernels      K1-K6 / K8 with inputs resident in HBM at 1 024 and 65 536 envs: the figures an HBM roofline can bind
  legs         (1 GPU) the same workload (i) through the float64 driver set-up of the unmodified reference
               (`dropin_env_steps_per_s`), (ii) with an inertia that changes on every substep fed from the host (the
               traffic of a MuJoCo-like backend), (iii) with that inertia computed on the GPU (row f1), (iv) the
               ego_forecast nets of BASELINE config 5 on this GPU's 1 024-slot shard, (v) with a physics substep that costs
               `--sim-cost-us` (20 us: the order of an mj_step of this humanoid; the surrogate's own is ~0.3 us) next to the
               CPU sampler at the same cost and to what the host threads alone allow, (vi) BASELINE config 4: frames/s of the
               state regressor's optimisation step (bf16 ResNet-18 encoder, batch 256 x 224 x 224)
  cpu_baseline the oracle's restatement of the reference CPU sampler (2 forked workers, float64, OMP=1) on a
               bounded sample, same physics backend (rank 0, N=1 only)
"""
import argparse
import faulthandler
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

faulthandler.register(signal.SIGUSR1, all_threads=True)       # `kill -USR1 <pid>`: where is a stuck run waiting?

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec
K1_BYTES_PER_ENV = (910 + 58 + 52 + 58 + 52 + 52) * 8   # qM + qfrc_bias + qpos[7:] + qvel + action + torque, float64
K1_MOVED_BYTES_PER_ENV = K1_BYTES_PER_ENV - 910 * 8      # ... without the inertia row: what moves while the backend's inertia is constant
K1_LINK_BYTES_PER_ENV = 22 * 64                         # the state row as the resident K1 reads it over PCIe: 22 whole lines (qpos | qvel | bias)


def cpu_baseline(dataset, steps, threads, extra_env=None, update_steps=0):
    """The oracle's restatement of the reference sampler on `threads` forked workers (and, with `update_steps`, of the
    reference's CPU update on the first episodes of that sample): a bounded sample in a subprocess."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", **(extra_env or {}))
    if not update_steps:            # README.md:25-27: OMP_NUM_THREADS=1 for the multi-process sampler (the update uses torch's threads)
        env.update(OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "oracle.cpu_env", "--dataset", dataset, "--threads", str(threads), "--steps", str(steps)]
    if update_steps:
        cmd += ["--update-steps", str(update_steps), "--update-threads", str(threads)]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        return {"value": None, "unit": "env-steps/s", "cores": threads, "kind": "port", "sample": "failed: " + out.stderr[-300:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    res = {"value": r["env_steps_per_s"], "unit": "env-steps/s", "cores": threads, "kind": "port",
           "sample": "oracle CPU sampler (reference structure: %d forked workers, batch-1 float64 policy, numpy reward/PD, "
                     "OMP_NUM_THREADS=1), %d env-steps of the same synthetic subject_03 workload in %.1f s, physics=%s"
                     % (threads, r["env_steps"], r["seconds"], r["physics"])}
    if "update" in r:
        u = r["update"]
        res["t_update"] = {"seconds": u["seconds"], "samples": u["samples"], "episodes": u["episodes"], "epochs": u["epochs"],
                           "torch_threads": u["torch_threads"], "samples_per_s": u["samples"] / max(u["seconds"], 1e-9),
                           "what": "oracle.ppo.update_params: the reference's update structure on the CPU (float64, LSTMCell loops over "
                                   "220-frame windows, 10 full-batch epochs) on the first episodes of the sample"}
    return res


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this container's CPU cgroup, or None."""
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0))
    except Exception:
        return None


def launch_ranks(n, argv):
    """Start the n ranks of this bench (one process per GPU) and wait for them. Rank 0 inherits stdout, so its JSON line
    is this command's output; the other ranks' stdout goes to stderr. Any rank failing takes the others down."""
    # (the probe socket stays open, SO_REUSEADDR, until the ranks are started: nobody else is handed the port in between)
    probe = socket.socket()
    probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL needs it on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    probe.close()
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def aggregate(steps_local, elapsed, world, device):
    """(env-steps of all ranks, max-over-ranks wall time)."""
    import torch
    if world == 1:
        return float(steps_local), float(elapsed)
    from egopose_amd import dist as D
    dev = D._comm_device(device)
    st = torch.tensor([float(steps_local)], dtype=torch.float64, device=dev)
    el = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(st)
    torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    return float(st.item()), float(el.item())


def base_line(args, world, total_steps, elapsed):
    return {"metric": "env-steps/sec (whole node) ego_mimic PPO", "value": total_steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None}


def dry_run(args):
    """Launcher / rendezvous / aggregation self-test without a GPU (gloo): every rank 'steps' a fixed amount of fake work."""
    import torch
    from egopose_amd import dist as D
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        torch.distributed.barrier()
    t0 = time.time()
    time.sleep(0.02 * (rank + 1))
    if world > 1:
        torch.distributed.barrier()
    total, elapsed = aggregate(100 * (rank + 1) * args.steps, time.time() - t0, world, "cpu")
    if rank == 0:
        res = base_line(args, world, total, elapsed)
        res.update(dtype="none", data="none", dry_run=True, env_steps=total,
                   config={"workload": "launcher self-test (no GPU work)", "parallelism": "dp%d" % world})
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _med(xs):
    xs = sorted(xs)
    n = len(xs)
    return None if n == 0 else (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2]))


def explain(args, world, eng, ro, t_sample, t_update, per_iter, iter_ms, iter_steps, thr0, thr1, probe0, probe1, info):
    """Everything that tells a slow run from a slow box, as plain numbers for the `config` dict (the record the driver keeps of
    this line retains `config` / `roofline` / `cpu_baseline` and drops unknown top-level keys): the reference prints T_sample /
    T_update per iteration (ego_pose/ego_mimic.py:115-126), this line carries them per iteration, with the median-iteration rate
    and its spread, the engine mode actually taken, the host, and the probe (egp_host_probe) before and after the timed region."""
    rates = [n * world / (ms * 1e-3) for n, ms in zip(iter_steps, iter_ms)]
    out = {
        "t_sample_s": round(t_sample, 4), "t_update_s": round(t_update, 4),
        "t_sample_ms_per_iteration": [p[0] for p in per_iter], "t_update_ms_per_iteration": [p[1] for p in per_iter],
        "t_sample_ms_median": _med([p[0] for p in per_iter]), "t_update_ms_median": _med([p[1] for p in per_iter]),
        "iteration_ms": [round(x, 1) for x in iter_ms],
        "rollout_only_env_steps_per_s": round(sum(iter_steps) * world / max(t_sample, 1e-9), 1),
        "env_steps_per_s_median_iteration": round(_med(rates), 1), "env_steps_per_s_min_iteration": round(min(rates), 1),
        "env_steps_per_s_max_iteration": round(max(rates), 1),
        "value_is": "env-steps of the K timed iterations / their wall time (bench contract); the median / min / max iteration beside it",
        "host_cgroup_throttled_events": (thr1[0] - thr0[0]) if (thr0 and thr1) else None,
        "host_cgroup_throttled_ms_all_threads": round((thr1[1] - thr0[1]) * 1e-3, 1) if (thr0 and thr1) else None,
        "engine_substeps_per_launch": eng.substeps_per_launch, "engine_envs_per_wave": eng.envs_per_wave,
        "engine_resident_capacity": eng.resident_capacity,
        "engine_go_words": {1: "vram(BAR)", 0: "pinned", -1: "n/a"}.get(int(eng.lib.egp_engine_go_words_in_vram(eng.handle)), "?"),
        "rollout_ticks": ro.timing.get("ticks"), "rollout_small_group_ticks": ro.timing.get("small_group_ticks"),
        "rollout_wait_s": round(ro.timing.get("wait", 0.0), 4), "rollout_setup_s": round(ro.timing.get("setup", 0.0), 4),
        "rollout_assemble_s": round(ro.timing.get("assemble", 0.0), 4),
        #  with prefetch_rollout the next pass's set-up (resets, context pool, noise: host work) runs inside update_params
        # behind the enqueued epochs: T_sample then excludes it and T_update contains it (hidden under the GPU's epochs); the
        # reference's T_sample includes its resets (agents/agent.py:29-76). Last rollout's figure:
        "rollout_setup_prepared_ms": round(ro.timing.get("setup", 0.0) * 1e3, 2) if ro.timing.get("setup_prepared") else 0.0,
        "t_split_note": "T_sample excludes the sampling set-up when it was prepared inside update_params (rollout_setup_prepared_ms > 0): "
                        "that host time is inside T_update; env-steps/s is end to end either way",
    }
    out.update({"host_" + k: v for k, v in info.items()})
    for tag, pr in (("probe", probe0), ("probe_after", probe1)):
        if pr is None:
            continue
        if "error" in pr:
            out[tag + "_error"] = pr["error"]
            continue
        out.update({tag + "_pcie_read_GBps": round(pr["pcie_read_gbps"], 2), tag + "_go_rtt_us_p50": round(pr["go_rtt_us_p50"], 2),
                    tag + "_go_rtt_us_p99": round(pr["go_rtt_us_p99"], 2), tag + "_go_rtt_us_max": round(pr["go_rtt_us_max"], 1),
                    tag + "_spin_gap_us_max": round(pr["spin_gap_us_max"], 1),
                    tag + "_spin_gap_us_median_thread_max": round(pr["spin_gap_us_median_of_thread_max"], 1),
                    tag + "_spin_lost_frac": round(pr["spin_lost_frac"], 5), tag + "_spin_gaps_over_5us": pr["spin_gaps_over_5us"]})
        if tag == "probe":
            out.update({"probe_large_bar": bool(pr["large_bar"]), "probe_go_in_vram": bool(pr["go_in_vram"]), "probe_spin_threads": pr["spin_threads"],
                        "probe_what": "egp_host_probe: 1024 pinned state rows read K1's way; go word -> row back round trip; spinning threads' clock gaps"})
    return out


CONFIG_FIRST = ("workload", "env_steps_per_s_changing_inertia_device", "env_steps_per_s_changing_inertia_host_fed",
                "env_steps_per_s_20us_substep", "k1_avg_launch_us_device_dynamics",
                "probe_go_rtt_us_p50", "probe_pcie_read_GBps", "probe_spin_gap_us_max", "host_loadavg_1m",
                "t_sample_ms_median", "t_update_ms_median", "rollout_setup_prepared_ms", "env_steps_per_s_median_iteration",
                "env_steps_per_s_min_iteration", "env_steps_per_s_max_iteration", "env_steps_per_s_2048_slots",
                "env_steps_per_s_4096_slots", "engine_substeps_per_launch", "engine_envs_per_wave", "physics", "host_threads_per_gpu",
                "envs_per_gpu", "parallelism", "frac_of_host_physics_ceiling_20us")


def order_config(cfg, legs):
    """The driver's record of this line keeps the first ~21 scalar keys of `config`: the physics modes a MuJoCo-shaped backend
    would see (from `legs`, which the record drops), the host probe and the sample / update split go first; everything else after."""
    legs = legs or {}

    def leg(name, key="env_steps_per_s"):
        v = legs.get(name, {}).get(key)
        return round(v, 1) if isinstance(v, float) else v
    lead = {"env_steps_per_s_changing_inertia_device": leg("changing_inertia_device_dynamics"),
            "env_steps_per_s_changing_inertia_host_fed": leg("changing_inertia_host_fed"),
            "env_steps_per_s_20us_substep": leg("simulator_cost_per_substep"),
            "frac_of_host_physics_ceiling_20us": leg("simulator_cost_per_substep", "frac_of_host_physics_ceiling"),
            "k1_avg_launch_us_device_dynamics": leg("changing_inertia_device_dynamics", "k1_avg_launch_us")}
    sweep = legs.get("envs_per_gpu_sweep", {})
    for n in (2048, 4096):
        best = [v.get("env_steps_per_s") for k, v in sweep.items() if k.startswith("slots_%d_" % n) and isinstance(v, dict) and v.get("env_steps_per_s")]
        lead["env_steps_per_s_%d_slots" % n] = round(max(best), 1) if best else None
    merged = dict(cfg, **lead)
    out = {k: merged[k] for k in CONFIG_FIRST if k in merged}
    out.update({k: v for k, v in merged.items() if k not in out})
    return out


def k1_roofline(tim, envs_per_group, eng, every, probe=None):
    """K1 roofline from the engine's event-bracketed launches (see the module docstring)."""
    if tim["k1_launches"] <= 0:
        return None
    total_s = tim["k1_ms"] * 1e-3
    avg_s = total_s / tim["k1_launches"]
    sub_per_launch = max(1, eng.substeps_per_launch)       # 15 when the resident K1 serves a whole env-step
    slots_per_launch = envs_per_group
    achieved = K1_BYTES_PER_ENV * tim["k1_env_substeps"] / total_s
    slot_based = K1_BYTES_PER_ENV * slots_per_launch * sub_per_launch / avg_s
    stepped_per_launch = tim["k1_env_substeps"] / float(sub_per_launch) / tim["k1_launches"]
    traffic, traffic_src = None, None
    try:      # HBM bytes from the committed PMC passes (rocprofv3 cannot run inside this process): tools/profile_round.sh
        pm = json.load(open(os.path.join(REPO, "profiles", "pmc_k1_traffic.json")))
        per = pm.get("hbm_bytes_per_stepped_env_substep") or pm["hbm_bytes_per_env_substep"]
        traffic = per * stepped_per_launch * sub_per_launch
        traffic_src = ("NOT measured in this run: committed profile taken at commit %s, " % pm.get("commit", "?")) + pm["source"] + "; " + pm["correction"]
    except Exception:
        pass
    # what the launch actually waits for: 15 host round trips and the PCIe read of the state rows (22 whole 64-byte lines per
    # env-substep); the inertia row (910 of the 1 182 "algorithmic" doubles) never moves while the backend's inertia is constant
    link_bytes = K1_LINK_BYTES_PER_ENV * tim["k1_env_substeps"]
    link_gbps = link_bytes / total_s / 1e9
    probe_gbps = (probe or {}).get("pcie_read_gbps")
    moved = K1_MOVED_BYTES_PER_ENV * tim["k1_env_substeps"] / total_s
    return {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
            "bound_in_rollout": "host round trips (%d per launch) + PCIe read of the state rows; HBM is not the limit here" % sub_per_launch,
            "pcie_frac": (link_gbps / probe_gbps) if probe_gbps else None, "pcie_GBps": link_gbps, "pcie_peak_GBps_probe": probe_gbps,
            "pcie_bytes_per_env_substep": K1_LINK_BYTES_PER_ENV,
            "alg_bytes_moved_per_env_substep": K1_MOVED_BYTES_PER_ENV, "achieved_moved_GBps": moved / 1e9, "frac_moved": moved / HBM_PEAK,
            "traffic": traffic, "traffic_unit": "bytes per launch (mean stepped envs)", "traffic_source": traffic_src,
            "kernel": "k_pd_server_tree58 (resident: one launch = 15 substeps, duration includes the waits for host physics)"
                      if sub_per_launch > 1 else "k_pd_torque_tree58<double>",
            "substeps_per_launch": sub_per_launch, "avg_launch_us": avg_s * 1e6,
            "stepped_envs_per_launch": stepped_per_launch, "slots_per_launch": slots_per_launch,
            "frac_slot_based": slot_based / HBM_PEAK, "achieved_slot_based": slot_based / 1e9,
            "event_pair_overhead_us_subtracted": tim["event_overhead_us"], "launches_timed": tim["k1_launches"],
            "event_sampling": "every %d-th env-step of each group inside the timed region" % every,
            "alg_bytes_per_env_substep": K1_BYTES_PER_ENV,
            "note": "the in-rollout launch is bound by the host physics round trip, not by HBM; see `kernels` for the "
                    "HBM-resident rates of the same arithmetic"}


def run_leg(make_trainer, steps, warmup, min_batch, every, env=None, default_dtype=None):
    """A short extra measurement of the same workload under another configuration (module docstring: `legs`)."""
    import torch
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    tr = None
    try:
        if default_dtype is not None:
            torch.set_default_dtype(default_dtype)       # the reference driver's process-wide setting (ego_mimic.py:31-32)
        tr = make_trainer()
        it = 0
        for _ in range(warmup):
            tr.iteration(it, min_batch)
            it += 1
        eng = tr.agent._get_rollout().engine
        eng.set_profile(True, every=every)
        eng.reset_timing()
        torch.cuda.synchronize()
        t0 = time.time()
        n_steps, t_sample, t_update, per_iter = 0, 0.0, 0.0, []
        for _ in range(steps):
            _, ts, tu, n = tr.iteration(it, min_batch)
            n_steps += n
            t_sample += ts
            t_update += tu
            per_iter.append([round(ts * 1e3, 1), round(tu * 1e3, 1)])
            it += 1
        torch.cuda.synchronize()
        elapsed = time.time() - t0
        tim = eng.timing()
        out = {"env_steps_per_s": n_steps / elapsed, "rollout_only_env_steps_per_s": n_steps / max(t_sample, 1e-9),
               "steps": steps, "warmup": warmup, "t_sample_s": t_sample, "t_update_s": t_update, "per_iteration_ms_sample_update": per_iter,
               "indicative": bool(steps < 3 or warmup < 2),      # fewer iterations than that is a smoke run, not a measurement
               "substeps_per_launch": eng.substeps_per_launch, "envs_per_wave": eng.envs_per_wave,
               "device_dynamics": bool(getattr(eng, "device_dynamics", False)),
               "inertia_uploads": int(eng.lib.egp_engine_inertia_uploads(eng.handle))}
        if tim["k1_launches"] > 0:
            out["k1_avg_launch_us"] = tim["k1_ms"] * 1e3 / tim["k1_launches"]
            out["k1_stepped_envs_per_launch"] = tim["k1_env_substeps"] / float(max(1, eng.substeps_per_launch)) / tim["k1_launches"]
        rt = tr.agent._get_rollout().timing                  # of the last rollout
        out.update({"env_steps_per_iteration": n_steps / max(1, steps), "ticks": rt.get("ticks"), "step_budget": rt.get("step_budget"),
                    "small_group_ticks": rt.get("small_group_ticks"), "small_group_tick_s": rt.get("small_group_tick_s")})
        return out
    except Exception as e:                       # a leg never takes the headline down
        return {"error": repr(e)[:300]}
    finally:
        if tr is not None:
            tr.close()
        torch.set_default_dtype(torch.float32)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2,
                    help="untimed iterations (the first two still grow the allocator's pools: batch sizes differ from rollout to rollout)")
    ap.add_argument("--envs", type=int, default=1024, help="env slots per GPU")
    ap.add_argument("--threads", type=int, default=0, help="host physics threads per GPU (0 = auto)")
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--min-batch", type=int, default=0, help="env-steps per GPU per iteration (0 = config: 50000)")
    ap.add_argument("--cfg", default="subject_03")
    ap.add_argument("--task", choices=["egomimic", "egoforecast"], default="egomimic",
                    help="egoforecast = BASELINE config 5's nets (VideoForecastNet, 90-step episodes, decayed reward)")
    ap.add_argument("--cpu-steps", type=int, default=24000, help="env-steps of the CPU baseline sample (~15 s on 2 cores)")
    ap.add_argument("--cpu-update-steps", type=int, default=1500, help="steps of the CPU sample the CPU update leg runs on (~10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-k1-events", action="store_true")
    ap.add_argument("--no-host-probe", action="store_true", help="skip egp_host_probe around the timed region")
    ap.add_argument("--k1-event-every", type=int, default=8, help="bracket K1 with HIP events on every Nth env-step")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra 1-GPU legs (drop-in dtype, changing inertia)")
    ap.add_argument("--sim-cost-us", type=float, default=20.0,
                    help="per-substep physics cost of the `simulator_cost_per_substep` leg (busy wait inside the surrogate's step)")
    ap.add_argument("--no-kernels", action="store_true", help="skip the HBM-resident kernel microbenchmarks")
    ap.add_argument("--leg-steps", type=int, default=3, help="timed iterations of every extra leg (after --leg-warmup untimed ones)")
    ap.add_argument("--leg-warmup", type=int, default=3)
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous self-test: no GPU work (gloo on CPU)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        # one command starts every rank (no torchrun needed); never fall back to one rank silently
        if not args.dry_run:
            import torch
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            # EGP_BENCH_SHARE_GPU=1 (self-test of the multi-rank path on a one-GPU box, with EGP_DIST_BACKEND=gloo): ranks
            # then share devices round-robin -- not a measurement
            if have < args.gpus and not (have >= 1 and os.environ.get("EGP_BENCH_SHARE_GPU") == "1"):
                raise SystemExit("--gpus %d but this node shows %d GPU(s)" % (args.gpus, have))
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, env_world))
    if args.dry_run:
        return dry_run(args)

    import numpy as np          # noqa: F401
    import torch
    from egopose_amd import dist as D
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from egopose_amd.bench_support import kernel_microbench, write_synthetic_dataset
    from egopose_amd.config import Config, ForecastConfig
    from egopose_amd.train import Trainer

    root = tempfile.mkdtemp(prefix="egp_bench_r%d_" % rank)
    write_synthetic_dataset(root, args.cfg, device_index=local)
    os.chdir(root)
    cfg_cls = ForecastConfig if args.task == "egoforecast" else Config
    cfg = cfg_cls(args.cfg, create_dirs=False)
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
    updater = tr.agent._get_updater()
    if world > 1 and updater is not None:
        updater.collective_ms()                 # (drop the warm-up's)
        updater.time_collectives = True
    # the box, measured right in front of the timed region and again behind it (rank 0; ~0.4 s each, the engine's threads asleep):
    # two runs of one commit that differ in env-steps/s differ here too -- or the code is what moved
    from egopose_amd.physics import host_info, host_probe
    probe0 = probe1 = None
    if rank == 0 and not args.no_host_probe:
        try:
            probe0 = host_probe(local, n_threads, 250)
        except Exception as e:
            probe0 = {"error": repr(e)[:200]}
    thr0 = cgroup_throttle()
    barrier()
    D.COLLECTIVES["count"] = 0                    # (after the barrier: only the timed iterations' collectives are counted)
    t0 = time.time()
    steps_local, t_sample, t_update, per_iter, iter_ms, iter_steps = 0, 0.0, 0.0, [], [], []
    for _ in range(args.steps):
        ti0 = time.time()
        log, ts, tu, n = tr.iteration(it, min_batch)
        iter_ms.append((time.time() - ti0) * 1e3)
        iter_steps.append(n)
        steps_local += n
        t_sample += ts
        t_update += tu
        per_iter.append((round(ts * 1e3, 1), round(tu * 1e3, 1)))
        it += 1
    n_collectives = D.COLLECTIVES["count"]
    barrier()
    elapsed = time.time() - t0
    thr1 = cgroup_throttle()
    if rank == 0 and probe0 is not None and "error" not in probe0:
        try:
            probe1 = host_probe(local, n_threads, 250)
        except Exception as e:
            probe1 = {"error": repr(e)[:200]}
    tim = eng.timing()
    ar_ms = updater.collective_ms() if (world > 1 and updater is not None) else []
    total_steps, elapsed = aggregate(steps_local, elapsed, world, dev)
    ro = tr.agent._get_rollout()
    res = None
    if rank == 0:
        res = base_line(args, world, total_steps, elapsed)
        res.update({
            "dtype": "f64 (rollout kernels K1-K6, physics state) + f32 (policy/value nets)", "data": "synthetic",
            "ranks_share_gpus": bool(world > torch.cuda.device_count()),
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
            "t_sample_s": t_sample, "t_update_s": t_update, "per_iteration_ms_sample_update": per_iter,
            "host_cgroup_throttled": None if not (thr0 and thr1) else {"events": thr1[0] - thr0[0], "usec_all_threads": thr1[1] - thr0[1]},
            "rollout_timing": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ro.timing.items()},
        })
        res["config"].update(explain(args, world, eng, ro, t_sample, t_update, per_iter, iter_ms, iter_steps, thr0, thr1, probe0, probe1,
                                     host_info(local)))
        res["roofline"] = k1_roofline(tim, args.envs / float(args.groups), eng, args.k1_event_every,
                                      probe0 if (probe0 and "error" not in probe0) else None)
        if res["roofline"] is not None:          # (the decomposition next to the kernel figure too: this dict is kept whole)
            c = res["config"]
            res["roofline"].update({"t_sample_ms_median": c["t_sample_ms_median"], "t_update_ms_median": c["t_update_ms_median"]})
        if world > 1:
            # SURVEY 8e / BASELINE.md G2: one fused gradient all-reduce per PPO epoch (476 109 floats), timed with HIP events on rank 0
            res["collectives"] = {"backend": torch.distributed.get_backend(), "ranks": world,
                                  "allreduce_ms_per_epoch": (sum(ar_ms) / len(ar_ms)) if ar_ms else None, "allreduces_timed": len(ar_ms),
                                  "allreduce_floats": int(updater.numel) if updater is not None else None,
                                  "collectives_per_iteration": n_collectives / max(1, args.steps),
                                  "per_iteration": "sampling pass: 1 all-gather (LoggerRL totals + observation-filter deltas, merged on the "
                                                   "device); update: 1 MAX (padded window length) + 1 all-gather (advantage moments + sample "
                                                   "counts, 5 x float64, Chan-merged on the device) + %d gradient all-reduces" % cfg.num_optim_epoch}
    tr.close()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0 and world == 1:
        if not args.no_kernels:
            try:
                res["kernels"] = {"what": "inputs resident in HBM, HIP events, float64, algorithmic bytes of SURVEY 8(d); us includes "
                                          "the Python/ctypes call (~8 us)", "peak_GBps": HBM_PEAK / 1e9,
                                  "rows": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
                                           for r in kernel_microbench((1024, 65536), device_index=local)]}
            except Exception as e:
                res["kernels"] = {"error": repr(e)[:300]}
        if not args.no_legs and args.task == "egomimic":
            mk32 = lambda: Trainer(cfg_cls(args.cfg, create_dirs=False), dev, torch.float32, num_envs=args.envs, num_threads=n_threads,
                                   num_groups=args.groups)

            mk64 = lambda: Trainer(cfg_cls(args.cfg, create_dirs=False), dev, torch.float64, num_envs=args.envs, num_threads=n_threads,
                                   num_groups=args.groups, plain_optim=True)
            ev, lw = args.k1_event_every, args.leg_warmup
            legs = {
                "dropin_float64_driver": run_leg(mk64, args.leg_steps, lw, min_batch, ev, default_dtype=torch.float64),
                "changing_inertia_host_fed": run_leg(mk32, args.leg_steps, lw, min_batch, ev, {"EGP_SURROGATE_ALWAYS_DIRTY": "1"}),
                "changing_inertia_device_dynamics": run_leg(mk32, args.leg_steps, lw, min_batch, ev,
                                                            {"EGP_SURROGATE_ALWAYS_DIRTY": "1", "EGP_DEVICE_DYNAMICS": "1"}),
            }
            # BASELINE config 5's nets on this GPU's shard (1 of the 4 x 1 024 env slots): VideoForecastNet front ends,
            # per-tick state LSTM, 90-step episodes, decayed reward
            mkf = lambda: Trainer(ForecastConfig(args.cfg, create_dirs=False), dev, torch.float32, num_envs=args.envs,
                                  num_threads=n_threads, num_groups=args.groups)
            legs["egoforecast_config5_shard"] = run_leg(mkf, args.leg_steps, lw, min_batch, ev)
            # the same pipeline when a physics substep costs what a real simulator's does (the surrogate's own is ~0.3 us;
            # an mj_step of this humanoid is tens of us): host-bound, and the CPU sampler beside it at the same cost
            cost = {"EGP_SURROGATE_SUBSTEP_US": str(args.sim_cost_us)}
            leg = run_leg(mk32, args.leg_steps, min(2, lw), min_batch, ev, cost)
            if not args.no_cpu_baseline and "error" not in leg:
                cb = cpu_baseline(root, max(200, args.cpu_steps // 6), 2, cost)
                leg.update({"cpu_sampler_env_steps_per_s": cb["value"], "cpu_sampler_cores": cb["cores"], "cpu_sampler_sample": cb["sample"]})
            leg["substep_cost_us"] = args.sim_cost_us
            if "error" not in leg:       # what the host threads alone allow: one env-step = substeps x cost on one thread
                leg["host_threads"] = n_threads
                leg["host_physics_ceiling_env_steps_per_s"] = n_threads / (leg["substeps_per_launch"] * args.sim_cost_us * 1e-6)
                leg["frac_of_host_physics_ceiling"] = leg["rollout_only_env_steps_per_s"] / leg["host_physics_ceiling_env_steps_per_s"]
            legs["simulator_cost_per_substep"] = leg
            # the reference's loop condition applied to all slots together (no new episode once the batch is there) against the
            # per-slot quota of the default: what the rollout's latency-bound tail costs, and what it buys in batch size
            legs["step_budget_global"] = run_leg(mk32, args.leg_steps, lw, min_batch, ev, {"EGP_STEP_BUDGET": "global"})
            # more env slots on the one GPU, same physics, the batch scaled with the slots (each slot keeps its 48-step quota): how far
            # the fixed latencies of a tick amortise -- the basis of the weak-scaling claim (1 024 slots = the headline; the resident
            # K1 serves them all: beyond 4 envs per CU a wave serves 2 or 4 envs in turn)
            sweep = {}
            for n_slots, n_groups in ((512, args.groups), (2048, args.groups), (4096, args.groups), (4096, 2 * args.groups)):
                mkn = lambda n_slots=n_slots, n_groups=n_groups: Trainer(cfg_cls(args.cfg, create_dirs=False), dev, torch.float32, num_envs=n_slots,
                                                                         num_threads=n_threads, num_groups=n_groups)
                r = run_leg(mkn, args.leg_steps, lw, min_batch * n_slots // args.envs, ev)
                key = "slots_%d_groups_%d" % (n_slots, n_groups)
                if "error" in r:
                    sweep[key] = {"error": r["error"]}
                else:
                    sweep[key] = {"env_steps_per_s": r["env_steps_per_s"], "rollout_only_env_steps_per_s": r["rollout_only_env_steps_per_s"],
                                  "ms_sample_update": [round(1e3 * r["t_sample_s"] / r["steps"], 1), round(1e3 * r["t_update_s"] / r["steps"], 1)],
                                  "ms_sample_update_per_iteration": r.get("per_iteration_ms_sample_update"), "steps": r["steps"], "warmup": r["warmup"],
                                  "env_steps_per_iteration": r["env_steps_per_iteration"], "substeps_per_launch": r["substeps_per_launch"],
                                  "envs_per_wave": r.get("envs_per_wave")}
            legs["envs_per_gpu_sweep"] = {k: json.dumps(v) for k, v in sweep.items()}
            # BASELINE config 4: the state regressor's optimisation step (ResNet-18 encoder in bf16 on the matrix cores)
            try:
                from egopose_amd.bench_support import statereg_config4
                legs["statereg_config4"] = statereg_config4(local)
            except Exception as e:
                legs["statereg_config4"] = {"error": repr(e)[:300]}
            res["legs"] = {k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()} for k, v in legs.items()}
            res["legs"]["envs_per_gpu_sweep"] = {k: json.loads(v) for k, v in res["legs"]["envs_per_gpu_sweep"].items()}
            res["legs"]["envs_per_gpu_sweep"]["slots_%d_groups_%d" % (args.envs, args.groups)] = {"env_steps_per_s": res["value"], "headline": True}
            res["dropin_env_steps_per_s"] = legs["dropin_float64_driver"].get("env_steps_per_s")
        if not args.no_cpu_baseline:
            # BASELINE config 1 / B1: 2 sampling workers (+ the CPU update on a bounded part of their sample); B2: as many
            # workers as this box's cores allow (the same bounded sample, so about nproc / 2 times shorter)
            res["cpu_baseline"] = cpu_baseline(root, args.cpu_steps, 2, update_steps=args.cpu_update_steps)
            nproc = max(2, cores - 2)
            res["cpu_baseline_nproc"] = cpu_baseline(root, args.cpu_steps * 2, nproc)
            for k in ("cpu_baseline", "cpu_baseline_nproc"):
                if res[k].get("value"):
                    res[k]["gpu_over_cpu_rollout"] = res["rollout_only_env_steps_per_s"] / res[k]["value"]
    if rank == 0:
        res["config"] = order_config(res["config"], res.get("legs"))
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
