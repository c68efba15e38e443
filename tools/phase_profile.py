"""Device kernels of ONE rollout and ONE update of the bench workload, separately (torch profiler over the two calls):
name, calls, total device time, classified as tools/classify_kernel_stats.py does. Writes gpurun_out/phase_profile.txt.

    python tools/phase_profile.py [--dtype float64]      (float64 = the drop-in driver's master/shadow set-up)
"""
import argparse
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch                                                    # noqa: E402
from torch.profiler import profile, ProfilerActivity            # noqa: E402
from classify_kernel_stats import cls                           # noqa: E402
from egopose_amd.bench_support import write_synthetic_dataset   # noqa: E402
from egopose_amd.config import Config                           # noqa: E402
from egopose_amd.physics import default_threads                 # noqa: E402
from egopose_amd.train import Trainer                           # noqa: E402


def table(prof, title, out):
    rows = [(e.key, e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
    if not rows:
        rows = [(e.key, e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0]
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    out.append("== %s: device time %.2f ms in %d kernel names, %d launches" % (title, tot, len(rows), sum(r[1] for r in rows)))
    acc = {}
    for k, c, ms in rows:
        a = acc.setdefault(cls(k), [0.0, 0])
        a[0] += ms
        a[1] += c
    for k, (ms, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        out.append("   %-52s %8.2f ms %6d launches" % (k, ms, c))
    out.append("")
    for k, c, ms in rows:
        out.append("%9.3f ms %6d  [%s] %s" % (ms, c, cls(k)[:12], k[:150]))
    out.append("")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "phase_profile.txt"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    root = tempfile.mkdtemp(prefix="egp_pp_")
    write_synthetic_dataset(root, "subject_03", device_index=0)
    os.chdir(root)
    cfg = Config("subject_03", create_dirs=False)
    tr = Trainer(cfg, dev, getattr(torch, args.dtype), num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
    for it in range(2):
        tr.iteration(it, cfg.min_batch_size)
    torch.cuda.synchronize()
    out = []
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        batch, log = tr.agent.sample(cfg.min_batch_size)
        torch.cuda.synchronize()
    table(prof, "rollout (Agent.sample, %d env-steps)" % log.num_steps, out)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        t = tr.agent.update_params(batch)
        torch.cuda.synchronize()
    table(prof, "update (AgentEgo.update_params, %.1f ms wall under the profiler)" % (t * 1e3), out)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out[:60]))
    tr.close()


if __name__ == "__main__":
    main()
