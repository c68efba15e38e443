"""bench.py --gpus N starts its own ranks (one process per GPU) when it is not under torchrun; here with the
gloo backend and no GPU work (--dry-run): rendezvous, barrier-bracketed timing, sum / max aggregation over ranks and
the rank-0-only JSON line. A mismatch between --gpus and the environment must be refused, never reported as n_gpus 1."""
import json
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(HIP_VISIBLE_DEVICES="", EGP_DIST_BACKEND="gloo", **kw)
    return env


def test_bench_launches_its_own_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "3"], env=_env(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["dry_run"] is True
    assert r["env_steps"] == 100 * 3 * (1 + 2)                      # sum over ranks
    assert r["ms_per_step"] * 3 >= 40.0 * 0.9                         # max over ranks (rank 1 sleeps 40 ms)
    assert abs(r["value"] - r["env_steps"] / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]


def test_bench_refuses_a_world_size_mismatch():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], env=_env(WORLD_SIZE="1"), capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and "{" not in out.stdout
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], env=_env(WORLD_SIZE="2", RANK="0", MASTER_PORT="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "{" not in out.stdout


def test_bench_refuses_more_ranks_than_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s)" in out.stderr and "{" not in out.stdout


def test_a_failing_rank_stops_the_launch():
    """Rank 1 dies before the rendezvous (bad MASTER_PORT handling is not needed: an argparse error is enough)."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "x"], env=_env(), capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "{" not in out.stdout


import pytest


@pytest.mark.gpu
def test_two_ranks_through_the_launcher_on_a_gpu():
    """The whole multi-rank path on real hardware: bench.py starts two ranks itself; with one GPU on the box they share it
    and exchange through gloo (EGP_BENCH_SHARE_GPU=1: a self-test, not a measurement) -- env shards per rank, advantage /
    filter / logger moments merged, one flat gradient all-reduce per epoch, rank-0 line with the summed env-steps."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(EGP_BENCH_SHARE_GPU="1", EGP_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--envs", "128", "--min-batch", "2048",
                          "--threads", "3"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
    assert r["env_steps"] >= 2 * 2048 and r["value"] > 0 and r["roofline"] is not None
    assert "legs" not in r and "cpu_baseline" not in r          # 1-GPU extras only
