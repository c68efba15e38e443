#!/usr/bin/env python3
"""Group a rocprofv3 `*kernel_stats.csv` (bench.py run) by what the kernels belong to. Usage: classify_kernel_stats.py CSV [iterations]"""
import collections
import csv
import sys


def cls(n):
    if "k_probe_" in n: return "host probe (egp_host_probe: outside the timed region)"
    if "k_pd_server" in n or "k_pd_torque" in n: return "K1 (stable PD)"
    if "k_lstm" in n: return "LSTM recurrences (HIP)"
    if "k_gemm" in n or "k_gemv_rows" in n or "k_rank1" in n or "k_colsum" in n: return "GEMM (HIP, split bf16 MFMA + thin float32 products)"
    if n.startswith("Cijk") or "rocblas" in n.lower(): return "GEMM (library)"
    if "k_policy" in n: return "policy step (HIP)"
    if "k_ppo_loss" in n or "k_adam" in n or "k_sqnorm" in n: return "update tail (HIP: losses, clip + Adam)"
    if "egp::" in n or "k_engine" in n or "k_zf" in n: return "other egp kernels (K2-K6, engine)"
    if "copyBuffer" in n or "fillBuffer" in n: return "copy / fill"
    if "Adam" in n or "multi_tensor" in n: return "optimizer"
    return "torch elementwise / reduce / index"


def main():
    rows = list(csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#")))
    iters = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    acc, calls = collections.Counter(), collections.Counter()
    for r in rows:
        c = cls(r["Name"])
        acc[c] += float(r["TotalDurationNs"])
        calls[c] += int(r["Calls"])
    tot = sum(acc.values())
    print("total kernel time %.1f ms over %g iterations" % (tot / 1e6, iters))
    for k, v in acc.most_common():
        print("%-38s %8.1f ms %7d calls  %6.1f ms / iteration" % (k, v / 1e6, calls[k], v / 1e6 / iters))
    print("\ntop kernels outside K1:")
    for r in [r for r in rows if "k_pd_server" not in r["Name"]][:22]:
        print("%8.2f ms %6s  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], r["Name"][:130]))


if __name__ == "__main__":
    main()
