"""Timeline of the rollout's per-tick GPU chain from a rocprofv3 --kernel-trace CSV:
    resident K1 (env-step k) -> K3+K6 partial -> K6 apply -> policy (tick k+1) -> next resident K1
prints, per phase of the rollout (by tick index), the median duration of each link and of the gaps between them.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python tools/probes/sample_time.py 3
    python tools/probes/chain_gaps.py gpurun_out/trace
"""
import csv, glob, os, sys
import numpy as np

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
assert files, "no kernel_trace.csv under " + root
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "0")))
rows.sort()
names = sorted({r[2].split("(")[0][:60] for r in rows})
print("%d kernel records; distinct kernels: %d" % (len(rows), len(names)))


def kind(n):
    if "k_pd_server" in n:
        return "K1"
    if "k_policy" in n:
        return "policy"
    if "zf" in n and "partial" in n or "k_post_step" in n:
        return "zf1"
    if "zf" in n and ("apply" in n or "merge" in n):
        return "zf2"
    if "k_reward" in n:
        return "K2"
    return None


ev = [(s, e, kind(n), n) for s, e, n, _ in rows if kind(n)]
print("kinds:", {k: sum(1 for x in ev if x[2] == k) for k in ("K1", "policy", "zf1", "zf2", "K2")})
# the chain as the stream sees it: zf1 -> zf2 -> policy (consecutive on the rollout's stream)
chains = []
i = 0
seq = [x for x in ev if x[2] in ("zf1", "zf2", "policy")]
while i + 2 < len(seq):
    a, b, c = seq[i], seq[i + 1], seq[i + 2]
    if (a[2], b[2], c[2]) == ("zf1", "zf2", "policy"):
        chains.append((a, b, c))
        i += 3
    else:
        i += 1
k1 = [x for x in ev if x[2] == "K1"]
k1_starts = np.array([x[0] for x in k1])
k1_ends = np.array(sorted(x[1] for x in k1))
out = []
for a, b, c in chains:
    # the env-step this chain follows: the latest K1 that ended before zf1 started; the one it feeds: first K1 starting after the policy ended
    j = np.searchsorted(k1_ends, a[0], side="right") - 1
    n = np.searchsorted(k1_starts, c[1], side="left")
    pre_gap = (a[0] - k1_ends[j]) / 1e3 if j >= 0 else np.nan
    nxt_gap = (k1_starts[n] - c[1]) / 1e3 if n < len(k1_starts) else np.nan
    out.append(((a[1] - a[0]) / 1e3, (b[0] - a[1]) / 1e3, (b[1] - b[0]) / 1e3, (c[0] - b[1]) / 1e3, (c[1] - c[0]) / 1e3, pre_gap, nxt_gap,
                (c[1] - a[0]) / 1e3))
out = np.array(out)
print("%d chains" % len(out))
hdr = ("zf1", "gap", "zf2", "gap", "policy", "K1end->zf1", "policy->K1", "zf1->policy end")
print("median us over all chains: " + "  ".join("%s %.1f" % (h, v) for h, v in zip(hdr, np.nanmedian(out, 0))))
q = len(out) // 4
for p in range(4):
    seg = out[p * q:(p + 1) * q]
    print("quarter %d: " % p + "  ".join("%s %.1f" % (h, v) for h, v in zip(hdr, np.nanmedian(seg, 0))))
d = np.array([(x[1] - x[0]) / 1e3 for x in k1])
print("K1 resident launches: %d, median %.1f us, mean %.1f us, sum %.1f ms" % (len(d), np.median(d), d.mean(), d.sum() / 1e3))
