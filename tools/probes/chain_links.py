"""The default tick chain from a rocprofv3 kernel trace (tools/probes/chain_gaps.sh): per tick without resets
    K1 (env-step k, group's stream) end -> filter statistics -> policy step with the apply pass -> K1 (env-step k + 1) start
median duration of every link and gap, and their sum per tick."""
import csv, glob, os, sys
import numpy as np
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
K1 = [(s, e) for s, e, n in rows if "k_pd_server" in n]
pol = [(s, e) for s, e, n in rows if "k_policy_gaussian" in n and ", true>(" in n]
zf1 = [(s, e) for s, e, n in rows if "k_zf_partial" in n]
k1_starts = np.array([s for s, _ in K1]); k1_ends = np.array(sorted(e for _, e in K1)); zf_ends = np.array([e for _, e in zf1]); zf_starts = np.array([s for s, _ in zf1])
out = []
for ps, pe in pol:
    i = np.searchsorted(zf_ends, ps, side="right") - 1            # the statistics kernel right in front of this policy step
    j = np.searchsorted(k1_starts, pe)                            # the env-step kernel it feeds
    if i < 0 or j >= len(K1):
        continue
    zs, ze = zf1[i]
    if ps - ze > 40e3 or k1_starts[j] - pe > 60e3:
        continue
    m = np.searchsorted(k1_ends, zs, side="right") - 1            # the env-step whose end released the statistics kernel (latest K1 end before it)
    if m < 0:
        continue
    out.append(((zs - k1_ends[m]) / 1e3, (ze - zs) / 1e3, (ps - ze) / 1e3, (pe - ps) / 1e3, (k1_starts[j] - pe) / 1e3))
a = np.array(out)
names = ["K1 end -> stats", "stats", "stats -> policy", "policy (+apply)", "policy -> K1"]
print("%d ticks matched" % len(a))
print("median us: " + "  ".join("%s %.1f" % (n, v) for n, v in zip(names, np.median(a, 0))) + "   sum %.1f" % np.median(a.sum(1)))
print("(K1 end -> stats is a lower bound of that link when the two groups' env-steps overlap: the latest K1 end may be the other group's)")
