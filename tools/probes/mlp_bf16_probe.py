"""The update's two MLPs (243 -> 300 -> 200 -> 52 | 1, ReLU) forward + backward at the bench batch (139 264 rows):
float32 library GEMMs (today) against bfloat16 GEMMs with float32 master weights (torch.autocast, and hand-cast operands).
Prints ms per forward+backward and the worst relative error of the parameter gradients against float64.
Usage: mlp_bf16_probe.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.nets import MLP, PolicyGaussian, Value

n = int(sys.argv[1]) if len(sys.argv) > 1 else 139264
dev = "cuda"
torch.manual_seed(0)
pol = PolicyGaussian(MLP(243, [300, 200], "relu"), 52, log_std=-2.3, fix_std=True).to(dev)
val = Value(MLP(243, [300, 200], "relu")).to(dev)
x = torch.randn(n, 243, device=dev)
a = torch.randn(n, 52, device=dev) * 0.1
ret = torch.randn(n, 1, device=dev)


def step(mode):
    for m in (pol, val):
        for p in m.parameters():
            p.grad = None
    if mode == "fp32":
        mean, std = pol.mean_std(x)
        v = val(x)
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            mean, std = pol.mean_std(x)
            v = val(x)
        mean, v = mean.float(), v.float()
    loss = ((a - mean) / std).pow(2).sum() / n + (v - ret).pow(2).sum() / n
    loss.backward()
    return loss


def bench(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# float64 reference gradients
pol64, val64 = __import__("copy").deepcopy(pol).double(), __import__("copy").deepcopy(val).double()
m64, s64 = pol64.mean_std(x.double())
l64 = ((a.double() - m64) / s64).pow(2).sum() / n + (val64(x.double()) - ret.double()).pow(2).sum() / n
l64.backward()
ref = [p.grad for p in list(pol64.parameters()) + list(val64.parameters()) if p.grad is not None]
for mode in ("fp32", "bf16-autocast"):
    t = bench(lambda: step(mode))
    step(mode)
    got = [p.grad for p in list(pol.parameters()) + list(val.parameters()) if p.grad is not None]
    err = max(float((g.double() - r).norm() / r.norm()) for g, r in zip(got, ref))
    print("%-14s %.2f ms per fwd+bwd of both MLPs, worst relative gradient error (per tensor, 2-norm) %.2e" % (mode, t, err))
