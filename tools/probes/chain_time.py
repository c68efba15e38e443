"""Forward + backward of the update's policy / value heads at the bench's batch size: chained launches vs one launch per layer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ["EGP_MLP_CHAIN"] = "1"
from egopose_amd import gemm as G, chain as CH
from egopose_amd.nets import MLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 134656
torch.manual_seed(0)
R = n // 200 * 220 + 2000
ctx2d = torch.randn(max(R, n), 128, device="cuda")
idx = torch.randperm(ctx2d.shape[0], device="cuda")[:n].contiguous()
x = torch.randn(n, 115, device="cuda")
for n_out in (52, 1):
    mlp, head = MLP(243, (300, 200), "relu").cuda(), torch.nn.Linear(200, n_out).cuda()
    dout = torch.randn(n, n_out, device="cuda")
    for name, fn in (("chain", CH.chain_mlp_head), ("layers", G.gather_mlp_head_layers)):
        res = {}
        for phase in ("fwd", "fwd+bwd"):
            ts = []
            for it in range(8):
                gi = G.GatheredInput(ctx2d.clone().requires_grad_(True), idx, x)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = fn(gi, mlp.affine_layers, head)
                if phase != "fwd":
                    out.backward(dout)
                b.record(); b.synchronize()
                ts.append(a.elapsed_time(b))
            res[phase] = sorted(ts[2:])[len(ts[2:]) // 2]
        ts = []
        with torch.no_grad():
            for it in range(8):
                gi = G.GatheredInput(ctx2d, idx, x)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); out = fn(gi, mlp.affine_layers, head); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b))
        print("n_out %2d  %-6s  fwd %.3f ms   fwd+bwd %.3f ms   no-grad fwd %.3f ms" % (n_out, name, res["fwd"], res["fwd+bwd"], sorted(ts[2:])[3]))
