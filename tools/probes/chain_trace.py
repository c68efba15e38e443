"""Phase stamps of one workgroup's first tile of the chained MLP launches (library built with EGP_BUILD_DEFS=-DEGP_CHAIN_TRACE=<wg>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["EGP_MLP_CHAIN"] = "1"
from egopose_amd import gemm as G, chain as CH, _lib
from egopose_amd.nets import MLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 134656
torch.manual_seed(0)
ctx2d = torch.randn(n + 2000, 128, device="cuda")
idx = torch.randperm(ctx2d.shape[0], device="cuda")[:n].contiguous()
x = torch.randn(n, 115, device="cuda")
mlp, head = MLP(243, (300, 200), "relu").cuda(), torch.nn.Linear(200, 52).cuda()
dout = torch.randn(n, 52, device="cuda")
lib = _lib.load()
lib.egp_chain_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["product 1", "epilogue 1", "(zero acc2)", "product 2", "epilogue 2 (+zero)", "product 3", "output"]
def read(tag):
    out = np.zeros(32, np.int64)
    assert lib.egp_chain_trace_read(out.ctypes.data, 1) == 0
    t = np.diff(out[:7]) / 100.0
    print(tag, " ".join("%s %.2f" % (nm, v) for nm, v in zip(["p1", "ep1", "p2", "ep2", "p3", "out"], t)), " tile %.2f us;  stage-boundary wait of all tiles: %d cycles" % ((out[6] - out[0]) / 100.0, out[16]))
for rep in range(3):
    gi = G.GatheredInput(ctx2d.clone().requires_grad_(True), idx, x)
    lib.egp_chain_trace_read(np.zeros(32, np.int64).ctypes.data, 1)
    out = CH.chain_mlp_head(gi, mlp.affine_layers, head)
    torch.cuda.synchronize()
    read("forward ")
    out.backward(dout)
    torch.cuda.synchronize()
    read("backward")
