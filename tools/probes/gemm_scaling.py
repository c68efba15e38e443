"""How the persistent split-operand GEMM scales with the tile count of weight-gradient (rcrc, split-K) and forward (kckc) shapes:
time per launch, per k-tile and workgroup, fp32-equivalent TF/s. Usage: python tools/probes/gemm_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd import gemm as G

dev = torch.device("cuda")
K = 131072


def timeit(f, it=30):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


print("weight gradients  dW[M][N] = dy[K][M]^T x[K][N], K = %d" % K)
for M, N in [(128, 128), (128, 256), (256, 128), (256, 256), (384, 256), (300, 244), (200, 301), (512, 128), (300, 128)]:
    dy = torch.randn(K, M, device=dev); x = torch.randn(K, N, device=dev)
    for splits in (None,):
        sp = G.pick_splits(M, N, K)
        us = timeit(lambda: G.gemm(dy, x, False, False, splits=sp, want_bias_grad=False))
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        ktiles = K / 32 / sp
        print("  M %4d N %4d: %2d tiles x %3d splits = %3d items, %7.1f us, %5.2f us per k-tile, %6.1f TF/s, HBM min %5.1f us" % (
            M, N, tiles, sp, tiles * sp, us, us / ktiles, 2.0 * M * N * K / us / 1e6, (M + N) * K * 4 / 4.5e6))
print("forward  y[K][N] = x[K][Kin] W[N][Kin]^T")
for Kin, N in [(256, 128), (256, 256), (243, 300), (300, 200), (128, 512), (128, 1024), (512, 128), (1024, 128)]:
    x = torch.randn(K, Kin, device=dev); W = torch.randn(N, Kin, device=dev); b = torch.randn(N, device=dev)
    us = timeit(lambda: G.linear_fwd(x, W, b, True))
    tiles = (K // 128) * ((N + 127) // 128)
    per_wg = tiles / 256.0 * ((Kin + 31) // 32)
    print("  Kin %4d N %4d: %5d tiles, %7.1f us, %5.2f us per k-tile and workgroup, %6.1f TF/s, HBM min %5.1f us" % (
        Kin, N, tiles, us, us / per_wg, 2.0 * K * Kin * N / us / 1e6, (Kin + N) * K * 4 / 4.5e6))
