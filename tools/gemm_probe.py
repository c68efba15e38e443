"""Time the update's MLP GEMM shapes (fp32) with the default BLAS pick vs torch TunableOp. Usage: gemm_probe.py [rows]"""
import sys, time, os
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 139264
dev = "cuda"
dims = [(243, 300), (256, 300), (300, 200), (304, 200), (300, 208), (200, 52), (200, 1)]


def bench(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(tag):
    tot = 0
    for (k, m) in dims:
        x = torch.randn(n, k, device=dev); w = torch.randn(m, k, device=dev); b = torch.randn(m, device=dev)
        gy = torch.randn(n, m, device=dev)
        t_f = bench(lambda: torch.addmm(b, x, w.t()))
        t_d = bench(lambda: gy @ w)
        t_w = bench(lambda: gy.t() @ x)
        fl = 2 * n * k * m
        print("%s  %3dx%3d  fwd %7.1f us (%5.1f TF/s)  dgrad %7.1f us (%5.1f)  wgrad %7.1f us (%5.1f)" % (tag, k, m, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6))
        tot += t_f + t_d + t_w
    print(tag, "total us", round(tot))


run("default ")
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(20)
tn.set_filename(os.environ.get("TUNE_OUT", "/tmp/tunable.csv"))
t0 = time.time(); run("tuning  "); print("tuning took", round(time.time() - t0, 1), "s")
run("tuned   ")
