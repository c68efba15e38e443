"""Does a rollout slow down when float32-class products run next to it? (the critic's epochs under the next
rollout.) A queue of products of the 4-wave kernel (k_gemm_bf16x<64, 6>: <= 160 VGPRs, fits beside a resident K1 wave on a SIMD) is
enqueued on a side stream -- plain, or created with a CU mask -- right before `sample`; reported: T_sample and when the queue drained.
    EGP_GEMM_WS=0 python tools/probes/corun_probe.py [n_products] [masked CUs]"""
import ctypes, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.train import Trainer
from egopose_amd.physics import default_threads
from egopose_amd import gemm as G
n_prod = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 192
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_corun_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
tr.agent.prefetch_rollout = False
for it in range(3):
    tr.iteration(it, cfg.min_batch_size)
x = torch.randn(134000, 300, device=dev); W = torch.randn(64, 300, device=dev); b = torch.zeros(64, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(n):
    words = (ctypes.c_uint32 * 8)(*[0] * 8)
    for c in range(n):                      # CU c of the device-wide numbering (XCDs interleaved by the runtime)
        words[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
def priority_stream(lowest=True):
    lo, hi = ctypes.c_int(), ctypes.c_int()
    assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, lo.value if lowest else hi.value) == 0        # 1 = hipStreamNonBlocking
    print("stream priorities: least %d greatest %d -> side stream at %d" % (lo.value, hi.value, lo.value if lowest else hi.value))
    return torch.cuda.ExternalStream(s.value)
def one_product_us():
    torch.cuda.synchronize(); a, e = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(20): G.linear_fwd(x, W, b)
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / 20 * 1e3
print("one product alone: %.1f us" % one_product_us())
if os.environ.get("CORUN_MAIN_HIGH", "0") == "1":
    torch.cuda.set_stream(priority_stream(False))
    print("the rollout's own stream: highest priority")
def run(label, stream):
    rows = []
    for rep in range(4):
        torch.cuda.synchronize()
        done = torch.cuda.Event(True); beg = torch.cuda.Event(True)
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                beg.record()
                for _ in range(n_prod): G.linear_fwd(x, W, b)
                done.record()
        t0 = time.time()
        batch, log = tr.agent.sample(cfg.min_batch_size)
        ts = time.time() - t0
        torch.cuda.synchronize()
        rows.append((ts * 1e3, beg.elapsed_time(done) if stream is not None else 0.0, log.num_steps))
    print("%-28s T_sample %s ms | side queue drained after %s ms" % (label, " ".join("%.1f" % r[0] for r in rows), " ".join("%.1f" % r[1] for r in rows)))
run("alone", None)
run("side stream, all CUs", torch.cuda.Stream(dev))
run("side stream, lowest priority", priority_stream(True))
run("side stream, %d CUs" % n_cu, masked_stream(n_cu))
run("alone again", None)
