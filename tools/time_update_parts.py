import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.nets import MLP, PolicyGaussian, Value, VideoStateNet, RNN
dev = "cuda"
torch.manual_seed(0)
T, B, D, N = 220, 2233, 128, 133000
def tm(fn, it=5):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.time() - t) / it * 1e3
rnn = RNN(D, 128, "lstm", bi_dir=True).to(dev)
x = torch.randn(T, B, D, device=dev)
with torch.no_grad():
    print("bi-LSTM fwd no_grad  ms", round(tm(lambda: rnn(x)), 2))
def fb():
    rnn.zero_grad(); y = rnn(x); y.sum().backward()
print("bi-LSTM fwd+bwd      ms", round(tm(fb), 2))
val = Value(MLP(243, [300, 200], "relu")).to(dev)
xs = torch.randn(N, 243, device=dev)
def mfb():
    val.zero_grad(); val(xs).sum().backward()
print("MLP value fwd+bwd    ms", round(tm(mfb), 2))
with torch.no_grad():
    print("MLP value fwd        ms", round(tm(lambda: val(xs)), 2))
idx = torch.randint(0, 200 * B, (N,), device=dev)
ctx = torch.randn(200 * B, 128, device=dev, requires_grad=True)
def gfb():
    ctx.grad = None; ctx.index_select(0, idx).sum().backward()
print("gather fwd+bwd       ms", round(tm(gfb), 2))
# packed / truncated variants: forward direction over 70 frames only
x70 = x[:70].contiguous()
def fb70():
    rnn.zero_grad(); y = rnn(x70); y.sum().backward()
print("bi-LSTM fwd+bwd T=70 ms", round(tm(fb70), 2))
