"""What a dependency between two small kernels costs on this box: same stream (in order) vs across streams through an event.
All launches are enqueued ahead of the GPU (a long spin kernel holds the queues back), so the numbers are GPU-side only."""
import time
import torch

dev = torch.device("cuda", 0)
x = torch.zeros(512 * 115, device=dev, dtype=torch.float64)
y = torch.zeros_like(x)
hold = torch.zeros(64 << 20, device=dev)


def spin():                       # ~ a millisecond of work in front of the measured chain
    for _ in range(4):
        hold.mul_(1.0)


def chain_same(n):
    s = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        spin()
        e0.record()
        for _ in range(n):
            x.add_(1.0)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def chain_cross(n, timing=False):
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        spin()
        e0.record()
    cur, other = a, b
    for i in range(n):
        ev = torch.cuda.Event(enable_timing=timing)
        with torch.cuda.stream(cur):
            (x if i % 2 == 0 else y).add_(1.0)
            ev.record()
        other.wait_event(ev)
        cur, other = other, cur
    with torch.cuda.stream(cur):
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for _ in range(2):
    print("same stream, in order      : %.2f us per kernel" % chain_same(200))
    print("alternating streams + event: %.2f us per kernel" % chain_cross(200))
    print("   (events with timing)    : %.2f us per kernel" % chain_cross(200, True))
# one kernel alone, host-synchronised: launch + completion latency as the host sees it
torch.cuda.synchronize()
t0 = time.time()
for _ in range(200):
    x.add_(1.0)
    torch.cuda.synchronize()
print("launch + synchronize       : %.2f us" % ((time.time() - t0) / 200 * 1e6))
