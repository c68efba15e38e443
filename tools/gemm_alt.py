import torch, time
T,B=220,2233
d=torch.randn(T,B,256,device="cuda"); x=torch.randn(T,B,128,device="cuda"); h=torch.randn(T,B,64,device="cuda")
d2=d.view(T*B,256); x2=x.view(T*B,128); h2=h.view(T*B,64)
def tm(fn,it=10):
    fn(); torch.cuda.synchronize(); t=time.time()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.time()-t)/it*1e3
ref=d2.t().mm(x2)
print("d2.t().mm(x2)      ", round(tm(lambda: d2.t().mm(x2)),3))
print("(x2.t().mm(d2)).t()", round(tm(lambda: x2.t().mm(d2).t()),3))
print("bmm over T + sum   ", round(tm(lambda: torch.bmm(d.transpose(1,2), x).sum(0)),3), (torch.bmm(d.transpose(1,2), x).sum(0)-ref).abs().max().item())
for ch in (10, 20, 44):
    dd=d.view(ch, T//ch*B, 256); xx=x.view(ch, T//ch*B, 128)
    print("bmm chunks", ch, round(tm(lambda: torch.bmm(dd.transpose(1,2), xx).sum(0)),3))
xh=torch.cat((x2,h2),1)
print("cat[x,h] one gemm  ", round(tm(lambda: d2.t().mm(torch.cat((x2,h2),1))),3))
print("d2.t().mm(h2)      ", round(tm(lambda: d2.t().mm(h2)),3))
print("d2.sum(0)          ", round(tm(lambda: d2.sum(0)),3))
ones=torch.ones(T*B,1,device="cuda")
print("gx proj addmm      ", round(tm(lambda: torch.addmm(torch.zeros(256,device='cuda'), x2, torch.randn(128,256,device='cuda'))),3))
