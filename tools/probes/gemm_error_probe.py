import sys; sys.path.insert(0,'.')
import torch
from egopose_amd.gemm import gemm
g=torch.Generator(device='cuda').manual_seed(1)
for (M,N,K) in [(1000,300,243),(300,243,4097),(2000,1024,128)]:
    A=torch.randn(M,K,device='cuda',generator=g); B=torch.randn(N,K,device='cuda',generator=g)
    ref=A.double()@B.double().t(); scale=(A.double().abs()@B.double().abs().t())
    for name,c in (('lib',A@B.t()),('x6',gemm(A,B,terms=6)),('x3',gemm(A,B,terms=3)),('x1',gemm(A,B,terms=1))):
        e=(c.double()-ref).abs()
        print(M,N,K,name,'rel %.2e'%float((c.double()-ref).norm()/ref.norm()),'max err/scale %.2e'%float((e/scale).max()))
