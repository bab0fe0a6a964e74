import sys; sys.path.insert(0,'.')
import torch, numpy as np
from msmd_amd import ops
torch.manual_seed(0)
for td in (torch.float32, torch.bfloat16):
    for (M,N,K) in ((64,64,32),(64,64,64),(128,128,128),(256,256,256)):
        a=torch.randn(M,K,device='cuda').to(td); w=torch.randn(N,K,device='cuda').to(td)
        ref=(a.float()@w.float().t())
        out=ops.gemm(a,w).float()
        torch.cuda.synchronize()
        err=(out-ref).abs().max().item()
        print(td,M,N,K,'err',err)
        if err>0.1 and M==64 and K==32:
            # identity probe
            a=torch.zeros(M,K,device='cuda',dtype=td); 
            for i in range(min(M,K)): a[i,i]=1
            w=torch.arange(N*K,device='cuda',dtype=torch.float32).reshape(N,K).to(td)/100
            out=ops.gemm(a,w).float(); ref=a.float()@w.float().t()
            print('out[:4,:8]',out[:4,:8]); print('ref[:4,:8]',ref[:4,:8])
            print('out[16:20,:8]',out[16:20,:8]); print('ref',ref[16:20,:8])
