"""Times one batched refresh of remembered weight conversions (weightprep._refresh) and checks it against torch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import weightprep, ops
from torch import nn
torch.manual_seed(0)
ps=[nn.Parameter(torch.randn(r,c,device='cuda')) for (r,c) in [(64,64),(256,64),(64,256),(128,128),(512,128),(320,320),(1280,320),(320,1280),(512,512),(2048,512),(512,2048),(100,36),(33,65)]*6]
outs=[weightprep.transpose(p,p.shape[0],p.shape[1],torch.bfloat16, reverse_cols=(i%5==0)) for i,p in enumerate(ps)]
outs2=[weightprep.cast(p,torch.bfloat16) for p in ps]
for p in ps: p.data.mul_(2.0)
weightprep.bump()
o=weightprep.transpose(ps[0],64,64,torch.bfloat16,reverse_cols=True)
torch.cuda.synchronize()
for i,p in enumerate(ps):
    ref=p.detach().t().contiguous() if i%5 else p.detach().flip(1).t().contiguous()
    assert torch.equal(outs[i], ref.to(torch.bfloat16)), i
    assert torch.equal(outs2[i], p.detach().to(torch.bfloat16))
st=weightprep._state[0]
torch.cuda.synchronize(); t=time.time()
for _ in range(50): weightprep._refresh(st, torch.device('cuda',0))
torch.cuda.synchronize(); print('refresh us', (time.time()-t)/50*1e6, 'elements', sum(p.numel() for p in ps)*2)
