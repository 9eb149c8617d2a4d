import torch,sys
sys.path.insert(0,".")
from sanerf_hq_amd import raymarching as rm
for T in (129,65,33,2):
    for kind in (0,1):
        a=rm.jitter(None,3,T,kind,device="cuda").cpu()[0]
        b=torch.linspace(0,1,T) if kind==0 else torch.linspace(0.5/T,1-0.5/T,steps=T)
        d=(a-b).abs()
        print(T,kind,float(d.max()), int((d>0).sum()), (d>0).nonzero().flatten()[:5].tolist())
