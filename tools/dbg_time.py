import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvtabular_amd import kernels as K, _lib
n=45_000_000; dev=torch.device("cuda",0)
for card,s in ((39884406.,1.05),(39043.,1.1)):
    g=torch.Generator(device=dev).manual_seed(int(card))
    u=torch.rand(n,device=dev,dtype=torch.float64,generator=g)
    x=((card**(1-s)-1)*u+1)**(1/(1-s))
    keys=((x.floor().clamp_(1,card).to(torch.int64)*2654435761)%(2**31)).to(torch.int32)
    del u,x
    for it in range(3):
        j=K.DenseCountJob(keys,None,None,hint=6_000_000); j.path=1
        st=torch.zeros(1,_lib.STATE_WORDS,dtype=torch.int64,device=dev); j.state=st[0]
        torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); j.launch(); b.record(); torch.cuda.synchronize()
    print(int(card), round(a.elapsed_time(b)*1e3), st.cpu().tolist()[0][2:4])
