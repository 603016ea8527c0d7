import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvtabular_amd import kernels as K
from nvtabular_amd.device import pack_bitmap_device
n=45_000_000; dev=torch.device("cuda",0); card=39043.; s=1.1
g=torch.Generator(device=dev).manual_seed(1)
u=torch.rand(n,device=dev,dtype=torch.float64,generator=g)
x=((card**(1-s)-1)*u+1)**(1/(1-s))
keys=((x.floor().clamp_(1,card).to(torch.int64)*2654435761)%(2**31)).to(torch.int32)
del u,x
for name,valid in (("none",None),("all-valid",pack_bitmap_device(torch.ones(n,dtype=torch.bool,device=dev))),("30% null",pack_bitmap_device(torch.rand(n,device=dev)>=0.3))):
    for it in range(2): K.dense_count(keys,valid,None,hint=39042)
    K.profile_begin(); K.dense_count(keys,valid,None,hint=39042); p=K.profile_end()
    print(name, {k:round(v[0]*1e3) for k,v in p.items()})
