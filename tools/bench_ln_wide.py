import sys; sys.path.insert(0,'.')
import torch
from applecider_amd import hipops as H
dev=torch.device('cuda')
x=torch.randn(8192,3072,device=dev,requires_grad=True); w=torch.ones(3072,device=dev,requires_grad=True); b=torch.zeros(3072,device=dev,requires_grad=True)
go=torch.randn(8192,3072,device=dev)
def f():
    y=H.layer_norm(x,w,b,1e-5,act="gelu"); y.backward(go)
for _ in range(3): f()
torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize(); print("LN 8192x3072 fwd+bwd ms", s.elapsed_time(e)/10)
