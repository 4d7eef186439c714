import sys; sys.path.insert(0,'/root/reference'); sys.path.insert(0,'/root/repo')
import torch, numpy as np
from models.SmaAt_UNet import SmaAt_UNet
from oracle import smaat_oracle as O
torch.manual_seed(0)
net = SmaAt_UNet(12,1).train()
sd = {k:v.clone() for k,v in net.state_dict().items()}
xn, yn = O.synthetic_precip(2,12,64,64,seed=3)
x = torch.from_numpy(xn); y = torch.rand(2,64,64)*0.3
gs=[]
for amp in (False, True):
    net.load_state_dict(sd); net.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
        out = net(x)
    loss = torch.nn.functional.mse_loss(out.float().squeeze(1), y, reduction="sum")/2
    loss.backward()
    gs.append({k:p.grad.clone() for k,p in net.named_parameters()})
f0 = torch.cat([g.flatten() for g in gs[0].values()]); f1 = torch.cat([gs[1][k].flatten() for k in gs[0]])
print("stock autocast flat-gradient cosine", float((f0*f1).sum()/(f0.norm()*f1.norm())), "rel", float((f1-f0).norm()/f0.norm()))
for k in list(gs[0])[:6]+list(gs[0])[-4:]:
    a,b=gs[0][k],gs[1][k]
    print(k, float((b-a).norm()/a.norm().clamp(min=1e-30)))
