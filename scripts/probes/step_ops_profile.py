#!/usr/bin/env python
"""Which host-level operators launch the small fill / copy kernels of a training step (torch.profiler, one step)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import smaat_unet_amd as S  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = S.SmaAt_UNet(12, 1).to(dev).train()
x = torch.rand(32, 12, 288, 288, device=dev)
y = torch.rand(32, 288, 288, device=dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3, foreach=True)


def step():
    out = m(x)
    loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.size(0)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=False)
rows = sorted(ev, key=lambda e: -e.count)
print(f"{'name':60s} {'count':>6s} {'cuda_us':>10s}")
for e in rows[:60]:
    print(f"{e.key[:60]:60s} {e.count:6d} {getattr(e, 'device_time_total', 0.0):10.1f}")
