#!/usr/bin/env python
"""Soak: two identical training runs (same seed, same batches) of the default f32 path must agree BIT FOR BIT after many steps --
every kernel has fixed reduction orders, and the operand maxima of the fp16 split are order-independent (atomicMax on bit patterns).
A race in any kernel (round 5 met one in a build that is no longer instantiated) shows up as a diverging loss."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import smaat_unet_amd as S  # noqa: E402

dev = torch.device("cuda:0")
steps = int(os.environ.get("SOAK_STEPS", "150"))
B, H = int(os.environ.get("SOAK_BATCH", "16")), int(os.environ.get("SOAK_SIZE", "288"))


def run():
    torch.manual_seed(0)
    m = S.SmaAt_UNet(12, 1).to(dev).train()
    if os.environ.get("SMAAT_ADAM", "one") == "one":  # (bench.py's default since round 6)
        from smaat_unet_amd.optim import Adam as OneLaunchAdam
        opt = OneLaunchAdam(m.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, foreach=True)
    g = torch.Generator().manual_seed(7)
    losses = []
    for i in range(steps):
        u = torch.rand(B, 12, H, H, generator=g)
        x = torch.where(u > 0.7, (u - 0.7) / 0.3 * 0.5, torch.zeros(())).to(dev)
        y = (torch.rand(B, H, H, generator=g) * 0.3).to(dev)
        out = m(x)
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / B
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), [p.detach().clone() for p in m.parameters()]


t0 = time.time()
la, pa = run()
lb, pb = run()
same_loss = bool(torch.equal(la, lb))
same_par = all(torch.equal(a, b) for a, b in zip(pa, pb))
first = next((i for i in range(steps) if la[i] != lb[i]), None)
print(f"{steps} steps x 2 runs at batch {B}, {H}x{H}: losses bit-identical {same_loss}, parameters bit-identical {same_par}, "
      f"first differing step {first}, loss {la[0].item():.4f} -> {la[-1].item():.4f}, finite {bool(torch.isfinite(la).all())}, {time.time() - t0:.0f} s")
sys.exit(0 if (same_loss and same_par) else 1)
