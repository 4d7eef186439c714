"""Round 6: what one training step costs the HOST (Python + autograd + ctypes launches), measured where the GPU is not the limit
(batch 1, 64 x 64: the same ~350 launches, a few ms of GPU work), against the 27.7 ms the GPU needs at batch 32.
python scripts/probes/host_cost_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import smaat_unet_amd as S  # noqa: E402
from smaat_unet_amd.optim import Adam  # noqa: E402


def run(batch, size, which, steps=30):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = S.SmaAt_UNet(12, 1).to(dev).train()
    opt = Adam(model.parameters(), lr=1e-3) if which == "one" else torch.optim.Adam(model.parameters(), lr=1e-3, foreach=True)
    x = torch.rand(batch, 12, size, size, device=dev)
    y = torch.rand(batch, size, size, device=dev)

    def step():
        out = model(x)
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / batch
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / steps * 1e3
    return t_host, t_all


def main():
    for which in ("one", "foreach"):
        for batch, size in ((1, 64), (32, 288)):
            h, a = run(batch, size, which)
            print(f"Adam={which:8s} batch {batch:2d} {size}x{size}: host loop {h:7.2f} ms/step (no sync), with final sync {a:7.2f} ms/step "
                  f"[{os.cpu_count()} host cores]", flush=True)


if __name__ == "__main__":
    main()
