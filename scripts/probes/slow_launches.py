#!/usr/bin/env python
"""Which launches of one training step are NOT running at streaming rate: HIP events around every entry point of the library
(smaat_unet_amd/_lib.py Profiler) for one step of the BASELINE config-2 workload (batch 32, 288 x 288, f32), every call priced with
the algorithmic bytes of its work model, and the calls listed that take more than 25 us at less than 3.5 TB/s
(streaming kernels reach 4.3-5.5 TB/s) -- candidates for a dependent chain or a grid too small for the chip, the way the attention
kernels of the deep levels were.  Calls without a work model (attention, small finalize kernels) are listed by time alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import smaat_unet_amd as S  # noqa: E402
from smaat_unet_amd import _lib  # noqa: E402


def main():
    batch = int(os.environ.get("SL_BATCH", "32"))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = S.SmaAt_UNet(12, 1).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = torch.randn(batch, 12, 288, 288, device=dev)
    y = torch.randn(batch, 288, 288, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(model(x).squeeze(1), y, reduction="sum") / batch
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof = _lib.Profiler()
    step()
    torch.cuda.synchronize()
    rows = []
    for name, e0, e1, args in prof.records:
        ms = e0.elapsed_time(e1)
        wm = _lib.WORK_MODELS.get(name)
        fl, by = wm(args) if wm is not None else (0.0, 0.0)
        ints = [a for a in args if isinstance(a, int) and 0 < a < 100000]
        rows.append((ms, name, by, fl, ints[-8:]))
    prof.close()
    tot = sum(r[0] for r in rows)
    print(f"{len(rows)} calls, {tot:.2f} ms inside entry points")
    print("--- priced calls > 25 us below 3.5 TB/s (and below 150 TFLOP/s), slowest first")
    for ms, name, by, fl, ints in sorted(rows, reverse=True):
        if by and ms > 0.025 and by / ms / 1e6 < 3500 and fl / ms / 1e9 < 150:
            print(f"  {ms * 1e3:8.1f} us  {by / ms / 1e6:7.0f} GB/s {fl / ms / 1e9:6.1f} TF  {name:34s} {ints}")
    print("--- calls without a work model > 25 us")
    for ms, name, by, fl, ints in sorted(rows, reverse=True):
        if not by and ms > 0.025:
            print(f"  {ms * 1e3:8.1f} us  {name:34s} {ints}")


if __name__ == "__main__":
    main()
