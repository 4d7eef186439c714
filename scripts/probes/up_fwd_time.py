#!/usr/bin/env python
"""Time of the bilinear 2x forward on the four decoder levels at batch 32 (output written into the second half of a concatenation
buffer, as in the network)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
N = 32
for c, h in ((64, 144), (128, 72), (256, 36), (512, 18)):
    ho = 2 * h
    x = torch.randn(N, c, h, h, device=dev)
    cat = torch.full((N, 2 * c, ho, ho), float("nan"), device=dev)

    def f():
        return L.smaat_upsample2x_fwd(x.data_ptr(), c * h * h, cat.data_ptr() + 4 * c * ho * ho, 2 * c * ho * ho, N, c, h, h, ho, ho, 0, 0, st)
    assert f() == 0
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = torch.nn.functional.interpolate(x[:2], scale_factor=2, mode="bilinear", align_corners=True)
    err = float((cat[:2, c:] - ref).abs().max())
    print(f"{c:4d} x {h:3d}^2 -> {ho:3d}^2   {ms * 1e3:7.1f} us {4.0 * N * c * (h * h + ho * ho) / 1e9 / ms:5.2f} TB/s   max |out - torch| {err:.2e}", flush=True)
