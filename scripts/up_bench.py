#!/usr/bin/env python
"""Bilinear 2x upsample forward (into the concatenation buffer) / backward and maxpool timing at the four decoder /
encoder levels of BASELINE config 2 (batch 32).  SMAAT_UP_ROWS=0 selects the element-per-thread kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smaat_unet_amd import _lib  # noqa: E402

LEVELS = [(512, 18), (256, 36), (128, 72), (64, 144)]  # (channels of x1, its H = W)


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    N = int(os.environ.get("LB_BATCH", "32"))
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tf = tb = 0.0
    for c, h in LEVELS:
        ho = 2 * h
        x = torch.randn(N, c, h, h, device=dev)
        cat = torch.empty(N, 2 * c, ho, ho, device=dev)
        dcat = torch.randn(N, 2 * c, ho, ho, device=dev)
        dx = torch.empty_like(x)
        off = 4 * c * ho * ho

        def f():
            assert L.smaat_upsample2x_fwd(x.data_ptr(), c * h * h, cat.data_ptr() + off, 2 * c * ho * ho, N, c, h, h, ho, ho,
                                          0, 0, st) == 0

        def b():
            assert L.smaat_upsample2x_bwd(dcat.data_ptr() + off, 2 * c * ho * ho, dx.data_ptr(), c * h * h, N, c, h, h, ho,
                                          ho, 0, 0, st) == 0

        a, bb = timeit(f), timeit(b)
        gb = 4.0 * N * c * (h * h + ho * ho) / 1e6
        print(f"C={c:4d} {h:3d}^2 -> {ho:3d}^2  fwd {a:7.3f} ms {gb / a:7.1f} GB/s | bwd {bb:7.3f} ms {gb / bb:7.1f} GB/s", flush=True)
        tf, tb = tf + a, tb + bb
    print(f"totals ms: fwd {tf:.3f} bwd {tb:.3f}")


if __name__ == "__main__":
    main()
