#!/usr/bin/env python
"""Depthwise 3x3 forward / backward timing per layer shape of BASELINE config 2 (batch 32, 288x288).
SMAAT_DW_ROWS=0 selects the LDS strip kernels, default = the register row-streaming kernels (dwrows.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smaat_unet_amd import _lib  # noqa: E402

SHAPES = [(12, 288), (64, 288), (128, 288), (64, 144), (128, 144), (256, 144), (128, 72), (256, 72), (512, 72),
          (256, 36), (512, 36), (1024, 36), (512, 18)]


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
    tf = tb = tr = 0.0
    for cin, h in SHAPES:
        w, k, p = h, cin * 2, h * h
        x = torch.randn(N, cin, h, w, device=dev)
        w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
        y = torch.empty(N, k, h, w, device=dev)
        dy = torch.randn(N, k, h, w, device=dev)
        dx = torch.empty(N, cin, h, w, device=dev)
        rows = L.smaat_dw3x3_bwd_ws_rows(N, cin, h, w)
        ws = torch.empty(rows, k, 10, device=dev)
        dwd, dbd = torch.empty(k, 9, device=dev), torch.empty(k, device=dev)
        sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
        mean, invstd = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
        rpart = torch.empty(2, rows - 1, cin, device=dev)

        def f_fwd():
            assert L.smaat_dw3x3_fwd(x.data_ptr(), cin * p, sc.data_ptr(), sh.data_ptr(), w_dw.data_ptr(), b_dw.data_ptr(),
                                     y.data_ptr(), k * p, N, cin, 2, h, w, st) == 0

        def f_bwd():
            assert L.smaat_dw3x3_bwd(x.data_ptr(), cin * p, dy.data_ptr(), k * p, w_dw.data_ptr(), dx.data_ptr(), cin * p,
                                     ws.data_ptr(), dwd.data_ptr(), dbd.data_ptr(), N, cin, 2, h, w, st) == 0

        def f_bnred():
            assert L.smaat_dw3x3_bwd_bnred(x.data_ptr(), cin * p, sc.data_ptr(), sh.data_ptr(), dy.data_ptr(), k * p,
                                           w_dw.data_ptr(), dx.data_ptr(), cin * p, ws.data_ptr(), dwd.data_ptr(),
                                           dbd.data_ptr(), mean.data_ptr(), invstd.data_ptr(), rpart.data_ptr(), N, cin, 2,
                                           h, w, st) == 0

        a, b, c = timeit(f_fwd), timeit(f_bwd), timeit(f_bnred)
        gf = 4.0 * N * 3 * cin * p / 1e6  # fwd: read Cin, write 2 Cin planes
        gb = 4.0 * N * 4 * cin * p / 1e6  # bwd: read 2 Cin (dY) + Cin (x), write Cin
        print(f"Cin={cin:5d} {h:3d}^2  fwd {a:7.3f} ms {gf / a:7.1f} GB/s | bwd {b:7.3f} ms {gb / b:7.1f} GB/s | "
              f"bwd+bnred {c:7.3f} ms {gb / c:7.1f} GB/s", flush=True)
        tf, tb, tr = tf + a, tb + b, tr + c
    print(f"totals ms: fwd {tf:.3f} bwd {tb:.3f} bnred {tr:.3f}")


if __name__ == "__main__":
    main()
