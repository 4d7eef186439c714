#!/usr/bin/env python
"""Do a power-/MFMA-bound kernel and an HBM-bound kernel overlap when they are launched on two HIP streams?  The weight
gradient of a deep layer (k_wgrad_split: 256 persistent workgroups, one per CU) beside the depthwise backward of a 288^2
layer (register streaming, no LDS).  Sequential time vs two-stream time of the same launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

L = _lib.get()
dev = torch.device("cuda:0")
N = 32


def mk_wgrad(cin, cout, h):
    k, p = 2 * cin, h * h
    y = torch.randn(N, k, h, h, device=dev)
    dz = torch.randn(N, cout, h, h, device=dev)
    ns = L.smaat_wgrad_num_splits(N, h, h, cout, k)
    ws = torch.empty(ns, cout, k, device=dev)
    dw = torch.empty(cout, k, device=dev)

    def f(st):
        assert L.smaat_pointwise_wgrad(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws.data_ptr(), dw.data_ptr(), N, k, cout, h, h, st) == 0
    return f


def mk_dwb(cin, h):
    k, p = 2 * cin, h * h
    x = torch.randn(N, cin, h, h, device=dev)
    dy = torch.randn(N, k, h, h, device=dev)
    dx = torch.empty(N, cin, h, h, device=dev)
    w_dw = torch.randn(k, 9, device=dev)
    ws2 = torch.empty(L.smaat_dw3x3_bwd_ws_rows(N, cin, h, h), k, 10, device=dev)
    dwd, dbd = torch.empty(k, 9, device=dev), torch.empty(k, device=dev)

    def f(st):
        assert L.smaat_dw3x3_bwd(x.data_ptr(), cin * p, dy.data_ptr(), k * p, w_dw.data_ptr(), dx.data_ptr(), cin * p, ws2.data_ptr(),
                                 dwd.data_ptr(), dbd.data_ptr(), N, cin, 2, h, h, st) == 0
    return f


def mk_bn(c, h):
    p = h * h
    dy, z = torch.randn(N, c, h, h, device=dev), torch.randn(N, c, h, h, device=dev)
    dz = torch.empty_like(z)
    v = [torch.rand(c, device=dev) + 0.5 for _ in range(4)]
    coef = torch.rand(3, c, device=dev) * 0.01

    def f(st):
        assert L.smaat_bn_bwd_apply(dy.data_ptr(), c * p, z.data_ptr(), c * p, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                    v[3].data_ptr(), coef.data_ptr(), dz.data_ptr(), c * p, N, c, p, 1, st) == 0
    return f


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for gname, g, reps_g in (("wgrad up1.0 (K 2048, M 512, 36^2)", mk_wgrad(1024, 512, 36), 4),
                         ("wgrad up2.0 (K 1024, M 256, 72^2)", mk_wgrad(512, 256, 72), 4),
                         ("wgrad up3.0 (K 512, M 128, 144^2)", mk_wgrad(256, 128, 144), 4)):
    for mname, m, reps_m in (("dw bwd 64ch 288^2", mk_dwb(64, 288), 4), ("bn_bwd_apply 64ch 288^2", mk_bn(64, 288), 8)):
        cur = torch.cuda.current_stream()

        def seq():
            for _ in range(reps_g):
                g(cur.cuda_stream)
            for _ in range(reps_m):
                m(cur.cuda_stream)

        def par():
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            for i in range(max(reps_g, reps_m)):
                if i < reps_g:
                    g(sa.cuda_stream)
                if i < reps_m:
                    m(sb.cuda_stream)
            cur.wait_stream(sa)
            cur.wait_stream(sb)

        def only_g():
            for _ in range(reps_g):
                g(cur.cuda_stream)

        def only_m():
            for _ in range(reps_m):
                m(cur.cuda_stream)

        tg, tm, ts, tp = timed(only_g), timed(only_m), timed(seq), timed(par)
        print(f"{gname:36s} + {mname:24s}: gemm {tg:6.3f}  mem {tm:6.3f}  sequential {ts:6.3f}  two streams {tp:6.3f} ms  "
              f"({100 * (ts - tp) / ts:+.0f} %)")
