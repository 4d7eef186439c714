#!/usr/bin/env python
"""One layer's recompute weight gradient (csrc/dswgrad.hip) and, beside it, the streamed split kernel on the kept depthwise
output: a few launches each, for rocprofv3 counter passes (scripts/gpu_pmc_dswg.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

N, cin, cout, h = 32, int(os.environ.get("DG_CIN", 64)), int(os.environ.get("DG_COUT", 64)), int(os.environ.get("DG_H", 288))
L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
k, p = cin * 2, h * h
x = torch.randn(N, cin, h, h, device=dev)
y = torch.randn(N, k, h, h, device=dev)
w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
dz = torch.randn(N, cout, h, h, device=dev)
ws = torch.empty(L.smaat_dsconv_wgrad_split_num_splits(N, cin, cout, h, h), cout, k, device=dev)
ws2 = torch.empty(L.smaat_wgrad_num_splits(N, h, h, cout, k), cout, k, device=dev)
dw = torch.empty(cout, k, device=dev)
if os.environ.get("DG_BF16", "0") == "1":  # mixed precision: typed recompute kernel next to the streamed bf16 weight gradient
    N = int(os.environ.get("DG_BATCH", 64))
    BF = torch.bfloat16
    x = torch.randn(N, cin, h, h, device=dev).to(BF)
    y = torch.randn(N, k, h, h, device=dev).to(BF)
    dz = torch.randn(N, cout, h, h, device=dev).to(BF)
    ws = torch.empty(L.smaat_dsconv_wgrad_split_num_splits(N, cin, cout, h, h), cout, k, device=dev)
    ws2 = torch.empty(L.smaat_wgrad_num_splits(N, h, h, cout, k), cout, k, device=dev)
    for _ in range(int(os.environ.get("DG_ITERS", 3))):
        assert L.smaat_dsconv_wgrad_split_t(x.data_ptr(), 1, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(), 1,
                                            cout * p, ws.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, h, st) == 0
        assert L.smaat_pointwise_wgrad_bf16(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws2.data_ptr(), dw.data_ptr(), N, k, cout,
                                            h, h, st) == 0
    torch.cuda.synchronize()
    print("ok bf16")
    sys.exit(0)
for _ in range(int(os.environ.get("DG_ITERS", 3))):
    assert L.smaat_dsconv_wgrad_split(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(), cout * p,
                                      ws.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, h, st) == 0
    assert L.smaat_pointwise_wgrad(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws2.data_ptr(), dw.data_ptr(), N, k, cout, h, h,
                                   st) == 0
torch.cuda.synchronize()
print("ok")
