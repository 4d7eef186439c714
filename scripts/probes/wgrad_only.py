#!/usr/bin/env python
"""One layer's pointwise weight gradient, a few launches (for rocprofv3 counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

N, cin, cout, h = 32, int(os.environ.get("DG_CIN", 512)), int(os.environ.get("DG_COUT", 256)), int(os.environ.get("DG_H", 72))
L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
k, p = cin * 2, h * h
y = torch.randn(N, k, h, h, device=dev)
dz = torch.randn(N, cout, h, h, device=dev)
ns = L.smaat_wgrad_num_splits(N, h, h, cout, k)
ws = torch.empty(ns, cout, k, device=dev)
dw = torch.empty(cout, k, device=dev)
for _ in range(int(os.environ.get("DG_ITERS", 3))):
    assert L.smaat_pointwise_wgrad(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws.data_ptr(), dw.data_ptr(), N, k, cout,
                                   h, h, st) == 0
torch.cuda.synchronize()
print("ok")
