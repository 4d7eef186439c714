#!/usr/bin/env python
"""One layer's split data-gradient GEMM, a few launches (for rocprofv3 counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

N, cin, cout, h = 32, int(os.environ.get("DG_CIN", 128)), int(os.environ.get("DG_COUT", 64)), int(os.environ.get("DG_H", 288))
L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
k, p = cin * 2, h * h
w_pw = torch.randn(cout, k, device=dev) * 0.1
wtt = w_pw.t().contiguous()
pl_b = torch.empty(3, k, (cout + 15) // 16 * 16, dtype=torch.int16, device=dev)
dz = torch.randn(N, cout, h, h, device=dev)
dy = torch.empty(N, k, h, h, device=dev)
assert L.smaat_split_planes(wtt.data_ptr(), k, cout, pl_b.data_ptr(), st) == 0
for _ in range(int(os.environ.get("DG_ITERS", 3))):
    assert L.smaat_pointwise_fwd_split(dz.data_ptr(), cout * p, pl_b.data_ptr(), None, dy.data_ptr(), k * p, None,
                                       N, cout, k, h, h, st) == 0
torch.cuda.synchronize()
print("ok")
