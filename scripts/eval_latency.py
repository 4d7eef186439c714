#!/usr/bin/env python
"""eval-mode batch-1 forward latency (run under rocprofv3 --kernel-trace --stats to see the kernel list)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smaat_unet_amd as S  # noqa: E402
from bench import fwd_latency  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda:0")
model = S.SmaAt_UNet(12, 1).to(dev)
print(fwd_latency(model, 288, dev, iters=int(os.environ.get("ITERS", "50"))))
