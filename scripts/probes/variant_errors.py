#!/usr/bin/env python
"""Per-parameter gradient error of a sibling-network golden case against the fp64 reference anchors."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import smaat_unet_amd as S  # noqa: E402
from oracle import params as oparams  # noqa: E402
from tests.test_host_emu import check_summary  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "unetds4cbam_k4_n1_32"
dev = torch.device("cuda:0")
g = np.load(os.path.join("tests", "golden", name + ".npz"))
meta = json.loads(str(g["meta"]))
cls = {0: S.UNetDS, 4: S.UNetDSAttention4CBAMs}[meta["cbams"]]
model = cls(n_channels=meta["n_channels"], n_classes=meta["n_classes"], kernels_per_layer=meta["kpl"])
P = oparams.fill(oparams.unetds_keys(meta["n_channels"], meta["n_classes"], meta["kpl"], 16, meta["cbams"]), meta["param_seed"])
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
model.to(dev).train()
x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
y = model(x)
print("logits rel", np.linalg.norm(y.detach().cpu().numpy() - g["logits"]) / np.linalg.norm(g["logits"]))
(y * torch.from_numpy(g["cot"]).to(dev)).sum().backward()
rows = []
for k, p in model.named_parameters():
    if ".double_conv." in k and k.endswith(("depthwise.bias", "pointwise.bias")):
        continue
    rows.append((check_summary(g, "grad64/" + k, p.grad.cpu().numpy()), float(g["noise/" + k]), k))
rows.sort(reverse=True)
for r in rows[:14]:
    print("gpu_vs_fp64 %.2e  ref32_vs_fp64 %.2e  %s" % r)
