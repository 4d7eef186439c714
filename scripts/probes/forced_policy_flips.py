#!/usr/bin/env python
"""VERDICT r4 next #1a: the reference fixture that failed in round 4 when the row-walking kernels were forced onto small planes
(tests/test_gpu_model.py::test_unet_vs_reference_golden[unet_3x21_n1_32-auto] under SMAAT_WGRAD_RECOMPUTE=all SMAAT_FWD_ROWS=all: one
gradient tensor 6.0e-3 against a floor of 5e-3).  Here: the same fixture under the default and the forced policy (three-term split in
both, as in round 4), per-tensor errors against the reference's fp64 anchors, and the ReLU decisions that differ between the two runs."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smaat_unet_amd import ops  # noqa: E402
from tests.test_gpu_model import _load_model  # noqa: E402
from tests.test_host_emu import check_summary, rel  # noqa: E402
from tests.tie_flips import differing_decisions, record_pre_activations  # noqa: E402

DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "unet_3x21_n1_32"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
meta = json.loads(str(g["meta"]))
ops.policy.f16_split = False
res = {}
for policy in ("auto", "all"):
    ops.policy.wgrad_recompute = policy
    ops.invalidate_weight_images()
    model, _ = _load_model(meta)
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    rec = []
    with record_pre_activations(rec):
        logits = model(x)
    tgt = torch.from_numpy(g["target"]).to(DEV)
    loss = (torch.nn.functional.mse_loss(logits.squeeze(1), tgt, reduction="sum") / meta["n"]) if meta["loss"] == "mse" else (logits * tgt).sum()
    loss.backward()
    errs = {}
    for k, p in model.named_parameters():
        key = "grad64/" + k
        if key in g.files or key + "#l2" in g.files:
            try:
                errs[k] = (check_summary(g, key, p.grad.cpu().numpy()), float(g["noise/" + k]) if "noise/" + k in g.files else float("nan"))
            except Exception:  # noqa: BLE001
                pass
    res[policy] = (rec, errs, rel(logits.detach().cpu().numpy(), g["logits"]))
    worst = sorted(errs.items(), key=lambda kv: -kv[1][0])[:4]
    print(f"WGRAD_RECOMPUTE={policy}: logits {res[policy][2]:.2e}; worst gradient tensors vs fp64 (ours, reference fp32 noise):",
          [(k, f"{a:.2e}", f"{b:.2e}") for k, (a, b) in worst])
flips = differing_decisions(res["all"][0], res["auto"][0])
print(f"{len(flips)} ReLU decisions differ between the two policies:")
for i, e, a, b, r in flips:
    print(f"   half {i:2d} element {e}: forced {a:+.3e}  default {b:+.3e}   (rms of the tensor {r:.3f})")
over = [(k, v) for k, v in res["all"][1].items() if v[0] > max(3 * v[1], 5e-3) and not (v[1] != v[1])]
print("tensors over max(3 x noise, 5e-3) under the forced policy:", over)
