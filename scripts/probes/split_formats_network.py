#!/usr/bin/env python
"""CPU study for the next round: the reference-generated NETWORK fixtures (logits, loss, per-parameter gradients against the
float64 anchors, running statistics) run through the emulated C ABI with the f32-storage GEMMs switched to a two-term fp16
operand split (tests/emu_backend.py: SMAAT_EMU_GEMM=f16x2), next to the default emulation (plain float32 products).
Prints, per fixture and mode, the logits error and the worst per-tensor gradient error relative to the fixture's own
float32-vs-float64 noise -- the quantities tests/test_host_emu.py::test_unet bounds (logits < 1e-4; gradients <= max(3 x noise, 5e-3))."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import emu_backend  # noqa: E402

emu_backend.install()
import smaat_unet_amd as S  # noqa: E402
from oracle import params as oparams  # noqa: E402
from smaat_unet_amd import ops as _ops  # noqa: E402
from tests.test_host_emu import check_param_grads, check_summary, rel  # noqa: E402

_ops.policy.split_policy = "all"  # every supported layer through the split-GEMM wiring
GOLD = os.path.join(ROOT, "tests", "golden")
for name in ("unet_12x1_n2_32", "unet_3x21_n1_32", "unet_12x1_n2_64x48"):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        continue
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    kpl = meta.get("kpl", 2)
    for mode in ("f32", "f16x2"):
        os.environ["SMAAT_EMU_GEMM"] = "" if mode == "f32" else mode
        _ops._PLANES.clear()
        _ops._PLANES_TABLE.clear()
        P = oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], kpl, 16, meta["param_seed"])
        model = S.SmaAt_UNet(meta["n_channels"], meta["n_classes"], kernels_per_layer=kpl)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
        model.train()
        x = torch.from_numpy(g["x"]).requires_grad_(True)
        logits = model(x)
        e_log = rel(logits.detach().numpy(), g["logits"])
        if meta["loss"] == "mse":
            loss = torch.nn.functional.mse_loss(logits.squeeze(1), torch.from_numpy(g["target"]), reduction="sum") / meta["n"]
        else:
            loss = (logits * torch.from_numpy(g["target"])).sum()
        loss.backward()
        bad = check_param_grads(g, [(k, p.grad.numpy()) for k, p in model.named_parameters()])
        e_dx = check_summary(g, "dx64", x.grad.numpy())
        print(f"{name:22s} {mode:6s} logits rel {e_log:.2e}   dx vs f64 {e_dx:.2e} (fixture f32 noise {float(g['noise/dx']):.2e})   "
              f"parameter-gradient tensors over their bound: {len(bad)}" + (f"  worst {bad[0]}" if bad else ""), flush=True)
