#!/usr/bin/env python
"""How much of a network fixture's per-tensor gradient error is the ROUNDING ORDER of the attention kernels: the same
fixture (tests/golden/unet_12x1_n2_64x48: planes of 64 x 48 ... 4 x 3 pixels, not tie-free) with the channels of the
three-pass attention backward / the spatial pooling split over 1, 2, 4, 8 waves (SMAAT_CBAM_CS, read once per process:
one subprocess per setting) and with the three-pass route off.  Prints error / noise of the worst tensors."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from tests.test_gpu_model import _load_model, DEV
from tests.test_host_emu import check_summary
g = np.load(os.path.join(%r, "tests", "golden", sys.argv[1] + ".npz"))
meta = json.loads(str(g["meta"]))
model, _ = _load_model(meta)
x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
logits = model(x)
tgt = torch.from_numpy(g["target"]).to(DEV)
loss = torch.nn.functional.mse_loss(logits.squeeze(1), tgt, reduction="sum") / meta["n"] if meta["loss"] == "mse" else (logits * tgt).sum()
loss.backward()
rows = []
for k, p in model.named_parameters():
    if ".double_conv." in "." + k and k.endswith(("depthwise.bias", "pointwise.bias")):
        continue
    if p.numel() == 1:
        continue
    e, noise = check_summary(g, "grad64/" + k, p.grad.cpu().numpy()), float(g["noise/" + k])
    rows.append((e / max(3 * noise, 5e-3), e, noise, k))
rows.sort(reverse=True)
print(json.dumps(rows[:4]))
''' % (ROOT, ROOT)


def main():
    fixtures = sys.argv[1:] or ["unet_12x1_n2_64x48"]
    for fx in fixtures:
        for env in ({"SMAAT_CBAM_THREE_PASS": "0"}, {"SMAAT_CBAM_CS": "1"}, {"SMAAT_CBAM_CS": "2"}, {"SMAAT_CBAM_CS": "4"},
                    {"SMAAT_CBAM_CS": "8"}, {}):
            r = subprocess.run([sys.executable, "-c", CHILD, fx], env={**os.environ, **env}, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("[")]
            if not line:
                print(fx, env, "FAILED", r.stderr[-400:])
                continue
            rows = json.loads(line[-1])
            print(f"{fx} {str(env):36s} " + "  ".join(f"{k.replace('spatial_att.', 'sp.').replace('channel_att.', 'ch.')}: {e:.2e} (noise {n:.1e}, {q:.2f} of bound)" for q, e, n, k in rows[:3]), flush=True)


if __name__ == "__main__":
    main()
