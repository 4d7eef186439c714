import os, sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests import emu_backend
from tests import test_eval_and_big as B
from smaat_unet_amd import ops
emu_backend.install()
gd = '/root/repo/tests/golden'
name = sys.argv[1]
for mode in ("off", "h3", "h4"):
    ops.policy.f16_split = mode != "off"
    os.environ["SMAAT_EMU_H4"] = "1" if mode == "h4" else ""
    ops.invalidate_weight_images()
    rep = {}
    try:
        B.run_big(gd, name, torch.device("cpu"), report=rep)
        st = "PASS"
    except AssertionError as e:
        st = "FAIL"
    t = rep["train"]["per_tensor"]
    ks = [k for k in t if 'pointwise.weight' in k]
    print(mode, st, "logits %.2e" % rep["train"]["logits"], " ".join(f"{k.split('.')[0]}.{k.split('.')[-3]}:{t[k][0]:.1e}" for k in ks))
