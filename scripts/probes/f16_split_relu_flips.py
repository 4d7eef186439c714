import os, sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests import emu_backend
from tests import test_eval_and_big as B
from smaat_unet_amd import ops
import smaat_unet_amd as S
from oracle import params as oparams
emu_backend.install()
gd = '/root/repo/tests/golden'
name = "unet_12x1_n3_64x48_eval"
g = np.load(os.path.join(gd, name + ".npz")); meta = json.loads(str(g["meta"]))
(x, target), _ = B.big_inputs(meta)
P = oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], 2, 16, meta["param_seed"])
rec = {}
orig = ops._affine_act_raw
orig_bn = ops._bn_finalize_raw
def run(f16):
    ops.policy.f16_split = f16; ops.invalidate_weight_images()
    model = S.SmaAt_UNet(meta["n_channels"], meta["n_classes"])
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}); model.train()
    pre = []
    # capture (z, scale, shift) of every BN: hook _half_forward
    hf = ops._half_forward
    def wrap(*a, **k):
        r = hf(*a, **k)
        y, z, st = r[0], r[1], r[2]
        pre.append((z.detach().clone(), st.detach().clone()))
        return r
    ops._half_forward = wrap
    try:
        model(torch.from_numpy(x))
    finally:
        ops._half_forward = hf
    return pre
a = run(False); b = run(True)
names = ["inc.0","inc.1","d1.0","d1.1","d2.0","d2.1","d3.0","d3.1","d4.0","d4.1","u1.0","u1.1","u2.0","u2.1","u3.0","u3.1","u4.0","u4.1"]
for nm, (za, sa), (zb, sb) in zip(names, a, b):
    pa = za * sa[2][None,:,None,None] + sa[3][None,:,None,None]
    pb = zb * sb[2][None,:,None,None] + sb[3][None,:,None,None]
    flips = ((pa > 0) != (pb > 0))
    rms = float(pa.pow(2).mean().sqrt())
    nf = int(flips.sum())
    info = ""
    if nf:
        idx = flips.nonzero()[:3]
        info = " ".join(f"[{tuple(i.tolist())}: {float(pa[tuple(i)]):+.2e} vs {float(pb[tuple(i)]):+.2e}]" for i in idx)
    print(f"{nm:6s} n={pa.numel():8d} flips={nf} rms={rms:.2e} maxdiff={float((pa-pb).abs().max()):.2e} {info}")
