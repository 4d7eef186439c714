"""Generate tests/golden/* by running the REAL reference (imported read-only from
/root/reference) on seeded inputs.  Runs only in the build container; the GPU box
has no /root/reference and only ever reads the committed fixtures.

    python oracle/gen_golden.py

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from models.SmaAt_UNet import SmaAt_UNet  # noqa: E402  (reference)
from models.layers import CBAM, ChannelAttention, DepthwiseSeparableConv, SpatialAttention  # noqa: E402
from models.unet_parts import OutConv  # noqa: E402
from models.unet_parts_depthwise_separable import DoubleConvDS, DownDS, UpDS  # noqa: E402

from oracle import params as oparams  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
torch.manual_seed(0)


def t2n(t):
    return t.detach().cpu().numpy().copy()


def load_np_state(mod, P):
    sd = mod.state_dict()
    assert list(sd.keys()) == [k for k in P.keys()], "key order mismatch"
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    mod.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})


def randomize_(mod, rng):
    """Replace every parameter with seeded numpy values (BN affine randomised)."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if p.ndim == 1 and ("bn" in name or ".1." in name or ".4." in name) and name.endswith("weight") and \
                    isinstance(dict(mod.named_modules()).get(name.rsplit(".", 1)[0]), torch.nn.BatchNorm2d):
                p.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, p.shape).astype(np.float32)))
            else:
                p.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, p.shape).astype(np.float32)))


def module_case(store, tag, mod, inputs, rng):
    """Run fwd+bwd of a reference module with a random cotangent, store everything."""
    mod.train()
    randomize_(mod, rng)
    for k, v in mod.state_dict().items():
        store[f"{tag}/param/{k}"] = t2n(v)
    xs = [torch.from_numpy(a).requires_grad_(True) for a in inputs]
    out = mod(*xs)
    cot = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
    (out * cot).sum().backward()
    for i, (a, x) in enumerate(zip(inputs, xs)):
        store[f"{tag}/in{i}"] = a
        store[f"{tag}/din{i}"] = t2n(x.grad)
    store[f"{tag}/out"] = t2n(out)
    store[f"{tag}/cot"] = t2n(cot)
    for k, p in mod.named_parameters():
        store[f"{tag}/grad/{k}"] = t2n(p.grad)
    for k, v in mod.state_dict().items():
        if "running" in k or "num_batches" in k:
            store[f"{tag}/after/{k}"] = t2n(v)


def relu_sparse(rng, shape):
    """post-ReLU-like activations: ~45% exact zeros (exercises max/argmax ties)."""
    a = rng.standard_normal(shape).astype(np.float32)
    return np.maximum(a, 0)


def gen_ops():
    rng = np.random.default_rng(42)
    s = {}
    module_case(s, "dsconv_k2", DepthwiseSeparableConv(6, 10, kernel_size=3, padding=1, kernels_per_layer=2),
                [rng.standard_normal((2, 6, 9, 11)).astype(np.float32)], rng)
    module_case(s, "dsconv_k1", DepthwiseSeparableConv(5, 7, kernel_size=3, padding=1, kernels_per_layer=1),
                [rng.standard_normal((2, 5, 8, 8)).astype(np.float32)], rng)
    module_case(s, "dsconv_k4", DepthwiseSeparableConv(3, 8, kernel_size=3, padding=1, kernels_per_layer=4),
                [rng.standard_normal((1, 3, 6, 10)).astype(np.float32)], rng)
    module_case(s, "doubleconv", DoubleConvDS(6, 16, kernels_per_layer=2),
                [rng.standard_normal((3, 6, 12, 10)).astype(np.float32)], rng)
    module_case(s, "doubleconv_mid", DoubleConvDS(8, 4, mid_channels=12, kernels_per_layer=2),
                [rng.standard_normal((2, 8, 8, 8)).astype(np.float32)], rng)
    module_case(s, "down", DownDS(6, 12, kernels_per_layer=2),
                [relu_sparse(rng, (2, 6, 12, 16))], rng)
    module_case(s, "down_odd", DownDS(4, 8, kernels_per_layer=2),
                [relu_sparse(rng, (2, 4, 11, 13))], rng)
    module_case(s, "up", UpDS(16, 6, bilinear=True, kernels_per_layer=2),
                [rng.standard_normal((2, 8, 5, 6)).astype(np.float32),
                 rng.standard_normal((2, 8, 10, 12)).astype(np.float32)], rng)
    module_case(s, "up_pad", UpDS(8, 4, bilinear=True, kernels_per_layer=2),
                [rng.standard_normal((2, 4, 5, 6)).astype(np.float32),
                 rng.standard_normal((2, 4, 11, 13)).astype(np.float32)], rng)
    module_case(s, "chatt", ChannelAttention(32, reduction_ratio=16),
                [relu_sparse(rng, (3, 32, 7, 9))], rng)
    module_case(s, "spatt", SpatialAttention(kernel_size=7),
                [relu_sparse(rng, (3, 10, 9, 12))], rng)
    module_case(s, "cbam", CBAM(32, reduction_ratio=16),
                [relu_sparse(rng, (2, 32, 10, 10))], rng)
    module_case(s, "cbam_small", CBAM(64, reduction_ratio=16),
                [relu_sparse(rng, (2, 64, 4, 4))], rng)
    module_case(s, "outconv", OutConv(16, 3),
                [rng.standard_normal((2, 16, 6, 7)).astype(np.float32)], rng)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **s)
    print("ops.npz:", len(s), "arrays")


def summarize(store, tag, arr):
    a = np.asarray(arr, np.float32).ravel()
    store[tag + "#l2"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    store[tag + "#sum"] = np.float64(a.astype(np.float64).sum())
    store[tag + "#n"] = np.int64(a.size)
    if a.size <= 8192:
        store[tag + "#full"] = np.asarray(arr, np.float32)
    else:
        idx = np.linspace(0, a.size - 1, 4096).astype(np.int64)
        store[tag + "#idx"] = idx
        store[tag + "#vals"] = a[idx]


def gen_unet(name, n_channels, n_classes, n, h, w, loss_kind, seed):
    P = oparams.make_smaat_params(n_channels, n_classes, 2, 16, seed)
    model = SmaAt_UNet(n_channels, n_classes)
    load_np_state(model, P)
    model.train()
    rng = np.random.default_rng(seed + 100)
    if loss_kind == "mse":
        u = rng.random((n, n_channels, h, w), dtype=np.float32)
        x = np.where(u > 0.7, (u - 0.7) / 0.3 * 0.5, 0).astype(np.float32)
        target = (rng.random((n, h, w), dtype=np.float32) * 0.3).astype(np.float32)
    else:
        x = rng.standard_normal((n, n_channels, h, w)).astype(np.float32)
        target = rng.standard_normal((n, n_classes, h, w)).astype(np.float32)  # random cotangent
    acts = {}
    hooks = []
    for nm in ["inc", "cbam1", "down1", "cbam2", "down2", "cbam3", "down3", "cbam4", "down4", "cbam5", "up1", "up2",
               "up3", "up4"]:
        hooks.append(getattr(model, nm).register_forward_hook(lambda m, i, o, nm=nm: acts.__setitem__(nm, t2n(o))))
    xt = torch.from_numpy(x).requires_grad_(True)
    logits = model(xt)
    if loss_kind == "mse":
        # reference: models/regression_lightning.py:57-65
        loss = torch.nn.functional.mse_loss(logits.squeeze(1), torch.from_numpy(target), reduction="sum") / n
    else:
        loss = (logits * torch.from_numpy(target)).sum()
    loss.backward()
    for hk in hooks:
        hk.remove()
    s = {"x": x, "target": target, "logits": t2n(logits), "loss": np.float64(loss.item()),
         "meta": np.array(json.dumps(dict(n_channels=n_channels, n_classes=n_classes, n=n, h=h, w=w, loss=loss_kind,
                                          param_seed=seed)))}
    rename = dict(inc="x1", cbam1="x1Att", down1="x2", cbam2="x2Att", down2="x3", cbam3="x3Att", down3="x4",
                  cbam4="x4Att", down4="x5", cbam5="x5Att", up1="u1", up2="u2", up3="u3", up4="u4")
    for k, v in acts.items():
        summarize(s, "act/" + rename[k], v)
    summarize(s, "dx", t2n(xt.grad))
    for k, p in model.named_parameters():
        summarize(s, "grad/" + k, t2n(p.grad))
    for k, v in model.state_dict().items():
        if "running" in k:
            s["after/" + k] = t2n(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **s)
    print(name, "loss", loss.item(), "arrays", len(s))


def gen_keys():
    out = {}
    for (nc, ncl) in ((12, 1), (3, 21)):
        sd = SmaAt_UNet(nc, ncl).state_dict()
        mine = oparams.smaat_unet_keys(nc, ncl)
        ref = [(k, tuple(v.shape)) for k, v in sd.items()]
        assert ref == [(k, tuple(s)) for k, s in mine], "oracle.params key list != reference state_dict"
        out[f"{nc}_{ncl}"] = [[k, list(s)] for k, s in ref]
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("state_dict keys verified:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    gen_keys()
    gen_ops()
    gen_unet("unet_12x1_n2_32", 12, 1, 2, 32, 32, "mse", 0)
    gen_unet("unet_12x1_n2_64x48", 12, 1, 2, 64, 48, "mse", 1)
    gen_unet("unet_3x21_n1_32", 3, 21, 1, 32, 32, "cot", 2)
