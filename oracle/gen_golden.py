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


GENERIC_DSCONV = {  # tag: (ctor arguments of the reference's DepthwiseSeparableConv, input shape)
    "dsconv_g5": (dict(in_channels=4, output_channels=6, kernel_size=5, padding=2, kernels_per_layer=3), (2, 4, 9, 10)),
    "dsconv_g3p0": (dict(in_channels=5, output_channels=7, kernel_size=3), (2, 5, 8, 9)),  # the module's defaults
    "dsconv_g1": (dict(in_channels=6, output_channels=4, kernel_size=1, padding=0, kernels_per_layer=2), (2, 6, 5, 7)),
    "dsconv_g7p1": (dict(in_channels=3, output_channels=5, kernel_size=7, padding=1, kernels_per_layer=5), (1, 3, 12, 11)),
    "dsconv_g3k3": (dict(in_channels=4, output_channels=8, kernel_size=3, padding=1, kernels_per_layer=3), (2, 4, 8, 8)),
    "dsconv_g3p2": (dict(in_channels=2, output_channels=3, kernel_size=3, padding=2, kernels_per_layer=2), (1, 2, 6, 4)),
}


def gen_ops_generic():
    """DepthwiseSeparableConv outside the 3x3 / padding 1 / kpl in {1, 2, 4} configuration the network uses
    (models/layers.py:35-45 accepts any kernel_size / padding / kernels_per_layer)"""
    rng = np.random.default_rng(4242)
    s = {}
    for tag, (kw, shape) in GENERIC_DSCONV.items():
        module_case(s, tag, DepthwiseSeparableConv(**kw), [rng.standard_normal(shape).astype(np.float32)], rng)
    # a whole block at a kernels_per_layer the fused kernels are not built for (the reference accepts any integer)
    module_case(s, "doubleconv_k3", DoubleConvDS(5, 8, kernels_per_layer=3), [rng.standard_normal((2, 5, 8, 10)).astype(np.float32)], rng)
    np.savez_compressed(os.path.join(OUT, "ops_generic.npz"), **s)
    print("ops_generic.npz:", len(s), "arrays")


def module_case_eval(store, tag, mod, inputs, rng):
    """eval-mode counterpart of module_case (reference call stack D: model.eval() inference, and fine-tuning with
    frozen statistics): random parameters AND random running statistics, forward + backward with a random
    cotangent in eval mode (BatchNorm = a fixed per-channel affine map; conv biases in front of it now DO have a
    gradient)."""
    randomize_(mod, rng)
    with torch.no_grad():
        for name, buf in mod.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, buf.shape).astype(np.float32)))
            elif name.endswith("running_var"):
                buf.copy_(torch.from_numpy(rng.uniform(0.5, 2.0, buf.shape).astype(np.float32)))
    mod.eval()
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
    for k, v in mod.state_dict().items():  # eval mode must not touch the running statistics
        if "running" in k or "num_batches" in k:
            store[f"{tag}/after/{k}"] = t2n(v)


def gen_ops_eval():
    rng = np.random.default_rng(4242)
    s = {}
    module_case_eval(s, "doubleconv", DoubleConvDS(6, 16, kernels_per_layer=2),
                     [rng.standard_normal((3, 6, 12, 10)).astype(np.float32)], rng)
    # K = 128 / 256 and Cout = 128: the branch of the inference policy that keeps the split GEMMs
    module_case_eval(s, "doubleconv_wide", DoubleConvDS(64, 128, kernels_per_layer=2),
                     [rng.standard_normal((1, 64, 8, 8)).astype(np.float32)], rng)
    module_case_eval(s, "doubleconv_18", DoubleConvDS(8, 12, kernels_per_layer=2),       # W % 4 != 0
                     [rng.standard_normal((2, 8, 18, 18)).astype(np.float32)], rng)
    module_case_eval(s, "down", DownDS(6, 12, kernels_per_layer=2), [relu_sparse(rng, (2, 6, 12, 16))], rng)
    module_case_eval(s, "up_pad", UpDS(8, 4, bilinear=True, kernels_per_layer=2),
                     [rng.standard_normal((2, 4, 5, 6)).astype(np.float32),
                      rng.standard_normal((2, 4, 11, 13)).astype(np.float32)], rng)
    module_case_eval(s, "spatt", SpatialAttention(kernel_size=7), [relu_sparse(rng, (3, 10, 9, 12))], rng)
    module_case_eval(s, "cbam", CBAM(32, reduction_ratio=16), [relu_sparse(rng, (2, 32, 10, 10))], rng)
    module_case_eval(s, "dsconv_k2", DepthwiseSeparableConv(6, 10, kernel_size=3, padding=1, kernels_per_layer=2),
                     [rng.standard_normal((2, 6, 9, 11)).astype(np.float32)], rng)
    np.savez_compressed(os.path.join(OUT, "ops_eval.npz"), **s)
    print("ops_eval.npz:", len(s), "arrays")


def _tie_margin(mod, xs64):
    """smallest relative distance of any selection in `mod` (fp64 run) from a tie: |pre-ReLU value| / rms, gap between
    the two largest entries of every MaxPool2d window / AdaptiveMaxPool2d plane / channel-max of SpatialAttention,
    each divided by the rms of the tensor.  A fixture with margin >> fp32 round-off is the SAME function for every
    correct fp32 implementation."""
    margins = []

    def rms(t):
        return float(t.pow(2).mean().sqrt()) + 1e-30

    def relu_pre(m, inp):
        margins.append(float(inp[0].detach().abs().min()) / rms(inp[0].detach()))

    def maxpool_pre(m, inp):
        win = torch.nn.functional.unfold(inp[0].reshape(-1, 1, *inp[0].shape[2:]), 2, stride=2)  # [NC][4][L]
        top = win.topk(2, dim=1).values
        margins.append(float((top[:, 0] - top[:, 1]).min()) / rms(inp[0]))

    def amax_pre(m, inp):
        top = inp[0].flatten(2).topk(2, dim=2).values
        margins.append(float((top[..., 0] - top[..., 1]).min()) / rms(inp[0]))

    def spatt_pre(m, inp):
        top = inp[0].topk(2, dim=1).values
        margins.append(float((top[:, 0] - top[:, 1]).min()) / rms(inp[0]))
    hooks = []
    for m in mod.modules():
        if isinstance(m, torch.nn.ReLU):
            hooks.append(m.register_forward_pre_hook(relu_pre))
        elif isinstance(m, torch.nn.MaxPool2d):
            hooks.append(m.register_forward_pre_hook(maxpool_pre))
        elif isinstance(m, torch.nn.AdaptiveMaxPool2d):
            hooks.append(m.register_forward_pre_hook(amax_pre))
        elif isinstance(m, SpatialAttention):
            hooks.append(m.register_forward_pre_hook(spatt_pre))
    mod(*xs64)
    for hk in hooks:
        hk.remove()
    return min(margins) if margins else 1.0


def module_case_strict(store, tag, make_mod, make_inputs, rng, margin=2e-4, max_tries=400):
    """Tie-free block fixture with fp64 anchors (VERDICT r1 "next" 1c): parameters / inputs are redrawn until every
    ReLU / max selection of the block is at least `margin` (relative to the rms of its tensor) away from a tie in an
    fp64 run -- three orders of magnitude above fp32 round-off, so no correct fp32 implementation can take another
    branch.  Stored: fp32 reference results, fp64 anchors and the reference's own fp32-vs-fp64 error per tensor.
    tests/test_strict_blocks.py holds the HIP path to 2 x that error (floor: see the test)."""
    for attempt in range(max_tries):
        mod = make_mod()
        mod.train()
        randomize_(mod, rng)
        inputs = make_inputs(rng)
        m64 = make_mod()
        m64.load_state_dict(mod.state_dict())
        m64 = m64.double().train()
        mg = _tie_margin(m64, [torch.from_numpy(a).double() for a in inputs])
        if mg >= margin:
            break
    else:
        raise RuntimeError(f"{tag}: no tie-free draw")
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    for k, v in state0.items():
        store[f"{tag}/param/{k}"] = t2n(v)
    relv = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))  # noqa: E731
    xs = [torch.from_numpy(a).requires_grad_(True) for a in inputs]
    out = mod(*xs)
    cot = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
    (out * cot).sum().backward()
    m64 = make_mod()
    m64.load_state_dict(state0)
    m64 = m64.double().train()
    xs64 = [torch.from_numpy(a).double().requires_grad_(True) for a in inputs]
    out64 = m64(*xs64)
    (out64 * cot.double()).sum().backward()
    store[f"{tag}/out"], store[f"{tag}/out64"], store[f"{tag}/cot"] = t2n(out), t2n(out64), t2n(cot)
    store[f"{tag}/noise/out"] = np.float64(relv(t2n(out), t2n(out64)))
    store[f"{tag}/margin"] = np.float64(mg)
    for i, (a, x, x64) in enumerate(zip(inputs, xs, xs64)):
        store[f"{tag}/in{i}"] = a
        store[f"{tag}/din64_{i}"] = t2n(x64.grad)
        store[f"{tag}/noise/din{i}"] = np.float64(relv(t2n(x.grad), t2n(x64.grad)))
    for (k, p), (_, p64) in zip(mod.named_parameters(), m64.named_parameters()):
        store[f"{tag}/grad64/{k}"] = t2n(p64.grad)
        store[f"{tag}/noise/grad/{k}"] = np.float64(relv(t2n(p.grad), t2n(p64.grad)))
    print(f"  {tag}: margin {mg:.1e} after {attempt + 1} draws, worst fp32 noise",
          max(float(v) for k, v in store.items() if k.startswith(f"{tag}/noise/")
              and not k.endswith(("depthwise.bias", "pointwise.bias"))))  # (exact-zero gradients in front of a BatchNorm)


def gen_ops_strict():
    rng = np.random.default_rng(777)
    s = {}
    f32 = lambda r, *shape: r.standard_normal(shape).astype(np.float32)  # noqa: E731
    pos = lambda r, *shape: (np.abs(r.standard_normal(shape)) + 0.05).astype(np.float32)  # noqa: E731
    module_case_strict(s, "doubleconv_k2", lambda: DoubleConvDS(6, 16, kernels_per_layer=2),
                       lambda r: [f32(r, 2, 6, 12, 12)], rng)
    module_case_strict(s, "doubleconv_k1", lambda: DoubleConvDS(8, 8, kernels_per_layer=1),
                       lambda r: [f32(r, 2, 8, 8, 8)], rng)
    module_case_strict(s, "doubleconv_k4", lambda: DoubleConvDS(4, 16, kernels_per_layer=4),
                       lambda r: [f32(r, 1, 4, 12, 8)], rng)
    module_case_strict(s, "doubleconv_odd", lambda: DoubleConvDS(6, 10, mid_channels=12, kernels_per_layer=2),
                       lambda r: [f32(r, 2, 6, 9, 11)], rng)               # W % 4 != 0: the non-strip kernels
    module_case_strict(s, "down_k2", lambda: DownDS(6, 12, kernels_per_layer=2), lambda r: [f32(r, 2, 6, 16, 16)], rng)
    module_case_strict(s, "up_k2", lambda: UpDS(16, 6, bilinear=True, kernels_per_layer=2),
                       lambda r: [f32(r, 2, 8, 4, 6), f32(r, 2, 8, 8, 12)], rng)
    module_case_strict(s, "up_pad_k4", lambda: UpDS(8, 4, bilinear=True, kernels_per_layer=4),
                       lambda r: [f32(r, 1, 4, 4, 4), f32(r, 1, 4, 9, 10)], rng)
    module_case_strict(s, "cbam_32", lambda: CBAM(32, reduction_ratio=16), lambda r: [pos(r, 2, 32, 8, 8)], rng)
    module_case_strict(s, "cbam_64_rr8", lambda: CBAM(64, reduction_ratio=8), lambda r: [pos(r, 1, 64, 6, 6)], rng)
    module_case_strict(s, "up_convt_k2", lambda: UpDS(16, 6, bilinear=False, kernels_per_layer=2),
                       lambda r: [f32(r, 2, 16, 4, 6), f32(r, 2, 8, 8, 12)], rng)
    module_case_strict(s, "up_convt_pad_k1", lambda: UpDS(8, 4, bilinear=False, kernels_per_layer=1),
                       lambda r: [f32(r, 1, 8, 4, 4), f32(r, 1, 4, 9, 11)], rng)
    np.savez_compressed(os.path.join(OUT, "ops_strict.npz"), **s)
    print("ops_strict.npz:", len(s), "arrays")


def gen_ops_strict_rows():
    """Round 5 (VERDICT r4 next #1a): tie-free block fixtures whose shapes ROUTE THROUGH the row-walking fused forward
    (csrc/dsrows.hip) and the recompute weight gradient (csrc/dswgrad.hip): kernels_per_layer 2, W % 32 == 0, <= 64 output
    channels, 32 ... 128 input channels -- DoubleConvDS(32, 64) (K = 64 and K = 128 halves) and UpDS(128, 64) (K = 256 and
    K = 128 halves), on planes small enough that a draw with every ReLU >= 2e-4 (of the tensor's rms) away from a tie exists
    (the chance per draw is ~exp(-1.6e-4 * elements): 8 x 32 planes).  Own file and own seed: ops_strict.npz stays as it is."""
    rng = np.random.default_rng(5151)
    s = {}
    f32 = lambda r, *shape: r.standard_normal(shape).astype(np.float32)  # noqa: E731
    module_case_strict(s, "rows_doubleconv_32_64", lambda: DoubleConvDS(32, 64, kernels_per_layer=2),
                       lambda r: [f32(r, 1, 32, 8, 32)], rng, max_tries=20000)
    module_case_strict(s, "rows_doubleconv_64_64_w64", lambda: DoubleConvDS(64, 64, kernels_per_layer=2),
                       lambda r: [f32(r, 1, 64, 4, 64)], rng, max_tries=20000)
    module_case_strict(s, "rows_up_128_64", lambda: UpDS(128, 64, bilinear=True, kernels_per_layer=2),
                       lambda r: [f32(r, 1, 64, 4, 16), f32(r, 1, 64, 8, 32)], rng, max_tries=20000)
    np.savez_compressed(os.path.join(OUT, "ops_strict_rows.npz"), **s)
    print("ops_strict_rows.npz:", len(s), "arrays")


def summarize(store, tag, arr, full_max=8192, nsample=4096, store_idx=True):
    """small tensors in full; large ones as l2 norm + sum + `nsample` evenly spaced samples (the sample positions
    are np.linspace(0, size - 1, nsample) -- stored, or with store_idx=False recomputed by the reader)"""
    a = np.asarray(arr, np.float32).ravel()
    store[tag + "#l2"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    store[tag + "#sum"] = np.float64(a.astype(np.float64).sum())
    store[tag + "#n"] = np.int64(a.size)
    if a.size <= full_max:
        store[tag + "#full"] = np.asarray(arr, np.float32)
    else:
        idx = np.linspace(0, a.size - 1, nsample).astype(np.int64)
        if store_idx:
            store[tag + "#idx"] = idx
        store[tag + "#vals"] = a[idx]


def gen_unet(name, n_channels, n_classes, n, h, w, loss_kind, seed, kpl=2):
    P = oparams.make_smaat_params(n_channels, n_classes, kpl, 16, seed)
    model = SmaAt_UNet(n_channels, n_classes, kernels_per_layer=kpl)
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
                                          param_seed=seed, kpl=kpl)))}
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
    # round 3: fp64 anchors + the reference's own fp32-vs-fp64 error per gradient tensor, so that the tests can use the
    # per-tensor 3 x noise rule of the benchmark-size fixtures instead of one flat bound
    m64 = SmaAt_UNet(n_channels, n_classes, kernels_per_layer=kpl).double()
    m64.load_state_dict({k: torch.from_numpy(np.asarray(v)).double() if np.asarray(v).dtype == np.float32
                         else torch.from_numpy(np.asarray(v)) for k, v in P.items()})
    m64.train()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    lg64 = m64(x64)
    if loss_kind == "mse":
        l64 = torch.nn.functional.mse_loss(lg64.squeeze(1), torch.from_numpy(target).double(), reduction="sum") / n
    else:
        l64 = (lg64 * torch.from_numpy(target).double()).sum()
    l64.backward()
    relv = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))  # noqa: E731
    s["noise/logits"] = np.float64(relv(t2n(logits), lg64.detach().numpy()))
    summarize(s, "dx64", x64.grad.numpy().astype(np.float32))
    s["noise/dx"] = np.float64(relv(t2n(xt.grad), x64.grad.numpy()))
    p32 = dict(model.named_parameters())
    worst = 0.0
    for k, p in m64.named_parameters():
        g64 = p.grad.numpy()
        summarize(s, "grad64/" + k, g64.astype(np.float32))
        s["noise/" + k] = np.float64(relv(t2n(p32[k].grad), g64))
        if not (".double_conv." in k and k.endswith(("depthwise.bias", "pointwise.bias"))):
            worst = max(worst, float(s["noise/" + k]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **s)
    print(name, "loss", loss.item(), "arrays", len(s), "worst fp32-vs-fp64 gradient noise of the reference", worst)


def _loss(kind, logits, target, n):
    if kind == "precip":  # reference: models/regression_lightning.py:57-65
        return torch.nn.functional.mse_loss(logits.squeeze(1), target, reduction="sum") / n
    return torch.nn.functional.cross_entropy(logits, target)  # reference: train_SmaAtUNet.py:183 (nn.CrossEntropyLoss())


_ACT_NAMES = ["inc", "cbam1", "down1", "cbam2", "down2", "cbam3", "down3", "cbam4", "down4", "cbam5", "up1", "up2",
              "up3", "up4"]
_ACT_RENAME = dict(inc="x1", cbam1="x1Att", down1="x2", cbam2="x2Att", down2="x3", cbam3="x3Att", down3="x4",
                   cbam4="x4Att", down4="x5", cbam5="x5Att", up1="u1", up2="u2", up3="u3", up4="u4")


def _forward_checkpointed(model, x):
    """models/SmaAt_UNet.py:41-57 with every top-level block of the REFERENCE model under activation checkpointing:
    the fp64 anchor of the batch-32 case would otherwise need ~66 GB.  Same blocks, same wiring, same arithmetic (a
    checkpointed block recomputes its forward in backward: the values are identical; only the running statistics of
    this throw-away fp64 model are updated twice, and they are not stored)."""
    from torch.utils.checkpoint import checkpoint
    ck = lambda m, *a: checkpoint(m, *a, use_reentrant=False)  # noqa: E731
    x1 = ck(model.inc, x)
    x1a = ck(model.cbam1, x1)
    x2 = ck(model.down1, x1)
    x2a = ck(model.cbam2, x2)
    x3 = ck(model.down2, x2)
    x3a = ck(model.cbam3, x3)
    x4 = ck(model.down3, x3)
    x4a = ck(model.cbam4, x4)
    x5 = ck(model.down4, x4)
    x5a = ck(model.cbam5, x5)
    u = ck(model.up1, x5a, x4a)
    u = ck(model.up2, u, x3a)
    u = ck(model.up3, u, x2a)
    u = ck(model.up4, u, x1a)
    return model.outc(u)


def gen_unet_big(name, kind, n_channels, n_classes, n, h, w, seed, n_eval=1, lean64=False):
    """Benchmark-size cases (VERDICT r1 missing #7): the REAL reference at BASELINE.json's sizes, train AND eval mode.
    Inputs are regenerated from the seed (oracle.params.synthetic_case), every tensor is stored as a summary.
      train/*  : one training step from the seeded parameters (logits, loss, hooked activations, gradients with
                 their fp64 anchors + the reference's own fp32-vs-fp64 noise, running statistics afterwards)
      eval/*   : after that step, model.eval(): forward at batch n and batch n_eval on a second input (per-frame
                 forward latency path, reference call stack D), and the gradients of an eval-mode backward
                 (BatchNorm running statistics are constants there)"""
    P = oparams.make_smaat_params(n_channels, n_classes, 2, 16, seed)
    x, target = oparams.synthetic_case(kind, n, n_channels, h, w, n_classes, seed + 100)
    xe, te = oparams.synthetic_case(kind, n, n_channels, h, w, n_classes, seed + 200)
    model = SmaAt_UNet(n_channels, n_classes)
    load_np_state(model, P)
    model.train()
    acts = {}
    hooks = [getattr(model, nm).register_forward_hook(lambda m, i, o, nm=nm: acts.__setitem__(nm, t2n(o)))
             for nm in _ACT_NAMES]
    xt = torch.from_numpy(x).requires_grad_(True)
    logits = model(xt)
    loss = _loss(kind, logits, torch.from_numpy(target), n)
    loss.backward()
    s = {"meta": np.array(json.dumps(dict(kind=kind, n_channels=n_channels, n_classes=n_classes, n=n, h=h, w=w,
                                          param_seed=seed, n_eval=n_eval, torch=torch.__version__,
                                          threads=torch.get_num_threads()))),
         "train/loss": np.float64(loss.item())}
    summarize(s, "train/logits", t2n(logits), store_idx=False)
    for k, v in acts.items():
        summarize(s, "train/act/" + _ACT_RENAME[k], v, store_idx=False)
    summarize(s, "train/dx", t2n(xt.grad), store_idx=False)
    g32 = {k: t2n(p.grad) for k, p in model.named_parameters()}
    for k, v in g32.items():
        summarize(s, "train/grad/" + k, v, 2048, 2048, store_idx=False)
    for k, v in model.state_dict().items():
        if "running" in k:
            s["train/after/" + k] = t2n(v)
    # fp64 anchors of the training gradients
    m64 = SmaAt_UNet(n_channels, n_classes)
    load_np_state(m64, P)
    m64 = m64.double().train()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    t64 = torch.from_numpy(target).double() if kind == "precip" else torch.from_numpy(target)
    _loss(kind, _forward_checkpointed(m64, x64) if lean64 else m64(x64), t64, n).backward()
    worst = 0.0
    for k, p64 in m64.named_parameters():
        g64 = p64.grad.numpy()
        summarize(s, "train/grad64/" + k, g64.astype(np.float32), 2048, 2048, store_idx=False)
        noise = float(np.linalg.norm(g32[k].astype(np.float64) - g64) / max(np.linalg.norm(g64), 1e-30))
        s["train/noise/" + k] = np.float64(noise)
        if not (".double_conv." in k and k.endswith(("depthwise.bias", "pointwise.bias"))):
            worst = max(worst, noise)
    del m64
    # ---- eval mode (running statistics of the step above) ----
    for hk in hooks:
        hk.remove()
    model.eval()
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        summarize(s, "eval/logits_b1", t2n(model(torch.from_numpy(xe[:n_eval]))), store_idx=False)
    xet = torch.from_numpy(xe).requires_grad_(True)
    le = model(xet)
    summarize(s, "eval/logits", t2n(le), store_idx=False)
    losse = _loss(kind, le, torch.from_numpy(te), n)
    losse.backward()
    s["eval/loss"] = np.float64(losse.item())
    summarize(s, "eval/dx", t2n(xet.grad), store_idx=False)
    for k, p in model.named_parameters():
        summarize(s, "eval/grad/" + k, t2n(p.grad), 2048, 2048, store_idx=False)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **s)
    print(name, "train loss", loss.item(), "eval loss", losse.item(), "arrays", len(s), "worst fp32-vs-fp64 grad", worst)



def _pack_f16(store, tag, tensors):
    """a list of float32 arrays as ONE float16 vector with a per-tensor scale (max-abs -> 1): 2 bytes per element,
    relative error 2^-11 per element -- enough for norms and cosines of 4 M-element gradients"""
    scales = np.array([max(float(np.abs(t).max()), 1e-30) for t in tensors], np.float64)
    store[tag + "#scale"] = scales
    store[tag + "#sizes"] = np.array([t.size for t in tensors], np.int64)
    store[tag + "#f16"] = np.concatenate([(t.ravel().astype(np.float64) / s).astype(np.float16)
                                          for t, s in zip(tensors, scales)])


def gen_autocast(name, n, h, w, param_seed, input_seed, steps=4):
    """Mixed-precision yardstick from the REFERENCE itself (VERDICT r3 next #1b).  The reference's mixed precision is
    Lightning `precision="16-mixed"` = torch.autocast around models/SmaAt_UNet.py:41-57 with the loss
    (models/regression_lightning.py:57-65) in float32.  Three runs of the reference from the same state on the same
    batch: float32, float64, and float32 parameters under torch.autocast("cpu", torch.bfloat16).  Stored: the float32
    run's logits (full) and flat gradient (float16, per-tensor scale), the float64 logits, and how far the AUTOCAST run is
    from both (logits rel-L2, 1 - cosine of the flat gradient, per-step losses of `steps` Adam steps, lr 1e-3) -- the
    tests bound this implementation's bf16 mode by 1.25 x those distances."""
    from oracle import smaat_oracle as O
    P = oparams.make_smaat_params(12, 1, 2, 16, param_seed)
    xn, yn = O.synthetic_precip(n, 12, h, w, seed=input_seed)

    def run(mode):
        model = SmaAt_UNet(12, 1)
        load_np_state(model, P)
        dt = torch.float64 if mode == "f64" else torch.float32
        model = model.to(dt).train()
        x, y = torch.from_numpy(xn).to(dt), torch.from_numpy(yn).to(dt)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        first, grads, losses = None, None, []
        for _ in range(steps):
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=(mode == "autocast")):
                out = model(x)
            loss = torch.nn.functional.mse_loss(out.float().squeeze(1) if mode == "autocast" else out.squeeze(1), y,
                                                reduction="sum") / n
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if first is None:
                first = out.detach().double().numpy()
                grads = [p.grad.detach().double().numpy().copy() for p in model.parameters()]
                names = [k for k, _ in model.named_parameters()]
            opt.step()
            losses.append(float(loss.item()))
        return first, grads, losses, names

    o32, g32, l32, names = run("f32")
    o64, g64, l64, _ = run("f64")
    oac, gac, lac, _ = run("autocast")
    flat = lambda g: np.concatenate([t.ravel() for t in g])  # noqa: E731
    f32v, f64v, facv = flat(g32), flat(g64), flat(gac)
    rl = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
    cs = lambda a, b: float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))  # noqa: E731
    s = {"meta": np.array(json.dumps(dict(n=n, h=h, w=w, n_channels=12, n_classes=1, param_seed=param_seed,
                                          input_seed=input_seed, steps=steps, lr=1e-3, torch=torch.__version__,
                                          names=names))),
         "logits32": o32.astype(np.float32), "logits64": o64.astype(np.float32),
         "losses32": np.array(l32), "losses64": np.array(l64), "losses_autocast": np.array(lac),
         "autocast/logits_vs32": np.float64(rl(oac, o32)), "autocast/logits_vs64": np.float64(rl(oac, o64)),
         "f32/logits_vs64": np.float64(rl(o32, o64)),
         "autocast/one_minus_cos_vs32": np.float64(1.0 - cs(facv, f32v)),
         "autocast/one_minus_cos_vs64": np.float64(1.0 - cs(facv, f64v)),
         "f32/one_minus_cos_vs64": np.float64(1.0 - cs(f32v, f64v)),
         "autocast/grad_rel_vs32": np.float64(rl(facv, f32v))}
    # round 5 (VERDICT r4 next #1b): the same two distances PER PARAMETER TENSOR (order = meta["names"]), so that a wrong
    # gradient in one mixed-precision layer cannot hide in the norm of the flat vector
    s["autocast/one_minus_cos_per_tensor_vs32"] = np.array([1.0 - cs(a.ravel(), b.ravel()) if np.linalg.norm(b) > 0 else 0.0
                                                            for a, b in zip(gac, g32)])
    s["autocast/rel_per_tensor_vs32"] = np.array([rl(a.ravel(), b.ravel()) if np.linalg.norm(b) > 0 else 0.0
                                                  for a, b in zip(gac, g32)])
    s["f32/rel_per_tensor_vs64"] = np.array([rl(a.ravel(), b.ravel()) if np.linalg.norm(b) > 0 else 0.0
                                             for a, b in zip(g32, g64)])
    _pack_f16(s, "grad32", [t.astype(np.float32) for t in g32])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **s)
    print(name, {k: float(v) for k, v in s.items() if "/" in k and np.ndim(v) == 0}, "losses32", l32, "autocast", lac, "f64", l64)


def gen_eval_noise(name, k=2):
    """The reference's OWN float32-vs-float64 distance on the eval-mode input gradient of a benchmark-size case (the
    fixture `name` stores fp64 anchors for the training gradients only).  Eval mode has no cross-sample coupling
    (BatchNorm on the running statistics), so the first k samples of the eval batch are evaluated on their own, with
    the running statistics the fixture recorded after its training step: dx of sample i here = dx of sample i in the
    batch (up to the 1/n of the loss, which the relative error does not see)."""
    g = np.load(os.path.join(OUT, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    P = oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], 2, 16, meta["param_seed"])
    for key in g.files:
        if key.startswith("train/after/"):
            P[key[12:]] = g[key]
    xe, te = oparams.synthetic_case(meta["kind"], meta["n"], meta["n_channels"], meta["h"], meta["w"], meta["n_classes"],
                                    meta["param_seed"] + 200)
    res = {}
    for dt in (torch.float32, torch.float64):
        m = SmaAt_UNet(meta["n_channels"], meta["n_classes"])
        load_np_state(m, P)
        m = m.to(dt).eval()
        x = torch.from_numpy(xe[:k]).to(dt).requires_grad_(True)
        t = torch.from_numpy(te[:k])
        t = t.to(dt) if meta["kind"] == "precip" else t
        out = m(x)
        (_loss(meta["kind"], out, t, k) * (k / meta["n"])).backward()  # the batch's loss restricted to these samples
        res[dt] = (out.detach().double().numpy(), x.grad.double().numpy())
    rl = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
    s = {"meta": np.array(json.dumps(dict(case=name, samples=k))),
         "eval/noise/logits": np.float64(rl(res[torch.float32][0], res[torch.float64][0])),
         "eval/noise/dx": np.float64(rl(res[torch.float32][1], res[torch.float64][1]))}
    # cross-check with the batch run of the fixture: the same values at the sampled positions of these k samples
    full = np.zeros((meta["n"],) + res[torch.float32][1].shape[1:], np.float32)
    full[:k] = res[torch.float32][1]
    idx = np.linspace(0, full.size - 1, 4096).astype(np.int64)
    sel = idx < res[torch.float32][1].size
    s["eval/dx_batch_vs_alone"] = np.float64(rl(full.ravel()[idx][sel].astype(np.float64),
                                                g["eval/dx#vals"][sel].astype(np.float64)))
    # ... and the reference against ITSELF on the whole batch: the same float32 evaluation sample by sample on ONE thread
    # (another summation order inside ATen) against the fixture's 8-thread batch run, measured exactly as the tests measure
    # (at the fixture's 4096 sampled positions, where a single ReLU / max-pool decision that flips under round-off can
    # dominate: 2.9e-3 at batch 32 although the full-tensor distance is 3.8e-4).  The eval-mode dx bound of the tests is
    # 1.5 x this figure (floor 2e-3).
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    m = SmaAt_UNet(meta["n_channels"], meta["n_classes"])
    load_np_state(m, P)
    m.eval()
    rows = []
    for i in range(meta["n"]):
        x = torch.from_numpy(xe[i:i + 1]).requires_grad_(True)
        (_loss(meta["kind"], m(x), torch.from_numpy(te[i:i + 1]), 1) / meta["n"]).backward()
        rows.append(x.grad.numpy().copy())
    torch.set_num_threads(nt)
    alone = np.concatenate(rows)
    ref = g["eval/dx#vals"].astype(np.float64)
    s["eval/self/dx_sampled"] = np.float64(rl(alone.ravel()[idx].astype(np.float64), ref))
    np.savez_compressed(os.path.join(OUT, name + "_evalnoise.npz"), **s)
    print(name + "_evalnoise", {k_: float(v) for k_, v in s.items() if k_ != "meta"})


class _RefVariant(torch.nn.Module):
    """The sibling networks of /root/reference/models/unet_precip_regression_lightning.py wired from the
    REFERENCE blocks.  The reference classes themselves are Lightning modules (`lightning` is not in this
    image), so their constructor (:87-106 UNetDS, :168-190 UNetDSAttention4CBAMs) and forward (:108-118,
    :192-208) are restated here line by line; every block is the reference's own."""

    def __init__(self, n_channels, n_classes, kpl, cbams, rr=16):
        super().__init__()
        self.cbams = cbams
        self.inc = DoubleConvDS(n_channels, 64, kernels_per_layer=kpl)
        if cbams:
            self.cbam1 = CBAM(64, reduction_ratio=rr)
        self.down1 = DownDS(64, 128, kernels_per_layer=kpl)
        if cbams:
            self.cbam2 = CBAM(128, reduction_ratio=rr)
        self.down2 = DownDS(128, 256, kernels_per_layer=kpl)
        if cbams:
            self.cbam3 = CBAM(256, reduction_ratio=rr)
        self.down3 = DownDS(256, 512, kernels_per_layer=kpl)
        if cbams:
            self.cbam4 = CBAM(512, reduction_ratio=rr)
        self.down4 = DownDS(512, 512, kernels_per_layer=kpl)
        self.up1 = UpDS(1024, 256, True, kernels_per_layer=kpl)
        self.up2 = UpDS(512, 128, True, kernels_per_layer=kpl)
        self.up3 = UpDS(256, 64, True, kernels_per_layer=kpl)
        self.up4 = UpDS(128, 64, True, kernels_per_layer=kpl)
        self.outc = OutConv(64, n_classes)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        if self.cbams:  # :192-208: attention on the four skips, un-attended bottleneck
            s1, s2, s3, s4 = self.cbam1(x1), self.cbam2(x2), self.cbam3(x3), self.cbam4(x4)
        else:           # :108-118
            s1, s2, s3, s4 = x1, x2, x3, x4
        x = self.up1(x5, s4)
        x = self.up2(x, s3)
        x = self.up3(x, s2)
        x = self.up4(x, s1)
        return self.outc(x)


def gen_variant(name, cbams, kpl, n_channels, n_classes, n, h, w, seed):
    keys = oparams.unetds_keys(n_channels, n_classes, kpl, 16, cbams)
    model = _RefVariant(n_channels, n_classes, kpl, cbams)
    ref = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref == [(k, tuple(s)) for k, s in keys], "oracle.params.unetds_keys != reference-block state_dict"
    load_np_state(model, oparams.fill(keys, seed))
    model.train()
    rng = np.random.default_rng(seed + 100)
    u = rng.random((n, n_channels, h, w), dtype=np.float32)
    x = np.where(u > 0.6, (u - 0.6) / 0.4, 0).astype(np.float32)
    cot = rng.standard_normal((n, n_classes, h, w)).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    logits = model(xt)
    (logits * torch.from_numpy(cot)).sum().backward()
    s = {"x": x, "cot": cot, "logits": t2n(logits),
         "meta": np.array(json.dumps(dict(n_channels=n_channels, n_classes=n_classes, n=n, h=h, w=w, kpl=kpl,
                                          cbams=cbams, param_seed=seed)))}
    summarize(s, "dx", t2n(xt.grad), 1024, 1024)
    for k, p in model.named_parameters():
        summarize(s, "grad/" + k, t2n(p.grad), 1024, 1024)
    for k, v in model.state_dict().items():
        if "running" in k:
            s["after/" + k] = t2n(v)
    # the same reference blocks in float64: these random-parameter networks amplify fp32 round-off end to
    # end (the reference's OWN fp32 gradients are up to 3e-2 from its fp64 ones), so the parity test bounds
    # "error against fp64" by the reference's own figure where the plain bound fails
    m64 = _RefVariant(n_channels, n_classes, kpl, cbams)
    load_np_state(m64, oparams.fill(keys, seed))
    m64 = m64.double().train()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    (m64(x64) * torch.from_numpy(cot).double()).sum().backward()
    worst = 0.0
    for (k, p), (_, p64) in zip(model.named_parameters(), m64.named_parameters()):
        g64 = p64.grad.numpy()
        summarize(s, "grad64/" + k, g64.astype(np.float32), 1024, 1024)
        noise = float(np.linalg.norm(t2n(p.grad).astype(np.float64) - g64) / max(np.linalg.norm(g64), 1e-30))
        s["noise/" + k] = np.float64(noise)
        if not (".double_conv." in k and k.endswith(("depthwise.bias", "pointwise.bias"))):  # exact-zero gradients
            worst = max(worst, noise)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **s)
    print(name, "arrays", len(s), "logits l2", float(np.linalg.norm(s["logits"])), "worst fp32-vs-fp64 grad", worst)


def _variant_grads(model, x, cot, dtype):
    model = model.to(dtype).train()
    for p in model.parameters():
        p.grad = None
    xt = torch.from_numpy(x).to(dtype).requires_grad_(True)
    logits = model(xt)
    (logits * torch.from_numpy(cot).to(dtype)).sum().backward()
    return t2n(logits), t2n(xt.grad), {k: t2n(p.grad) for k, p in model.named_parameters()}


def gen_variant_strict(name, cbams, kpl, n_channels, n_classes, n, h, w, seed0, max_tries=200, draws=8, convt=False):
    """Sibling-network fixture with fp64 anchors and a MEASURED noise floor.
    End-to-end gradients of these small random-parameter networks are not a smooth function of round-off: an
    activation within ~1e-5 (relative) of zero, or a max-pool / CBAM-max near-tie, takes the other branch under
    any change of summation order, and ONE flipped element moves a gradient tensor by 1e-4 .. 4e-3 -- in the
    reference itself as well (measured below).  A fixture cannot be searched free of such elements at the fp32
    noise level of the forward pass (~1e-5 on the logits for the reference AND for the HIP path): there are
    ~4e5 activations per case.  So the fixture
      (a) is searched (input seed) for a case whose fp32 reference gradients sit within 1e-4 of the fp64 ones on
          every tensor, with 8 threads and with 1 thread: the stored fp32/fp64 tensors themselves contain no flip;
      (b) records what the reference's gradients do under perturbations of the size of fp32 forward noise:
          `draws` fp64 runs on x * (1 + 1e-6 n); per tensor the largest relative change ("sens/<key>") and the
          largest over all tensors ("sens_global").
    tests/test_host_emu.py::run_variant holds every tensor of the HIP path (against the fp64 anchors) to
    2 x sens_global: the error of ONE flip anywhere in the reference -- no percentile, no minimum over references."""
    # convt: the reference SmaAt_UNet itself with bilinear=False (ConvTranspose2d up path, models/SmaAt_UNet.py:31-37,
    # unet_parts_depthwise_separable.py:72-73) instead of one of the sibling networks
    keys = (oparams.smaat_unet_keys(n_channels, n_classes, kpl, 16, bilinear=False) if convt
            else oparams.unetds_keys(n_channels, n_classes, kpl, 16, cbams))
    zero = lambda k: ".double_conv." in k and k.endswith(("depthwise.bias", "pointwise.bias"))  # noqa: E731
    relv = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))  # noqa: E731
    for seed in range(seed0, seed0 + max_tries):
        P = oparams.fill(keys, seed)
        rng = np.random.default_rng(seed + 100)
        u = rng.random((n, n_channels, h, w), dtype=np.float32)
        x = np.where(u > 0.6, (u - 0.6) / 0.4, 0).astype(np.float32)
        cot = rng.standard_normal((n, n_classes, h, w)).astype(np.float32)

        def fresh():
            m = (SmaAt_UNet(n_channels, n_classes, kernels_per_layer=kpl, bilinear=False) if convt
                 else _RefVariant(n_channels, n_classes, kpl, cbams))
            load_np_state(m, P)
            return m
        torch.set_num_threads(8)
        lg32, dx32, g32 = _variant_grads(fresh(), x, cot, torch.float32)
        _, dx64, g64 = _variant_grads(fresh(), x, cot, torch.float64)
        noise = {k: relv(g32[k], g64[k]) for k in g32 if not zero(k)}
        if max(noise.values()) > 1e-4:
            continue
        torch.set_num_threads(1)
        _, _, g32b = _variant_grads(fresh(), x, cot, torch.float32)
        torch.set_num_threads(8)
        if max(relv(g32b[k], g64[k]) for k in noise) > 1e-4:
            continue
        break
    else:
        raise RuntimeError(f"{name}: no flip-free fp32 reference run in {max_tries} seeds")
    sens = {k: 0.0 for k in noise}
    sens_dx = 0.0
    for _ in range(draws):
        xp = x.astype(np.float64) * (1 + 1e-6 * rng.standard_normal(x.shape))
        m64 = fresh().double().train()
        xpt = torch.from_numpy(xp).requires_grad_(True)
        (m64(xpt) * torch.from_numpy(cot).double()).sum().backward()
        for k, p in m64.named_parameters():
            if k in sens:
                sens[k] = max(sens[k], relv(p.grad.numpy(), g64[k]))
        sens_dx = max(sens_dx, relv(xpt.grad.numpy(), dx64))
    model = fresh().train()
    model(torch.from_numpy(x))  # running statistics after one step
    s = {"x": x, "cot": cot, "logits": lg32,
         "meta": np.array(json.dumps(dict(n_channels=n_channels, n_classes=n_classes, n=n, h=h, w=w, kpl=kpl,
                                          cbams=cbams, param_seed=seed, strict=True, convt=bool(convt)))),
         "sens_global": np.float64(max(max(sens.values()), sens_dx))}
    summarize(s, "dx", dx32, 1024, 1024)
    summarize(s, "dx64", dx64.astype(np.float32), 1024, 1024)
    s["noise/dx"] = np.float64(relv(dx32, dx64))
    for k in g32:
        summarize(s, "grad/" + k, g32[k], 1024, 1024)
        summarize(s, "grad64/" + k, g64[k].astype(np.float32), 1024, 1024)
        s["noise/" + k] = np.float64(relv(g32[k], g64[k]))
        if k in sens:
            s["sens/" + k] = np.float64(sens[k])
    for k, v in model.state_dict().items():
        if "running" in k:
            s["after/" + k] = t2n(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **s)
    print(name, "seed", seed, "worst fp32-vs-fp64 grad", max(noise.values()), "sens_global", float(s["sens_global"]),
          "median sens", float(np.median(list(sens.values()))), "arrays", len(s))


def gen_metrics():
    """tests/golden/precip_metrics.npz from the reference's own PrecipitationMetrics
    (/root/reference/metric/precipitation_metrics.py).  `torchmetrics` is not installed; the class only uses
    Metric.__init__ and Metric.add_state, so a ten-line stand-in for that base class is enough to import and run
    the reference code itself."""
    import types
    tm = types.ModuleType("torchmetrics")

    class Metric:  # stand-in for torchmetrics.Metric: stores the states as attributes
        def __init__(self, dist_sync_on_step=False):
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default.clone()
            setattr(self, name, default.clone())

    tm.Metric = Metric
    sys.modules["torchmetrics"] = tm
    from metric.precipitation_metrics import PrecipitationMetrics as RefMetrics
    rng = np.random.default_rng(77)
    s = {}
    cases = [dict(tag="default", threshold=0.5, denormalize=True, shape=(3, 1, 24, 20), squeeze=True),
             dict(tag="nodenorm", threshold=0.5, denormalize=False, shape=(2, 1, 16, 16), squeeze=False),
             dict(tag="thr2", threshold=2.0, denormalize=True, shape=(4, 1, 9, 11), squeeze=True)]
    for c in cases:
        m = RefMetrics(threshold=c["threshold"], denormalize=c["denormalize"])
        for b in range(3):
            n, ch, h, w = c["shape"]
            u = rng.random((n, h, w), dtype=np.float32)
            target = np.where(u > 0.6, (u - 0.6) * 0.02, 0).astype(np.float32)  # normalised rain rates, mostly dry
            preds = (target + 0.002 * rng.standard_normal((n, ch, h, w)).astype(np.float32)[:, 0]).astype(np.float32)
            preds = preds[:, None] if not c["squeeze"] else preds[:, None]  # model output [N,1,H,W]
            if b == 1 and c["tag"] == "default":
                bad = preds.copy()
                bad[0, 0, 0, 0] = np.nan  # a NaN batch must be ignored (:46-48)
                m.update(torch.from_numpy(bad), torch.from_numpy(target))
                s[f"{c['tag']}/b{b}/preds"], s[f"{c['tag']}/b{b}/target"] = bad, target
                continue
            m.update(torch.from_numpy(preds), torch.from_numpy(target))
            s[f"{c['tag']}/b{b}/preds"], s[f"{c['tag']}/b{b}/target"] = preds, target
        out = m.compute()
        for k, v in out.items():
            s[f"{c['tag']}/compute/{k}"] = np.float64(float(v))
        for k in ("total_loss", "total_loss_denorm", "total_samples", "total_pixels", "total_tp", "total_fp", "total_tn",
                  "total_fn"):
            s[f"{c['tag']}/state/{k}"] = np.float64(float(getattr(m, k)))
        s[f"{c['tag']}/cfg"] = np.array(json.dumps(dict(threshold=c["threshold"], denormalize=c["denormalize"])))
    np.savez_compressed(os.path.join(OUT, "precip_metrics.npz"), **s)
    print("precip_metrics.npz:", len(s), "arrays")


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


def _jobs():
    J = {}
    J["metrics"] = gen_metrics
    J["keys"] = gen_keys
    J["ops"] = gen_ops
    J["ops_eval"] = gen_ops_eval
    J["ops_strict"] = gen_ops_strict
    J["ops_strict_rows"] = gen_ops_strict_rows
    J["ops_generic"] = gen_ops_generic
    # benchmark-size cases, train + eval mode (inputs regenerated from the seed, summaries only)
    J["unet_12x1_n2_288"] = lambda: gen_unet_big("unet_12x1_n2_288", "precip", 12, 1, 2, 288, 288, 7)   # configs[1] shape
    J["unet_3x21_n2_256"] = lambda: gen_unet_big("unet_3x21_n2_256", "voc", 3, 21, 2, 256, 256, 8)      # configs[4] shape
    J["unet_12x1_n3_64x48_eval"] = lambda: gen_unet_big("unet_12x1_n3_64x48_eval", "precip", 12, 1, 3, 64, 48, 9)
    # round 4 (VERDICT r3 next #1a): the EXACT BASELINE.json configs -- batch 32 at 288 x 288 (configs[1], ~35 GB and
    # minutes of CPU; the fp64 anchor runs block-checkpointed) and batch 16 VOC at 256 x 256 (configs[4])
    J["unet_12x1_n32_288"] = lambda: gen_unet_big("unet_12x1_n32_288", "precip", 12, 1, 32, 288, 288, 17, lean64=True)
    J["unet_3x21_n16_256"] = lambda: gen_unet_big("unet_3x21_n16_256", "voc", 3, 21, 16, 256, 256, 18, lean64=True)
    J["unet_12x1_n32_288_evalnoise"] = lambda: gen_eval_noise("unet_12x1_n32_288")
    J["unet_3x21_n16_256_evalnoise"] = lambda: gen_eval_noise("unet_3x21_n16_256")
    J["unet_12x1_n2_288_evalnoise"] = lambda: gen_eval_noise("unet_12x1_n2_288")
    # round 4 (VERDICT r3 next #1b): the reference under torch.autocast(bfloat16)
    J["autocast_bf16_n2_64"] = lambda: gen_autocast("autocast_bf16_n2_64", 2, 64, 64, 3, 11)
    J["autocast_bf16_n2_288"] = lambda: gen_autocast("autocast_bf16_n2_288", 2, 288, 288, 7, 107)
    J["unet_12x1_n2_32"] = lambda: gen_unet("unet_12x1_n2_32", 12, 1, 2, 32, 32, "mse", 0)
    J["unet_12x1_n2_64x48"] = lambda: gen_unet("unet_12x1_n2_64x48", 12, 1, 2, 64, 48, "mse", 1)
    J["unet_3x21_n1_32"] = lambda: gen_unet("unet_3x21_n1_32", 3, 21, 1, 32, 32, "cot", 2)
    # round 4 (VERDICT r3 missing #6): the network at kernels_per_layer = 3 (models/SmaAt_UNet.py:12 accepts any integer):
    # outside the fused kernels, every block on the general depthwise geometry path
    J["unet_4x2_k3_n2_32"] = lambda: gen_unet("unet_4x2_k3_n2_32", 4, 2, 2, 32, 32, "cot", 12, kpl=3)
    # sibling networks: no attention / four CBAMs, kernels_per_layer 1, 2 and 4;
    # 48 x 40: the width is not a multiple of 16, so UpDS has to F.pad (unet_parts_depthwise_separable.py:78-81).
    # Tie-free fixtures (gen_variant_strict), incl. the four-CBAM network at kernels_per_layer = 4
    J["strict_unetds_k2_n2_32"] = lambda: gen_variant_strict("strict_unetds_k2_n2_32", 0, 2, 12, 1, 2, 32, 32, 1000)
    J["strict_unetds_k1_n1_48x40"] = lambda: gen_variant_strict("strict_unetds_k1_n1_48x40", 0, 1, 5, 2, 1, 48, 40, 2000)
    J["strict_unetds_k4_n1_32"] = lambda: gen_variant_strict("strict_unetds_k4_n1_32", 0, 4, 3, 2, 1, 32, 32, 3000)
    J["strict_unetds4cbam_k2_n2_32"] = lambda: gen_variant_strict("strict_unetds4cbam_k2_n2_32", 4, 2, 12, 1, 2, 32, 32, 4000)
    J["strict_unetds4cbam_k4_n1_32"] = lambda: gen_variant_strict("strict_unetds4cbam_k4_n1_32", 4, 4, 3, 2, 1, 32, 32, 5000)
    J["strict_smaat_convt_k2_n2_32"] = lambda: gen_variant_strict("strict_smaat_convt_k2_n2_32", 5, 2, 12, 1, 2, 32, 32,
                                                                  6000, convt=True)  # bilinear=False
    return J


if __name__ == "__main__":
    # `python oracle/gen_golden.py` regenerates everything; `python oracle/gen_golden.py NAME ...` only those fixtures
    jobs = _jobs()
    for nm in (sys.argv[1:] or list(jobs)):
        jobs[nm]()
