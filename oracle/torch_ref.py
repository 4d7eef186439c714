"""torch.nn.functional restatement of the reference forward (CPU, autograd backward).

TEST INFRASTRUCTURE ONLY (same rules as oracle/smaat_oracle.py).  This is the closest thing
to "the reference on the host cores" that can travel to the GPU box (the real reference under
/root/reference cannot): the same ATen CPU operators (MKLDNN convolutions, native_batch_norm,
...) that models/SmaAt_UNet.py:41-57 dispatches to, driven from the flat state_dict.  It is the
`cpu_baseline` leg of bench.py (kind = "port") and a second oracle in the tests; it is pinned
against the reference-generated goldens by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# Mixed-precision emulation (BASELINE configs[3]): when set to a predicate f(x) -> bool, the pointwise convs of the
# DepthwiseSeparableConvs for which it is true see their two operands rounded to bf16 (round-to-nearest-even) and
# accumulate in f32 -- exactly what smaat_unet_amd's "bf16" matrix mode computes.  Used by the tests only.
PW_BF16 = None


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _half(P, x, pre, bn, kpl, training, buffers):
    # DepthwiseSeparableConv (models/layers.py:47-50) -> BatchNorm2d -> ReLU
    x = F.conv2d(x, P[pre + ".depthwise.weight"], P[pre + ".depthwise.bias"], padding=1, groups=x.shape[1])
    if PW_BF16 is not None and PW_BF16(x):
        x = F.conv2d(_bf16(x), _bf16(P[pre + ".pointwise.weight"]), P[pre + ".pointwise.bias"])
    else:
        x = F.conv2d(x, P[pre + ".pointwise.weight"], P[pre + ".pointwise.bias"])
    x = F.batch_norm(x, buffers[bn + ".running_mean"], buffers[bn + ".running_var"], P[bn + ".weight"],
                     P[bn + ".bias"], training, 0.1, 1e-5)
    return F.relu(x)


def _double(P, x, pre, kpl, training, buffers):
    x = _half(P, x, pre + ".double_conv.0", pre + ".double_conv.1", kpl, training, buffers)
    return _half(P, x, pre + ".double_conv.3", pre + ".double_conv.4", kpl, training, buffers)


def _cbam(P, x, pre, training, buffers):
    # ChannelAttention models/layers.py:105-111
    ca = pre + ".channel_att.MLP"
    avg = F.adaptive_avg_pool2d(x, 1).flatten(1)
    mx = F.adaptive_max_pool2d(x, 1).flatten(1)

    def mlp(v):
        return F.linear(F.relu(F.linear(v, P[ca + ".1.weight"], P[ca + ".1.bias"])), P[ca + ".3.weight"],
                        P[ca + ".3.bias"])
    x = x * torch.sigmoid(mlp(avg) + mlp(mx))[:, :, None, None]
    # SpatialAttention models/layers.py:122-129
    sp = pre + ".spatial_att"
    m = torch.cat([x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True)[0]], dim=1)
    m = F.conv2d(m, P[sp + ".conv.weight"], None, padding=P[sp + ".conv.weight"].shape[-1] // 2)
    m = F.batch_norm(m, buffers[sp + ".bn.running_mean"], buffers[sp + ".bn.running_var"], P[sp + ".bn.weight"],
                     P[sp + ".bn.bias"], training, 0.1, 1e-5)
    return x * torch.sigmoid(m)


def _up(P, x1, x2, pre, kpl, training, buffers):
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double(P, torch.cat([x2, x1], dim=1), pre + ".conv", kpl, training, buffers)


def forward(P, x, kpl=2, training=True, buffers=None):
    """P: name -> tensor (parameters, requires_grad as the caller wishes); buffers: running stats."""
    if buffers is None:
        buffers = {k: v for k, v in P.items() if "running" in k}
    x1 = _double(P, x, "inc", kpl, training, buffers)
    x1a = _cbam(P, x1, "cbam1", training, buffers)
    x2 = _double(P, F.max_pool2d(x1, 2), "down1.maxpool_conv.1", kpl, training, buffers)
    x2a = _cbam(P, x2, "cbam2", training, buffers)
    x3 = _double(P, F.max_pool2d(x2, 2), "down2.maxpool_conv.1", kpl, training, buffers)
    x3a = _cbam(P, x3, "cbam3", training, buffers)
    x4 = _double(P, F.max_pool2d(x3, 2), "down3.maxpool_conv.1", kpl, training, buffers)
    x4a = _cbam(P, x4, "cbam4", training, buffers)
    x5 = _double(P, F.max_pool2d(x4, 2), "down4.maxpool_conv.1", kpl, training, buffers)
    x5a = _cbam(P, x5, "cbam5", training, buffers)
    u = _up(P, x5a, x4a, "up1", kpl, training, buffers)
    u = _up(P, u, x3a, "up2", kpl, training, buffers)
    u = _up(P, u, x2a, "up3", kpl, training, buffers)
    u = _up(P, u, x1a, "up4", kpl, training, buffers)
    return F.conv2d(u, P["outc.conv.weight"], P["outc.conv.bias"])


def params_from_numpy(Pn, requires_grad=True):
    out = {}
    for k, v in Pn.items():
        t = torch.from_numpy(v.copy()) if hasattr(v, "shape") else torch.tensor(v)
        if t.dtype == torch.float32 and "running" not in k and requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def train_step(P, x, target):
    """forward + MSE(sum)/N (models/regression_lightning.py:57-65) + backward."""
    for p in P.values():
        if p.grad is not None:
            p.grad = None
    logits = forward(P, x)
    loss = F.mse_loss(logits.squeeze(1), target, reduction="sum") / target.shape[0]
    loss.backward()
    return loss.detach(), logits.detach()
