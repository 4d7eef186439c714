"""Deterministic numpy parameter generator keyed like the reference state_dict.

TEST INFRASTRUCTURE ONLY (see oracle/smaat_oracle.py header).

The key order/shapes restate what SmaAt_UNet.__init__ builds
(/root/reference/models/SmaAt_UNet.py:8-39 with the block definitions in
models/unet_parts_depthwise_separable.py:13-36,45-50,59-70, models/layers.py:35-45,
91-103,115-120,133-136 and models/unet_parts.py:68-70).  oracle/gen_golden.py
asserts the list equals the real reference's state_dict keys and shapes.

A 16 MB state_dict is too big for a fixture, so goldens are produced from these
seeded numpy parameters instead; the values are drawn with torch-default-like
bounds (U(+-1/sqrt(fan_in))) but BN affine parameters are randomised so that
gamma/beta paths are exercised.
"""
from __future__ import annotations

import numpy as np


def _dsconv(keys, pre, cin, cout, kpl):
    keys.append((pre + ".depthwise.weight", (cin * kpl, 1, 3, 3)))
    keys.append((pre + ".depthwise.bias", (cin * kpl,)))
    keys.append((pre + ".pointwise.weight", (cout, cin * kpl, 1, 1)))
    keys.append((pre + ".pointwise.bias", (cout,)))


def _bn(keys, pre, c):
    keys.append((pre + ".weight", (c,)))
    keys.append((pre + ".bias", (c,)))
    keys.append((pre + ".running_mean", (c,)))
    keys.append((pre + ".running_var", (c,)))
    keys.append((pre + ".num_batches_tracked", ()))


def double_conv_ds_keys(keys, pre, cin, cout, cmid, kpl):
    cmid = cmid or cout
    _dsconv(keys, pre + ".double_conv.0", cin, cmid, kpl)
    _bn(keys, pre + ".double_conv.1", cmid)
    _dsconv(keys, pre + ".double_conv.3", cmid, cout, kpl)
    _bn(keys, pre + ".double_conv.4", cout)


def cbam_keys(keys, pre, c, rr):
    keys.append((pre + ".channel_att.MLP.1.weight", (c // rr, c)))
    keys.append((pre + ".channel_att.MLP.1.bias", (c // rr,)))
    keys.append((pre + ".channel_att.MLP.3.weight", (c, c // rr)))
    keys.append((pre + ".channel_att.MLP.3.bias", (c,)))
    keys.append((pre + ".spatial_att.conv.weight", (1, 2, 7, 7)))
    _bn(keys, pre + ".spatial_att.bn", 1)


def _up_keys(k, pre, cin, cout, kpl, bilinear):
    """UpDS (models/unet_parts_depthwise_separable.py:59-73): bilinear -> DoubleConvDS(cin, cout, cin // 2);
    else ConvTranspose2d(cin, cin // 2, 2, stride=2) (weight [cin][cin // 2][2][2]) + DoubleConvDS(cin, cout)"""
    if bilinear:
        double_conv_ds_keys(k, pre + ".conv", cin, cout, cin // 2, kpl)
    else:
        k.append((pre + ".up.weight", (cin, cin // 2, 2, 2)))
        k.append((pre + ".up.bias", (cin // 2,)))
        double_conv_ds_keys(k, pre + ".conv", cin, cout, None, kpl)


def smaat_unet_keys(n_channels, n_classes, kpl=2, rr=16, bilinear=True):
    """[(name, shape)] in reference state_dict order (models/SmaAt_UNet.py:23-39)."""
    if not bilinear:
        k = []
        double_conv_ds_keys(k, "inc", n_channels, 64, None, kpl)
        chans = (64, 128, 256, 512, 1024)
        for lvl in range(1, 5):
            cbam_keys(k, f"cbam{lvl}", chans[lvl - 1], rr)
            double_conv_ds_keys(k, f"down{lvl}.maxpool_conv.1", chans[lvl - 1], chans[lvl], None, kpl)
        cbam_keys(k, "cbam5", 1024, rr)
        _up_keys(k, "up1", 1024, 512, kpl, False)
        _up_keys(k, "up2", 512, 256, kpl, False)
        _up_keys(k, "up3", 256, 128, kpl, False)
        _up_keys(k, "up4", 128, 64, kpl, False)
        k.append(("outc.conv.weight", (n_classes, 64, 1, 1)))
        k.append(("outc.conv.bias", (n_classes,)))
        return k
    k = []
    double_conv_ds_keys(k, "inc", n_channels, 64, None, kpl)
    cbam_keys(k, "cbam1", 64, rr)
    double_conv_ds_keys(k, "down1.maxpool_conv.1", 64, 128, None, kpl)
    cbam_keys(k, "cbam2", 128, rr)
    double_conv_ds_keys(k, "down2.maxpool_conv.1", 128, 256, None, kpl)
    cbam_keys(k, "cbam3", 256, rr)
    double_conv_ds_keys(k, "down3.maxpool_conv.1", 256, 512, None, kpl)
    cbam_keys(k, "cbam4", 512, rr)
    double_conv_ds_keys(k, "down4.maxpool_conv.1", 512, 512, None, kpl)
    cbam_keys(k, "cbam5", 512, rr)
    double_conv_ds_keys(k, "up1.conv", 1024, 256, 512, kpl)
    double_conv_ds_keys(k, "up2.conv", 512, 128, 256, kpl)
    double_conv_ds_keys(k, "up3.conv", 256, 64, 128, kpl)
    double_conv_ds_keys(k, "up4.conv", 128, 64, 64, kpl)
    k.append(("outc.conv.weight", (n_classes, 64, 1, 1)))
    k.append(("outc.conv.bias", (n_classes,)))
    return k


def unetds_keys(n_channels, n_classes, kpl=2, rr=16, cbams=0):
    """[(name, shape)] of the sibling networks of models/unet_precip_regression_lightning.py (bilinear=True):
    cbams = 0 -> UNetDS (:86-118), 4 -> UNetDSAttention4CBAMs (:167-208), 5 -> UNetDSAttention (:121-164)."""
    k = []
    double_conv_ds_keys(k, "inc", n_channels, 64, None, kpl)
    chans = (64, 128, 256, 512, 512)
    for lvl in range(1, 5):
        if cbams >= lvl:
            cbam_keys(k, f"cbam{lvl}", chans[lvl - 1], rr)
        double_conv_ds_keys(k, f"down{lvl}.maxpool_conv.1", chans[lvl - 1], chans[lvl], None, kpl)
    if cbams >= 5:
        cbam_keys(k, "cbam5", 512, rr)
    double_conv_ds_keys(k, "up1.conv", 1024, 256, 512, kpl)
    double_conv_ds_keys(k, "up2.conv", 512, 128, 256, kpl)
    double_conv_ds_keys(k, "up3.conv", 256, 64, 128, kpl)
    double_conv_ds_keys(k, "up4.conv", 128, 64, 64, kpl)
    k.append(("outc.conv.weight", (n_classes, 64, 1, 1)))
    k.append(("outc.conv.bias", (n_classes,)))
    return k


def fill(keys, seed=0):
    """name -> np.ndarray (float32; int64 for num_batches_tracked)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in keys:
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros((), np.int64)
        elif name.endswith("running_mean"):
            out[name] = np.zeros(shape, np.float32)
        elif name.endswith("running_var"):
            out[name] = np.ones(shape, np.float32)
        elif ".double_conv.1." in name or ".double_conv.4." in name or ".bn." in name:
            if name.endswith(".weight"):
                out[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            else:
                out[name] = rng.uniform(-0.2, 0.2, shape).astype(np.float32)
        else:
            if len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
            elif len(shape) == 2:
                fan_in = shape[1]
            else:
                fan_in = None
            if fan_in is None:  # bias: reuse the bound of the layer's weight (previous entry)
                fan_in = out["__last_fan_in"]
            else:
                out["__last_fan_in"] = fan_in
            bound = 1.0 / np.sqrt(fan_in)
            out[name] = rng.uniform(-bound, bound, shape).astype(np.float32)
    out.pop("__last_fan_in", None)
    return out


def make_smaat_params(n_channels=12, n_classes=1, kpl=2, rr=16, seed=0):
    return fill(smaat_unet_keys(n_channels, n_classes, kpl, rr), seed)


def synthetic_case(kind, n, c, h, w, n_classes, seed):
    """Deterministic numpy inputs for the large golden cases (regenerated from the seed on both sides, so the
    fixtures hold summaries only).  kind "precip": SURVEY 8(d) sparse radar frames + a dense target in [0, 0.3]
    (reference loss models/regression_lightning.py:57-65); kind "voc": ImageNet-normalised-like images ~ N(0, 1)
    and integer class maps (reference train_SmaAtUNet.py:178-183, utils/dataset_VOC.py:134-137)."""
    rng = np.random.default_rng(seed)
    if kind == "precip":
        u = rng.random((n, c, h, w), dtype=np.float32)
        x = np.where(u > 0.7, (u - 0.7) / 0.3 * 0.5, 0.0).astype(np.float32)
        y = (rng.random((n, h, w), dtype=np.float32) * 0.3).astype(np.float32)
        return x, y
    if kind == "voc":
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        y = rng.integers(0, n_classes, (n, h, w)).astype(np.int64)
        return x, y
    raise ValueError(kind)
