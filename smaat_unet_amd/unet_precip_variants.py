"""Sibling architectures of SmaAt-UNet built from the same MI355X blocks.

Drop-ins for the network definitions of /root/reference/models/unet_precip_regression_lightning.py:
  UNetDS                  (:86-118)   depthwise-separable U-Net without attention
  UNetDSAttention         (:121-164)  = the SmaAt-UNet wiring (CBAM on every level, bottleneck included)
  UNetDSAttention4CBAMs   (:167-208)  CBAM on the four skip connections only (un-attended bottleneck)
Same submodule names (=> the reference `state_dict` keys of each network, checked against the reference
blocks in oracle/gen_golden.py) and the same forward wiring.  The reference classes are Lightning modules
configured through an `hparams` namespace; these take either that namespace / a dict (`hparams=...`) or
plain keyword arguments -- the training loop around them (Lightning) is out of scope.
"""
from __future__ import annotations

from torch import nn

from .SmaAt_UNet import SmaAt_UNet
from .layers import CBAM
from .unet_parts import OutConv
from .unet_parts_depthwise_separable import DoubleConvDS, DownDS, UpDS

_DEFAULTS = dict(n_channels=12, n_classes=1, kernels_per_layer=2, bilinear=True, reduction_ratio=16)


def _hp(hparams, kw):
    cfg = dict(_DEFAULTS)
    if hparams is not None:
        src = hparams if isinstance(hparams, dict) else vars(hparams)
        cfg.update({k: src[k] for k in _DEFAULTS if k in src})
    unknown = set(kw) - set(_DEFAULTS)
    if unknown:
        raise TypeError(f"unexpected arguments {sorted(unknown)}")
    cfg.update(kw)
    return cfg


class UNetDS(nn.Module):
    """reference: models/unet_precip_regression_lightning.py:86-118"""

    def __init__(self, hparams=None, **kw):
        super().__init__()
        c = _hp(hparams, kw)
        self.n_channels, self.n_classes, self.bilinear = c["n_channels"], c["n_classes"], c["bilinear"]
        kpl = c["kernels_per_layer"]
        self.inc = DoubleConvDS(self.n_channels, 64, kernels_per_layer=kpl)
        self.down1 = DownDS(64, 128, kernels_per_layer=kpl)
        self.down2 = DownDS(128, 256, kernels_per_layer=kpl)
        self.down3 = DownDS(256, 512, kernels_per_layer=kpl)
        factor = 2 if self.bilinear else 1
        self.down4 = DownDS(512, 1024 // factor, kernels_per_layer=kpl)
        self.up1 = UpDS(1024, 512 // factor, self.bilinear, kernels_per_layer=kpl)
        self.up2 = UpDS(512, 256 // factor, self.bilinear, kernels_per_layer=kpl)
        self.up3 = UpDS(256, 128 // factor, self.bilinear, kernels_per_layer=kpl)
        self.up4 = UpDS(128, 64, self.bilinear, kernels_per_layer=kpl)
        self.outc = OutConv(64, self.n_classes)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up1(x5, x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        x = self.up4(x, x1)
        return self.outc(x)


class UNetDSAttention(SmaAt_UNet):
    """reference: models/unet_precip_regression_lightning.py:121-164 (the SmaAt-UNet wiring)"""

    def __init__(self, hparams=None, **kw):
        c = _hp(hparams, kw)
        super().__init__(c["n_channels"], c["n_classes"], kernels_per_layer=c["kernels_per_layer"],
                         bilinear=c["bilinear"], reduction_ratio=c["reduction_ratio"])


class UNetDSAttention4CBAMs(nn.Module):
    """reference: models/unet_precip_regression_lightning.py:167-208"""

    def __init__(self, hparams=None, **kw):
        super().__init__()
        c = _hp(hparams, kw)
        self.n_channels, self.n_classes, self.bilinear = c["n_channels"], c["n_classes"], c["bilinear"]
        kpl, rr = c["kernels_per_layer"], c["reduction_ratio"]
        self.inc = DoubleConvDS(self.n_channels, 64, kernels_per_layer=kpl)
        self.cbam1 = CBAM(64, reduction_ratio=rr)
        self.down1 = DownDS(64, 128, kernels_per_layer=kpl)
        self.cbam2 = CBAM(128, reduction_ratio=rr)
        self.down2 = DownDS(128, 256, kernels_per_layer=kpl)
        self.cbam3 = CBAM(256, reduction_ratio=rr)
        self.down3 = DownDS(256, 512, kernels_per_layer=kpl)
        self.cbam4 = CBAM(512, reduction_ratio=rr)
        factor = 2 if self.bilinear else 1
        self.down4 = DownDS(512, 1024 // factor, kernels_per_layer=kpl)
        self.up1 = UpDS(1024, 512 // factor, self.bilinear, kernels_per_layer=kpl)
        self.up2 = UpDS(512, 256 // factor, self.bilinear, kernels_per_layer=kpl)
        self.up3 = UpDS(256, 128 // factor, self.bilinear, kernels_per_layer=kpl)
        self.up4 = UpDS(128, 64, self.bilinear, kernels_per_layer=kpl)
        self.outc = OutConv(64, self.n_classes)

    def forward(self, x):
        if not SmaAt_UNet._fusable(self):
            return self._forward_modular(x)
        # fused skip wiring of SmaAt_UNet.forward (CBAM output written straight into the decoder's
        # concatenation buffer, one dX for CBAM + max-pool), without the bottleneck attention
        ups = (self.up4, self.up3, self.up2, self.up1)
        cats = []
        h = self.inc(x)
        for cbam, down, up in zip((self.cbam1, self.cbam2, self.cbam3, self.cbam4),
                                  (self.down1, self.down2, self.down3, self.down4), ups):
            c_extra = up.conv.double_conv[0].depthwise.in_channels - h.shape[1]
            cat, pooled = cbam.forward_pool_cat(h, c_extra)
            cats.append(cat)
            h = down.maxpool_conv[1](pooled)
        for up, cat in zip(reversed(ups), reversed(cats)):
            h = up.forward_into(h, cat)
        return self.outc(h)

    def _forward_modular(self, x):
        x1 = self.inc(x)
        x1Att = self.cbam1(x1)
        x2 = self.down1(x1)
        x2Att = self.cbam2(x2)
        x3 = self.down2(x2)
        x3Att = self.cbam3(x3)
        x4 = self.down3(x3)
        x4Att = self.cbam4(x4)
        x5 = self.down4(x4)
        x = self.up1(x5, x4Att)
        x = self.up2(x, x3Att)
        x = self.up3(x, x2Att)
        x = self.up4(x, x1Att)
        return self.outc(x)
