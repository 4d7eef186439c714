"""Drop-in for /root/reference/models/SmaAt_UNet.py: same constructor, attributes,
submodule names (=> identical 214 state_dict keys) and forward wiring (:41-57)."""
from __future__ import annotations

from torch import nn

from .layers import CBAM
from .unet_parts import OutConv
from .unet_parts_depthwise_separable import DoubleConvDS, DownDS, UpDS


class SmaAt_UNet(nn.Module):
    def __init__(
        self,
        n_channels,
        n_classes,
        kernels_per_layer=2,
        bilinear=True,
        reduction_ratio=16,
    ):
        super().__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear

        self.inc = DoubleConvDS(self.n_channels, 64, kernels_per_layer=kernels_per_layer)
        self.cbam1 = CBAM(64, reduction_ratio=reduction_ratio)
        self.down1 = DownDS(64, 128, kernels_per_layer=kernels_per_layer)
        self.cbam2 = CBAM(128, reduction_ratio=reduction_ratio)
        self.down2 = DownDS(128, 256, kernels_per_layer=kernels_per_layer)
        self.cbam3 = CBAM(256, reduction_ratio=reduction_ratio)
        self.down3 = DownDS(256, 512, kernels_per_layer=kernels_per_layer)
        self.cbam4 = CBAM(512, reduction_ratio=reduction_ratio)
        factor = 2 if self.bilinear else 1
        self.down4 = DownDS(512, 1024 // factor, kernels_per_layer=kernels_per_layer)
        self.cbam5 = CBAM(1024 // factor, reduction_ratio=reduction_ratio)
        self.up1 = UpDS(1024, 512 // factor, self.bilinear, kernels_per_layer=kernels_per_layer)
        self.up2 = UpDS(512, 256 // factor, self.bilinear, kernels_per_layer=kernels_per_layer)
        self.up3 = UpDS(256, 128 // factor, self.bilinear, kernels_per_layer=kernels_per_layer)
        self.up4 = UpDS(128, 64, self.bilinear, kernels_per_layer=kernels_per_layer)

        self.outc = OutConv(64, self.n_classes)

    def _fusable(self):
        """the fused skip wiring bypasses the `forward` of cbamN / downN.maxpool / upN: keep the
        module-by-module path whenever a user hooked one of them (or uses an exotic configuration)."""
        mods = [self.cbam1, self.cbam2, self.cbam3, self.cbam4, self.down1, self.down2, self.down3, self.down4,
                self.up1, self.up2, self.up3, self.up4]
        for top in mods:
            for mm in top.modules():
                if mm._forward_hooks or mm._forward_pre_hooks or mm._backward_hooks:
                    return False
        return self.bilinear

    def forward(self, x):
        # NB (reference :41-57): the encoder continues from the UN-attended x_i; the CBAM
        # outputs feed only the skip connections and the bottleneck.
        if not self._fusable():
            return self._forward_modular(x)
        ups = (self.up4, self.up3, self.up2, self.up1)
        cats = []
        h = self.inc(x)
        for cbam, down, up in zip((self.cbam1, self.cbam2, self.cbam3, self.cbam4),
                                  (self.down1, self.down2, self.down3, self.down4), ups):
            c_extra = up.conv.double_conv[0].depthwise.in_channels - h.shape[1]
            cat, pooled = cbam.forward_pool_cat(h, c_extra)   # skip written straight into the decoder's cat buffer
            cats.append(cat)
            h = down.maxpool_conv[1](pooled)
        h = self.cbam5(h)
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
        x5Att = self.cbam5(x5)
        x = self.up1(x5Att, x4Att)
        x = self.up2(x, x3Att)
        x = self.up3(x, x2Att)
        x = self.up4(x, x1Att)
        return self.outc(x)
