"""Drop-in module classes for /root/reference/models/unet_parts_depthwise_separable.py."""
from __future__ import annotations

from torch import nn

from . import ops
from .layers import DepthwiseSeparableConv, _bn_args


class DoubleConvDS(nn.Module):
    """(DepthwiseSeparableConv => BN => ReLU) * 2 -- reference :10-39.  Each half runs as the
    fused dsconv+BN-stats kernel, a finalize and one BN-apply+ReLU pass."""

    def __init__(self, in_channels, out_channels, mid_channels=None, kernels_per_layer=1):
        super().__init__()
        widths = (in_channels, mid_channels or out_channels, out_channels)
        stages = []
        for cin, cout in zip(widths[:-1], widths[1:]):  # Sequential indices 0,1,2 / 3,4,5 as in the reference
            stages += [DepthwiseSeparableConv(cin, cout, kernel_size=3, padding=1, kernels_per_layer=kernels_per_layer),
                       nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
        self.double_conv = nn.Sequential(*stages)

    @staticmethod
    def _half(x, conv: DepthwiseSeparableConv, bn: nn.BatchNorm2d):
        g, b, rm, rv, training, momentum, eps = _bn_args(bn)
        if not conv._fast_geometry():
            # a kernels_per_layer (or, for a hand-built block, a kernel size / padding) outside the fused configuration:
            # general depthwise kernels, then pointwise GEMM + BatchNorm + ReLU as one node
            dw = conv.depthwise
            y = ops.depthwise_any(x, dw.weight, dw.bias, conv.kernels_per_layer_, dw.padding[0], dw.padding[1])
            return ops.pointwise_bn_relu(y, conv.pointwise.weight, conv.pointwise.bias, g, b, rm, rv, training, momentum, eps)
        return ops.dsconv_bn_relu(x, conv.depthwise.weight, conv.depthwise.bias, conv.pointwise.weight,
                                  conv.pointwise.bias, g, b, rm, rv, training, momentum, eps,
                                  conv.kernels_per_layer_)

    # ---- inference fast path: BatchNorm folded into the pointwise weights, cached until a tensor changes ----
    EVAL_FAST_PATH = True

    def _folded_half(self, i):
        conv, bn = self.double_conv[3 * i], self.double_conv[3 * i + 1]
        src = (conv.pointwise.weight, conv.pointwise.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        try:
            # (data pointer, version counter) of every tensor that enters the fold + the matrix mode the operand images
            # were built for.  Writes through `.data` do not bump the version counters: after such updates call the
            # network's invalidate_eval_cache() (or DoubleConvDS.invalidate_eval_cache()).
            key = (tuple((t.data_ptr(), t._version) if t is not None else None for t in src)
                   + (bn.eps, ops._lib.get().smaat_split_mode()))
        except Exception:  # noqa: BLE001  (FakeTensors under torch.export / compile: fold inside the traced graph)
            return conv.depthwise.weight, conv.depthwise.bias, ops.fold_bn_into_pointwise(*src, bn.eps)
        cache = self.__dict__.setdefault("_fold_cache", {})
        if cache.get(i, (None,))[0] != key:
            cache[i] = (key, ops.fold_bn_into_pointwise(*src, bn.eps))
        return conv.depthwise.weight, conv.depthwise.bias, cache[i][1]

    def invalidate_eval_cache(self):
        """forget the BatchNorm-folded weights (see UNetDSFamily.invalidate_eval_cache)"""
        self.__dict__["_fold_cache"] = {}
        return self

    def _eval_fast_ok(self, hooked, x):
        import torch
        seq = self.double_conv
        # (the inference operator set is f32; a bf16 activation -- mixed precision with a hooked block upstream -- takes
        # the general operators below, which dispatch on the stored type)
        return (self.EVAL_FAST_PATH and not hooked and not self.training and not torch.is_grad_enabled()
                and x.dtype == torch.float32
                and all(seq[j].track_running_stats and seq[j].running_mean is not None and not seq[j].training
                        for j in (1, 4))
                and seq[0].kernels_per_layer_ == seq[3].kernels_per_layer_)

    def forward(self, x, head=None, defer=False):
        """head (internal, set by the network's fused wiring): the nn.Conv2d(C, 1, 1) of the OutConv that consumes this
        block; the call then returns OutConv(block(x)) as one autograd node (ops.double_conv_ds(..., head=...)).
        defer (internal, same wiring): return (z2, st2) -- the block output with its last BatchNorm + ReLU (coefficient
        rows st2[2], st2[3]) left to the consumer -- when the block runs as the fused training node; a plain tensor otherwise."""
        seq = self.double_conv
        hooked = any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in seq.modules())
        import torch
        from . import train_ops
        if not (seq[0]._fast_geometry() and seq[3]._fast_geometry()):
            y = self._half(self._half(x, seq[0], seq[1]), seq[3], seq[4])  # half by half on the general kernels
            if head is None:
                return y
            return ops.pointwise(y, head.weight, head.bias)
        if (train_ops.active() and torch.is_grad_enabled() and not hooked
                and seq[0].kernels_per_layer_ == seq[3].kernels_per_layer_):
            # traceable wiring: the block as the custom operator smaat::double_conv_ds (no head / deferred-activation fusion)
            for conv in (seq[0], seq[3]):
                conv._check_geometry()
            halves = [(seq[i].depthwise.weight, seq[i].depthwise.bias, seq[i].pointwise.weight, seq[i].pointwise.bias)
                      + _bn_args(seq[i + 1]) for i in (0, 3)]
            y = train_ops.double_conv_ds(x, halves[0], halves[1], seq[0].kernels_per_layer_)
            return y if head is None else train_ops.pointwise(y, head.weight, head.bias)
        if defer:
            import torch
            if (not hooked and torch.is_grad_enabled() and seq[0].kernels_per_layer_ == seq[3].kernels_per_layer_
                    and seq[4].affine):
                for conv in (seq[0], seq[3]):
                    conv._check_geometry()
                halves = [(seq[i].depthwise.weight, seq[i].depthwise.bias, seq[i].pointwise.weight,
                           seq[i].pointwise.bias) + _bn_args(seq[i + 1]) for i in (0, 3)]
                return ops.double_conv_ds(x, halves[0], halves[1], seq[0].kernels_per_layer_, defer=True)
        if head is not None:
            import torch
            if hooked or not torch.is_grad_enabled() or seq[0].kernels_per_layer_ != seq[3].kernels_per_layer_:
                y = self.forward(x)  # the plain block, then the OutConv on its own
                if not torch.is_grad_enabled() and y.dtype == torch.float32:
                    return torch.ops.smaat.pointwise_infer(y, head.weight, head.bias)
                return ops.pointwise(y, head.weight, head.bias)
            for conv in (seq[0], seq[3]):
                conv._check_geometry()
            halves = [(seq[i].depthwise.weight, seq[i].depthwise.bias, seq[i].pointwise.weight, seq[i].pointwise.bias)
                      + _bn_args(seq[i + 1]) for i in (0, 3)]
            return ops.double_conv_ds(x, halves[0], halves[1], seq[0].kernels_per_layer_, head=(head.weight, head.bias))
        if self._eval_fast_ok(hooked, x):
            for conv in (seq[0], seq[3]):
                conv._check_geometry()
            return ops.double_conv_ds_eval(x, self._folded_half(0), self._folded_half(1), seq[0].kernels_per_layer_)
        if hooked or seq[0].kernels_per_layer_ != seq[3].kernels_per_layer_:
            x = self._half(x, seq[0], seq[1])
            return self._half(x, seq[3], seq[4])
        halves = []
        for conv, bn in ((seq[0], seq[1]), (seq[3], seq[4])):
            conv._check_geometry()
            halves.append((conv.depthwise.weight, conv.depthwise.bias, conv.pointwise.weight, conv.pointwise.bias)
                          + _bn_args(bn))
        return ops.double_conv_ds(x, halves[0], halves[1], seq[0].kernels_per_layer_)


class _MaxPool2(nn.MaxPool2d):
    def forward(self, x):
        import torch
        if not torch.is_grad_enabled() and x.dtype == torch.float32:  # (the inference operator set is f32)
            return torch.ops.smaat.maxpool2_infer(x)
        return ops.maxpool2(x)


class DownDS(nn.Module):
    """Downscaling with maxpool then double conv -- reference :42-53."""

    def __init__(self, in_channels, out_channels, kernels_per_layer=1):
        super().__init__()
        conv = DoubleConvDS(in_channels, out_channels, kernels_per_layer=kernels_per_layer)
        self.maxpool_conv = nn.Sequential(_MaxPool2(2), conv)

    def forward(self, x):
        return self.maxpool_conv(x)


class UpDS(nn.Module):
    """Upscaling then double conv -- reference :56-86.  bilinear=True: nn.Upsample(x2, bilinear, align_corners) and a
    DoubleConvDS with mid = in/2 (:64-70); bilinear=False: nn.ConvTranspose2d(in, in/2, 2, stride=2) and a plain
    DoubleConvDS (:72-73).  Either way the upsampled map is written straight into the concatenation buffer."""

    def __init__(self, in_channels, out_channels, bilinear=True, kernels_per_layer=1):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:
            # `up` is kept as a (parameter-free) submodule for interface parity; the upsampling itself runs fused
            # with the pad + concatenation in ops.upsample_cat / ops.upsample_into
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConvDS(in_channels, out_channels, mid_channels=in_channels // 2,
                                     kernels_per_layer=kernels_per_layer)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)  # parameters only
            self.conv = DoubleConvDS(in_channels, out_channels, kernels_per_layer=kernels_per_layer)

    def forward(self, x1, x2):
        import torch
        if self.bilinear:
            if not torch.is_grad_enabled() and x1.dtype == torch.float32 and x2.dtype == torch.float32:
                return self.conv(torch.ops.smaat.upsample_cat_infer(x1, x2))
            return self.conv(ops.upsample_cat(x1, x2))
        return self.conv(ops.upconv_cat(x1, x2, self.up.weight, self.up.bias))

    def forward_into(self, x1, cat, head=None):
        """same as forward(x1, x2) when x2 already sits in channels [0, C2) of `cat`
        ([N, C2 + C1', H2, W2]): the upsampled x1 is written behind it, no torch.cat copy.
        head: see DoubleConvDS.forward."""
        kw = {} if head is None else {"head": head}
        if self.bilinear:
            import torch
            from . import train_ops
            if train_ops.active() and torch.is_grad_enabled():
                return self.conv(train_ops.upsample_into(cat, x1, cat.shape[1] - x1.shape[1]), **kw)
            if not torch.is_grad_enabled() and x1.dtype == torch.float32:
                torch.ops.smaat.upsample_into_(cat, x1, cat.shape[1] - x1.shape[1])
                return self.conv(cat, **kw)
            return self.conv(ops.upsample_into(cat, x1, cat.shape[1] - x1.shape[1]), **kw)
        co = self.up.out_channels
        return self.conv(ops.upconv_into(cat, x1, self.up.weight, self.up.bias, cat.shape[1] - co), **kw)


class OutConv(nn.Module):
    """reference :89-95 (duplicate of models/unet_parts.py:67-73)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        import torch
        from . import train_ops
        if train_ops.active() and torch.is_grad_enabled():
            return train_ops.pointwise(x, self.conv.weight, self.conv.bias)
        if not torch.is_grad_enabled() and x.dtype == torch.float32:
            return torch.ops.smaat.pointwise_infer(x, self.conv.weight, self.conv.bias)
        return ops.pointwise(x, self.conv.weight, self.conv.bias)
