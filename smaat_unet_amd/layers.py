"""Drop-in module classes for /root/reference/models/layers.py (the four classes on the
hot path): same constructor signatures, attribute names and state_dict keys; `forward`
dispatches to the gfx950 kernels (smaat_unet_amd.ops).  No CPU path."""
from __future__ import annotations

import contextlib
import threading

import torch
from torch import nn

from . import ops

_TLS = threading.local()


@contextlib.contextmanager
def batched_counters():
    """Inside this context the `num_batches_tracked += 1` of every train-mode BatchNorm that runs (reference:
    torch.nn.BatchNorm2d.forward) is deferred and applied on exit with ONE multi-tensor add instead of one tiny
    kernel per layer (23 per SmaAt_UNet step).  The network's forward wraps itself in it; a BatchNorm with
    momentum=None needs the counter's value and increments immediately."""
    if getattr(_TLS, "pending", None) is not None:  # nested: the outermost context flushes
        yield
        return
    _TLS.pending = []
    try:
        yield
    finally:
        pend, _TLS.pending = _TLS.pending, None
        if pend:
            # a BatchNorm module called k times inside the context appears k times: add k ONCE per tensor (duplicate
            # tensors in one multi-tensor apply are not guaranteed to accumulate)
            counts = {}
            for t in pend:
                e = counts.setdefault(id(t), [t, 0])
                e[1] += 1
            with torch.no_grad():
                for k in sorted({n for _, n in counts.values()}):
                    torch._foreach_add_([t for t, n in counts.values() if n == k], k)


def _bn_args(bn: nn.BatchNorm2d):
    """(gamma, beta, running_mean, running_var, training, momentum, eps) with torch's
    exponential_average_factor rule (momentum=None -> cumulative average)."""
    momentum = bn.momentum
    if bn.training and bn.track_running_stats:
        if bn.num_batches_tracked is not None:
            pend = getattr(_TLS, "pending", None)
            if pend is not None and bn.momentum is not None:
                pend.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
            if bn.momentum is None:
                momentum = 1.0 / float(bn.num_batches_tracked)
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    return bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps


class DepthwiseSeparableConv(nn.Module):
    """reference: models/layers.py:34-50.  The 3x3 / padding 1 / kernels_per_layer in {1, 2, 4} configuration every network
    of the reference uses runs on the fused / row-streaming kernels; any other kernel_size, padding or
    kernels_per_layer the reference's constructor accepts runs on the general depthwise kernels
    (`smaat_dwconv_*_any`, f32) followed by the pointwise GEMM."""

    def __init__(self, in_channels, output_channels, kernel_size, padding=0, kernels_per_layer=1):
        super().__init__()
        self.depthwise = nn.Conv2d(
            in_channels,
            in_channels * kernels_per_layer,
            kernel_size=kernel_size,
            padding=padding,
            groups=in_channels,
        )
        self.pointwise = nn.Conv2d(in_channels * kernels_per_layer, output_channels, kernel_size=1)
        self.kernels_per_layer_ = kernels_per_layer

    def _fast_geometry(self):
        """the configuration the fused kernels are built for (and the only one DoubleConvDS constructs)"""
        dw = self.depthwise
        return (dw.kernel_size == (3, 3) and dw.padding == (1, 1) and dw.stride == (1, 1) and dw.dilation == (1, 1)
                and self.kernels_per_layer_ in (1, 2, 4))

    def _check_geometry(self):
        if not self._fast_geometry():
            raise NotImplementedError("the fused DoubleConvDS kernels take depthwise 3x3, stride 1, padding 1, "
                                      "kernels_per_layer in {1, 2, 4}")

    def forward(self, x):
        dw = self.depthwise
        if self._fast_geometry():
            return ops.dsconv(x, dw.weight, dw.bias, self.pointwise.weight, self.pointwise.bias, self.kernels_per_layer_)
        if dw.stride != (1, 1) or dw.dilation != (1, 1) or not isinstance(dw.padding, tuple) or dw.padding_mode != "zeros":
            raise NotImplementedError("depthwise stride / dilation / string padding modes: not constructible through "
                                      "DepthwiseSeparableConv.__init__ and not built")
        y = ops.depthwise_any(x, dw.weight, dw.bias, self.kernels_per_layer_, dw.padding[0], dw.padding[1])
        return ops.pointwise(y, self.pointwise.weight, self.pointwise.bias)


class Flatten(nn.Module):
    """reference: models/layers.py:85-87 (kept so that MLP indices 1 and 3 match)."""

    def forward(self, x):
        return x.view(x.size(0), -1)


class ChannelAttention(nn.Module):
    """reference: models/layers.py:90-111."""

    def __init__(self, input_channels, reduction_ratio=16):
        super().__init__()
        self.input_channels = input_channels
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.MLP = nn.Sequential(
            Flatten(),
            nn.Linear(input_channels, input_channels // reduction_ratio),
            nn.ReLU(),
            nn.Linear(input_channels // reduction_ratio, input_channels),
        )

    def _mlp_params(self):
        return self.MLP[1].weight, self.MLP[1].bias, self.MLP[3].weight, self.MLP[3].bias

    def forward(self, x):
        w1, b1, w2, b2 = self._mlp_params()
        return ops.cbam(x, w1, b1, w2, b2, None, None, None, None, None, False, None, 0.0, True, False)


class SpatialAttention(nn.Module):
    """reference: models/layers.py:114-129."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 7), "kernel size must be 3 or 7"
        padding = 3 if kernel_size == 7 else 1
        self.conv = nn.Conv2d(2, 1, kernel_size=kernel_size, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(1)

    def forward(self, x):
        g, b, rm, rv, training, momentum, eps = _bn_args(self.bn)
        return ops.cbam(x, None, None, None, None, self.conv.weight, g, b, rm, rv, training, momentum, eps, False,
                        True)


class CBAM(nn.Module):
    """reference: models/layers.py:132-141; both halves run as one fused operator."""

    def __init__(self, input_channels, reduction_ratio=16, kernel_size=7):
        super().__init__()
        self.channel_att = ChannelAttention(input_channels, reduction_ratio=reduction_ratio)
        self.spatial_att = SpatialAttention(kernel_size=kernel_size)

    EVAL_FAST_PATH = True

    def _eval_fast(self):
        bn = self.spatial_att.bn
        return (self.EVAL_FAST_PATH and not self.training and not bn.training and not torch.is_grad_enabled()
                and bn.track_running_stats and bn.running_mean is not None)

    @staticmethod
    def _split_lazy(x):
        """x is a tensor, or (z, st) from a block whose last BatchNorm + ReLU is deferred to this consumer (st rows:
        mean, invstd, scale, shift)"""
        if isinstance(x, tuple):
            return x[0], (x[1][2], x[1][3])
        return x, None

    def forward(self, x):
        x, lazy = self._split_lazy(x)
        w1, b1, w2, b2 = self.channel_att._mlp_params()
        sp = self.spatial_att
        from . import train_ops
        if lazy is None and train_ops.active() and torch.is_grad_enabled():
            g, b, rm, rv, training, momentum, eps = _bn_args(sp.bn)
            return train_ops.cbam(x, w1, b1, w2, b2, sp.conv.weight, g, b, rm, rv, training, momentum, eps)
        if lazy is not None:
            g, b, rm, rv, training, momentum, eps = _bn_args(sp.bn)
            return ops.cbam(x, w1, b1, w2, b2, sp.conv.weight, g, b, rm, rv, training, momentum, eps, True, True,
                            lazy=lazy)
        if self._eval_fast() and x.dtype == torch.float32:  # inference: three launches, BatchNorm(1) on the running
            # statistics (the inference operator set is f32: a bf16 activation takes the general operator below)

            return torch.ops.smaat.cbam_infer(x, w1, b1, w2, b2, sp.conv.weight, sp.bn.weight, sp.bn.bias,
                                              sp.bn.running_mean, sp.bn.running_var, sp.bn.eps)
        g, b, rm, rv, training, momentum, eps = _bn_args(sp.bn)
        return ops.cbam(x, w1, b1, w2, b2, sp.conv.weight, g, b, rm, rv, training, momentum, eps, True, True)

    def forward_pool_cat_forked(self, x, c_extra, side):
        """forward_pool_cat for the captured inference graph at small batch: (cat, pooled, keepalive) with the attention
        running on the stream `side` (ops.cbam_eval_forked), or None when that path does not apply"""
        if isinstance(x, tuple) or not self._eval_fast() or not x.is_cuda or x.dtype != torch.float32:
            return None
        w1, b1, w2, b2 = self.channel_att._mlp_params()
        sp = self.spatial_att
        n, c, h, w = x.shape
        cat = torch.empty((n, c + c_extra, h, w), dtype=x.dtype, device=x.device)
        r = ops.cbam_eval_forked(x, w1, b1, w2, b2, sp.conv.weight, sp.bn.weight, sp.bn.bias, sp.bn.running_mean,
                                 sp.bn.running_var, sp.bn.eps, cat[:, :c], side)
        if r is None:
            return None
        return cat, r[0], r[1] + (cat,)

    def forward_pool_cat(self, x, c_extra):
        """(cat, pooled): CBAM(x) written into channels [0, C) of a fresh [N, C + c_extra, H, W]
        concatenation buffer, and maxpool2(x) -- the two consumers of an encoder level in
        SmaAt_UNet.forward, sharing one backward pass over x."""
        x, lazy = self._split_lazy(x)
        w1, b1, w2, b2 = self.channel_att._mlp_params()
        sp = self.spatial_att
        from . import train_ops
        if lazy is None and train_ops.active() and torch.is_grad_enabled():
            g, b, rm, rv, training, momentum, eps = _bn_args(sp.bn)
            return train_ops.cbam_pool_cat(x, w1, b1, w2, b2, sp.conv.weight, g, b, rm, rv, training, momentum, eps, c_extra)
        if lazy is not None:
            g, b, rm, rv, training, momentum, eps = _bn_args(sp.bn)
            return ops.cbam_pool_cat(x, w1, b1, w2, b2, sp.conv.weight, g, b, rm, rv, training, momentum, eps, c_extra,
                                     lazy=lazy)
        if self._eval_fast() and x.dtype == torch.float32:

            return torch.ops.smaat.cbam_pool_cat_infer(x, w1, b1, w2, b2, sp.conv.weight, sp.bn.weight, sp.bn.bias,
                                                       sp.bn.running_mean, sp.bn.running_var, sp.bn.eps, c_extra)
        g, b, rm, rv, training, momentum, eps = _bn_args(sp.bn)
        return ops.cbam_pool_cat(x, w1, b1, w2, b2, sp.conv.weight, g, b, rm, rv, training, momentum, eps, c_extra)
