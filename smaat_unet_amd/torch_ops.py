"""`torch.ops.smaat.*`: the inference operator set as PyTorch custom operators (torch.library.custom_op) with
fake (meta) kernels, so that an eval-mode `SmaAt_UNet` traces under `torch.export` / `torch.compile` / `make_fx` into a
graph of ~45 opaque gfx950 kernels-behind-operators instead of breaking on the ctypes calls.

The modules' inference fast path (eval mode under `torch.no_grad()`, reference call stack D:
calc_metrics_test_set.py:119) calls these operators; training goes through the `torch.autograd.Function`s of
`smaat_unet_amd.ops` (whose forward/backward pairs keep un-materialised activations and fused reductions between
them, which an operator-per-tensor decomposition would have to give up).  The four differentiable functional
operators registered in ops.py (`smaat::dsconv`, `pointwise`, `maxpool2`, `upsample_cat`) remain.

  smaat::split_planes          f32 weight matrix -> three bf16 planes (chunk-major), models/layers.py:45 weights
  smaat::dsconv_folded         DepthwiseSeparableConv + eval BatchNorm2d (folded) + ReLU   (layers.py:47-50,
                               unet_parts_depthwise_separable.py:24-26)
  smaat::cbam_infer            CBAM(x)                                                       (layers.py:132-141)
  smaat::cbam_pool_cat_infer   (cat buffer with CBAM(x) in its first C channels, maxpool2(x))  (SmaAt_UNet.py:43-50)
  smaat::upsample_into_        cat[:, c_off:] = pad(upsample2x(x1)), in place   (unet_parts_depthwise_separable.py:64,76-85)
  smaat::upsample_cat_infer    cat([x2, pad(upsample2x(x1))])
  smaat::pointwise_infer       OutConv                                                       (unet_parts.py:67-73)
  smaat::maxpool2_infer        MaxPool2d(2)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import _lib, ops


@custom_op("smaat::split_planes", mutates_args=())
def split_planes(w2d: Tensor) -> Tensor:
    return ops._split_planes_raw(w2d.contiguous())


@split_planes.register_fake
def _(w2d):
    r, c = w2d.shape
    return w2d.new_empty((3, r, (c + 15) // 16 * 16), dtype=torch.int16)


@custom_op("smaat::dsconv_folded", mutates_args=())
def dsconv_folded(x: Tensor, w_dw: Tensor, b_dw: Optional[Tensor], w_fold: Tensor, wt_fold: Tensor,
                  planes: Optional[Tensor], b_fold: Tensor, kpl: int, relu: bool) -> Tensor:
    return ops.dsconv_folded(x, w_dw, b_dw, dict(w=w_fold, wt=wt_fold, planes=planes, b=b_fold), kpl, relu_out=relu)


@dsconv_folded.register_fake
def _(x, w_dw, b_dw, w_fold, wt_fold, planes, b_fold, kpl, relu):
    n, _, h, w = x.shape
    return x.new_empty((n, w_fold.shape[0], h, w))


@custom_op("smaat::cbam_infer", mutates_args=())
def cbam_infer(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, wconv: Tensor, gamma: Optional[Tensor],
               beta: Optional[Tensor], rm: Tensor, rv: Tensor, eps: float) -> Tensor:
    return ops.cbam_eval(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, eps)


@cbam_infer.register_fake
def _(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, eps):
    return x.new_empty(x.shape)


@custom_op("smaat::cbam_pool_cat_infer", mutates_args=())
def cbam_pool_cat_infer(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, wconv: Tensor,
                        gamma: Optional[Tensor], beta: Optional[Tensor], rm: Tensor, rv: Tensor, eps: float,
                        c_extra: int) -> Tuple[Tensor, Tensor]:
    n, c, h, w = x.shape
    cat = torch.empty((n, c + c_extra, h, w), dtype=x.dtype, device=x.device)
    _, pooled = ops.cbam_eval(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, eps, out=cat[:, :c], pool=True)
    return cat, pooled


@cbam_pool_cat_infer.register_fake
def _(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, eps, c_extra):
    n, c, h, w = x.shape
    return x.new_empty((n, c + c_extra, h, w)), x.new_empty((n, c, h // 2, w // 2))


@custom_op("smaat::upsample_into_", mutates_args=("cat",))
def upsample_into_(cat: Tensor, x1: Tensor, c_off: int) -> None:
    with torch.no_grad():
        ops._UpsampleInto.apply(cat, x1, c_off, None)


@upsample_into_.register_fake
def _(cat, x1, c_off):
    return None


@custom_op("smaat::upsample_cat_infer", mutates_args=())
def upsample_cat_infer(x1: Tensor, x2: Tensor) -> Tensor:
    with torch.no_grad():
        return ops._UpsampleCat.apply(x1, x2)


@upsample_cat_infer.register_fake
def _(x1, x2):
    n, c2, ho, wo = x2.shape
    return x2.new_empty((n, c2 + x1.shape[1], ho, wo))


@custom_op("smaat::pointwise_infer", mutates_args=())
def pointwise_infer(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    with torch.no_grad():
        return ops._Pointwise.apply(x, w, b)


@pointwise_infer.register_fake
def _(x, w, b):
    n, _, h, wd = x.shape
    return x.new_empty((n, w.shape[0], h, wd))


@custom_op("smaat::maxpool2_infer", mutates_args=())
def maxpool2_infer(x: Tensor) -> Tensor:
    with torch.no_grad():
        return ops._MaxPool2.apply(x)


@maxpool2_infer.register_fake
def _(x):
    n, c, h, w = x.shape
    return x.new_empty((n, c, h // 2, w // 2))


def inference_mode_active(module) -> bool:
    """eval mode under no_grad: the modules switch to the operators above"""
    return (not module.training) and (not torch.is_grad_enabled())
