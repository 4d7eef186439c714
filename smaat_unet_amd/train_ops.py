"""`torch.ops.smaat.*` TRAINING operators: the fused autograd nodes of the training path as PyTorch custom operators
(torch.library.custom_op + register_fake + register_autograd), so that a whole TRAIN STEP -- forward, loss, backward --
of `SmaAt_UNet` traces under `make_fx` / `torch.compile` / `torch.export` into a graph of opaque gfx950
kernels-behind-operators (north_star: "exposed to the Python host through PyTorch-ROCm custom ops"; reference boundary
models/SmaAt_UNet.py:41-57).  Round 2 had only the inference set (torch_ops.py); the training path went through
`torch.autograd.Function`s whose ctypes calls a tracer cannot see through.

    smaat::double_conv_ds      (DepthwiseSeparableConv => BN => ReLU) * 2   unet_parts_depthwise_separable.py:10-39
    smaat::cbam_pool_cat       CBAM(x) into the decoder's concatenation buffer + MaxPool2d(2)(x)   SmaAt_UNet.py:43-50
    smaat::cbam_train          CBAM(x)                                                             layers.py:132-141
    smaat::upsample_into       cat[:, c_off:] = pad(upsample2x(x1)) (functional: returns the filled buffer)   :64,76-85
    smaat::pointwise_train     OutConv                                                             unet_parts.py:67-73
each with a `*_bwd` companion operator that the registered autograd formula calls, and fake (meta) kernels for both.

They run the SAME host code and kernels as the autograd.Functions of ops.py (`_half_forward`, `_cbam_forward_impl`, ...),
in both precisions.  What an operator-per-node decomposition cannot express are the two cross-node fusions of the
default path -- the head fusion (last block + OutConv as one node, no 64-channel output tensor) and the deferred encoder
activation (an encoder block hands its pre-BatchNorm tensor to the attention block) -- so the traceable wiring is ~3 %
slower per step and is opt-in:  `with smaat_unet_amd.traceable_training():`  (or SMAAT_TRACEABLE=1).
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import List, Optional

import torch
from torch import Tensor
from torch.library import custom_op

from . import _lib, ops

_TLS = threading.local()


@contextlib.contextmanager
def traceable_training(enabled=True):
    """modules called inside use the custom-operator wiring of the training path (see the module docstring)"""
    prev = getattr(_TLS, "on", None)
    _TLS.on = bool(enabled)
    try:
        yield
    finally:
        _TLS.on = prev


def active():
    on = getattr(_TLS, "on", None)
    return (os.environ.get("SMAAT_TRACEABLE", "0") == "1") if on is None else on


def _e(ref):  # placeholder for "no tensor" in an operator result (operators return tensors only)
    return ref.new_empty((0,))


def _opt(t):  # ... and back
    return None if t is None or t.numel() == 0 else t


def _mo(m):  # momentum: None (cumulative average) travels as -1
    return -1.0 if m is None else float(m)


def _mo_back(m):
    return None if m < 0 else m


# --------------------------------------------------------------------------------------------------------------------
# DoubleConvDS
# --------------------------------------------------------------------------------------------------------------------
def _dc_plan(x, w_pw1, g1, kpl, bf16, keep):
    """(fuse the first activation into its consumers?, element dtype of the activations)"""
    n, _, h, w = x.shape
    bf = bf16 or x.dtype == torch.bfloat16
    fuse = (ops.policy.fuse_first_activation and keep and g1 is not None and bool(_lib.get().smaat_dw3x3_strip_ok(kpl, h, w))
            and (not bf or kpl <= 2))
    return fuse, (torch.bfloat16 if bf else x.dtype)


# The operators are FUNCTIONAL (an autograd formula can only be registered for a functional operator): the running
# statistics go in read-only and their updated values come back as results; the thin Python wrappers below copy them into
# the module buffers (a traced buffer mutation, eight floats-per-channel vectors).
@custom_op("smaat::double_conv_ds", mutates_args=())
def double_conv_ds_op(x: Tensor, w_dw1: Tensor, b_dw1: Optional[Tensor], w_pw1: Tensor, b_pw1: Optional[Tensor],
                      g1: Optional[Tensor], be1: Optional[Tensor], rm1: Optional[Tensor], rv1: Optional[Tensor],
                      w_dw2: Tensor, b_dw2: Optional[Tensor], w_pw2: Tensor, b_pw2: Optional[Tensor], g2: Optional[Tensor],
                      be2: Optional[Tensor], rm2: Optional[Tensor], rv2: Optional[Tensor], tr1: bool, mo1: float, eps1: float,
                      tr2: bool, mo2: float, eps2: float, kpl: int, bf16: bool, keep: bool) -> List[Tensor]:
    """-> [y2, z1, st1, ydw1, y1, z2, st2, ydw2, rm1', rv1', rm2', rv2']  (ydw*: kept depthwise outputs, y1: empty when the
    activation is applied on load, rm' / rv': the updated running statistics, empty when not tracked / not training)"""
    ops._check(x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2)
    ops._expect_dsconv(x, w_dw1, b_dw1, w_pw1, b_pw1, kpl)
    w_dw1, w_pw1, w_dw2, w_pw2 = (t.contiguous() for t in (w_dw1, w_pw1, w_dw2, w_pw2))
    rm1, rv1, rm2, rv2 = (t.clone() if t is not None else None for t in (rm1, rv1, rm2, rv2))
    fuse, _ = _dc_plan(x, w_pw1, g1, kpl, bf16, keep)
    with ops.precision("bf16" if bf16 else None):
        y1, z1, st1, ydw1, _ = ops._half_forward(x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, tr1, _mo_back(mo1), eps1, kpl,
                                                 keep, want_act=not fuse)
        y2, z2, st2, ydw2, _ = ops._half_forward(z1 if fuse else y1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2, tr2,
                                                 _mo_back(mo2), eps2, kpl, keep, in_aff=(st1[2], st1[3]) if fuse else None)
    return [y2, z1, st1, ydw1 if ydw1 is not None else _e(x), y1 if y1 is not None else _e(x), z2, st2,
            ydw2 if ydw2 is not None else _e(x)] + [t if (t is not None and tr) else _e(x)
                                                    for t, tr in ((rm1, tr1), (rv1, tr1), (rm2, tr2), (rv2, tr2))]


@double_conv_ds_op.register_fake
def _(x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2, tr1, mo1, eps1, tr2, mo2,
      eps2, kpl, bf16, keep):
    n, cin, h, w = x.shape
    c1, c2 = w_pw1.shape[0], w_pw2.shape[0]
    fuse, dt = _dc_plan(x, w_pw1, g1, kpl, bf16, keep)
    act = lambda c: x.new_empty((n, c, h, w), dtype=dt)  # noqa: E731
    st = lambda c: x.new_empty((4, c), dtype=torch.float32)  # noqa: E731
    e = x.new_empty((0,))
    stat = lambda t, tr, c: x.new_empty((c,), dtype=torch.float32) if (t is not None and tr) else e  # noqa: E731
    return [act(c2), act(c1), st(c1), act(cin * kpl) if keep else e, e if fuse else act(c1), act(c2), st(c2),
            act(c1 * kpl) if keep else e, stat(rm1, tr1, c1), stat(rv1, tr1, c1), stat(rm2, tr2, c2), stat(rv2, tr2, c2)]


@custom_op("smaat::double_conv_ds_bwd", mutates_args=())
def double_conv_ds_bwd_op(dy2: Tensor, x: Tensor, w_dw1: Tensor, b_dw1: Optional[Tensor], w_pw1: Tensor, g1: Optional[Tensor],
                          z1: Tensor, st1: Tensor, ydw1: Tensor, y1: Tensor, w_dw2: Tensor, b_dw2: Optional[Tensor],
                          w_pw2: Tensor, g2: Optional[Tensor], z2: Tensor, st2: Tensor, ydw2: Tensor, kpl: int, ts1: bool,
                          ts2: bool, hb1: bool, hb2: bool, need_dx: bool) -> List[Tensor]:
    """-> [dx, dw_dw1, db_dw1, dw_pw1, db_pw1, dg1, dbe1, dw_dw2, db_dw2, dw_pw2, db_pw2, dg2, dbe2] (empty = no gradient)"""
    y1 = _opt(y1)
    fuse = y1 is None
    gr2, red = ops._half_backward(z1 if fuse else y1, w_dw2, b_dw2, w_pw2, g2, z2, st2, _opt(ydw2), dy2, kpl, ts2,
                                  (b_dw2 is not None, hb2), True, bnred=(st1[0], st1[1]) if fuse else None,
                                  in_aff=(st1[2], st1[3]) if fuse else None)
    gr1, _ = ops._half_backward(x, w_dw1, b_dw1, w_pw1, g1, z1, st1, _opt(ydw1), gr2[0], kpl, ts1, (b_dw1 is not None, hb1),
                                need_dx, pre_part=red)
    out = list(gr1) + list(gr2[1:])
    # (an operator's results must not alias each other: the exactly-zero bias gradients are slices of one zero arena,
    # ops._zero_grad_words -- this operator runs on autograd's thread, where the thread-local `active()` is not visible)
    return [(t.clone() if t._base is not None else t) if t is not None else _e(x) for t in out]


@double_conv_ds_bwd_op.register_fake
def _(dy2, x, w_dw1, b_dw1, w_pw1, g1, z1, st1, ydw1, y1, w_dw2, b_dw2, w_pw2, g2, z2, st2, ydw2, kpl, ts1, ts2, hb1, hb2, need_dx):
    f32 = lambda t: t.new_empty(t.shape, dtype=torch.float32)  # noqa: E731
    e = x.new_empty((0,))
    c1, c2 = w_pw1.shape[0], w_pw2.shape[0]
    vec = lambda c: x.new_empty((c,), dtype=torch.float32)  # noqa: E731
    return [x.new_empty(x.shape) if need_dx else e, f32(w_dw1), vec(w_dw1.shape[0]) if b_dw1 is not None else e, f32(w_pw1),
            vec(c1) if hb1 else e, vec(c1) if g1 is not None else e, vec(c1) if g1 is not None else e, f32(w_dw2),
            vec(w_dw2.shape[0]) if b_dw2 is not None else e, f32(w_pw2), vec(c2) if hb2 else e,
            vec(c2) if g2 is not None else e, vec(c2) if g2 is not None else e]


def _dc_setup(ctx, inputs, output):
    (x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2, tr1, mo1, eps1, tr2, mo2, eps2,
     kpl, bf16, keep) = inputs
    y2, z1, st1, ydw1, y1, z2, st2, ydw2 = output[:8]
    ctx.save_for_backward(x, w_dw1, b_dw1, w_pw1, g1, z1, st1, ydw1, y1, w_dw2, b_dw2, w_pw2, g2, z2, st2, ydw2)
    ctx.kpl = kpl
    ctx.ts = (tr1 or rm1 is None, tr2 or rm2 is None)
    ctx.hb = (b_pw1 is not None, b_pw2 is not None)
    ctx.need_dx = ctx.needs_input_grad[0]


def _dc_backward(ctx, *grads):
    grads = grads[0] if len(grads) == 1 and isinstance(grads[0], (list, tuple)) else grads
    x, w_dw1, b_dw1, w_pw1, g1, z1, st1, ydw1, y1, w_dw2, b_dw2, w_pw2, g2, z2, st2, ydw2 = ctx.saved_tensors
    g = torch.ops.smaat.double_conv_ds_bwd(grads[0].contiguous(), x, w_dw1, b_dw1, w_pw1, g1, z1, st1, ydw1, y1, w_dw2, b_dw2, w_pw2,
                                           g2, z2, st2, ydw2, ctx.kpl, ctx.ts[0], ctx.ts[1], ctx.hb[0], ctx.hb[1], ctx.need_dx)
    dx, dwd1, dbd1, dwp1, dbp1, dg1, dbe1, dwd2, dbd2, dwp2, dbp2, dg2, dbe2 = (_opt(t) for t in g)
    return (dx, dwd1, dbd1, dwp1, dbp1, dg1, dbe1, None, None, dwd2, dbd2, dwp2, dbp2, dg2, dbe2, None, None) + (None,) * 9


double_conv_ds_op.register_autograd(_dc_backward, setup_context=_dc_setup)


def double_conv_ds(x, half1, half2, kpl):
    """half = (w_dw, b_dw, w_pw, b_pw, gamma, beta, running_mean, running_var, training, momentum, eps) as ops.double_conv_ds"""
    a, b = half1, half2
    keep = torch.is_grad_enabled()
    bf16 = ops.mixed_precision_active() or x.dtype == torch.bfloat16
    r = torch.ops.smaat.double_conv_ds(x, *a[:8], *b[:8], a[8], _mo(a[9]), a[10], b[8], _mo(b[9]), b[10], kpl, bf16, keep)
    with torch.no_grad():
        for buf, new in ((a[6], r[8]), (a[7], r[9]), (b[6], r[10]), (b[7], r[11])):
            if buf is not None and new.numel():
                buf.copy_(new)
    return r[0]


# --------------------------------------------------------------------------------------------------------------------
# CBAM (+ MaxPool2d + concatenation buffer)
# --------------------------------------------------------------------------------------------------------------------
# ops._cbam_forward_impl saves (x, w1, w2, wconv, gamma, avg, mx, amax, ha, hm, sc, maps, conv, st, gate, amaxc): the first five
# are the operator's own inputs (an operator must not return its inputs), the other eleven are returned and saved by the formula
# (amaxc, the channel index map of the three-pass backward, is an empty int32 tensor when that route is off or not applicable)
_N_CBAM_SAVED = 11


def _cbam_has_index_map(h, w):
    return ops.policy.cbam_three_pass and (h * w) % 4 == 0


def _cbam_fake_saved(x, w1):
    n, c, h, w = x.shape
    f = lambda *s: x.new_empty(s, dtype=torch.float32)  # noqa: E731
    cr = w1.shape[0]
    return [f(n, c), f(n, c), x.new_empty((n, c), dtype=torch.int32), f(n, cr), f(n, cr), f(n, c), f(n, 2, h, w), f(n, 1, h, w),
            f(4, 1), f(n, 1, h, w), x.new_empty((n, h, w) if _cbam_has_index_map(h, w) else (0,), dtype=torch.int32)]


@custom_op("smaat::cbam_pool_cat", mutates_args=())
def cbam_pool_cat_op(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, wconv: Tensor, gamma: Optional[Tensor],
                     beta: Optional[Tensor], rm: Optional[Tensor], rv: Optional[Tensor], training: bool, momentum: float,
                     eps: float, c_extra: int, pool: bool) -> List[Tensor]:
    """-> [cat (CBAM(x) in channels [0, C) of a [N, C + c_extra, H, W] buffer), maxpool2(x) or empty, avg, mx, amax, ha, hm, sc,
    maps, conv, st, gate, amaxc or empty, rm', rv']"""
    xx, x_bs = ops._planes(x)
    n, c, h, w = xx.shape
    rm, rv = (t.clone() if t is not None else None for t in (rm, rv))
    cat = torch.empty((n, c + c_extra, h, w), dtype=xx.dtype, device=xx.device)
    _, saved, _ = ops._cbam_forward_impl(xx, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, _mo_back(momentum), eps, True,
                                         True, out=cat[:, :c], pool=(got := []) if pool else None)
    pooled = (got[0] if got else ops._maxpool2_fwd_raw(xx, x_bs)) if pool else _e(xx)
    sv = list(saved[5:])
    assert len(sv) == _N_CBAM_SAVED and (sv[-1] is not None) == _cbam_has_index_map(h, w)
    if sv[-1] is None:
        sv[-1] = xx.new_empty((0,), dtype=torch.int32)
    return [cat, pooled] + sv + [t if (t is not None and training) else _e(xx) for t in (rm, rv)]


@cbam_pool_cat_op.register_fake
def _(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra, pool):
    n, c, h, w = x.shape
    stat = lambda t: x.new_empty((1,), dtype=torch.float32) if (t is not None and training) else x.new_empty((0,))  # noqa: E731
    return [x.new_empty((n, c + c_extra, h, w)), x.new_empty((n, c, h // 2, w // 2)) if pool else x.new_empty((0,))] + \
        _cbam_fake_saved(x, w1) + [stat(rm), stat(rv)]


@custom_op("smaat::cbam_pool_cat_bwd", mutates_args=())
def cbam_pool_cat_bwd_op(dcat: Tensor, dpooled: Tensor, x: Tensor, w1: Tensor, w2: Tensor, wconv: Tensor,
                         gamma: Optional[Tensor], saved: List[Tensor], train_stats: bool) -> List[Tensor]:
    """-> [dx, dw1, db1, dw2, db2, dwconv, dgamma, dbeta]"""
    c = x.shape[1]
    saved = list(saved)
    if saved[-1].numel() == 0:
        saved[-1] = None  # (no channel index map: the gate / main / final sequence)
    sv = (ops._planes(x)[0], w1.contiguous(), w2.contiguous(), wconv.contiguous(), gamma) + tuple(saved)
    pooled = ops._planes(dpooled) if dpooled.numel() else None
    g = list(ops._cbam_backward_impl(sv, (True, True, train_stats), dcat[:, :c], pooled=pooled))
    for i in (1, 2, 3, 4):  # the MLP gradients are views of one reduction buffer: an operator's results must not alias
        g[i] = g[i].clone() if g[i] is not None else None
    return [t if t is not None else _e(dcat) for t in g]


@cbam_pool_cat_bwd_op.register_fake
def _(dcat, dpooled, x, w1, w2, wconv, gamma, saved, train_stats):
    c = x.shape[1]
    f = lambda *s: x.new_empty(s, dtype=torch.float32)  # noqa: E731
    cr = w1.shape[0]
    g = gamma is not None
    return [x.new_empty(x.shape), f(cr, c), f(cr), f(c, cr), f(c), f(*wconv.shape), f(1) if g else f(0), f(1) if g else f(0)]


def _cpc_setup(ctx, inputs, output):
    x, w1, _b1, w2, _b2, wconv, gamma = inputs[:7]
    ctx.train_stats = bool(inputs[10] or inputs[8] is None)
    ctx.save_for_backward(x, w1, w2, wconv, gamma, *output[2:2 + _N_CBAM_SAVED])


def _cpc_backward(ctx, *grads):
    grads = grads[0] if len(grads) == 1 and isinstance(grads[0], (list, tuple)) else grads
    x, w1, w2, wconv, gamma, *saved = ctx.saved_tensors
    dcat, dpooled = grads[0], grads[1]
    if dcat is None:
        dcat = x.new_zeros(x.shape)
    if dpooled is None:
        dpooled = x.new_empty((0,))
    g = torch.ops.smaat.cbam_pool_cat_bwd(dcat.contiguous(), dpooled.contiguous(), x, w1, w2, wconv, gamma, list(saved),
                                          ctx.train_stats)
    dx, dw1, db1, dw2, db2, dwconv, dgamma, dbeta = (_opt(t) for t in g)
    return (dx, dw1, db1, dw2, db2, dwconv, dgamma, dbeta) + (None,) * 7


cbam_pool_cat_op.register_autograd(_cpc_backward, setup_context=_cpc_setup)


def _cbam_call(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra, pool):
    r = torch.ops.smaat.cbam_pool_cat(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, _mo(momentum), eps, c_extra, pool)
    with torch.no_grad():
        for buf, new in ((rm, r[2 + _N_CBAM_SAVED]), (rv, r[3 + _N_CBAM_SAVED])):
            if buf is not None and new.numel():
                buf.copy_(new)
    return r


def cbam_pool_cat(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra):
    r = _cbam_call(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra, True)
    return r[0], r[1]


def cbam(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps):
    """CBAM(x) as the same operator without extra channels and without the pooled output"""
    return _cbam_call(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, 0, False)[0]


# --------------------------------------------------------------------------------------------------------------------
# Upsample into the concatenation buffer (functional form: the filled buffer is a new value for the tracer; the kernel
# writes in place and the operator declares the mutation)
# --------------------------------------------------------------------------------------------------------------------
@custom_op("smaat::upsample_into", mutates_args=("cat",))
def upsample_into_op(cat: Tensor, x1: Tensor, c_off: int) -> Tensor:
    with torch.no_grad():
        ops._UpsampleInto.apply(cat, x1, c_off, None)
    return cat.new_empty((0,))  # (the result is the mutated `cat`)


@upsample_into_op.register_fake
def _(cat, x1, c_off):
    return cat.new_empty((0,))


@custom_op("smaat::upsample_into_bwd", mutates_args=())
def upsample_into_bwd_op(dcat: Tensor, c_off: int, c1: int, h: int, w: int) -> Tensor:
    n, ct, ho, wo = dcat.shape
    dx1 = torch.empty((n, c1, h, w), dtype=dcat.dtype, device=dcat.device)
    pt, pl = (ho - 2 * h) // 2, (wo - 2 * w) // 2
    ops._upsample_bwd_raw(dcat.data_ptr() + dcat.element_size() * c_off * ho * wo, ct * ho * wo, dx1, n, c1, h, w, ho, wo, pt, pl,
                          ops._stream(dcat))
    return dx1


@upsample_into_bwd_op.register_fake
def _(dcat, c_off, c1, h, w):
    return dcat.new_empty((dcat.shape[0], c1, h, w))


class _UpsampleIntoTraceable(torch.autograd.Function):
    """autograd glue only (no kernels here): forward = the mutating operator, backward = its companion operator"""

    @staticmethod
    def forward(ctx, cat, x1, c_off):
        torch.ops.smaat.upsample_into(cat, x1, c_off)
        ctx.geom = (c_off, x1.shape[1], x1.shape[2], x1.shape[3])
        ctx.mark_dirty(cat)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        c_off, c1, h, w = ctx.geom
        dcat = dcat.contiguous()
        return dcat, torch.ops.smaat.upsample_into_bwd(dcat, c_off, c1, h, w), None


def upsample_into(cat, x1, c_off):
    return _UpsampleIntoTraceable.apply(cat, x1, c_off)


# --------------------------------------------------------------------------------------------------------------------
# OutConv
# --------------------------------------------------------------------------------------------------------------------
@custom_op("smaat::pointwise_train", mutates_args=())
def pointwise_train_op(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    with torch.no_grad():
        return ops._Pointwise.apply(x, w, b)


@pointwise_train_op.register_fake
def _(x, w, b):
    n, _, h, wd = x.shape
    return x.new_empty((n, w.shape[0], h, wd), dtype=torch.float32)


@custom_op("smaat::pointwise_train_bwd", mutates_args=())
def pointwise_train_bwd_op(dz: Tensor, x: Tensor, w: Tensor, has_bias: bool, need_dx: bool) -> List[Tensor]:
    m, c = w.shape[0], w.shape[1]
    w = w.contiguous()
    if x.dtype == torch.bfloat16:
        dzb = dz.to(torch.bfloat16)
        dx = ops._pointwise_bf16_raw(dzb, ops._bf16_planes_raw(w.reshape(m, c), transpose=True), None, c)[0] if need_dx else None
        dw = ops._pointwise_wgrad_raw(x, dzb, m)
    else:
        dx = ops._pointwise_raw(dz, w.reshape(m, c), None, c) if need_dx else None
        dw = ops._pointwise_wgrad_raw(x, dz, m)
    db = ops._channel_sum_raw(dz) if has_bias else None
    return [t if t is not None else _e(dz) for t in (dx, dw, db)]


@pointwise_train_bwd_op.register_fake
def _(dz, x, w, has_bias, need_dx):
    e = dz.new_empty((0,))
    return [x.new_empty(x.shape) if need_dx else e, w.new_empty(w.shape, dtype=torch.float32),
            w.new_empty((w.shape[0],), dtype=torch.float32) if has_bias else e]


def _pw_setup(ctx, inputs, output):
    x, w, b = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = b is not None
    ctx.need_dx = ctx.needs_input_grad[0]


def _pw_backward(ctx, dz):
    x, w = ctx.saved_tensors
    dx, dw, db = (_opt(t) for t in torch.ops.smaat.pointwise_train_bwd(dz.contiguous(), x, w, ctx.has_bias, ctx.need_dx))
    return dx, dw, db


pointwise_train_op.register_autograd(_pw_backward, setup_context=_pw_setup)


def pointwise(x, w, b):
    return torch.ops.smaat.pointwise_train(x, w, b)
