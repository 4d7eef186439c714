"""SmaAt-UNet and its depthwise-separable siblings on the MI355X blocks.

`SmaAt_UNet` is the drop-in for /root/reference/models/SmaAt_UNet.py (constructor :8-15, attributes :17-20,
submodule names :23-39 => the same 214 `state_dict` keys in the same order, wiring :41-57).  The network is
described by a small table instead of one statement per layer: an encoder of four `DownDS` levels behind `inc`,
`cbam_levels` attention blocks (5: every level and the bottleneck, 4: the skips only, 0: none) and a decoder of
four `UpDS` levels.  Submodules are registered in the reference's order (inc, cbam1, down1, cbam2, ... , up1..4,
outc), which is what fixes the `state_dict` key order.
"""
from __future__ import annotations

import os

from torch import nn

from . import ops
from .layers import CBAM, batched_counters
from .unet_parts import OutConv
from .layers import DepthwiseSeparableConv
from .unet_parts_depthwise_separable import DoubleConvDS, DownDS, UpDS

_ENCODER_WIDTHS = (64, 128, 256, 512)


class UNetDSFamily(nn.Module):
    """inc -> 4 x (CBAM on the skip?, DownDS) -> (CBAM on the bottleneck?) -> 4 x UpDS -> OutConv"""

    def __init__(self, n_channels, n_classes, kernels_per_layer, bilinear, reduction_ratio, cbam_levels):
        super().__init__()
        if cbam_levels not in (0, 4, 5):
            raise ValueError("cbam_levels must be 0, 4 or 5")
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        self.cbam_levels = cbam_levels
        shrink = 2 if bilinear else 1
        widths = _ENCODER_WIDTHS + (1024 // shrink,)
        kw = dict(kernels_per_layer=kernels_per_layer)
        self.inc = DoubleConvDS(n_channels, widths[0], **kw)
        for lvl in range(1, 5):
            if cbam_levels >= lvl:
                setattr(self, f"cbam{lvl}", CBAM(widths[lvl - 1], reduction_ratio=reduction_ratio))
            setattr(self, f"down{lvl}", DownDS(widths[lvl - 1], widths[lvl], **kw))
        if cbam_levels == 5:
            self.cbam5 = CBAM(widths[4], reduction_ratio=reduction_ratio)
        cat_channels = 1024
        for i in range(1, 5):  # up_i takes the concatenation [skip, upsampled] of cat_channels channels
            out = 64 if i == 4 else cat_channels // 2 // shrink
            setattr(self, f"up{i}", UpDS(cat_channels, out, bilinear, **kw))
            cat_channels //= 2
        self.outc = OutConv(64, n_classes)

    # -- precision -----------------------------------------------------------------------------------------
    def set_precision(self, mode):
        """"f32" (default) or "bf16" = mixed precision (BASELINE configs[3]): activations and their gradients are stored as
        bfloat16, pointwise GEMMs run on the bf16 matrix pipe with f32 accumulation, parameters / gradients of parameters
        / BatchNorm statistics stay f32.  The reference has no such switch; the equivalent there is Lightning's
        precision="bf16-mixed" (torch.autocast around models/SmaAt_UNet.py:41-57), which this module honours too: under
        torch.autocast(device_type="cuda", dtype=torch.bfloat16) the forward runs in mixed precision without this call.
        Inputs and logits stay float32.  Applies to the training / grad-enabled path; the eval fast path is f32.
        None = follow the surrounding `smaat_unet_amd.precision(...)` context / autocast."""
        if mode not in (None, "f32", "bf16"):
            raise ValueError("precision must be 'f32', 'bf16' or None")
        self._precision = mode
        self.__dict__["_graphs"] = {}
        return self

    def _mixed_precision_ok(self, x):
        """bf16 activation storage is built for the configurations the reference trains (bilinear up path, 3x3 depthwise
        with kernels_per_layer 1, 2 or 4, input height and width multiples of 32: every level then has an even plane and
        every upsampled width is a multiple of 4).  Anything else runs with f32 storage -- mixed precision is an
        optimisation, not a contract on results -- and says so once."""
        ok = (x.dim() == 4 and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0 and getattr(self, "bilinear", True)
              and all(m._fast_geometry() for m in self.modules() if isinstance(m, DepthwiseSeparableConv)))
        if not ok and not self.__dict__.get("_warned_mixed"):
            self.__dict__["_warned_mixed"] = True
            import warnings
            warnings.warn("smaat_unet_amd: mixed precision (bf16 activation storage) is built for the bilinear up path, "
                          "kernels_per_layer in {1, 2, 4} and input sizes that are multiples of 32; this call runs with f32 "
                          f"storage (input {tuple(x.shape)})", stacklevel=3)
        return ok

    # -- pieces ------------------------------------------------------------------------------------------
    def _levels(self):
        downs = [getattr(self, f"down{l}") for l in range(1, 5)]
        ups = [getattr(self, f"up{i}") for i in range(1, 5)]
        cbams = [getattr(self, f"cbam{l}", None) for l in range(1, 6)]
        return downs, ups, cbams

    def _fusable(self):
        """the fused skip wiring bypasses the `forward` of cbamN / downN.maxpool / upN: keep the
        module-by-module path whenever a user hooked one of them (or uses an exotic configuration)."""
        if self.cbam_levels < 4:
            return False
        downs, ups, cbams = self._levels()
        # (inc, outc and the bottleneck attention are called as modules, but with internal keyword arguments / tuple
        # values of the deferred-activation and head fusions: a hook on them gets the module-by-module path as well)
        for top in downs + ups + [c for c in cbams if c is not None] + [self.inc, self.outc]:
            for mm in top.modules():
                if mm._forward_hooks or mm._forward_pre_hooks or mm._backward_hooks:
                    return False
        return True

    # ---- inference: the whole forward as ONE captured hipGraph, owned by the module -----------------------------
    MAX_EVAL_GRAPHS = 8  # captured input shapes kept per module (least recently used one dropped beyond that)
    # Opt-in (SMAAT_FORK_ATTENTION=1): capture the skip connections' attention as parallel branches of the graph.  Measured on
    # MI355X / ROCm 7.0 at batch 1, 288 x 288: 0.711 ms against 0.687 ms without -- the graph executor does not overlap the
    # branches (each replay still runs its kernels back to back) and the two extra dependency edges per level cost ~5 us
    # each; kept as a switch for runtimes that do.
    FORK_ATTENTION = os.environ.get("SMAAT_FORK_ATTENTION", "0") == "1"
    FORK_MAX_PIXELS = 4 * 288 * 288  # N * H * W up to which the captured graph forks the attention branches

    def enable_eval_graph(self, enabled=True, clone_output=True):
        """Inference (eval mode under no_grad, reference call stack D): capture the forward for every input shape
        seen into a hipGraph and replay it -- ~40 kernel launches become one graph launch.  The graph is rebuilt when
        a parameter / buffer changes or the shape changes.  clone_output=False returns the graph's static output buffer
        (overwritten by the next call) and saves one copy.

        Change detection is cheap by construction (it runs on every call of a 0.6 ms forward): the sum of the autograd
        version counters of the parameters and buffers -- every in-place torch operation, optimizer step and
        load_state_dict bumps one -- plus a dirty flag raised by train(), .to() / .half() / ... (`_apply`),
        load_state_dict, set_precision and invalidate_eval_cache().  Writes through `.data` (EMA / SWA weight swaps,
        `.data`-based clamping) bypass the version counters: call `invalidate_eval_cache()` after them."""
        self._graph_enabled = bool(enabled)
        self._graph_clone = bool(clone_output)
        self._graphs = {}
        self._sig_tensors = None
        return self

    def invalidate_eval_cache(self):
        """Drop everything the inference fast path derived from the weights: captured hipGraphs, BatchNorm folded into
        the pointwise weights and their operand images (DoubleConvDS._fold_cache).  Needed after weight updates that
        bypass torch's version counters (writes through `.data`); everything else is detected automatically."""
        self._graphs = {}
        self._sig_tensors = None
        for m in self.modules():
            if "_fold_cache" in m.__dict__:
                m.__dict__["_fold_cache"] = {}
        return self

    def train(self, mode=True):
        if mode:  # weights are about to change: captured inference state is stale from here on
            self.__dict__["_graphs"] = {}
        return super().train(mode)

    def _apply(self, fn, *a, **kw):
        self.__dict__["_graphs"] = {}
        self.__dict__["_sig_tensors"] = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self.invalidate_eval_cache()
        return r

    def _weights_signature(self):
        """sum of the autograd version counters of every parameter and buffer, or -1 (never equal to a stored
        signature) when a Parameter / buffer OBJECT was replaced since the list was cached -- `module.weight =
        nn.Parameter(...)`, `parent.load_state_dict(sd, assign=True)`, parametrize, swap_tensors: the cached graph holds
        the old storage and would silently replay the old weights.  One pass of ~360 dict lookups (~15 us).
        Remaining blind spot: writes through `.data` (see enable_eval_graph)."""
        ents = self.__dict__.get("_sig_tensors")
        if ents is None:  # (rebuilt after _apply / load_state_dict / invalidate_eval_cache)
            ents = []
            for m in self.modules():
                ents += [(m._parameters, k, t) for k, t in m._parameters.items() if t is not None]
                ents += [(m._buffers, k, t) for k, t in m._buffers.items() if t is not None]
            self.__dict__["_sig_tensors"] = ents
        v = 0
        for d, k, t in ents:
            if d.get(k) is not t:
                self.__dict__["_sig_tensors"] = None
                return -1
            v += t._version
        return v

    def _graph_forward(self, x):
        import torch
        key = (tuple(x.shape), x.dtype, x.device)
        sig = self._weights_signature()
        ent = self._graphs.get(key)
        if ent is not None and (sig < 0 or ent["sig"] != sig):
            self.invalidate_eval_cache()  # (the folded weights are stale as well)
            ent = None
        if sig < 0:
            self.invalidate_eval_cache()
        if ent is None:
            while len(self._graphs) >= self.MAX_EVAL_GRAPHS:
                self._graphs.pop(next(iter(self._graphs)))
            static_in = x.detach().clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            # small batches: every kernel under-fills the chip and the forward is a chain of ~60 dependent launches; the
            # skip connections' attention then runs as a parallel branch of the graph (ops.cbam_eval_forked)
            fork = self.FORK_ATTENTION and x.shape[0] * x.shape[2] * x.shape[3] <= self.FORK_MAX_PIXELS
            self.__dict__["_fork_stream"] = torch.cuda.Stream(device=x.device) if fork else None
            try:
                with torch.cuda.stream(side):  # warm-up outside the capture: BatchNorm folding, lazy kernel attributes
                    self._forward_impl(static_in)
                torch.cuda.current_stream(x.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    static_out = self._forward_impl(static_in)
            finally:
                self.__dict__["_fork_stream"] = None
            ent = self._graphs[key] = dict(sig=self._weights_signature(), graph=g, x=static_in, y=static_out)
        else:
            self._graphs[key] = self._graphs.pop(key)  # most recently used last
        ent["x"].copy_(x)
        ent["graph"].replay()
        return ent["y"].clone() if self._graph_clone else ent["y"]

    def forward(self, x):
        import torch
        if (getattr(self, "_graph_enabled", False) and not self.training and not torch.is_grad_enabled() and x.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            return self._graph_forward(x)
        prec = getattr(self, "_precision", None)
        if (prec == "bf16" or (prec is None and ops.mixed_precision_active())) and not self._mixed_precision_ok(x):
            prec = "f32"  # (an enclosing torch.autocast / precision("bf16") is overridden for this call tree)
        with batched_counters(), ops.precision(prec):
            # (batched_counters: one add for all num_batches_tracked counters of the step)
            out = self._forward_impl(x)
        return out.float() if out.dtype != x.dtype and x.dtype.is_floating_point else out

    def _forward_impl(self, x):
        # NB (reference SmaAt_UNet.py:41-57): the encoder continues from the UN-attended x_i; the CBAM
        # outputs feed only the skip connections and the bottleneck.
        if not self._fusable():
            return self._forward_modular(x)
        downs, ups, cbams = self._levels()
        cats = []
        import torch
        # deferred activation (training path): an encoder block hands out its pre-BatchNorm tensor + coefficients and
        # the attention block that consumes it applies BatchNorm + ReLU inside its channel-pooling kernel, which also
        # writes the activated tensor -- the separate pass over every encoder output disappears
        from . import train_ops
        defer = self.FUSE_ENCODER_ACT and torch.is_grad_enabled() and not train_ops.active()
        h = self.inc(x, defer=True) if defer else self.inc(x)
        side = self.__dict__.get("_fork_stream")  # set by _graph_forward around the warm-up and the capture (small batches)
        keep = []
        for lvl in range(4):
            up = ups[3 - lvl]  # the decoder level that consumes this skip
            ch = (h[0] if isinstance(h, tuple) else h).shape[1]
            c_extra = up.conv.double_conv[0].depthwise.in_channels - ch
            forked = cbams[lvl].forward_pool_cat_forked(h, c_extra, side) if side is not None else None
            if forked is not None:  # the skip's attention runs beside the deeper encoder levels (joined below)
                cat, pooled, ka = forked
                keep.append(ka)
            else:
                cat, pooled = cbams[lvl].forward_pool_cat(h, c_extra)  # skip written straight into the decoder's cat buffer
            cats.append(cat)
            last = downs[lvl].maxpool_conv[1]
            h = last(pooled, defer=True) if (defer and cbams[4] is not None or defer and lvl < 3) else last(pooled)
        if cbams[4] is not None:
            h = cbams[4](h)
        if keep:
            torch.cuda.current_stream(x.device).wait_stream(side)
            keep.clear()  # (allocated on this stream: reusable from here on)
        head = self._fused_head()
        for i, (up, cat) in enumerate(zip(ups, reversed(cats))):
            if head is not None and i == len(ups) - 1:
                return up.forward_into(h, cat, head=head)  # last decoder block + OutConv as one node
            h = up.forward_into(h, cat)
        return self.outc(h)

    # class-level switches of the two cross-module fusions of the training path (environment overrides for A/B runs)
    FUSE_HEAD = os.environ.get("SMAAT_FUSE_HEAD", "1") != "0"
    FUSE_ENCODER_ACT = os.environ.get("SMAAT_FUSE_ENC", "1") != "0"

    def _fused_head(self):
        """the OutConv's nn.Conv2d when it has ONE output channel and can be fused with the BatchNorm + ReLU in front of
        it (training path: the 64-channel block output and its gradient are never materialised), else None"""
        import torch
        conv = self.outc.conv
        from . import train_ops
        if (not self.FUSE_HEAD or train_ops.active() or not torch.is_grad_enabled() or conv.out_channels != 1
                or conv.kernel_size != (1, 1)
                or any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in self.outc.modules())):
            return None
        return conv

    def _forward_modular(self, x):
        """module-by-module wiring (what the reference spells out statement by statement)"""
        downs, ups, cbams = self._levels()
        feats = [self.inc(x)]
        for down in downs:
            feats.append(down(feats[-1]))
        att = [f if c is None else c(f) for f, c in zip(feats, cbams)]
        h = att[4]
        for up, skip in zip(ups, reversed(att[:4])):
            h = up(h, skip)
        return self.outc(h)


class SmaAt_UNet(UNetDSFamily):
    def __init__(self, n_channels, n_classes, kernels_per_layer=2, bilinear=True, reduction_ratio=16):
        super().__init__(n_channels, n_classes, kernels_per_layer, bilinear, reduction_ratio, cbam_levels=5)
