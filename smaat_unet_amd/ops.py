"""PyTorch-ROCm autograd operators over the C ABI of libsmaat_hip.so.

Every forward/backward here only enqueues hand-written HIP kernels on the current
stream (plus tiny torch ops on [C]-sized vectors).  PyTorch provides device memory,
streams and autograd bookkeeping only.  No CPU fallback: host tensors are rejected.

Mapping to the reference (HansBambel/SmaAt-UNet):
  dsconv_bn_relu   DepthwiseSeparableConv -> BatchNorm2d -> ReLU   (models/layers.py:47-50,
                   models/unet_parts_depthwise_separable.py:17-36)
  dsconv           DepthwiseSeparableConv alone                    (models/layers.py:34-50)
  pointwise        OutConv                                          (models/unet_parts.py:67-73)
  maxpool2         nn.MaxPool2d(2)                                  (unet_parts_depthwise_separable.py:48)
  upsample_cat     nn.Upsample + F.pad + torch.cat([x2, x1])         (:64, :76-85)
  cbam             ChannelAttention / SpatialAttention / CBAM        (models/layers.py:90-141)
"""
from __future__ import annotations

import contextlib
import os
import weakref
import threading

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32


# --------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------

class Policy:
    """The kernel-selection switches of this module in ONE place: which entry point of include/smaat_hip.h runs
    for a layer follows from the shape, the storage type and these fields (DESIGN.md section 4 has the table).  Defaults are the
    measured policy; every field has the environment variable that sets it at import for A/B runs.  Tests and probes patch
    fields of `ops.policy` (monkeypatch.setattr(ops.policy, "f16_split", False)) -- no module globals, no re-import."""

    def __init__(self):
        e = os.environ.get
        self.f16_split = e("SMAAT_F16_SPLIT", "1") != "0"
        self.f16_min_samples = int(e("SMAAT_F16_MIN_SAMPLES", "4096"))
        self.split_policy = e("SMAAT_SPLIT_POLICY", "auto")
        self.fuse_dw_split = e("SMAAT_FUSE_DW", "auto")
        self.wgrad_recompute = e("SMAAT_WGRAD_RECOMPUTE", "auto")
        self.bf16_recompute = e("SMAAT_BF16_RECOMPUTE", "1") != "0"
        self.fwd_rows = e("SMAAT_FWD_ROWS", "auto")
        self.cbam_three_pass = e("SMAAT_CBAM_THREE_PASS", "1") != "0"
        self.plane_cache = e("SMAAT_PLANE_CACHE", "1") != "0"
        self.keep_depthwise_output = True
        self.fuse_first_activation = True
        self.splitk_train_budget = int(e("SMAAT_SPLITK_TRAIN", "2048"))
        self.fused_bwd = e("SMAAT_FUSED_BWD", "0") == "1"
        self.fwd_rows_h = e("SMAAT_FWD_ROWS_H", "1") != "0"

    def snapshot(self):
        return dict(vars(self))


policy = Policy()

def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def _check(*tensors, acts=1, bf16=True):
    """Operator-boundary type check.  The first `acts` tensors are activations (or activation gradients): float32, or
    bfloat16 when the operator has a bf16-storage kernel family (bf16=True; they dispatch on _dt()).  Every other tensor
    is a parameter or buffer and must be float32: the kernels read and WRITE them (running statistics) as 4-byte
    floats, so a model converted with .bfloat16() / Lightning "bf16-true" must fail here, not read out of bounds."""
    for i, t in enumerate(tensors):
        if t is None:
            continue
        if not t.is_cuda and not _lib._ALLOW_HOST_POINTERS:
            raise _lib.SmaatHipError("smaat_unet_amd operators need ROCm (cuda) tensors: there is no CPU fallback")
        if i < acts:
            if t.dtype == torch.float32 or (bf16 and t.dtype == torch.bfloat16):
                continue
            raise TypeError("smaat_unet_amd: this operator takes float32 activations"
                            + (" (or bfloat16 ones in mixed precision)" if bf16 else " only (no bf16-storage kernel)")
                            + f", got {t.dtype}")
        elif t.dtype != torch.float32:
            raise TypeError(f"smaat_unet_amd: parameters and buffers must be float32 (master weights; mixed precision "
                            f"stores only activations as bfloat16), got {t.dtype} -- use model.set_precision('bf16') or "
                            f"torch.autocast instead of model.bfloat16()")


def _dt(t):
    """dtype code of the C ABI (include/smaat_hip.h: SMAAT_DT_F32 = 0, SMAAT_DT_BF16 = 1)"""
    return 1 if t.dtype == torch.bfloat16 else 0


def _is_bf(t):
    return t is not None and t.dtype == torch.bfloat16


# ---- precision policy ----------------------------------------------------------------------------------------------
# "bf16" = mixed precision (BASELINE configs[3]): every activation tensor and its gradient is STORED as bfloat16, the
# pointwise GEMMs run one bf16 MFMA per product, accumulation / BatchNorm statistics / weights / weight gradients stay
# f32.  It is selected per call tree, not per process: `with precision("bf16"):` around a forward (the networks do this
# for themselves after `model.set_precision("bf16")`), or torch.autocast(device_type="cuda", dtype=torch.bfloat16) -- what
# Lightning's precision="bf16-mixed" wraps the reference's modules in.  Only the blocks that receive an f32 tensor (the
# stem) consult it; everything downstream follows the dtype of its input, and the backward follows the saved tensors.
_PREC = threading.local()


@contextlib.contextmanager
def precision(mode):
    """mode: "f32" | "bf16" | None (None = leave the surrounding setting)"""
    if mode not in (None, "f32", "bf16"):
        raise ValueError("precision must be 'f32' or 'bf16'")
    prev = getattr(_PREC, "mode", None)
    if mode is not None:
        _PREC.mode = mode
    try:
        yield
    finally:
        _PREC.mode = prev


def mixed_precision_active():
    mode = getattr(_PREC, "mode", None)
    if mode is not None:
        return mode == "bf16"
    try:
        return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
    except Exception:  # noqa: BLE001  (older torch signatures)
        return False


def _expect(t, shape, name):
    """argument validation at the operator boundary: a mismatching weight / activation shape must surface as a
    Python exception naming the tensor, not as a wrong read inside a kernel (None = any size, optional tensors pass)"""
    if t is None:
        return
    if t.dim() != len(shape) or any(e is not None and int(a) != int(e) for a, e in zip(t.shape, shape)):
        raise ValueError(f"{name}: expected shape {tuple('*' if e is None else e for e in shape)}, got {tuple(t.shape)}")


def _expect_dsconv(x, w_dw, b_dw, w_pw, b_pw, kpl, bn=()):
    if x.dim() != 4:
        raise ValueError(f"input: expected [N, C, H, W], got {tuple(x.shape)}")
    if kpl not in (1, 2, 4):
        raise NotImplementedError(f"kernels_per_layer={kpl}: the gfx950 kernels are built for 1, 2 and 4")
    cin = x.shape[1]
    k = cin * kpl
    _expect(w_dw, (k, 1, 3, 3), "depthwise.weight")
    _expect(b_dw, (k,), "depthwise.bias")
    _expect(w_pw, (None, k, 1, 1), "pointwise.weight")
    cout = w_pw.shape[0]
    _expect(b_pw, (cout,), "pointwise.bias")
    for name, t in bn:
        _expect(t, (cout,), name)
    return cout


def _planes(t):
    """(tensor usable by the kernels, batch stride): NCHW with dense [C][H][W] planes; the
    batch stride may be larger than C*H*W (channel slice of a cat buffer)."""
    n, c, h, w = t.shape
    st = t.stride()
    if st[3] == 1 and st[2] == w and (c == 1 or st[1] == h * w) and (n == 1 or st[0] >= c * h * w):
        return t, (st[0] if n > 1 else c * h * w)
    t = t.contiguous()
    return t, c * h * w


def _new(ref, *shape, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=ref.device)


# --------------------------------------------------------------------------------------
# raw kernel wrappers (thin; shapes derived from tensors)
# --------------------------------------------------------------------------------------
def _dsconv_fwd_raw(x, w_dw, b_dw, w_pw, b_pw, kpl, want_stats, in_scale=None, in_shift=None, want_y=False):
    L = _lib.get()
    x, x_bs = _planes(x)
    n, cin, h, w = x.shape
    cout = w_pw.shape[0]
    k = cin * kpl
    wt = w_pw.reshape(cout, k).t().contiguous()
    z = _new(x, n, cout, h, w)
    part = None
    slots = 0
    if want_stats:
        slots = L.smaat_pw_num_slots(n, h, w, cout)
        part = _new(x, 3, slots, cout)
    y = _new(x, n, k, h, w) if want_y else None
    _lib.check(L.smaat_dsconv_fwd(_ptr(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(wt),
                                  _ptr(b_pw), _ptr(z), cout * h * w, _ptr(part), _ptr(y), n, cin, kpl, cout, h, w,
                                  _stream(x)), "smaat_dsconv_fwd")
    if want_y:
        return z, part, slots, y
    return z, part, slots


def _pointwise_raw(x, wt, bias, m):
    """out[n][m][p] = sum_c wt[c][m] x[n][c][p] + bias[m]; wt is [C][m] contiguous."""
    L = _lib.get()
    x, x_bs = _planes(x)
    n, c, h, w = x.shape
    out = _new(x, n, m, h, w)
    _lib.check(L.smaat_pointwise_fwd(_ptr(x), x_bs, _ptr(wt), _ptr(bias), _ptr(out), m * h * w, None, n, c, m, h, w,
                                     _stream(x)), "smaat_pointwise_fwd")
    return out


# --------------------------------------------------------------------------------------
# bf16-split matrix path (include/smaat_hip.h "bf16-split matrix path"): f32 operands split exactly
# into three bf16 terms, six bf16 MFMAs per product, f32 accumulation.  On when the library says so
# (env SMAAT_SPLIT, default on); shapes it does not handle use the fused f32-MFMA kernel.
# --------------------------------------------------------------------------------------
def _split_on():
    return bool(_lib.get().smaat_split_enabled())


# Measured policy (profiles/r1, MI355X, batch 32): in training every supported layer runs on the split GEMMs
# (standalone strip depthwise kernel + persistent split GEMM; +1.7 % on the step against keeping the fused
# f32-MFMA forward for the plane-dominated layers, which is MFMA-bound at the f32 rate).  Inference
# (running statistics, typically batch 1) keeps the fused f32 kernel for the narrow layers: one launch
# instead of three matters more there than the matrix rate.
# policy.splitk_train_budget: workgroup items a sliced training GEMM may use (0 = off)
# policy.fuse_first_activation: DoubleConvDS: apply the first BatchNorm + ReLU on load instead of writing y1
# policy.split_policy: "auto" = the measured policy; "all" = every supported shape


def _split_all():
    # mode 1 (plain bf16 operands, one MFMA per product) is 6x cheaper on the matrix pipe: use it wherever supported
    return policy.split_policy == "all" or _lib.get().smaat_split_mode() == 1


def _split_fwd_ok(k, cout, train=True):
    if not _split_on():
        return False
    return _split_all() or train or (k >= 128 and cout >= 128)


def _split_dgrad_ok(cout, k):
    # the persistent split kernel beats the f32-MFMA one on every data gradient with a GEMM-sized output
    # (profiles/r1/r1p); the 24-channel stem stays on the f32 kernel
    return _split_on() and (_split_all() or (cout >= 64 and k >= 64))


# ---- two-term fp16 split (round 5; csrc/splitmma.hip NT == 2): three fp16 MFMAs per product instead of six bf16 ones ------
# On by default for the f32-storage training GEMMs whose operands come with their maxima: the forward pointwise GEMM (the
# depthwise kernel that writes its operand also leaves max |y|), the data gradient and the streamed weight gradient (the
# BatchNorm-backward apply kernel leaves max |dz|).  Everything else -- fused forwards, inference, ConvTranspose, OutConv,
# callers that pass no maxima -- runs the exact three-term bf16 split as before.  SMAAT_F16_SPLIT=0 switches it off (A/B).
# policy.f16_split (default os.environ.get("SMAAT_F16_SPLIT", "1") != "0")
# ... and only where the train-mode BatchNorm behind the GEMM averages at least this many samples per channel (N * H * W).
# The two-term split's per-product error is ~3x an f32 rounding (22-23 significant bits against exact products): invisible at
# the benchmark sizes (every reference fixture at 288^2 / 256^2 holds at unchanged bounds), but on planes of a few pixels one
# ReLU decision that flips at round-off level moves whole gradient tensors by 5e-3 ... 1e-2 (a 4 x 3 bottleneck plane at batch
# 3 is 36 samples: one flip is 3 % of them), and three times the forward noise means three times the flips -- the
# counter-examples are recorded in profiles/r5/f16_split_small_plane_counterexamples.txt (SMAAT_F16_MIN_SAMPLES=0 reproduces
# them).  BASELINE configs: 32 x 18 x 18 = 10,368 at the bottleneck of configs[1], 16 x 16 x 16 = 4,096 for configs[4].
# policy.f16_min_samples (default int(os.environ.get("SMAAT_F16_MIN_SAMPLES", "4096")))
_AMAX_LOCK = threading.Lock()
_AMAX_ARENA = {}  # (device, stream) -> [int32 tensor of zeros, next free word]


def _f16_on():
    if not policy.f16_split or not _split_on() or _lib.get().smaat_split_mode() != 3:
        return False
    from . import train_ops
    return not train_ops.active()  # (the traceable operators declare their saved tensors up front)


AMAX_WORDS = 1024  # SMAAT_AMAX_WORDS of include/smaat_hip.h: one operand-maximum buffer


def _amax_words(ref, n):
    """n fresh ZERO amax buffers (n * AMAX_WORDS int32 words) on ref's device: the operand-maximum buffers of
    include/smaat_hip.h "two-term fp16 split".  Handed out from a zero-filled arena and never reused, so a training step
    costs no fill launch: one 4 MB fill per 1024 buffers (~28 steps)."""
    n *= AMAX_WORDS
    if ref.is_cuda and torch.cuda.is_current_stream_capturing():
        return torch.zeros(n, dtype=torch.int32, device=ref.device)
    key = (ref.device, _stream(ref))  # (per stream: the arena's one fill launch is ordered before every use on that stream)
    with _AMAX_LOCK:
        a = _AMAX_ARENA.get(key)
        if a is None or a[1] + n > a[0].numel():
            a = [torch.zeros(max(1 << 20, n), dtype=torch.int32, device=ref.device), 0]
            _AMAX_ARENA[key] = a
        w = a[0][a[1]:a[1] + n]
        a[1] += n
    return w


# ---- maxima of tensors that live across autograd nodes (round 6) ------------------------------------------------------
# The row-walking fused forward on the two-term fp16 split (smaat_dsconv_fwd_rows_h) needs a bound of |x| BEFORE it runs.  For
# a decoder concatenation buffer that bound is the maximum its two writers leave (cbam_pool_cat: channels [0, C), upsample_into:
# the rest).  It travels beside the tensor OBJECT -- autograd hands the same Python object from node to node in eager mode --
# in a side table keyed by id(), guarded by a weak reference (the id of a dead object can be reused) and by the tensor's version
# counter (anything else that writes the tensor through torch invalidates the entry).  No entry, or a stale one: the consumer
# runs the exact three-term kernel as before.
_X_AMAX = {}  # id(tensor) -> [weakref, version, [(c_lo, c_hi, amax buffer), ...]]


def _note_x_amax(t, c_lo, c_hi, buf, extend=False):
    ent = _X_AMAX.get(id(t))
    if extend and ent is not None and ent[0]() is t:
        ent[1] = t._version
        ent[2].append((c_lo, c_hi, buf))
        return
    key = id(t)
    _X_AMAX[key] = [weakref.ref(t, lambda _r, k=key: _X_AMAX.pop(k, None)), t._version, [(c_lo, c_hi, buf)]]


def _x_amax_entry(t):
    """the live, current entry of t or None"""
    ent = _X_AMAX.get(id(t))
    if ent is None or ent[0]() is not t or ent[1] != t._version:
        return None
    return ent


def _x_amax_of(t):
    """-> (amax buffer, second amax buffer | None) covering every channel of t, or None"""
    ent = _x_amax_entry(t)
    if ent is None or t.dim() != 4 or len(ent[2]) > 2:
        return None
    rs = sorted(ent[2], key=lambda r: r[0])
    if rs[0][0] != 0 or rs[-1][1] != t.shape[1] or (len(rs) == 2 and rs[0][1] != rs[1][0]):
        return None
    return rs[0][2], (rs[1][2] if len(rs) == 2 else None)


def _want_x_amax(t):
    """producers: leave the maximum of what they write when a two-term fused forward could use it"""
    return policy.fwd_rows_h and policy.fwd_rows != "off" and t.dtype == F32 and _f16_on()


_ZERO_ARENA = {}  # (device, stream) -> [float32 tensor of zeros, next free element]


def _zero_grad_words(ref, n):
    """n float32 ZEROS on ref's device that nobody else will ever be handed: the exactly-zero gradient of a convolution bias
    in front of a train-mode BatchNorm (18 per training step; SURVEY 8c "zero-gradient trap").  A slice of a zero-filled arena
    instead of a fill launch per bias: autograd adopts the slice as `.grad` (it is referenced by nothing else), in-place
    arithmetic on it stays inside the slice, and a slice is never reused, so the arena is refilled once per ~200 steps.
    ALIASING: such a `.grad` is a VIEW into a 4 MB storage it shares with other parameters' zero gradients and with
    slices not yet handed out -- value-level use (optimizers, clipping, all-reduce, `.clone()`) cannot tell, but storage-level
    operations can: `torch.save(p.grad)` pickles the whole storage (save `p.grad.clone()`), `set_` / `resize_` / `share_memory_`
    act on the arena.  `SMAAT_ZERO_ARENA=0` hands out owning tensors instead (18 fill launches per step)."""
    if os.environ.get("SMAAT_ZERO_ARENA", "1") == "0":
        return torch.zeros(n, dtype=torch.float32, device=ref.device)
    from . import train_ops
    if train_ops.active() or (ref.is_cuda and torch.cuda.is_current_stream_capturing()):
        return torch.zeros(n, dtype=torch.float32, device=ref.device)  # (outputs of a torch.library operator must not alias)
    key = (ref.device, _stream(ref))
    with _AMAX_LOCK:
        a = _ZERO_ARENA.get(key)
        if a is None or a[1] + n > a[0].numel():
            a = [torch.zeros(max(1 << 20, n), dtype=torch.float32, device=ref.device), 0]
            _ZERO_ARENA[key] = a
        w = a[0][a[1]:a[1] + n]
        a[1] += (n + 3) // 4 * 4  # (16-byte aligned slices)
    return w


def set_matrix_mode(mode):
    """Arithmetic of the f32-STORAGE matrix path (process-wide A/B switch; precision proper is chosen per call tree, see
    `precision`).  Returns the previous mode string.
      "f32_split"        (default) f32-class split GEMMs on the 16-bit matrix pipe: the TWO-term fp16 operand split (three MFMAs
                         per product, operands to 22-23 bits, ~3 x an f32 rounding per product) where the operand maxima are at
                         hand and the BatchNorm behind the GEMM averages >= policy.f16_min_samples samples (i.e. it switches
                         with batch x plane size), the exact three-term bf16 split (six MFMAs) everywhere else;
      "f32_split_exact"  the exact three-term bf16 split everywhere (= "f32_split" with policy.f16_split off);
      "f32"              f32-MFMA kernels only (the reference's multiplier width; bench.py's `f32_mfma_only` leg);
      "bf16"             single-term bf16 operands on f32 storage (round-2 mixed mode; bench.py --precision bf16_operands)."""
    codes = {"f32": 0, "bf16": 1, "f32_split": 3, "f32_split_exact": 3}
    prev = _lib.get().smaat_set_split_mode(codes[mode])
    was_exact = not policy.f16_split
    if mode in ("f32_split", "f32_split_exact"):
        if policy.f16_split != (mode == "f32_split"):
            policy.f16_split = mode == "f32_split"
            invalidate_weight_images()
    name = {0: "f32", 1: "bf16", 2: "f32_split2", 3: "f32_split"}[prev]
    return "f32_split_exact" if (name == "f32_split" and was_exact) else name


# ---- operand images of the pointwise weights: one refresh launch per optimizer step -------------------------------------------
# A training step needs the image of every pointwise weight (forward) and of its transpose (data gradient): 35 launches of ~5 us
# doing a few KB of work each, one after the other in the stream (0.19 ms of a 32.6 ms step).  Every one of them goes stale at the
# same moment -- the optimizer step -- so images live in a cache keyed by the weight's storage, and the first stale image asked
# for after a weight update refreshes ALL registered stale images in ONE launch (smaat_weight_planes_multi).  A use is served from
# the cache only if the weight tensor is the same object at the same address with the same version counter; in-place updates
# (optimizer, load_state_dict, DDP broadcast) bump the version, replaced parameters are new objects; and because writes through
# `.data` bump nothing, an image is served at most ONCE per refresh -- its second use (a new pass over the network) refreshes all
# images again, so such writes are honoured from the next pass on, at one launch per pass.  The returned image is shared: callers
# only read it.  SMAAT_PLANE_CACHE=0 restores one launch per use.
# policy.plane_cache (default os.environ.get("SMAAT_PLANE_CACHE", "1") != "0")
_PLANES = {}
_PLANES_TABLE = {}  # tuple of entry ids -> device descriptor table
_PLANES_LOCK = threading.RLock()  # (replicas / a background evaluation thread share the cache)


def invalidate_weight_images():
    """Forget every cached operand image: the call to make after writing weights through `.data` (no version counter sees
    such a write) when the very next use must see it.  Without it a write through `.data` is seen from the SECOND use of the
    image after its last refresh on (see _weight_planes)."""
    with _PLANES_LOCK:
        _PLANES.clear()
        _PLANES_TABLE.clear()


class _PlaneEntry:
    __slots__ = ("ref", "off", "src", "version", "planes", "r", "c", "kind", "src_t", "dev", "stream", "mode", "nblk", "used")


def _weight_planes(w2d, transpose, kind):
    """kind 0: split planes (smaat_split_planes / _t), kind 2: bf16 image (smaat_bf16_planes), kind 3: fp16 two-term image +
    scale exponent (smaat_split_planes_h); see the note above"""
    with _PLANES_LOCK:
        return _weight_planes_locked(w2d, transpose, kind)


def _weight_planes_locked(w2d, transpose, kind):
    L = _lib.get()
    r, c = (w2d.shape[1], w2d.shape[0]) if transpose else w2d.shape
    if kind == 0:
        cp = (c + 15) // 16 * 16
        shape = (3, r, cp)
    elif kind == 3:
        cp = (c + 15) // 16 * 16
        shape = (int(L.smaat_split_planes_h_bytes(r, c)) // 2,)
    else:
        cp = (c + 31) // 32 * 32
        shape = (cp // 16, r, 16)

    def direct(planes=None):
        if planes is None:
            planes = torch.empty(shape, dtype=torch.int16, device=w2d.device)
        if kind == 0:
            fn = L.smaat_split_planes_t if transpose else L.smaat_split_planes
            _lib.check(fn(_ptr(w2d), r, c, _ptr(planes), _stream(w2d)), "smaat_split_planes")
        elif kind == 3:
            _lib.check(L.smaat_split_planes_h(_ptr(w2d), r, c, _ptr(planes), 1 if transpose else 0, _stream(w2d)),
                       "smaat_split_planes_h")
        else:
            _lib.check(L.smaat_bf16_planes(_ptr(w2d), r, c, _ptr(planes), 1 if transpose else 0, _stream(w2d)), "smaat_bf16_planes")
        return planes

    base = w2d._base if w2d._base is not None else w2d
    if (not policy.plane_cache or not w2d.is_contiguous() or base.dtype != torch.float32 or base.is_inference()
            or (w2d.is_cuda and torch.cuda.is_current_stream_capturing())):
        return direct()  # (inference tensors have no version counter; a capture must not bake in a cache decision)
    stream = _stream(w2d)
    mode = L.smaat_split_mode() if kind == 0 else -1
    key = (w2d.data_ptr(), r, c, bool(transpose), kind, mode, stream)
    e = _PLANES.get(key)
    if e is not None and e.ref() is base:
        if e.version == base._version:
            if not e.used:
                e.used = True
                return e.planes
            # second use of an image since its last refresh with (as far as the version counters say) unchanged weights --
            # gradient accumulation, repeated evaluation, or weights written through `.data`, which no counter sees.  THIS image
            # is refreshed by itself (one small launch) and nothing else is touched: a refresh of all images here would hand the
            # images of OTHER modules another unrefreshed use, and a `.data` write to one of those in between would be served
            # stale (A(x); B(x); A(x); B.w.data.add_(); B(x)).  The one-launch refresh below stays what the training
            # loop sees: an optimizer step bumps every version counter.
            direct(e.planes)
            return e.planes
    else:
        e = _PlaneEntry()
        e.ref, e.off, e.src = weakref.ref(base), w2d.data_ptr() - base.data_ptr(), w2d.data_ptr()
        e.version, e.r, e.c, e.kind, e.src_t = -1, r, c, kind, 1 if transpose else 0
        e.dev, e.stream, e.mode, e.used = w2d.device, stream, mode, False
        e.nblk = (r * cp + 255) // 256
        e.planes = torch.empty(shape, dtype=torch.int16, device=w2d.device)
        _PLANES[key] = e
    # refresh every stale image of this device / stream / split mode in one launch
    cur_mode = L.smaat_split_mode()
    stale, dead = [], []
    for k, x in _PLANES.items():
        b = x.ref()
        if b is None or b.data_ptr() + x.off != x.src:
            dead.append(k)
        elif x.dev == e.dev and x.stream == stream and (x.kind != 0 or x.mode == cur_mode) and x.version != b._version:
            stale.append((x, b._version))
    for k in dead:
        del _PLANES[k]
    ids = tuple(id(x) for x, _ in stale)
    table = _PLANES_TABLE.get(ids)
    if table is None:
        if len(_PLANES_TABLE) > 16:
            _PLANES_TABLE.clear()
        rows, b0, hp = [], 0, 0
        for x, _ in stale:
            rows.append([x.src, x.planes.data_ptr(), x.r, x.c, x.kind, x.src_t, b0, x.nblk])
            b0 += x.nblk
            if x.kind == 3:
                hp += L.smaat_split_planes_h_pieces(x.r, x.c)
        table = (torch.tensor(rows, dtype=torch.int64).to(e.dev), b0, [x for x, _ in stale], hp)  # (keeps the entries alive)
        _PLANES_TABLE[ids] = table
    if table[3] > 0:  # fp16 two-term images among them: their maxima first (a second launch), then all images
        _lib.check(L.smaat_weight_planes_multi_h(_ptr(table[0]), len(stale), table[1], table[3], stream),
                   "smaat_weight_planes_multi_h")
    else:
        _lib.check(L.smaat_weight_planes_multi(_ptr(table[0]), len(stale), table[1], stream), "smaat_weight_planes_multi")
    for x, v in stale:
        x.version, x.used = v, False
    e.used = True
    return e.planes


def _split_planes_raw(w2d, transpose=False):
    """w2d [R][C] f32 contiguous -> int16 bf16 planes, 3 * R * ceil16(C) elements (chunk-major [ceil16(C)/16][3][R][16]).
    transpose=True: the planes of w2d^T (w2d is [C][R]), read straight from w2d.  The result is a shared cached image
    (_weight_planes): read-only for the caller."""
    return _weight_planes(w2d, transpose, 0)


def _split_planes_h_raw(w2d, transpose=False):
    """w2d [R][C] f32 -> the fp16 two-term image of w2d * 2^kexp with its trailer (smaat_split_planes_h); transpose=True: of
    w2d^T (w2d stored [C][R]).  A shared cached image (_weight_planes): read-only for the caller."""
    return _weight_planes(w2d, transpose, 3)


def _pointwise_split_raw(x, planes, bias, m, want_stats=False, amax=None):
    """out[n][m][p] = sum_c A[m][c] x[n][c][p] + bias[m] with A given as split planes.  amax (int32 word holding the bit
    pattern of max |x|, written by the kernel that produced x): planes is then an fp16 image (_split_planes_h_raw) and the
    GEMM runs the two-term fp16 split"""
    L = _lib.get()
    x, x_bs = _planes(x)
    n, c, h, w = x.shape
    out = _new(x, n, m, h, w)
    part, slots = None, 0
    if want_stats:
        slots = L.smaat_pw_split_num_slots(n, h, w)
        part = _new(x, 3, slots, m)
    if amax is not None:
        _lib.check(L.smaat_pointwise_fwd_split_h(_ptr(x), x_bs, _ptr(amax), _ptr(planes), _ptr(bias), _ptr(out), m * h * w,
                                                 _ptr(part), n, c, m, h, w, _stream(x)), "smaat_pointwise_fwd_split_h")
        return out, part, slots
    _lib.check(L.smaat_pointwise_fwd_split(_ptr(x), x_bs, _ptr(planes), _ptr(bias), _ptr(out), m * h * w, _ptr(part),
                                           n, c, m, h, w, _stream(x)), "smaat_pointwise_fwd_split")
    return out, part, slots


def _dw3x3_fwd_raw(x, w_dw, b_dw, kpl, in_scale=None, in_shift=None, out_dtype=None, amx=None):
    """standalone depthwise 3x3 forward; None when the library does not handle the shape.  out_dtype: torch.float32
    (default: the dtype of x) or torch.bfloat16 (mixed precision: x may be f32 -- the stem -- or bf16).
    amx (f32 only): {"w": two amax buffers [y | dz], ...}; the kernel leaves max |y| in the first and amx["y"] = True, when the
    row-streaming kernel takes the shape"""
    L = _lib.get()
    x, x_bs = _planes(x)
    n, cin, h, w = x.shape
    k = cin * kpl
    out_dtype = out_dtype or x.dtype
    y = _new(x, n, k, h, w, dtype=out_dtype)
    if _is_bf(x) or out_dtype == BF16:
        rc = L.smaat_dw3x3_fwd_t(_ptr(x), _dt(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(y),
                                 _dt(y), k * h * w, n, cin, kpl, h, w, _stream(x))
        if rc == -2:
            raise NotImplementedError(f"depthwise 3x3 with bf16 storage: shape [{n},{cin},{h},{w}] kpl={kpl} not built")
        _lib.check(rc, "smaat_dw3x3_fwd_t")
        return y
    if amx is not None:
        rc = L.smaat_dw3x3_fwd_amax(_ptr(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(y), k * h * w,
                                    _ptr(amx["w"]), n, cin, kpl, h, w, _stream(x))
        if rc == 0:
            amx["y"] = True
            return y
        if rc != -2:
            _lib.check(rc, "smaat_dw3x3_fwd_amax")
    rc = L.smaat_dw3x3_fwd(_ptr(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(y), k * h * w,
                           n, cin, kpl, h, w, _stream(x))
    if rc == -2:
        return None
    _lib.check(rc, "smaat_dw3x3_fwd")
    return y


def _dsconv_fwd_split(x, w_dw, b_dw, w_pw, b_pw, kpl, want_stats, in_scale=None, in_shift=None, amx=None):
    """depthwise kernel + split pointwise GEMM; returns (z, part, slots, y) or None (unsupported shape).  amx: see
    _dw3x3_fwd_raw -- with the maximum of the depthwise output at hand the GEMM runs the two-term fp16 split"""
    y = _dw3x3_fwd_raw(x, w_dw, b_dw, kpl, in_scale, in_shift, amx=amx)
    if y is None:
        return None
    cout = w_pw.shape[0]
    ay = amx["w"][:AMAX_WORDS] if (amx is not None and amx.get("y")) else None
    planes = _split_planes_h_raw(w_pw.reshape(cout, -1)) if ay is not None else _split_planes_raw(w_pw.reshape(cout, -1))
    L = _lib.get()
    n, k, h, w = y.shape
    # layers that leave the chip under-filled (18 x 18 at batch 32: 384 serial chains of 64 chunks): the contraction is cut
    # into slices that run as virtual images; the slice reduction writes z and its BatchNorm partials
    s_k = L.smaat_pointwise_splitk_slices(n, k, cout, h, w, policy.splitk_train_budget) if policy.splitk_train_budget > 0 else 1
    if s_k > 1:
        z = _new(y, n, cout, h, w)
        ws = _new(y, n * s_k * cout * h * w)
        part, slots = None, 0
        if want_stats:
            slots = L.smaat_pw_split_num_slots(n, h, w)
            part = _new(y, 3, slots, cout)
        if ay is not None:
            rc = L.smaat_pointwise_fwd_split_k_h(_ptr(y), k * h * w, _ptr(ay), _ptr(planes), _ptr(b_pw), _ptr(z), cout * h * w,
                                                 _ptr(part), _ptr(ws), s_k, n, k, cout, h, w, _stream(y))
        else:
            rc = L.smaat_pointwise_fwd_split_k(_ptr(y), k * h * w, _ptr(planes), _ptr(b_pw), _ptr(z), cout * h * w, _ptr(part),
                                               _ptr(ws), s_k, n, k, cout, h, w, 0, _stream(y))
        if rc == 0:
            return z, part, slots, y
        if rc != -2:
            _lib.check(rc, "smaat_pointwise_fwd_split_k")
    z, part, slots = _pointwise_split_raw(y, planes, b_pw, cout, want_stats, amax=ay)
    return z, part, slots, y


# ---- mixed precision (bf16 storage): depthwise kernel (bf16 out) + bf16 GEMM fed by LDS-DMA (csrc/bf16gemm.hip) ----
def _bf16_planes_raw(w2d, transpose=False):
    """w2d [R][C] f32 -> bf16 image [ceil32(C)/16][R][16]; transpose=True: the image of w2d^T (w2d stored [C][R]).  A shared
    cached image (_weight_planes): read-only for the caller."""
    return _weight_planes(w2d, transpose, 2)


def _pointwise_bf16_raw(x, planes, bias, m, want_stats=False, out_dtype=BF16, relu=False):
    """out[n][m][p] = sum_c A[m][c] x[n][c][p] + bias[m]; x bf16, A a bf16 image, f32 accumulation, out bf16 | f32"""
    L = _lib.get()
    x, x_bs = _planes(x)
    assert x.dtype == BF16
    n, c, h, w = x.shape
    out = _new(x, n, m, h, w, dtype=out_dtype)
    part, slots = None, 0
    if want_stats:
        slots = L.smaat_pw_split_num_slots(n, h, w)
        part = _new(x, 3, slots, m)
    rc = L.smaat_pointwise_fwd_bf16(_ptr(x), x_bs, _ptr(planes), _ptr(bias), _ptr(out), m * h * w, _dt(out), _ptr(part), n, c,
                                    m, h, w, 1 if relu else 0, _stream(x))
    if rc == -2:
        raise NotImplementedError(f"bf16 pointwise GEMM: [{n},{c},{h},{w}] -> {m} channels not built (odd plane size?)")
    _lib.check(rc, "smaat_pointwise_fwd_bf16")
    return out, part, slots


def _dsconv_fwd_bf16(x, w_dw, b_dw, w_pw, b_pw, kpl, want_stats, in_scale=None, in_shift=None):
    """mixed-precision DepthwiseSeparableConv forward: (z bf16, part, slots, y_dw bf16); x f32 (stem) or bf16"""
    y = _dw3x3_fwd_raw(x, w_dw, b_dw, kpl, in_scale, in_shift, out_dtype=BF16)
    cout = w_pw.shape[0]
    planes = _bf16_planes_raw(w_pw.reshape(cout, -1))
    z, part, slots = _pointwise_bf16_raw(y, planes, b_pw, cout, want_stats)
    return z, part, slots, y


# Fused depthwise -> split GEMM forward (csrc/dsconv_split.hip): the 2x-expanded depthwise tensor stays in LDS.
# Measured (profiles/r2/layer_bench_fused.txt, batch 32): 1.3-1.4x faster than the depthwise kernel + GEMM pair on the
# 288^2 layers when the depthwise output is NOT needed afterwards, but slower than the pair when it has to be written
# as a side product for the streamed weight gradient (the write is 2/3 of the traffic it saves).  Policy:
#   "auto" = use it where the depthwise output is not kept (forward under no_grad, eval-mode forward) on the
#            plane-dominated layers (Cout <= 128: one or two 64-channel tiles re-run the cheap depthwise stage);
#   "train" = also in training (side output written); "all" = every supported shape; "off".
# policy.fuse_dw_split (default os.environ.get("SMAAT_FUSE_DW", "auto"))


# Round 4: weight gradient with the depthwise output recomputed by the GEMM's producer waves (csrc/dswgrad.hip).  Where
# it applies, training runs the fused forward WITHOUT the side output and keeps no depthwise tensor at all: the
# 2x-expanded tensor is neither written (forward) nor read (weight gradient).  "auto" = the measured policy
# (profiles/r4), "all" = every shape the kernels take, "off" = the round-3 behaviour.
# policy.wgrad_recompute (default os.environ.get("SMAAT_WGRAD_RECOMPUTE", "auto"))
# policy.bf16_recompute: the same policy under bf16 storage (A/B switch)


def _recompute_wgrad_ok(n, cin, h, w, kpl, cout):
    if policy.wgrad_recompute == "off" or policy.fuse_dw_split == "off" or not _split_on():
        return False
    from . import train_ops
    if train_ops.active():  # (the traceable operators declare their saved tensors up front: kept depthwise outputs)
        return False
    L = _lib.get()
    if not L.smaat_dsconv_wgrad_split_ok(kpl, cout, h, w) or L.smaat_dsconv_split_num_slots(n, h, w) <= 0:
        return False
    if policy.wgrad_recompute == "all":
        return True
    # a 64-channel K tile per workgroup: layers with fewer input channels (the 12-channel stem) would idle most producer
    # threads; plane-dominated layers only (the deep layers are not HBM-bound in f32)
    return cin >= 32 and h * w >= 16384


def _recompute_operand_ok(x):
    """what smaat_dsconv_wgrad_split(_t) asks of x beyond the shape (it returns -2 otherwise, and by then -- in the backward --
    nothing has been kept to fall back on): 16-byte aligned rows of a dense-plane tensor, a batch stride that is a
    multiple of 4 elements, one image below 4 GiB.  The network's own buffers always qualify; an offset or strided view
    handed in by a caller may not, and then the forward keeps the depthwise output as it did before round 4."""
    xx, x_bs = _planes(x)
    return (xx.data_ptr() % 16 == 0 and x_bs % 4 == 0 and xx.shape[1] * xx.shape[2] * xx.shape[3] * xx.element_size() < (1 << 32))


def _fused_dw_ok(n, h, w, kpl, cout, keep_y, cin=0):
    if policy.fuse_dw_split == "off" or kpl != 2 or not _split_on():
        return False
    if _lib.get().smaat_dsconv_split_num_slots(n, h, w) <= 0:
        return False
    if policy.fuse_dw_split == "all":
        return True
    if not keep_y or policy.fuse_dw_split == "train":
        return cout <= 128
    # training keeps the depthwise tensor for the weight gradient: the kernel then writes it as a side output.
    # Measured per layer on MI355X (profiles/r2/layer_bench_r2v.txt): that wins where one 64-row output tile covers
    # Cout and the reduction is long enough to amortise the tile prologue (288^2, K = 256: 1.52 vs 1.73 ms), is a tie
    # at K = 128 (0.99 vs 0.97 ms) and loses where the depthwise stage is recomputed per output tile (Cout = 128).
    return cout <= 64 and cin * kpl >= 256 and h * w >= 65536


def _dsconv_fwd_fused(x, w_dw, b_dw, w_pw, b_pw, kpl, want_stats, in_scale=None, in_shift=None, want_y=False):
    """fused depthwise + split pointwise GEMM; returns (z, part, slots, y) or None (unsupported shape)"""
    L = _lib.get()
    x, x_bs = _planes(x)
    n, cin, h, w = x.shape
    cout = w_pw.shape[0]
    k = cin * kpl
    slots = L.smaat_dsconv_split_num_slots(n, h, w)
    if slots <= 0:
        return None
    planes = _split_planes_raw(w_pw.reshape(cout, -1))
    z = _new(x, n, cout, h, w)
    part = _new(x, 3, slots, cout) if want_stats else None
    y = _new(x, n, k, h, w) if want_y else None
    rc = L.smaat_dsconv_fwd_split(_ptr(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(planes),
                                  _ptr(b_pw), _ptr(z), cout * h * w, _ptr(part), _ptr(y), n, cin, kpl, cout, h, w,
                                  _stream(x))
    if rc == -2:
        return None
    _lib.check(rc, "smaat_dsconv_fwd_split")
    return z, part, (slots if want_stats else 0), y


# Row-walking fused forward (csrc/dsrows.hip, round 4): the depthwise window in the producer threads' registers, the
# pointwise weight's MFMA fragments in the consumer waves' registers; no depthwise side output (the weight gradient
# recomputes it).  "auto": wherever the kernel takes the shape and nothing has to be kept; "off": the tile kernel.
# policy.fwd_rows (default os.environ.get("SMAAT_FWD_ROWS", "auto"))


def _dsconv_fwd_rows(x, w_dw, b_dw, w_pw, b_pw, kpl, want_stats, in_scale=None, in_shift=None, out_dtype=None, amx=None,
                     xb=None):
    """-> (z, part, slots, None) or None when the row-walking kernel does not take the shape.  x f32 -> z f32 (split planes);
    out_dtype = torch.bfloat16: mixed precision (x f32 | bf16, bf16 weight image, z bf16).
    amx (f32 storage; see _half_forward): the kernel also leaves max |y| of the depthwise output it forms -- never stored -- in
    the first amax buffer, for the two-term fp16 recompute weight gradient of the backward.
    xb (with amx; round 6): a bound of |x| -- {"amax": buffer, "amax2": buffer | None} (maxima of x itself) or {"amax": buffer of
    max |u|, "prev_w": [Cin][K'] weight, "prev_b": bias | None} (x = prev_w . u + prev_b) -- with which the GEMM itself runs the
    two-term fp16 split (smaat_dsconv_fwd_rows_h)"""
    if policy.fwd_rows == "off":
        return None
    L = _lib.get()
    x, x_bs = _planes(x)
    n, cin, h, w = x.shape
    cout = w_pw.shape[0]
    if not L.smaat_dsconv_rows_ok(kpl, cin, cout, h, w):
        return None
    out_dtype = out_dtype or x.dtype
    # (K = 256 with the exact three-term split -- up4.0 -- used to stay on the tile kernel: 96 registers of resident weight
    # fragments per consumer wave spilled.  Its third weight plane now lives in LDS: 0.83 vs 1.09 ms, profiles/r4.)
    slots = L.smaat_dsconv_rows_num_slots(n, h, w)
    z = _new(x, n, cout, h, w, dtype=out_dtype)
    part = _new(x, 3, slots, cout) if want_stats else None
    if (amx is not None and amx.get("w") is not None and x.dtype == F32 and out_dtype == F32 and xb is not None
            and policy.fwd_rows_h and kpl == 2):
        pw = xb.get("prev_w")
        rc = L.smaat_dsconv_fwd_rows_h(_ptr(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(xb["amax"]),
                                       _ptr(xb.get("amax2")), _ptr(pw), _ptr(xb.get("prev_b")) if pw is not None else None,
                                       pw.shape[1] if pw is not None else 0, _ptr(_split_planes_h_raw(w_pw.reshape(cout, -1))),
                                       _ptr(b_pw), _ptr(z), cout * h * w, _ptr(part), _ptr(amx["w"][:AMAX_WORDS]), None, n, cin, kpl,
                                       cout, h, w, _stream(x))
        if rc == 0:
            amx["y"] = True
            return z, part, (slots if want_stats else 0), None
        if rc != -2:
            _lib.check(rc, "smaat_dsconv_fwd_rows_h")
    # (the three-term / bf16 image is asked for only here: an image that was requested once is refreshed after every weight update)
    planes = (_bf16_planes_raw(w_pw.reshape(cout, -1)) if out_dtype == BF16 else _split_planes_raw(w_pw.reshape(cout, -1)))
    if amx is not None and amx.get("w") is not None and x.dtype == F32 and out_dtype == F32:
        rc = L.smaat_dsconv_fwd_rows_amax(_ptr(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(planes),
                                          _ptr(b_pw), _ptr(z), cout * h * w, _ptr(part), _ptr(amx["w"][:AMAX_WORDS]), n, cin, kpl,
                                          cout, h, w, _stream(x))
        if rc == 0:
            amx["y"] = True
            return z, part, (slots if want_stats else 0), None
        if rc != -2:
            _lib.check(rc, "smaat_dsconv_fwd_rows_amax")
        return None
    rc = L.smaat_dsconv_fwd_rows(_ptr(x), _dt(x), x_bs, _ptr(in_scale), _ptr(in_shift), _ptr(w_dw), _ptr(b_dw), _ptr(planes),
                                 _ptr(b_pw), _ptr(z), _dt(z), cout * h * w, _ptr(part), n, cin, kpl, cout, h, w, _stream(x))
    if rc == -2:
        return None
    _lib.check(rc, "smaat_dsconv_fwd_rows")
    return z, part, (slots if want_stats else 0), None


def _bn_finalize_raw(part, slots, c, count, bias_shift, gamma, beta, eps, momentum, rm, rv):
    L = _lib.get()
    st = _new(part, 4, c)
    _lib.check(L.smaat_bn_finalize(_ptr(part), slots, c, float(count), _ptr(bias_shift), _ptr(gamma), _ptr(beta),
                                   float(eps), float(momentum), _ptr(rm), _ptr(rv), _ptr(st[0]), _ptr(st[1]),
                                   _ptr(st[2]), _ptr(st[3]), _stream(part)), "smaat_bn_finalize")
    return st  # rows: mean, invstd, scale, shift


def _bn_eval_coefs_raw(rm, rv, gamma, beta, eps):
    """st rows (mean, invstd, scale, shift) from the running statistics: one launch"""
    L = _lib.get()
    c = rm.numel()
    st = _new(rm, 4, c)
    _lib.check(L.smaat_bn_eval_coefs(_ptr(rm), _ptr(rv), _ptr(gamma), _ptr(beta), float(eps), c, _ptr(st), _stream(rm)),
               "smaat_bn_eval_coefs")
    return st


def _affine_act_raw(z, scale, shift, relu, out=None):
    L = _lib.get()
    z, z_bs = _planes(z)
    n, c, h, w = z.shape
    if out is None:
        out = _new(z, n, c, h, w, dtype=z.dtype)
    out_t, o_bs = _planes(out)
    assert out_t is out
    if _is_bf(z) or _is_bf(out):
        _lib.check(L.smaat_affine_act_t(_ptr(z), _dt(z), z_bs, _ptr(scale), _ptr(shift), _ptr(out), _dt(out), o_bs, n, c,
                                        h * w, 1 if relu else 0, _stream(z)), "smaat_affine_act_t")
        return out
    _lib.check(L.smaat_affine_act(_ptr(z), z_bs, _ptr(scale), _ptr(shift), _ptr(out), o_bs, n, c, h * w,
                                  1 if relu else 0, _stream(z)), "smaat_affine_act")
    return out


def _bn_bwd_raw(dy, z, st, gamma, relu, train, pre_part=None, amax=None):
    """dz, dgamma, dbeta for y = relu?(bn(z)).  st rows: mean, invstd, scale, shift.
    pre_part = (part [2][slots][C], slots): the reduction already produced by smaat_dw3x3_bwd_bnred.
    amax (f32 storage only): int32 word that receives the bit pattern of max |dz| (two-term fp16 split of the consumers)"""
    L = _lib.get()
    dy, dy_bs = _planes(dy)
    z, z_bs = _planes(z)
    n, c, h, w = z.shape
    p = h * w
    s = _stream(z)
    typed = _is_bf(z) or _is_bf(dy)
    if typed and dy.dtype != z.dtype:  # (a gradient that autograd handed over in another dtype)
        dy = dy.to(z.dtype)
    if pre_part is not None:
        part, slots = pre_part
    else:
        slots = L.smaat_plane_num_slots(n, p)
        part = _new(z, 2, slots, c)
        if typed:
            _lib.check(L.smaat_bn_bwd_reduce_t(_ptr(dy), _dt(dy), dy_bs, _ptr(z), _dt(z), z_bs, _ptr(st[2]), _ptr(st[3]),
                                               _ptr(st[0]), _ptr(st[1]), _ptr(part), n, c, p, 1 if relu else 0, None, s),
                       "smaat_bn_bwd_reduce_t")
        else:
            _lib.check(L.smaat_bn_bwd_reduce(_ptr(dy), dy_bs, _ptr(z), z_bs, _ptr(st[2]), _ptr(st[3]), _ptr(st[0]),
                                             _ptr(st[1]), _ptr(part), n, c, p, 1 if relu else 0, s),
                       "smaat_bn_bwd_reduce")
    dgamma = _new(z, c)
    dbeta = _new(z, c)
    coef = _new(z, 3, c)
    _lib.check(L.smaat_bn_bwd_finalize(_ptr(part), slots, c, float(n * p), _ptr(gamma), _ptr(st[1]), _ptr(dgamma),
                                       _ptr(dbeta), _ptr(coef), s), "smaat_bn_bwd_finalize")
    if not train:  # eval mode: statistics are constants -> no mean/var terms
        coef[1:].zero_()
    dz = _new(z, n, c, h, w, dtype=z.dtype)
    if typed:
        _lib.check(L.smaat_bn_bwd_apply_t(_ptr(dy), _dt(dy), dy_bs, _ptr(z), _dt(z), z_bs, _ptr(st[2]), _ptr(st[3]),
                                          _ptr(st[0]), _ptr(st[1]), _ptr(coef), _ptr(dz), _dt(dz), c * p, n, c, p,
                                          1 if relu else 0, None, s), "smaat_bn_bwd_apply_t")
        return dz, dgamma, dbeta
    if amax is not None:
        _lib.check(L.smaat_bn_bwd_apply_amax(_ptr(dy), dy_bs, None, _ptr(z), z_bs, _ptr(st[2]), _ptr(st[3]), _ptr(st[0]),
                                             _ptr(st[1]), _ptr(coef), _ptr(dz), c * p, _ptr(amax), n, c, p, 1 if relu else 0, s),
                   "smaat_bn_bwd_apply_amax")
        return dz, dgamma, dbeta
    _lib.check(L.smaat_bn_bwd_apply(_ptr(dy), dy_bs, _ptr(z), z_bs, _ptr(st[2]), _ptr(st[3]), _ptr(st[0]),
                                    _ptr(st[1]), _ptr(coef), _ptr(dz), c * p, n, c, p, 1 if relu else 0, s),
               "smaat_bn_bwd_apply")
    return dz, dgamma, dbeta


def _bn_bwd_head_raw(dlog, w_out, z, st, gamma, train, amax=None):
    """BatchNorm + ReLU backward when the consumer of y = relu(bn(z)) is a 1x1 convolution to ONE channel whose output
    gradient is dlog [N][1][H][W]: dy = w_out[c] * dlog is formed on the fly (smaat_bn_bwd_*_head), never stored.
    -> dz, dgamma, dbeta, dw_out [1][C][1][1]"""
    L = _lib.get()
    dlog, dl_bs = _planes(dlog)
    z, z_bs = _planes(z)
    n, c, h, w = z.shape
    p = h * w
    s = _stream(z)
    wv = w_out.reshape(-1).contiguous()
    slots = L.smaat_plane_num_slots(n, p)
    part = _new(z, 3, slots, c)
    typed = _is_bf(z)
    if typed:
        dlog = dlog.float() if dlog.dtype != F32 else dlog
        _lib.check(L.smaat_bn_bwd_reduce_t(_ptr(dlog), 0, dl_bs, _ptr(z), _dt(z), z_bs, _ptr(st[2]), _ptr(st[3]), _ptr(st[0]),
                                           _ptr(st[1]), _ptr(part), n, c, p, 1, _ptr(wv), s), "smaat_bn_bwd_reduce_t(head)")
    else:
        _lib.check(L.smaat_bn_bwd_reduce_head(_ptr(dlog), dl_bs, _ptr(wv), _ptr(z), z_bs, _ptr(st[2]), _ptr(st[3]),
                                              _ptr(st[0]), _ptr(st[1]), _ptr(part), n, c, p, s), "smaat_bn_bwd_reduce_head")
    dgamma, dbeta, coef = _new(z, c), _new(z, c), _new(z, 3, c)
    _lib.check(L.smaat_bn_bwd_finalize(_ptr(part), slots, c, float(n * p), _ptr(gamma), _ptr(st[1]), _ptr(dgamma),
                                       _ptr(dbeta), _ptr(coef), s), "smaat_bn_bwd_finalize")
    if not train:
        coef[1:].zero_()
    dz = _new(z, n, c, h, w, dtype=z.dtype)
    if typed:
        _lib.check(L.smaat_bn_bwd_apply_t(_ptr(dlog), 0, dl_bs, _ptr(z), _dt(z), z_bs, _ptr(st[2]), _ptr(st[3]), _ptr(st[0]),
                                          _ptr(st[1]), _ptr(coef), _ptr(dz), _dt(dz), c * p, n, c, p, 1, _ptr(wv), s),
                   "smaat_bn_bwd_apply_t(head)")
    elif amax is not None:
        _lib.check(L.smaat_bn_bwd_apply_amax(_ptr(dlog), dl_bs, _ptr(wv), _ptr(z), z_bs, _ptr(st[2]), _ptr(st[3]), _ptr(st[0]),
                                             _ptr(st[1]), _ptr(coef), _ptr(dz), c * p, _ptr(amax), n, c, p, 1, s),
                   "smaat_bn_bwd_apply_amax(head)")
    else:
        _lib.check(L.smaat_bn_bwd_apply_head(_ptr(dlog), dl_bs, _ptr(wv), _ptr(z), z_bs, _ptr(st[2]), _ptr(st[3]),
                                             _ptr(st[0]), _ptr(st[1]), _ptr(coef), _ptr(dz), c * p, n, c, p, s),
                   "smaat_bn_bwd_apply_head")
    dw_out = _new(z, c)
    _lib.check(L.smaat_reduce_rows(_ptr(part[2]), slots, c, _ptr(dw_out), 1.0, s), "smaat_reduce_rows")
    return dz, dgamma, dbeta, dw_out.view(1, c, 1, 1)


def _outconv1_fwd_raw(z, scale, shift, w_out, b_out):
    """logits [N][1][H][W] = OutConv(relu(z * scale + shift)) for ONE output channel, the activation never written"""
    L = _lib.get()
    z, z_bs = _planes(z)
    n, c, h, w = z.shape
    out = _new(z, n, 1, h, w)
    wv = w_out.reshape(-1).contiguous()
    if _is_bf(z):
        _lib.check(L.smaat_outconv1_fwd_t(_ptr(z), _dt(z), z_bs, _ptr(scale), _ptr(shift), _ptr(wv), _ptr(b_out), _ptr(out),
                                          h * w, n, c, h * w, _stream(z)), "smaat_outconv1_fwd_t")
        return out
    _lib.check(L.smaat_outconv1_fwd(_ptr(z), z_bs, _ptr(scale), _ptr(shift), _ptr(wv), _ptr(b_out), _ptr(out), h * w, n, c,
                                    h * w, _stream(z)), "smaat_outconv1_fwd")
    return out


def _channel_sum_raw(x):
    L = _lib.get()
    x, x_bs = _planes(x)
    n, c, h, w = x.shape
    ws = _new(x, L.smaat_plane_num_slots(n, h * w), c)
    out = _new(x, c)
    if _is_bf(x):
        _lib.check(L.smaat_channel_sum_t(_ptr(x), _dt(x), x_bs, n, c, h * w, _ptr(ws), _ptr(out), _stream(x)),
                   "smaat_channel_sum_t")
        return out
    _lib.check(L.smaat_channel_sum(_ptr(x), x_bs, n, c, h * w, _ptr(ws), _ptr(out), _stream(x)), "smaat_channel_sum")
    return out


def _pointwise_wgrad_raw(y, dz, m, amax_y=None, amax_dz=None):
    """dW[m][k] = sum_{n,p} dz[n][m][p] * y[n][k][p]  -> tensor [m][k][1][1].  amax_y / amax_dz (both or neither; f32
    storage): the operands' maximum words -> two-term fp16 split"""
    L = _lib.get()
    y, y_bs = _planes(y)
    dz, dz_bs = _planes(dz)
    n, k, h, w = y.shape
    ns = L.smaat_wgrad_num_splits(n, h, w, m, k)
    ws = _new(y, ns, m, k)
    dw = _new(y, m, k, 1, 1)
    if _is_bf(y) or _is_bf(dz):  # mixed precision: bf16 operands straight from HBM, f32 accumulation and result
        if y.dtype != BF16:
            y, y_bs = _planes(y.to(BF16))
        if dz.dtype != BF16:
            dz, dz_bs = _planes(dz.to(BF16))
        rc = L.smaat_pointwise_wgrad_bf16(_ptr(y), y_bs, _ptr(dz), dz_bs, _ptr(ws), _ptr(dw), n, k, m, h, w, _stream(y))
        if rc == -2:
            raise NotImplementedError(f"bf16 pointwise weight gradient: [{n},{k},{h},{w}] x {m} not built (odd plane size?)")
        _lib.check(rc, "smaat_pointwise_wgrad_bf16")
        return dw
    if amax_y is not None and amax_dz is not None:
        _lib.check(L.smaat_pointwise_wgrad_h(_ptr(y), y_bs, _ptr(amax_y), _ptr(dz), dz_bs, _ptr(amax_dz), _ptr(ws), _ptr(dw), n,
                                             k, m, h, w, _stream(y)), "smaat_pointwise_wgrad_h")
        return dw
    _lib.check(L.smaat_pointwise_wgrad(_ptr(y), y_bs, _ptr(dz), dz_bs, _ptr(ws), _ptr(dw), n, k, m, h, w,
                                       _stream(y)), "smaat_pointwise_wgrad")
    return dw


def _dsconv_bwd_raw(x, w_dw, b_dw, w_pw, dz, kpl, need_dx, y=None, bnred=None, in_aff=None, amx=None):
    """gradients of z = pointwise(depthwise(x)) given dz: dx, dw_dw, db_dw, dw_pw.
    y: the depthwise output kept by the forward (streamed weight gradient); when None the
    memory-lean kernel recomputes it from x."""
    L = _lib.get()
    x, x_bs = _planes(x)
    dz, dz_bs = _planes(dz)
    n, cin, h, w = x.shape
    cout = w_pw.shape[0]
    k = cin * kpl
    s = _stream(x)
    if _is_bf(dz):
        return _dsconv_bwd_bf16(x, x_bs, w_dw, b_dw, w_pw, dz, kpl, need_dx, y, bnred, in_aff)
    if _is_bf(x):  # the forward of this block fell back to f32 storage (_bf16_storage_ok) on a bf16 input
        x, x_bs = _planes(x.float())
    # amx = {"w": int32 words [max |y|, max |dz|], "y": bool, "dz": bool}: which maxima the producing kernels left
    # ("wdz": the words THIS backward pass filled -- _half_backward hands every pass fresh zero words)
    a_dz = amx.get("wdz", amx["w"][AMAX_WORDS:]) if (amx is not None and amx.get("dz")) else None
    a_y = amx["w"][:AMAX_WORDS] if (amx is not None and amx.get("y")) else None
    if y is not None:
        dw_pw = _pointwise_wgrad_raw(y, dz, cout, a_y if a_dz is not None else None, a_dz if a_y is not None else None)
    else:
        # no depthwise tensor was kept: it is recomputed from x (with the previous activation applied on load) inside
        # the weight-gradient kernel -- on the split matrix path where that kernel takes the shape (the training policy,
        # _recompute_wgrad_ok), else by the memory-lean f32-MFMA kernel (ops.KEEP_DEPTHWISE_OUTPUT = False)
        isc, ish = in_aff if in_aff is not None else (None, None)
        dw_pw = _new(x, cout, k, 1, 1)
        rc = -2
        if _split_on() and policy.wgrad_recompute != "off" and L.smaat_dsconv_wgrad_split_ok(kpl, cout, h, w):
            ws = _new(x, L.smaat_dsconv_wgrad_split_num_splits(n, cin, cout, h, w), cout, k)
            if a_y is not None and a_dz is not None:  # the forward left max |y|, the BatchNorm apply max |dz|: fp16 split
                rc = L.smaat_dsconv_wgrad_split_h(_ptr(x), x_bs, _ptr(isc), _ptr(ish), _ptr(w_dw), _ptr(b_dw), _ptr(a_y), _ptr(dz),
                                                  dz_bs, _ptr(a_dz), _ptr(ws), _ptr(dw_pw), n, cin, kpl, cout, h, w, s)
            else:
                rc = L.smaat_dsconv_wgrad_split(_ptr(x), x_bs, _ptr(isc), _ptr(ish), _ptr(w_dw), _ptr(b_dw), _ptr(dz), dz_bs,
                                                _ptr(ws), _ptr(dw_pw), n, cin, kpl, cout, h, w, s)
            if rc != -2:
                _lib.check(rc, "smaat_dsconv_wgrad_split")
        if rc == -2:
            ns = L.smaat_dsconv_wgrad_num_splits(n, h, w, cout, k)
            ws = _new(x, ns, cout, k)
            _lib.check(L.smaat_dsconv_wgrad(_ptr(x), x_bs, _ptr(isc), _ptr(ish), _ptr(w_dw), _ptr(b_dw), _ptr(dz), dz_bs,
                                            _ptr(ws), _ptr(dw_pw), n, cin, kpl, cout, h, w, s), "smaat_dsconv_wgrad")
        del ws
    # round 6: dgrad GEMM + depthwise backward in ONE kernel where it takes the shape (the 288^2 layers: Cout = 64, Cin 64 | 128):
    # dY = W^T dZ is formed row by row on chip and never written (include/smaat_hip.h "fused BACKWARD"); dx bit-identical
    if (policy.fused_bwd and a_dz is not None and need_dx and _split_dgrad_ok(cout, k) and (in_aff is None or bnred is not None)
            and L.smaat_dsconv_bwd_rows_ok(kpl, cin, cout, h, w)):
        planes_t = _split_planes_h_raw(w_pw.reshape(cout, k), transpose=True)
        rows = L.smaat_dsconv_bwd_rows_num_rows(n, cin, h, w)
        dx = _new(x, n, cin, h, w)
        ws2 = _new(x, rows, k, 10)
        dw_dw = _new(x, k, 1, 3, 3)
        db_dw = _new(x, k)
        isc, ish = in_aff if in_aff is not None else (None, None)
        mean, invstd = bnred if in_aff is not None else (None, None)
        rpart = _new(x, 2, rows, cin) if in_aff is not None else None
        rc = L.smaat_dsconv_bwd_rows_h(_ptr(x), x_bs, _ptr(isc), _ptr(ish), _ptr(mean), _ptr(invstd), _ptr(dz), dz_bs, _ptr(a_dz),
                                       _ptr(planes_t), _ptr(w_dw), _ptr(dx), cin * h * w, _ptr(ws2), _ptr(dw_dw), _ptr(db_dw),
                                       _ptr(rpart), n, cin, kpl, cout, h, w, s)
        if rc != -2:
            _lib.check(rc, "smaat_dsconv_bwd_rows_h")
            if in_aff is not None:
                return dx, dw_dw, db_dw, dw_pw, (rpart, rows)
            if bnred is not None:
                return dx, dw_dw, db_dw, dw_pw, None
            return dx, dw_dw, db_dw, dw_pw
    # data gradient of the pointwise conv: dY = W^T dZ  (wt := w_pw in its natural [Cout][K] layout)
    if _split_dgrad_ok(cout, k):
        # A[m' = k][c = co] = w_pw[co][k]: planes of the transposed weight
        if a_dz is not None:
            planes_t = _split_planes_h_raw(w_pw.reshape(cout, k), transpose=True)
        else:
            planes_t = _split_planes_raw(w_pw.reshape(cout, k), transpose=True)
        dy, _, _ = _pointwise_split_raw(dz, planes_t, None, k, amax=a_dz)
    else:
        dy = _new(x, n, k, h, w)
        _lib.check(L.smaat_pointwise_fwd(_ptr(dz), dz_bs, _ptr(w_pw), None, _ptr(dy), k * h * w, None, n, cout, k, h,
                                         w, s), "smaat_pointwise_fwd(dgrad)")
    # depthwise backward
    dx = _new(x, n, cin, h, w) if need_dx else None
    ws2 = _new(x, L.smaat_dw3x3_bwd_ws_rows(n, cin, h, w), k, 10)
    dw_dw = _new(x, k, 1, 3, 3)
    db_dw = _new(x, k)
    red = None
    if in_aff is not None:
        # x is the PRE-BatchNorm tensor of the previous half (its activation is applied on load): the strip kernel
        # also reduces that BatchNorm's backward sums.  The forward only leaves the activation unmaterialised
        # when smaat_dw3x3_strip_ok says this kernel takes the shape, so -2 here is a host/library mismatch.
        if bnred is None or not need_dx:
            raise _lib.SmaatHipError("depthwise backward with an on-load activation needs bnred=(mean, invstd) and dx")
        mean, invstd = bnred
        isc, ish = in_aff
        rows = L.smaat_dw3x3_bwd_ws_rows(n, cin, h, w) - 1
        rpart = _new(x, 2, rows, cin)
        rc = L.smaat_dw3x3_bwd_bnred(_ptr(x), x_bs, _ptr(isc), _ptr(ish), _ptr(dy), k * h * w, _ptr(w_dw), _ptr(dx),
                                     cin * h * w, _ptr(ws2), _ptr(dw_dw), _ptr(db_dw), _ptr(mean), _ptr(invstd),
                                     _ptr(rpart), n, cin, kpl, h, w, s)
        if rc == -2:
            raise _lib.SmaatHipError(
                f"smaat_dw3x3_bwd_bnred does not take [{n},{cin},{h},{w}] kpl={kpl} although smaat_dw3x3_strip_ok did "
                "when the forward ran (SMAAT_DWB_STRIP changed between forward and backward?)")
        _lib.check(rc, "smaat_dw3x3_bwd_bnred")
        return dx, dw_dw, db_dw, dw_pw, (rpart, rows)
    _lib.check(L.smaat_dw3x3_bwd(_ptr(x), x_bs, _ptr(dy), k * h * w, _ptr(w_dw), _ptr(dx), cin * h * w, _ptr(ws2),
                                 _ptr(dw_dw), _ptr(db_dw), n, cin, kpl, h, w, s), "smaat_dw3x3_bwd")
    if bnred is not None:
        return dx, dw_dw, db_dw, dw_pw, red
    return dx, dw_dw, db_dw, dw_pw


def _dsconv_bwd_bf16(x, x_bs, w_dw, b_dw, w_pw, dz, kpl, need_dx, y, bnred, in_aff):
    """mixed-precision form of _dsconv_bwd_raw: dz, y (the kept depthwise output) and the depthwise-output gradient are
    bf16; x is bf16, or f32 for the stem (then dx, if wanted, is f32 too).  Same return convention."""
    L = _lib.get()
    n, cin, h, w = x.shape
    cout = w_pw.shape[0]
    k = cin * kpl
    s = _stream(x)
    if y is None:
        # nothing was kept (the training policy of the plane-dominated layers, _recompute_wgrad_ok): the typed
        # weight-gradient kernel recomputes the depthwise output from x with the previous activation applied on load
        if not L.smaat_dsconv_wgrad_split_ok(kpl, cout, h, w):
            raise _lib.SmaatHipError("mixed precision keeps the depthwise output for the weight gradient except where "
                                     "smaat_dsconv_wgrad_split_t takes the shape (ops.KEEP_DEPTHWISE_OUTPUT = False is an "
                                     "f32-only option)")
        isc_, ish_ = in_aff if in_aff is not None else (None, None)
        dw_pw = _new(x, cout, k, 1, 1)
        ws = _new(x, L.smaat_dsconv_wgrad_split_num_splits(n, cin, cout, h, w), cout, k)
        dzp, dz_bs = _planes(dz)
        _lib.check(L.smaat_dsconv_wgrad_split_t(_ptr(x), _dt(x), x_bs, _ptr(isc_), _ptr(ish_), _ptr(w_dw), _ptr(b_dw), _ptr(dzp), _dt(dzp),
                                                dz_bs, _ptr(ws), _ptr(dw_pw), n, cin, kpl, cout, h, w, s),
                   "smaat_dsconv_wgrad_split_t")
        del ws
    else:
        dw_pw = _pointwise_wgrad_raw(y, dz, cout)
    planes_t = _bf16_planes_raw(w_pw.reshape(cout, k), transpose=True)  # A[k][co] = w_pw[co][k]
    dy, _, _ = _pointwise_bf16_raw(dz, planes_t, None, k)
    dx = _new(x, n, cin, h, w, dtype=x.dtype) if need_dx else None
    rows = L.smaat_dw3x3_bwd_ws_rows(n, cin, h, w)
    ws2 = _new(x, rows, k, 10)
    dw_dw = _new(x, k, 1, 3, 3)
    db_dw = _new(x, k)
    rpart = None
    isc = ish = mean = invstd = None
    if in_aff is not None:
        if bnred is None or not need_dx:
            raise _lib.SmaatHipError("depthwise backward with an on-load activation needs bnred=(mean, invstd) and dx")
        mean, invstd = bnred
        isc, ish = in_aff
        rpart = _new(x, 2, rows - 1, cin)
    rc = L.smaat_dw3x3_bwd_t(_ptr(x), _dt(x), x_bs, _ptr(isc), _ptr(ish), _ptr(dy), _dt(dy), k * h * w, _ptr(w_dw), _ptr(dx),
                             _dt(dx) if dx is not None else _dt(x), cin * h * w, _ptr(ws2), _ptr(dw_dw), _ptr(db_dw),
                             _ptr(mean), _ptr(invstd), _ptr(rpart), n, cin, kpl, h, w, s)
    if rc == -2:
        raise NotImplementedError(f"depthwise 3x3 backward with bf16 storage: [{n},{cin},{h},{w}] kpl={kpl} not built")
    _lib.check(rc, "smaat_dw3x3_bwd_t")
    if in_aff is not None:
        return dx, dw_dw, db_dw, dw_pw, (rpart, rows - 1)
    if bnred is not None:
        return dx, dw_dw, db_dw, dw_pw, None
    return dx, dw_dw, db_dw, dw_pw


# keep the depthwise output of the forward for the backward (streamed weight gradient).  Set to
# False to trade speed for memory: the backward then recomputes it inside the wgrad kernel.
# dgrad GEMM + depthwise backward in one kernel (csrc/dsbwd.hip).  OFF by default: correct (dX bit-identical to the two-kernel form)
# but measured SLOWER -- inc.1 at batch 32: 1.23 ms against 0.40 + 0.53 ms, up4.0 2.48 against 0.76 + 1.06; the step 29.4 against
# 28.0 ms (profiles/r6/dsbwd_ablate_r6i.txt: one workgroup of 12 waves per CU and a barrier per row leave every wave's chain of
# waits exposed; the dword-granular row accesses a 30-column stride forces cost 0.5 ms of the 1.23).  SMAAT_FUSED_BWD=1 selects it.
# policy.fused_bwd (default os.environ.get("SMAAT_FUSED_BWD", "0") == "1")
# policy.keep_depthwise_output (default True)


# --------------------------------------------------------------------------------------
# DepthwiseSeparableConv (+ BatchNorm2d + ReLU)
# --------------------------------------------------------------------------------------
def _bf16_storage_ok(h, w, kpl):
    """planes the bf16-storage depthwise / GEMM kernels take: an even width (rows of 4-element groups, the last one possibly
    2 wide; an even plane for the GEMM's 2-pixel stores), kernels_per_layer 1, 2 or 4"""
    return w % 2 == 0 and h >= 1 and kpl in (1, 2, 4)


def _half_forward(x, w_dw, b_dw, w_pw, b_pw, gamma, beta, rm, rv, training, momentum, eps, kpl, keep_y, in_aff=None,
                  want_act=True, amx=None, xb=None):
    """DepthwiseSeparableConv -> BatchNorm2d -> ReLU.  Returns y, z, st, y_dw, use_batch_stats.
    in_aff = (scale, shift): x is a PRE-BatchNorm tensor and relu(x*scale + shift) is applied on load.
    want_act=False: do not materialise y (the consumer applies this BatchNorm + ReLU on load).
    amx (dict, filled in place; None = exact three-term split everywhere): the operand maxima of this half for the two-term
    fp16 split -- {"w": int32 words [max |y_dw|, max |dz|], "y": produced by the forward, "dz": to be produced by the backward}
    xb: a bound of |x| for the row-walking fused forward on the two-term split (_dsconv_fwd_rows)"""
    n, cin, h, w = x.shape
    cout = w_pw.shape[0]
    use_batch_stats = training or rm is None
    y_dw = None
    isc, ish = in_aff if in_aff is not None else (None, None)
    rs = None
    bf = _is_bf(x) or mixed_precision_active()
    if bf and not _bf16_storage_ok(h, w, kpl):
        # a plane the bf16-storage kernel families do not take (odd width ...): this block runs with f32 storage -- mixed
        # precision is an optimisation, not a contract on results (also for a standalone block under autocast)
        bf = False
        if _is_bf(x):
            x = x.float()
    rec = (keep_y and _recompute_wgrad_ok(n, cin, h, w, kpl, cout) and (not bf or policy.bf16_recompute)
           and _recompute_operand_ok(x))
    if rec:
        keep_y = False  # the weight gradient recomputes the depthwise output from x: nothing to keep (y_dw = None)
    if amx is not None:
        amx.update(w=None, y=False, dz=False)
        if not bf and use_batch_stats and n * h * w >= policy.f16_min_samples and _f16_on():
            amx["w"] = _amax_words(x, 2)
            amx["dz"] = True  # (the backward's BatchNorm apply kernel will leave max |dz| in the second buffer)
    if bf:  # mixed precision: bf16 depthwise output, bf16 GEMM, bf16 z (the f32 kernel families are not involved)
        if rec or not keep_y:  # no depthwise tensor wanted: the row-walking fused kernel where it takes the shape
            rs = _dsconv_fwd_rows(x, w_dw, b_dw, w_pw, b_pw, kpl, use_batch_stats, isc, ish, out_dtype=BF16)
        if rs is None:
            rs = _dsconv_fwd_bf16(x, w_dw, b_dw, w_pw, b_pw, kpl, use_batch_stats, isc, ish)
            if rec:
                rs = rs[:3] + (None,)
    elif _split_on() and _fused_dw_ok(n, h, w, kpl, cout, keep_y, cin):
        if not keep_y:
            rs = _dsconv_fwd_rows(x, w_dw, b_dw, w_pw, b_pw, kpl, use_batch_stats, isc, ish, amx=amx if rec else None,
                                  xb=xb if rec else None)
        if rs is None:
            rs = _dsconv_fwd_fused(x, w_dw, b_dw, w_pw, b_pw, kpl, use_batch_stats, isc, ish, want_y=keep_y)
    if rs is None and not bf and _split_fwd_ok(cin * kpl, cout, use_batch_stats):
        rs = _dsconv_fwd_split(x, w_dw, b_dw, w_pw, b_pw, kpl, use_batch_stats, isc, ish,
                               amx=amx if (amx is not None and amx["w"] is not None) else None)
    if rs is not None and use_batch_stats:
        z, part, slots, y_dw = rs
        if not keep_y:
            y_dw = None
        st = _bn_finalize_raw(part, slots, cout, n * h * w, b_pw, gamma, beta, eps,
                              momentum if momentum is not None else 0.0, rm if training else None,
                              rv if training else None)
    elif use_batch_stats:
        r = _dsconv_fwd_raw(x, w_dw, b_dw, w_pw, b_pw, kpl, True, in_scale=isc, in_shift=ish, want_y=keep_y)
        z, part, slots = r[:3]
        y_dw = r[3] if keep_y else None
        st = _bn_finalize_raw(part, slots, cout, n * h * w, b_pw, gamma, beta, eps,
                              momentum if momentum is not None else 0.0, rm if training else None,
                              rv if training else None)
    else:
        if rs is not None:
            z, y_dw = rs[0], (rs[3] if keep_y else None)
        else:
            r = _dsconv_fwd_raw(x, w_dw, b_dw, w_pw, b_pw, kpl, False, in_scale=isc, in_shift=ish, want_y=keep_y)
            z = r[0]
            y_dw = r[3] if keep_y else None
        st = _bn_eval_coefs_raw(rm, rv, gamma, beta, eps)
    y = _affine_act_raw(z, st[2], st[3], True) if want_act else None
    return y, z, st, y_dw, use_batch_stats


def _half_backward(x, w_dw, b_dw, w_pw, gamma, z, st, y_dw, dy, kpl, train_stats, has_bias, need_dx, pre_part=None,
                   bnred=None, in_aff=None, head=None, amx=None):
    """-> (dx, dw_dw, db_dw, dw_pw, db_pw, dgamma, dbeta), red.  pre_part: this BatchNorm's backward sums
    (from the following block's depthwise backward); in_aff=(scale, shift) + bnred=(mean, invstd) of the PREVIOUS
    BatchNorm: x is its input, the activation is applied on load and its backward sums are emitted (`red`)."""
    if amx is not None and (amx.get("w") is None or not amx.get("dz") or _is_bf(z) or _is_bf(dy) or not _f16_on()):
        amx = None  # (no maxima words, or a storage / mode change since the forward)
    if amx is not None:
        # max |dz| is accumulated with atomic max into words that "must hold 0 on entry" (include/smaat_hip.h): a SECOND backward
        # over the same graph (retain_graph, several losses, Jacobian rows) must not start from the first one's maximum -- a much
        # smaller dz would be scaled by a stale power of two and lose its second term.  Fresh zero words per pass
        # (a slice of the arena: no launch); the forward's words keep max |y|, which does not change between passes.
        amx = dict(amx, wdz=_amax_words(z, 1))
    a_dz = amx["wdz"] if amx is not None else None
    if head is not None:  # dy is w_out (x) dlog, formed on the fly; head["dw"] receives the 1x1 conv's weight gradient
        dz, dgamma, dbeta, head["dw"] = _bn_bwd_head_raw(head["dlog"], head["w"], z, st, gamma, train_stats, amax=a_dz)
    else:
        dz, dgamma, dbeta = _bn_bwd_raw(dy, z, st, gamma, True, train_stats, pre_part=pre_part, amax=a_dz)
    r = _dsconv_bwd_raw(x, w_dw, b_dw, w_pw, dz, kpl, need_dx, y=y_dw, bnred=bnred, in_aff=in_aff, amx=amx)
    dx, dw_dw, db_dw, dw_pw = r[:4]
    red = r[4] if bnred is not None else None
    if train_stats:
        # a bias in front of a train-mode BatchNorm has an exactly-zero gradient
        db_pw = _zero_grad_words(dgamma, dgamma.numel()) if has_bias[1] else None
    else:
        db_pw = _channel_sum_raw(dz) if has_bias[1] else None
    if not has_bias[0]:
        db_dw = None
    if gamma is None:
        dgamma = dbeta = None
    return (dx, dw_dw, db_dw, dw_pw, db_pw, dgamma, dbeta), red


class _DSConvBNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_dw, b_dw, w_pw, b_pw, gamma, beta, rm, rv, training, momentum, eps, kpl):
        _check(x, w_dw, b_dw, w_pw, b_pw, gamma, beta, rm, rv)
        _expect_dsconv(x, w_dw, b_dw, w_pw, b_pw, kpl, (("bn.weight", gamma), ("bn.bias", beta), ("bn.running_mean", rm),
                                                         ("bn.running_var", rv)))
        w_dw = w_dw.contiguous()
        w_pw = w_pw.contiguous()
        # (mixed precision: the bf16 backward streams the kept depthwise output whatever input needs a gradient -- also
        # when only the BatchNorm affine parameters do)
        bfm = _is_bf(x) or mixed_precision_active()
        keep_y = ((policy.keep_depthwise_output or bfm)
                  and any(ctx.needs_input_grad[:7 if bfm else 4]))  # forward runs under no_grad
        amx = {} if keep_y else None
        xa = _x_amax_of(x) if amx is not None else None
        y, z, st, y_dw, ubs = _half_forward(x, w_dw, b_dw, w_pw, b_pw, gamma, beta, rm, rv, training, momentum, eps,
                                            kpl, keep_y, amx=amx, xb=dict(amax=xa[0], amax2=xa[1]) if xa else None)
        ctx.amx = amx
        ctx.save_for_backward(x, w_dw, b_dw, w_pw, gamma, z, st, y_dw)
        ctx.kpl = kpl
        ctx.train_stats = ubs
        ctx.has_bias = (b_dw is not None, b_pw is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_dw, b_dw, w_pw, gamma, z, st, y_dw = ctx.saved_tensors
        g, _ = _half_backward(x, w_dw, b_dw, w_pw, gamma, z, st, y_dw, dy, ctx.kpl, ctx.train_stats, ctx.has_bias,
                              ctx.needs_input_grad[0], amx=ctx.amx)
        return g + (None,) * 6


class _DoubleConvDS(torch.autograd.Function):
    """(DepthwiseSeparableConv => BN => ReLU) * 2 as ONE autograd node (reference
    unet_parts_depthwise_separable.py:10-39), so that the backward of the second half can hand the first
    BatchNorm's backward reduction to the first half (computed inside the depthwise backward kernel, which
    already streams dX and y1): one pass over (dy1, z1) less."""

    @staticmethod
    def forward(ctx, x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2,
                tr1, mo1, eps1, tr2, mo2, eps2, kpl, w_out=None, b_out=None, defer=False):
        _check(x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2, w_out,
               b_out)
        c1 = _expect_dsconv(x, w_dw1, b_dw1, w_pw1, b_pw1, kpl, (("double_conv.1.weight", g1), ("double_conv.1.bias", be1),
                                                                  ("double_conv.1.running_mean", rm1),
                                                                  ("double_conv.1.running_var", rv1)))
        _expect(w_dw2, (c1 * kpl, 1, 3, 3), "double_conv.3.depthwise.weight")
        _expect(b_dw2, (c1 * kpl,), "double_conv.3.depthwise.bias")
        _expect(w_pw2, (None, c1 * kpl, 1, 1), "double_conv.3.pointwise.weight")
        for nm, t in (("pointwise.bias", b_pw2), ("double_conv.4.weight", g2), ("double_conv.4.bias", be2),
                      ("double_conv.4.running_mean", rm2), ("double_conv.4.running_var", rv2)):
            _expect(t, (w_pw2.shape[0],), "double_conv.3/4 " + nm)
        w_dw1, w_pw1, w_dw2, w_pw2 = (t.contiguous() for t in (w_dw1, w_pw1, w_dw2, w_pw2))
        keep_y = (policy.keep_depthwise_output or _is_bf(x) or mixed_precision_active()) and any(ctx.needs_input_grad[:17])
        # the activation y1 = relu(bn1(z1)) is never written when every consumer can apply it on load:
        # the second half's forward (depthwise stage) and its depthwise backward (strip kernel: W % 4 == 0),
        # and the weight gradient reads the kept depthwise output, not y1
        n, _, h, w = x.shape
        bf = _is_bf(x) or mixed_precision_active()
        fuse = (policy.fuse_first_activation and keep_y and g1 is not None
                and bool(_lib.get().smaat_dw3x3_strip_ok(kpl, h, w))
                and (not bf or kpl <= 2))  # (bf16 storage: the row-streaming backward, kernels_per_layer <= 2)
        amx1, amx2 = ({}, {}) if keep_y else (None, None)  # operand maxima of the two halves (two-term fp16 split)
        xa = _x_amax_of(x) if amx1 is not None else None  # (a concatenation buffer whose writers left their maxima)
        y1, z1, st1, ydw1, ubs1 = _half_forward(x, w_dw1, b_dw1, w_pw1, b_pw1, g1, be1, rm1, rv1, tr1, mo1, eps1, kpl,
                                                keep_y, want_act=not fuse, amx=amx1,
                                                xb=dict(amax=xa[0], amax2=xa[1]) if xa else None)
        # head: an OutConv with ONE output channel consumes the block (w_out [1][C][1][1]): the block output is not
        # written, the logits come from the pre-BatchNorm tensor with the activation applied on load
        head = w_out is not None
        if head:
            _expect(w_out, (1, w_pw2.shape[0], 1, 1), "outc.conv.weight")
            _expect(b_out, (1,), "outc.conv.bias")
        # the second half reads z1 = w_pw1 . y1 + b_pw1 with the activation applied on load: where the first half left max |y1|,
        # |z1| is bounded per channel through the weight (smaat_dsconv_fwd_rows_h, prev_w form)
        xb2 = None
        if fuse and amx1 is not None and amx1.get("y") and amx1.get("w") is not None and z1.dtype == F32:
            xb2 = dict(amax=amx1["w"][:AMAX_WORDS], prev_w=w_pw1.reshape(w_pw1.shape[0], -1), prev_b=b_pw1)
        y2, z2, st2, ydw2, ubs2 = _half_forward(z1 if fuse else y1, w_dw2, b_dw2, w_pw2, b_pw2, g2, be2, rm2, rv2, tr2,
                                                mo2, eps2, kpl, keep_y, in_aff=(st1[2], st1[3]) if fuse else None,
                                                want_act=not (head or defer), amx=amx2, xb=xb2)
        ctx.amx = (amx1, amx2)
        ctx.save_for_backward(x, w_dw1, b_dw1, w_pw1, g1, be1, z1, st1, ydw1, y1, w_dw2, b_dw2, w_pw2, g2, z2, st2,
                              ydw2, w_out)
        ctx.fuse = fuse
        ctx.kpl = kpl
        ctx.train_stats = (ubs1, ubs2)
        ctx.has_bias = ((b_dw1 is not None, b_pw1 is not None), (b_dw2 is not None, b_pw2 is not None))
        ctx.head = (True, b_out is not None) if head else None
        if head:
            return _outconv1_fwd_raw(z2, st2[2], st2[3], w_out, b_out)
        if defer:
            # Deferred activation: the node hands out the PRE-BatchNorm tensor z2 with the coefficients of
            # y2 = relu(z2 * scale + shift); its consumer (the attention block of the fused encoder wiring) applies and
            # materialises the activation inside its first kernel and returns, as the gradient of this output, the
            # gradient with respect to y2 -- which is what backward() below expects in every mode.
            ctx.mark_non_differentiable(st2)
            return z2, st2  # st2 rows: mean, invstd, scale, shift
        return y2

    @staticmethod
    def backward(ctx, dy2, *_unused):
        (x, w_dw1, b_dw1, w_pw1, g1, be1, z1, st1, ydw1, y1, w_dw2, b_dw2, w_pw2, g2, z2, st2,
         ydw2, w_out) = ctx.saved_tensors
        head = None
        if ctx.head is not None:  # dy2 is the gradient of the logits [N][1][H][W]
            head = dict(dlog=dy2.contiguous(), w=w_out, dw=None)
        gr2, red = _half_backward(z1 if ctx.fuse else y1, w_dw2, b_dw2, w_pw2, g2, z2, st2, ydw2, dy2, ctx.kpl,
                                  ctx.train_stats[1], ctx.has_bias[1], True,
                                  bnred=(st1[0], st1[1]) if ctx.fuse else None,
                                  in_aff=(st1[2], st1[3]) if ctx.fuse else None, head=head, amx=ctx.amx[1])
        gr1, _ = _half_backward(x, w_dw1, b_dw1, w_pw1, g1, z1, st1, ydw1, gr2[0], ctx.kpl, ctx.train_stats[0],
                                ctx.has_bias[0], ctx.needs_input_grad[0], pre_part=red, amx=ctx.amx[0])
        ghead = (None, None)
        if head is not None:
            ghead = (head["dw"], _channel_sum_raw(head["dlog"]) if ctx.head[1] else None)
        return gr1 + (None, None) + gr2[1:] + (None, None) + (None,) * 7 + ghead + (None,)


def double_conv_ds(x, half1, half2, kpl, head=None, defer=False):
    """half = (w_dw, b_dw, w_pw, b_pw, gamma, beta, running_mean, running_var, training, momentum, eps).
    head = (w_out [1][C][1][1], b_out [1] or None): returns OutConv(block(x)) for an OutConv with ONE output channel,
    with the block output and its gradient never materialised (see _DoubleConvDS)."""
    a, b = half1, half2
    if head is not None:
        return _DoubleConvDS.apply(x, *a[:8], *b[:8], a[8], a[9], a[10], b[8], b[9], b[10], kpl, head[0], head[1])
    if defer:  # -> (z2, st2): the consumer applies relu(z2 * st2[2] + st2[3]), see _DoubleConvDS.forward
        return _DoubleConvDS.apply(x, *a[:8], *b[:8], a[8], a[9], a[10], b[8], b[9], b[10], kpl, None, None, True)
    return _DoubleConvDS.apply(x, *a[:8], *b[:8], a[8], a[9], a[10], b[8], b[9], b[10], kpl)


def dsconv_bn_relu(x, w_dw, b_dw, w_pw, b_pw, gamma, beta, running_mean, running_var, training, momentum, eps, kpl):
    return _DSConvBNReLU.apply(x, w_dw, b_dw, w_pw, b_pw, gamma, beta, running_mean, running_var, training,
                               momentum, eps, kpl)


# --------------------------------------------------------------------------------------
# Inference fast path (reference call stack D: model.eval() forward, calc_metrics_test_set.py:119):
# BatchNorm folded into the pointwise conv once per set of weights,
#     w' = w * gamma / sqrt(running_var + eps),   b' = (b - running_mean) * gamma / sqrt(running_var + eps) + beta
# (models/unet_parts_depthwise_separable.py:25,34 in eval mode is exactly this affine map), so a half block is ONE
# fused depthwise -> pointwise launch; the ReLU is applied by the consumer on load (in_scale = 1, in_shift = 0) or by
# one streaming pass for the block output.
# --------------------------------------------------------------------------------------
def fold_bn_into_pointwise(w_pw, b_pw, gamma, beta, rm, rv, eps):
    """-> dict(w [Cout][K] folded, b [Cout] folded, wt [K][Cout] (f32 kernels), planes (split kernels))"""
    cout = w_pw.shape[0]
    with torch.no_grad():
        s = torch.rsqrt(rv.double() + eps)
        if gamma is not None:
            s = s * gamma.double()
        w2 = (w_pw.reshape(cout, -1).double() * s[:, None]).float().contiguous()
        b0 = b_pw.double() if b_pw is not None else torch.zeros(cout, dtype=torch.float64, device=w_pw.device)
        b2 = (b0 - rm.double()) * s
        if beta is not None:
            b2 = b2 + beta.double()
        b2 = b2.float().contiguous()
        fold = dict(w=w2, b=b2, wt=w2.t().contiguous())
        fold["planes"] = torch.ops.smaat.split_planes(w2) if _split_on() else None
    return fold


_UNIT = {}


def _unit_affine(c, ref):
    key = (c, ref.device)
    if key not in _UNIT:
        _UNIT[key] = (torch.ones(c, dtype=torch.float32, device=ref.device),
                      torch.zeros(c, dtype=torch.float32, device=ref.device))
    return _UNIT[key]


def dsconv_folded(x, w_dw, b_dw, fold, kpl, relu_out=True):
    """relu?(pointwise'(depthwise(x))) with BatchNorm folded into the pointwise conv and the ReLU fused into the GEMM
    epilogue: a whole DepthwiseSeparableConv -> BatchNorm2d(eval) -> ReLU half block.  No autograd (inference)."""
    _check(x, w_dw, b_dw, bf16=False)
    L = _lib.get()
    x, x_bs = _planes(x)
    n, cin, h, w = x.shape
    cout = fold["w"].shape[0]
    ro = 1 if relu_out else 0
    z = _new(x, n, cout, h, w)
    if fold["planes"] is not None and kpl == 2 and policy.fuse_dw_split != "off" and L.smaat_dsconv_split_num_slots(n, h, w) > 0:
        rc = L.smaat_dsconv_fwd_split_act(_ptr(x), x_bs, None, None, _ptr(w_dw), _ptr(b_dw), _ptr(fold["planes"]),
                                          _ptr(fold["b"]), _ptr(z), cout * h * w, n, cin, kpl, cout, h, w, ro, _stream(x))
        if rc == 0:
            return z
        if rc != -2:
            _lib.check(rc, "smaat_dsconv_fwd_split_act")
    if fold["planes"] is not None and _split_fwd_ok(cin * kpl, cout, train=False):
        # GEMM-sized layers: depthwise kernel + persistent split GEMM (2 launches) beat the f32-MFMA fused kernel
        y = _dw3x3_fwd_raw(x, w_dw, b_dw, kpl)
        if y is not None:
            # (few tiles, long contraction -- batch 1 on the deep levels: the library cuts the contraction into slices)
            wsn = L.smaat_pointwise_splitk_ws_floats(n, cin * kpl, cout, h, w)
            ws = _new(x, wsn) if wsn > 0 else None
            _lib.check(L.smaat_pointwise_fwd_split_act_k(_ptr(y), cin * kpl * h * w, _ptr(fold["planes"]), _ptr(fold["b"]),
                                                         _ptr(z), cout * h * w, _ptr(ws), n, cin * kpl, cout, h, w, ro,
                                                         _stream(x)), "smaat_pointwise_fwd_split_act_k")
            return z
    _lib.check(L.smaat_dsconv_fwd_act(_ptr(x), x_bs, None, None, _ptr(w_dw), _ptr(b_dw), _ptr(fold["wt"]),
                                      _ptr(fold["b"]), _ptr(z), cout * h * w, n, cin, kpl, cout, h, w, ro, _stream(x)),
               "smaat_dsconv_fwd_act")
    return z


def double_conv_ds_eval(x, half1, half2, kpl):
    """eval-mode DoubleConvDS under no_grad: half = (w_dw, b_dw, fold).  One fused launch per half on the
    plane-dominated layers, depthwise + GEMM on the deep ones; no BatchNorm / ReLU kernels at all.  Goes through
    torch.ops.smaat.dsconv_folded (smaat_unet_amd/torch_ops.py) so that the inference graph is traceable."""
    for w_dw, b_dw, f in (half1, half2):
        x = torch.ops.smaat.dsconv_folded(x, w_dw, b_dw, f["w"], f["wt"], f["planes"], f["b"], kpl, True)
    return x


class _DSConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_dw, b_dw, w_pw, b_pw, kpl):
        _check(x, w_dw, b_dw, w_pw, b_pw, bf16=False)
        _expect_dsconv(x, w_dw, b_dw, w_pw, b_pw, kpl)
        w_dw = w_dw.contiguous()
        w_pw = w_pw.contiguous()
        keep_y = policy.keep_depthwise_output and any(ctx.needs_input_grad[:4])
        rs = (_dsconv_fwd_split(x, w_dw, b_dw, w_pw, b_pw, kpl, False)
              if _split_fwd_ok(w_pw.shape[1], w_pw.shape[0], any(ctx.needs_input_grad[:5])) else None)
        if rs is not None:
            r = (rs[0], None, 0, rs[3])
        else:
            r = _dsconv_fwd_raw(x, w_dw, b_dw, w_pw, b_pw, kpl, False, want_y=keep_y)
            if not keep_y:
                r = (r[0], None, 0, None)
        z = r[0]
        ctx.save_for_backward(x, w_dw, b_dw, w_pw, r[3] if keep_y else None)
        ctx.kpl = kpl
        ctx.has_bias = (b_dw is not None, b_pw is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w_dw, b_dw, w_pw, y_dw = ctx.saved_tensors
        dx, dw_dw, db_dw, dw_pw = _dsconv_bwd_raw(x, w_dw, b_dw, w_pw, dz, ctx.kpl, ctx.needs_input_grad[0], y=y_dw)
        db_pw = _channel_sum_raw(dz) if ctx.has_bias[1] else None
        if not ctx.has_bias[0]:
            db_dw = None
        return dx, dw_dw, db_dw, dw_pw, db_pw, None


def dsconv(x, w_dw, b_dw, w_pw, b_pw, kpl):
    return _DSConv.apply(x, w_dw, b_dw, w_pw, b_pw, kpl)


# --------------------------------------------------------------------------------------
# depthwise convolution of any geometry (DepthwiseSeparableConv outside the network's 3x3 / padding 1 / kpl in {1, 2, 4})
# --------------------------------------------------------------------------------------
class _DepthwiseAny(torch.autograd.Function):
    """reference models/layers.py:38-44,48 for any kernel_size / padding / kernels_per_layer (stride 1, dilation 1); f32"""

    @staticmethod
    def forward(ctx, x, w, b, kpl, ph, pw):
        _check(x, w, b)
        if x.dim() != 4:
            raise ValueError(f"input: expected [N, C, H, W], got {tuple(x.shape)}")
        if x.dtype != F32:
            raise NotImplementedError("the general depthwise kernels are f32 only (mixed precision covers the 3x3 / padding 1 "
                                      "layers of the network)")
        n, cin, h, wd = x.shape
        _expect(w, (cin * kpl, 1, None, None), "depthwise.weight")
        _expect(b, (cin * kpl,), "depthwise.bias")
        kh, kw = w.shape[2], w.shape[3]
        ho, wo = h + 2 * ph - kh + 1, wd + 2 * pw - kw + 1
        if ho < 1 or wo < 1:
            raise RuntimeError(f"Calculated padded input size per channel: ({h + 2 * ph} x {wd + 2 * pw}). Kernel size: "
                               f"({kh} x {kw}). Kernel size can't be greater than actual input size")
        L = _lib.get()
        x, x_bs = _planes(x)
        w = w.contiguous()
        y = _new(x, n, cin * kpl, ho, wo)
        _lib.check(L.smaat_dwconv_fwd_any(_ptr(x), x_bs, _ptr(w), _ptr(b), _ptr(y), cin * kpl * ho * wo, n, cin, kpl, h, wd, kh,
                                          kw, ph, pw, _stream(x)), "smaat_dwconv_fwd_any")
        ctx.save_for_backward(x, w)
        ctx.geom = (kpl, ph, pw, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        kpl, ph, pw, has_b = ctx.geom
        L = _lib.get()
        x, x_bs = _planes(x)
        dy, dy_bs = _planes(dy.float())
        n, cin, h, wd = x.shape
        kh, kw = w.shape[2], w.shape[3]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2])
        dx = _new(x, n, cin, h, wd) if need_x else None
        dw = _new(x, *w.shape) if need_w else None
        db = _new(x, cin * kpl) if (need_w and has_b) else None
        _lib.check(L.smaat_dwconv_bwd_any(_ptr(x), x_bs, _ptr(dy), dy_bs, _ptr(w), _ptr(dx), cin * h * wd, _ptr(dw), _ptr(db), n,
                                          cin, kpl, h, wd, kh, kw, ph, pw, _stream(x)), "smaat_dwconv_bwd_any")
        return dx, dw, db, None, None, None


def depthwise_any(x, w, b, kpl, ph, pw):
    return _DepthwiseAny.apply(x, w, b, kpl, ph, pw)


class _PointwiseBNReLU(torch.autograd.Function):
    """nn.Conv2d(K, M, 1) -> BatchNorm2d -> ReLU behind a general depthwise stage (a DoubleConvDS half whose
    DepthwiseSeparableConv is outside the fused configuration; reference unet_parts_depthwise_separable.py:17-36):
    f32-MFMA GEMM with the statistics partials in its epilogue, finalize, one apply pass; f32."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rm, rv, training, momentum, eps):
        _check(x, w, b, gamma, beta, rm, rv)
        _expect(w, (None, x.shape[1], 1, 1), "pointwise.weight")
        m, c = w.shape[0], w.shape[1]
        _expect(b, (m,), "pointwise.bias")
        for nm, t in (("bn.weight", gamma), ("bn.bias", beta), ("bn.running_mean", rm), ("bn.running_var", rv)):
            _expect(t, (m,), nm)
        if x.dtype != F32:
            raise NotImplementedError("general DepthwiseSeparableConv geometries are f32 only")
        L = _lib.get()
        w = w.contiguous()
        x, x_bs = _planes(x)
        n, _, h, wd = x.shape
        use_batch_stats = training or rm is None
        wt = w.reshape(m, c).t().contiguous()
        if use_batch_stats:
            z = _new(x, n, m, h, wd)
            slots = L.smaat_pw_num_slots(n, h, wd, m)
            part = _new(x, 3, slots, m)
            _lib.check(L.smaat_pointwise_fwd(_ptr(x), x_bs, _ptr(wt), _ptr(b), _ptr(z), m * h * wd, _ptr(part), n, c, m, h, wd,
                                             _stream(x)), "smaat_pointwise_fwd")
            st = _bn_finalize_raw(part, slots, m, n * h * wd, b, gamma, beta, eps, momentum if momentum is not None else 0.0,
                                  rm if training else None, rv if training else None)
        else:
            z = _pointwise_raw(x, wt, b, m)
            st = _bn_eval_coefs_raw(rm, rv, gamma, beta, eps)
        y = _affine_act_raw(z, st[2], st[3], True)
        ctx.save_for_backward(x, w, gamma, z, st)
        ctx.flags = (use_batch_stats, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, gamma, z, st = ctx.saved_tensors
        train_stats, has_bias = ctx.flags
        m, c = w.shape[0], w.shape[1]
        dz, dgamma, dbeta = _bn_bwd_raw(dy, z, st, gamma, True, train_stats)
        dx = _pointwise_raw(dz, w.reshape(m, c), None, c) if ctx.needs_input_grad[0] else None
        dw = _pointwise_wgrad_raw(x, dz, m).reshape(w.shape)
        db = None
        if has_bias:  # a bias in front of a train-mode BatchNorm has an exactly-zero gradient
            db = _zero_grad_words(dgamma, dgamma.numel()) if train_stats else _channel_sum_raw(dz)
        if gamma is None:
            dgamma = dbeta = None
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None


def pointwise_bn_relu(x, w, b, gamma, beta, rm, rv, training, momentum, eps):
    return _PointwiseBNReLU.apply(x, w, b, gamma, beta, rm, rv, training, momentum, eps)


# --------------------------------------------------------------------------------------
# OutConv (plain 1x1)
# --------------------------------------------------------------------------------------
class _Pointwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _check(x, w, b)
        if x.dim() != 4:
            raise ValueError(f"input: expected [N, C, H, W], got {tuple(x.shape)}")
        _expect(w, (None, x.shape[1], 1, 1), "conv.weight")
        _expect(b, (w.shape[0],), "conv.bias")
        w = w.contiguous()
        m, c = w.shape[0], w.shape[1]
        if _is_bf(x):  # mixed precision: bf16 activations in, f32 logits out
            out, _, _ = _pointwise_bf16_raw(x, _bf16_planes_raw(w.reshape(m, c)), b, m, out_dtype=F32)
        else:
            wt = w.reshape(m, c).t().contiguous()
            out = _pointwise_raw(x, wt, b, m)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        m, c = w.shape[0], w.shape[1]
        dx = None
        if _is_bf(x):
            dzb = dz.to(BF16)  # (n_classes channels: a small tensor) -- the bf16 GEMMs take bf16 operands
            if ctx.needs_input_grad[0]:
                dx, _, _ = _pointwise_bf16_raw(dzb, _bf16_planes_raw(w.reshape(m, c), transpose=True), None, c)
            dw = _pointwise_wgrad_raw(x, dzb, m)
            db = _channel_sum_raw(dz) if ctx.has_bias else None
            return dx, dw, db
        if ctx.needs_input_grad[0]:
            dx = _pointwise_raw(dz, w.reshape(m, c), None, c)  # wt[c'=m][m'=c] = w natural
        dw = _pointwise_wgrad_raw(x, dz, m)
        db = _channel_sum_raw(dz) if ctx.has_bias else None
        return dx, dw, db


def pointwise(x, w, b):
    return _Pointwise.apply(x, w, b)


# --------------------------------------------------------------------------------------
# MaxPool2d(2)
# --------------------------------------------------------------------------------------
class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _check(x)
        L = _lib.get()
        x, x_bs = _planes(x)
        n, c, h, w = x.shape
        y = _maxpool2_fwd_raw(x, x_bs)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.get()
        (x,) = ctx.saved_tensors
        x, x_bs = _planes(x)
        dy, dy_bs = _planes(dy)
        n, c, h, w = x.shape
        dx = _new(x, n, c, h, w, dtype=x.dtype)
        _maxpool2_bwd_raw(x, x_bs, dy, dy_bs, dx, 0)
        return dx


def _maxpool2_fwd_raw(x, x_bs):
    L = _lib.get()
    n, c, h, w = x.shape
    y = _new(x, n, c, h // 2, w // 2, dtype=x.dtype)
    if _is_bf(x):
        _lib.check(L.smaat_maxpool2_fwd_t(_ptr(x), x_bs, _ptr(y), c * (h // 2) * (w // 2), n, c, h, w, 1, _stream(x)),
                   "smaat_maxpool2_fwd_t")
    else:
        _lib.check(L.smaat_maxpool2_fwd(_ptr(x), x_bs, _ptr(y), c * (h // 2) * (w // 2), n, c, h, w, _stream(x)),
                   "smaat_maxpool2_fwd")
    return y


def _maxpool2_bwd_raw(x, x_bs, dy, dy_bs, dx, accum):
    """dx (+)= maxpool2 backward; x, dy, dx share one dtype"""
    L = _lib.get()
    n, c, h, w = x.shape
    if _is_bf(x):
        if dy.dtype != BF16:
            dy, dy_bs = _planes(dy.to(BF16))
        _lib.check(L.smaat_maxpool2_bwd_t(_ptr(x), x_bs, _ptr(dy), dy_bs, _ptr(dx), c * h * w, n, c, h, w, accum, 1,
                                          _stream(x)), "smaat_maxpool2_bwd_t")
    else:
        _lib.check(L.smaat_maxpool2_bwd(_ptr(x), x_bs, _ptr(dy), dy_bs, _ptr(dx), c * h * w, n, c, h, w, accum,
                                        _stream(x)), "smaat_maxpool2_bwd")


def maxpool2(x):
    return _MaxPool2.apply(x)


# --------------------------------------------------------------------------------------
# Upsample(x2, bilinear, align_corners=True) + F.pad + cat([x2, x1_up], dim=1)
# --------------------------------------------------------------------------------------
def _upsample_fwd_raw(x1, x1_bs, dst_ptr, dst_bs, n, c1, h, w, ho, wo, pt, pl, s):
    L = _lib.get()
    if _is_bf(x1):
        rc = L.smaat_upsample2x_fwd_t(_ptr(x1), x1_bs, dst_ptr, dst_bs, n, c1, h, w, ho, wo, pt, pl, 1, s)
        if rc == -2:
            raise NotImplementedError(f"bilinear upsample with bf16 storage: [{n},{c1},{h},{w}] -> {ho}x{wo} not built "
                                      "(output width must be a multiple of 4)")
        _lib.check(rc, "smaat_upsample2x_fwd_t")
    else:
        _lib.check(L.smaat_upsample2x_fwd(_ptr(x1), x1_bs, dst_ptr, dst_bs, n, c1, h, w, ho, wo, pt, pl, s),
                   "smaat_upsample2x_fwd")


def _upsample_bwd_raw(src_ptr, src_bs, dx1, n, c1, h, w, ho, wo, pt, pl, s):
    L = _lib.get()
    if _is_bf(dx1):
        rc = L.smaat_upsample2x_bwd_t(src_ptr, src_bs, _ptr(dx1), c1 * h * w, n, c1, h, w, ho, wo, pt, pl, 1, s)
        if rc == -2:
            raise NotImplementedError(f"bilinear upsample backward with bf16 storage: [{n},{c1},{h},{w}] <- {ho}x{wo} not built")
        _lib.check(rc, "smaat_upsample2x_bwd_t")
    else:
        _lib.check(L.smaat_upsample2x_bwd(src_ptr, src_bs, _ptr(dx1), c1 * h * w, n, c1, h, w, ho, wo, pt, pl, s),
                   "smaat_upsample2x_bwd")


def _copy_planes_raw(src_ptr, s_bs, dst_ptr, d_bs, n, plane_len, esize, s):
    """dst[n][:plane_len] = src[n][:plane_len] for tensors of `esize`-byte elements (strides in elements)"""
    L = _lib.get()
    if esize == 2:  # bf16: move pairs of elements as floats
        if (plane_len | s_bs | d_bs) & 1:
            raise NotImplementedError("plane copy of bf16 tensors with an odd element count")
        plane_len, s_bs, d_bs = plane_len // 2, s_bs // 2, d_bs // 2
    _lib.check(L.smaat_copy_planes(src_ptr, s_bs, dst_ptr, d_bs, n, plane_len, 0, s), "smaat_copy_planes")


class _UpsampleCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2):
        _check(x1, x2, acts=2)
        L = _lib.get()
        x1, x1_bs = _planes(x1)
        x2, x2_bs = _planes(x2)
        n, c1, h, w = x1.shape
        n2, c2, ho, wo = x2.shape
        assert n == n2
        dy_, dx_ = ho - 2 * h, wo - 2 * w
        if dy_ < 0 or dx_ < 0:
            raise NotImplementedError("UpDS with a skip connection smaller than the upsampled map (negative pad)")
        pt, pl = dy_ // 2, dx_ // 2
        if x2.dtype != x1.dtype:
            x2, x2_bs = _planes(x2.to(x1.dtype))
        cat = _new(x1, n, c2 + c1, ho, wo, dtype=x1.dtype)
        es = cat.element_size()
        cbs = (c1 + c2) * ho * wo
        s = _stream(x1)
        _copy_planes_raw(_ptr(x2), x2_bs, _ptr(cat), cbs, n, c2 * ho * wo, es, s)
        _upsample_fwd_raw(x1, x1_bs, cat.data_ptr() + es * c2 * ho * wo, cbs, n, c1, h, w, ho, wo, pt, pl, s)
        ctx.geom = (n, c1, h, w, c2, ho, wo, pt, pl)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        L = _lib.get()
        n, c1, h, w, c2, ho, wo, pt, pl = ctx.geom
        dcat = dcat.contiguous()
        es = dcat.element_size()
        cbs = (c1 + c2) * ho * wo
        s = _stream(dcat)
        dx1 = dx2 = None
        if ctx.needs_input_grad[1]:
            dx2 = _new(dcat, n, c2, ho, wo, dtype=dcat.dtype)
            _copy_planes_raw(_ptr(dcat), cbs, _ptr(dx2), c2 * ho * wo, n, c2 * ho * wo, es, s)
        if ctx.needs_input_grad[0]:
            dx1 = _new(dcat, n, c1, h, w, dtype=dcat.dtype)
            _upsample_bwd_raw(dcat.data_ptr() + es * c2 * ho * wo, cbs, dx1, n, c1, h, w, ho, wo, pt, pl, s)
        return dx1, dx2


def upsample_cat(x1, x2):
    return _UpsampleCat.apply(x1, x2)


# --------------------------------------------------------------------------------------
# CBAM (channel attention and/or spatial attention)
# --------------------------------------------------------------------------------------
def _cbam_forward_impl(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, use_ch, use_sp,
                       out=None, lazy=None, pool=None, out_amax=None):
    """out = spatial_att(channel_att(x)); `out` may be a channel slice of a larger buffer (dense
    planes, any batch stride).  Returns (out, saved tensors, flags).
    pool = []: the caller also needs maxpool2(x) (an encoder level): when the shape allows, the channel pooling kernel
    produces it in the same pass and it is appended to the list; otherwise the list stays empty.
    lazy = (scale, shift): x is the PRE-BatchNorm tensor of the block in front (deferred activation, see
    _DoubleConvDS): relu(x * scale + shift) is formed and written by the channel pooling kernel; saved[0] is that
    activated tensor.
    out_amax (f32 storage): an amax buffer that receives max |out| (smaat_cbam_apply_amax)"""
    _check(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv)
    if lazy is not None and not use_ch:  # no pooling kernel to fuse with: materialise up front
        x = _affine_act_raw(x, lazy[0], lazy[1], True)
        lazy = None
    if x.dim() != 4:
        raise ValueError(f"input: expected [N, C, H, W], got {tuple(x.shape)}")
    if use_ch:
        _expect(w1, (None, x.shape[1]), "channel_att.MLP.1.weight")
        _expect(b1, (w1.shape[0],), "channel_att.MLP.1.bias")
        _expect(w2, (x.shape[1], w1.shape[0]), "channel_att.MLP.3.weight")
        _expect(b2, (x.shape[1],), "channel_att.MLP.3.bias")
    if use_sp:
        if wconv.dim() != 4 or tuple(wconv.shape[:2]) != (1, 2) or wconv.shape[2] != wconv.shape[3] or wconv.shape[2] not in (3, 7):
            raise ValueError(f"spatial_att.conv.weight: expected (1, 2, k, k) with k in (3, 7), got {tuple(wconv.shape)}")
        for nm, t in (("bn.weight", gamma), ("bn.bias", beta), ("bn.running_mean", rm), ("bn.running_var", rv)):
            _expect(t, (1,), "spatial_att." + nm)
    L = _lib.get()
    x, x_bs = _planes(x)
    n, c, h, w = x.shape
    p = h * w
    s_ = _stream(x)
    dev = x
    bf = _is_bf(x)  # mixed precision: x / out are bf16; the per-(n, c) vectors, maps and the gate stay f32
    amaxc = None
    if use_ch:
        cr = w1.shape[0]
        w1 = w1.contiguous()
        w2 = w2.contiguous()
        avg = _new(dev, n, c)
        mx = _new(dev, n, c)
        amax = _new(dev, n, c, dtype=torch.int32)
        fused_pool = False
        if pool is not None and w % 4 == 0 and h >= 2:
            y = _new(dev, n, c, h, w, dtype=x.dtype) if lazy is not None else None
            pooled = _new(dev, n, c, h // 2, w // 2, dtype=x.dtype)
            rc = L.smaat_cbam_chpool_pool_t(_ptr(x), x_bs, _ptr(lazy[0]) if lazy is not None else None,
                                            _ptr(lazy[1]) if lazy is not None else None, _ptr(y) if y is not None else None,
                                            c * p, _ptr(pooled), c * (h // 2) * (w // 2), n, c, h, w, _ptr(avg), _ptr(mx),
                                            _ptr(amax), _dt(x), s_)
            if rc != -2:
                _lib.check(rc, "smaat_cbam_chpool_pool_t")
                fused_pool = True
                pool.append(pooled)
                if y is not None:
                    x, x_bs = y, c * p
        if fused_pool:
            pass
        elif lazy is not None:
            y = _new(dev, n, c, h, w, dtype=x.dtype)
            if bf:
                _lib.check(L.smaat_cbam_chpool_t(_ptr(x), x_bs, _ptr(lazy[0]), _ptr(lazy[1]), _ptr(y), c * p, n, c, p,
                                                 _ptr(avg), _ptr(mx), _ptr(amax), 1, s_), "smaat_cbam_chpool_t")
            else:
                _lib.check(L.smaat_cbam_chpool_act(_ptr(x), x_bs, _ptr(lazy[0]), _ptr(lazy[1]), _ptr(y), c * p, n, c, p,
                                                   _ptr(avg), _ptr(mx), _ptr(amax), s_), "smaat_cbam_chpool_act")
            x, x_bs = y, c * p
        elif bf:
            _lib.check(L.smaat_cbam_chpool_t(_ptr(x), x_bs, None, None, None, 0, n, c, p, _ptr(avg), _ptr(mx), _ptr(amax), 1,
                                             s_), "smaat_cbam_chpool_t")
        else:
            _lib.check(L.smaat_cbam_chpool(_ptr(x), x_bs, n, c, p, _ptr(avg), _ptr(mx), _ptr(amax), s_),
                       "smaat_cbam_chpool")
        ha = _new(dev, n, cr)
        hm = _new(dev, n, cr)
        sc = _new(dev, n, c)
        _lib.check(L.smaat_cbam_mlp(_ptr(avg), _ptr(mx), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), n, c, cr,
                                    _ptr(ha), _ptr(hm), _ptr(sc), s_), "smaat_cbam_mlp")
    else:
        avg = mx = amax = ha = hm = None
        sc = torch.ones(n, c, dtype=torch.float32, device=x.device)  # (f32 in every precision)
    if use_sp:
        wconv = wconv.contiguous()
        ks = wconv.shape[-1]
        maps = _new(dev, n, 2, h, w)
        rc = -2
        if policy.cbam_three_pass and use_ch:
            # + the channel index of the per-pixel maximum, which lets the backward split the channels over waves
            amaxc = _new(dev, n, h, w, dtype=torch.int32)
            rc = L.smaat_cbam_sppool_idx_t(_ptr(x), x_bs, _ptr(sc), n, c, p, _ptr(maps), _ptr(amaxc), 1 if bf else 0, s_)
            if rc == -2:
                amaxc = None
            else:
                _lib.check(rc, "smaat_cbam_sppool_idx_t")
        if rc == 0:
            pass
        elif bf:
            _lib.check(L.smaat_cbam_sppool_t(_ptr(x), x_bs, _ptr(sc), n, c, p, _ptr(maps), 1, s_), "smaat_cbam_sppool_t")
        else:
            _lib.check(L.smaat_cbam_sppool(_ptr(x), x_bs, _ptr(sc), n, c, p, _ptr(maps), s_), "smaat_cbam_sppool")
        nb = L.smaat_cbam_spconv_blocks(n, h, w)
        conv = _new(dev, n, 1, h, w)
        use_batch_stats = training or rm is None
        part = _new(dev, 3, nb, 1)
        _lib.check(L.smaat_cbam_spconv(_ptr(maps), _ptr(wconv), ks, n, h, w, _ptr(conv), _ptr(part), s_),
                   "smaat_cbam_spconv")
        if use_batch_stats:
            st = _bn_finalize_raw(part, nb, 1, n * p, None, gamma, beta, eps,
                                  momentum if momentum is not None else 0.0, rm if training else None,
                                  rv if training else None)
        else:
            st = _bn_eval_coefs_raw(rm, rv, gamma, beta, eps)
        gate = _new(dev, n, 1, h, w)
        _lib.check(L.smaat_cbam_gate(_ptr(conv), _ptr(st[2]), _ptr(st[3]), n * p, _ptr(gate), s_),
                   "smaat_cbam_gate")
    else:
        maps = conv = st = None
        use_batch_stats = False
        gate = torch.ones(n, 1, h, w, dtype=torch.float32, device=x.device)
    if out is None:
        out = _new(dev, n, c, h, w, dtype=x.dtype)
    out_t, o_bs = _planes(out)
    assert out_t is out, "cbam: the output slice must have dense [C][H][W] planes"
    assert out.dtype == x.dtype
    if bf:
        _lib.check(L.smaat_cbam_apply_t(_ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(out), o_bs, n, c, p, 1, s_),
                   "smaat_cbam_apply_t")
    elif out_amax is not None:
        _lib.check(L.smaat_cbam_apply_amax(_ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(out), o_bs, _ptr(out_amax), n, c, p, s_),
                   "smaat_cbam_apply_amax")
    else:
        _lib.check(L.smaat_cbam_apply(_ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(out), o_bs, n, c, p, s_),
                   "smaat_cbam_apply")
    saved = (x, w1, w2, wconv, gamma, avg, mx, amax, ha, hm, sc, maps, conv, st, gate, amaxc)
    return out, saved, (use_ch, use_sp, use_batch_stats)


# policy.cbam_three_pass: (0: the gate / main / final sequence everywhere)


def _cbam_mlp_backward(L, dev, ds, sc, avg, mx, ha, hm, w1, w2, n, c, s_):
    """shared-MLP backward of the channel attention -> dw1, db1, dw2, db2, davg[n][c], dmx[n][c]"""
    cr = w1.shape[0]
    pgs = c * cr + c + cr * c + cr
    pg = _new(dev, n, pgs)
    davg = _new(dev, n, c)
    dmx = _new(dev, n, c)
    _lib.check(L.smaat_cbam_bwd_mlp(_ptr(ds), _ptr(sc), _ptr(avg), _ptr(mx), _ptr(ha), _ptr(hm), _ptr(w1),
                                    _ptr(w2), n, c, cr, _ptr(pg), _ptr(davg), _ptr(dmx), s_),
               "smaat_cbam_bwd_mlp")
    pgr = _new(dev, pgs)
    _lib.check(L.smaat_reduce_rows(_ptr(pg), n, pgs, _ptr(pgr), 1.0, s_), "smaat_reduce_rows")
    dw2 = pgr[:c * cr].view(c, cr)
    db2 = pgr[c * cr:c * cr + c]
    dw1 = pgr[c * cr + c:c * cr + c + cr * c].view(cr, c)
    db1 = pgr[c * cr + c + cr * c:]
    return dw1, db1, dw2, db2, davg, dmx


def _cbam_backward_impl(saved, flags, dout, pooled=None):
    """-> dx, dw1, db1, dw2, db2, dwconv, dgamma, dbeta.  `dout` may be a channel slice of a larger
    gradient buffer (dense planes, any batch stride).  pooled = (d maxpool2(x), batch stride): the gradient of the
    MaxPool2d that reads x as well is added to dx (in the same pass as the channel-attention terms when the shape
    allows)."""
    L = _lib.get()
    x, w1, w2, wconv, gamma, avg, mx, amax, ha, hm, sc, maps, conv, st, gate, amaxc = saved
    use_ch, use_sp, train_stats = flags
    x, x_bs = _planes(x)
    bf = _is_bf(x)
    if dout.dtype != x.dtype:
        dout = dout.to(x.dtype)
    dout, do_bs = _planes(dout)
    n, c, h, w = x.shape
    p = h * w
    s_ = _stream(x)
    dev = x
    dwconv = dgamma = dbeta = None
    # both halves of the attention (+ the MaxPool2d that reads x too at the encoder levels): three passes that write dx once
    # and split the channels over waves (csrc/cbam.hip, k_cbam_bwd_gate_ds_v4) when the forward left the channel index of
    # the per-pixel maximum and shapes and alignments allow, else the gate / main / final sequence
    three = False
    dspart3 = None
    if amaxc is not None and use_sp and use_ch:
        dpl, dp_bs = pooled if pooled is not None else (None, 0)
        if dpl is not None and dpl.dtype != x.dtype:
            dpl, dp_bs = _planes(dpl.to(x.dtype))
            pooled = (dpl, dp_bs)
        three = L.smaat_cbam_bwd3_ok(_ptr(x), x_bs, _ptr(dout), do_bs, _ptr(dpl) if dpl is not None else None, dp_bs, n, c, h, w,
                                     1 if bf else 0) == 1
    if use_sp:
        ks = wconv.shape[-1]
        nbp = L.smaat_cbam_pix_blocks(n, p)
        dbn = _new(dev, n, p)
        part = _new(dev, 2, nbp, 1)
        if three:
            dspart3 = _new(dev, 2 * (nbp // n), n, c)  # [gate pass | ds2 pass][blocks per image][n][c]
            _lib.check(L.smaat_cbam_bwd_gate_ds_t(_ptr(dout), do_bs, _ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(conv), _ptr(st[0]),
                                                  _ptr(st[1]), n, c, p, _ptr(dbn), _ptr(part), _ptr(dspart3), 1 if bf else 0, s_),
                       "smaat_cbam_bwd_gate_ds_t")
        elif bf:
            _lib.check(L.smaat_cbam_bwd_gate_t(_ptr(dout), do_bs, _ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(conv),
                                               _ptr(st[0]), _ptr(st[1]), n, c, p, _ptr(dbn), _ptr(part), 1, s_),
                       "smaat_cbam_bwd_gate_t")
        else:
            _lib.check(L.smaat_cbam_bwd_gate(_ptr(dout), do_bs, _ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(conv),
                                             _ptr(st[0]), _ptr(st[1]), n, c, p, _ptr(dbn), _ptr(part), s_),
                       "smaat_cbam_bwd_gate")
        dgamma = _new(dev, 1)
        dbeta = _new(dev, 1)
        coef = _new(dev, 3, 1)
        _lib.check(L.smaat_bn_bwd_finalize(_ptr(part), nbp, 1, float(n * p), _ptr(gamma), _ptr(st[1]),
                                           _ptr(dgamma), _ptr(dbeta), _ptr(coef), s_), "smaat_bn_bwd_finalize")
        if not train_stats:
            coef[1:].zero_()
        nb = L.smaat_cbam_spconv_blocks(n, h, w)
        dmaps = _new(dev, n, 2, h, w)
        wpart = _new(dev, nb, 2 * ks * ks)
        _lib.check(L.smaat_cbam_bwd_spconv(_ptr(dbn), _ptr(conv), _ptr(st[0]), _ptr(st[1]), _ptr(coef),
                                           _ptr(maps), _ptr(wconv), ks, n, h, w, _ptr(dmaps), _ptr(wpart), s_),
                   "smaat_cbam_bwd_spconv")
        dwconv = _new(dev, 1, 2, ks, ks)
        _lib.check(L.smaat_reduce_rows(_ptr(wpart), nb, 2 * ks * ks, _ptr(dwconv), 1.0, s_), "smaat_reduce_rows")
        if gamma is None:
            dgamma = dbeta = None
    else:
        # no spatial half: gate == 1, no pooled-map gradients
        maps = torch.full((n, 2, h, w), float("inf"), dtype=torch.float32, device=x.device)
        dmaps = torch.zeros(n, 2, h, w, dtype=torch.float32, device=x.device)
    nbp = L.smaat_cbam_pix_blocks(n, p)
    dx = _new(dev, n, c, h, w, dtype=x.dtype)
    if three:
        per = nbp // n
        dt = 1 if bf else 0
        _lib.check(L.smaat_cbam_bwd_ds2_t(_ptr(x), x_bs, _ptr(dmaps), _ptr(amaxc), n, c, p, _ptr(dspart3[per:]), dt, s_),
                   "smaat_cbam_bwd_ds2_t")
        ds = _new(dev, n, c)
        _lib.check(L.smaat_reduce_rows(_ptr(dspart3), 2 * per, n * c, _ptr(ds), 1.0, s_), "smaat_reduce_rows")
        dw1, db1, dw2, db2, davg, dmx = _cbam_mlp_backward(L, dev, ds, sc, avg, mx, ha, hm, w1, w2, n, c, s_)
        dpl, dp_bs = pooled if pooled is not None else (None, 0)
        _lib.check(L.smaat_cbam_bwd_apply_t(_ptr(dout), do_bs, _ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(dmaps), _ptr(amaxc),
                                            _ptr(davg), _ptr(dmx), _ptr(amax), _ptr(dpl) if dpl is not None else None, dp_bs,
                                            n, c, h, w, _ptr(dx), c * p, dt, s_), "smaat_cbam_bwd_apply_t")
        return dx, dw1, db1, dw2, db2, dwconv, dgamma, dbeta
    dspart = _new(dev, nbp, c)
    if bf:
        _lib.check(L.smaat_cbam_bwd_main_t(_ptr(dout), do_bs, _ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(maps),
                                           _ptr(dmaps), n, c, p, _ptr(dx), c * p, _ptr(dspart), 1, s_),
                   "smaat_cbam_bwd_main_t")
    else:
        _lib.check(L.smaat_cbam_bwd_main(_ptr(dout), do_bs, _ptr(x), x_bs, _ptr(sc), _ptr(gate), _ptr(maps),
                                         _ptr(dmaps), n, c, p, _ptr(dx), c * p, _ptr(dspart), s_),
                   "smaat_cbam_bwd_main")
    dw1 = db1 = dw2 = db2 = None
    pool_done = False
    if use_ch:
        per = nbp // n
        ds = _new(dev, n, c)
        # dspart is [per][n][c]: one deterministic row reduction
        _lib.check(L.smaat_reduce_rows(_ptr(dspart), per, n * c, _ptr(ds), 1.0, s_), "smaat_reduce_rows")
        dw1, db1, dw2, db2, davg, dmx = _cbam_mlp_backward(L, dev, ds, sc, avg, mx, ha, hm, w1, w2, n, c, s_)
        rc = -2
        if pooled is not None:  # + the backward of the MaxPool2d that reads x too, in the same pass over dx
            dpl, dp_bs = pooled
            if dpl.dtype != x.dtype:
                dpl, dp_bs = _planes(dpl.to(x.dtype))
                pooled = (dpl, dp_bs)
            if bf:
                rc = L.smaat_cbam_bwd_final_pool_t(_ptr(dx), c * p, _ptr(davg), _ptr(dmx), _ptr(amax), _ptr(x), x_bs,
                                                   _ptr(dpl), dp_bs, n, c, h, w, 1, s_)
            else:
                rc = L.smaat_cbam_bwd_final_pool(_ptr(dx), c * p, _ptr(davg), _ptr(dmx), _ptr(amax), _ptr(x), x_bs, _ptr(dpl),
                                                 dp_bs, n, c, h, w, s_)
            if rc not in (0, -2):
                _lib.check(rc, "smaat_cbam_bwd_final_pool")
        if rc == -2:
            if bf:
                _lib.check(L.smaat_cbam_bwd_final_t(_ptr(dx), c * p, _ptr(davg), _ptr(dmx), _ptr(amax), n, c, p, 1, s_),
                           "smaat_cbam_bwd_final_t")
            else:
                _lib.check(L.smaat_cbam_bwd_final(_ptr(dx), c * p, _ptr(davg), _ptr(dmx), _ptr(amax), n, c, p, s_),
                           "smaat_cbam_bwd_final")
        pool_done = rc == 0
    if pooled is not None and not pool_done:
        dpl, dp_bs = pooled
        _maxpool2_bwd_raw(x, x_bs, dpl, dp_bs, dx, 1)
    return dx, dw1, db1, dw2, db2, dwconv, dgamma, dbeta


def cbam_eval(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, eps, out=None, pool=False):
    """Inference CBAM (eval mode, no autograd): three launches -- channel pooling, shared MLP + channel-wise
    mean/max maps, spatial conv + BatchNorm(1) on the running statistics + sigmoid + the final product -- and,
    with pool=True, MaxPool2d(2) of the un-attended input from the same loads.  Returns out or (out, pooled)."""
    _check(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, bf16=False)
    L = _lib.get()
    x, x_bs = _planes(x)
    n, c, h, w = x.shape
    p = h * w
    cr = w1.shape[0]
    s_ = _stream(x)
    w1, w2, wconv = w1.contiguous(), w2.contiguous(), wconv.contiguous()
    avg, mx = _new(x, n, c), _new(x, n, c)
    amax = _new(x, n, c, dtype=torch.int32)
    _lib.check(L.smaat_cbam_chpool(_ptr(x), x_bs, n, c, p, _ptr(avg), _ptr(mx), _ptr(amax), s_), "smaat_cbam_chpool")
    sc = _new(x, n, c)
    maps = _new(x, n, 2, h, w)
    _lib.check(L.smaat_cbam_eval_pool(_ptr(x), x_bs, _ptr(avg), _ptr(mx), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), n, c,
                                      cr, p, _ptr(sc), _ptr(maps), s_), "smaat_cbam_eval_pool")
    if out is None:
        out = _new(x, n, c, h, w)
    out_t, o_bs = _planes(out)
    assert out_t is out, "cbam: the output slice must have dense [C][H][W] planes"
    pooled = _new(x, n, c, h // 2, w // 2) if pool else None
    _lib.check(L.smaat_cbam_eval_apply(_ptr(x), x_bs, _ptr(sc), _ptr(maps), _ptr(wconv), wconv.shape[-1], _ptr(gamma),
                                       _ptr(beta), _ptr(rm), _ptr(rv), float(eps), n, c, h, w, _ptr(out), o_bs,
                                       _ptr(pooled), c * (h // 2) * (w // 2) if pool else 0, s_), "smaat_cbam_eval_apply")
    return (out, pooled) if pool else out


def cbam_eval_forked(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, eps, out, side):
    """Inference CBAM of an encoder level with the attention taken OFF the encoder's critical path (small-batch latency:
    every kernel of a batch-1 forward under-fills the chip, so the kernel chain, not the work, is the time).  The first
    launch -- channel pooling + MaxPool2d(2) in one pass -- runs on the current stream and yields `pooled`, which is all
    the next encoder level waits for; the shared MLP + channel maps and the spatial gate + product run on the HIP stream
    `side`, concurrently with the deeper encoder levels.  The caller joins (`current.wait_stream(side)`) before the decoder
    reads `out`, and keeps the returned tensors alive until then (they were allocated on the current stream).
    Returns (pooled, keepalive) or None when the one-pass pooling kernel does not take the shape (W % 4 != 0)."""
    _check(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv)
    L = _lib.get()
    x, x_bs = _planes(x)
    n, c, h, w = x.shape
    if w % 4 or h < 2 or x.dtype != torch.float32:
        return None
    p = h * w
    cr = w1.shape[0]
    w1, w2, wconv = w1.contiguous(), w2.contiguous(), wconv.contiguous()
    avg, mx = _new(x, n, c), _new(x, n, c)
    amax = _new(x, n, c, dtype=torch.int32)
    sc = _new(x, n, c)
    maps = _new(x, n, 2, h, w)
    pooled = _new(x, n, c, h // 2, w // 2)
    out_t, o_bs = _planes(out)
    assert out_t is out, "cbam: the output slice must have dense [C][H][W] planes"
    rc = L.smaat_cbam_chpool_pool_t(_ptr(x), x_bs, None, None, None, 0, _ptr(pooled), c * (h // 2) * (w // 2), n, c, h, w,
                                    _ptr(avg), _ptr(mx), _ptr(amax), _dt(x), _stream(x))
    if rc == -2:
        return None
    _lib.check(rc, "smaat_cbam_chpool_pool_t")
    side.wait_stream(torch.cuda.current_stream(x.device))
    ss = side.cuda_stream
    _lib.check(L.smaat_cbam_eval_pool(_ptr(x), x_bs, _ptr(avg), _ptr(mx), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), n, c,
                                      cr, p, _ptr(sc), _ptr(maps), ss), "smaat_cbam_eval_pool")
    _lib.check(L.smaat_cbam_eval_apply(_ptr(x), x_bs, _ptr(sc), _ptr(maps), _ptr(wconv), wconv.shape[-1], _ptr(gamma),
                                       _ptr(beta), _ptr(rm), _ptr(rv), float(eps), n, c, h, w, _ptr(out), o_bs,
                                       None, 0, ss), "smaat_cbam_eval_apply")
    return pooled, (x, avg, mx, amax, sc, maps, w1, w2, wconv, out)


class _CBAM(torch.autograd.Function):
    """out = spatial_att(channel_att(x)); either half can be switched off (standalone
    ChannelAttention / SpatialAttention modules reuse the same kernels)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, use_ch, use_sp,
                lazy_scale=None, lazy_shift=None):
        lazy = (lazy_scale, lazy_shift) if lazy_scale is not None else None
        out, saved, flags = _cbam_forward_impl(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum,
                                               eps, use_ch, use_sp, lazy=lazy)
        ctx.save_for_backward(*saved)
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, dout):
        g = _cbam_backward_impl(ctx.saved_tensors, ctx.flags, dout)
        return g + (None,) * 9


class _CBAMPoolCat(torch.autograd.Function):
    """One encoder level of SmaAt_UNet.forward (reference models/SmaAt_UNet.py:43-50): the level output
    x feeds BOTH CBAM (-> skip connection) and MaxPool2d(2) (-> next DownDS).
      cat[:, :C]  = CBAM(x)   written straight into the decoder's concatenation buffer
                              (channels [C, C + c_extra) are filled later by upsample_into)
      pooled      = maxpool2(x)
    Backward: ONE dx = cbam_bwd(dcat[:, :C]) (+)= maxpool_bwd(dpooled) -- no torch.cat copies, no
    gradient-accumulation add."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra,
                lazy_scale=None, lazy_shift=None, amax_out=None):
        L = _lib.get()
        x, x_bs = _planes(x)
        n, c, h, w = x.shape
        cat = _new(x, n, c + c_extra, h, w, dtype=x.dtype)
        lazy = (lazy_scale, lazy_shift) if lazy_scale is not None else None
        # amax_out (a list, filled in place): the maximum of the skip half of the buffer, for the decoder block that reads it
        oa = _amax_words(x, 1) if (amax_out is not None and _want_x_amax(x)) else None
        _, saved, flags = _cbam_forward_impl(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps,
                                             True, True, out=cat[:, :c], lazy=lazy, pool=(got := []), out_amax=oa)
        if oa is not None:
            amax_out.append(oa)
        if got:
            pooled = got[0]
        else:
            x, x_bs = _planes(saved[0])  # (the activated tensor when the activation was deferred)
            pooled = _maxpool2_fwd_raw(x, x_bs)
        ctx.save_for_backward(*saved)
        ctx.flags = flags
        return cat, pooled

    @staticmethod
    def backward(ctx, dcat, dpooled):
        L = _lib.get()
        saved = ctx.saved_tensors
        x = saved[0]
        n, c, h, w = x.shape
        if dcat is not None:
            pooled = _planes(dpooled) if dpooled is not None else None
            g = _cbam_backward_impl(saved, ctx.flags, dcat[:, :c], pooled=pooled)
            dx = g[0]
        else:
            g = (None,) * 8
            dx = torch.zeros_like(x)
            if dpooled is not None:
                xx, x_bs = _planes(x)
                dpooled, dp_bs = _planes(dpooled)
                _maxpool2_bwd_raw(xx, x_bs, dpooled, dp_bs, dx, 1)
        return (dx,) + tuple(g[1:]) + (None,) * 9


def cbam_pool_cat(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra, lazy=None):
    """lazy = (scale, shift): x is a deferred-activation block output (see double_conv_ds(..., defer=True))"""
    lz = lazy if lazy is not None else (None, None)
    cat, pooled = _CBAMPoolCat.apply(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, c_extra,
                                     lz[0], lz[1], (got := []))
    if got:  # max |cat[:, :C]| was left by the apply kernel (_X_AMAX)
        _note_x_amax(cat, 0, x.shape[1], got[0])
    return cat, pooled


class _UpsampleInto(torch.autograd.Function):
    """cat[:, c_off:] = pad(upsample2x(x1)) in place (reference unet_parts_depthwise_separable.py:64,76-85:
    Upsample + F.pad + the second operand of torch.cat); the first c_off channels already hold the skip."""

    @staticmethod
    def forward(ctx, cat, x1, c_off, amax_out=None):
        _check(cat, x1, acts=2)
        L = _lib.get()
        x1, x1_bs = _planes(x1)
        n, c1, h, w = x1.shape
        n2, ct, ho, wo = cat.shape
        assert n == n2 and ct == c_off + c1 and cat.is_contiguous()
        dy_, dx_ = ho - 2 * h, wo - 2 * w
        if dy_ < 0 or dx_ < 0:
            raise NotImplementedError("UpDS with a skip connection smaller than the upsampled map (negative pad)")
        pt, pl = dy_ // 2, dx_ // 2
        if x1.dtype != cat.dtype:
            # a block in front fell back to f32 storage (_bf16_storage_ok) while the skip was stored as bf16, or the other
            # way round: the upsampled map follows the buffer it is written into
            x1, x1_bs = _planes(x1.to(cat.dtype))
        done = False
        if amax_out is not None and x1.dtype == F32 and cat.dtype == F32:  # + max |cat[:, c_off:]| where the row kernel runs
            oa = _amax_words(x1, 1)
            rc = L.smaat_upsample2x_fwd_amax(_ptr(x1), x1_bs, cat.data_ptr() + 4 * c_off * ho * wo, ct * ho * wo, _ptr(oa), n, c1,
                                             h, w, ho, wo, pt, pl, _stream(x1))
            if rc == 0:
                amax_out.append(oa)
                done = True
            elif rc != -2:
                _lib.check(rc, "smaat_upsample2x_fwd_amax")
        if not done:
            _upsample_fwd_raw(x1, x1_bs, cat.data_ptr() + cat.element_size() * c_off * ho * wo, ct * ho * wo, n, c1, h, w, ho, wo,
                              pt, pl, _stream(x1))
        ctx.geom = (n, c1, h, w, c_off, ho, wo, pt, pl)
        ctx.mark_dirty(cat)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        L = _lib.get()
        n, c1, h, w, c_off, ho, wo, pt, pl = ctx.geom
        dcat = dcat.contiguous()
        dx1 = None
        if ctx.needs_input_grad[1]:
            dx1 = _new(dcat, n, c1, h, w, dtype=dcat.dtype)
            _upsample_bwd_raw(dcat.data_ptr() + dcat.element_size() * c_off * ho * wo, (c_off + c1) * ho * wo, dx1, n, c1, h,
                              w, ho, wo, pt, pl, _stream(dcat))
        return dcat, dx1, None, None


def upsample_into(cat, x1, c_off):
    if _x_amax_entry(cat) is None or not _want_x_amax(cat):
        return _UpsampleInto.apply(cat, x1, c_off, None)
    out = _UpsampleInto.apply(cat, x1, c_off, (got := []))
    if got and out is cat:  # both writers of the buffer have left their maxima (the in-place write moved the version on)
        _note_x_amax(cat, c_off, cat.shape[1], got[0], extend=True)
    return out


# --------------------------------------------------------------------------------------
# ConvTranspose2d(C, C/2, kernel_size=2, stride=2) + F.pad + cat (UpDS with bilinear=False, reference
# models/unet_parts_depthwise_separable.py:72-73,76-85): a pointwise GEMM with 4*Cout rows + a 2x2 pixel shuffle that
# writes straight into channels [c_off, c_off + Cout) of the concatenation buffer.
# --------------------------------------------------------------------------------------
def _gemm_rows(x, a2d, m):
    """out[n][m][p] = sum_c a2d[m][c] x[n][c][p] on the split GEMM when it is on, else the f32 MFMA kernel"""
    if _split_on():
        out, _, _ = _pointwise_split_raw(x, _split_planes_raw(a2d.contiguous()), None, m)
        return out
    return _pointwise_raw(x, a2d.t().contiguous(), None, m)


def _upconv_forward(x1, w, b, cat, c_off):
    L = _lib.get()
    if _is_bf(x1) or _is_bf(cat):
        raise NotImplementedError("UpDS(bilinear=False) (ConvTranspose2d up path) is built for f32 storage only; "
                                  "mixed precision takes the bilinear up path")
    x1, _ = _planes(x1)
    x1 = x1.contiguous()
    n, c1, h, wd = x1.shape
    co = w.shape[1]
    if tuple(w.shape) != (c1, co, 2, 2):
        raise ValueError(f"ConvTranspose2d weight {tuple(w.shape)} does not match an input with {c1} channels")
    _, ct, ho, wo = cat.shape
    dy_, dx_ = ho - 2 * h, wo - 2 * wd
    if dy_ < 0 or dx_ < 0:
        raise NotImplementedError("UpDS with a skip connection smaller than the upsampled map (negative pad)")
    pt, pl = dy_ // 2, dx_ // 2
    a2d = w.permute(2, 3, 1, 0).reshape(4 * co, c1)  # row (a*2+b)*Co + co = w[:, co, a, b]
    t = _gemm_rows(x1, a2d, 4 * co)
    _lib.check(L.smaat_pixel_shuffle2_fwd(_ptr(t), 4 * co * h * wd, _ptr(b), cat.data_ptr() + 4 * c_off * ho * wo,
                                          ct * ho * wo, n, co, h, wd, ho, wo, pt, pl, _stream(x1)),
               "smaat_pixel_shuffle2_fwd")
    return (n, c1, h, wd, co, ct, ho, wo, pt, pl)


def _upconv_backward(geom, x1, w, has_bias, dcat, c_off, need_dx):
    L = _lib.get()
    n, c1, h, wd, co, ct, ho, wo, pt, pl = geom
    dcat = dcat.contiguous()
    dslice = dcat[:, c_off:c_off + co]
    dt = _new(dcat, n, 4 * co, h, wd)
    _lib.check(L.smaat_pixel_shuffle2_bwd(dslice.data_ptr(), ct * ho * wo, _ptr(dt), 4 * co * h * wd, n, co, h, wd, ho,
                                          wo, pt, pl, _stream(dcat)), "smaat_pixel_shuffle2_bwd")
    a2d = w.permute(2, 3, 1, 0).reshape(4 * co, c1)
    dx1 = _gemm_rows(dt, a2d.t(), c1) if need_dx else None          # dX = A^T dT
    da = _pointwise_wgrad_raw(x1, dt, 4 * co).reshape(2, 2, co, c1)   # dA[(a,b,co)][ci]
    dw = da.permute(3, 2, 0, 1).contiguous()
    db = _channel_sum_raw(dt).view(4, co).sum(0) if has_bias else None   # over the un-padded image = over the four dt planes
    return dx1, dw, db


class _UpConvInto(torch.autograd.Function):
    """cat[:, c_off:] = pad(conv_transpose2x2(x1)) in place; the first c_off channels already hold the skip"""

    @staticmethod
    def forward(ctx, cat, x1, w, b, c_off):
        _check(cat, x1, w, b, acts=2)
        assert cat.is_contiguous() and cat.shape[1] == c_off + w.shape[1]
        ctx.geom = _upconv_forward(x1, w, b, cat, c_off)
        ctx.save_for_backward(x1, w)
        ctx.c_off, ctx.has_bias = c_off, b is not None
        ctx.mark_dirty(cat)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        x1, w = ctx.saved_tensors
        dx1, dw, db = _upconv_backward(ctx.geom, x1, w, ctx.has_bias, dcat, ctx.c_off, ctx.needs_input_grad[1])
        return dcat, dx1, dw, db, None


class _UpConvCat(torch.autograd.Function):
    """cat([x2, pad(conv_transpose2x2(x1))], dim=1)"""

    @staticmethod
    def forward(ctx, x1, x2, w, b):
        _check(x1, x2, w, b, acts=2)
        L = _lib.get()
        x2, x2_bs = _planes(x2)
        n2, c2, ho, wo = x2.shape
        co = w.shape[1]
        cat = _new(x1, n2, c2 + co, ho, wo)
        _lib.check(L.smaat_copy_planes(_ptr(x2), x2_bs, _ptr(cat), (c2 + co) * ho * wo, n2, c2 * ho * wo, 0,
                                       _stream(x1)), "smaat_copy_planes")
        ctx.geom = _upconv_forward(x1, w, b, cat, c2)
        ctx.save_for_backward(x1, w)
        ctx.c2, ctx.has_bias = c2, b is not None
        return cat

    @staticmethod
    def backward(ctx, dcat):
        L = _lib.get()
        x1, w = ctx.saved_tensors
        n, c1, h, wd, co, ct, ho, wo, pt, pl = ctx.geom
        dcat = dcat.contiguous()
        dx2 = None
        if ctx.needs_input_grad[1]:
            dx2 = _new(dcat, n, ctx.c2, ho, wo)
            _lib.check(L.smaat_copy_planes(_ptr(dcat), ct * ho * wo, _ptr(dx2), ctx.c2 * ho * wo, n, ctx.c2 * ho * wo, 0,
                                           _stream(dcat)), "smaat_copy_planes")
        dx1, dw, db = _upconv_backward(ctx.geom, x1, w, ctx.has_bias, dcat, ctx.c2, ctx.needs_input_grad[0])
        return dx1, dx2, dw, db


def upconv_into(cat, x1, w, b, c_off):
    return _UpConvInto.apply(cat, x1, w, b, c_off)


def upconv_cat(x1, x2, w, b):
    return _UpConvCat.apply(x1, x2, w, b)


def cbam(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, use_ch=True, use_sp=True, lazy=None):
    if lazy is not None:
        return _CBAM.apply(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, use_ch, use_sp,
                           lazy[0], lazy[1])
    return _CBAM.apply(x, w1, b1, w2, b2, wconv, gamma, beta, rm, rv, training, momentum, eps, use_ch, use_sp)


# --------------------------------------------------------------------------------------
# torch.ops.smaat.* registration (inference-style functional entry points; training goes
# through the autograd Functions above, which the modules call)
# --------------------------------------------------------------------------------------
def _register_torch_ops():
    try:
        lib = torch.library.Library("smaat", "DEF")
    except Exception:  # already defined (module re-import)
        return
    lib.define("dsconv(Tensor x, Tensor w_dw, Tensor? b_dw, Tensor w_pw, Tensor? b_pw, int kpl) -> Tensor")
    lib.define("pointwise(Tensor x, Tensor w, Tensor? b) -> Tensor")
    lib.define("maxpool2(Tensor x) -> Tensor")
    lib.define("upsample_cat(Tensor x1, Tensor x2) -> Tensor")

    # CompositeImplicitAutograd: the autograd.Functions inside record their own backward
    key = "CompositeImplicitAutograd"
    lib.impl("dsconv", lambda x, wd, bd, wp, bp, kpl: _DSConv.apply(x, wd, bd, wp, bp, kpl), key)
    lib.impl("pointwise", lambda x, w, b: _Pointwise.apply(x, w, b), key)
    lib.impl("maxpool2", lambda x: _MaxPool2.apply(x), key)
    lib.impl("upsample_cat", lambda a, b: _UpsampleCat.apply(a, b), key)
    globals()["_TORCH_LIB"] = lib


_register_torch_ops()
