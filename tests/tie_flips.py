"""TEST INFRASTRUCTURE: attribution of a whole-network gradient deviation to ReLU decisions at a tie.

The reference-generated network fixtures that are NOT tie-free (tests/golden/unet_*: every correct fp32 implementation
takes thousands of ReLU decisions on pre-activations within round-off of zero) bound gradients by a multiple of the
reference's own fp32-vs-fp64 figure.  On planes of a few pixels ONE such decision is a visible fraction of a BatchNorm's
samples (a 4 x 3 plane at batch 3: 1/36), and a run whose forward differs from another correct run by f32 round-off can
land on the other side of a tie: every gradient upstream of it then moves by 5e-3 ... 1e-2 (profiles/r5/
f16_split_small_plane_counterexamples.txt shows one such element).  `attribute()` decides whether a bound violation of a run
is of that kind, by comparing it with a second run of THIS implementation on the exact three-term split:

  * the exact run satisfies the bound;
  * the two runs take different ReLU decisions somewhere, and EVERY differing decision sits on a pre-activation that is
    within `tie` (relative to the rms of its tensor) of zero in BOTH runs.

Anything else -- a difference away from a tie, or no difference at all -- is a real failure.

Round 6: attribution alone says WHY the run left the bound, not that its gradients are right given the
decisions it took.  `check_against_masked_oracle()` therefore re-derives the fp64 anchor with the run's OWN recorded ReLU
decisions imposed (oracle/smaat_oracle.py train_step_loss_and_grads(relu_masks=...)) and holds every gradient tensor to the
ordinary per-tensor bound against THAT anchor: an accepted tie flip is followed by an oracle check, not by a shrug."""
import contextlib

import numpy as np
import torch

from smaat_unet_amd import ops


@contextlib.contextmanager
def record_pre_activations(store):
    """store.append((z, st)) for every DepthwiseSeparableConv + BatchNorm + ReLU half the network runs (z: pre-BatchNorm
    tensor, st rows: mean, invstd, scale, shift); the ReLU's argument is z * scale + shift"""
    orig = ops._half_forward

    def wrapped(*a, **k):
        r = orig(*a, **k)
        store.append((r[1].detach().float().clone(), r[2].detach().clone()))
        return r

    ops._half_forward = wrapped
    try:
        yield
    finally:
        ops._half_forward = orig


def differing_decisions(a, b):
    """[(layer index, element index, value in run a, value in run b, rms of the tensor)] for every ReLU argument whose sign
    differs between the two recordings"""
    out = []
    for i, ((za, sa), (zb, sb)) in enumerate(zip(a, b)):
        pa = za * sa[2][None, :, None, None] + sa[3][None, :, None, None]
        pb = zb * sb[2][None, :, None, None] + sb[3][None, :, None, None]
        d = (pa > 0) != (pb > 0)
        if bool(d.any()):
            rms = float(pa.double().pow(2).mean().sqrt())
            for idx in d.nonzero().tolist():
                t = tuple(idx)
                out.append((i, t, float(pa[t]), float(pb[t]), rms))
    return out


class BoundViolation(AssertionError):
    """the per-tensor gradient bound of a network fixture is violated (the only failure a tie flip may explain: logits,
    loss and running statistics are asserted with plain AssertionErrors and are never forgiven -- ADVICE r5)"""


def relu_masks(rec):
    """the ReLU decisions a recorded run took: one boolean numpy array per DoubleConvDS half, in execution order"""
    return [((z * st[2][None, :, None, None] + st[3][None, :, None, None]) > 0).cpu().numpy() for z, st in rec]


def check_against_masked_oracle(P, x, target, loss, rec, named_grads, noise_of, factor, floor=5e-3, kpl=2, cotangent=None,
                                skip=lambda k: False):
    """fp64 oracle with the decisions of `rec` imposed; every gradient tensor of named_grads (name, array) must be within
    max(factor * noise_of(name), floor) of it (rel-L2), the single-number gradients judged together as one vector.
    Returns (violations, anchor gradients)."""
    from oracle import smaat_oracle as O
    P64 = {k: np.asarray(v, np.float64) for k, v in P.items()}
    tgt = np.asarray(target) if loss == "ce" else np.asarray(target, np.float64)
    _, G, dx, _ = O.train_step_loss_and_grads(P64, np.asarray(x, np.float64), tgt, kpl, relu_masks=relu_masks(rec), loss=loss,
                                              cotangent=None if cotangent is None else np.asarray(cotangent, np.float64))
    bad, so, sr = [], [], []
    for k, gk in named_grads:
        if skip(k):
            continue
        gk, ref = np.asarray(gk, np.float64), np.asarray(G[k], np.float64)
        if gk.size == 1:
            so.append(float(gk.ravel()[0]))
            sr.append(float(ref.ravel()[0]))
            continue
        e = np.linalg.norm(gk - ref) / max(np.linalg.norm(ref), 1e-30)
        if e > max(factor * noise_of(k), floor):
            bad.append((k, float(e), float(noise_of(k))))
    if so:
        e = np.linalg.norm(np.array(so) - np.array(sr)) / max(np.linalg.norm(sr), 1e-30)
        if e > max(factor * max(noise_of("<single>"), 0.0), 2e-2):
            bad.append(("<single-number gradients as one vector>", float(e), 0.0))
    G["<dx>"] = dx
    return sorted(bad, key=lambda t: -t[1]), G


def attribute(run, tie=2e-4, flag="f16_split", sink=None):
    """run(on: bool, store) -> None or raises AssertionError (the bound check of the test, on a fresh model with the feature
    named by `flag` -- a boolean field of smaat_unet_amd.ops.policy whose two settings differ at f32 round-off level: the two-term
    fp16 split (default), or cbam_three_pass, whose kernels add the channels of the attention maps in another order -- on /
    off; it must run its forward inside `record_pre_activations(store)`).
    Only a BoundViolation of the default run can be forgiven; any other AssertionError (logits, loss, running statistics)
    propagates.  sink (dict): receives rec = the recording of the default run (for check_against_masked_oracle).
    Returns the list of tie flips when the violation of run(True) is attributable to them; raises otherwise."""
    prev = getattr(ops.policy, flag)
    rec = {}
    err = {}
    try:
        for f16 in (True, False):
            setattr(ops.policy, flag, f16)
            ops.invalidate_weight_images()
            rec[f16] = []
            try:
                run(f16, rec[f16])
                err[f16] = None
            except BoundViolation as e:  # noqa: PERF203
                err[f16] = e
    finally:
        setattr(ops.policy, flag, prev)
        ops.invalidate_weight_images()
    if sink is not None:
        sink["rec"] = rec[True]
    if err[True] is None:
        return []
    if err[False] is not None:
        raise err[False]  # the exact path violates the bound as well: not a question of ties
    flips = differing_decisions(rec[True], rec[False])
    if not flips:
        raise err[True]
    off_tie = [f for f in flips if max(abs(f[2]), abs(f[3])) > tie * f[4]]
    if off_tie:
        raise AssertionError(f"ReLU decisions differ AWAY from a tie: {off_tie[:4]}") from err[True]
    return flips
