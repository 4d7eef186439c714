"""Mixed precision (BASELINE.json configs[3], the reference's Lightning `precision="16-mixed"`) against a yardstick the
REFERENCE produced (VERDICT r3 next #1b, ADVICE r3: the old bounds were hand-picked constants).

oracle/gen_golden.py::gen_autocast ran /root/reference/models/SmaAt_UNet.py three times from the same state on the same
batch -- float32, float64, and float32 parameters under torch.autocast("cpu", torch.bfloat16) with the loss in float32
(models/regression_lightning.py:57-65) -- and stored the float32 run (logits, flat gradient, four Adam-step losses) plus
the distance of the AUTOCAST run from it.  The rule for this implementation's bf16-storage mode:

    distance(ours bf16, reference f32)  <=  1.25 x distance(reference autocast, reference f32)

on (a) logits rel-L2, (b) 1 - cosine of the flat gradient, (c) the largest relative loss deviation over the four steps
(yardstick (c): the larger of autocast-vs-f32 and f64-vs-f32 of the reference, see `yardstick`).
The f32 mode of the same code must sit at round-off distance from the reference's f32 run (1e-4 logits).

CPU (`-m "not gpu"`): through the numpy emulation of the C ABI (tests/emu_backend.py: f32 twins + one bf16 rounding per stored
tensor) at 64 x 64 -- checks the host wiring and that the yardstick is attainable.  GPU: the HIP kernels at 64 x 64 and
288 x 288."""
import json
import os

import numpy as np
import pytest
import torch

import smaat_unet_amd as S
from oracle import params as oparams
from oracle import smaat_oracle as O

FACTOR = 1.25
FACTOR_T = 2.0  # per TENSOR: the reference's per-tensor figure is ONE sample of a noisy quantity (its worst tensors sit at cos 0.6)


def load_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    sizes, scales, q = g["grad32#sizes"], g["grad32#scale"], g["grad32#f16"]
    off = np.concatenate([[0], np.cumsum(sizes)])
    g32 = {nm: q[off[i]:off[i + 1]].astype(np.float64) * scales[i] for i, nm in enumerate(meta["names"])}
    return g, meta, g32


def run_ours(meta, dev, mode, steps):
    P = oparams.make_smaat_params(12, 1, 2, 16, meta["param_seed"])
    model = S.SmaAt_UNet(12, 1)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
    model.to(dev).train().set_precision(mode)
    xn, yn = O.synthetic_precip(meta["n"], 12, meta["h"], meta["w"], seed=meta["input_seed"])
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=meta["lr"])
    first, grads, losses = None, None, []
    for _ in range(steps):
        out = model(x)
        assert out.dtype == torch.float32
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / meta["n"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if first is None:
            first = out.detach().cpu().double().numpy()
            grads = {k: p.grad.detach().cpu().double().numpy().ravel() for k, p in model.named_parameters()}
        opt.step()
        losses.append(float(loss.item()))
    return first, grads, losses


def distances(g, meta, g32, first, grads, losses):
    o32 = g["logits32"].astype(np.float64)
    a = np.concatenate([grads[k] for k in meta["names"]])
    b = np.concatenate([g32[k] for k in meta["names"]])
    l32 = g["losses32"][:len(losses)]
    return dict(logits=float(np.linalg.norm(first - o32) / np.linalg.norm(o32)),
                one_minus_cos=float(1.0 - (a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b))),
                loss=float(np.max(np.abs(np.asarray(losses) - l32) / l32)))


def yardstick(g):
    """distances of the reference's autocast run from its float32 run.  The four-step loss trajectory of the float32 run is
    itself only defined up to its distance from the float64 run (chaotic Adam steps from a random init: 4.2 % at step 4 of
    the 288 x 288 case, where autocast happens to land within 1.4 %), so the loss yardstick is the larger of the two."""
    lac, l32, l64 = g["losses_autocast"], g["losses32"], g["losses64"]
    return dict(logits=float(g["autocast/logits_vs32"]), one_minus_cos=float(g["autocast/one_minus_cos_vs32"]),
                loss=float(max(np.max(np.abs(lac - l32) / l32), np.max(np.abs(l64 - l32) / l32))))


def _zero_grad_key(k):
    """conv biases in front of a train-mode BatchNorm: exactly-zero true gradient, no direction to compare"""
    return ".double_conv." in "." + k and k.endswith(("depthwise.bias", "pointwise.bias"))


def per_tensor(names, ours, ref):
    """(1 - cosine, rel-L2) of every parameter-gradient tensor of `ours` against `ref` (dicts name -> flat float64 array);
    the single-number tensors (BatchNorm(1) affine of the spatial attentions) are judged together as one vector"""
    out, sa, sb = {}, [], []
    for k in names:
        if _zero_grad_key(k):
            continue
        a, b = ours[k], ref[k]
        if a.size == 1:
            sa.append(a[0])
            sb.append(b[0])
            continue
        nb = np.linalg.norm(b)
        out[k] = (float(1.0 - (a * b).sum() / max(np.linalg.norm(a) * nb, 1e-300)), float(np.linalg.norm(a - b) / max(nb, 1e-300)))
    if sa:
        a, b = np.array(sa), np.array(sb)
        out["<single-number tensors, as one vector>"] = (float(1.0 - (a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b))),
                                                         float(np.linalg.norm(a - b) / np.linalg.norm(b)))
    return out


def per_tensor_yardstick(g, meta):
    """the same two distances of the REFERENCE's autocast run from its float32 run, per tensor (round-5 keys of the fixture)"""
    oc, rl = g["autocast/one_minus_cos_per_tensor_vs32"], g["autocast/rel_per_tensor_vs32"]
    y = {k: (float(oc[i]), float(rl[i])) for i, k in enumerate(meta["names"]) if not _zero_grad_key(k)}
    single = [k for k in y if g["grad32#sizes"][meta["names"].index(k)] == 1]
    if single:  # (their per-tensor cosine is 0 or 2: meaningless one by one)
        for k in single:
            y.pop(k)
    return y


def check_per_tensor(names, ours, ref, yard, dump=None):
    """a wrong gradient in ONE mixed-precision layer must not pass.  Every tensor's distance from the f32
    gradient is bounded by FACTOR_T x what stock autocast does to the same tensor of the REFERENCE -- or, where the reference's
    autocast happens to leave a tensor almost untouched (it rounds at other places than bf16 storage does: e.g. the
    attention MLPs see f32 pools there), by FACTOR x the 90th percentile of the reference's own per-tensor 1 - cos (0.26-0.27
    in the fixtures; single tensors of the reference move by up to 0.41) and FACTOR x the largest per-tensor rel-L2 the
    reference shows (1.03 / 1.11: under autocast whole tensors of the reference change their norm by a factor of two).  A sign
    error is 1 - cos = 2; a dropped gradient term rotates its tensor far beyond 0.34."""
    table = per_tensor(names, ours, ref)
    floor_cos = float(np.percentile([v[0] for v in yard.values()], 90))
    floor_rel = float(max(v[1] for v in yard.values()))  # (magnitudes are the noisier half: the largest the reference shows)
    bad = []
    for k, (oc, rl) in table.items():
        yc, yr = yard.get(k, (None, None))
        if yc is None:
            continue
        if oc > FACTOR_T * max(yc, floor_cos) or rl > FACTOR_T * max(yr, floor_rel):
            bad.append((k, oc, yc, rl, yr))
    if dump and os.path.isdir(os.path.dirname(dump)):
        with open(dump, "w") as fh:
            json.dump(dict(factor=FACTOR_T, floor_one_minus_cos=floor_cos, floor_rel=floor_rel,
                           per_tensor={k: dict(one_minus_cos=v[0], rel=v[1], reference_autocast=list(yard.get(k, (None, None))))
                                       for k, v in table.items()}), fh, indent=1)
    assert not bad, bad[:8]
    return table


def check(golden_dir, name, dev, report_dir=None, f32_too=True):
    g, meta, g32 = load_case(golden_dir, name)
    ref = yardstick(g)
    first, grads, losses = run_ours(meta, dev, "bf16", meta["steps"])
    ours = distances(g, meta, g32, first, grads, losses)
    table = check_per_tensor(meta["names"], grads, g32, per_tensor_yardstick(g, meta),
                             dump=os.path.join(report_dir, f"autocast_per_tensor_{name}_{dev.type}.json") if report_dir else None)
    rep = dict(case=name, reference_autocast_vs_reference_f32=ref, ours_bf16_vs_reference_f32=ours, factor=FACTOR,
               worst_tensor_one_minus_cos=max(table.items(), key=lambda kv: kv[1][0]),
               worst_tensor_rel=max(table.items(), key=lambda kv: kv[1][1]))
    if f32_too:
        rep["ours_f32_vs_reference_f32"] = f = distances(g, meta, g32, *run_ours(meta, dev, "f32", 1))
        # the f32 mode is at round-off distance from the reference's f32 run (the fixture's float16 gradient adds 2^-11
        # per element: 1 - cos ~ 1e-7); the reference's own f32-vs-f64 figures are stored beside it
        assert f["logits"] < 1e-4 and f["one_minus_cos"] < 1e-4 and f["loss"] < 1e-4, rep
    if report_dir and os.path.isdir(report_dir):
        with open(os.path.join(report_dir, f"autocast_yardstick_{name}_{dev.type}.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    for k in ("logits", "one_minus_cos", "loss"):
        assert ours[k] <= FACTOR * ref[k], (k, rep)
    assert ours["logits"] > 1e-4, rep  # the mode really stores bf16
    return rep


def test_fixture_is_what_the_generator_documents(golden_dir):
    for name in ("autocast_bf16_n2_64", "autocast_bf16_n2_288"):
        g, meta, g32 = load_case(golden_dir, name)
        assert list(g32) == [k for k, _ in S.SmaAt_UNet(12, 1).named_parameters()]
        assert sum(v.size for v in g32.values()) == 4033537
        y = yardstick(g)
        # stock autocast moves this random-init network's logits by 0.1 .. 0.3 and its gradient direction by 0.1 .. 0.4;
        # the reference's own f32 run is 4-5 orders of magnitude closer to f64
        assert 0.05 < y["logits"] < 0.4 and 0.05 < y["one_minus_cos"] < 0.5, y
        assert float(g["f32/logits_vs64"]) < 1e-4 and float(g["f32/one_minus_cos_vs64"]) < 1e-5


def test_bf16_mode_within_the_reference_autocast_yardstick_host_emulation(golden_dir):
    from tests import emu_backend
    emu_backend.install()
    try:
        check(golden_dir, "autocast_bf16_n2_64", torch.device("cpu"))
    finally:
        emu_backend.uninstall()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["autocast_bf16_n2_64", "autocast_bf16_n2_288"])
def test_bf16_mode_within_the_reference_autocast_yardstick_gpu(golden_dir, name):
    check(golden_dir, name, torch.device("cuda:0"), report_dir="gpurun_out")


@pytest.mark.gpu
def test_bf16_storage_at_the_quoted_batch_64(golden_dir):
    """BASELINE.json configs[3] AS QUOTED: SmaAt_UNet(12, 1), 288 x 288, batch 64, bf16 activation storage (VERDICT r4 next #1b).
    No CPU run of the reference exists at this batch under autocast (the fixtures are at batch 2); the comparison is with the
    SAME model in f32 on the GPU -- itself pinned to the reference at batch 32 by tests/test_eval_and_big.py -- and the bounds
    are the reference's own autocast-vs-f32 distances at 288 x 288 (tests/golden/autocast_bf16_n2_288.npz) x 1.25: logits,
    flat gradient, loss, AND every parameter-gradient tensor by itself.  (A larger batch averages the rounding noise of
    more pixels: the batch-2 yardstick is an upper bound for batch 64, not a tuned one.)"""
    dev = torch.device("cuda:0")
    g, meta, _ = load_case(golden_dir, "autocast_bf16_n2_288")
    yard, yard_t = yardstick(g), per_tensor_yardstick(g, meta)
    P = oparams.make_smaat_params(12, 1, 2, 16, meta["param_seed"])
    xn, yn = O.synthetic_precip(64, 12, 288, 288, seed=6400)
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    res = {}
    for mode in ("f32", "bf16"):
        model = S.SmaAt_UNet(12, 1)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
        model.to(dev).train().set_precision(mode)
        out = model(x)
        assert out.dtype == torch.float32 and out.shape == (64, 1, 288, 288)
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 64
        loss.backward()
        res[mode] = (out.detach().double().cpu().numpy(), float(loss.item()),
                     {k: p.grad.detach().double().cpu().numpy().ravel() for k, p in model.named_parameters()})
        del model, out, loss
        torch.cuda.empty_cache()
    (o32, l32, g32), (ob, lb, gb) = res["f32"], res["bf16"]
    names = meta["names"]
    a, b = np.concatenate([gb[k] for k in names]), np.concatenate([g32[k] for k in names])
    ours = dict(logits=float(np.linalg.norm(ob - o32) / np.linalg.norm(o32)),
                one_minus_cos=float(1.0 - (a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b))),
                loss=abs(lb - l32) / abs(l32))
    table = check_per_tensor(names, gb, g32, yard_t, dump="gpurun_out/autocast_per_tensor_b64_288_cuda.json")
    rep = dict(case="12->1, 288x288, batch 64, bf16 storage vs f32 on the GPU", ours=ours, yardstick_batch2=yard, factor=FACTOR,
               worst_tensor_one_minus_cos=max(table.items(), key=lambda kv: kv[1][0]),
               worst_tensor_rel=max(table.items(), key=lambda kv: kv[1][1]))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/bf16_b64_288_vs_f32.json", "w") as fh:
            json.dump(rep, fh, indent=1)
    assert np.isfinite(ob).all() and ours["logits"] > 1e-4, rep  # really stored as bf16
    for k in ("logits", "one_minus_cos", "loss"):
        assert ours[k] <= FACTOR * yard[k], (k, rep)
