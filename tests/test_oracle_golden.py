"""The numpy oracle is pinned against outputs of the real reference (tests/golden/*,
produced by oracle/gen_golden.py from /root/reference).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import params as oparams
from oracle import smaat_oracle as O


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def P(ops, tag):
    pre = f"{tag}/param/"
    return {k[len(pre):]: ops[k] for k in ops.files if k.startswith(pre)}


def G(ops, tag):
    pre = f"{tag}/grad/"
    return {k[len(pre):]: ops[k] for k in ops.files if k.startswith(pre)}


@pytest.mark.parametrize("tag,kpl", [("dsconv_k2", 2), ("dsconv_k1", 1), ("dsconv_k4", 4)])
def test_dsconv(ops, tag, kpl):
    p, g = P(ops, tag), G(ops, tag)
    x, cot = ops[f"{tag}/in0"], ops[f"{tag}/cot"]
    y = O.dw3x3_fwd(x, p["depthwise.weight"], p["depthwise.bias"], kpl)
    z = O.pw1x1_fwd(y, p["pointwise.weight"], p["pointwise.bias"])
    assert rel(z, ops[f"{tag}/out"]) < 2e-6
    dy, dwp, dbp = O.pw1x1_bwd(y, p["pointwise.weight"], cot)
    dx, dwd, dbd = O.dw3x3_bwd(x, p["depthwise.weight"], dy, kpl)
    assert rel(dx, ops[f"{tag}/din0"]) < 2e-6
    assert rel(dwp, g["pointwise.weight"]) < 2e-6
    assert rel(dbp, g["pointwise.bias"]) < 2e-6
    assert rel(dwd, g["depthwise.weight"]) < 2e-6
    assert rel(dbd, g["depthwise.bias"]) < 2e-6


GENERIC_DSCONV = {  # tag: ctor arguments (the reference cases of oracle/gen_golden.py GENERIC_DSCONV)
    "dsconv_g5": dict(in_channels=4, output_channels=6, kernel_size=5, padding=2, kernels_per_layer=3),
    "dsconv_g3p0": dict(in_channels=5, output_channels=7, kernel_size=3),
    "dsconv_g1": dict(in_channels=6, output_channels=4, kernel_size=1, padding=0, kernels_per_layer=2),
    "dsconv_g7p1": dict(in_channels=3, output_channels=5, kernel_size=7, padding=1, kernels_per_layer=5),
    "dsconv_g3k3": dict(in_channels=4, output_channels=8, kernel_size=3, padding=1, kernels_per_layer=3),
    "dsconv_g3p2": dict(in_channels=2, output_channels=3, kernel_size=3, padding=2, kernels_per_layer=2),
}


@pytest.mark.parametrize("tag", sorted(GENERIC_DSCONV))
def test_dsconv_any_geometry(golden_dir, tag):
    """the general depthwise restatement against the REFERENCE module at kernel sizes / paddings / kernels_per_layer outside
    the network's 3x3 / 1 / {1, 2, 4} (models/layers.py:35-45 accepts them all)"""
    ops = np.load(os.path.join(golden_dir, "ops_generic.npz"))
    kw = GENERIC_DSCONV[tag]
    kpl, pad = kw.get("kernels_per_layer", 1), kw.get("padding", 0)
    p, g = P(ops, tag), G(ops, tag)
    x, cot = ops[f"{tag}/in0"], ops[f"{tag}/cot"]
    y = O.dwconv_fwd(x, p["depthwise.weight"], p["depthwise.bias"], kpl, pad)
    z = O.pw1x1_fwd(y, p["pointwise.weight"], p["pointwise.bias"])
    assert z.shape == ops[f"{tag}/out"].shape
    assert rel(z, ops[f"{tag}/out"]) < 2e-6
    dy, dwp, dbp = O.pw1x1_bwd(y, p["pointwise.weight"], cot)
    dx, dwd, dbd = O.dwconv_bwd(x, p["depthwise.weight"], dy, kpl, pad)
    assert rel(dx, ops[f"{tag}/din0"]) < 2e-6
    assert rel(dwp, g["pointwise.weight"]) < 2e-6
    assert rel(dbp, g["pointwise.bias"]) < 2e-6
    assert rel(dwd, g["depthwise.weight"]) < 2e-6
    assert rel(dbd, g["depthwise.bias"]) < 2e-6
    if kw["kernel_size"] == 3 and pad == 1:  # the 3x3 restatement is the same function at this geometry
        assert np.array_equal(y, O.dw3x3_fwd(x, p["depthwise.weight"], p["depthwise.bias"], kpl))


@pytest.mark.parametrize("tag,kpl", [("doubleconv", 2), ("doubleconv_mid", 2)])
def test_doubleconv(ops, tag, kpl):
    p, g = P(ops, tag), G(ops, tag)
    p = {"m." + k: v for k, v in p.items()}
    tape = O.Tape()
    out = O.double_conv_ds_fwd(p, "m", ops[f"{tag}/in0"], kpl, tape)
    assert rel(out, ops[f"{tag}/out"]) < 5e-6
    Gd = {}
    dx = O.double_conv_ds_bwd(p, Gd, "m", kpl, tape, ops[f"{tag}/cot"])
    assert rel(dx, ops[f"{tag}/din0"]) < 2e-4
    for k, v in g.items():
        if k.endswith("depthwise.bias") or k.endswith("pointwise.bias"):
            # true gradient is exactly 0 (bias feeds a train-mode BN): roundoff only
            wn = np.linalg.norm(g[k.replace("bias", "weight")])
            assert np.abs(Gd["m." + k]).max() <= 1e-3 * wn + 1e-5
            continue
        assert rel(Gd["m." + k], v) < 2e-4, k
    # running stats (momentum 0.1, unbiased var)
    t = tape.d["m#0"]
    cnt = t["z"].shape[0] * t["z"].shape[2] * t["z"].shape[3]
    rm, rv = O.bn_running_update(np.zeros_like(t["mean"]), np.ones_like(t["var"]), t["mean"], t["var"], cnt)
    # the fixture modules had default running stats (0 / 1) before the step
    assert rel(rm, ops[f"{tag}/after/double_conv.1.running_mean"]) < 1e-5
    assert rel(rv, ops[f"{tag}/after/double_conv.1.running_var"]) < 1e-5


@pytest.mark.parametrize("tag", ["down", "down_odd"])
def test_down(ops, tag):
    p = {"m." + k: v for k, v in P(ops, tag).items()}
    tape = O.Tape()
    out = O.down_fwd(p, "m", ops[f"{tag}/in0"], 2, tape)
    assert rel(out, ops[f"{tag}/out"]) < 5e-6
    Gd = {}
    dx = O.down_bwd(p, Gd, "m", 2, tape, ops[f"{tag}/cot"])
    assert rel(dx, ops[f"{tag}/din0"]) < 2e-4


@pytest.mark.parametrize("tag", ["up", "up_pad"])
def test_up(ops, tag):
    p = {"m." + k: v for k, v in P(ops, tag).items()}
    tape = O.Tape()
    out = O.up_fwd(p, "m", ops[f"{tag}/in0"], ops[f"{tag}/in1"], 2, tape)
    assert rel(out, ops[f"{tag}/out"]) < 5e-6
    Gd = {}
    dx1, dx2 = O.up_bwd(p, Gd, "m", 2, tape, ops[f"{tag}/cot"])
    assert rel(dx1, ops[f"{tag}/din0"]) < 2e-4
    assert rel(dx2, ops[f"{tag}/din1"]) < 2e-4


def test_upsample_alone():
    import torch
    x = np.random.default_rng(0).standard_normal((2, 3, 7, 9)).astype(np.float32)
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    xt = torch.from_numpy(x).requires_grad_(True)
    y = up(xt)
    cot = torch.randn_like(y)
    (y * cot).sum().backward()
    assert rel(O.upsample2x_fwd(x), y.detach().numpy()) < 1e-6
    assert rel(O.upsample2x_bwd(x.shape, cot.numpy()), xt.grad.numpy()) < 1e-6


def test_channel_att(ops):
    tag = "chatt"
    p, g = P(ops, tag), G(ops, tag)
    x = ops[f"{tag}/in0"]
    y, c = O.channel_att_fwd(x, p["MLP.1.weight"], p["MLP.1.bias"], p["MLP.3.weight"], p["MLP.3.bias"])
    assert rel(y, ops[f"{tag}/out"]) < 2e-6
    dx, dw1, db1, dw2, db2 = O.channel_att_bwd(x, p["MLP.1.weight"], p["MLP.1.bias"], p["MLP.3.weight"],
                                               p["MLP.3.bias"], c, ops[f"{tag}/cot"])
    assert rel(dx, ops[f"{tag}/din0"]) < 1e-5
    assert rel(dw1, g["MLP.1.weight"]) < 1e-5
    assert rel(db1, g["MLP.1.bias"]) < 1e-5
    assert rel(dw2, g["MLP.3.weight"]) < 1e-5
    assert rel(db2, g["MLP.3.bias"]) < 1e-5


def test_spatial_att(ops):
    tag = "spatt"
    p, g = P(ops, tag), G(ops, tag)
    x = ops[f"{tag}/in0"]
    y, c = O.spatial_att_fwd(x, p["conv.weight"], p["bn.weight"], p["bn.bias"])
    assert rel(y, ops[f"{tag}/out"]) < 5e-6
    dx, dwc, dg, db = O.spatial_att_bwd(x, p["conv.weight"], p["bn.weight"], c, ops[f"{tag}/cot"])
    assert rel(dx, ops[f"{tag}/din0"]) < 1e-4
    assert rel(dwc, g["conv.weight"]) < 1e-4
    assert rel(dg, g["bn.weight"]) < 1e-4
    assert rel(db, g["bn.bias"]) < 1e-4


@pytest.mark.parametrize("tag", ["cbam", "cbam_small"])
def test_cbam(ops, tag):
    p = {"m." + k: v for k, v in P(ops, tag).items()}
    g = G(ops, tag)
    tape = O.Tape()
    y = O.cbam_fwd(p, "m", ops[f"{tag}/in0"], tape)
    assert rel(y, ops[f"{tag}/out"]) < 5e-6
    Gd = {}
    dx = O.cbam_bwd(p, Gd, "m", tape, ops[f"{tag}/cot"])
    assert rel(dx, ops[f"{tag}/din0"]) < 1e-4
    for k, v in g.items():
        assert rel(Gd["m." + k], v) < 2e-4, k


def test_outconv(ops):
    tag = "outconv"
    p, g = P(ops, tag), G(ops, tag)
    x = ops[f"{tag}/in0"]
    assert rel(O.pw1x1_fwd(x, p["conv.weight"], p["conv.bias"]), ops[f"{tag}/out"]) < 2e-6
    dx, dw, db = O.pw1x1_bwd(x, p["conv.weight"], ops[f"{tag}/cot"])
    assert rel(dx, ops[f"{tag}/din0"]) < 2e-6 and rel(dw, g["conv.weight"]) < 2e-6 and rel(db, g["conv.bias"]) < 2e-6


def check_summary(store, tag, arr, tol):
    a = np.asarray(arr, np.float32)
    assert a.size == int(store[tag + "#n"])
    ref_l2 = float(store[tag + "#l2"])
    if tag + "#full" in store.files:
        ref = store[tag + "#full"]
        err = np.linalg.norm(a.astype(np.float64) - ref) / max(ref_l2, 1e-30)
    else:
        idx = store[tag + "#idx"]
        ref = store[tag + "#vals"]
        got = a.ravel()[idx]
        err = np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        l2 = np.sqrt((a.astype(np.float64) ** 2).sum())
        assert abs(l2 - ref_l2) <= 10 * tol * max(ref_l2, 1e-30), (tag, l2, ref_l2)
    return err


@pytest.mark.parametrize("name", ["unet_12x1_n2_32", "unet_12x1_n2_64x48", "unet_3x21_n1_32"])
def test_unet_end_to_end(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    Pm = oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], 2, 16, meta["param_seed"])
    logits, tape, acts = O.smaat_unet_fwd(Pm, g["x"])
    assert rel(logits, g["logits"]) < 1e-4
    for k in ["x1", "x1Att", "x3", "x5", "x5Att", "u1", "u4"]:
        assert check_summary(g, "act/" + k, acts[k], 1e-4) < 1e-4, k
    if meta["loss"] == "mse":
        loss, dl = O.mse_sum_over_batch(logits, g["target"])
        assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    else:
        dl = g["target"]
    Gd, dx = O.smaat_unet_bwd(Pm, tape, dl)
    # end-to-end gradient noise floor of the reference itself is ~2-5e-3
    worst = 0.0
    for k in Gd:
        if ".double_conv." in k and (k.endswith("depthwise.bias") or k.endswith("pointwise.bias")):
            continue  # exact-zero gradients, roundoff only
        e = check_summary(g, "grad/" + k, Gd[k], 2e-2)
        worst = max(worst, e)
        assert e < 2e-2, (k, e)
    assert check_summary(g, "dx", dx, 2e-2) < 2e-2
    # BN running stats after one step
    t = tape.d["inc#0"]
    cnt = t["z"].shape[0] * t["z"].shape[2] * t["z"].shape[3]
    rm, rv = O.bn_running_update(Pm["inc.double_conv.1.running_mean"], Pm["inc.double_conv.1.running_var"],
                                 t["mean"], t["var"], cnt)
    assert rel(rm, g["after/inc.double_conv.1.running_mean"]) < 1e-4
    assert rel(rv, g["after/inc.double_conv.1.running_var"]) < 1e-4


def test_state_dict_keys(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_keys.json")) as f:
        ref = json.load(f)
    for tag, (nc, ncl) in {"12_1": (12, 1), "3_21": (3, 21)}.items():
        mine = [[k, list(s)] for k, s in oparams.smaat_unet_keys(nc, ncl)]
        assert mine == ref[tag]
        assert len(mine) == 214


@pytest.mark.parametrize("name", ["unet_12x1_n2_32", "unet_12x1_n2_64x48"])
def test_torch_functional_port_vs_golden(golden_dir, name):
    """oracle/torch_ref.py (bench.py's cpu_baseline port) reproduces the reference."""
    import torch
    from oracle import torch_ref
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    P = torch_ref.params_from_numpy(oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], 2, 16,
                                                               meta["param_seed"]))
    loss, logits = torch_ref.train_step(P, torch.from_numpy(g["x"]), torch.from_numpy(g["target"]))
    assert rel(logits.numpy(), g["logits"]) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for k in ["outc.conv.weight", "up4.conv.double_conv.3.pointwise.weight", "inc.double_conv.0.depthwise.weight",
              "cbam3.channel_att.MLP.1.weight"]:
        assert check_summary(g, "grad/" + k, P[k].grad.numpy(), 2e-2) < 2e-2, k
