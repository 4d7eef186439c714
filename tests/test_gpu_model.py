"""GPU: the module-level drop-ins (HIP path through the C ABI) against
  * the reference-generated golden fixtures (tests/golden, made by oracle/gen_golden.py),
  * the numpy oracle on seeded inputs at sizes it finishes in seconds,
  * size-independent properties at BASELINE.json's full size (288x288, batch 32).
Tolerances (north_star: 1e-4 relative fp32 on outputs): forward rel-L2 <= 1e-4;
per-op gradients <= 3e-4; end-to-end gradients <= 2e-2 (the reference disagrees with ITSELF
at 2-5e-3 end to end)."""
import json
import os

import numpy as np
import pytest
import torch

import smaat_unet_amd as S
from oracle import params as oparams
from oracle import smaat_oracle as O
from tests.test_host_emu import VARIANTS, check_summary, rel, run_variant

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def run_case(ops, tag, mod, tol_out=2e-5, tol_grad=3e-4, zero_bias=True, tol_stat=1e-5):
    pre = f"{tag}/param/"
    mod.load_state_dict({k[len(pre):]: torch.from_numpy(ops[k]) for k in ops.files if k.startswith(pre)})
    mod.to(DEV).train()
    ins, i = [], 0
    while f"{tag}/in{i}" in ops.files:
        ins.append(torch.from_numpy(ops[f"{tag}/in{i}"]).to(DEV).requires_grad_(True))
        i += 1
    out = mod(*ins)
    assert rel(out.detach().float().cpu().numpy(), ops[f"{tag}/out"]) < tol_out  # (.float(): bf16 outputs in mixed precision)
    (out * torch.from_numpy(ops[f"{tag}/cot"]).to(DEV)).sum().backward()
    for i, x in enumerate(ins):
        assert rel(x.grad.cpu().numpy(), ops[f"{tag}/din{i}"]) < tol_grad, f"din{i}"
    for k, p in mod.named_parameters():
        ref = ops[f"{tag}/grad/{k}"]
        if zero_bias and ".double_conv." in "." + k and (k.endswith("depthwise.bias") or k.endswith("pointwise.bias")):
            wn = np.linalg.norm(ops[f"{tag}/grad/{k.replace('bias', 'weight')}"])
            assert np.abs(p.grad.cpu().numpy()).max() <= 1e-3 * wn + 1e-5, k
            continue
        assert rel(p.grad.cpu().numpy(), ref) < tol_grad, k
    for k, v in mod.state_dict().items():
        if "running" in k:
            assert rel(v.cpu().numpy(), ops[f"{tag}/after/{k}"]) < tol_stat, k


@pytest.mark.parametrize("tag,ctor,zb", [
    ("dsconv_k2", lambda: S.DepthwiseSeparableConv(6, 10, kernel_size=3, padding=1, kernels_per_layer=2), False),
    ("dsconv_k1", lambda: S.DepthwiseSeparableConv(5, 7, kernel_size=3, padding=1, kernels_per_layer=1), False),
    ("dsconv_k4", lambda: S.DepthwiseSeparableConv(3, 8, kernel_size=3, padding=1, kernels_per_layer=4), False),
    ("doubleconv", lambda: S.DoubleConvDS(6, 16, kernels_per_layer=2), True),
    ("doubleconv_mid", lambda: S.DoubleConvDS(8, 4, mid_channels=12, kernels_per_layer=2), True),
    ("down", lambda: S.DownDS(6, 12, kernels_per_layer=2), True),
    ("down_odd", lambda: S.DownDS(4, 8, kernels_per_layer=2), True),
    ("up", lambda: S.UpDS(16, 6, bilinear=True, kernels_per_layer=2), True),
    ("up_pad", lambda: S.UpDS(8, 4, bilinear=True, kernels_per_layer=2), True),
    ("chatt", lambda: S.ChannelAttention(32, reduction_ratio=16), True),
    ("spatt", lambda: S.SpatialAttention(kernel_size=7), True),
    ("cbam", lambda: S.CBAM(32, reduction_ratio=16), True),
    ("cbam_small", lambda: S.CBAM(64, reduction_ratio=16), True),
    ("outconv", lambda: S.OutConv(16, 3), False),
])
def test_module_vs_reference_golden(ops, tag, ctor, zb):
    run_case(ops, tag, ctor(), zero_bias=zb)


GENERIC_DSCONV = {  # tag: ctor arguments (the reference cases of oracle/gen_golden.py GENERIC_DSCONV)
    "dsconv_g5": dict(in_channels=4, output_channels=6, kernel_size=5, padding=2, kernels_per_layer=3),
    "dsconv_g3p0": dict(in_channels=5, output_channels=7, kernel_size=3),
    "dsconv_g1": dict(in_channels=6, output_channels=4, kernel_size=1, padding=0, kernels_per_layer=2),
    "dsconv_g7p1": dict(in_channels=3, output_channels=5, kernel_size=7, padding=1, kernels_per_layer=5),
    "dsconv_g3k3": dict(in_channels=4, output_channels=8, kernel_size=3, padding=1, kernels_per_layer=3),
    "dsconv_g3p2": dict(in_channels=2, output_channels=3, kernel_size=3, padding=2, kernels_per_layer=2),
}


@pytest.mark.parametrize("tag", sorted(GENERIC_DSCONV) + ["doubleconv_k3"])
def test_any_geometry_modules_vs_reference_golden(golden_dir, tag):
    """the reference's DepthwiseSeparableConv accepts any kernel_size / padding / kernels_per_layer (models/layers.py:35-45):
    outside the fused configuration the module runs the general depthwise kernels (smaat_dwconv_*_any) + the pointwise GEMM;
    a DoubleConvDS at kernels_per_layer = 3 runs half by half"""
    g = np.load(os.path.join(golden_dir, "ops_generic.npz"))
    if tag == "doubleconv_k3":
        run_case(g, tag, S.DoubleConvDS(5, 8, kernels_per_layer=3), zero_bias=True)
    else:
        run_case(g, tag, S.DepthwiseSeparableConv(**GENERIC_DSCONV[tag]), zero_bias=False)


def test_network_at_kernels_per_layer_3_trains():
    """SmaAt_UNet(kernels_per_layer=3): constructible in the reference, outside the fused kernels -- the network runs (general
    path), matches the same weights evaluated by torch's own convolutions, and produces finite gradients for every parameter"""
    import torch.nn.functional as F
    torch.manual_seed(0)
    model = S.SmaAt_UNet(4, 2, kernels_per_layer=3).to(DEV).train()
    x = torch.randn(2, 4, 32, 32, device=DEV)
    out = model(x)
    assert out.shape == (2, 2, 32, 32)
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    # first half of the stem against torch: depthwise (groups) conv -> 1x1 conv
    c0 = model.inc.double_conv[0]
    cpu = lambda t: t.detach().cpu()  # noqa: E731  (torch's CPU convolutions: no MIOpen dependency on the box)
    ref = F.conv2d(F.conv2d(cpu(x), cpu(c0.depthwise.weight), cpu(c0.depthwise.bias), padding=1, groups=4),
                   cpu(c0.pointwise.weight), cpu(c0.pointwise.bias))
    got = cpu(c0(x))
    assert float((got - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.parametrize("name", VARIANTS)
def test_sibling_networks_vs_reference_golden(golden_dir, name):
    """UNetDS / UNetDSAttention4CBAMs, kernels_per_layer 1, 2, 4, a size that needs the UpDS padding: tie-free
    fixtures, every gradient tensor within max(2 x reference-fp32 error, 1e-4) of the fp64 anchors"""
    report = {}
    try:
        run_variant(golden_dir, name, DEV, report=report)
    finally:
        if os.path.isdir("gpurun_out"):
            with open(f"gpurun_out/variant_{name}.json", "w") as f:
                json.dump(report, f, indent=1, default=float)


def _per_tensor_in_the_autocast_class(model, ga, gb, fixture="autocast_bf16_n2_64"):
    """every parameter-gradient tensor of `ga` against `gb` (flat vectors in model.parameters() order), bounded tensor by
    tensor by what stock autocast does to the REFERENCE (tests/test_autocast_yardstick.py::check_per_tensor): a wrong
    gradient in one mixed-precision layer cannot hide in the norm of the flat vector"""
    from tests.test_autocast_yardstick import check_per_tensor, load_case, per_tensor_yardstick
    g, meta, _ = load_case(os.path.join(os.path.dirname(__file__), "golden"), fixture)
    names = [k for k, _ in model.named_parameters()]
    assert names == meta["names"]
    sizes = [p.numel() for p in model.parameters()]
    off = np.concatenate([[0], np.cumsum(sizes)])
    da = {k: ga[off[i]:off[i + 1]].double().cpu().numpy() for i, k in enumerate(names)}
    db = {k: gb[off[i]:off[i + 1]].double().cpu().numpy() for i, k in enumerate(names)}
    return check_per_tensor(names, da, db, per_tensor_yardstick(g, meta))


def _load_model(meta):
    kpl = meta.get("kpl", 2)
    P = oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], kpl, 16, meta["param_seed"])
    model = S.SmaAt_UNet(meta["n_channels"], meta["n_classes"], kernels_per_layer=kpl)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
    return model.to(DEV).train(), P


@pytest.mark.parametrize("name,policy", [("unet_12x1_n2_32", "auto"), ("unet_12x1_n2_64x48", "auto"),
                                         ("unet_3x21_n1_32", "auto"), ("unet_12x1_n2_32", "all"),
                                         ("unet_12x1_n2_64x48", "all"),
                                         ("unet_4x2_k3_n2_32", "auto")])  # kernels_per_layer = 3: reference fixture, general path
def test_unet_vs_reference_golden(golden_dir, name, policy, monkeypatch):
    from smaat_unet_amd import ops as _ops
    monkeypatch.setattr(_ops.policy, "split_policy", policy)  # "all": every supported layer on the bf16-split path
    if name != "unet_12x1_n2_64x48":
        return _unet_vs_reference_golden(golden_dir, name)
    # bottleneck planes of 4 x 3 pixels at batch 2: a violation of the per-tensor bound is accepted IF tests/tie_flips.py
    # attributes it to ReLU decisions at a tie between this run and the run with the gate / main / final attention kernels
    # of the same build (round 5: the three-pass attention kernels add the channels of the pooled maps in chunks; with eight
    # chunks this fixture lands on the other side of one tie -- scripts/probes/cbam_split_rounding.py)
    from tests.tie_flips import attribute, check_against_masked_oracle, record_pre_activations
    cap, sink = {}, {}

    def run(on, store):
        with record_pre_activations(store):
            _unet_vs_reference_golden(golden_dir, name, capture=cap if on else None)
    flips = attribute(run, flag="cbam_three_pass", sink=sink)
    if flips:
        # round 6: an accepted flip is followed by an ORACLE check -- the fp64 anchor re-derived with the decisions THIS run took,
        # every gradient tensor held to the ordinary bound against it
        g = cap["g"]
        bad, _ = check_against_masked_oracle(cap["P"], g["x"], g["target"], "mse" if cap["meta"]["loss"] == "mse" else "dot",
                                             sink["rec"], cap["grads"], lambda k: float(g["noise/" + k]) if "noise/" + k in g.files else 0.0,
                                             NOISE_FACTOR_SMALL, cotangent=g["target"],
                                             skip=lambda k: ".double_conv." in "." + k and k.endswith(("depthwise.bias", "pointwise.bias")))
        assert not bad, ("gradients do not match the fp64 oracle under this run's own ReLU decisions", bad[:6])
    if flips and os.path.isdir("gpurun_out"):
        with open(f"gpurun_out/golden_{name}_{policy}_tie_flips.json", "w") as f:
            json.dump([dict(half=i, element=list(e), three_pass=a, sequence=b, rms=r) for i, e, a, b, r in flips], f, indent=1)


from tests.test_host_emu import SMALL_NET_NOISE_FACTOR as NOISE_FACTOR_SMALL  # noqa: E402  (2 x since round 6)


def _unet_vs_reference_golden(golden_dir, name, capture=None):
    from tests.tie_flips import BoundViolation
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    model, _ = _load_model(meta)
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    logits = model(x)
    e = rel(logits.detach().cpu().numpy(), g["logits"])
    assert e < 1e-4, e
    tgt = torch.from_numpy(g["target"]).to(DEV)
    if meta["loss"] == "mse":  # reference: models/regression_lightning.py:57-65
        loss = torch.nn.functional.mse_loss(logits.squeeze(1), tgt, reduction="sum") / meta["n"]
        assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    else:
        loss = (logits * tgt).sum()
    loss.backward()
    # per tensor against the fp64 anchor of the reference: no worse than 3 x the reference's own fp32 error on that
    # tensor (floor 5e-3: one ReLU flip at forward round-off level), the rule of the benchmark-size fixtures
    # (tests/test_eval_and_big.py run_big) -- not one flat 2e-2 for every tensor
    from tests.test_host_emu import check_param_grads
    grads = [(k, p.grad.cpu().numpy()) for k, p in model.named_parameters()]
    if capture is not None:
        capture.update(g=g, meta=meta, grads=grads, P=oparams.make_smaat_params(meta["n_channels"], meta["n_classes"], meta.get("kpl", 2),
                                                                                 16, meta["param_seed"]))
    bad = check_param_grads(g, grads)
    if bad:
        raise BoundViolation(bad[:6])
    if not check_summary(g, "dx64", x.grad.cpu().numpy()) < max(NOISE_FACTOR_SMALL * float(g["noise/dx"]), 5e-3):
        raise BoundViolation("dx")
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("after/"):
            assert rel(sd[k[6:]].cpu().numpy(), g[k]) < 1e-4, k


def test_unet_vs_oracle_288():
    """one 288x288 frame, forward + backward, against the oracle.  End-to-end gradients are
    judged the way SURVEY.md 8(c) prescribes: per tensor, our error against the FP64 oracle
    must be no worse than 2x the error that FP32 CPU arithmetic itself shows against FP64
    (numpy fp32 oracle and the ATen-CPU port oracle/torch_ref.py; floor 1e-3)."""
    from oracle import torch_ref
    meta = dict(n_channels=12, n_classes=1, param_seed=3)
    model, P = _load_model(meta)
    xn, yn = O.synthetic_precip(1, 12, 288, 288, seed=1234)
    loss_o, G32, dx32, acts = O.train_step_loss_and_grads(P, xn, yn)
    P64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in P.items()}
    _, G64, dx64, acts64 = O.train_step_loss_and_grads(P64, xn.astype(np.float64), yn.astype(np.float64))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Pt = torch_ref.params_from_numpy(P)
    torch_ref.train_step(Pt, torch.from_numpy(xn), torch.from_numpy(yn))
    x = torch.from_numpy(xn).to(DEV).requires_grad_(True)
    logits = model(x)
    assert rel(logits.detach().cpu().numpy(), acts["logits"]) < 1e-4
    assert rel(logits.detach().cpu().numpy(), acts64["logits"]) < 1e-4
    loss = torch.nn.functional.mse_loss(logits.squeeze(1), torch.from_numpy(yn).to(DEV), reduction="sum") / 1
    assert abs(loss.item() - float(loss_o)) < 1e-4 * abs(float(loss_o))
    loss.backward()
    bad, table = [], {}
    skip = lambda k: ".double_conv." in k and k.endswith(("depthwise.bias", "pointwise.bias"))  # noqa: E731
    for k, p in model.named_parameters():
        if skip(k):
            continue
        ours = rel(p.grad.cpu().numpy(), G64[k])
        ref = max(rel(G32[k], G64[k]), rel(Pt[k].grad.numpy(), G64[k]))
        table[k] = (ours, rel(G32[k], G64[k]), rel(Pt[k].grad.numpy(), G64[k]))
        if ours > max(3.0 * ref, 5e-3):
            bad.append((k, ours, ref))
    flat_o = np.concatenate([G64[k].ravel() for k, _ in model.named_parameters() if not skip(k)])
    flat_m = np.concatenate([p.grad.cpu().numpy().ravel() for k, p in model.named_parameters() if not skip(k)])
    flat_t = np.concatenate([Pt[k].grad.numpy().ravel() for k, _ in model.named_parameters() if not skip(k)])
    summary = dict(flat_ours_vs_fp64=rel(flat_m, flat_o), flat_aten_cpu_vs_fp64=rel(flat_t, flat_o),
                   worst_ours=max(table.items(), key=lambda kv: kv[1][0]),
                   worst_aten=max(table.items(), key=lambda kv: kv[1][2]))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/grad_errors_288.json", "w") as f:
            json.dump(dict(summary=summary, per_tensor=table), f, indent=1, default=float)
    assert not bad, (bad[:8], summary)
    assert rel(x.grad.cpu().numpy(), dx64) < max(2.0 * rel(dx32, dx64), 1e-3)
    assert summary["flat_ours_vs_fp64"] < max(2.0 * summary["flat_aten_cpu_vs_fp64"], 3e-3)


def test_full_size_properties():
    """BASELINE config 2 size (batch 32, 12x288x288): properties that need no oracle run."""
    torch.manual_seed(0)
    model = S.SmaAt_UNet(12, 1).to(DEV)
    xn, yn = O.synthetic_precip(32, 12, 288, 288, seed=7)
    x, y = torch.from_numpy(xn).to(DEV), torch.from_numpy(yn).to(DEV)
    # (a) eval mode is per-sample: a batch of 32 == two batches of 16 (different tile configs, same k order)
    model.train()
    with torch.no_grad():
        model(x[:4])  # move running stats off their init
    model.eval()
    with torch.no_grad():
        full = model(x)
        halves = torch.cat([model(x[:16]), model(x[16:])])
    assert torch.isfinite(full).all()
    assert rel(full.cpu().numpy(), halves.cpu().numpy()) < 1e-6
    # (b) train-mode BN invariants on the first fused stage: mean(bn(z)) = beta, var = gamma^2 var/(var+eps)
    from smaat_unet_amd import ops as K
    c0, bn0 = model.inc.double_conv[0], model.inc.double_conv[1]
    with torch.no_grad():
        bn0.weight.uniform_(0.5, 1.5)
        bn0.bias.uniform_(-0.3, 0.3)
    z, part, slots = K._dsconv_fwd_raw(x, c0.depthwise.weight.detach(), c0.depthwise.bias.detach(),
                                       c0.pointwise.weight.detach(), c0.pointwise.bias.detach(), 2, True)
    st = K._bn_finalize_raw(part, slots, 64, 32 * 288 * 288, c0.pointwise.bias.detach(), bn0.weight.detach(),
                            bn0.bias.detach(), 1e-5, 0.1, None, None)
    a = K._affine_act_raw(z, st[2], st[3], False).double()
    m = a.mean(dim=(0, 2, 3)).float()
    v = a.var(dim=(0, 2, 3), unbiased=False).float()
    zvar = z.double().var(dim=(0, 2, 3), unbiased=False).float()
    assert (m - bn0.bias.detach()).abs().max().item() < 2e-5
    assert rel((v / (bn0.weight.detach() ** 2 * zvar / (zvar + 1e-5))).cpu().numpy(), np.ones(64)) < 1e-4
    # (c) backward is linear in the cotangent: scaling the loss by 2 scales every gradient by exactly 2
    model.train()
    grads = []
    for scale in (1.0, 2.0):
        model.zero_grad(set_to_none=True)
        for mod in model.modules():  # same BN momentum state does not matter; stats are batch stats
            pass
        out = model(x)
        (torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 32 * scale).backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    for g1, g2 in zip(*grads):
        assert torch.equal(g1 * 2.0, g2)
    tot = torch.cat([g.flatten() for g in grads[0]])
    assert torch.isfinite(tot).all() and tot.abs().sum().item() > 0


def test_bf16_mixed_precision_mode(ops):
    """BASELINE configs[3]: pointwise GEMMs with bf16 operands / f32 accumulation (ops.set_matrix_mode("bf16")).
    SURVEY 8(c)(5) asks ~1e-2 against the fp32 oracle.  That bound holds where the error is the bf16 rounding class
    and is checked there:
      (1) one GEMM: against fp64 with the operands rounded to bf16 (round-to-nearest-even) <= 2e-6 -- the mode
          computes EXACTLY "bf16 operands, f32 accumulation", nothing else -- and against the unrounded fp64 product
          5e-4 .. 5e-3 (the rounding class, not more and not less);
      (2) every DoubleConvDS / DownDS / UpDS block against the reference's fp32 goldens: output <= 1e-2 (gradients,
          which pass through two BatchNorm backward passes on 12x16 maps, <= 0.15).
    End to end this randomly initialised network amplifies ANY per-op perturbation by 60-150x (f32: 1e-7 per op ->
    1.5e-5 on the logits), and every 1e-6 difference in an f32 activation flips bf16 roundings downstream,
    so the logits sit 0.1 from the fp32 path and 3e-2 from the fp32 ATen port with the SAME operand rounding emulated
    (oracle/torch_ref.py PW_BF16): both are recorded in gpurun_out/bf16_mode.json; asserted are only that the emulated
    oracle is CLOSER than the fp32 path (the mode is what it says), the loss within 1 % of it, and that training
    still reduces the loss."""
    from oracle import torch_ref
    from smaat_unet_amd import ops as K
    meta = dict(n_channels=12, n_classes=1, param_seed=3)
    xn, yn = O.synthetic_precip(2, 12, 64, 64, seed=11)
    x, y = torch.from_numpy(xn).to(DEV), torch.from_numpy(yn).to(DEV)

    def run(steps=1):
        model, P = _load_model(meta)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        first, losses = None, []
        for _ in range(steps):
            out = model(x)
            if first is None:
                first = out.detach().cpu().numpy()
            loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 2
            opt.zero_grad(set_to_none=True)
            loss.backward()
            assert all(torch.isfinite(p.grad).all() for p in model.parameters())
            opt.step()
            losses.append(loss.item())
        return first, losses, P

    ref_out, ref_losses, P = run()
    prev = K.set_matrix_mode("bf16")
    report = {}
    try:
        # (1) one layer against fp64
        g = torch.Generator().manual_seed(0)
        xa = torch.randn(2, 256, 36, 36, generator=g).to(DEV)
        w = (torch.randn(128, 256, generator=g) * 0.1).to(DEV)
        pl = K._split_planes_raw(w)
        z, _, _ = K._pointwise_split_raw(xa, pl, None, 128)
        ref = torch.einsum("mk,nkp->nmp", w.double(), xa.double().flatten(2)).view(2, 128, 36, 36)
        e_op = ((z.double() - ref).norm() / ref.norm()).item()
        assert 5e-4 < e_op < 5e-3, e_op
        bf = lambda t: t.to(torch.bfloat16).double()  # noqa: E731  (round to nearest even)
        ref_r = torch.einsum("mk,nkp->nmp", bf(w), bf(xa).flatten(2)).view(2, 128, 36, 36)
        e_exact = ((z.double() - ref_r).norm() / ref_r.norm()).item()
        assert e_exact < 2e-6, e_exact
        # (2) blocks against the fp32 goldens of the reference
        for tag, ctor in (("doubleconv", lambda: S.DoubleConvDS(6, 16, kernels_per_layer=2)),
                          ("down", lambda: S.DownDS(6, 12, kernels_per_layer=2)),
                          ("up", lambda: S.UpDS(16, 6, bilinear=True, kernels_per_layer=2))):
            run_case(ops, tag, ctor(), tol_out=1e-2, tol_grad=0.15, tol_stat=1e-2)
        # (3) the network against the fp32 oracle with the same operand rounding
        out, losses, _ = run(steps=4)
        torch_ref.PW_BF16 = lambda t: t.shape[-1] % 4 == 0  # the layers the split kernels take (ops._split_all)
        try:
            Pt = torch_ref.params_from_numpy(P, requires_grad=False)
            with torch.no_grad():
                emu = torch_ref.forward(Pt, torch.from_numpy(xn), training=True)
            loss_emu = (torch.nn.functional.mse_loss(emu.squeeze(1), torch.from_numpy(yn), reduction="sum") / 2).item()
        finally:
            torch_ref.PW_BF16 = None
        report = dict(per_op_vs_fp64=e_op, logits_vs_bf16_emulating_oracle=rel(out, emu.numpy()),
                      logits_vs_fp32_path=rel(out, ref_out), loss=losses[0], loss_emulated=loss_emu,
                      loss_fp32=ref_losses[0])
        report["gemm_vs_fp64_with_bf16_operands"] = e_exact
        assert report["logits_vs_bf16_emulating_oracle"] < 0.5 * report["logits_vs_fp32_path"], report
        assert abs(losses[0] - loss_emu) < 1e-2 * abs(loss_emu), report
        assert report["logits_vs_fp32_path"] > 1e-5                      # the mode really rounds
        assert losses[-1] < losses[0], losses
    finally:
        K.set_matrix_mode(prev)
        if os.path.isdir("gpurun_out"):
            with open("gpurun_out/bf16_mode.json", "w") as f:
                json.dump(report, f, indent=1, default=float)
    again, _, _ = run()                  # back on the default path: bit-identical to the first f32 run
    assert np.array_equal(again, ref_out)


@pytest.mark.skipif(os.environ.get("SMAAT_DW_ROWS", "1") == "0" or os.environ.get("SMAAT_UP_ROWS", "1") == "0",
                    reason="row-streaming kernels switched off: mixed precision is not available")
def test_bf16_storage_mixed_precision(ops):
    """BASELINE configs[3], real mixed precision (model.set_precision("bf16")): every activation tensor and its gradient is
    STORED as bfloat16, f32 arithmetic / accumulation / BatchNorm statistics, f32 master weights and weight gradients.
      (1) blocks against the reference's fp32 goldens: outputs <= 1e-2, gradients <= 0.15;
      (2) the network on the GPU against the SAME mixed-precision arithmetic evaluated by the numpy emulation of the C ABI
          (tests/emu_backend.py: f32 twins + one rounding per stored tensor): the loss within 1 %, the logits at less than
          half the distance of the f32 path, the flat gradient at cosine > 0.9 -- the kernels compute the mode they claim;
          the residual is 1-ulp bf16 flips caused by f32 accumulation order, amplified by the network;
      (3) against the fp32 path: no further than 1.25 x what stock torch.autocast(bfloat16) does to the REFERENCE modules on
          this case (fixture autocast_bf16_n2_64 generated from /root/reference: logits 0.180, flat gradient 1 - cos 0.220);
          numbers in gpurun_out/bf16_storage.json;
      (4) every activation the autograd graph keeps is bf16 (2 bytes per element), logits and parameter gradients are f32;
      (5) training reduces the loss; the default path afterwards is bit-identical to before."""
    from tests import emu_backend
    meta = dict(n_channels=12, n_classes=1, param_seed=3)
    xn, yn = O.synthetic_precip(2, 12, 64, 64, seed=11)
    x, y = torch.from_numpy(xn).to(DEV), torch.from_numpy(yn).to(DEV)
    report = {}
    # (1)
    with S.precision("bf16"):
        for tag, ctor in (("doubleconv", lambda: S.DoubleConvDS(6, 16, kernels_per_layer=2)),
                          ("down", lambda: S.DownDS(6, 12, kernels_per_layer=2)),
                          ("up", lambda: S.UpDS(16, 6, bilinear=True, kernels_per_layer=2))):
            run_case(ops, tag, ctor(), tol_out=1e-2, tol_grad=0.15, tol_stat=1e-2)

    def run(mode, steps=1, dev=DEV):
        model, _ = _load_model(meta)
        model = model.to(dev)
        model.set_precision(mode)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        xx, yy = x.to(dev), y.to(dev)
        first, losses, g = None, [], None
        for _ in range(steps):
            out = model(xx)
            assert out.dtype == torch.float32
            loss = torch.nn.functional.mse_loss(out.squeeze(1), yy, reduction="sum") / 2
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if first is None:
                first = out.detach().cpu().numpy()
                g = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu()
                assert all(p.grad.dtype == torch.float32 for p in model.parameters())
            assert all(torch.isfinite(p.grad).all() for p in model.parameters())
            opt.step()
            losses.append(loss.item())
        return first, losses, g

    f32_out, f32_losses, f32_g = run("f32")
    out, losses, g = run("bf16", steps=4)
    # (4) what the graph saved
    model, _ = _load_model(meta)
    model.set_precision("bf16")
    saved = []
    with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append((t.dtype, tuple(t.shape))), t)[1], lambda t: t):
        model(x)
    # activation-sized tensors: [N = 2][C][H][W] maps (parameters, per-channel vectors and partial sums are f32 by design)
    acts = [(d, sh) for d, sh in saved if len(sh) == 4 and sh[0] == 2 and sh[2] > 1 and sh[3] > 1]
    n_bf = sum(int(np.prod(sh)) for d, sh in acts if d == torch.bfloat16)
    f32_acts = [(d, sh) for d, sh in acts if d == torch.float32]
    report["saved_activation_elements"] = dict(bf16=n_bf, f32=sum(int(np.prod(sh)) for _, sh in f32_acts))
    # f32: the network input and the 1- / 2-channel attention maps (gate, conv, pooled maps) only
    assert all(sh[1] <= 2 or sh == (2, 12, 64, 64) for _, sh in f32_acts), f32_acts
    assert n_bf > 20 * report["saved_activation_elements"]["f32"]
    # (2) the same arithmetic on the host
    emu_backend.install()
    try:
        emu_out, emu_losses, emu_g = run("bf16", dev=torch.device("cpu"))
    finally:
        emu_backend.uninstall()
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))  # noqa: E731
    report.update(logits_vs_emulated_mixed_precision=rel(out, emu_out), logits_vs_fp32_path=rel(out, f32_out),
                  grad_cos_vs_emulated=cos(g, emu_g), grad_cos_vs_fp32=cos(g, f32_g), loss=losses[0],
                  loss_emulated=emu_losses[0], loss_fp32=f32_losses[0], losses=losses)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/bf16_storage.json", "w") as f:
            json.dump(report, f, indent=1, default=float)
    # the mode is what it says: the emulation of the SAME mixed-precision arithmetic is much closer than the f32 path
    # (measured 0.06 vs 0.18: every 1e-6 accumulation-order difference flips bf16 roundings downstream, and this
    # random-init network amplifies any per-op perturbation 60-150x)
    assert report["logits_vs_emulated_mixed_precision"] < 0.5 * report["logits_vs_fp32_path"], report
    assert report["grad_cos_vs_emulated"] > 0.9, report
    # (3) the yardstick is what stock autocast does to the REFERENCE on this very case (tests/golden/autocast_bf16_n2_64.npz,
    # same parameters / batch: logits 0.180, 1 - cosine 0.220); tests/test_autocast_yardstick.py holds the full rule
    from tests.test_autocast_yardstick import FACTOR, yardstick
    yard = yardstick(np.load(os.path.join(os.path.dirname(__file__), "golden", "autocast_bf16_n2_64.npz")))
    report["reference_autocast_yardstick"] = yard
    assert report["logits_vs_fp32_path"] < FACTOR * yard["logits"], report
    assert 1.0 - report["grad_cos_vs_fp32"] < FACTOR * yard["one_minus_cos"], report
    # ... and tensor by tensor: against the f32 path and against the emulation of the same mixed-precision arithmetic
    names_model, _ = _load_model(meta)
    _per_tensor_in_the_autocast_class(names_model, g, f32_g)
    _per_tensor_in_the_autocast_class(names_model, g, emu_g)
    assert abs(losses[0] - emu_losses[0]) < 1e-2 * abs(emu_losses[0]), report
    assert losses[-1] < losses[0], losses
    again, _, _ = run("f32")
    assert np.array_equal(again, f32_out)


def test_training_trajectory_against_the_aten_reference():
    """Twelve Adam steps (reference models/regression_lightning.py:48,57-65: MSE(sum)/N, Adam lr 1e-3) from the same
    initial state on the same batches: the GPU path against the reference's arithmetic on the CPU (oracle/torch_ref.py,
    pinned to reference-generated fixtures).  Training this network is chaotic at round-off level -- Adam's first steps
    are sign-like, so coordinates whose gradient is round-off noise move by +-lr in a random direction: the reference run
    in float32 separates from the same run in float64 by 1e-2 within a few steps.  So the yardstick is measured in the
    test: at every step this implementation must stay within 4x the distance of the reference's own float32 run from
    its float64 run so far (floor 1e-3); step 0 -- before any update -- is compared at 1e-5.  What this
    adds to the single-step fixtures: state carried between steps (running statistics counters, optimizer-visible
    gradients of every parameter, no stale cached weights in the training path) and a loss that falls as the reference's."""
    from oracle import torch_ref
    seed, steps = 11, 12
    Pn = oparams.make_smaat_params(12, 1, 2, 16, seed)
    batches = [O.synthetic_precip(2, 12, 64, 64, seed=100 + i) for i in range(steps)]

    def reference(dtype):
        P = {k: (v.detach().to(dtype).requires_grad_(v.requires_grad) if v.is_floating_point() else v)
             for k, v in torch_ref.params_from_numpy(Pn).items()}
        opt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3)
        out = []
        for xn, yn in batches:
            rl, _ = torch_ref.train_step(P, torch.from_numpy(xn).to(dtype), torch.from_numpy(yn).to(dtype))
            opt.step()
            out.append(float(rl))
        return out

    nthr = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        ref_a, ref_b = reference(torch.float64), reference(torch.float32)
    finally:
        torch.set_num_threads(nthr)
    model = S.SmaAt_UNet(12, 1)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in Pn.items()})
    model.to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for xn, yn in batches:
        y = torch.from_numpy(yn).to(DEV)
        out = model(torch.from_numpy(xn).to(DEV))
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.shape[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    self_div = [abs(a - b) / abs(a) for a, b in zip(ref_a, ref_b)]
    ours = [abs(a - b) / abs(b) for a, b in zip(losses, ref_a)]
    report = {"losses": losses, "reference_fp64": ref_a, "reference_fp32": ref_b, "reference_fp32_vs_fp64": self_div,
              "ours_vs_reference": ours}
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/trajectory.json", "w") as f:
            json.dump(report, f, indent=1)
    assert ours[0] < 1e-5, report
    for i in range(1, steps):
        assert ours[i] <= max(4.0 * max(self_div[:i + 1]), 1e-3), (i, report)
    assert ref_a[-1] < 0.2 * ref_a[0] and losses[-1] < 0.2 * losses[0]  # both learn this stream
    assert all(int(v) == steps for k, v in model.state_dict().items() if "num_batches" in k)
    assert all(torch.isfinite(v).all() for v in model.state_dict().values())


@pytest.mark.parametrize("shape", [(1, 64, 64), (3, 48, 80), (2, 80, 48), (1, 96, 32), (2, 36, 36), (1, 40, 72), (5, 32, 32),
                                   (2, 128, 64), (1, 100, 52), (2, 18, 22)])
def test_shape_sweep_vs_the_aten_reference(shape):
    """Every kernel family has shape-specialised forms (row-streaming / strip / element kernels, two-column last groups,
    sliced GEMMs below a fill threshold, padded up path for odd levels): a sweep over batch sizes and H x W -- multiples of
    16 and not, H != W, levels that become odd after pooling -- of the whole network, forward and every gradient, against
    the reference's arithmetic on the CPU (oracle/torch_ref.py) in float32 AND float64.  Bounds: logits 1e-4 rel-L2 of the
    float32 reference; the flat gradient no further from the float64 one than 2x what the float32 reference itself is, or
    2x what the float64 gradient moves under a 1e-6 input perturbation (decision flips: 6e-4 ... 1e-2 at these sizes, the
    larger of the two yardsticks), every tensor's cosine with the float64 gradient >= 0.999."""
    from oracle import torch_ref
    n, h, w = shape
    Pn = oparams.make_smaat_params(12, 1, 2, 16, 21)
    xn, yn = O.synthetic_precip(n, 12, h, w, seed=300 + h + w)
    nthr = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        P = torch_ref.params_from_numpy(Pn)
        rl, rlog = torch_ref.train_step(P, torch.from_numpy(xn), torch.from_numpy(yn))
        def p64():
            return {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v)
                    for k, v in torch_ref.params_from_numpy(Pn).items()}

        P64 = p64()
        x64, y64 = torch.from_numpy(xn).double(), torch.from_numpy(yn).double()
        torch_ref.train_step(P64, x64, y64)
        # how far the EXACT gradient moves when the input moves by 1e-6 (ReLU / max-pool / arg-max decisions at round-off
        # distance from a tie flip): the reference's float32 run can land on either side of those, and so can this one
        sens = []
        for sd in range(4):
            Pp = p64()
            gen = torch.Generator().manual_seed(sd)
            torch_ref.train_step(Pp, x64 * (1 + 1e-6 * torch.randn(x64.shape, generator=gen, dtype=torch.float64)), y64)
            d2 = sum(float((Pp[k].grad - P64[k].grad).norm() ** 2) for k, v in P64.items() if v.requires_grad)
            n2 = sum(float(P64[k].grad.norm() ** 2) for k, v in P64.items() if v.requires_grad)
            sens.append((d2 / n2) ** 0.5)
        sens = sorted(sens)[len(sens) // 2]
    finally:
        torch.set_num_threads(nthr)
    model = S.SmaAt_UNet(12, 1)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in Pn.items()})
    model.to(DEV).train()
    y = torch.from_numpy(yn).to(DEV)
    out = model(torch.from_numpy(xn).to(DEV))
    loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.shape[0]
    loss.backward()
    e = float((out.detach().cpu() - rlog).norm() / rlog.norm())
    assert e < 1e-4, ("logits", shape, e)
    assert abs(float(loss.detach()) - float(rl)) < 1e-4 * abs(float(rl))
    num = ref_num = den = 0.0
    for k, p in model.named_parameters():
        g, r32, r = p.grad.detach().cpu().double(), P[k].grad.double(), P64[k].grad
        num += float((g - r).norm() ** 2)
        ref_num += float((r32 - r).norm() ** 2)
        den += float(r.norm() ** 2)
        zero_by_construction = ".double_conv." in k and (k.endswith("pointwise.bias") or k.endswith("depthwise.bias"))
        if not zero_by_construction and float(r.norm()) > 1e-9:
            cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-300))
            assert cos > 0.999, (k, shape, cos)
    ours, ref = (num / den) ** 0.5, (ref_num / den) ** 0.5
    if os.path.isdir("gpurun_out"):
        with open(f"gpurun_out/shape_sweep_{n}x{h}x{w}.json", "w") as f:
            json.dump({"ours_vs_fp64": ours, "reference_fp32_vs_fp64": ref, "fp64_sensitivity_to_1e-6_input": sens, "logits": e}, f)
    assert ours <= max(2.0 * ref, 2.0 * sens, 1e-4), (shape, ours, ref, sens)
    # the same shapes through the inference fast path (BatchNorm folded on the running statistics both sides just updated,
    # fused / sliced kernels, module-owned hipGraph): eval logits against the reference's eval forward
    with torch.no_grad():
        ref_eval = torch_ref.forward(P, torch.from_numpy(xn), training=False)
        model.eval()
        ev = model(torch.from_numpy(xn).to(DEV))
        model.enable_eval_graph()
        evg = model(torch.from_numpy(xn).to(DEV))
        model.enable_eval_graph(False)
    ee = float((ev.cpu() - ref_eval).norm() / ref_eval.norm())
    assert ee < 1e-4, ("eval logits", shape, ee)
    assert torch.equal(ev, evg)


@pytest.mark.parametrize("kpl,shape", [(1, (2, 48, 64)), (4, (2, 48, 64)), (1, (1, 36, 20)), (4, (3, 32, 32)), (3, (2, 32, 48))])
def test_kernels_per_layer_sweep_vs_the_aten_reference(kpl, shape):
    """SmaAt_UNet at kernels_per_layer 1, 4 (strip / row kernels with other register shapes) and 3 (general depthwise path),
    forward and flat gradient against the ATen restatement in float32 / float64 with the yardsticks of the shape sweep"""
    import torch.nn.functional as F
    from oracle import torch_ref
    n, h, w = shape
    Pn = oparams.make_smaat_params(12, 1, kpl, 16, 41)
    xn, yn = O.synthetic_precip(n, 12, h, w, seed=500 + h + w + kpl)

    def ref_step(dtype, x):
        P = {k: (v.detach().to(dtype).requires_grad_(v.requires_grad) if v.is_floating_point() else v)
             for k, v in torch_ref.params_from_numpy(Pn).items()}
        logits = torch_ref.forward(P, x.to(dtype), kpl=kpl)
        loss = F.mse_loss(logits.squeeze(1), torch.from_numpy(yn).to(dtype), reduction="sum") / n
        loss.backward()
        return P, logits.detach(), float(loss)

    nthr = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        x0 = torch.from_numpy(xn)
        P32, rlog, rl = ref_step(torch.float32, x0)
        P64, _, _ = ref_step(torch.float64, x0)
        gen = torch.Generator().manual_seed(0)
        Pp, _, _ = ref_step(torch.float64, x0.double() * (1 + 1e-6 * torch.randn(x0.shape, generator=gen, dtype=torch.float64)))
    finally:
        torch.set_num_threads(nthr)
    model = S.SmaAt_UNet(12, 1, kernels_per_layer=kpl)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in Pn.items()})
    model.to(DEV).train()
    y = torch.from_numpy(yn).to(DEV)
    out = model(torch.from_numpy(xn).to(DEV))
    loss = F.mse_loss(out.squeeze(1), y, reduction="sum") / n
    loss.backward()
    e = float((out.detach().cpu() - rlog).norm() / rlog.norm())
    assert e < 1e-4, ("logits", kpl, shape, e)
    num = ref_num = sens_num = den = 0.0
    for k, p in model.named_parameters():
        g, r = p.grad.detach().cpu().double(), P64[k].grad
        num += float((g - r).norm() ** 2)
        ref_num += float((P32[k].grad.double() - r).norm() ** 2)
        sens_num += float((Pp[k].grad - r).norm() ** 2)
        den += float(r.norm() ** 2)
    ours, ref, sens = (num / den) ** 0.5, (ref_num / den) ** 0.5, (sens_num / den) ** 0.5
    assert ours <= max(2.0 * ref, 2.0 * sens, 1e-4), (kpl, shape, ours, ref, sens)


@pytest.mark.skipif(os.environ.get("SMAAT_DW_ROWS", "1") == "0" or os.environ.get("SMAAT_UP_ROWS", "1") == "0",
                    reason="row-streaming kernels switched off: mixed precision is not available")
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 32), (3, 32, 64), (2, 128, 64), (1, 32, 32)])
def test_mixed_precision_shape_sweep(shape):
    """bf16 activation storage over the shapes it is built for (multiples of 32, H != W, batch 1-3; the 18-wide / 2-wide
    deepest planes take the two-column row kernels and the narrow LDS-DMA form): the step runs on bf16 tensors, is finite,
    stays in the accuracy class of autocast (logits within 0.3 rel-L2 of the f32 path at random initialisation, gradient
    cosine > 0.7, loss within 3 %), and a second run is bit-identical (no atomics, fixed reduction orders)."""
    n, h, w = shape
    Pn = oparams.make_smaat_params(12, 1, 2, 16, 31)
    xn, yn = O.synthetic_precip(n, 12, h, w, seed=400 + h + w)

    models = []

    def run(prec):
        model = S.SmaAt_UNet(12, 1)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in Pn.items()})
        model.to(DEV).train().set_precision(prec)
        y = torch.from_numpy(yn).to(DEV)
        out = model(torch.from_numpy(xn).to(DEV))
        loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / y.shape[0]
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        models.append(model)
        return out.detach(), float(loss.detach()), g

    o32, l32, g32 = run("f32")
    ob, lb, gb = run("bf16")
    ob2, lb2, gb2 = run("bf16")
    # (tensor-by-tensor bounds live in tests/test_autocast_yardstick.py, at the sizes the reference's per-tensor yardstick was
    # generated at: on these small planes single attention tensors are noise -- cbam1's MLP gradient reverses between two
    # correct evaluations at 32 x 64 -- and the flat vector is the meaningful quantity)
    assert ob.dtype == torch.float32 and torch.isfinite(ob).all() and torch.isfinite(gb).all()
    assert torch.equal(ob, ob2) and torch.equal(gb, gb2) and lb == lb2
    rel = float((ob - o32).norm() / o32.norm())
    cos = float((gb.double() * g32.double()).sum() / (gb.double().norm() * g32.double().norm()))
    assert 1e-4 < rel < 0.3, (shape, rel)          # (really rounded, and in the class of autocast)
    assert cos > 0.7, (shape, cos)
    assert abs(lb - l32) < 0.03 * abs(l32), (shape, lb, l32)


def test_voc_config_256_batch16():
    """BASELINE configs[4]: SmaAt_UNet(3, 21) on 256x256, batch 16, CrossEntropyLoss (reference
    train_SmaAtUNet.py:178-183): three Adam steps reduce the loss, everything stays finite."""
    torch.manual_seed(0)
    model = S.SmaAt_UNet(3, 21).to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 3, 256, 256, generator=g).to(DEV)
    y = torch.randint(0, 21, (16, 256, 256), generator=g).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        out = model(x)
        assert out.shape == (16, 21, 256, 256)
        loss = torch.nn.functional.cross_entropy(out, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_library_is_the_hip_one():
    from smaat_unet_amd import _lib
    L = _lib.get()
    assert type(L).__name__ == "_Lib" and os.path.basename(_lib.LIB_PATH) == "libsmaat_hip.so"
    with open(f"/proc/{os.getpid()}/maps") as f:
        assert "libsmaat_hip.so" in f.read()


def test_ddp_rccl_single_rank(tmp_path):
    """the bucketed all-reduce path of smaat_unet_amd/ddp.py on RCCL (backend "nccl") with one rank on the GPU box
    (multi-rank behaviour is covered on CPU by tests/test_ddp_gloo.py): hooks fire, collectives run on device
    buffers, gradients equal the plain backward."""
    import torch.distributed as dist
    from smaat_unet_amd.ddp import FlatGradAllReduce
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    try:
        meta = dict(n_channels=12, n_classes=1, param_seed=3)
        xn, yn = O.synthetic_precip(2, 12, 64, 64, seed=5)
        x, y = torch.from_numpy(xn).to(DEV), torch.from_numpy(yn).to(DEV)
        model, _ = _load_model(meta)
        out = model(x)
        (torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 2).backward()
        ref = [p.grad.clone() for p in model.parameters()]
        model2, _ = _load_model(meta)
        ddp = FlatGradAllReduce(model2, buckets=2, overlap=True, force_collectives=True)
        for _ in range(2):
            ddp.zero_grad()
            out = model2(x)
            (torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 2).backward()
            assert all(ddp._launched), "the post-accumulate hooks did not launch the bucket all-reduces"
            flat = ddp.finish()
        torch.cuda.synchronize()
        assert flat.is_cuda and flat.numel() == 4033537
        for p, g in zip(model2.parameters(), ref):
            assert torch.equal(p.grad, g)
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """bench.py's multi-rank control flow (rank-0-only profiling legs, barriers, max-over-ranks timing, one JSON line) with
    two ranks on the ONE GPU of the test box: SMAAT_BENCH_BACKEND=gloo lets both ranks use cuda:0 (RCCL needs one device per
    rank).  The driver's multi-GPU runs use the same code path with backend nccl; no scaling number is claimed from this."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMAAT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--size", "64", "--no-cpu-baseline", "--no-alt", "--no-latency", "--no-eager-baseline", "--no-input-pipeline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    assert rec["kernels"]  # the rank-0 per-kernel leg ran while rank 1 waited at the final barrier


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_traceable_training_operators_on_gpu(mode, monkeypatch):
    """the torch.library training operators (smaat_unet_amd/train_ops.py) run the same kernels as the autograd.Function
    wiring: same logits (f32: the head / deferred-activation fusions only change which kernel applies an activation),
    same running statistics, gradients at the round-off level of the default path.  (The traceable operators keep the
    exact three-term split -- their saved tensors are declared up front, the operand maxima of the two-term fp16 split are
    not among them -- so the default path is switched to it as well: this test compares wirings at equal arithmetic.)"""
    from smaat_unet_amd import ops as _ops
    monkeypatch.setattr(_ops.policy, "f16_split", False)
    meta = dict(n_channels=12, n_classes=1, param_seed=3)
    xn, yn = O.synthetic_precip(2, 12, 64, 64, seed=11)
    x, y = torch.from_numpy(xn).to(DEV), torch.from_numpy(yn).to(DEV)
    res = []
    for traceable in (False, True):
        model, _ = _load_model(meta)
        model.set_precision(mode)
        with S.traceable_training(traceable):
            out = model(x)
            loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 2
            loss.backward()
        res.append((out.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]),
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    (o0, g0, s0), (o1, g1, s1) = res
    assert o1.dtype == torch.float32
    assert float((o1 - o0).norm() / o0.norm()) < (1e-5 if mode == "f32" else 5e-2)
    for k in s0:
        assert torch.allclose(s0[k], s1[k], rtol=1e-4 if mode == "f32" else 2e-2, atol=1e-6), k
    assert float((g1 - g0).norm() / g0.norm()) < (5e-3 if mode == "f32" else 0.6)


def test_training_runs_are_bit_reproducible_with_the_f16_split(monkeypatch):
    """two identical runs of eight Adam steps (fresh batch every step) end in bit-identical parameters: fixed reduction
    orders everywhere, and the operand maxima of the two-term fp16 split -- forced onto every layer here -- are
    order-independent (atomicMax on bit patterns).  scripts/probes/soak_determinism.py is the long form (150 steps at
    288 x 288, profiles/r5/soak_determinism_r5.txt)."""
    from smaat_unet_amd import ops as _ops
    monkeypatch.setattr(_ops.policy, "f16_min_samples", 0)

    def run():
        torch.manual_seed(0)
        m = S.SmaAt_UNet(12, 1).to(DEV).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(3)
        losses = []
        for _ in range(8):
            x = torch.rand(3, 12, 96, 64, generator=g).to(DEV)
            y = torch.rand(3, 96, 64, generator=g).to(DEV)
            loss = torch.nn.functional.mse_loss(m(x).squeeze(1), y, reduction="sum") / 3
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, [p.detach().clone() for p in m.parameters()]

    (la, pa), (lb, pb) = run(), run()
    assert la == lb and all(np.isfinite(la)) and la[-1] < la[0]
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
