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
