#!/usr/bin/env python
"""Error of the GEMM-family entry points against an fp64 reference computed on the GPU, for the
current SMAAT_SPLIT mode (0 = f32 MFMA, 3 = three-term bf16 split, 2 = two-term).  Run once per mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smaat_unet_amd import _lib  # noqa: E402


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def main():
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    mode = os.environ.get("SMAAT_SPLIT", "0")
    torch.manual_seed(0)
    for (N, cin, cout, h) in [(4, 64, 64, 288), (4, 128, 256, 72), (8, 512, 512, 36), (8, 1024, 512, 18),
                              (2, 12, 64, 288)]:
        k, p = cin * 2, h * h
        y = torch.randn(N, k, h, h, device=dev) * torch.rand(N, k, 1, 1, device=dev) * 3
        dz = torch.randn(N, cout, h, h, device=dev) * 0.05
        w = torch.randn(cout, k, device=dev) * 0.1
        out = {}
        if "wgrad" in os.environ.get("SA_OPS", "wgrad dgrad fwd"):
            ns = L.smaat_wgrad_num_splits(N, h, h, cout, k)
            ws = torch.empty(ns, cout, k, device=dev)
            dw = torch.empty(cout, k, device=dev)
            assert L.smaat_pointwise_wgrad(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws.data_ptr(), dw.data_ptr(),
                                           N, k, cout, h, h, st) == 0
            ref = torch.einsum("nmp,nkp->mk", dz.double().flatten(2), y.double().flatten(2))
            out["wgrad"] = rel(dw, ref)
        if "dgrad" in os.environ.get("SA_OPS", "wgrad dgrad fwd"):
            dy = torch.empty(N, k, h, h, device=dev)
            if L.smaat_split_enabled():
                wtt = w.t().contiguous()  # [k][cout]
                pl = torch.empty(3, k, (cout + 15) // 16 * 16, dtype=torch.int16, device=dev)
                assert L.smaat_split_planes(wtt.data_ptr(), k, cout, pl.data_ptr(), st) == 0
                assert L.smaat_pointwise_fwd_split(dz.data_ptr(), cout * p, pl.data_ptr(), None, dy.data_ptr(), k * p,
                                                   None, N, cout, k, h, h, st) == 0
            else:
                assert L.smaat_pointwise_fwd(dz.data_ptr(), cout * p, w.data_ptr(), None, dy.data_ptr(), k * p, None,
                                             N, cout, k, h, h, st) == 0
            ref = torch.einsum("mk,nmp->nkp", w.double(), dz.double().flatten(2)).view(N, k, h, h)
            out["dgrad"] = rel(dy, ref)
        if "fwd" in os.environ.get("SA_OPS", "wgrad dgrad fwd"):
            x = torch.randn(N, cin, h, h, device=dev).relu_()
            w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
            b_pw = torch.randn(cout, device=dev)
            wt = w.t().contiguous()
            z = torch.empty(N, cout, h, h, device=dev)
            if L.smaat_split_enabled():
                yy = torch.empty(N, k, h, h, device=dev)
                assert L.smaat_dw3x3_fwd(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), yy.data_ptr(), k * p,
                                         N, cin, 2, h, h, st) == 0
                pl = torch.empty(3, cout, (k + 15) // 16 * 16, dtype=torch.int16, device=dev)
                assert L.smaat_split_planes(w.data_ptr(), cout, k, pl.data_ptr(), st) == 0
                assert L.smaat_pointwise_fwd_split(yy.data_ptr(), k * p, pl.data_ptr(), b_pw.data_ptr(), z.data_ptr(),
                                                   cout * p, None, N, k, cout, h, h, st) == 0
            else:
                assert L.smaat_dsconv_fwd(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                          wt.data_ptr(), b_pw.data_ptr(), z.data_ptr(), cout * p, None, None, N, cin,
                                          2, cout, h, h, st) == 0
            yd = torch.nn.functional.conv2d(x.double(), w_dw.double().view(k, 1, 3, 3), b_dw.double(), padding=1,
                                            groups=cin)
            ref = torch.einsum("mk,nkp->nmp", w.double(), yd.flatten(2)).view(N, cout, h, h) + \
                b_pw.double().view(1, -1, 1, 1)
            out["fwd"] = rel(z, ref)
        torch.cuda.synchronize()
        print(f"split={mode} N={N} K={k} M={cout} {h}^2  " + "  ".join(f"{a} {b:.3e}" for a, b in out.items()),
              flush=True)


if __name__ == "__main__":
    main()
