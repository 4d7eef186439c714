#!/usr/bin/env python
"""Per-layer A/B of the split GEMMs at the BASELINE config-2 shapes (batch 32, 288 x 288 input): the exact three-term bf16 split
(six MFMAs per product) against the two-term fp16 split (three; round 5), for the forward pointwise GEMM, the data gradient
and the streamed weight gradient.  Operand images are prepared outside the timed region (once per step in the network).
Prints a table + gpurun_out/layer_bench_f16.json."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smaat_unet_amd import _lib  # noqa: E402
from scripts.layer_bench import LAYERS, timeit  # noqa: E402


def amax_word(t):
    w = torch.zeros(1024, dtype=torch.int32, device=t.device)  # SMAAT_AMAX_WORDS
    w[0] = np.array([float(t.abs().max())], np.float32).view(np.int32)[0].item()
    return w


def main():
    N = int(os.environ.get("LB_BATCH", "32"))
    only = os.environ.get("LB_ONLY", "")
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rows, tot = [], dict(fwd3=0.0, fwdh=0.0, dg3=0.0, dgh=0.0, wg3=0.0, wgh=0.0)
    for name, cin, cout, h in LAYERS:
        if only and not any(o in name for o in only.split(",")):
            continue
        w = h
        k, p = cin * 2, h * h
        if k < 64:
            continue  # (the 24-channel stem stays on the f32-MFMA data gradient)
        y = torch.randn(N, k, h, w, device=dev)
        dz = torch.randn(N, cout, h, w, device=dev) * 1e-3
        w_pw = torch.randn(cout, k, device=dev) * 0.1
        b_pw = torch.randn(cout, device=dev)
        z = torch.empty(N, cout, h, w, device=dev)
        dy = torch.empty(N, k, h, w, device=dev)
        part = torch.empty(3, L.smaat_pw_split_num_slots(N, h, w), cout, device=dev)
        ws = torch.empty(L.smaat_wgrad_num_splits(N, h, w, cout, k), cout, k, device=dev)
        dw = torch.empty(cout, k, device=dev)
        pl_f = torch.empty(3, cout, (k + 15) // 16 * 16, dtype=torch.int16, device=dev)
        pl_b = torch.empty(3, k, (cout + 15) // 16 * 16, dtype=torch.int16, device=dev)
        ph_f = torch.empty(L.smaat_split_planes_h_bytes(cout, k) // 2, dtype=torch.int16, device=dev)
        ph_b = torch.empty(L.smaat_split_planes_h_bytes(k, cout) // 2, dtype=torch.int16, device=dev)
        assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl_f.data_ptr(), st) == 0
        assert L.smaat_split_planes_t(w_pw.data_ptr(), k, cout, pl_b.data_ptr(), st) == 0
        assert L.smaat_split_planes_h(w_pw.data_ptr(), cout, k, ph_f.data_ptr(), 0, st) == 0
        assert L.smaat_split_planes_h(w_pw.data_ptr(), k, cout, ph_b.data_ptr(), 1, st) == 0
        ay, adz = amax_word(y), amax_word(dz)

        def fwd3():
            assert L.smaat_pointwise_fwd_split(y.data_ptr(), k * p, pl_f.data_ptr(), b_pw.data_ptr(), z.data_ptr(), cout * p,
                                               part.data_ptr(), N, k, cout, h, w, st) == 0

        def fwdh():
            assert L.smaat_pointwise_fwd_split_h(y.data_ptr(), k * p, ay.data_ptr(), ph_f.data_ptr(), b_pw.data_ptr(), z.data_ptr(),
                                                 cout * p, part.data_ptr(), N, k, cout, h, w, st) == 0

        def dg3():
            assert L.smaat_pointwise_fwd_split(dz.data_ptr(), cout * p, pl_b.data_ptr(), None, dy.data_ptr(), k * p, None, N, cout,
                                               k, h, w, st) == 0

        def dgh():
            assert L.smaat_pointwise_fwd_split_h(dz.data_ptr(), cout * p, adz.data_ptr(), ph_b.data_ptr(), None, dy.data_ptr(), k * p,
                                                 None, N, cout, k, h, w, st) == 0

        def wg3():
            assert L.smaat_pointwise_wgrad(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws.data_ptr(), dw.data_ptr(), N, k, cout, h,
                                           w, st) == 0

        def wgh():
            assert L.smaat_pointwise_wgrad_h(y.data_ptr(), k * p, ay.data_ptr(), dz.data_ptr(), cout * p, adz.data_ptr(), ws.data_ptr(),
                                             dw.data_ptr(), N, k, cout, h, w, st) == 0

        t = {}
        for rep in range(2):  # interleaved: a, b, a, b
            for nm, fn in (("fwd3", fwd3), ("fwdh", fwdh), ("dg3", dg3), ("dgh", dgh), ("wg3", wg3), ("wgh", wgh)):
                t[nm] = min(t.get(nm, 1e9), timeit(fn, iters=int(os.environ.get("LB_ITERS", "5"))))
        fl = 2.0 * N * k * cout * p
        if L.smaat_dsconv_rows_ok(2, cin, cout, h, w) and L.smaat_dsconv_wgrad_split_ok(2, cout, h, w):
            # the row-walking pair: fused forward with / without the maximum of its depthwise output, recompute weight gradient
            # on the three-term / two-term split
            x = torch.randn(N, cin, h, w, device=dev)
            w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
            partr = torch.empty(3, L.smaat_dsconv_rows_num_slots(N, h, w), cout, device=dev)
            wsr = torch.empty(L.smaat_dsconv_wgrad_split_num_splits(N, cin, cout, h, w), cout, k, device=dev)
            ayr = torch.zeros(1024, dtype=torch.int32, device=dev)

            def rf():
                assert L.smaat_dsconv_fwd_rows(x.data_ptr(), 0, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), pl_f.data_ptr(),
                                               b_pw.data_ptr(), z.data_ptr(), 0, cout * p, partr.data_ptr(), N, cin, 2, cout, h, w, st) == 0

            def rfa():
                assert L.smaat_dsconv_fwd_rows_amax(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), pl_f.data_ptr(),
                                                    b_pw.data_ptr(), z.data_ptr(), cout * p, partr.data_ptr(), ayr.data_ptr(), N, cin, 2,
                                                    cout, h, w, st) == 0

            def rw3():
                assert L.smaat_dsconv_wgrad_split(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(),
                                                  cout * p, wsr.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, w, st) == 0

            def rwh():
                assert L.smaat_dsconv_wgrad_split_h(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), ayr.data_ptr(),
                                                    dz.data_ptr(), cout * p, adz.data_ptr(), wsr.data_ptr(), dw.data_ptr(), N, cin, 2,
                                                    cout, h, w, st) == 0
            rfa()
            tr = {}
            for rep in range(2):
                for nm, fn in (("rows_fwd", rf), ("rows_fwd_amax", rfa), ("rwg3", rw3), ("rwgh", rwh)):
                    tr[nm] = min(tr.get(nm, 1e9), timeit(fn, iters=5))
            print(f"{name:8s}   row-walking pair: fused fwd {tr['rows_fwd']:.3f} ms, + max|y| {tr['rows_fwd_amax']:.3f} ms | recompute wgrad "
                  f"3xbf16 {tr['rwg3']:.3f} ms {fl / tr['rwg3'] / 1e9:6.1f} TF  2xfp16 {tr['rwgh']:.3f} ms {fl / tr['rwgh'] / 1e9:6.1f} TF", flush=True)
            t.update(tr)
            del x
        r = dict(layer=name, k=k, cout=cout, hw=h, gflop=fl / 1e9, **{a + "_ms": v for a, v in t.items()})
        rows.append(r)
        for a in tot:
            tot[a] += t.get(a, 0.0)
        tf = lambda ms: fl / ms / 1e9  # noqa: E731
        print(f"{name:8s} K={k:5d} M={cout:4d} {h:3d}^2 | fwd 3xbf16 {t['fwd3']:7.3f} ms {tf(t['fwd3']):6.1f} TF  2xfp16 {t['fwdh']:7.3f} ms "
              f"{tf(t['fwdh']):6.1f} TF | dgrad {t['dg3']:7.3f} {tf(t['dg3']):6.1f}  {t['dgh']:7.3f} {tf(t['dgh']):6.1f} | wgrad "
              f"{t['wg3']:7.3f} {tf(t['wg3']):6.1f}  {t['wgh']:7.3f} {tf(t['wgh']):6.1f}", flush=True)
        del y, dz, z, dy, ws
    print("totals ms:", {a: round(v, 3) for a, v in tot.items()})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/layer_bench_f16.json", "w") as f:
        json.dump(dict(batch=N, rows=rows, totals=tot), f, indent=1)


if __name__ == "__main__":
    main()
