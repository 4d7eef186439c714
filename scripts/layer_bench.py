#!/usr/bin/env python
"""Per-layer timing of the GEMM-family entry points at BASELINE config-2 shapes
(batch 32, 288x288 input): fused dsconv forward, dgrad, streamed wgrad, dw backward.
Prints a table + JSON (gpurun_out/layer_bench.json)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smaat_unet_amd import _lib  # noqa: E402

LAYERS = [  # name, Cin, Cout, H
    ("inc.0", 12, 64, 288), ("inc.1", 64, 64, 288), ("down1.0", 64, 128, 144), ("down1.1", 128, 128, 144),
    ("down2.0", 128, 256, 72), ("down2.1", 256, 256, 72), ("down3.0", 256, 512, 36), ("down3.1", 512, 512, 36),
    ("down4.0", 512, 512, 18), ("down4.1", 512, 512, 18), ("up1.0", 1024, 512, 36), ("up1.1", 512, 256, 36),
    ("up2.0", 512, 256, 72), ("up2.1", 256, 128, 72), ("up3.0", 256, 128, 144), ("up3.1", 128, 64, 144),
    ("up4.0", 128, 64, 288), ("up4.1", 64, 64, 288),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    N = int(os.environ.get("LB_BATCH", "32"))
    only = os.environ.get("LB_ONLY", "")
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, dwb=0.0)
    for name, cin, cout, h in LAYERS:
        if only and not any(o in name for o in only.split(",")):
            continue
        w = h
        k = cin * 2
        p = h * w
        x = torch.randn(N, cin, h, w, device=dev)
        w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
        w_pw = torch.randn(cout, k, device=dev) * 0.1
        wt = w_pw.t().contiguous()
        b_pw = torch.randn(cout, device=dev)
        z = torch.empty(N, cout, h, w, device=dev)
        y = torch.empty(N, k, h, w, device=dev)
        slots = L.smaat_pw_num_slots(N, h, w, cout)
        part = torch.empty(3, slots, cout, device=dev)
        dz = torch.randn(N, cout, h, w, device=dev)
        dy = torch.empty(N, k, h, w, device=dev)
        dx = torch.empty(N, cin, h, w, device=dev)
        ns = L.smaat_wgrad_num_splits(N, h, w, cout, k)
        ws = torch.empty(ns, cout, k, device=dev)
        dw = torch.empty(cout, k, device=dev)
        ws2 = torch.empty(L.smaat_dw3x3_bwd_ws_rows(N, cin, h, w), k, 10, device=dev)
        dwd, dbd = torch.empty(k, 9, device=dev), torch.empty(k, device=dev)

        split = bool(L.smaat_split_enabled())
        pl_f = torch.empty(3, cout, (k + 15) // 16 * 16, dtype=torch.int16, device=dev)
        pl_b = torch.empty(3, k, (cout + 15) // 16 * 16, dtype=torch.int16, device=dev)
        wtt = w_pw.t().contiguous()
        slots_s = L.smaat_pw_split_num_slots(N, h, w)
        part_s = torch.empty(3, slots_s, cout, device=dev)

        def f_fwd_split():
            assert L.smaat_dw3x3_fwd(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), y.data_ptr(), k * p, N, cin,
                                     2, h, w, st) == 0
            assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl_f.data_ptr(), st) == 0
            assert L.smaat_pointwise_fwd_split(y.data_ptr(), k * p, pl_f.data_ptr(), b_pw.data_ptr(), z.data_ptr(),
                                               cout * p, part_s.data_ptr(), N, k, cout, h, w, st) == 0

        slots_f = L.smaat_dsconv_split_num_slots(N, h, w)
        part_f = torch.empty(3, max(slots_f, 1), cout, device=dev)

        def f_fused(want_y):
            assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl_f.data_ptr(), st) == 0
            assert L.smaat_dsconv_fwd_split(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                            pl_f.data_ptr(), b_pw.data_ptr(), z.data_ptr(), cout * p, part_f.data_ptr(),
                                            y.data_ptr() if want_y else None, N, cin, 2, cout, h, w, st) == 0

        def f_wgrad_recompute():
            nsr = L.smaat_dsconv_wgrad_num_splits(N, h, w, cout, k)
            wsr = torch.empty(nsr, cout, k, device=dev)
            assert L.smaat_dsconv_wgrad(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(),
                                        cout * p, wsr.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, w, st) == 0

        def f_wgrad_recompute_split():
            nsr = L.smaat_dsconv_wgrad_split_num_splits(N, cin, cout, h, w)
            wsr = torch.empty(nsr, cout, k, device=dev)
            assert L.smaat_dsconv_wgrad_split(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(), dz.data_ptr(),
                                              cout * p, wsr.data_ptr(), dw.data_ptr(), N, cin, 2, cout, h, w, st) == 0

        def f_gemm_split():
            assert L.smaat_pointwise_fwd_split(y.data_ptr(), k * p, pl_f.data_ptr(), b_pw.data_ptr(), z.data_ptr(),
                                               cout * p, part_s.data_ptr(), N, k, cout, h, w, st) == 0

        def f_dgrad_split():
            assert L.smaat_split_planes(wtt.data_ptr(), k, cout, pl_b.data_ptr(), st) == 0
            assert L.smaat_pointwise_fwd_split(dz.data_ptr(), cout * p, pl_b.data_ptr(), None, dy.data_ptr(), k * p, None,
                                               N, cout, k, h, w, st) == 0

        def f_fwd():
            assert L.smaat_dsconv_fwd(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                      wt.data_ptr(), b_pw.data_ptr(), z.data_ptr(), cout * p, part.data_ptr(),
                                      y.data_ptr(), N, cin, 2, cout, h, w, st) == 0

        def f_fwd_noy():
            assert L.smaat_dsconv_fwd(x.data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                      wt.data_ptr(), b_pw.data_ptr(), z.data_ptr(), cout * p, part.data_ptr(),
                                      None, N, cin, 2, cout, h, w, st) == 0

        def f_dgrad():
            assert L.smaat_pointwise_fwd(dz.data_ptr(), cout * p, w_pw.data_ptr(), None, dy.data_ptr(), k * p, None,
                                         N, cout, k, h, w, st) == 0

        def f_wgrad():
            assert L.smaat_pointwise_wgrad(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws.data_ptr(),
                                           dw.data_ptr(), N, k, cout, h, w, st) == 0

        def f_dwb():
            assert L.smaat_dw3x3_bwd(x.data_ptr(), cin * p, dy.data_ptr(), k * p, w_dw.data_ptr(), dx.data_ptr(),
                                     cin * p, ws2.data_ptr(), dwd.data_ptr(), dbd.data_ptr(), N, cin, 2, h, w,
                                     st) == 0

        fl = 2.0 * N * k * cout * p
        if split and w % 4 == 0:  # split path: fwd = depthwise kernel + split GEMM ("noY" column = the GEMM alone)
            t_f, t_fn, t_d = timeit(f_fwd_split), timeit(f_gemm_split), timeit(f_dgrad_split)
            t_w, t_b = timeit(f_wgrad), timeit(f_dwb)
        else:
            t_f, t_fn, t_d, t_w, t_b = (timeit(f_fwd), timeit(f_fwd_noy), timeit(f_dgrad), timeit(f_wgrad),
                                        timeit(f_dwb))
        extra = ""
        if split and slots_f > 0 and os.environ.get("LB_FUSED", "1") != "0":
            t_fu, t_fuy = timeit(lambda: f_fused(False)), timeit(lambda: f_fused(True))
            t_wr = timeit(f_wgrad_recompute) if os.environ.get("LB_F32_RECOMPUTE", "0") == "1" else float("nan")
            gb = 4.0 * N * (cin + cout) * p / 1e6
            extra = (f" | FUSED fwd {t_fu:7.3f} ms {fl / t_fu / 1e9:6.1f} TF {gb / t_fu:7.1f} GB/s (with y_out {t_fuy:7.3f})"
                     f" | f32 recompute-wgrad {t_wr:7.3f}")
            if L.smaat_dsconv_rows_ok(2, cin, cout, h, w):  # round 4: row-walking fused forward
                slots_r = L.smaat_dsconv_rows_num_slots(N, h, w)
                part_r = torch.empty(3, slots_r, cout, device=dev)

                def f_rows():
                    assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl_f.data_ptr(), st) == 0
                    assert L.smaat_dsconv_fwd_rows(x.data_ptr(), 0, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                                   pl_f.data_ptr(), b_pw.data_ptr(), z.data_ptr(), 0, cout * p, part_r.data_ptr(),
                                                   N, cin, 2, cout, h, w, st) == 0
                t_r = timeit(f_rows)
                extra += f" | ROWS fwd {t_r:7.3f} ms {fl / t_r / 1e9:6.1f} TF {gb / t_r:7.1f} GB/s"
            if L.smaat_dsconv_wgrad_split_ok(2, cout, h, w):  # round 4: split-path weight gradient that recomputes y from x
                t_ws = timeit(f_wgrad_recompute_split)
                gbw = 4.0 * N * (cin + cout) * p / 1e6
                extra += f" | SPLIT recompute-wgrad {t_ws:7.3f} ms {fl / t_ws / 1e9:6.1f} TF {gbw / t_ws:7.1f} GB/s"
        bw = 4.0 * N * (k + 2 * cin) * p
        rows.append(dict(layer=name, cin=cin, k=k, cout=cout, hw=h, gflop=fl / 1e9, fwd_ms=t_f, fwd_noy_ms=t_fn,
                         dgrad_ms=t_d, wgrad_ms=t_w, dwb_ms=t_b, fwd_tf=fl / t_f / 1e9, fwd_noy_tf=fl / t_fn / 1e9,
                         dgrad_tf=fl / t_d / 1e9, wgrad_tf=fl / t_w / 1e9, dwb_gbs=bw / t_b / 1e6))
        tot["fwd"] += t_f
        tot["dgrad"] += t_d
        tot["wgrad"] += t_w
        tot["dwb"] += t_b
        r = rows[-1]
        print(f"{name:8s} K={k:5d} M={cout:4d} {h:3d}^2  fwd {t_f:7.3f} ms {r['fwd_tf']:6.1f} TF (noY {t_fn:7.3f} "
              f"{r['fwd_noy_tf']:6.1f}) | dgrad {t_d:7.3f} {r['dgrad_tf']:6.1f} | wgrad {t_w:7.3f} {r['wgrad_tf']:6.1f}"
              f" | dwb {t_b:7.3f} ms {r['dwb_gbs']:7.1f} GB/s" + extra, flush=True)
        del x, z, y, dz, dy, dx, ws
    print("totals ms:", {k: round(v, 2) for k, v in tot.items()})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/layer_bench.json", "w") as f:
        json.dump(dict(batch=N, rows=rows, totals=tot), f, indent=1)


if __name__ == "__main__":
    main()
