#!/usr/bin/env python
"""Per-layer timing of the mixed-precision (bf16 storage) entry points at the BASELINE configs[3] shapes (batch 64,
288x288 input): depthwise forward, bf16 GEMM forward, data gradient, weight gradient, depthwise backward.  Every kernel
is HBM-bound in bf16, so the figure of merit is ALGORITHMIC bytes / time against 8 TB/s.
Prints a table + JSON (gpurun_out/layer_bench_bf16.json).  env: LB_BATCH (64), LB_ONLY (substring filter)."""
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
BF = torch.bfloat16


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
    N = int(os.environ.get("LB_BATCH", "64"))
    only = os.environ.get("LB_ONLY", "")
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    tot = dict(dwf=0.0, fwd=0.0, dgrad=0.0, wgrad=0.0, dwb=0.0)
    print(f"batch {N}: ms (algorithmic GB/s)")
    print(f"{'layer':9s} {'dw fwd':>14s} {'gemm fwd':>14s} {'dgrad':>14s} {'wgrad':>14s} {'dw bwd':>14s}")
    for name, cin, cout, h in LAYERS:
        if only and not any(o in name for o in only.split(",")):
            continue
        w = h
        k = cin * 2
        p = h * w
        x = torch.randn(N, cin, h, w, device=dev).to(BF)
        w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
        w_pw = torch.randn(cout, k, device=dev) * 0.1
        b_pw = torch.randn(cout, device=dev)
        y = torch.empty(N, k, h, w, device=dev, dtype=BF)
        z = torch.empty(N, cout, h, w, device=dev, dtype=BF)
        dz = torch.randn(N, cout, h, w, device=dev).to(BF)
        dy = torch.empty(N, k, h, w, device=dev, dtype=BF)
        dx = torch.empty(N, cin, h, w, device=dev, dtype=BF)
        slots = L.smaat_pw_split_num_slots(N, h, w)
        part = torch.empty(3, slots, cout, device=dev)
        pl = torch.empty(((k + 31) // 32 * 2, cout, 16), dtype=torch.int16, device=dev)
        plt = torch.empty(((cout + 31) // 32 * 2, k, 16), dtype=torch.int16, device=dev)
        assert L.smaat_bf16_planes(w_pw.data_ptr(), cout, k, pl.data_ptr(), 0, st) == 0
        assert L.smaat_bf16_planes(w_pw.data_ptr(), k, cout, plt.data_ptr(), 1, st) == 0
        ns = L.smaat_wgrad_num_splits(N, h, w, cout, k)
        ws = torch.empty(ns, cout, k, device=dev)
        dw = torch.empty(cout, k, device=dev)
        rows_ = L.smaat_dw3x3_bwd_ws_rows(N, cin, h, w)
        ws2 = torch.empty(rows_, k, 10, device=dev)
        dwd, dbd = torch.empty(k, 9, device=dev), torch.empty(k, device=dev)

        def chk(rc):
            assert rc == 0, rc

        t_dwf = timeit(lambda: chk(L.smaat_dw3x3_fwd_t(x.data_ptr(), 1, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                                       y.data_ptr(), 1, k * p, N, cin, 2, h, w, st)))
        t_fwd = timeit(lambda: chk(L.smaat_pointwise_fwd_bf16(y.data_ptr(), k * p, pl.data_ptr(), b_pw.data_ptr(), z.data_ptr(),
                                                              cout * p, 1, part.data_ptr(), N, k, cout, h, w, 0, st)))
        t_dg = timeit(lambda: chk(L.smaat_pointwise_fwd_bf16(dz.data_ptr(), cout * p, plt.data_ptr(), None, dy.data_ptr(), k * p,
                                                             1, None, N, cout, k, h, w, 0, st)))
        t_wg = timeit(lambda: chk(L.smaat_pointwise_wgrad_bf16(y.data_ptr(), k * p, dz.data_ptr(), cout * p, ws.data_ptr(),
                                                               dw.data_ptr(), N, k, cout, h, w, st)))
        t_dwb = timeit(lambda: chk(L.smaat_dw3x3_bwd_t(x.data_ptr(), 1, cin * p, None, None, dy.data_ptr(), 1, k * p,
                                                       w_dw.data_ptr(), dx.data_ptr(), 1, cin * p, ws2.data_ptr(), dwd.data_ptr(),
                                                       dbd.data_ptr(), None, None, None, N, cin, 2, h, w, st)))
        extra = ""
        if L.smaat_dsconv_rows_ok(2, cin, cout, h, w):  # round 4: row-walking fused forward + typed recompute weight gradient
            part_r = torch.empty(3, L.smaat_dsconv_rows_num_slots(N, h, w), cout, device=dev)
            t_rows = timeit(lambda: chk(L.smaat_dsconv_fwd_rows(x.data_ptr(), 1, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                                                pl.data_ptr(), b_pw.data_ptr(), z.data_ptr(), 1, cout * p,
                                                                part_r.data_ptr(), N, cin, 2, cout, h, w, st)))
            extra += f" | ROWS fwd {t_rows:6.3f} ({2.0 * N * (cin + cout) * p / (t_rows * 1e-3) / 1e9:5.0f}) vs dw+gemm {t_dwf + t_fwd:6.3f}"
        if L.smaat_dsconv_wgrad_split_ok(2, cout, h, w):
            wsr = torch.empty(L.smaat_dsconv_wgrad_split_num_splits(N, cin, cout, h, w), cout, k, device=dev)
            t_wr = timeit(lambda: chk(L.smaat_dsconv_wgrad_split_t(x.data_ptr(), 1, cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                                                   dz.data_ptr(), 1, cout * p, wsr.data_ptr(), dw.data_ptr(), N, cin, 2,
                                                                   cout, h, w, st)))
            extra += f" | recompute wgrad {t_wr:6.3f} ({2.0 * N * (cin + cout) * p / (t_wr * 1e-3) / 1e9:5.0f}) vs {t_wg:6.3f}"
        gb = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9  # noqa: E731
        b_dwf = 2.0 * N * (cin + k) * p
        b_g = 2.0 * N * (k + cout) * p
        b_dwb = 2.0 * N * (k + 2 * cin) * p
        r = dict(layer=name, dwf_ms=t_dwf, dwf_gbs=gb(b_dwf, t_dwf), fwd_ms=t_fwd, fwd_gbs=gb(b_g, t_fwd), dgrad_ms=t_dg,
                 dgrad_gbs=gb(b_g, t_dg), wgrad_ms=t_wg, wgrad_gbs=gb(b_g, t_wg), dwb_ms=t_dwb, dwb_gbs=gb(b_dwb, t_dwb))
        rows.append(r)
        tot["dwf"] += t_dwf
        tot["fwd"] += t_fwd
        tot["dgrad"] += t_dg
        tot["wgrad"] += t_wg
        tot["dwb"] += t_dwb
        print(f"{name:9s} {t_dwf:6.3f} ({r['dwf_gbs']:5.0f}) {t_fwd:6.3f} ({r['fwd_gbs']:5.0f}) {t_dg:6.3f} ({r['dgrad_gbs']:5.0f}) "
              f"{t_wg:6.3f} ({r['wgrad_gbs']:5.0f}) {t_dwb:6.3f} ({r['dwb_gbs']:5.0f})" + extra, flush=True)
        del x, y, z, dz, dy, dx, ws, ws2
    print("total ms: " + "  ".join(f"{k_} {v:.3f}" for k_, v in tot.items()))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/layer_bench_bf16.json", "w") as f:
        json.dump(dict(batch=N, rows=rows, total=tot), f, indent=1)


if __name__ == "__main__":
    main()
