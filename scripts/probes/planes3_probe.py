#!/usr/bin/env python
"""Exact-f32 GEMM from pre-split planes (k_pw_bf16 NT = 3, LDS-DMA, no producer waves) against the persistent split
GEMM (k_pw_split_p, operands split on the fly) on the deep-layer shapes of BASELINE configs[1] (batch 32): time,
algorithmic TFLOP/s, and whether the two results agree bit for bit.  fwd: [K -> M]; dgrad: [M -> K]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

LAYERS = [("down2.1", 512, 256, 72), ("down3.0", 512, 512, 36), ("down3.1", 1024, 512, 36), ("down4.0", 1024, 512, 18),
          ("up1.0", 2048, 512, 36), ("up1.1", 1024, 256, 36), ("up2.0", 1024, 256, 72), ("up2.1", 512, 128, 72),
          ("up3.0", 512, 128, 144), ("down1.1", 256, 128, 144)]


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
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tot = [0.0, 0.0, 0.0]
    for name, k, m, h in LAYERS:
        for what, cin, cout in (("fwd", k, m), ("dgrad", m, k)):
            p = h * h
            x = torch.randn(N, cin, h, h, device=dev)
            w = torch.randn(cout, cin, device=dev) * 0.1
            b = torch.randn(cout, device=dev)
            cp = (cin + 15) // 16 * 16
            wpl = torch.empty(3 * cout * cp, dtype=torch.int16, device=dev)
            assert L.smaat_split_planes(w.data_ptr(), cout, cin, wpl.data_ptr(), st) == 0
            slots = L.smaat_pw_split_num_slots(N, h, h)
            part0, part1 = torch.empty(3, slots, cout, device=dev), torch.empty(3, slots, cout, device=dev)
            o0, o1 = torch.empty(N, cout, h, h, device=dev), torch.empty(N, cout, h, h, device=dev)
            xp = torch.empty(3, N, cin, h, h, dtype=torch.bfloat16, device=dev)
            ps = N * cin * p
            t_split = timeit(lambda: L.smaat_split_act3(x.data_ptr(), cin * p, xp.data_ptr(), cin * p, ps, N, cin, p, st))
            t0 = timeit(lambda: L.smaat_pointwise_fwd_split(x.data_ptr(), cin * p, wpl.data_ptr(), b.data_ptr(), o0.data_ptr(),
                                                             cout * p, part0.data_ptr(), N, cin, cout, h, h, st))
            rc = L.smaat_pointwise_fwd_planes3(xp.data_ptr(), cin * p, ps, wpl.data_ptr(), b.data_ptr(), o1.data_ptr(), cout * p,
                                               part1.data_ptr(), N, cin, cout, h, h, 0, st)
            assert rc == 0, rc
            t1 = timeit(lambda: L.smaat_pointwise_fwd_planes3(xp.data_ptr(), cin * p, ps, wpl.data_ptr(), b.data_ptr(),
                                                               o1.data_ptr(), cout * p, part1.data_ptr(), N, cin, cout, h, h, 0, st))
            torch.cuda.synchronize()
            fl = 2.0 * N * cin * cout * p
            same = torch.equal(o0, o1)
            err = float((o0 - o1).abs().max())
            print(f"{name:8s} {what:5s} {cin:5d}->{cout:5d} {h:3d}^2  split-on-the-fly {t0:6.3f} ms {fl / t0 / 1e9:6.1f} TF | planes3 "
                  f"{t1:6.3f} ms {fl / t1 / 1e9:6.1f} TF | x{t0 / t1:4.2f} | splitter pass {t_split:6.3f} ms | bit-identical {same} "
                  f"(max abs diff {err:.1e}) stats equal {torch.equal(part0, part1)}", flush=True)
            tot[0] += t0
            tot[1] += t1
            tot[2] += t_split
    print(f"totals ms: split-on-the-fly {tot[0]:.3f}  planes3 {tot[1]:.3f}  stand-alone splitter passes {tot[2]:.3f}")


if __name__ == "__main__":
    main()
