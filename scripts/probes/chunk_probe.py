#!/usr/bin/env python
"""Producer -> consumer kernel pairs run over the batch in CHUNKS small enough for the intermediate tensor to stay in the
256 MB memory-side cache between the two launches (scripts/probes/mall_probe.py: a tensor read right after it was written
comes back at 5.3-5.9 TB/s up to ~192 MB, against 3.3-3.5 TB/s from HBM).  In-stream, no overlap between launches:
what is measured is (cache hits) - (fill/drain of the smaller launches).
  forward : depthwise 3x3 -> [y] -> split GEMM
  backward: split data-gradient GEMM -> [dy] -> depthwise backward"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

LAYERS = [("inc.1", 64, 64, 288), ("up4.0", 128, 64, 288), ("down1.1", 128, 128, 144), ("up3.0", 256, 128, 144),
          ("up2.0", 512, 256, 72), ("up1.0", 1024, 512, 36)]
N = int(os.environ.get("CP_BATCH", "32"))
L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, iters=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, cin, cout, h in LAYERS:
    w = h
    k, p = 2 * cin, h * h
    x = torch.randn(N, cin, h, w, device=dev)
    w_dw, b_dw = torch.randn(k, 9, device=dev) * 0.3, torch.randn(k, device=dev) * 0.1
    w_pw, b_pw = torch.randn(cout, k, device=dev) * 0.1, torch.randn(cout, device=dev)
    y = torch.empty(N, k, h, w, device=dev)
    z = torch.empty(N, cout, h, w, device=dev)
    dz = torch.randn(N, cout, h, w, device=dev)
    dy = torch.empty(N, k, h, w, device=dev)
    dx = torch.empty(N, cin, h, w, device=dev)
    pl_f = torch.empty(3, cout, (k + 15) // 16 * 16, dtype=torch.int16, device=dev)
    pl_b = torch.empty(3, k, (cout + 15) // 16 * 16, dtype=torch.int16, device=dev)
    wtt = w_pw.t().contiguous()
    assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl_f.data_ptr(), st) == 0
    assert L.smaat_split_planes(wtt.data_ptr(), k, cout, pl_b.data_ptr(), st) == 0
    part = torch.empty(3, L.smaat_pw_split_num_slots(N, h, w), cout, device=dev)
    ws2 = torch.empty(L.smaat_dw3x3_bwd_ws_rows(N, cin, h, w), k, 10, device=dev)
    dwd, dbd = torch.empty(k, 9, device=dev), torch.empty(k, device=dev)
    line = f"{name:8s} y = {N * k * p * 4 / 2**20:6.0f} MB |"
    for c in (N, 16, 8, 4, 2, 1):
        if c > N:
            continue

        def fwd():
            for n0 in range(0, N, c):
                assert L.smaat_dw3x3_fwd(x[n0:].data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                         y[n0:].data_ptr(), k * p, c, cin, 2, h, w, st) == 0
                assert L.smaat_pointwise_fwd_split(y[n0:].data_ptr(), k * p, pl_f.data_ptr(), b_pw.data_ptr(), z[n0:].data_ptr(),
                                                   cout * p, part.data_ptr(), c, k, cout, h, w, st) == 0

        def bwd():
            for n0 in range(0, N, c):
                assert L.smaat_pointwise_fwd_split(dz[n0:].data_ptr(), cout * p, pl_b.data_ptr(), None, dy[n0:].data_ptr(), k * p,
                                                   None, c, cout, k, h, w, st) == 0
                assert L.smaat_dw3x3_bwd(x[n0:].data_ptr(), cin * p, dy[n0:].data_ptr(), k * p, w_dw.data_ptr(), dx[n0:].data_ptr(),
                                         cin * p, ws2.data_ptr(), dwd.data_ptr(), dbd.data_ptr(), c, cin, 2, h, w, st) == 0

        def fwd2():  # producer on stream A, consumer on stream B, chunk-granular events: launches overlap across chunks
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            for n0 in range(0, N, c):
                assert L.smaat_dw3x3_fwd(x[n0:].data_ptr(), cin * p, None, None, w_dw.data_ptr(), b_dw.data_ptr(),
                                         y[n0:].data_ptr(), k * p, c, cin, 2, h, w, sa.cuda_stream) == 0
                ev = torch.cuda.Event()
                ev.record(sa)
                sb.wait_event(ev)
                assert L.smaat_pointwise_fwd_split(y[n0:].data_ptr(), k * p, pl_f.data_ptr(), b_pw.data_ptr(), z[n0:].data_ptr(),
                                                   cout * p, part.data_ptr(), c, k, cout, h, w, sb.cuda_stream) == 0
            cur.wait_stream(sa)
            cur.wait_stream(sb)

        two = f" 2-stream fwd {timeit(fwd2):6.3f}" if (os.environ.get("CP_TWO", "1") == "1" and c < N) else ""
        line += f"  c={c:2d}: fwd {timeit(fwd):6.3f} bwd {timeit(bwd):6.3f}{two} |"
    print(line, flush=True)
