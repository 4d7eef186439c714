#!/usr/bin/env python
"""The non-walking depthwise forward (k_dw3x3_fwd_lin) against the row walker (k_dw3x3_fwd_rows) on the depthwise shapes of the
step at batch 32: time of each (SMAAT_DW_LIN=1 / 0, read at every call) and bit-equality of y and of the published maximum."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402


def main():
    L = _lib.get()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    N = int(os.environ.get("DWL_BATCH", "32"))
    for cin, h in ((12, 288), (64, 288), (64, 144), (128, 144), (256, 144), (128, 72), (256, 72), (512, 72), (256, 36), (512, 36),
                   (1024, 36), (512, 18)):
        k, p = 2 * cin, h * h
        x = torch.randn(N, cin, h, h, device=dev)
        w = torch.randn(k, 9, device=dev) * 0.3
        b = torch.randn(k, device=dev) * 0.1
        sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.3
        res = {}
        for aff in (False, True):
            out = {}
            for mode in ("0", "1"):
                os.environ["SMAAT_DW_LIN"] = mode
                y = torch.full((N, k, h, h), float("nan"), device=dev)
                am = torch.zeros(1024, dtype=torch.int32, device=dev)

                def f():
                    return L.smaat_dw3x3_fwd_amax(x.data_ptr(), cin * p, sc.data_ptr() if aff else None, sh.data_ptr() if aff else None,
                                                  w.data_ptr(), b.data_ptr(), y.data_ptr(), k * p, am.data_ptr(), N, cin, 2, h, h, st)
                rc = f()
                if rc != 0:
                    out[mode] = None
                    continue
                for _ in range(2):
                    f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f()
                e1.record()
                torch.cuda.synchronize()
                out[mode] = (e0.elapsed_time(e1) / 10, y, int(am.max()))
            if out["0"] is None or out["1"] is None:
                res[aff] = "n/a"
                continue
            same = torch.equal(out["0"][1], out["1"][1]) and out["0"][2] == out["1"][2]
            gb = 4.0 * N * (cin + k) * p / 1e9
            res[aff] = f"walker {out['0'][0] * 1e3:7.1f} us {gb / out['0'][0]:5.2f} TB/s  lin {out['1'][0] * 1e3:7.1f} us {gb / out['1'][0]:5.2f} TB/s  {'bit-identical' if same else 'DIFFERENT'}"
        print(f"{cin:5d} x {h:3d}^2  plain: {res[False]}  |  act on load: {res[True]}", flush=True)
    os.environ.pop("SMAAT_DW_LIN", None)


if __name__ == "__main__":
    main()
