#!/usr/bin/env python
"""HBM ceilings of the box as seen by plain streaming kernels (torch fill / copy / reduce):
pure write, pure read, and 1:1 / 1:4 read:write mixes.  Used to price the write-dominated kernels
(pointwise data gradients, upsample) against what the memory system actually sustains."""
import torch


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    dev = torch.device("cuda:0")
    n = 680 * 1024 * 1024  # floats: 2.7 GB
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    x.normal_()
    gb = n * 4 / 1e9
    ms = t(lambda: y.zero_()); print(f"fill   (write only)     {gb / ms * 1e3:7.1f} GB/s  {ms:.3f} ms")
    ms = t(lambda: y.copy_(x)); print(f"copy   (1 read:1 write) {2 * gb / ms * 1e3:7.1f} GB/s  {ms:.3f} ms")
    ms = t(lambda: x.sum()); print(f"sum    (read only)      {gb / ms * 1e3:7.1f} GB/s  {ms:.3f} ms")
    q = x[: n // 4].view(1, -1)
    y4 = y.view(4, -1)
    ms = t(lambda: torch.add(q, 1.0, out=y4[0:1])); print(f"add    (small)          {2 * gb / 4 / ms * 1e3:7.1f} GB/s  {ms:.3f} ms")
    ms = t(lambda: y4.copy_(q.expand(4, -1))); print(f"bcast  (1 read:4 write) {(gb + gb / 4) / ms * 1e3:7.1f} GB/s  {ms:.3f} ms")


if __name__ == "__main__":
    main()
