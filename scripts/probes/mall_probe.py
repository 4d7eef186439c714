#!/usr/bin/env python
"""Does a tensor that one kernel WRITES come back from the memory-side cache (MI355X: 256 MB Infinity Cache) when the next
kernel READS it?  Producer = smaat_affine_act (streams src -> buf), consumer = smaat_cbam_chpool (reads buf once).  The
consumer is timed right after the producer ("warm") and after a 2 GB flush of other data ("cold"), for several sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
P = 288 * 288
flush_a = torch.empty(512 * 1024 * 1024 // 4, device=dev)
flush_b = torch.empty_like(flush_a)


def run(mb):
    planes = mb * 1024 * 1024 // (4 * P)
    src = torch.randn(1, planes, 288, 288, device=dev)
    buf = torch.empty_like(src)
    sc, sh = torch.ones(planes, device=dev), torch.zeros(planes, device=dev)
    avg, mx = torch.empty(planes, device=dev), torch.empty(planes, device=dev)
    am = torch.empty(planes, dtype=torch.int32, device=dev)

    def produce():
        assert L.smaat_affine_act(src.data_ptr(), planes * P, sc.data_ptr(), sh.data_ptr(), buf.data_ptr(), planes * P, 1, planes,
                                  P, 1, st) == 0

    def consume():
        assert L.smaat_cbam_chpool(buf.data_ptr(), planes * P, 1, planes, P, avg.data_ptr(), mx.data_ptr(), am.data_ptr(), st) == 0

    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(5):
            produce()
            if mode == "cold":
                flush_b.copy_(flush_a)
                flush_a.copy_(flush_b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            consume()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    nbytes = planes * P * 4
    print(f"{mb:5d} MB  read after write: warm {res['warm']*1e3:8.1f} us = {nbytes/res['warm']/1e9:6.2f} TB/s   "
          f"cold {res['cold']*1e3:8.1f} us = {nbytes/res['cold']/1e9:6.2f} TB/s")


for mb in (32, 64, 96, 128, 192, 256, 384, 512, 1024):
    run(mb)
