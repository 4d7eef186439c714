#!/usr/bin/env python
"""k_bn_bwd_apply with the (planes, segments) grid against the one-dimensional grid in address order (SMAAT_BN_LIN = segment length,
read once per process: one subprocess per setting), on the BatchNorm shapes of the step at batch 32."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from smaat_unet_amd import _lib
L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
out = []
for C, H in ((64, 288), (128, 144), (256, 72), (512, 36), (512, 18)):
    N, P = 32, H * H
    dy, z = torch.randn(N, C, H, H, device=dev), torch.randn(N, C, H, H, device=dev)
    dz = torch.empty_like(z)
    v = [torch.rand(C, device=dev) + 0.5 for _ in range(4)]
    coef = torch.rand(3, C, device=dev)
    am = torch.zeros(1024, dtype=torch.int32, device=dev)
    def f():
        assert L.smaat_bn_bwd_apply_amax(dy.data_ptr(), C * P, None, z.data_ptr(), C * P, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                         v[3].data_ptr(), coef.data_ptr(), dz.data_ptr(), C * P, am.data_ptr(), N, C, P, 1, st) == 0
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append(f"{C}x{H}^2 {ms * 1e3:7.1f} us {12.0 * N * C * P / ms / 1e6:6.0f} GB/s")
print(" | ".join(out))
''' % ROOT

for lin in ("0", "1024", "2048", "4096", "8192"):
    r = subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, "SMAAT_BN_LIN": lin}, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if "GB/s" in ln]
    print(f"SMAAT_BN_LIN={lin:5s} " + (line[-1] if line else "FAILED " + r.stderr[-300:]), flush=True)
