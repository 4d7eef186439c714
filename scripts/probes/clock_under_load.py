#!/usr/bin/env python
"""What does the shader clock do while the kernels run?  rocm-smi is sampled from a second thread while the main thread keeps
one kernel class launched back to back for a few seconds: the exact-f32 split GEMM of a deep layer (MFMA + L2 + HBM), the
bf16 GEMM of the same layer, a pure streaming kernel (BatchNorm backward apply), and idle.  The PMC "busy" percentages are
in cycles, so a clock below the nominal 2.4 GHz lowers the attainable TFLOP/s without showing in them."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib  # noqa: E402

L = _lib.get()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
N, cin, cout, h = 32, 1024, 512, 36
k, p = 2 * cin, h * h
y = torch.randn(N, k, h, h, device=dev)
w_pw, b_pw = torch.randn(cout, k, device=dev) * 0.1, torch.randn(cout, device=dev)
z = torch.empty(N, cout, h, h, device=dev)
pl = torch.empty(3, cout, (k + 15) // 16 * 16, dtype=torch.int16, device=dev)
assert L.smaat_split_planes(w_pw.data_ptr(), cout, k, pl.data_ptr(), st) == 0
part = torch.empty(3, L.smaat_pw_split_num_slots(N, h, h), cout, device=dev)
yb, zb = y.to(torch.bfloat16), torch.empty(N, cout, h, h, device=dev, dtype=torch.bfloat16)
plb = torch.empty(((k + 31) // 32 * 2, cout, 16), dtype=torch.int16, device=dev)
assert L.smaat_bf16_planes(w_pw.data_ptr(), cout, k, plb.data_ptr(), 0, st) == 0
big = torch.randn(32, 64, 288, 288, device=dev)
big2 = torch.empty_like(big)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)


def gemm_f32():
    assert L.smaat_pointwise_fwd_split(y.data_ptr(), k * p, pl.data_ptr(), b_pw.data_ptr(), z.data_ptr(), cout * p, part.data_ptr(),
                                       N, k, cout, h, h, st) == 0


def gemm_bf16():
    assert L.smaat_pointwise_fwd_bf16(yb.data_ptr(), k * p, plb.data_ptr(), b_pw.data_ptr(), zb.data_ptr(), cout * p, 1,
                                      part.data_ptr(), N, k, cout, h, h, 0, st) == 0


def stream():
    assert L.smaat_affine_act(big.data_ptr(), 64 * 288 * 288, sc.data_ptr(), sh.data_ptr(), big2.data_ptr(), 64 * 288 * 288, 32, 64,
                              288 * 288, 1, st) == 0


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return str(e)
    s = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out) or re.findall(r"sclk[^\n]*?\((\d+)Mhz\)", out)
    pw = re.findall(r"Power \(W\): ([\d.]+)", out) or re.findall(r"Socket Power[^\n]*?([\d.]+)\s*$", out, re.M)
    return f"sclk {s} MHz  power {pw} W"


steps_f32 = None  # (defined below)


def run(name, fn, seconds=4.0):
    stop = threading.Event()
    samples = []

    def sampler():
        time.sleep(1.0)
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.7)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        reps = 2 if fn is steps_f32 else 50
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
        n += reps
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    rate = ""
    if fn is steps_f32 and fn is not None:
        rate = f"  {32.0 * n / dt:7.1f} frames/s"
    if fn is gemm_f32 or fn is gemm_bf16:
        rate = f"  {2.0 * N * k * cout * p * n / dt / 1e12:7.1f} TFLOP/s algorithmic"
    print(f"{name:34s}{rate}")
    for s_ in samples:
        print("     ", s_)


import smaat_unet_amd as S  # noqa: E402

torch.manual_seed(0)
model = S.SmaAt_UNet(12, 1).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, foreach=True)
xs, ys = torch.rand(32, 12, 288, 288, device=dev), torch.rand(32, 288, 288, device=dev)


def make_step(m, x, yv):
    def f():
        out = m(x)
        loss = torch.nn.functional.mse_loss(out.squeeze(1).float(), yv, reduction="sum") / yv.size(0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    return f


step_f32 = make_step(model, xs, ys)


def steps_f32():  # run() launches fn 50 times between synchronisations: one step per call is enough here
    step_f32()


run("idle", None, 2.5)
run("whole training step, f32, B=32", steps_f32, 5.0)
run("split GEMM up1.0 (f32, 6 MFMAs)", gemm_f32)
run("bf16 GEMM up1.0", gemm_bf16)
run("streaming (affine_act 288^2)", stream)
out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
print(out[-1500:])
