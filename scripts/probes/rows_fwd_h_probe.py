"""Round 6: the row-walking fused forward on the two-term fp16 split with an a-priori bound of |y| (smaat_dsconv_fwd_rows_h)
against the three-term bf16 form it would replace (smaat_dsconv_fwd_rows_amax): error against an fp64 evaluation and time,
on the three layers of the step that run it (batch 32).  python scripts/probes/rows_fwd_h_probe.py [batch]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from smaat_unet_amd import _lib  # noqa: E402

AMAX_WORDS = 1024


def P(t):
    return None if t is None else t.data_ptr()


def publish(t):
    w = torch.zeros(AMAX_WORDS, dtype=torch.int32, device=t.device)
    w[0] = np.array([float(t.abs().max())], np.float32).view(np.int32)[0].item()
    return w


def amax_of(buf):
    return float(np.array([int(buf.max().item())], np.int32).view(np.float32)[0])


def run(L, dev, name, N, Cin, Cout, H, W, aff, xscale=1.0, loose=1.0):
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device="cpu").manual_seed(1)
    K = 2 * Cin
    x = (torch.randn(N, Cin, H, W, generator=g) * xscale).to(dev)
    w_dw = (torch.randn(K, 9, generator=g) * 0.3).to(dev)
    b_dw = (torch.randn(K, generator=g) * 0.3).to(dev)
    w_pw = (torch.randn(Cout, K, generator=g) * 0.2).to(dev)
    b_pw = torch.randn(Cout, generator=g).to(dev)
    sc = (torch.rand(Cin, generator=g) + 0.5).to(dev) if aff else None
    sh = (torch.randn(Cin, generator=g) * 0.3).to(dev) if aff else None
    Kp = (K + 15) // 16 * 16
    pl3 = torch.empty((3, Cout, Kp), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w_pw), Cout, K, P(pl3), st) == 0
    plh = torch.full((int(L.smaat_split_planes_h_bytes(Cout, K)) // 2,), -1, dtype=torch.int16, device=dev)
    assert L.smaat_split_planes_h(P(w_pw), Cout, K, P(plh), 0, st) == 0
    slots = L.smaat_dsconv_rows_num_slots(N, H, W)
    z3 = torch.empty((N, Cout, H, W), device=dev)
    zh = torch.empty_like(z3)
    part3 = torch.empty((3, slots, Cout), device=dev)
    parth = torch.empty_like(part3)
    ay3 = torch.zeros(AMAX_WORDS, dtype=torch.int32, device=dev)
    ayh = torch.zeros_like(ay3)
    az = torch.zeros_like(ay3)
    ax = publish(x * loose)

    def f3():
        return L.smaat_dsconv_fwd_rows_amax(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(pl3), P(b_pw), P(z3), Cout * H * W,
                                            P(part3), P(ay3), N, Cin, 2, Cout, H, W, st)

    def fh():
        return L.smaat_dsconv_fwd_rows_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ax), None, P(plh), P(b_pw), P(zh),
                                         Cout * H * W, P(parth), P(ayh), P(az), N, Cin, 2, Cout, H, W, st)

    assert f3() == 0
    assert fh() == 0
    torch.cuda.synchronize()
    # fp64 evaluation (two images are enough for the error figure)
    n2 = min(N, 2)
    xd = x[:n2].double()
    if aff:
        xd = torch.relu(xd * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    yd = torch.nn.functional.conv2d(xd, w_dw.double().view(K, 1, 3, 3), b_dw.double(), padding=1, groups=Cin)
    zd = torch.nn.functional.conv2d(yd, w_pw.double().view(Cout, K, 1, 1), b_pw.double())
    den = zd.abs().max()
    e3 = float((z3[:n2].double() - zd).abs().max() / den)
    eh = float((zh[:n2].double() - zd).abs().max() / den)
    r3 = float((z3[:n2].double() - zd).norm() / zd.norm())
    rh = float((zh[:n2].double() - zd).norm() / zd.norm())
    ymax = float(yd.abs().max())

    def timeit(f, n=20):
        for _ in range(3):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    t3, th = timeit(f3), timeit(fh)
    print(f"{name:8s} N={N} Cin={Cin:3d} aff={int(aff)} loose x{loose:g}: 3-term {t3:.3f} ms  fp16 2-term {th:.3f} ms ({th / t3:.2f}x) | "
          f"max err / max|z|: {e3:.2e} -> {eh:.2e}   rel-L2: {r3:.2e} -> {rh:.2e} | max|y| true {ymax:.3g}, kernel's {amax_of(ayh):.3g} "
          f"(3-term {amax_of(ay3):.3g}); max|z| {amax_of(az):.4g} vs {float(zh.abs().max()):.4g}; parts equal-ish "
          f"{float((part3 - parth).abs().max()):.2e}", flush=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L, dev = _lib.get(), torch.device("cuda:0")
    run(L, dev, "inc.1", B, 64, 64, 288, 288, True)
    run(L, dev, "up4.0", B, 128, 64, 288, 288, False)
    run(L, dev, "up4.1", B, 64, 64, 288, 288, True)
    run(L, dev, "small", 2, 64, 64, 32, 32, True)
    run(L, dev, "small", 2, 128, 64, 36, 96, False)
    run(L, dev, "loose", 2, 64, 64, 64, 64, True, loose=1024.0)
    run(L, dev, "loose", 2, 128, 64, 64, 64, False, loose=65536.0)
    run(L, dev, "tinyx", 2, 64, 64, 64, 64, False, xscale=1e-20)
    run(L, dev, "hugex", 2, 64, 64, 64, 64, False, xscale=1e20)


if __name__ == "__main__":
    main()
