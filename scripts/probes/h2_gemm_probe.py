"""Round 6: what a pointwise GEMM on PRE-SPLIT fp16 operand planes reaches -- the prototype
smaat_pointwise_fwd_h2_proto (csrc/h2gemm.hip: LDS-DMA, transposed LDS reads, no VALU in the loop) against the shipped
smaat_pointwise_fwd_split_h (producer waves split the f32 activation in VALU) on the GEMM-sized layers of the step, batch 32:
forward shapes (K = 2 Cin -> Cout) and data-gradient shapes (Cout -> K).  Error of both against fp64.
python scripts/probes/h2_gemm_probe.py [batch]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from smaat_unet_amd import _lib  # noqa: E402

AMAX_WORDS = 1024


def P(t):
    return None if t is None else t.data_ptr()


def kexp(t):
    am = np.array([float(t.abs().max())], np.float32).view(np.uint32)[0]
    e = (int(am) >> 23) & 0xFF
    if e == 255 or (int(am) & 0x7FFFFFFF) == 0:
        return 0
    return max(-126, min(126, 141 - e))


def planes_of(t, k):
    ts = t * (2.0 ** k)
    h = ts.half()
    g = (ts - h.float()).half()
    return h, g


def timeit(f, n=30):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def run(L, dev, name, N, C, M, H, W):
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device="cpu").manual_seed(1)
    Pn = H * W
    x = torch.randn(N, C, Pn, generator=g).to(dev)
    w = (torch.randn(M, C, generator=g) * 0.1).to(dev)
    bias = torch.randn(M, generator=g).to(dev)
    # shipped kernel
    plh = torch.full((int(L.smaat_split_planes_h_bytes(M, C)) // 2,), -1, dtype=torch.int16, device=dev)
    assert L.smaat_split_planes_h(P(w), M, C, P(plh), 0, st) == 0
    ax = torch.zeros(AMAX_WORDS, dtype=torch.int32, device=dev)
    ax[0] = np.array([float(x.abs().max())], np.float32).view(np.int32)[0].item()
    z0 = torch.empty(N, M, Pn, device=dev)
    f0 = lambda: L.smaat_pointwise_fwd_split_h(P(x), C * Pn, P(ax), P(plh), P(bias), P(z0), M * Pn, None, N, C, M, H, W, st)  # noqa: E731
    assert f0() == 0
    # prototype operands: activation planes [N][2][C][P], weight planes [2][C/16][M][16]
    kx, ka = kexp(x), kexp(w)
    xh, xg = planes_of(x, kx)
    xp = torch.stack([xh, xg], dim=1).contiguous()
    wh, wg = planes_of(w, ka)
    ap = torch.stack([t.view(M, C // 16, 16).permute(1, 0, 2).contiguous() for t in (wh, wg)], dim=0).contiguous()
    z1 = torch.empty(N, M, Pn, device=dev)
    res = {}
    for cfg in (0, 1, 2):
        f1 = lambda: L.smaat_pointwise_fwd_h2_proto(P(xp), 2 * C * Pn, C * Pn, P(ap), (C // 16) * M * 16, P(bias), P(z1), M * Pn, None,  # noqa: E731
                                                    N, C, M, H, W, ka + kx, cfg, st)
        z1.fill_(float("nan"))
        rc = f1()
        if rc != 0:
            res[cfg] = None
            continue
        torch.cuda.synchronize()
        n2 = min(N, 2)
        zd = torch.einsum("mc,ncp->nmp", w.double(), x[:n2].double()) + bias.double().view(1, -1, 1)
        e1 = float((z1[:n2].double() - zd).norm() / zd.norm())
        res[cfg] = (timeit(f1), e1, bool(torch.isfinite(z1).all()))
    n2 = min(N, 2)
    zd = torch.einsum("mc,ncp->nmp", w.double(), x[:n2].double()) + bias.double().view(1, -1, 1)
    e0 = float((z0[:n2].double() - zd).norm() / zd.norm())
    t0 = timeit(f0)
    fl = 2.0 * N * C * M * Pn
    s = f"{name:14s} N={N} {C:4d}->{M:4d} {H}x{W}: shipped {t0 * 1e3:7.1f} us {fl / t0 / 1e9:6.1f} TF (err {e0:.1e}) |"
    for cfg in (0, 1, 2):
        r = res[cfg]
        s += f" cfg{cfg} " + ("refused" if r is None else f"{r[0] * 1e3:7.1f} us {fl / r[0] / 1e9:6.1f} TF ({t0 / r[0]:.2f}x, err {r[1]:.1e}{'' if r[2] else ' NONFINITE'})") + " |"
    print(s, flush=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L, dev = _lib.get(), torch.device("cuda:0")
    shapes = [("down2.1 fwd", 512, 256, 72), ("down3.0 fwd", 512, 512, 36), ("down3.1 fwd", 1024, 512, 36), ("down4.0 fwd", 1024, 512, 18),
              ("up1.0 fwd", 2048, 512, 36), ("up1.1 fwd", 1024, 256, 36), ("up2.0 fwd", 1024, 256, 72), ("up2.1 fwd", 512, 128, 72),
              ("up3.0 fwd", 512, 128, 144), ("down1.1 fwd", 256, 128, 144),
              ("up1.0 dgrad", 512, 2048, 36), ("down3.1 dgrad", 512, 1024, 36), ("up2.0 dgrad", 256, 1024, 72), ("down2.1 dgrad", 256, 512, 72),
              ("up3.0 dgrad", 128, 512, 144), ("down1.1 dgrad", 128, 256, 144), ("up4.0 dgrad", 64, 256, 288)]
    for name, C, M, S in shapes:
        try:
            run(L, dev, name, B, C, M, S, S)
        except Exception as e:  # noqa: BLE001
            print(name, "failed:", repr(e)[:200], flush=True)


if __name__ == "__main__":
    main()
