#!/usr/bin/env python
"""debug: the fp16 recompute weight gradient with the previous activation applied on load (AFF)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smaat_unet_amd import _lib
from tests.test_gpu_kernels import P, T, rel, rnd, stream
from tests.test_gpu_f16_split import _amax_of, _amax_word, _publish
from oracle import smaat_oracle as O
L, dev = _lib.get(), torch.device("cuda:0")
for aff in (False, True):
    N, Cin, Cout, H, W = 2, 64, 64, 32, 32
    K = 2 * Cin
    xn = rnd(1, N, Cin, H, W)
    w_dwn, b_dwn = rnd(2, K, 9, scale=0.3), rnd(3, K, scale=0.3)
    scn = np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32)
    shn = rnd(7, Cin, scale=0.3)
    x, w_dw, b_dw = T(xn, dev), T(w_dwn, dev), T(b_dwn, dev)
    sc, sh = (T(scn, dev), T(shn, dev)) if aff else (None, None)
    xa = np.maximum(xn * scn[None, :, None, None] + shn[None, :, None, None], 0) if aff else xn
    y64 = O.dw3x3_fwd(xa.astype(np.float64), w_dwn.astype(np.float64).reshape(K, 1, 3, 3), b_dwn.astype(np.float64), 2)
    dzn = rnd(8, N, Cout, H, W) * 1e-3
    dz = T(dzn, dev)
    ref = np.einsum("nmp,nkp->mk", dzn.astype(np.float64).reshape(N, Cout, -1), y64.reshape(N, K, -1))
    w_pw, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    pl = torch.empty((3, Cout, K), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w_pw), Cout, K, P(pl), stream(dev)) == 0
    z = torch.empty((N, Cout, H, W), device=dev)
    ay = _amax_word(dev)
    assert L.smaat_dsconv_fwd_rows_amax(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(pl), P(b_pw), P(z), Cout * H * W, None, P(ay), N,
                                        Cin, 2, Cout, H, W, stream(dev)) == 0
    torch.cuda.synchronize()
    true_bits = int(np.array([np.abs(y64).max()], np.float32).view(np.uint32)[0])
    print("aff", aff, "amax from fwd kernel", hex(_amax_of(ay)), np.array([_amax_of(ay)], np.uint32).view(np.float32)[0], "true", hex(true_bits), np.abs(y64).max())
    adz = _publish(dz)
    ws = torch.empty((L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W), Cout, K), device=dev)
    for nm, aybuf in (("kernel amax", ay), ("exact amax", _publish(torch.from_numpy(y64.astype(np.float32))).to(dev))):
        dw = torch.empty((Cout, K), device=dev)
        assert L.smaat_dsconv_wgrad_split_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(aybuf), P(dz), Cout * H * W, P(adz), P(ws),
                                            P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
        print("   wgrad_h with", nm, "rel vs fp64", rel(dw.cpu().numpy(), ref))
    dw3 = torch.empty((Cout, K), device=dev)
    assert L.smaat_dsconv_wgrad_split(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(dz), Cout * H * W, P(ws), P(dw3), N, Cin, 2, Cout, H,
                                      W, stream(dev)) == 0
    print("   3-term rel vs fp64", rel(dw3.cpu().numpy(), ref))
    if aff:
        print("   repeatability and placement of the maximum inside the buffer:")
        ex = _publish(torch.from_numpy(y64.astype(np.float32)))
        for nm, mk in (("word 0", lambda: ex.clone()), ("word 0 again", lambda: ex.clone()), ("word 160", lambda: torch.roll(ex, 160)),
                       ("kernel buffer", lambda: ay.cpu().clone()), ("kernel buffer again", lambda: ay.cpu().clone()),
                       ("all 32 slots", lambda: ex[0].repeat(1024) * (torch.arange(1024) % 32 == 0))):
            b = mk().to(torch.int32).to(dev)
            dw = torch.empty((Cout, K), device=dev)
            assert L.smaat_dsconv_wgrad_split_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(b), P(dz), Cout * H * W, P(adz), P(ws),
                                                P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
            print("     ", nm, "max bits", hex(int(b.max())), "rel vs fp64", rel(dw.cpu().numpy(), ref))
