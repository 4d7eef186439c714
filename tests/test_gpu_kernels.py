"""GPU: every C-ABI entry point of libsmaat_hip.so against its numpy emulation
(tests/emu_backend.py, built on the oracle) on identical seeded inputs.
Tolerance: rel-L2 <= 1e-5 (fp32, north_star asks 1e-4) unless stated."""
import os

import numpy as np
import pytest
import torch

from smaat_unet_amd import _lib
from tests.emu_backend import EmuLib

pytestmark = pytest.mark.gpu


def _backends():
    return (("emu", EmuLib(), torch.device("cpu")), ("hip", _lib.get(), torch.device("cuda:0")))


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def P(t):
    return None if t is None else t.data_ptr()


def stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None


def both(case, *args, tol=1e-5, **kw):
    res = {}
    for name, L, dev in _backends():
        out = case(L, dev, *args, **kw)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        res[name] = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    for k in res["emu"]:
        e = rel(res["hip"][k], res["emu"][k])
        assert e <= tol, f"{k}: rel err {e:.3e} > {tol}  (max abs {np.abs(res['hip'][k]-res['emu'][k]).max():.3e})"
    return res


def part_stats(part):
    """per-tile BatchNorm partials [3][T][C] = (mean, M2, count) -> (total count, mean, biased variance) per channel,
    merged in fp64 exactly as include/smaat_hip.h specifies for smaat_bn_finalize"""
    pp = part.double()
    n = pp[2].sum(0)
    mean = (pp[2] * pp[0]).sum(0) / n.clamp(min=1)
    var = (pp[1] + pp[2] * (pp[0] - mean[None]) ** 2).sum(0) / n.clamp(min=1)
    return n, mean, var


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


# ----------------------------------------------------------------------------------------
def case_dsconv_fwd(L, dev, N, Cin, kpl, Cout, H, W, aff=False, pad_c=0, bias=True):
    K = Cin * kpl
    xfull = T(rnd(1, N, Cin + pad_c, H, W), dev)
    x = xfull[:, pad_c:]  # channel slice of a bigger buffer -> batch stride > Cin*H*W
    x_bs = (Cin + pad_c) * H * W
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    wt, b_pw = T(rnd(4, K, Cout, scale=0.2), dev), T(rnd(5, Cout), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    z = torch.full((N, Cout, H, W), float("nan"), device=dev)
    slots = L.smaat_pw_num_slots(N, H, W, Cout)
    part = torch.full((3, slots, Cout), float("nan"), device=dev)
    y = torch.full((N, K, H, W), float("nan"), device=dev)
    xptr = x.data_ptr()
    rc = L.smaat_dsconv_fwd(xptr, x_bs, P(sc), P(sh), P(w_dw), P(b_dw) if bias else None, P(wt),
                            P(b_pw) if bias else None, P(z), Cout * H * W, P(part), P(y), N, Cin, kpl, Cout, H, W,
                            stream(dev))
    assert rc == 0
    pn, pmean, pvar = part_stats(part)
    return dict(z=z, y=y, pn=pn, pmean=pmean, pvar=pvar)


DS_SHAPES = [
    # N, Cin, kpl, Cout, H, W
    (2, 12, 2, 64, 32, 32),     # inc.0-like, flattened tiles
    (2, 6, 2, 10, 9, 11),       # odd everything, partial co tile
    (1, 3, 4, 8, 6, 10),        # kpl 4
    (2, 5, 1, 7, 8, 8),         # kpl 1
    (2, 64, 2, 128, 36, 36),    # flattened mode, 128-wide co tile
    (3, 16, 2, 130, 18, 18),    # co not a multiple of 64/128
    (2, 64, 2, 64, 72, 72),     # 72-wide flattened
    (2, 32, 2, 64, 144, 144),   # 16-wide 2D tiles
    (4, 12, 2, 64, 288, 288),   # "big" config (256-pixel tiles, 8x32)
    (4, 8, 2, 128, 288, 288),   # big + 128 co tile
    (4, 40, 2, 130, 288, 288),  # big, 5 contraction chunks, 2 co tiles (one partial)
    (4, 20, 2, 40, 288, 288),   # big, 64-wide co tile (partial), 3 chunks
    (1, 4, 2, 16, 100, 100),    # partial 2D tiles
    (2, 24, 2, 32, 4, 4),       # tiny maps (64x64 config bottoms out at 4x4)
    (1, 8, 2, 16, 2, 2),
]


@pytest.mark.parametrize("shape", DS_SHAPES)
def test_dsconv_fwd(shape):
    both(case_dsconv_fwd, *shape)


def test_dsconv_fwd_affine_slice_nobias():
    both(case_dsconv_fwd, 2, 12, 2, 64, 32, 32, aff=True, pad_c=4)
    both(case_dsconv_fwd, 2, 6, 2, 10, 9, 11, bias=False)


def case_pointwise(L, dev, N, C, M, H, W, with_part=False):
    x = T(rnd(1, N, C, H, W), dev)
    wt, b = T(rnd(2, C, M, scale=0.2), dev), T(rnd(3, M), dev)
    out = torch.full((N, M, H, W), float("nan"), device=dev)
    slots = L.smaat_pw_num_slots(N, H, W, M)
    part = torch.zeros((3, slots, M), device=dev) if with_part else None
    assert L.smaat_pointwise_fwd(P(x), C * H * W, P(wt), P(b), P(out), M * H * W, P(part), N, C, M, H, W,
                                 stream(dev)) == 0
    r = dict(out=out)
    if with_part:
        r["pn"], r["pmean"], r["pvar"] = part_stats(part)
    return r


@pytest.mark.parametrize("shape", [(2, 64, 1, 32, 32), (2, 64, 21, 16, 16), (2, 16, 3, 6, 7), (2, 128, 256, 36, 36),
                                   (4, 64, 1, 288, 288), (2, 1, 64, 20, 20), (2, 512, 1024, 18, 18),
                                   (4, 40, 130, 288, 288)])
def test_pointwise_fwd(shape):
    both(case_pointwise, *shape)
    both(case_pointwise, *shape, with_part=True)


def case_dsconv_wgrad(L, dev, N, Cin, kpl, Cout, H, W):
    K = Cin * kpl
    x = T(rnd(1, N, Cin, H, W), dev)
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    dz = T(rnd(4, N, Cout, H, W), dev)
    ns = L.smaat_dsconv_wgrad_num_splits(N, H, W, Cout, K)
    ws = torch.empty((ns, Cout, K), device=dev)
    dw = torch.full((Cout, K), float("nan"), device=dev)
    assert L.smaat_dsconv_wgrad(P(x), Cin * H * W, None, None, P(w_dw), P(b_dw), P(dz), Cout * H * W, P(ws), P(dw), N,
                                Cin, kpl, Cout, H, W, stream(dev)) == 0
    return dict(dw=dw)


@pytest.mark.parametrize("shape", [(2, 12, 2, 64, 32, 32), (2, 6, 2, 10, 9, 11), (1, 3, 4, 8, 6, 10),
                                   (2, 5, 1, 7, 8, 8), (2, 64, 2, 128, 36, 36), (2, 40, 2, 130, 18, 18),
                                   (2, 16, 2, 64, 144, 144), (2, 12, 2, 64, 288, 288), (1, 4, 2, 16, 100, 100),
                                   (2, 24, 2, 32, 4, 4)])
def test_dsconv_wgrad(shape):
    both(case_dsconv_wgrad, *shape, tol=2e-5)


def case_dsconv_wgrad_split(L, dev, N, Cin, Cout, H, W, aff=False, bias=True, pad_c=0):
    """round 4: weight gradient on the split matrix path with the depthwise output recomputed by the producer waves
    (csrc/dswgrad.hip); x may be a channel slice of a larger buffer (batch stride), the previous BatchNorm + ReLU is
    optionally applied on load"""
    K = Cin * 2
    xfull = T(rnd(1, N, Cin + pad_c, H, W), dev)
    x = xfull[:, pad_c:]
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    dz = T(rnd(4, N, Cout, H, W), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    assert L.smaat_dsconv_wgrad_split_ok(2, Cout, H, W) == 1
    ns = L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W)
    ws = torch.full((ns, Cout, K), float("nan"), device=dev)
    dw = torch.full((Cout, K), float("nan"), device=dev)
    assert L.smaat_dsconv_wgrad_split(x.data_ptr(), (Cin + pad_c) * H * W, P(sc), P(sh), P(w_dw), P(b_dw) if bias else None,
                                      P(dz), Cout * H * W, P(ws), P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    return dict(dw=dw)


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 40, 50, 8, 64), (2, 100, 64, 36, 96), (1, 8, 16, 70, 32),
                                   (3, 130, 33, 5, 32), (2, 12, 64, 288, 288), (1, 64, 64, 50, 160), (2, 32, 1, 1, 32),
                                   (9, 4, 8, 2, 64), (1, 8, 16, 67, 32)])  # (67: a prime height -> a short last band)
def test_dsconv_wgrad_split(shape):
    both(case_dsconv_wgrad_split, *shape, tol=1e-5)
    both(case_dsconv_wgrad_split, *shape, aff=True, bias=False, pad_c=3, tol=1e-5)


def case_dsconv_wgrad_split_t(L, dev, N, Cin, Cout, H, W, x_bf=True, aff=False, pad_c=0):
    """typed form, mixed precision: bf16 dz (the MFMA operand as stored), x bf16 or f32 (the stem), plain bf16 operands"""
    K = Cin * 2
    xfull = T(rnd(1, N, Cin + pad_c, H, W), dev)
    if x_bf:
        xfull = xfull.to(torch.bfloat16)
    x = xfull[:, pad_c:]
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    dz = T(rnd(4, N, Cout, H, W), dev).to(torch.bfloat16)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    ws = torch.full((L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W), Cout, K), float("nan"), device=dev)
    dw = torch.full((Cout, K), float("nan"), device=dev)
    assert L.smaat_dsconv_wgrad_split_t(x.data_ptr(), 1 if x_bf else 0, (Cin + pad_c) * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(dz),
                                        1, Cout * H * W, P(ws), P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    return dict(dw=dw)


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 40, 50, 8, 64), (2, 100, 64, 36, 96), (1, 8, 16, 70, 32)])
def test_dsconv_wgrad_split_bf16_storage(shape):
    # (y is rounded to bf16 once before the MFMA: an f32 rounding difference in y moves single products by 2^-8; the sums
    # over thousands of pixels agree to ~1e-3 of the gradient norm at worst, measured ~1e-4)
    both(case_dsconv_wgrad_split_t, *shape, tol=2e-3)
    both(case_dsconv_wgrad_split_t, *shape, x_bf=False, aff=True, pad_c=3, tol=2e-3)


def test_dsconv_wgrad_split_against_fp64_and_the_streamed_kernel():
    """f32-class error: against an fp64 evaluation the recompute kernel is as close as the streamed split kernel
    (k_wgrad_split on the kept depthwise output) and as the f32-MFMA kernel; it refuses what it does not take"""
    L, dev = _lib.get(), torch.device("cuda:0")
    N, Cin, Cout, H, W = 2, 96, 64, 72, 64
    K = 2 * Cin
    x, dz = rnd(1, N, Cin, H, W), rnd(4, N, Cout, H, W)
    w_dw, b_dw = rnd(2, K, 9, scale=0.3), rnd(3, K, scale=0.3)
    from oracle import smaat_oracle as O
    y64 = O.dw3x3_fwd(x.astype(np.float64), w_dw.astype(np.float64).reshape(K, 1, 3, 3), b_dw.astype(np.float64), 2)
    ref = np.einsum("nmp,nkp->mk", dz.astype(np.float64).reshape(N, Cout, -1), y64.reshape(N, K, -1))
    new = case_dsconv_wgrad_split(L, dev, N, Cin, Cout, H, W)["dw"].cpu().numpy()
    yt = torch.empty((N, K, H, W), device=dev)
    assert L.smaat_dw3x3_fwd(P(T(x, dev)), Cin * H * W, None, None, P(T(w_dw, dev)), P(T(b_dw, dev)), P(yt), K * H * W, N, Cin, 2,
                             H, W, stream(dev)) == 0
    ws = torch.empty((L.smaat_wgrad_num_splits(N, H, W, Cout, K), Cout, K), device=dev)
    old = torch.empty((Cout, K), device=dev)
    assert L.smaat_pointwise_wgrad(P(yt), K * H * W, P(T(dz, dev)), Cout * H * W, P(ws), P(old), N, K, Cout, H, W,
                                   stream(dev)) == 0
    e_new, e_old = rel(new, ref), rel(old.cpu().numpy(), ref)
    assert e_new < 2e-6 and e_new < 3 * e_old + 2e-7, (e_new, e_old)
    # run to run bit-identical (fixed-order reduction, no atomics)
    again = case_dsconv_wgrad_split(L, dev, N, Cin, Cout, H, W)["dw"].cpu().numpy()
    assert np.array_equal(new, again)
    # refusals
    assert L.smaat_dsconv_wgrad_split_ok(2, 64, 18, 18) == 0 and L.smaat_dsconv_wgrad_split_ok(4, 64, 32, 32) == 0
    assert L.smaat_dsconv_wgrad_split_ok(2, 128, 32, 32) == 0
    t = torch.zeros(1, 4, 18, 18, device=dev)
    assert L.smaat_dsconv_wgrad_split(P(t), 4 * 324, None, None, P(t), None, P(t), 4 * 324, P(t), P(t), 1, 4, 2, 4, 18, 18,
                                      stream(dev)) == -2


def case_pointwise_wgrad(L, dev, N, C, M, H, W):
    x, dz = T(rnd(1, N, C, H, W), dev), T(rnd(2, N, M, H, W), dev)
    ns = L.smaat_wgrad_num_splits(N, H, W, M, C)
    ws = torch.empty((ns, M, C), device=dev)
    dw = torch.full((M, C), float("nan"), device=dev)
    assert L.smaat_pointwise_wgrad(P(x), C * H * W, P(dz), M * H * W, P(ws), P(dw), N, C, M, H, W, stream(dev)) == 0
    return dict(dw=dw)


@pytest.mark.parametrize("shape", [(2, 64, 1, 32, 32), (2, 64, 21, 16, 16), (2, 16, 3, 6, 7), (2, 64, 1, 288, 288),
                                   (2, 128, 64, 36, 36), (2, 24, 64, 64, 64), (3, 130, 70, 9, 11), (2, 256, 200, 18, 18),
                                   (1, 1024, 512, 18, 18), (2, 256, 64, 144, 144), (1, 5, 3, 7, 9)])
def test_pointwise_wgrad(shape):
    both(case_pointwise_wgrad, *shape, tol=2e-5)


def case_dw_bwd(L, dev, N, Cin, kpl, H, W, need_dx=True):
    K = Cin * kpl
    x, dy = T(rnd(1, N, Cin, H, W), dev), T(rnd(2, N, K, H, W), dev)
    w_dw = T(rnd(3, K, 9, scale=0.3), dev)
    dx = torch.full((N, Cin, H, W), float("nan"), device=dev) if need_dx else None
    ws = torch.empty((L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W), K, 10), device=dev)
    dw, db = torch.full((K, 9), float("nan"), device=dev), torch.full((K,), float("nan"), device=dev)
    assert L.smaat_dw3x3_bwd(P(x), Cin * H * W, P(dy), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws), P(dw), P(db), N,
                             Cin, kpl, H, W, stream(dev)) == 0
    r = dict(dw=dw, db=db)
    if need_dx:
        r["dx"] = dx
    return r


@pytest.mark.parametrize("shape", [(2, 12, 2, 32, 32), (2, 6, 2, 9, 11), (1, 3, 4, 6, 10), (2, 5, 1, 8, 8),
                                   (2, 8, 2, 36, 36), (2, 4, 2, 144, 144), (2, 3, 2, 288, 288), (1, 4, 2, 100, 100),
                                   (2, 8, 2, 4, 4), (1, 8, 2, 2, 2), (1, 3, 4, 8, 12), (2, 4, 2, 72, 72),
                                   (1, 5, 2, 10, 12), (1, 2, 2, 50, 64), (1, 2, 1, 21, 48),
                                   # row-streaming kernels: 4 / 6 / 40 float4 columns, planes over several waves and blocks
                                   (3, 64, 2, 16, 16), (2, 96, 2, 32, 24), (1, 3, 2, 64, 160), (3, 130, 1, 16, 12),
                                   (1, 2, 2, 1, 8), (1, 3, 2, 3, 4), (2, 3, 1, 2, 12),
                                   # plane packing (several images' planes of one channel per wave): full and partial image
                                   # groups, the two-column last group of the 18 x 18 bottleneck planes, one DPP row per plane
                                   (3, 20, 2, 18, 18), (5, 7, 1, 18, 18), (8, 16, 2, 18, 18), (7, 9, 2, 12, 8), (6, 4, 2, 8, 20)])
def test_dw3x3_bwd(shape):
    both(case_dw_bwd, *shape, tol=2e-5)
    both(case_dw_bwd, *shape, need_dx=False, tol=2e-5)


# ----------------------------------------------------------------------------------------
# bf16-split matrix path: exact three-term operand split, six bf16 MFMAs per product
def case_pw_split(L, dev, N, C, M, H, W, with_part=False):
    x = T(rnd(1, N, C, H, W) * np.exp(rnd(7, N, C, 1, 1)), dev)   # wide dynamic range across channels
    w, b = T(rnd(2, M, C, scale=0.2), dev), T(rnd(3, M), dev)
    Cp = (C + 15) // 16 * 16
    pl = torch.full((3, M, Cp), -1, dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w), M, C, P(pl), stream(dev)) == 0
    out = torch.full((N, M, H, W), float("nan"), device=dev)
    slots = L.smaat_pw_split_num_slots(N, H, W)
    part = torch.full((3, slots, M), float("nan"), device=dev) if with_part else None
    assert L.smaat_pointwise_fwd_split(P(x), C * H * W, P(pl), P(b), P(out), M * H * W, P(part), N, C, M, H, W,
                                       stream(dev)) == 0
    r = dict(out=out, planes=pl.to(torch.int32))
    if with_part:
        r["pn"], r["pmean"], r["pvar"] = part_stats(part)
    return r


@pytest.mark.parametrize("shape", [(2, 64, 1, 32, 32), (2, 64, 21, 16, 16), (2, 16, 3, 6, 7), (2, 128, 256, 36, 36),
                                   (4, 64, 70, 288, 288), (2, 1, 64, 20, 20), (2, 512, 1024, 18, 18),
                                   (4, 40, 130, 288, 288), (2, 24, 64, 32, 32), (1, 37, 19, 5, 9),
                                   # persistent kernel, several items per workgroup with ONE chunk per item (the bias /
                                   # statistics slots rotate every barrier), full and partial chunk, 1 and 2 channel tiles
                                   (4, 16, 64, 288, 288), (3, 8, 200, 144, 144)])
def test_pointwise_fwd_split(shape):
    both(case_pw_split, *shape)
    both(case_pw_split, *shape, with_part=True)


@pytest.mark.parametrize("shape", [(2, 32, 64, 32, 32), (2, 64, 128, 36, 36), (1, 16, 64, 2, 2), (3, 24, 70, 18, 18)])
def test_bn_partials_keep_the_variance_under_a_large_mean(shape):
    """per-tile (mean, M2, count) partials merged pairwise.  Outputs with |mean| = 1e3..1e4 x std
    (sums of z and z^2 in f32 would lose the variance entirely) must still give the variance of the produced z to
    1e-3 and the mean to 1e-6, in all three GEMM families."""
    N, C, M, H, W = shape
    L, dev = _lib.get(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(N, C, H, W, generator=g) * 1e-2 + 5.0).to(dev)    # mean 5, std 3e-3
    w = (torch.rand(M, C, generator=g) * 0.5 + 0.75).to(dev)          # positive weights: |out mean| ~ 5 C, std ~ 2e-2
    Cp = (C + 15) // 16 * 16
    pl = torch.empty((3, M, Cp), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w), M, C, P(pl), stream(dev)) == 0
    outs = {}
    z = torch.empty(N, M, H, W, device=dev)
    part = torch.full((3, L.smaat_pw_split_num_slots(N, H, W), M), float("nan"), device=dev)
    assert L.smaat_pointwise_fwd_split(P(x), C * H * W, P(pl), None, P(z), M * H * W, P(part), N, C, M, H, W,
                                       stream(dev)) == 0
    outs["split"] = (z.clone(), part)
    z2 = torch.empty(N, M, H, W, device=dev)
    part2 = torch.full((3, L.smaat_pw_num_slots(N, H, W, M), M), float("nan"), device=dev)
    wt = w.t().contiguous()
    assert L.smaat_pointwise_fwd(P(x), C * H * W, P(wt), None, P(z2), M * H * W, P(part2), N, C, M, H, W,
                                 stream(dev)) == 0
    outs["f32"] = (z2, part2)
    torch.cuda.synchronize()
    for name, (zz, pp) in outs.items():
        n, mean, var = part_stats(pp)
        raw = zz.double()  # no bias: the stored z ARE the accumulators the statistics were taken from
        assert torch.equal(n, torch.full_like(n, N * H * W)), name
        ref_mean, ref_var = raw.mean(dim=(0, 2, 3)), raw.var(dim=(0, 2, 3), unbiased=False)
        assert ((mean - ref_mean).abs() / ref_mean.abs()).max().item() < 1e-6, name
        assert (ref_mean.abs() / ref_var.sqrt()).min().item() > 1e3, "the case must be offset-dominated"
        # Tile means are stored as f32: with |mean| = 1e4 x the between-tile differences, their rounding moves the
        # between-tile term (1/128 of the variance here) by ~1e-2 of itself -> 1e-4 of the variance.  One-tile cases
        # of the split family reproduce the variance of the stored values to the last digit.  f32 family on a 2 x 2
        # map (four values per channel): one of the four differs by one ulp OF THE MEAN between the accumulator the
        # statistics see and the stored value (2e-3 of this variance) -- bounded, not explained.
        tol = 5e-3 if (name == "f32" and H * W * N <= 4) else 2e-4
        assert ((var - ref_var).abs() / ref_var).max().item() < tol, (name, ((var - ref_var).abs() / ref_var).max())


def case_dsconv_fwd_split(L, dev, N, Cin, Cout, H, W, aff=False, pad_c=0, bias=True, want_y=True):
    """fused depthwise -> split GEMM forward (smaat_dsconv_fwd_split), kernels_per_layer = 2"""
    K = Cin * 2
    xfull = T(rnd(1, N, Cin + pad_c, H, W) * np.exp(rnd(7, 1, Cin + pad_c, 1, 1)), dev)  # wide range across channels
    x = xfull[:, pad_c:]
    x_bs = (Cin + pad_c) * H * W
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    w, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    Kp = (K + 15) // 16 * 16
    pl = torch.full((3, Cout, Kp), -1, dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w), Cout, K, P(pl), stream(dev)) == 0
    z = torch.full((N, Cout, H, W), float("nan"), device=dev)
    slots = L.smaat_dsconv_split_num_slots(N, H, W)
    assert slots > 0
    part = torch.full((3, slots, Cout), float("nan"), device=dev)
    y = torch.full((N, K, H, W), float("nan"), device=dev) if want_y else None
    rc = L.smaat_dsconv_fwd_split(x.data_ptr(), x_bs, P(sc), P(sh), P(w_dw), P(b_dw) if bias else None, P(pl),
                                  P(b_pw) if bias else None, P(z), Cout * H * W, P(part), P(y), N, Cin, 2, Cout, H, W,
                                  stream(dev))
    assert rc == 0
    pn, pmean, pvar = part_stats(part)
    r = dict(z=z, pn=pn, pmean=pmean, pvar=pvar)
    if want_y:
        r["y"] = y
    return r


@pytest.mark.parametrize("shape", [
    # N, Cin, Cout, H, W
    (2, 12, 64, 32, 32),      # 4 x 32 tiles, partial last chunk (K = 24)
    (2, 8, 10, 16, 16),       # 8 x 16 tiles, partial channel tile
    (1, 64, 64, 64, 64),      # 8 chunks
    (2, 16, 130, 48, 48),     # three channel tiles (the last partial), 8 x 16 tiles
    (2, 5, 7, 20, 32),        # H not a multiple of the tile height, odd channel counts
    (1, 3, 64, 8, 80),        # W = 80: 8 x 16 tiles, one tile row
    (2, 12, 64, 288, 288),    # inc.0
    (1, 128, 64, 288, 288),   # up4.0: 16 chunks
    (2, 64, 128, 144, 144),   # down1.0: W = 144 -> 8 x 16 tiles, two channel tiles
    (1, 40, 64, 32, 32),      # 5 chunks: the 3-deep load groups wrap with a tail
    (1, 2, 3, 4, 32),         # ONE partial chunk, one tile
])
def test_dsconv_fwd_split(shape):
    both(case_dsconv_fwd_split, *shape, tol=2e-6)
    both(case_dsconv_fwd_split, *shape, aff=True, pad_c=4, bias=False, want_y=False, tol=2e-6)


def case_dsconv_fwd_rows(L, dev, N, Cin, Cout, H, W, aff=False, pad_c=0, bias=True, x_bf=False, z_bf=False):
    """row-walking fused forward (smaat_dsconv_fwd_rows, csrc/dsrows.hip): f32 storage (split planes) or bf16 storage"""
    K = Cin * 2
    xf = rnd(1, N, Cin + pad_c, H, W) * np.exp(rnd(7, 1, Cin + pad_c, 1, 1))  # wide range across channels
    xfull = T(xf, dev)
    if x_bf:
        xfull = xfull.to(torch.bfloat16)
    x = xfull[:, pad_c:]
    x_bs = (Cin + pad_c) * H * W
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    w, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    if z_bf:
        pl = torch.full(((K + 31) // 32 * 2, Cout, 16), -1, dtype=torch.int16, device=dev)
        assert L.smaat_bf16_planes(P(w), Cout, K, P(pl), 0, stream(dev)) == 0
    else:
        pl = torch.full((3, Cout, (K + 15) // 16 * 16), -1, dtype=torch.int16, device=dev)
        assert L.smaat_split_planes(P(w), Cout, K, P(pl), stream(dev)) == 0
    z = torch.full((N, Cout, H, W), float("nan"), device=dev).to(torch.bfloat16 if z_bf else torch.float32)
    assert L.smaat_dsconv_rows_ok(2, Cin, Cout, H, W) == 1
    slots = L.smaat_dsconv_rows_num_slots(N, H, W)
    part = torch.full((3, slots, Cout), float("nan"), device=dev)
    rc = L.smaat_dsconv_fwd_rows(x.data_ptr(), 1 if x_bf else 0, x_bs, P(sc), P(sh), P(w_dw), P(b_dw) if bias else None, P(pl),
                                 P(b_pw) if bias else None, P(z), 1 if z_bf else 0, Cout * H * W, P(part), N, Cin, 2, Cout, H, W,
                                 stream(dev))
    assert rc == 0
    pn, pmean, pvar = part_stats(part)
    return dict(z=z.float(), pn=pn, pmean=pmean, pvar=pvar)


ROWS_SHAPES = [
    # N, Cin, Cout, H, W
    (2, 64, 64, 32, 32),      # one strip, one band, K = 128
    (1, 8, 10, 5, 64),        # two strips, partial channel tile, K = 16
    (2, 40, 50, 36, 96),      # three strips, K = 80 (contraction steps beyond K are zero)
    (1, 128, 64, 70, 32),     # two channels per producer thread (K = 256), bands of 32 with a short last one
    (3, 72, 33, 9, 32),       # K = 144: the second channel of most producer threads is beyond Cin
    (2, 64, 64, 288, 288),    # the inc.1 / up4.1 geometry
    (9, 16, 1, 2, 64),        # more items than fit one workgroup each, a single output channel
    (1, 8, 16, 67, 32),       # a prime height above 64: 32-row bands with a short last one (3 rows)
]


@pytest.mark.parametrize("shape", ROWS_SHAPES)
def test_dsconv_fwd_rows(shape):
    both(case_dsconv_fwd_rows, *shape, tol=2e-6)
    both(case_dsconv_fwd_rows, *shape, aff=True, pad_c=4, bias=False, tol=2e-6)


@pytest.mark.parametrize("shape", ROWS_SHAPES[:5])
def test_dsconv_fwd_rows_bf16_storage(shape):
    """mixed precision: bf16 x (or the f32 stem input) -> bf16 z, one bf16 MFMA per product, f32 accumulation and statistics.
    Against the emulation (f32 depthwise, ONE rounding of y, bf16 weights, fp64 sum): an f32 accumulation-order difference can
    flip the final bf16 rounding of z by one ulp (2^-8 relative) on a few elements"""
    both(case_dsconv_fwd_rows, *shape, x_bf=True, z_bf=True, tol=2e-3)
    both(case_dsconv_fwd_rows, *shape, x_bf=False, z_bf=True, aff=True, pad_c=4, tol=2e-3)


def test_dsconv_fwd_rows_refuses_what_it_does_not_handle():
    L, dev = _lib.get(), torch.device("cuda:0")
    assert L.smaat_dsconv_rows_ok(2, 64, 64, 18, 18) == 0 and L.smaat_dsconv_rows_ok(4, 64, 64, 32, 32) == 0
    assert L.smaat_dsconv_rows_ok(2, 64, 128, 32, 32) == 0 and L.smaat_dsconv_rows_ok(2, 12, 64, 32, 32) == 0
    assert L.smaat_dsconv_rows_ok(2, 256, 64, 32, 32) == 0 and L.smaat_dsconv_rows_num_slots(2, 18, 18) == 0
    t = torch.zeros(1, 8, 18, 18, device=dev)
    assert L.smaat_dsconv_fwd_rows(P(t), 0, 8 * 324, None, None, P(t), None, P(t), None, P(t), 0, 8 * 324, None, 1, 8, 2, 8, 18, 18,
                                   stream(dev)) == -2


def test_dsconv_fwd_split_refuses_what_it_does_not_handle():
    L, dev = _lib.get(), torch.device("cuda:0")
    assert L.smaat_dsconv_split_num_slots(2, 18, 18) == 0 and L.smaat_dsconv_split_num_slots(2, 4, 16) == 0
    x = torch.zeros(1, 4, 18, 18, device=dev)
    assert L.smaat_dsconv_fwd_split(P(x), 4 * 324, None, None, P(x), None, P(x), None, P(x), 4 * 324, None, None, 1, 4, 2, 4,
                                    18, 18, stream(dev)) == -2
    x = torch.zeros(1, 4, 32, 32, device=dev)
    assert L.smaat_dsconv_fwd_split(P(x), 4 * 1024, None, None, P(x), None, P(x), None, P(x), 4 * 1024, None, None, 1, 4, 4,
                                    4, 32, 32, stream(dev)) == -2  # kernels_per_layer 4


def case_dw_fwd(L, dev, N, Cin, kpl, H, W, bias=True, pad_c=0, aff=False):
    K = Cin * kpl
    xfull = T(rnd(1, N, Cin + pad_c, H, W), dev)
    x = xfull[:, pad_c:]
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    y = torch.full((N, K, H, W), float("nan"), device=dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    rc = L.smaat_dw3x3_fwd(x.data_ptr(), (Cin + pad_c) * H * W, P(sc), P(sh), P(w_dw), P(b_dw) if bias else None, P(y),
                           K * H * W, N, Cin, kpl, H, W, stream(dev))
    assert rc == 0
    return dict(y=y)


@pytest.mark.parametrize("shape", [(2, 12, 2, 32, 32), (2, 5, 1, 8, 8), (2, 8, 2, 36, 36), (2, 4, 2, 144, 144),
                                   (2, 3, 2, 288, 288), (1, 4, 2, 100, 100), (2, 8, 2, 4, 4), (1, 3, 4, 8, 12),
                                   (2, 4, 2, 72, 72), (1, 5, 2, 10, 12), (1, 2, 2, 50, 64), (1, 2, 1, 21, 48),
                                   (1, 2, 2, 3, 4), (1, 2, 2, 1, 8), (3, 64, 2, 16, 16), (2, 96, 2, 32, 24), (1, 3, 2, 64, 160),
                                   # rows that are not 16-byte aligned: the flat-copy small-plane kernel
                                   (2, 20, 2, 18, 18), (3, 5, 1, 9, 11), (1, 3, 4, 7, 5), (2, 37, 2, 6, 6), (1, 2, 2, 39, 41),
                                   (5, 7, 4, 18, 18), (8, 16, 2, 18, 18), (7, 9, 2, 12, 8)])  # (plane packing)
def test_dw3x3_fwd(shape):
    both(case_dw_fwd, *shape)
    both(case_dw_fwd, *shape, bias=False, pad_c=4)
    both(case_dw_fwd, *shape, aff=True)


@pytest.mark.parametrize("shape", [(2, 5, 1, 8, 8), (2, 8, 2, 36, 36), (2, 4, 2, 144, 144), (1, 3, 2, 288, 288), (1, 3, 2, 4, 4),
                                   (2, 4, 2, 72, 72), (1, 5, 2, 10, 12), (1, 2, 2, 50, 64), (1, 2, 1, 21, 48), (1, 2, 2, 3, 4),
                                   (1, 2, 2, 2, 8), (3, 64, 2, 16, 16), (2, 33, 2, 32, 24), (1, 3, 2, 64, 160)])
@pytest.mark.parametrize("aff", [False, True])
def test_dw3x3_fwd_not_walking_is_the_walker_bit_for_bit(shape, aff, monkeypatch):
    """round 5: k_dw3x3_fwd_lin (a lane owns one output position and loads its three window rows itself; waves in address
    order) against k_dw3x3_fwd_rows (a lane walks down a band with the window in registers): the same window construction and
    tap order, so y and the published maximum must agree bit for bit -- f32 and bf16 storage, x a channel slice of a wider
    buffer, every plane size the launcher would give either kernel (SMAAT_DW_LIN=1 forces the new one wherever it applies)."""
    L, dev = _lib.get(), torch.device("cuda:0")
    N, Cin, kpl, H, W = shape
    K, pad_c = Cin * kpl, 3
    xfull = T(rnd(1, N, Cin + pad_c, H, W), dev)
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    for xdt, ydt in ((0, 0), (1, 1), (0, 1)):
        xs = xfull if xdt == 0 else xfull.to(torch.bfloat16)
        es = 4 if xdt == 0 else 2
        out = []
        for mode in ("0", "1"):
            monkeypatch.setenv("SMAAT_DW_LIN", mode)
            y = torch.full((N, K, H, W), float("nan"), dtype=torch.float32 if ydt == 0 else torch.bfloat16, device=dev)
            rc = L.smaat_dw3x3_fwd_t(xs.data_ptr() + es * pad_c * H * W, xdt, (Cin + pad_c) * H * W, P(sc), P(sh), P(w_dw), P(b_dw),
                                     P(y), ydt, K * H * W, N, Cin, kpl, H, W, stream(dev))
            am = None
            if xdt == 0 and ydt == 0 and rc == 0:
                am = torch.zeros(1024, dtype=torch.int32, device=dev)
                y2 = torch.full((N, K, H, W), float("nan"), device=dev)
                rc2 = L.smaat_dw3x3_fwd_amax(xs.data_ptr() + es * pad_c * H * W, (Cin + pad_c) * H * W, P(sc), P(sh), P(w_dw), P(b_dw),
                                             P(y2), K * H * W, P(am), N, Cin, kpl, H, W, stream(dev))
                if rc2 == 0:
                    torch.cuda.synchronize()
                    assert torch.equal(y, y2)
                    assert int(am.max()) == int(y2.abs().max().view(torch.int32))
                else:
                    assert rc2 == -2
                    am = None
            torch.cuda.synchronize()
            out.append((rc, y, None if am is None else int(am.max())))
        assert out[0][0] == out[1][0]
        if out[0][0] == 0:
            assert torch.equal(out[0][1].view(torch.int16 if ydt else torch.int32), out[1][1].view(torch.int16 if ydt else torch.int32))
            assert not bool(torch.isnan(out[1][1].float()).any())
            assert out[0][2] == out[1][2]
        else:
            assert out[0][0] == -2


def case_dw_bwd_bnred(L, dev, N, Cin, kpl, H, W, gamma_mode="normal"):
    """x = the PRE-BatchNorm tensor z; y = relu(z*sc + sh) is recomputed on load and the kernel also reduces that
    BatchNorm's backward sums with zhat = (z - mean) * invstd (valid for gamma == 0 / tiny gamma)"""
    K = Cin * kpl
    z = rnd(1, N, Cin, H, W) * 1.3 + 0.2
    mean = z.mean(axis=(0, 2, 3)).astype(np.float32)
    invstd = (1.0 / np.sqrt(z.var(axis=(0, 2, 3)) + 1e-5)).astype(np.float32)
    gam = np.random.default_rng(8).uniform(0.5, 1.5, Cin).astype(np.float32)
    if gamma_mode == "zero":
        gam[::2] = 0.0            # a channel whose BatchNorm weight is exactly 0: y = relu(beta), dgamma still defined
    elif gamma_mode == "tiny":
        gam[:] = 1e-6             # |beta| >> |gamma * zhat|: (y - beta) / gamma would cancel
    bet = (np.abs(rnd(9, Cin, scale=0.3)) + 0.05).astype(np.float32)  # positive: the gamma == 0 channels stay active
    sc = (gam * invstd).astype(np.float32)
    sh = (bet - mean * sc).astype(np.float32)
    x, sc, sh, mean, invstd = T(z, dev), T(sc, dev), T(sh, dev), T(mean, dev), T(invstd, dev)
    dy = T(rnd(2, N, K, H, W), dev)
    w_dw = T(rnd(3, K, 9, scale=0.3), dev)
    dx = torch.full((N, Cin, H, W), float("nan"), device=dev)
    rows = L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W)
    ws = torch.empty((rows, K, 10), device=dev)
    rpart = torch.full((2, rows - 1, Cin), float("nan"), device=dev)
    dw, db = torch.full((K, 9), float("nan"), device=dev), torch.full((K,), float("nan"), device=dev)
    assert L.smaat_dw3x3_strip_ok(kpl, H, W) == 1
    rc = L.smaat_dw3x3_bwd_bnred(P(x), Cin * H * W, P(sc), P(sh), P(dy), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws),
                                 P(dw), P(db), P(mean), P(invstd), P(rpart), N, Cin, kpl, H, W, stream(dev))
    assert rc == 0
    # without the activation coefficients the fused form is refused (the caller runs the two kernels separately)
    assert L.smaat_dw3x3_bwd_bnred(P(x), Cin * H * W, None, None, P(dy), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws),
                                   P(dw), P(db), P(mean), P(invstd), P(rpart), N, Cin, kpl, H, W, stream(dev)) == -2
    return dict(dw=dw, db=db, dx=dx, r1=rpart[0].double().sum(0), r2=rpart[1].double().sum(0))


@pytest.mark.parametrize("shape", [(2, 12, 2, 32, 32), (2, 5, 1, 8, 8), (2, 8, 2, 36, 36), (2, 4, 2, 144, 144),
                                   (2, 3, 2, 288, 288), (1, 4, 2, 100, 100), (1, 3, 4, 8, 12), (1, 5, 2, 10, 12),
                                   (3, 64, 2, 16, 16), (2, 96, 2, 32, 24), (1, 3, 2, 64, 160),
                                   (3, 20, 2, 18, 18), (8, 16, 2, 18, 18), (7, 9, 2, 12, 8)])  # (plane packing)
def test_dw3x3_bwd_bnred(shape):
    both(case_dw_bwd_bnred, *shape, tol=2e-5)


@pytest.mark.parametrize("mode", ["zero", "tiny"])
def test_dw3x3_bwd_bnred_degenerate_gamma(mode):
    r = both(case_dw_bwd_bnred, 2, 8, 2, 36, 36, gamma_mode=mode, tol=2e-5)
    assert np.abs(r["hip"]["r2"]).min() > 0  # sum g*zhat (= dgamma) is NOT silently zero on the gamma == 0 channels


def test_dw3x3_strip_ok_matches_the_kernels():
    """the exported predicate is what ops.py uses to leave the first activation unmaterialised: it must agree with
    what the two strip kernels accept"""
    L, dev = _lib.get(), torch.device("cuda:0")
    for (kpl, H, W) in [(2, 32, 32), (1, 8, 8), (4, 8, 12), (2, 18, 18), (2, 6, 6), (3, 8, 8), (2, 3, 4), (2, 288, 288)]:
        ok = L.smaat_dw3x3_strip_ok(kpl, H, W)
        N, Cin = 1, 2
        K = Cin * kpl
        x = torch.randn(N, Cin, H, W, device=dev)
        y = torch.empty(N, K, H, W, device=dev)
        w = torch.randn(K, 9, device=dev)
        sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        if kpl in (1, 2, 4):
            rc_f = L.smaat_dw3x3_fwd(P(x), Cin * H * W, P(sc), P(sh), P(w), None, P(y), K * H * W, N, Cin, kpl, H, W,
                                     stream(dev))
        else:
            rc_f = -2
        rows = L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W)
        ws, rp = torch.empty(rows, K, 10, device=dev), torch.empty(2, rows - 1, Cin, device=dev)
        dx, dw, db = torch.empty_like(x), torch.empty(K, 9, device=dev), torch.empty(K, device=dev)
        rc_b = L.smaat_dw3x3_bwd_bnred(P(x), Cin * H * W, P(sc), P(sh), P(y), K * H * W, P(w), P(dx), Cin * H * W, P(ws),
                                       P(dw), P(db), P(sc), P(sc), P(rp), N, Cin, kpl, H, W, stream(dev)) \
            if 1 <= kpl <= 4 else -2
        torch.cuda.synchronize()
        if ok:
            assert rc_f == 0 and rc_b == 0, (kpl, H, W, rc_f, rc_b)
        else:
            assert rc_b in (-2, -1), (kpl, H, W, rc_b)


def case_bn_eval(L, dev, C):
    rm, rv = T(rnd(1, C, scale=0.3), dev), T(np.random.default_rng(2).uniform(0.2, 2, C).astype(np.float32), dev)
    g, b = T(np.random.default_rng(3).uniform(0.5, 1.5, C).astype(np.float32), dev), T(rnd(4, C, scale=0.2), dev)
    st = torch.full((4, C), float("nan"), device=dev)
    assert L.smaat_bn_eval_coefs(P(rm), P(rv), P(g), P(b), 1e-5, C, P(st), stream(dev)) == 0
    st2 = torch.full((4, C), float("nan"), device=dev)
    assert L.smaat_bn_eval_coefs(P(rm), P(rv), None, None, 1e-5, C, P(st2), stream(dev)) == 0
    return dict(st=st, st_noaffine=st2)


@pytest.mark.parametrize("C", [1, 64, 300])
def test_bn_eval_coefs(C):
    both(case_bn_eval, C, tol=1e-6)


# ----------------------------------------------------------------------------------------
def case_bn(L, dev, N, C, H, W, relu=1, slice_pad=0):
    Pn = H * W
    zfull = T(rnd(1, N, C + slice_pad, H, W) * 1.7 + 0.3, dev)
    z = zfull[:, slice_pad:]
    z_bs = (C + slice_pad) * Pn
    gamma = T(np.random.default_rng(2).uniform(0.5, 1.5, C).astype(np.float32), dev)
    beta = T(rnd(3, C, scale=0.2), dev)
    bias = T(rnd(4, C, scale=0.5), dev)
    # per-tile partials (mean, M2, count) of (z - bias): three unequal "tiles" in slots 1, 3, 4 of 5 (0 and 2 empty)
    zz = (z - bias[None, :, None, None]).double().flatten(2)  # [N][C][P]
    part = torch.zeros((3, 5, C), device=dev)
    cuts = [0, Pn // 4, Pn // 4 + max(Pn // 3, 1), Pn]
    for slot, (a, b) in zip((1, 3, 4), zip(cuts[:-1], cuts[1:])):
        piece = zz[:, :, a:b]
        if piece.shape[2] == 0:
            continue
        m = piece.mean(dim=(0, 2))
        part[0, slot] = m.float()
        part[1, slot] = ((piece - part[0, slot].double()[None, :, None]) ** 2).sum(dim=(0, 2)).float()
        part[2, slot] = float(N * (b - a))
    rm, rv = T(rnd(5, C, scale=0.1), dev), T(np.random.default_rng(6).uniform(0.5, 2, C).astype(np.float32), dev)
    st = torch.empty((4, C), device=dev)
    s = stream(dev)
    assert L.smaat_bn_finalize(P(part), 5, C, float(N * Pn), P(bias), P(gamma), P(beta), 1e-5, 0.1, P(rm), P(rv),
                               P(st[0]), P(st[1]), P(st[2]), P(st[3]), s) == 0
    y = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_affine_act(z.data_ptr(), z_bs, P(st[2]), P(st[3]), P(y), C * Pn, N, C, Pn, relu, s) == 0
    dy = T(rnd(7, N, C, H, W), dev)
    slots = L.smaat_plane_num_slots(N, Pn)
    bpart = torch.empty((2, slots, C), device=dev)
    assert L.smaat_bn_bwd_reduce(P(dy), C * Pn, z.data_ptr(), z_bs, P(st[2]), P(st[3]), P(st[0]), P(st[1]), P(bpart),
                                 N, C, Pn, relu, s) == 0
    dgamma, dbeta, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty((3, C), device=dev)
    assert L.smaat_bn_bwd_finalize(P(bpart), slots, C, float(N * Pn), P(gamma), P(st[1]), P(dgamma), P(dbeta), P(coef),
                                   s) == 0
    dz = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_bn_bwd_apply(P(dy), C * Pn, z.data_ptr(), z_bs, P(st[2]), P(st[3]), P(st[0]), P(st[1]), P(coef),
                                P(dz), C * Pn, N, C, Pn, relu, s) == 0
    return dict(st=st, rm=rm, rv=rv, y=y, dgamma=dgamma, dbeta=dbeta, coef=coef, dz=dz)


@pytest.mark.parametrize("shape", [(2, 8, 12, 10), (3, 5, 9, 7), (2, 64, 72, 72), (2, 4, 288, 288), (4, 1, 36, 36)])
def test_bn_chain(shape):
    both(case_bn, *shape, tol=2e-5)
    both(case_bn, *shape, relu=0, slice_pad=3, tol=2e-5)


def case_misc(L, dev, N, C, H, W):
    Pn = H * W
    x = T(rnd(1, N, C, H, W), dev)
    s = stream(dev)
    ws = torch.empty((L.smaat_plane_num_slots(N, Pn), C), device=dev)
    cs = torch.empty(C, device=dev)
    assert L.smaat_channel_sum(P(x), C * Pn, N, C, Pn, P(ws), P(cs), s) == 0
    rows = T(rnd(2, 7, 1000), dev)
    rr = torch.empty(1000, device=dev)
    assert L.smaat_reduce_rows(P(rows), 7, 1000, P(rr), 0.5, s) == 0
    rows2 = T(rnd(4, 1037, 3001), dev)  # tall: exercises the two-level path (part is scratch)
    rr2 = torch.empty(3001, device=dev)
    assert L.smaat_reduce_rows(P(rows2), 1037, 3001, P(rr2), 1.0, s) == 0
    big = torch.zeros((N, C + 3, H, W), device=dev)
    assert L.smaat_copy_planes(P(x), C * Pn, big.data_ptr() + 4 * 2 * Pn, (C + 3) * Pn, N, C * Pn, 0, s) == 0
    assert L.smaat_copy_planes(P(x), C * Pn, big.data_ptr() + 4 * 2 * Pn, (C + 3) * Pn, N, C * Pn, 1, s) == 0
    return dict(cs=cs, rr=rr, rr2=rr2, big=big)


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (2, 16, 64, 64), (1, 1, 288, 288)])
def test_misc(shape):
    both(case_misc, *shape)


def case_pool_up(L, dev, N, C, H, W, Ho, Wo):
    s = stream(dev)
    x = T(np.maximum(rnd(1, N, C, H, W), 0), dev)  # relu-like: exercises ties at 0
    y = torch.full((N, C, H // 2, W // 2), float("nan"), device=dev)
    assert L.smaat_maxpool2_fwd(P(x), C * H * W, P(y), C * (H // 2) * (W // 2), N, C, H, W, s) == 0
    dy = T(rnd(2, N, C, H // 2, W // 2), dev)
    dx = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_maxpool2_bwd(P(x), C * H * W, P(dy), C * (H // 2) * (W // 2), P(dx), C * H * W, N, C, H, W, 0,
                                s) == 0
    # upsample into a padded slice of a cat buffer
    pt, pl = (Ho - 2 * H) // 2, (Wo - 2 * W) // 2
    cat = torch.full((N, C + 2, Ho, Wo), float("nan"), device=dev)
    assert L.smaat_upsample2x_fwd(P(x), C * H * W, cat.data_ptr() + 4 * 2 * Ho * Wo, (C + 2) * Ho * Wo, N, C, H, W, Ho,
                                  Wo, pt, pl, s) == 0
    dcat = T(rnd(3, N, C + 2, Ho, Wo), dev)
    dxu = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_upsample2x_bwd(dcat.data_ptr() + 4 * 2 * Ho * Wo, (C + 2) * Ho * Wo, P(dxu), C * H * W, N, C, H, W,
                                  Ho, Wo, pt, pl, s) == 0
    return dict(y=y, dx=dx, up=cat[:, 2:], dxu=dxu)


@pytest.mark.parametrize("shape", [(2, 3, 8, 8, 16, 16), (2, 4, 5, 6, 11, 13), (1, 2, 11, 13, 22, 26),
                                   (2, 8, 18, 18, 36, 36), (1, 2, 144, 144, 288, 288), (2, 3, 2, 2, 4, 4),
                                   (1, 2, 2, 3, 5, 7),
                                   # row-walking upsample kernels: padded (pad_l % 4 == 0), unpadded, many bands
                                   (2, 4, 6, 8, 16, 24), (2, 5, 10, 12, 20, 24), (1, 3, 72, 72, 144, 144),
                                   (1, 2, 7, 10, 16, 28), (3, 70, 4, 4, 8, 8), (1, 2, 36, 36, 72, 72)])
def test_pool_upsample(shape):
    both(case_pool_up, *shape)


@pytest.mark.parametrize("shape", [(64, 256), (37, 19), (512, 1024), (1, 5)])
def test_split_planes_transposed(shape):
    """smaat_split_planes_t(w [C][R]) == smaat_split_planes(w^T): the data gradient takes the planes of the transposed
    pointwise weight without a transposed copy"""
    L, dev = _lib.get(), torch.device("cuda:0")
    C, R = shape
    w = T(rnd(1, C, R), dev)
    Cp = (C + 15) // 16 * 16
    a = torch.full((3, R, Cp), -1, dtype=torch.int16, device=dev)
    b = torch.full((3, R, Cp), -2, dtype=torch.int16, device=dev)
    wt = w.t().contiguous()
    assert L.smaat_split_planes(P(wt), R, C, P(a), stream(dev)) == 0
    assert L.smaat_split_planes_t(P(w), R, C, P(b), stream(dev)) == 0
    assert torch.equal(a, b)


def test_weight_planes_multi_is_the_single_matrix_kernels():
    """smaat_weight_planes_multi: the images of many weight matrices (split planes of a matrix / of a transpose, bf16 images
    in both orientations, ragged sizes) in ONE launch, bit-identical to the single-matrix entry points"""
    L, dev = _lib.get(), torch.device("cuda:0")
    BF = 1
    mats = [(64, 128, 0, 0), (128, 64, 0, 1), (37, 19, 0, 0), (19, 37, 0, 1), (1, 5, 0, 0), (512, 1024, 0, 1), (64, 24, 0, 0),
            (64, 128, 2, 0), (128, 64, 2, 1), (37, 19, 2, 0), (21, 64, 2, 1), (1024, 512, 2, 0)]  # (R, C, kind, src_t)
    rows, keep, b0 = [], [], 0
    for i, (R, C, kind, src_t) in enumerate(mats):
        w = T(rnd(10 + i, C, R) if src_t else rnd(10 + i, R, C), dev)  # src_t: stored [C][R]
        Cp = (C + 31) // 32 * 32 if kind == 2 else (C + 15) // 16 * 16
        n16 = R * Cp * (1 if kind == 2 else 3)
        ref = torch.full((n16,), -1, dtype=torch.int16, device=dev)
        got = torch.full((n16,), -2, dtype=torch.int16, device=dev)
        if kind == 2:
            assert L.smaat_bf16_planes(P(w), R, C, P(ref), src_t, stream(dev)) == 0
        else:
            assert (L.smaat_split_planes_t if src_t else L.smaat_split_planes)(P(w), R, C, P(ref), stream(dev)) == 0
        nb = (R * Cp + 255) // 256
        rows.append([w.data_ptr(), got.data_ptr(), R, C, kind, src_t, b0, nb])
        b0 += nb
        keep.append((w, ref, got))
    desc = torch.tensor(rows, dtype=torch.int64).to(dev)
    assert L.smaat_weight_planes_multi(P(desc), len(rows), b0, stream(dev)) == 0
    torch.cuda.synchronize()
    for (R, C, kind, src_t), (_, ref, got) in zip(mats, keep):
        assert torch.equal(ref, got), (R, C, kind, src_t)
    assert L.smaat_weight_planes_multi(None, 1, 1, stream(dev)) == -1
    del BF


@pytest.mark.parametrize("shape", [(1, 512, 256, 36, 36), (1, 1024, 512, 18, 18), (2, 256, 256, 36, 36),
                                   (1, 2048, 512, 36, 36), (1, 64, 64, 32, 32), (1, 256, 128, 72, 72)])
def test_pointwise_split_k_slices(shape):
    """inference GEMM with the contraction cut into K slices == the un-split kernel (same products, another order of
    the f32 sums) on the shapes where the library splits, identical where it does not"""
    L, dev = _lib.get(), torch.device("cuda:0")
    N, C, M, H, W = shape
    x = T(np.maximum(rnd(1, N, C, H, W), 0), dev)
    w, b = T(rnd(2, M, C, scale=0.1), dev), T(rnd(3, M), dev)
    pl = torch.empty((3, M, (C + 15) // 16 * 16), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w), M, C, P(pl), stream(dev)) == 0
    ref = torch.full((N, M, H, W), float("nan"), device=dev)
    assert L.smaat_pointwise_fwd_split_act(P(x), C * H * W, P(pl), P(b), P(ref), M * H * W, N, C, M, H, W, 1,
                                           stream(dev)) == 0
    nws = L.smaat_pointwise_splitk_ws_floats(N, C, M, H, W)
    ws = torch.full((max(nws, 1),), float("nan"), device=dev)
    out = torch.full((N, M, H, W), float("nan"), device=dev)
    assert L.smaat_pointwise_fwd_split_act_k(P(x), C * H * W, P(pl), P(b), P(out), M * H * W, P(ws) if nws else None, N, C,
                                             M, H, W, 1, stream(dev)) == 0
    if nws == 0:
        assert torch.equal(out, ref)
    else:
        assert rel(out.cpu().numpy(), ref.cpu().numpy()) < 1e-6
    if shape[1] >= 512:
        assert nws > 0  # the deep batch-1 layers are the ones this exists for


def case_chpool_act(L, dev, N, C, H, W, pad_c=0):
    """channel pooling with the BatchNorm + ReLU of the block in front applied on load and the activated tensor written
    out: y bit-identical to smaat_affine_act, pools identical to pooling that y"""
    Pn = H * W
    zf = T(rnd(1, N, C + pad_c, H, W) * 1.2 - 0.1, dev)
    z_bs = (C + pad_c) * Pn
    sc, sh = T(np.random.default_rng(2).uniform(0.5, 1.5, C).astype(np.float32), dev), T(rnd(3, C, scale=0.3), dev)
    s = stream(dev)
    y = torch.full((N, C, H, W), float("nan"), device=dev)
    avg, mx = torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
    amax = torch.empty(N, C, dtype=torch.int32, device=dev)
    assert L.smaat_cbam_chpool_act(P(zf), z_bs, P(sc), P(sh), P(y), C * Pn, N, C, Pn, P(avg), P(mx), P(amax), s) == 0
    y2 = torch.empty_like(y)
    assert L.smaat_affine_act(P(zf), z_bs, P(sc), P(sh), P(y2), C * Pn, N, C, Pn, 1, s) == 0
    a2, m2 = torch.empty_like(avg), torch.empty_like(mx)
    am2 = torch.empty_like(amax)
    assert L.smaat_cbam_chpool(P(y2), C * Pn, N, C, Pn, P(a2), P(m2), P(am2), s) == 0
    assert torch.equal(y, y2) and torch.equal(avg, a2) and torch.equal(mx, m2) and torch.equal(amax, am2)
    return dict(y=y, avg=avg, mx=mx, amax=amax.float())


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 3, 5, 7), (2, 64, 144, 144), (3, 5, 9, 12)])
def test_cbam_chpool_act(shape):
    both(case_chpool_act, *shape)
    both(case_chpool_act, *shape, pad_c=2)


def case_head(L, dev, N, C, H, W, pad_c=0):
    """OutConv with one output channel fused with the BatchNorm + ReLU in front of it: forward, backward reduction
    (+ the conv's weight gradient), backward apply"""
    Pn = H * W
    zf = T(rnd(1, N, C + pad_c, H, W) * 1.5 + 0.3, dev)
    z_bs = (C + pad_c) * Pn
    mean = zf[:, :C].mean(dim=(0, 2, 3)).contiguous()
    invstd = (1.0 / torch.sqrt(zf[:, :C].var(dim=(0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    gam = T(np.random.default_rng(2).uniform(0.5, 1.5, C).astype(np.float32), dev)
    bet = T(rnd(3, C, scale=0.3), dev)
    sc = (gam * invstd).contiguous()
    sh = (bet - mean * sc).contiguous()
    w, b = T(rnd(4, C, scale=0.4), dev), T(rnd(5, 1), dev)
    s = stream(dev)
    out = torch.full((N, 1, H, W), float("nan"), device=dev)
    assert L.smaat_outconv1_fwd(P(zf), z_bs, P(sc), P(sh), P(w), P(b), P(out), Pn, N, C, Pn, s) == 0
    dlog = T(rnd(6, N, 1, H, W), dev)
    slots = L.smaat_plane_num_slots(N, Pn)
    part = torch.full((3, slots, C), float("nan"), device=dev)
    assert L.smaat_bn_bwd_reduce_head(P(dlog), Pn, P(w), P(zf), z_bs, P(sc), P(sh), P(mean), P(invstd), P(part), N, C, Pn,
                                      s) == 0
    dgamma, dbeta, coef = (torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(3, C, device=dev))
    assert L.smaat_bn_bwd_finalize(P(part), slots, C, float(N * Pn), P(gam), P(invstd), P(dgamma), P(dbeta), P(coef), s) == 0
    dz = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_bn_bwd_apply_head(P(dlog), Pn, P(w), P(zf), z_bs, P(sc), P(sh), P(mean), P(invstd), P(coef), P(dz),
                                     C * Pn, N, C, Pn, s) == 0
    return dict(out=out, dgamma=dgamma, dbeta=dbeta, dw=part[2].double().sum(0), dz=dz)


@pytest.mark.parametrize("shape", [(2, 64, 32, 32), (1, 5, 7, 9), (2, 64, 288, 288), (3, 16, 20, 12)])
def test_outconv1_head(shape):
    both(case_head, *shape, tol=2e-5)
    both(case_head, *shape, pad_c=3, tol=2e-5)


def case_final_pool(L, dev, N, C, H, W, pad_c=0):
    """cbam_bwd_final + maxpool2 backward in one pass == the two separate kernels (bit for bit: same adds, same order)"""
    Pn = H * W
    x = T(np.maximum(rnd(1, N, C + pad_c, H, W), 0), dev)  # relu-like: ties at 0 inside windows
    dpool = T(rnd(2, N, C, H // 2, W // 2), dev)
    davg, dmx = T(rnd(3, N, C), dev), T(rnd(4, N, C), dev)
    amax = torch.from_numpy(np.random.default_rng(5).integers(0, Pn, (N, C)).astype(np.int32)).to(dev)
    dx0 = T(rnd(6, N, C, H, W), dev)
    s = stream(dev)
    dx = dx0.clone()
    rc = L.smaat_cbam_bwd_final_pool(P(dx), C * Pn, P(davg), P(dmx), P(amax), P(x), (C + pad_c) * Pn, P(dpool),
                                     C * (H // 2) * (W // 2), N, C, H, W, s)
    ref = dx0.clone()
    assert L.smaat_cbam_bwd_final(P(ref), C * Pn, P(davg), P(dmx), P(amax), N, C, Pn, s) == 0
    assert L.smaat_maxpool2_bwd(P(x), (C + pad_c) * Pn, P(dpool), C * (H // 2) * (W // 2), P(ref), C * Pn, N, C, H, W, 1,
                                s) == 0
    if W % 4 == 0 and H >= 2:
        assert rc == 0
        assert torch.equal(dx, ref)
    else:
        assert rc == -2
        dx = ref
    return dict(dx=dx)


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (2, 5, 9, 12), (1, 2, 288, 288), (3, 70, 4, 4), (2, 4, 6, 10),
                                   (2, 64, 36, 36), (1, 3, 2, 4)])
def test_cbam_final_pool(shape):
    both(case_final_pool, *shape)
    both(case_final_pool, *shape, pad_c=2)


# ----------------------------------------------------------------------------------------
def case_cbam_eval(L, dev, N, C, H, W, ks=7, rr=16, pool=True, pad_c=0):
    """inference CBAM: chpool -> eval_pool (MLP + channel-wise maps) -> eval_apply (conv + BN(1) running stats + sigmoid
    + product, + maxpool2 of the input), output written into a channel slice of a larger buffer"""
    Pn, Cr = H * W, max(C // rr, 1)
    x = T(np.maximum(rnd(1, N, C, H, W), 0) + 0.01 * rnd(11, N, C, H, W), dev)
    w1, b1 = T(rnd(2, Cr, C, scale=0.3), dev), T(rnd(3, Cr, scale=0.1), dev)
    w2, b2 = T(rnd(4, C, Cr, scale=0.3), dev), T(rnd(5, C, scale=0.1), dev)
    wc = T(rnd(6, 2, ks, ks, scale=0.2), dev)
    g, b = T(np.array([1.3], np.float32), dev), T(np.array([-0.2], np.float32), dev)
    rm, rv = T(np.array([0.15], np.float32), dev), T(np.array([0.7], np.float32), dev)
    s = stream(dev)
    avg, mx = torch.empty((N, C), device=dev), torch.empty((N, C), device=dev)
    amax = torch.empty((N, C), dtype=torch.int32, device=dev)
    assert L.smaat_cbam_chpool(P(x), C * Pn, N, C, Pn, P(avg), P(mx), P(amax), s) == 0
    sc, maps = torch.full((N, C), float("nan"), device=dev), torch.full((N, 2, H, W), float("nan"), device=dev)
    assert L.smaat_cbam_eval_pool(P(x), C * Pn, P(avg), P(mx), P(w1), P(b1), P(w2), P(b2), N, C, Cr, Pn, P(sc), P(maps),
                                  s) == 0
    cat = torch.full((N, C + pad_c, H, W), float("nan"), device=dev)
    pooled = torch.full((N, C, H // 2, W // 2), float("nan"), device=dev) if pool else None
    assert L.smaat_cbam_eval_apply(P(x), C * Pn, P(sc), P(maps), P(wc), ks, P(g), P(b), P(rm), P(rv), 1e-5, N, C, H, W,
                                   P(cat), (C + pad_c) * Pn, P(pooled), C * (H // 2) * (W // 2) if pool else 0, s) == 0
    r = dict(s=sc, maps=maps, out=cat[:, :C])
    if pool:
        r["pooled"] = pooled
    return r


@pytest.mark.parametrize("shape", [(2, 32, 10, 10), (1, 64, 288, 288), (2, 512, 18, 18), (1, 128, 36, 36), (3, 10, 9, 12),
                                   (2, 16, 7, 33), (1, 256, 72, 72)])
def test_cbam_eval(shape):
    both(case_cbam_eval, *shape, tol=2e-6)
    both(case_cbam_eval, *shape, ks=3, rr=8, pool=False, pad_c=3, tol=2e-6)


def case_pixel_shuffle(L, dev, N, Co, H, W, Ho, Wo, slice_pad=0):
    """ConvTranspose2d(k=2, s=2) tail: 2x2 pixel shuffle + bias into a padded slice of a cat buffer, and its inverse"""
    t = T(rnd(1, N, 4 * Co, H, W), dev)
    bias = T(rnd(2, Co), dev)
    cat = torch.full((N, Co + slice_pad, Ho, Wo), float("nan"), device=dev)
    pt, pl = (Ho - 2 * H) // 2, (Wo - 2 * W) // 2
    s = stream(dev)
    assert L.smaat_pixel_shuffle2_fwd(P(t), 4 * Co * H * W, P(bias), cat.data_ptr() + 4 * slice_pad * Ho * Wo,
                                      (Co + slice_pad) * Ho * Wo, N, Co, H, W, Ho, Wo, pt, pl, s) == 0
    dcat = T(rnd(3, N, Co + slice_pad, Ho, Wo), dev)
    dt = torch.full((N, 4 * Co, H, W), float("nan"), device=dev)
    assert L.smaat_pixel_shuffle2_bwd(dcat.data_ptr() + 4 * slice_pad * Ho * Wo, (Co + slice_pad) * Ho * Wo, P(dt),
                                      4 * Co * H * W, N, Co, H, W, Ho, Wo, pt, pl, s) == 0
    return dict(out=cat[:, slice_pad:], dt=dt)


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 10, 12), (2, 4, 5, 6, 11, 13), (1, 3, 1, 1, 2, 2), (2, 64, 36, 36, 72, 72)])
def test_pixel_shuffle(shape):
    both(case_pixel_shuffle, *shape, tol=0)
    both(case_pixel_shuffle, *shape, slice_pad=5, tol=0)


def case_cbam(L, dev, N, C, H, W, ks=7, rr=16):
    Pn, Cr = H * W, max(C // rr, 1)
    s = stream(dev)
    x = T(np.maximum(rnd(1, N, C, H, W), 0), dev)
    w1, b1 = T(rnd(2, Cr, C, scale=0.3), dev), T(rnd(3, Cr, scale=0.1), dev)
    w2, b2 = T(rnd(4, C, Cr, scale=0.3), dev), T(rnd(5, C, scale=0.1), dev)
    wc = T(rnd(6, 2, ks, ks, scale=0.2), dev)
    gamma, beta = T(np.array([1.3], np.float32), dev), T(np.array([0.1], np.float32), dev)
    avg, mx = torch.empty((N, C), device=dev), torch.empty((N, C), device=dev)
    amax = torch.empty((N, C), dtype=torch.int32, device=dev)
    assert L.smaat_cbam_chpool(P(x), C * Pn, N, C, Pn, P(avg), P(mx), P(amax), s) == 0
    ha, hm, sc = torch.empty((N, Cr), device=dev), torch.empty((N, Cr), device=dev), torch.empty((N, C), device=dev)
    assert L.smaat_cbam_mlp(P(avg), P(mx), P(w1), P(b1), P(w2), P(b2), N, C, Cr, P(ha), P(hm), P(sc), s) == 0
    maps = torch.empty((N, 2, H, W), device=dev)
    assert L.smaat_cbam_sppool(P(x), C * Pn, P(sc), N, C, Pn, P(maps), s) == 0
    nb = L.smaat_cbam_spconv_blocks(N, H, W)
    conv, part = torch.empty((N, 1, H, W), device=dev), torch.empty((3, nb, 1), device=dev)
    assert L.smaat_cbam_spconv(P(maps), P(wc), ks, N, H, W, P(conv), P(part), s) == 0
    st = torch.empty((4, 1), device=dev)
    assert L.smaat_bn_finalize(P(part), nb, 1, float(N * Pn), None, P(gamma), P(beta), 1e-5, 0.1, None, None,
                               P(st[0]), P(st[1]), P(st[2]), P(st[3]), s) == 0
    gate = torch.empty((N, 1, H, W), device=dev)
    assert L.smaat_cbam_gate(P(conv), P(st[2]), P(st[3]), N * Pn, P(gate), s) == 0
    out = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_cbam_apply(P(x), C * Pn, P(sc), P(gate), P(out), C * Pn, N, C, Pn, s) == 0
    # ---- backward
    dout = T(rnd(7, N, C, H, W), dev)
    nbp = L.smaat_cbam_pix_blocks(N, Pn)
    dbn, bpart = torch.empty((N, Pn), device=dev), torch.empty((2, nbp, 1), device=dev)
    assert L.smaat_cbam_bwd_gate(P(dout), C * Pn, P(x), C * Pn, P(sc), P(gate), P(conv), P(st[0]), P(st[1]), N, C, Pn,
                                 P(dbn), P(bpart), s) == 0
    dgamma, dbeta, coef = torch.empty(1, device=dev), torch.empty(1, device=dev), torch.empty((3, 1), device=dev)
    assert L.smaat_bn_bwd_finalize(P(bpart), nbp, 1, float(N * Pn), P(gamma), P(st[1]), P(dgamma), P(dbeta), P(coef),
                                   s) == 0
    dmaps, wpart = torch.empty((N, 2, H, W), device=dev), torch.empty((nb, 2 * ks * ks), device=dev)
    assert L.smaat_cbam_bwd_spconv(P(dbn), P(conv), P(st[0]), P(st[1]), P(coef), P(maps), P(wc), ks, N, H, W, P(dmaps),
                                   P(wpart), s) == 0
    dx = torch.full((N, C, H, W), float("nan"), device=dev)
    dspart = torch.empty((nbp, C), device=dev)
    assert L.smaat_cbam_bwd_main(P(dout), C * Pn, P(x), C * Pn, P(sc), P(gate), P(maps), P(dmaps), N, C, Pn, P(dx),
                                 C * Pn, P(dspart), s) == 0
    per = nbp // N
    ds = dspart.view(per, N, C).double().sum(0).float().contiguous()
    pgs = C * Cr + C + Cr * C + Cr
    pg, davg, dmx = torch.empty((N, pgs), device=dev), torch.empty((N, C), device=dev), torch.empty((N, C), device=dev)
    assert L.smaat_cbam_bwd_mlp(P(ds), P(sc), P(avg), P(mx), P(ha), P(hm), P(w1), P(w2), N, C, Cr, P(pg), P(davg),
                                P(dmx), s) == 0
    dx_main = dx.clone()
    assert L.smaat_cbam_bwd_final(P(dx), C * Pn, P(davg), P(dmx), P(amax), N, C, Pn, s) == 0
    return dict(avg=avg, mx=mx, amax=amax.float(), ha=ha, hm=hm, sc=sc, maps=maps, conv=conv,
                pmean=part_stats(part)[1], pvar=part_stats(part)[2], st=st, gate=gate, out=out, dbn=dbn,
                bsum=bpart.double().sum(1), dgamma=dgamma, dbeta=dbeta, coef=coef, dmaps=dmaps,
                dwc=wpart.double().sum(0), dx_main=dx_main, ds=ds, pg=pg, davg=davg, dmx=dmx, dx=dx)


@pytest.mark.parametrize("shape", [(2, 32, 10, 10), (2, 64, 4, 4), (3, 16, 9, 12), (2, 64, 72, 72), (2, 512, 18, 18),
                                   (1, 64, 288, 288), (2, 128, 37, 41)])
def test_cbam_chain(shape):
    both(case_cbam, *shape, tol=3e-5)


def test_cbam_ks3():
    both(case_cbam, 2, 32, 10, 10, ks=3, tol=3e-5)


def case_cbam_three_pass(L, dev, N, C, H, W, pad_c=0, pool=True):
    """the three-pass backward of a level's attention (gate + ds1 | ds2 | apply [+ pool]) against the gate / main /
    final[_pool] sequence: with one wave per 256 pixels dbn and its BatchNorm partials bit for bit (channels split over
    waves: up to the f32 summation order), ds up to the f32 summation order, and -- fed the SAME davg / dmx -- dx bit for
    bit; the forward kernel's maps against smaat_cbam_sppool and its index map against the scan rule of
    k_cbam_bwd_main.  x / dout are channel slices of wider buffers when pad_c > 0."""
    Pn = H * W
    s = stream(dev)
    xw = T(np.maximum(rnd(1, N, C + pad_c, H, W), 0), dev)
    dw = T(rnd(7, N, C + pad_c, H, W), dev)
    x, dout = xw[:, pad_c:], dw[:, :C]
    bs = (C + pad_c) * Pn
    XP, DP = xw.data_ptr() + 4 * pad_c * Pn, dw.data_ptr()
    sc = T(np.random.default_rng(2).uniform(0.2, 0.9, (N, C)).astype(np.float32), dev)
    maps0 = torch.empty((N, 2, H, W), device=dev)
    assert L.smaat_cbam_sppool(XP, bs, P(sc), N, C, Pn, P(maps0), s) == 0
    maps = torch.full((N, 2, H, W), float("nan"), device=dev)
    amaxc = torch.full((N, H, W), -1, dtype=torch.int32, device=dev)
    assert L.smaat_cbam_sppool_idx_t(XP, bs, P(sc), N, C, Pn, P(maps), P(amaxc), 0, s) == 0
    gate = T(np.random.default_rng(3).uniform(0.1, 0.9, (N, 1, H, W)).astype(np.float32), dev)
    conv = T(rnd(5, N, 1, H, W), dev)
    mean, invstd = T(rnd(6, 1, scale=0.1), dev), T(np.array([0.8], np.float32), dev)
    dmaps = T(rnd(8, N, 2, H, W, scale=0.1), dev)
    davg, dmx = T(rnd(9, N, C, scale=0.1), dev), T(rnd(10, N, C, scale=0.1), dev)
    amax = T(np.random.default_rng(11).integers(0, Pn, (N, C)).astype(np.int32), dev)
    Ho, Wo = H // 2, W // 2
    dpool = T(rnd(12, N, C, Ho, Wo), dev) if pool else None
    nbp = L.smaat_cbam_pix_blocks(N, Pn)
    per = nbp // N
    # ---- the sequence it replaces
    dbn0, pa0 = torch.empty((N, Pn), device=dev), torch.empty((2, nbp, 1), device=dev)
    assert L.smaat_cbam_bwd_gate(DP, bs, XP, bs, P(sc), P(gate), P(conv), P(mean), P(invstd), N, C, Pn, P(dbn0), P(pa0), s) == 0
    dx0 = torch.full((N, C, H, W), float("nan"), device=dev)
    dsp0 = torch.empty((nbp, C), device=dev)
    assert L.smaat_cbam_bwd_main(DP, bs, XP, bs, P(sc), P(gate), P(maps0), P(dmaps), N, C, Pn, P(dx0), C * Pn, P(dsp0), s) == 0
    if pool:
        assert L.smaat_cbam_bwd_final_pool(P(dx0), C * Pn, P(davg), P(dmx), P(amax), XP, bs, P(dpool), C * Ho * Wo, N, C, H, W,
                                           s) == 0
    else:
        assert L.smaat_cbam_bwd_final(P(dx0), C * Pn, P(davg), P(dmx), P(amax), N, C, Pn, s) == 0
    # ---- three passes
    assert L.smaat_cbam_bwd3_ok(XP, bs, DP, bs, P(dpool), C * Ho * Wo if pool else 0, N, C, H, W, 0) == 1
    dbn1, pa1 = torch.empty((N, Pn), device=dev), torch.empty((2, nbp, 1), device=dev)
    dsp = torch.full((2 * per, N, C), float("nan"), device=dev)
    assert L.smaat_cbam_bwd_gate_ds_t(DP, bs, XP, bs, P(sc), P(gate), P(conv), P(mean), P(invstd), N, C, Pn, P(dbn1), P(pa1),
                                      P(dsp), 0, s) == 0
    assert L.smaat_cbam_bwd_ds2_t(XP, bs, P(dmaps), P(amaxc), N, C, Pn, dsp.data_ptr() + 4 * per * N * C, 0, s) == 0
    dx1 = torch.full((N, C, H, W), float("nan"), device=dev)
    assert L.smaat_cbam_bwd_apply_t(DP, bs, XP, bs, P(sc), P(gate), P(dmaps), P(amaxc), P(davg), P(dmx), P(amax), P(dpool),
                                    C * Ho * Wo if pool else 0, N, C, H, W, P(dx1), C * Pn, 0, s) == 0
    if dev.type == "cuda":
        torch.cuda.synchronize()
    # the index map: the first channel whose x * s equals the maximum
    xs = x * sc[:, :, None, None]
    eq = xs == maps0[:, 1:2]
    first = (eq & (eq.cumsum(1) == 1)).float().argmax(1)
    assert torch.equal(amaxc.long(), first)
    assert torch.equal(maps[:, 1], maps0[:, 1])
    assert float((maps[:, 0] - maps0[:, 0]).abs().max()) <= 1e-6 * float(xs.abs().max())
    assert torch.equal(dx0, dx1), float((dx0 - dx1).abs().max())
    mag1 = (dout.abs() * x.abs() * sc[:, :, None, None]).sum(1).reshape(N, Pn)  # magnitudes of the terms of dgate[n][p]
    assert bool(((dbn0 - dbn1).abs() <= 1e-6 * mag1 + 1e-30).all())
    if os.environ.get("SMAAT_CBAM_CS", "") == "1":
        assert torch.equal(dbn0, dbn1) and torch.equal(pa0, pa1) and torch.equal(maps, maps0)
    ds0 = dsp0.view(per, N, C).double().sum(0)
    ds1 = dsp.double().sum(0)
    mag = ((dout.double().abs() * gate.double() + dmaps[:, 0:1].double().abs() / C + dmaps[:, 1:2].double().abs()) *
           x.double().abs()).sum((2, 3))  # sum of the magnitudes of the terms of ds[n][c]
    assert bool(((ds0 - ds1).abs() <= 2e-6 * mag + 1e-30).all()), float(((ds0 - ds1).abs() / (mag + 1e-30)).max())
    return dict(dbn=dbn1, bsum=pa1.double().sum(1), ds=ds1.float(), dx=dx1, maps=maps, amaxc=amaxc.float())


@pytest.mark.parametrize("shape", [(2, 32, 10, 12), (2, 64, 4, 4), (3, 5, 8, 12), (2, 256, 72, 72), (1, 64, 288, 288),
                                   (2, 7, 18, 20), (1, 3, 2, 4), (2, 512, 36, 36), (2, 130, 6, 8)])
def test_cbam_three_pass_backward(shape):
    both(case_cbam_three_pass, *shape, tol=3e-5)
    both(case_cbam_three_pass, *shape, pad_c=3, tol=3e-5)
    both(case_cbam_three_pass, *shape, pool=False, tol=3e-5)


def test_cbam_three_pass_without_pooling_takes_any_plane_of_whole_float4():
    both(case_cbam_three_pass, 2, 512, 18, 18, pool=False, tol=3e-5)   # the last level of the network
    both(case_cbam_three_pass, 2, 96, 5, 12, pool=False, tol=3e-5)


def test_cbam_three_pass_declines_shapes_it_does_not_take():
    L = _lib.get()
    x = torch.zeros(1, 2, 6, 6, device="cuda")
    assert L.smaat_cbam_bwd3_ok(P(x), 72, P(x), 72, None, 0, 1, 2, 6, 6, 0) == 1      # no pooling: any plane of whole float4
    assert L.smaat_cbam_bwd3_ok(P(x), 72, P(x), 72, P(x), 18, 1, 2, 6, 6, 0) == 0     # pooling: W % 4
    x = torch.zeros(1, 2, 5, 8, device="cuda")
    assert L.smaat_cbam_bwd3_ok(P(x), 80, P(x), 80, P(x), 16, 1, 2, 5, 8, 0) == 0     # pooling: odd H
    x = torch.zeros(1, 2, 3, 5, device="cuda")
    assert L.smaat_cbam_bwd3_ok(P(x), 30, P(x), 30, None, 0, 1, 2, 3, 5, 0) == 0      # H * W % 4
    x = torch.zeros(1, 2, 4, 8, device="cuda")
    assert L.smaat_cbam_bwd3_ok(P(x), 64, P(x), 64, P(x), 16, 1, 2, 4, 8, 0) == 1
    assert L.smaat_cbam_bwd3_ok(P(x) + 4, 64, P(x), 64, None, 0, 1, 2, 4, 8, 0) == 0  # misaligned planes
    dx = torch.zeros(1, 2, 6, 6, device="cuda")
    f = torch.zeros(64, device="cuda")
    i = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert L.smaat_cbam_bwd_apply_t(P(dx), 72, P(dx), 72, P(f), P(f), P(f), P(i), P(f), P(f), P(i), P(dx), 18, 1, 2, 6, 6, P(dx), 72,
                                    0, 0) == -2
    assert L.smaat_cbam_sppool_idx_t(P(dx), 30, P(f), 1, 2, 15, P(f), P(i), 0, 0) == -2
