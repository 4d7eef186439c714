"""GPU: the two-term fp16 split GEMMs (round 5; csrc/splitmma.hip NT == 2, include/smaat_hip.h "two-term fp16 split") and the
kernels that produce their operand maxima, through the C ABI.

reference arithmetic: nn.Conv2d(K, Cout, 1) forward and its autograd, /root/reference models/layers.py:45,49 (f32 in ATen).
Bars: every GEMM against an fp64 evaluation NEXT TO the exact three-term bf16 split it replaces (f32-class: no worse than
3 x that kernel's error + 2e-7) and against the numpy twin (tests/emu_backend.py evaluates the same three fp16 products);
operand images bit-exact against the twin; maxima bit-exact; producers' main outputs bit-identical to the entry points
without the side output; adversarial operands (one channel 1e8 above the rest, gradients in the denormal range, all-zero
planes, NaN) stay f32-class / propagate."""
import numpy as np
import pytest
import torch

from smaat_unet_amd import _lib
from tests.emu_backend import EmuLib, f16_kexp
from tests.test_gpu_kernels import P, T, both, part_stats, rel, rnd, stream

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


AMAX_WORDS = 1024  # SMAAT_AMAX_WORDS


def _amax_word(dev):
    return torch.zeros(AMAX_WORDS, dtype=torch.int32, device=dev)


def _bits(x):
    return int(np.array([x], np.float32).view(np.uint32)[0])


def _publish(t):
    """an amax word holding max |t| as the producing kernels would leave it"""
    w = torch.zeros(AMAX_WORDS, dtype=torch.int32, device=t.device)
    m = float(t.abs().max()) if t.numel() else 0.0
    w[0] = np.array([m], np.float32).view(np.int32)[0].item()
    return w


def _amax_of(buf):
    """the maximum an amax buffer holds (bit patterns of non-negative floats: integer max = float max)"""
    return int(buf.max().item())


def _h_image(L, dev, w, transposed=False):
    R, C = (w.shape[1], w.shape[0]) if transposed else w.shape
    pl = torch.full((int(L.smaat_split_planes_h_bytes(R, C)) // 2,), -1, dtype=torch.int16, device=dev)
    assert L.smaat_split_planes_h(P(w), R, C, P(pl), 1 if transposed else 0, stream(dev)) == 0
    return pl


# ------------------------------------------------------------------------------------------------ operand images
def case_planes_h(L, dev, R, C, transposed, scale):
    w = T(rnd(5, C, R, scale=scale) if transposed else rnd(5, R, C, scale=scale), dev)
    pl = _h_image(L, dev, w, transposed)
    Cp = (C + 15) // 16 * 16
    n = (Cp // 16) * 2 * R * 16
    return dict(image=pl[:n].to(torch.int32), kexp=pl[n:n + 2].view(torch.int32).clone())


@pytest.mark.parametrize("shape", [(64, 128, False, 0.2), (128, 64, True, 0.2), (70, 37, False, 3e-9), (19, 200, True, 4e5),
                                   (512, 2048, False, 0.05), (2048, 512, True, 0.05), (1, 1, False, 1.0)])
def test_split_planes_h_is_the_twin_bit_for_bit(shape):
    r = both(case_planes_h, *shape, tol=0.0)
    assert r["hip"]["kexp"][0] == r["emu"]["kexp"][0]


def test_weight_planes_multi_h_equals_the_single_matrix_entry_points():
    """kind-3 rows next to kind-0 rows in one refresh: images and exponents bit-identical to smaat_split_planes_h / smaat_split_planes"""
    L, dev = _lib.get(), DEV
    ws = [T(rnd(10 + i, r, c, scale=s), dev) for i, (r, c, s) in enumerate([(64, 128, 0.1), (300, 70, 2.0), (128, 512, 1e-3), (64, 64, 1.0)])]
    kinds = [3, 3, 0, 3]
    src_t = [0, 1, 0, 0]
    rows, outs, b0, hp = [], [], 0, 0
    for w, kind, st in zip(ws, kinds, src_t):
        R, C = (w.shape[1], w.shape[0]) if st else w.shape
        Cp = (C + 15) // 16 * 16
        if kind == 3:
            o = torch.full((int(L.smaat_split_planes_h_bytes(R, C)) // 2,), -1, dtype=torch.int16, device=dev)
            hp += L.smaat_split_planes_h_pieces(R, C)
        else:
            o = torch.full((3 * R * Cp,), -1, dtype=torch.int16, device=dev)
        nb = (R * Cp + 255) // 256
        rows.append([w.data_ptr(), o.data_ptr(), R, C, kind, st, b0, nb])
        b0 += nb
        outs.append(o)
    desc = torch.tensor(rows, dtype=torch.int64).to(dev)
    assert L.smaat_weight_planes_multi_h(P(desc), len(rows), b0, hp, stream(dev)) == 0
    torch.cuda.synchronize()
    for w, kind, st, o in zip(ws, kinds, src_t, outs):
        R, C = (w.shape[1], w.shape[0]) if st else w.shape
        if kind == 3:
            ref = _h_image(L, dev, w, bool(st))
            n = ((C + 15) // 16) * 2 * R * 16 + 2  # image + the exponent word (the scratch behind it is unspecified)
            assert torch.equal(o[:n], ref[:n])
        else:
            ref = torch.empty_like(o)
            assert L.smaat_split_planes(P(w), R, C, P(ref), stream(dev)) == 0
            assert torch.equal(o, ref)


# ------------------------------------------------------------------------------------------------ producers of the maxima
@pytest.mark.parametrize("shape", [(2, 12, 2, 32, 32), (3, 20, 2, 18, 18), (2, 4, 2, 144, 144), (1, 3, 4, 8, 12), (2, 5, 1, 8, 8),
                                   (2, 64, 2, 36, 36), (1, 8, 2, 288, 288), (7, 9, 2, 12, 8), (1, 2, 2, 1, 8)])
@pytest.mark.parametrize("aff", [False, True])
def test_dw3x3_fwd_amax(shape, aff):
    """y bit-identical to smaat_dw3x3_fwd; the word holds exactly max |y| (bit pattern)"""
    L, dev = _lib.get(), DEV
    N, Cin, kpl, H, W = shape
    K = Cin * kpl
    x = T(rnd(1, N, Cin, H, W), dev)
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    y0 = torch.full((N, K, H, W), float("nan"), device=dev)
    y1 = torch.full((N, K, H, W), float("nan"), device=dev)
    assert L.smaat_dw3x3_fwd(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(y0), K * H * W, N, Cin, kpl, H, W, stream(dev)) == 0
    am = _amax_word(dev)
    assert L.smaat_dw3x3_fwd_amax(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(y1), K * H * W, P(am), N, Cin, kpl, H, W,
                                  stream(dev)) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert _amax_of(am) == _amax_of(_publish(y1)), (_amax_of(am), _amax_of(_publish(y1)))
    used = am.nonzero().flatten().cpu().numpy()
    assert len(used) >= 1 and (used % 32 == 0).all()  # partial maxima only in words 0, 32, 64, ... (one per 128-byte line)


def test_dw3x3_fwd_amax_refuses_what_the_row_kernels_do_not_take():
    L, dev = _lib.get(), DEV
    t = torch.zeros(1, 4, 9, 11, device=dev)
    y = torch.zeros(1, 8, 9, 11, device=dev)
    am = _amax_word(dev)
    assert L.smaat_dw3x3_fwd_amax(P(t), 4 * 99, None, None, P(t), None, P(y), 8 * 99, P(am), 1, 4, 2, 9, 11, stream(dev)) == -2
    assert L.smaat_dw3x3_fwd_amax(P(t), 4 * 99, None, None, P(t), None, P(y), 8 * 99, None, 1, 4, 2, 9, 11, stream(dev)) == -1


@pytest.mark.parametrize("shape", [(2, 64, 32 * 32), (3, 20, 18 * 18), (2, 7, 99), (1, 64, 288 * 288), (4, 130, 36 * 36)])
@pytest.mark.parametrize("head", [False, True])
def test_bn_bwd_apply_amax(shape, head):
    """dz bit-identical to smaat_bn_bwd_apply / _head; the word holds exactly max |dz|"""
    L, dev = _lib.get(), DEV
    N, C, Pn = shape
    z = T(rnd(1, N, C, Pn), dev)
    dy = T(rnd(2, N, 1 if head else C, Pn, scale=3e-4), dev)
    hw = T(rnd(8, C), dev) if head else None
    scale, shift = T(np.abs(rnd(3, C)) + 0.5, dev), T(rnd(4, C, scale=0.2), dev)
    mean, invstd = T(rnd(5, C, scale=0.1), dev), T(np.abs(rnd(6, C)) + 0.5, dev)
    coef = T(rnd(9, 3, C, scale=0.1), dev)
    dz0 = torch.full((N, C, Pn), float("nan"), device=dev)
    dz1 = torch.full((N, C, Pn), float("nan"), device=dev)
    dy_bs = Pn if head else C * Pn
    if head:
        assert L.smaat_bn_bwd_apply_head(P(dy), dy_bs, P(hw), P(z), C * Pn, P(scale), P(shift), P(mean), P(invstd), P(coef), P(dz0),
                                         C * Pn, N, C, Pn, stream(dev)) == 0
    else:
        assert L.smaat_bn_bwd_apply(P(dy), dy_bs, P(z), C * Pn, P(scale), P(shift), P(mean), P(invstd), P(coef), P(dz0), C * Pn, N,
                                    C, Pn, 1, stream(dev)) == 0
    am = _amax_word(dev)
    assert L.smaat_bn_bwd_apply_amax(P(dy), dy_bs, P(hw), P(z), C * Pn, P(scale), P(shift), P(mean), P(invstd), P(coef), P(dz1),
                                     C * Pn, P(am), N, C, Pn, 1, stream(dev)) == 0
    torch.cuda.synchronize()
    assert torch.equal(dz0, dz1)
    assert _amax_of(am) == _amax_of(_publish(dz1))


# ------------------------------------------------------------------------------------------------ GEMMs
def case_pw_split_h(L, dev, N, C, M, H, W, with_part=False, x=None, w=None, slices=0, bias=True):
    x = T(rnd(1, N, C, H, W) * np.exp(rnd(7, N, C, 1, 1)) if x is None else x, dev)   # wide dynamic range across channels
    w = T(rnd(2, M, C, scale=0.2) if w is None else w, dev)
    b = T(rnd(3, M), dev) if bias else None
    pl = _h_image(L, dev, w)
    am = _publish(x)
    out = torch.full((N, M, H, W), float("nan"), device=dev)
    slots = L.smaat_pw_split_num_slots(N, H, W)
    part = torch.full((3, slots, M), float("nan"), device=dev) if with_part else None
    if slices:
        ws = torch.empty(N * slices * M * H * W, device=dev)
        rc = L.smaat_pointwise_fwd_split_k_h(P(x), C * H * W, P(am), P(pl), P(b), P(out), M * H * W, P(part), P(ws), slices, N, C, M,
                                             H, W, stream(dev))
    else:
        rc = L.smaat_pointwise_fwd_split_h(P(x), C * H * W, P(am), P(pl), P(b), P(out), M * H * W, P(part), N, C, M, H, W,
                                           stream(dev))
    assert rc == 0, rc
    r = dict(out=out)
    if with_part:
        r["pn"], r["pmean"], r["pvar"] = part_stats(part)
    return r


PW_SHAPES = [(2, 64, 1, 32, 32), (2, 64, 21, 16, 16), (2, 16, 3, 6, 7), (2, 128, 256, 36, 36), (4, 64, 70, 288, 288),
             (2, 1, 64, 20, 20), (2, 512, 1024, 18, 18), (4, 40, 130, 288, 288), (2, 24, 64, 32, 32), (1, 37, 19, 5, 9),
             (4, 16, 64, 288, 288), (3, 8, 200, 144, 144)]


@pytest.mark.parametrize("shape", PW_SHAPES)
def test_pointwise_fwd_split_h(shape):
    both(case_pw_split_h, *shape, tol=2e-6)
    both(case_pw_split_h, *shape, with_part=True, tol=2e-5)  # (the variance partials: as test_pointwise_fwd_split)


@pytest.mark.parametrize("shape", [(2, 256, 64, 18, 18, 2), (3, 1024, 512, 18, 18, 4), (1, 512, 130, 6, 10, 2)])
def test_pointwise_fwd_split_k_h(shape):
    *s, S = shape
    both(case_pw_split_h, *s, with_part=True, slices=S, tol=2e-5)


def _fp64_pw(x, w, b):
    return np.einsum("mc,nchw->nmhw", w.astype(np.float64), x.astype(np.float64)) + b.astype(np.float64)[None, :, None, None]


@pytest.mark.parametrize("shape", [(2, 128, 64, 72, 64), (2, 1024, 512, 18, 18), (4, 256, 128, 36, 36), (1, 64, 64, 288, 288)])
def test_pointwise_fwd_split_h_against_fp64_next_to_the_three_term_split(shape):
    L, dev = _lib.get(), DEV
    N, C, M, H, W = shape
    x = rnd(1, N, C, H, W) * np.exp(rnd(7, N, C, 1, 1))
    w, b = rnd(2, M, C, scale=0.2), rnd(3, M)
    ref = _fp64_pw(x, w, b)
    new = case_pw_split_h(L, dev, N, C, M, H, W)["out"].cpu().numpy()
    from tests.test_gpu_kernels import case_pw_split
    old = case_pw_split(L, dev, N, C, M, H, W)["out"].cpu().numpy()
    e_new, e_old = rel(new, ref), rel(old, ref)
    assert e_new < 1e-6 and e_new < 3 * e_old + 2e-7, (e_new, e_old)
    again = case_pw_split_h(L, dev, N, C, M, H, W)["out"].cpu().numpy()
    assert np.array_equal(new, again)  # bit-reproducible


def test_pointwise_fwd_split_h_adversarial_operands():
    """(a) one input channel 1e8 above the rest, (b) one weight row 1e8 above the rest, (c) an operand whose values sit in the
    denormal range of f32 / far below fp16's range, (d) all-zero operand, (e) a NaN: f32-class in rel-L2 (a-c), per output row
    bounded (b), exact zeros + bias (d), propagated (e)"""
    L, dev = _lib.get(), DEV
    N, C, M, H, W = 2, 128, 64, 16, 32
    b = rnd(3, M)
    # (a)
    x = rnd(1, N, C, H, W)
    x[:, 5] *= 1e8
    w = rnd(2, M, C, scale=0.2)
    out = case_pw_split_h(L, dev, N, C, M, H, W, x=x, w=w)["out"].cpu().numpy()
    assert rel(out, _fp64_pw(x, w, b)) < 1e-6
    # with that channel's weights zeroed the result is made of the SMALL channels only: they sit 2^26 below the tensor's
    # maximum, their second fp16 term is subnormal -- the documented loss: relative error ~2^-13 on those elements, i.e. an
    # absolute error <= 2^-39 of the operand maximum
    w0 = w.copy()
    w0[:, 5] = 0
    out0 = case_pw_split_h(L, dev, N, C, M, H, W, x=x, w=w0)["out"].cpu().numpy()
    ref0 = _fp64_pw(x, w0, b)
    assert np.abs(out0 - ref0).max() <= 2.0 ** -36 * np.abs(x).max() * np.abs(w0).sum(1).max()
    # (b)
    w2 = rnd(2, M, C, scale=0.2)
    w2[7] *= 1e8
    x2 = rnd(1, N, C, H, W)
    out2 = case_pw_split_h(L, dev, N, C, M, H, W, x=x2, w=w2)["out"].cpu().numpy()
    ref2 = _fp64_pw(x2, w2, b)
    assert rel(out2, ref2) < 1e-6 and rel(out2[:, 7], ref2[:, 7]) < 1e-6
    small = np.arange(M) != 7
    assert np.abs(out2[:, small] - ref2[:, small]).max() <= 2.0 ** -36 * np.abs(w2).max() * np.abs(x2).sum(1).max()
    # (c) tiny operand: the scale lifts it into range
    zb = np.zeros(M, np.float32)
    x3 = rnd(1, N, C, H, W) * np.float32(1e-36)
    out3 = case_pw_split_h(L, dev, N, C, M, H, W, x=x3, w=w, bias=False)["out"].cpu().numpy()
    assert rel(out3, _fp64_pw(x3, w, zb)) < 1e-5  # (results near the f32 denormal boundary themselves)
    x4 = rnd(1, N, C, H, W) * np.float32(1e-20)
    out4 = case_pw_split_h(L, dev, N, C, M, H, W, x=x4, w=w, bias=False)["out"].cpu().numpy()
    assert rel(out4, _fp64_pw(x4, w, zb)) < 1e-6
    # (d)
    out5 = case_pw_split_h(L, dev, N, C, M, H, W, x=np.zeros((N, C, H, W), np.float32), w=w)["out"].cpu().numpy()
    assert np.array_equal(out5, np.broadcast_to(b[None, :, None, None], out5.shape))
    # (e)
    x6 = rnd(1, N, C, H, W)
    x6[1, 3, 2, 2] = np.nan
    out6 = case_pw_split_h(L, dev, N, C, M, H, W, x=x6, w=w)["out"].cpu().numpy()
    assert np.isnan(out6[1, :, 2, 2]).all() and np.isfinite(out6[0]).all()


def case_wgrad_h(L, dev, N, C, M, H, W, x=None, dz=None):
    x = T(rnd(1, N, C, H, W) if x is None else x, dev)
    dz = T(rnd(2, N, M, H, W) * np.exp(2 * rnd(8, N, M, 1, 1)) * 1e-4 if dz is None else dz, dev)  # log-normal gradient scales
    ns = L.smaat_wgrad_num_splits(N, H, W, M, C)
    ws = torch.empty((ns, M, C), device=dev)
    dw = torch.full((M, C), float("nan"), device=dev)
    ax, adz = _publish(x), _publish(dz)
    assert L.smaat_pointwise_wgrad_h(P(x), C * H * W, P(ax), P(dz), M * H * W, P(adz), P(ws), P(dw), N, C, M, H, W, stream(dev)) == 0
    return dict(dw=dw)


@pytest.mark.parametrize("shape", [(2, 64, 1, 32, 32), (2, 64, 21, 16, 16), (2, 16, 3, 6, 7), (2, 64, 1, 288, 288),
                                   (2, 128, 64, 36, 36), (2, 24, 64, 64, 64), (3, 130, 70, 9, 11), (2, 256, 200, 18, 18),
                                   (1, 1024, 512, 18, 18), (2, 256, 64, 144, 144), (1, 5, 3, 7, 9)])
def test_pointwise_wgrad_h(shape):
    both(case_wgrad_h, *shape, tol=2e-5)


@pytest.mark.parametrize("shape", [(2, 256, 64, 72, 64), (2, 1024, 512, 18, 18), (2, 128, 128, 144, 144)])
def test_pointwise_wgrad_h_against_fp64_next_to_the_three_term_split(shape):
    L, dev = _lib.get(), DEV
    N, C, M, H, W = shape
    x = rnd(1, N, C, H, W)
    dz = rnd(2, N, M, H, W) * np.exp(2 * rnd(8, N, M, 1, 1)) * 1e-4
    dz[:, :, ::7, ::5] *= 1e4  # outliers
    ref = np.einsum("nmp,nkp->mk", dz.astype(np.float64).reshape(N, M, -1), x.astype(np.float64).reshape(N, C, -1))
    new = case_wgrad_h(L, dev, N, C, M, H, W, x=x, dz=dz)["dw"].cpu().numpy()
    ns = L.smaat_wgrad_num_splits(N, H, W, M, C)
    ws = torch.empty((ns, M, C), device=dev)
    old = torch.empty((M, C), device=dev)
    assert L.smaat_pointwise_wgrad(P(T(x, dev)), C * H * W, P(T(dz, dev)), M * H * W, P(ws), P(old), N, C, M, H, W, stream(dev)) == 0
    e_new, e_old = rel(new, ref), rel(old.cpu().numpy(), ref)
    assert e_new < 2e-6 and e_new < 3 * e_old + 2e-7, (e_new, e_old)
    assert np.array_equal(new, case_wgrad_h(L, dev, N, C, M, H, W, x=x, dz=dz)["dw"].cpu().numpy())


def test_pointwise_wgrad_h_denormal_range_gradient_and_zero_planes():
    L, dev = _lib.get(), DEV
    N, C, M, H, W = 2, 128, 64, 16, 32
    x = rnd(1, N, C, H, W)
    x[:, 10] = 0  # an all-zero plane of y
    dz = rnd(2, N, M, H, W) * np.float32(3e-39)  # f32 denormals
    dz[:, 3] = 0
    ref = np.einsum("nmp,nkp->mk", dz.astype(np.float64).reshape(N, M, -1), x.astype(np.float64).reshape(N, C, -1))
    dw = case_wgrad_h(L, dev, N, C, M, H, W, x=x, dz=dz)["dw"].cpu().numpy()
    assert np.all(dw[3] == 0) and np.all(dw[:, 10] == 0)
    # (dz itself carries ~8 significant bits down there; the GEMM must not lose more than the operand has)
    assert rel(dw, ref) < 1e-5
    dz2 = rnd(2, N, M, H, W) * np.float32(1e-30)
    ref2 = np.einsum("nmp,nkp->mk", dz2.astype(np.float64).reshape(N, M, -1), x.astype(np.float64).reshape(N, C, -1))
    assert rel(case_wgrad_h(L, dev, N, C, M, H, W, x=x, dz=dz2)["dw"].cpu().numpy(), ref2) < 1e-6


def test_kexp_matches_the_twin_on_edge_maxima():
    """the scale exponent the kernels derive from a maximum word: spot values through a 1 x 1 GEMM"""
    L, dev = _lib.get(), DEV
    for m in (1.0, 65504.0, 3e-39, 1e-45, 3e38, 2.0 ** -113, 0.75):
        x = np.full((1, 16, 2, 2), m, np.float32)
        w = np.ones((1, 16), np.float32)
        out = case_pw_split_h(L, dev, 1, 16, 1, 2, 2, x=x, w=w)["out"].cpu().numpy()
        want = np.float32(16) * np.float32(m) + rnd(3, 1)[0]
        assert np.allclose(out, want, rtol=2e-6, atol=0), (m, out.ravel()[:2], want, f16_kexp(_bits(m)))


# ------------------------------------------------------------------------------------------------ the row-walking pair
def case_rows_amax_and_wgrad_h(L, dev, N, Cin, Cout, H, W, aff=False):
    """smaat_dsconv_fwd_rows_amax: z and the BatchNorm partials as smaat_dsconv_fwd_rows, + max |y|; then
    smaat_dsconv_wgrad_split_h with that buffer and the one of dz"""
    K = 2 * Cin
    x = T(rnd(1, N, Cin, H, W), dev)
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    w_pw, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    Kp = (K + 15) // 16 * 16
    pl = torch.empty((3, Cout, Kp), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w_pw), Cout, K, P(pl), stream(dev)) == 0
    slots = L.smaat_dsconv_rows_num_slots(N, H, W)
    z = torch.full((N, Cout, H, W), float("nan"), device=dev)
    part = torch.full((3, slots, Cout), float("nan"), device=dev)
    ay = _amax_word(dev)
    assert L.smaat_dsconv_fwd_rows_amax(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(pl), P(b_pw), P(z), Cout * H * W, P(part),
                                        P(ay), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    dz = T(rnd(8, N, Cout, H, W) * np.exp(2 * rnd(9, N, Cout, 1, 1)) * 1e-4, dev)
    adz = _publish(dz)
    ws = torch.empty((L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W), Cout, K), device=dev)
    dw = torch.full((Cout, K), float("nan"), device=dev)
    assert L.smaat_dsconv_wgrad_split_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ay), P(dz), Cout * H * W, P(adz), P(ws),
                                        P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    pn, pmean, pvar = part_stats(part)
    return dict(z=z, dw=dw, amax=torch.tensor([_amax_of(ay)], dtype=torch.int64), pn=pn, pmean=pmean)


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 40, 50, 8, 64), (2, 128, 64, 36, 96), (1, 8, 16, 70, 32)])
@pytest.mark.parametrize("aff", [False, True])
def test_rows_forward_amax_and_recompute_wgrad_h(shape, aff):
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W = shape
    r = both(case_rows_amax_and_wgrad_h, *shape, aff=aff, tol=1e-5)
    # the maximum is that of the depthwise output the standalone kernel writes (the producers form y in its tap order)
    K = 2 * Cin
    x = T(rnd(1, N, Cin, H, W), dev)
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    y = torch.empty((N, K, H, W), device=dev)
    assert L.smaat_dw3x3_fwd(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(y), K * H * W, N, Cin, 2, H, W, stream(dev)) == 0
    assert int(r["hip"]["amax"][0]) == _amax_of(_publish(y))
    # z is bit-identical to the entry point without the side output
    z0 = torch.empty((N, Cout, H, W), device=dev)
    w_pw, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    pl = torch.empty((3, Cout, (K + 15) // 16 * 16), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w_pw), Cout, K, P(pl), stream(dev)) == 0
    assert L.smaat_dsconv_fwd_rows(P(x), 0, Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(pl), P(b_pw), P(z0), 0, Cout * H * W, None,
                                   N, Cin, 2, Cout, H, W, stream(dev)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(z0.cpu().numpy(), r["hip"]["z"])


@pytest.mark.parametrize("aff", [False, True])
def test_recompute_wgrad_h_is_bit_reproducible(aff):
    """six calls on the same inputs: bit-identical (the scalar-math AFF build of this kernel was not -- profiles/r5/
    dswgrad_h_aff_scalar_build_nondeterministic.txt -- and is no longer instantiated)"""
    L, dev = _lib.get(), DEV
    first = case_rows_amax_and_wgrad_h(L, dev, 2, 64, 64, 32, 32, aff=aff)["dw"].cpu().numpy()
    for _ in range(5):
        assert np.array_equal(first, case_rows_amax_and_wgrad_h(L, dev, 2, 64, 64, 32, 32, aff=aff)["dw"].cpu().numpy())


def test_recompute_wgrad_h_against_fp64_next_to_the_three_term_kernel():
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W = 2, 96, 64, 72, 64
    K = 2 * Cin
    from oracle import smaat_oracle as O
    x, dzs = rnd(1, N, Cin, H, W), rnd(8, N, Cout, H, W) * np.exp(2 * rnd(9, N, Cout, 1, 1)) * 1e-4
    w_dw, b_dw = rnd(2, K, 9, scale=0.3), rnd(3, K, scale=0.3)
    y64 = O.dw3x3_fwd(x.astype(np.float64), w_dw.astype(np.float64).reshape(K, 1, 3, 3), b_dw.astype(np.float64), 2)
    ref = np.einsum("nmp,nkp->mk", dzs.astype(np.float64).reshape(N, Cout, -1), y64.reshape(N, K, -1))
    new = case_rows_amax_and_wgrad_h(L, dev, N, Cin, Cout, H, W)["dw"].cpu().numpy()
    ws = torch.empty((L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W), Cout, K), device=dev)
    old = torch.empty((Cout, K), device=dev)
    assert L.smaat_dsconv_wgrad_split(P(T(x, dev)), Cin * H * W, None, None, P(T(w_dw, dev)), P(T(b_dw, dev)), P(T(dzs, dev)),
                                      Cout * H * W, P(ws), P(old), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    e_new, e_old = rel(new, ref), rel(old.cpu().numpy(), ref)
    assert e_new < 2e-6 and e_new < 3 * e_old + 2e-7, (e_new, e_old)
    assert np.array_equal(new, case_rows_amax_and_wgrad_h(L, dev, N, Cin, Cout, H, W)["dw"].cpu().numpy())


def test_second_backward_over_a_retained_graph_uses_its_own_gradient_maximum():
    """max |dz| of a block is accumulated with atomic max into words that must hold 0 on entry.  A second backward over
    the same graph (retain_graph: several losses, Jacobian rows) with a cotangent 1e-8 times the first one's must not be scaled by
    the first pass's maximum -- its second fp16 term would vanish (2^-27 below the stale scale) and the operand itself would sit
    in the fp16 denormals.  Gradients are linear in the cotangent (BatchNorm statistics are fixed by the forward): pass 2 must be
    1e-8 x pass 1 to f32-class accuracy."""
    import smaat_unet_amd as S
    from smaat_unet_amd import ops as K
    torch.manual_seed(3)
    blk = S.unet_parts_depthwise_separable.DoubleConvDS(64, 64, kernels_per_layer=2).to(DEV).train()
    x = torch.randn(2, 64, 64, 64, device=DEV, requires_grad=True)
    assert 2 * 64 * 64 >= K.policy.f16_min_samples  # (the two-term fp16 split is what runs)
    out = blk(x)
    g = torch.randn_like(out)
    params = [p for p in blk.parameters()]
    first = torch.autograd.grad(out, [x] + params, grad_outputs=g, retain_graph=True)
    second = torch.autograd.grad(out, [x] + params, grad_outputs=g * 1e-8)
    for (name, _), a, b in zip([("x", None)] + list(blk.named_parameters()), first, second):
        if name.endswith("pointwise.bias"):  # exactly zero in front of a train-mode BatchNorm
            assert float(b.abs().max()) == 0.0
            continue
        if name.endswith("depthwise.bias"):  # true gradient 0, computed as a sum: roundoff only (SURVEY 8c "zero-gradient trap")
            continue
        na = float(a.double().norm())
        err = float((b.double() * 1e8 - a.double()).norm()) / max(na, 1e-30)
        assert err < 2e-5, (name, err)


# ------------------------------------------------------------------------------------------------ round 6: bit-repeat soak
SOAK_CALLS = int(__import__("os").environ.get("SMAAT_SOAK_CALLS", "1000"))


def _soak(call, outs, what):
    """`call()` launches one kernel into the tensors `outs`; SOAK_CALLS launches on identical inputs must all be bit-identical to
    the first (every kernel here has fixed reduction orders and no data-dependent atomics other than the order-independent maxima)"""
    call()
    torch.cuda.synchronize()
    first = [o.clone() for o in outs]
    bad = 0
    for i in range(1, SOAK_CALLS):
        for o in outs:
            o.fill_(float("nan")) if o.is_floating_point() else o.zero_()
        call()
        if not all(torch.equal(a, b) or (a.is_floating_point() and torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num()))
                   for a, b in zip(first, outs)):
            bad += 1
    assert bad == 0, f"{what}: {bad} of {SOAK_CALLS} calls differ from the first"


@pytest.mark.parametrize("aff", [False, True])
@pytest.mark.parametrize("shape", [(4, 64, 64, 288, 288), (4, 128, 64, 288, 288), (2, 64, 64, 32, 32)])
def test_soak_rows_forward_and_recompute_wgrad_h(shape, aff):
    """k_dsconv_rows_fwd + k_dsconv_wgrad_split<NT=2> (inline-asm loads, counted waits) at the 288 x 288 plane of BASELINE
    configs[1] (inc.1 / up4.1: 64 -> 128 -> 64; up4.0: 128 -> 256 -> 64; batch 4 = 36 items per workgroup as at batch 32 on 8x
    the chip) and on the shape round 5's nondeterministic instantiation was found on: 1,000 bit-identical calls each."""
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W = shape
    K = 2 * Cin
    x = T(rnd(1, N, Cin, H, W), dev)
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    w_pw, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    pl = torch.empty((3, Cout, (K + 15) // 16 * 16), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w_pw), Cout, K, P(pl), stream(dev)) == 0
    slots = L.smaat_dsconv_rows_num_slots(N, H, W)
    z = torch.empty((N, Cout, H, W), device=dev)
    part = torch.empty((3, slots, Cout), device=dev)
    ay = _amax_word(dev)

    def fwd():
        ay.zero_()
        assert L.smaat_dsconv_fwd_rows_amax(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(pl), P(b_pw), P(z), Cout * H * W, P(part),
                                            P(ay), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    _soak(fwd, [z, part], "k_dsconv_rows_fwd")
    ay_max = torch.zeros_like(ay)
    ay_max[0] = ay.max()  # (which slot of the buffer a wave publishes into is its own business; the maximum is what is defined)
    dz = T(rnd(8, N, Cout, H, W) * np.exp(2 * rnd(9, N, Cout, 1, 1)) * 1e-4, dev)
    adz = _publish(dz)
    ws = torch.empty((L.smaat_dsconv_wgrad_split_num_splits(N, Cin, Cout, H, W), Cout, K), device=dev)
    dw = torch.empty((Cout, K), device=dev)

    def wgrad():
        assert L.smaat_dsconv_wgrad_split_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ay_max), P(dz), Cout * H * W, P(adz),
                                            P(ws), P(dw), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    _soak(wgrad, [dw], "k_dsconv_wgrad_split<NT=2>")


@pytest.mark.parametrize("shape", [(4, 128, 128, 144, 144), (8, 1024, 512, 36, 36), (4, 256, 64, 288, 288)])
def test_soak_split_gemms_h(shape):
    """k_pw_split_p<NT=2> (forward / data gradient) and k_wgrad_split<NT=2> at layer shapes of BASELINE configs[1]:
    1,000 bit-identical calls each."""
    L, dev = _lib.get(), DEV
    N, C, M, H, W = shape
    x = T(rnd(1, N, C, H, W) * np.exp(rnd(7, N, C, 1, 1)), dev)
    w, b = T(rnd(2, M, C, scale=0.2), dev), T(rnd(3, M), dev)
    pl, am = _h_image(L, dev, w), _publish(x)
    out = torch.empty((N, M, H, W), device=dev)
    part = torch.empty((3, L.smaat_pw_split_num_slots(N, H, W), M), device=dev)

    def fwd():
        assert L.smaat_pointwise_fwd_split_h(P(x), C * H * W, P(am), P(pl), P(b), P(out), M * H * W, P(part), N, C, M, H, W, stream(dev)) == 0
    _soak(fwd, [out, part], "k_pw_split_p<NT=2>")
    dz = T(rnd(4, N, M, H, W) * np.exp(2 * rnd(8, N, M, 1, 1)) * 1e-4, dev)
    adz = _publish(dz)
    ws = torch.empty((L.smaat_wgrad_num_splits(N, H, W, M, C), M, C), device=dev)
    dw = torch.empty((M, C), device=dev)

    def wgrad():
        assert L.smaat_pointwise_wgrad_h(P(x), C * H * W, P(am), P(dz), M * H * W, P(adz), P(ws), P(dw), N, C, M, H, W, stream(dev)) == 0
    _soak(wgrad, [dw], "k_wgrad_split<NT=2>")


# ------------------------------------------------------------------------------------------------ round 6: fused backward
def _bwd_two_kernel_and_fused(L, dev, N, Cin, Cout, H, W, aff, gamma_mode="normal"):
    """(two-kernel results, fused results) of the DepthwiseSeparableConv backward on identical inputs: dgrad GEMM
    (smaat_pointwise_fwd_split_h on the transposed image) + smaat_dw3x3_bwd[_bnred]  vs  smaat_dsconv_bwd_rows_h"""
    K = 2 * Cin
    z = rnd(1, N, Cin, H, W) * 1.3 + 0.2
    w_pw = T(rnd(4, Cout, K, scale=0.2), dev)
    w_dw = T(rnd(3, K, 9, scale=0.3), dev)
    dz = T(rnd(8, N, Cout, H, W) * np.exp(2 * rnd(9, N, Cout, 1, 1)) * 1e-3, dev)
    adz = _publish(dz)
    pl_t = _h_image(L, dev, w_pw, transposed=True)
    x = T(z, dev)
    sc = sh = mean = invstd = None
    if aff:
        mean_n = z.mean(axis=(0, 2, 3)).astype(np.float32)
        invstd_n = (1.0 / np.sqrt(z.var(axis=(0, 2, 3)) + 1e-5)).astype(np.float32)
        gam = np.random.default_rng(8).uniform(0.5, 1.5, Cin).astype(np.float32)
        if gamma_mode == "zero":
            gam[::2] = 0.0
        bet = (np.abs(rnd(9, Cin, scale=0.3)) + 0.05).astype(np.float32)
        sc_n = (gam * invstd_n).astype(np.float32)
        sc, sh, mean, invstd = T(sc_n, dev), T((bet - mean_n * sc_n).astype(np.float32), dev), T(mean_n, dev), T(invstd_n, dev)
    # ---- two kernels
    dy = torch.full((N, K, H, W), float("nan"), device=dev)
    assert L.smaat_pointwise_fwd_split_h(P(dz), Cout * H * W, P(adz), P(pl_t), None, P(dy), K * H * W, None, N, Cout, K, H, W, stream(dev)) == 0
    rows = L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W)
    ws = torch.empty((rows, K, 10), device=dev)
    a = dict(dx=torch.full((N, Cin, H, W), float("nan"), device=dev), dw=torch.full((K, 9), float("nan"), device=dev),
             db=torch.full((K,), float("nan"), device=dev))
    if aff:
        rp = torch.full((2, rows - 1, Cin), float("nan"), device=dev)
        assert L.smaat_dw3x3_bwd_bnred(P(x), Cin * H * W, P(sc), P(sh), P(dy), K * H * W, P(w_dw), P(a["dx"]), Cin * H * W, P(ws), P(a["dw"]),
                                       P(a["db"]), P(mean), P(invstd), P(rp), N, Cin, 2, H, W, stream(dev)) == 0
        a["r1"], a["r2"] = rp[0].double().sum(0), rp[1].double().sum(0)
    else:
        assert L.smaat_dw3x3_bwd(P(x), Cin * H * W, P(dy), K * H * W, P(w_dw), P(a["dx"]), Cin * H * W, P(ws), P(a["dw"]), P(a["db"]), N, Cin, 2,
                                 H, W, stream(dev)) == 0
    # ---- fused
    assert L.smaat_dsconv_bwd_rows_ok(2, Cin, Cout, H, W) == 1
    rows2 = L.smaat_dsconv_bwd_rows_num_rows(N, Cin, H, W)
    ws2 = torch.full((rows2, K, 10), float("nan"), device=dev)
    b = dict(dx=torch.full((N, Cin, H, W), float("nan"), device=dev), dw=torch.full((K, 9), float("nan"), device=dev),
             db=torch.full((K,), float("nan"), device=dev))
    rp2 = torch.full((2, rows2, Cin), float("nan"), device=dev) if aff else None
    assert L.smaat_dsconv_bwd_rows_h(P(x), Cin * H * W, P(sc), P(sh), P(mean), P(invstd), P(dz), Cout * H * W, P(adz), P(pl_t), P(w_dw),
                                     P(b["dx"]), Cin * H * W, P(ws2), P(b["dw"]), P(b["db"]), P(rp2), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    if aff:
        b["r1"], b["r2"] = rp2[0].double().sum(0), rp2[1].double().sum(0)
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("aff", [False, True])
@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 64), (1, 128, 64, 24, 96), (2, 64, 64, 37, 70), (1, 64, 64, 9, 32), (3, 128, 64, 75, 45),
                                   (1, 64, 64, 288, 288)])
def test_fused_backward_equals_the_two_kernel_form(shape, aff):
    """smaat_dsconv_bwd_rows_h (dY formed by MFMA on chip, consumed in the same kernel) against the pair it replaces: dX BIT-identical
    (same MFMA sequence as k_pw_split_p<NT = 2>, same FMA order as k_dw3x3_bwd_rows), the depthwise weight / bias gradients and the
    previous BatchNorm's backward sums f32-class (their partial sums are cut differently: per workgroup instead of per wave).
    Shapes: strips that do not divide the width (70, 45: last tile partly outside), bands that do not divide the height, both
    channel-half counts, the 288 x 288 plane of BASELINE configs[1]."""
    L, dev = _lib.get(), DEV
    if aff and shape[4] % 2:
        pytest.skip("the two-kernel form has no on-load activation for odd widths (smaat_dw3x3_strip_ok)")
    a, b = _bwd_two_kernel_and_fused(L, dev, *shape, aff)
    if shape[4] % 4 == 0:
        assert torch.equal(a["dx"], b["dx"]), float((a["dx"] - b["dx"]).abs().max())
    else:  # (odd widths: the two-kernel form runs its general kernels, whose sums associate differently)
        assert rel(b["dx"].cpu().numpy(), a["dx"].cpu().numpy()) < 1e-6
    assert rel(b["dw"].cpu().numpy(), a["dw"].cpu().numpy()) < 2e-6
    assert rel(b["db"].cpu().numpy(), a["db"].cpu().numpy()) < 2e-6
    if aff:
        assert rel(b["r1"].cpu().numpy(), a["r1"].cpu().numpy()) < 1e-5
        assert rel(b["r2"].cpu().numpy(), a["r2"].cpu().numpy()) < 1e-5


def test_fused_backward_degenerate_gamma_and_refusals():
    L, dev = _lib.get(), DEV
    a, b = _bwd_two_kernel_and_fused(L, dev, 2, 64, 64, 32, 64, True, gamma_mode="zero")
    assert torch.equal(a["dx"], b["dx"])
    assert rel(b["r2"].cpu().numpy(), a["r2"].cpu().numpy()) < 1e-5 and rel(b["r1"].cpu().numpy(), a["r1"].cpu().numpy()) < 1e-5
    for bad in ((2, 32, 64, 32, 64), (2, 64, 128, 32, 64), (1, 64, 64, 32, 64), (2, 64, 64, 4, 64), (2, 64, 64, 32, 16)):
        assert L.smaat_dsconv_bwd_rows_ok(*bad) == 0, bad


def test_soak_fused_backward():
    """1,000 bit-identical calls at the 288 x 288 plane (inline-asm dz loads with counted waits in the MFMA waves)"""
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W = 4, 64, 64, 288, 288
    K = 2 * Cin
    x, dz = T(rnd(1, N, Cin, H, W), dev), T(rnd(8, N, Cout, H, W) * 1e-3, dev)
    w_pw, w_dw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(3, K, 9, scale=0.3), dev)
    adz, pl_t = _publish(dz), _h_image(L, dev, w_pw, transposed=True)
    rows = L.smaat_dsconv_bwd_rows_num_rows(N, Cin, H, W)
    ws, dx = torch.empty((rows, K, 10), device=dev), torch.empty((N, Cin, H, W), device=dev)
    dw, db = torch.empty((K, 9), device=dev), torch.empty((K,), device=dev)

    def call():
        assert L.smaat_dsconv_bwd_rows_h(P(x), Cin * H * W, None, None, None, None, P(dz), Cout * H * W, P(adz), P(pl_t), P(w_dw), P(dx),
                                         Cin * H * W, P(ws), P(dw), P(db), None, N, Cin, 2, Cout, H, W, stream(dev)) == 0
    _soak(call, [dx, dw, db], "k_dsconv_bwd_rows")


# ------------------------------------------------------------ round 6: the fused row-walking forward on the two-term split
def _rows_h_inputs(dev, N, Cin, Cout, H, W, aff, prev_K=0, xscale=1.0):
    K = 2 * Cin
    w_dw, b_dw = T(rnd(2, K, 9, scale=0.3), dev), T(rnd(3, K, scale=0.3), dev)
    w_pw, b_pw = T(rnd(4, Cout, K, scale=0.2), dev), T(rnd(5, Cout), dev)
    sc = T(np.random.default_rng(6).uniform(0.5, 1.5, Cin).astype(np.float32), dev) if aff else None
    sh = T(rnd(7, Cin, scale=0.3), dev) if aff else None
    if prev_K:  # x = prev_w . u + prev_b: the output of a previous pointwise convolution whose operand's maximum is known
        u = T(rnd(10, N, prev_K, H, W), dev)
        pw, pb = T(rnd(11, Cin, prev_K, scale=0.25), dev), T(rnd(12, Cin, scale=0.5), dev)
        x = (torch.einsum("ck,nkhw->nchw", pw.double(), u.double()) + pb.double().view(1, -1, 1, 1)).float().contiguous()
        ax = _publish(u)
    else:
        x = T(rnd(1, N, Cin, H, W) * xscale, dev)
        pw = pb = None
        ax = _publish(x)
    return x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax


def case_rows_h(L, dev, N, Cin, Cout, H, W, aff=False, prev_K=0, two=False):
    """smaat_dsconv_fwd_rows_h: z, BatchNorm partials, max |y| (true), max |z|"""
    K = 2 * Cin
    x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax = _rows_h_inputs(dev, N, Cin, Cout, H, W, aff, prev_K)
    ax2 = None
    if two:  # two writers of x: the first half of the channels in one buffer, the rest in the other
        ax, ax2 = _publish(x[:, :Cin // 2]), _publish(x[:, Cin // 2:])
    plh = _h_image(L, dev, w_pw)
    slots = L.smaat_dsconv_rows_num_slots(N, H, W)
    z = torch.full((N, Cout, H, W), float("nan"), device=dev)
    part = torch.full((3, slots, Cout), float("nan"), device=dev)
    ay, az = _amax_word(dev), _amax_word(dev)
    assert L.smaat_dsconv_fwd_rows_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ax), P(ax2), P(pw), P(pb), prev_K, P(plh),
                                     P(b_pw), P(z), Cout * H * W, P(part), P(ay), P(az), N, Cin, 2, Cout, H, W, stream(dev)) == 0
    pn, pmean, pvar = part_stats(part)
    return dict(z=z, amax_y=torch.tensor([_amax_of(ay)], dtype=torch.int64), amax_z=torch.tensor([_amax_of(az)], dtype=torch.int64),
                pn=pn, pmean=pmean, pvar=pvar)


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 40, 50, 8, 64), (2, 128, 64, 36, 96), (1, 8, 16, 70, 32)])
@pytest.mark.parametrize("form", ["amax", "two_writers", "prev_w"])
@pytest.mark.parametrize("aff", [False, True])
def test_rows_forward_h(shape, form, aff):
    """against the numpy twin (same three fp16 products, same a-priori scale exponent): f32 round-off class; the maxima exactly"""
    N, Cin, Cout, H, W = shape
    L, dev = _lib.get(), DEV
    r = both(case_rows_h, *shape, aff=aff, prev_K=24 if form == "prev_w" else 0, two=form == "two_writers", tol=1e-5)
    assert int(r["hip"]["amax_z"][0]) == _bits(float(np.abs(r["hip"]["z"]).max()))
    # the maximum of y is the TRUE one (not the bound), that of the depthwise output the standalone kernel writes (same tap order)
    x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax = _rows_h_inputs(dev, N, Cin, Cout, H, W, aff, 24 if form == "prev_w" else 0)
    y = torch.empty((N, 2 * Cin, H, W), device=dev)
    assert L.smaat_dw3x3_fwd(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(y), 2 * Cin * H * W, N, Cin, 2, H, W, stream(dev)) == 0
    torch.cuda.synchronize()
    assert int(r["hip"]["amax_y"][0]) == _amax_of(_publish(y))


@pytest.mark.parametrize("case", [(2, 64, 64, 64, 64, True, 0), (2, 128, 64, 36, 96, False, 0), (2, 64, 64, 64, 64, True, 24),
                                  (2, 128, 64, 32, 64, True, 256)])
def test_rows_forward_h_against_fp64_next_to_the_three_term_kernel(case):
    """error against an fp64 evaluation no worse than 3 x the exact three-term kernel's + 2e-7 (the bar of the other two-term
    GEMMs), with the bound coming from max |x| or through the previous GEMM's weight (K' = 24: the stem; 256: a decoder block)"""
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W, aff, prev_K = case
    K = 2 * Cin
    x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax = _rows_h_inputs(dev, N, Cin, Cout, H, W, aff, prev_K)
    plh = _h_image(L, dev, w_pw)
    pl3 = torch.empty((3, Cout, (K + 15) // 16 * 16), dtype=torch.int16, device=dev)
    assert L.smaat_split_planes(P(w_pw), Cout, K, P(pl3), stream(dev)) == 0
    zh, z3 = torch.empty((N, Cout, H, W), device=dev), torch.empty((N, Cout, H, W), device=dev)
    ay = _amax_word(dev)
    assert L.smaat_dsconv_fwd_rows_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ax), None, P(pw), P(pb), prev_K, P(plh),
                                     P(b_pw), P(zh), Cout * H * W, None, P(ay), None, N, Cin, 2, Cout, H, W, stream(dev)) == 0
    assert L.smaat_dsconv_fwd_rows(P(x), 0, Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(pl3), P(b_pw), P(z3), 0, Cout * H * W, None,
                                   N, Cin, 2, Cout, H, W, stream(dev)) == 0
    xd = x.double()
    if aff:
        xd = torch.relu(xd * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    yd = torch.nn.functional.conv2d(xd, w_dw.double().view(K, 1, 3, 3), b_dw.double(), padding=1, groups=Cin)
    zd = torch.nn.functional.conv2d(yd, w_pw.double().view(Cout, K, 1, 1), b_pw.double())
    eh, e3 = float((zh.double() - zd).norm() / zd.norm()), float((z3.double() - zd).norm() / zd.norm())
    assert eh <= 3 * e3 + 2e-7, (eh, e3)
    mh = float((zh.double() - zd).abs().max() / zd.abs().max())
    assert mh <= 2e-6, mh


def test_rows_forward_h_loose_bounds_zero_planes_and_nan():
    """a bound 2^16 too large costs at most the subnormal tail (still f32 class); all-zero input with biases; all-zero
    everything (scale exponent 0, exact zeros); a NaN in x comes out as NaNs"""
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W = 2, 64, 64, 32, 64
    K = 2 * Cin
    x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax = _rows_h_inputs(dev, N, Cin, Cout, H, W, False)
    plh = _h_image(L, dev, w_pw)

    def run(xx, axx, bdw=b_dw, bpw=b_pw):
        z = torch.full((N, Cout, H, W), float("nan"), device=dev)
        assert L.smaat_dsconv_fwd_rows_h(P(xx), Cin * H * W, None, None, P(w_dw), P(bdw), P(axx), None, None, None, 0, P(plh), P(bpw),
                                         P(z), Cout * H * W, None, None, None, N, Cin, 2, Cout, H, W, stream(dev)) == 0
        torch.cuda.synchronize()
        return z

    yd = torch.nn.functional.conv2d(x.double(), w_dw.double().view(K, 1, 3, 3), b_dw.double(), padding=1, groups=Cin)
    zd = torch.nn.functional.conv2d(yd, w_pw.double().view(Cout, K, 1, 1), b_pw.double())
    tight = float((run(x, ax).double() - zd).norm() / zd.norm())
    loose = float((run(x, _publish(x * 65536.0)).double() - zd).norm() / zd.norm())
    assert tight < 3e-7 and loose < 3e-6, (tight, loose)
    zero = torch.zeros_like(x)
    z0 = run(zero, _publish(zero))
    ref0 = torch.nn.functional.conv2d(torch.nn.functional.conv2d(zero.double(), w_dw.double().view(K, 1, 3, 3), b_dw.double(), padding=1,
                                                                 groups=Cin), w_pw.double().view(Cout, K, 1, 1), b_pw.double())
    assert float((z0.double() - ref0).norm() / ref0.norm()) < 3e-7
    assert torch.equal(run(zero, _publish(zero), bdw=None, bpw=None), torch.zeros(N, Cout, H, W, device=dev))
    xn = x.clone()
    xn[1, 3, 5, 7] = float("nan")
    zn = run(xn, _publish(x))
    assert torch.isnan(zn[1, :, 4:7, 6:9]).all() and not torch.isnan(zn[0]).any()


def test_rows_forward_h_refusals():
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W = 1, 64, 64, 16, 32
    x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax = _rows_h_inputs(dev, N, Cin, Cout, H, W, False, prev_K=24)
    plh = _h_image(L, dev, w_pw)
    z = torch.empty((N, Cout, H, W), device=dev)
    a = [P(x), Cin * H * W, None, None, P(w_dw), P(b_dw), P(ax), None, P(pw), P(pb), 24, P(plh), P(b_pw), P(z), Cout * H * W, None, None,
         None, N, Cin, 2, Cout, H, W, stream(dev)]
    assert L.smaat_dsconv_fwd_rows_h(*a) == 0
    b = list(a); b[6] = None
    assert L.smaat_dsconv_fwd_rows_h(*b) == -1                      # no maximum
    b = list(a); b[7] = P(ax)
    assert L.smaat_dsconv_fwd_rows_h(*b) == -1                      # the weight form has one maximum
    b = list(a); b[23] = 40; b[14] = Cout * H * 40
    assert L.smaat_dsconv_fwd_rows_h(*b) == -2                      # W % 32
    b = list(a); b[20] = 1
    assert L.smaat_dsconv_fwd_rows_h(*b) == -2                      # kernels_per_layer


@pytest.mark.parametrize("shape", [(2, 64, 288 * 288), (3, 128, 144 * 144), (2, 5, 77), (1, 3, 18 * 18)])
def test_cbam_apply_amax(shape):
    """out bit-identical to smaat_cbam_apply; the buffer holds max |out| exactly"""
    L, dev = _lib.get(), DEV
    N, C, Pn = shape
    x = T(rnd(1, N, C, Pn), dev)
    s = T(np.random.default_rng(2).uniform(0.1, 1.0, (N, C)).astype(np.float32), dev)
    gate = T(np.random.default_rng(3).uniform(0.1, 1.0, (N, Pn)).astype(np.float32), dev)
    o0, o1 = torch.full((N, C, Pn), float("nan"), device=dev), torch.full((N, C, Pn), float("nan"), device=dev)
    am = _amax_word(dev)
    assert L.smaat_cbam_apply(P(x), C * Pn, P(s), P(gate), P(o0), C * Pn, N, C, Pn, stream(dev)) == 0
    assert L.smaat_cbam_apply_amax(P(x), C * Pn, P(s), P(gate), P(o1), C * Pn, P(am), N, C, Pn, stream(dev)) == 0
    torch.cuda.synchronize()
    assert torch.equal(o0, o1)
    assert _amax_of(am) == _bits(float(o1.abs().max()))


@pytest.mark.parametrize("shape", [(2, 64, 144, 144, 288, 288), (2, 16, 9, 9, 20, 20), (1, 8, 18, 18, 36, 36), (1, 4, 5, 7, 10, 14)])
def test_upsample2x_fwd_amax(shape):
    """out bit-identical to smaat_upsample2x_fwd (into a channel slice of a larger buffer); max |out| exactly; -2 where the
    row-walking kernel does not take the shape"""
    L, dev = _lib.get(), DEV
    N, C, H, W, Ho, Wo = shape
    x = T(rnd(1, N, C, H, W), dev)
    pt, pl_ = (Ho - 2 * H) // 2, (Wo - 2 * W) // 2
    c0 = 3
    o0 = torch.full((N, C + c0, Ho, Wo), float("nan"), device=dev)
    o1 = torch.full((N, C + c0, Ho, Wo), float("nan"), device=dev)
    am = _amax_word(dev)
    assert L.smaat_upsample2x_fwd(P(x), C * H * W, o0.data_ptr() + 4 * c0 * Ho * Wo, (C + c0) * Ho * Wo, N, C, H, W, Ho, Wo, pt, pl_,
                                  stream(dev)) == 0
    rc = L.smaat_upsample2x_fwd_amax(P(x), C * H * W, o1.data_ptr() + 4 * c0 * Ho * Wo, (C + c0) * Ho * Wo, P(am), N, C, H, W, Ho, Wo,
                                     pt, pl_, stream(dev))
    if Wo % 4:
        assert rc == -2
        return
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(o0[:, c0:], o1[:, c0:]) and torch.isnan(o1[:, :c0]).all()
    assert _amax_of(am) == _bits(float(o1[:, c0:].abs().max()))


@pytest.mark.parametrize("case", [(4, 64, 64, 288, 288, True, 24), (4, 128, 64, 288, 288, False, 0), (4, 64, 64, 288, 288, True, 256),
                                  (2, 64, 64, 32, 32, True, 0)])
def test_soak_rows_forward_h(case):
    """k_dsconv_rows_fwd<NT=2> (inline-asm loads, counted waits; scripts/isa_hazards.py proves the ISA) on the three layers of the
    step that run it: 1,000 bit-identical calls"""
    L, dev = _lib.get(), DEV
    N, Cin, Cout, H, W, aff, prev_K = case
    x, w_dw, b_dw, w_pw, b_pw, sc, sh, pw, pb, ax = _rows_h_inputs(dev, N, Cin, Cout, H, W, aff, prev_K)
    plh = _h_image(L, dev, w_pw)
    slots = L.smaat_dsconv_rows_num_slots(N, H, W)
    z = torch.empty((N, Cout, H, W), device=dev)
    part = torch.empty((3, slots, Cout), device=dev)
    ay = _amax_word(dev)

    def fwd():
        ay.zero_()
        assert L.smaat_dsconv_fwd_rows_h(P(x), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(ax), None, P(pw), P(pb), prev_K, P(plh),
                                         P(b_pw), P(z), Cout * H * W, P(part), P(ay), None, N, Cin, 2, Cout, H, W, stream(dev)) == 0
    _soak(fwd, [z, part], "k_dsconv_rows_fwd<NT=2>")


# ------------------------------------------------------------ round 6: the prototype GEMM on pre-split planes (measurement only)
@pytest.mark.parametrize("shape", [(2, 64, 128, 16, 32), (1, 512, 256, 12, 24), (3, 32, 200, 8, 16)])
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_pointwise_fwd_h2_prototype_equals_the_shipped_two_term_gemm(shape, cfg):
    """smaat_pointwise_fwd_h2_proto (csrc/h2gemm.hip; not selected by ops.py): the same three fp16 products as
    smaat_pointwise_fwd_split_h on operand planes split OUTSIDE the kernel -- f32 round-off class against it and against fp64"""
    L, dev = _lib.get(), DEV
    N, C, M, H, W = shape
    Pn = H * W
    x, w, b = T(rnd(1, N, C, Pn), dev), T(rnd(2, M, C, scale=0.1), dev), T(rnd(3, M), dev)
    kx, ka = f16_kexp(_bits(float(x.abs().max()))), f16_kexp(_bits(float(w.abs().max())))

    def planes(t, k):
        ts = t * (2.0 ** k)
        h = ts.half()
        return h, (ts - h.float()).half()
    xh, xg = planes(x, kx)
    wh, wg = planes(w, ka)
    xp = torch.stack([xh, xg], dim=1).contiguous()
    ap = torch.stack([t.view(M, C // 16, 16).permute(1, 0, 2).contiguous() for t in (wh, wg)], dim=0).contiguous()
    z = torch.full((N, M, Pn), float("nan"), device=dev)
    assert L.smaat_pointwise_fwd_h2_proto(P(xp), 2 * C * Pn, C * Pn, P(ap), (C // 16) * M * 16, P(b), P(z), M * Pn, None, N, C, M, H, W,
                                          ka + kx, cfg, stream(dev)) == 0
    plh = _h_image(L, dev, w)
    z0 = torch.empty(N, M, Pn, device=dev)
    assert L.smaat_pointwise_fwd_split_h(P(x), C * Pn, P(_publish(x)), P(plh), P(b), P(z0), M * Pn, None, N, C, M, H, W, stream(dev)) == 0
    torch.cuda.synchronize()
    zd = torch.einsum("mc,ncp->nmp", w.double(), x.double()) + b.double().view(1, -1, 1)
    assert float((z - z0).abs().max() / z0.abs().max()) < 2e-6
    assert float((z.double() - zd).norm() / zd.norm()) < 6e-7


def test_pointwise_fwd_h2_prototype_refusals():
    L, dev = _lib.get(), DEV
    t = torch.zeros(1 << 16, dtype=torch.float16, device=dev)
    o = torch.zeros(1 << 16, device=dev)
    call = lambda C, M, H, W, cfg: L.smaat_pointwise_fwd_h2_proto(P(t), 2 * C * H * W, C * H * W, P(t), (C // 16) * M * 16, None, P(o), M * H * W,  # noqa: E731
                                                                  None, 1, C, M, H, W, 0, cfg, stream(dev))
    assert call(32, 64, 8, 16, 2) == -2    # M <= 64
    assert call(24, 128, 8, 16, 2) == -2   # Cin % 32
    assert call(32, 128, 3, 6, 2) == -2    # P % 8
    assert call(32, 128, 8, 16, 7) == -1   # configuration
