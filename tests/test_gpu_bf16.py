"""GPU: the mixed-precision (bf16 activation storage) entry points of libsmaat_hip.so, through the C ABI.

Two kinds of checks (include/smaat_hip.h "mixed precision"):
  * the bf16 GEMM / weight gradient against an fp64 evaluation of the same bf16-rounded operands (the products of two
    bf16 numbers are exact in f32, so only the accumulation order differs: <= 2e-6 before the output rounding);
  * every *_t entry point against its f32 twin -- which tests/test_gpu_kernels.py pins to the oracle -- run on the SAME
    values (inputs that are exactly representable in bf16): the arithmetic is the same f32 code, so a bf16 output must be
    the round-to-nearest-even of the f32 result, bit for bit, and an f32 output must be identical.
"""
import numpy as np
import pytest
import torch

from smaat_unet_amd import _lib

import os  # noqa: E402

# (bf16 storage has no fallback kernels: the A/B switches that turn the row-streaming kernels off turn it off as well)
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("SMAAT_DW_ROWS", "1") == "0" or os.environ.get("SMAAT_UP_ROWS", "1") == "0",
                                 reason="row-streaming kernels switched off: mixed precision is not available")]
DEV = torch.device("cuda:0")
F32, BF16 = 0, 1


def P(t):
    return None if t is None else t.data_ptr()


def S():
    return torch.cuda.current_stream(DEV).cuda_stream


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)).to(DEV)


def r16(t):
    """values exactly representable in bf16, kept as f32"""
    return t.to(torch.bfloat16).float()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def same_as_rounded(got_bf16, ref_f32, what):
    """got must be RNE(ref) bit for bit"""
    exp = ref_f32.to(torch.bfloat16)
    bad = (got_bf16.view(torch.int16) != exp.view(torch.int16))
    # -0.0 vs +0.0 and NaN payloads do not occur here; report the worst element otherwise
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} elements differ from RNE(f32 result); " \
                                f"max abs {float((got_bf16.float() - exp.float()).abs().max()):.3e}"


def part_stats(part):
    pp = part.double()
    n = pp[2].sum(0)
    mean = (pp[2] * pp[0]).sum(0) / n.clamp(min=1)
    var = (pp[1] + pp[2] * (pp[0] - mean[None]) ** 2).sum(0) / n.clamp(min=1)
    return n, mean, var


# ------------------------------------------------------------------------------------------------------------------
def test_bf16_planes_layout():
    L = _lib.get()
    for R, C, tr in ((64, 24, 0), (130, 72, 0), (48, 200, 1), (7, 33, 1)):
        w = rnd(R * 7 + C, R, C)
        src = w.t().contiguous() if tr else w
        Cp = (C + 31) // 32 * 32
        out = torch.full((Cp // 16, R, 16), -1, dtype=torch.int16, device=DEV)
        assert L.smaat_bf16_planes(P(src), R, C, P(out), tr, S()) == 0
        torch.cuda.synchronize()
        exp = torch.zeros(R, Cp, dtype=torch.bfloat16, device=DEV)
        exp[:, :C] = w.to(torch.bfloat16)
        exp = exp.view(R, Cp // 16, 16).permute(1, 0, 2).contiguous()
        assert torch.equal(out, exp.view(torch.int16)), (R, C, tr)


PW_SHAPES = [
    # N, Cin, M, H, W
    (2, 24, 64, 16, 24),     # stem-like: partial contraction chunk, 64-row tile, P % 128 == 0
    (2, 128, 64, 32, 36),    # P % 128 != 0 tail tile
    (3, 256, 128, 18, 18),   # P % 8 != 0: the 4-byte LDS-DMA variant, 128-row tile
    (2, 72, 200, 36, 36),    # partial chunk + two channel tiles (one partial) + tail tile
    (1, 64, 24, 12, 12),     # fewer rows than a tile (stem data gradient)
    (2, 6, 21, 16, 16),      # VOC-like tiny contraction
    (2, 512, 512, 18, 18),   # deep level, many chunks, 4-byte variant
    (1, 2048, 512, 36, 36),  # longest contraction of the network
]


@pytest.mark.parametrize("shape", PW_SHAPES)
@pytest.mark.parametrize("out_dt", [BF16, F32])
def test_pointwise_fwd_bf16(shape, out_dt):
    L = _lib.get()
    N, Cin, M, H, W = shape
    p = H * W
    pad_c = 3  # x is a channel slice of a larger buffer (batch stride > Cin * P)
    xfull = rnd(1, N, Cin + pad_c + (1 if (Cin + pad_c) * p % 2 else 0), H, W).to(torch.bfloat16)
    x = xfull[:, pad_c:pad_c + Cin]
    x_bs = xfull.shape[1] * p
    w = rnd(2, M, Cin, scale=0.2)
    b = rnd(3, M)
    planes = torch.empty(((Cin + 31) // 32 * 2, M, 16), dtype=torch.int16, device=DEV)
    assert L.smaat_bf16_planes(P(w), M, Cin, P(planes), 0, S()) == 0
    out = torch.full((N, M, H, W), float("nan"), dtype=torch.bfloat16 if out_dt == BF16 else torch.float32, device=DEV)
    slots = L.smaat_pw_split_num_slots(N, H, W)
    part = torch.full((3, slots, M), float("nan"), device=DEV)
    rc = L.smaat_pointwise_fwd_bf16(x.data_ptr(), x_bs, P(planes), P(b), P(out), M * p, out_dt, P(part), N, Cin, M, H, W, 0, S())
    assert rc == 0
    torch.cuda.synchronize()
    ref = torch.einsum("mc,nchw->nmhw", r16(w).double(), x.double())
    refb = ref + b.double()[None, :, None, None]
    if out_dt == F32:
        assert rel(out, refb) < 2e-6
    else:
        # one bf16 rounding of the f32 result: |err| <= 2^-9 |value| (+ f32 accumulation noise)
        err = (out.double() - refb).abs()
        assert bool((err <= refb.abs() * 2.0 ** -8 + 1e-5).all()), float(err.max())
        assert rel(out, refb) < 3e-3
    # BatchNorm partials: statistics of the raw accumulators (without the bias)
    pn, pmean, pvar = part_stats(part)
    assert torch.equal(pn, torch.full_like(pn, N * p))
    assert rel(pmean, ref.mean((0, 2, 3))) < 1e-5
    assert rel(pvar, ref.var((0, 2, 3), unbiased=False)) < 1e-5
    # fused ReLU epilogue, no partials
    out2 = torch.empty_like(out)
    assert L.smaat_pointwise_fwd_bf16(x.data_ptr(), x_bs, P(planes), P(b), P(out2), M * p, out_dt, None, N, Cin, M, H, W, 1,
                                      S()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out2, out.clamp(min=0))


def test_pointwise_fwd_bf16_dgrad_transposed_planes():
    """dY = W^T dZ with the image of the transposed weight taken straight from pointwise.weight"""
    L = _lib.get()
    N, K, Cout, H, W = 2, 48, 72, 20, 20
    dz = rnd(1, N, Cout, H, W).to(torch.bfloat16)
    w = rnd(2, Cout, K, scale=0.2)  # pointwise.weight [Cout][K]
    planes = torch.empty(((Cout + 31) // 32 * 2, K, 16), dtype=torch.int16, device=DEV)
    assert L.smaat_bf16_planes(P(w), K, Cout, P(planes), 1, S()) == 0
    dy = torch.empty((N, K, H, W), dtype=torch.bfloat16, device=DEV)
    assert L.smaat_pointwise_fwd_bf16(P(dz), Cout * H * W, P(planes), None, P(dy), K * H * W, BF16, None, N, Cout, K, H, W, 0,
                                      S()) == 0
    torch.cuda.synchronize()
    ref = torch.einsum("mk,nmhw->nkhw", r16(w).double(), dz.double())
    assert rel(dy, ref) < 3e-3


WG_SHAPES = [
    # N, K (rows of y), M (rows of dz), H, W
    (2, 128, 64, 16, 24),
    (2, 24, 64, 32, 36),     # fewer y rows than the 128-row tile
    (3, 256, 128, 18, 18),   # 4-byte LDS-DMA variant, pixel tail inside a chunk
    (2, 200, 72, 36, 36),    # partial tiles on both sides, chunk tail (1296 = 40 * 32 + 16)
    (1, 1024, 512, 18, 18),
    (2, 6, 21, 16, 16),
]


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_pointwise_wgrad_bf16(shape):
    L = _lib.get()
    N, K, M, H, W = shape
    y = rnd(1, N, K, H, W).to(torch.bfloat16)
    dz = rnd(2, N, M, H, W, scale=0.1).to(torch.bfloat16)
    ns = L.smaat_wgrad_num_splits(N, H, W, M, K)
    ws = torch.full((ns, M, K), float("nan"), device=DEV)
    dw = torch.full((M, K), float("nan"), device=DEV)
    assert L.smaat_pointwise_wgrad_bf16(P(y), K * H * W, P(dz), M * H * W, P(ws), P(dw), N, K, M, H, W, S()) == 0
    torch.cuda.synchronize()
    ref = torch.einsum("nmhw,nkhw->mk", dz.double(), y.double())
    assert rel(dw, ref) < 2e-6


# ------------------------------------------------------------------------------------------------------------------
# typed streaming kernels against their f32 twins
# ------------------------------------------------------------------------------------------------------------------
DW_SHAPES = [(2, 6, 2, 16, 24), (2, 5, 1, 12, 16), (1, 3, 4, 8, 8), (3, 8, 2, 18, 18), (2, 4, 2, 36, 36), (1, 4, 2, 288, 288),
             (5, 6, 2, 18, 18)]  # (plane packing with a partial image group)


@pytest.mark.parametrize("shape", DW_SHAPES)
@pytest.mark.parametrize("x_dt", [F32, BF16])
@pytest.mark.parametrize("aff", [False, True])
def test_dw3x3_fwd_t(shape, x_dt, aff):
    L = _lib.get()
    N, Cin, kpl, H, W = shape
    K = Cin * kpl
    x32 = r16(rnd(1, N, Cin, H, W))
    w_dw, b_dw = rnd(2, K, 9, scale=0.3), rnd(3, K, scale=0.3)
    sc = torch.rand(Cin, device=DEV) + 0.5 if aff else None
    sh = rnd(7, Cin, scale=0.3) if aff else None
    yref = torch.empty(N, K, H, W, device=DEV)
    rc = L.smaat_dw3x3_fwd(P(x32), Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(yref), K * H * W, N, Cin, kpl, H, W, S())
    assert rc == 0
    x = x32.to(torch.bfloat16) if x_dt == BF16 else x32
    y = torch.empty(N, K, H, W, dtype=torch.bfloat16, device=DEV)
    rc = L.smaat_dw3x3_fwd_t(P(x), x_dt, Cin * H * W, P(sc), P(sh), P(w_dw), P(b_dw), P(y), BF16, K * H * W, N, Cin, kpl, H, W,
                             S())
    assert rc == 0
    torch.cuda.synchronize()
    same_as_rounded(y, yref, "dw3x3_fwd_t")


@pytest.mark.parametrize("shape", [s for s in DW_SHAPES if s[2] <= 2])
@pytest.mark.parametrize("combo", [(BF16, BF16, BF16), (F32, BF16, F32)])
def test_dw3x3_bwd_t(shape, combo):
    L = _lib.get()
    N, Cin, kpl, H, W = shape
    K = Cin * kpl
    x_dt, dy_dt, dx_dt = combo
    x32, dy32 = r16(rnd(1, N, Cin, H, W)), r16(rnd(2, N, K, H, W))
    w_dw = rnd(3, K, 9, scale=0.3)
    rows = L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W)

    def run(typed):
        ws = torch.empty(rows, K, 10, device=DEV)
        dw, db = torch.empty(K, 9, device=DEV), torch.empty(K, device=DEV)
        if not typed:
            dx = torch.empty(N, Cin, H, W, device=DEV)
            rc = L.smaat_dw3x3_bwd(P(x32), Cin * H * W, P(dy32), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws), P(dw), P(db), N,
                                   Cin, kpl, H, W, S())
        else:
            x = x32.to(torch.bfloat16) if x_dt == BF16 else x32
            dy = dy32.to(torch.bfloat16)
            dx = torch.empty(N, Cin, H, W, dtype=torch.bfloat16 if dx_dt == BF16 else torch.float32, device=DEV)
            rc = L.smaat_dw3x3_bwd_t(P(x), x_dt, Cin * H * W, None, None, P(dy), dy_dt, K * H * W, P(w_dw), P(dx), dx_dt,
                                     Cin * H * W, P(ws), P(dw), P(db), None, None, None, N, Cin, kpl, H, W, S())
        assert rc == 0
        torch.cuda.synchronize()
        return dx, dw, db

    dx0, dw0, db0 = run(False)
    dx1, dw1, db1 = run(True)
    if dx_dt == BF16:
        same_as_rounded(dx1, dx0, "dw3x3_bwd_t dx")
    else:
        assert torch.equal(dx1, dx0)
    assert torch.equal(dw1, dw0) and torch.equal(db1, db0)


@pytest.mark.parametrize("shape", [(2, 6, 2, 16, 24), (2, 4, 2, 36, 36), (1, 4, 1, 72, 72), (3, 8, 2, 18, 18)])
def test_dw3x3_bwd_t_bnred(shape):
    """the fused BatchNorm reduction of the second half (x = pre-BatchNorm tensor, activation on load)"""
    L = _lib.get()
    N, Cin, kpl, H, W = shape
    K = Cin * kpl
    z32, dy32 = r16(rnd(1, N, Cin, H, W)), r16(rnd(2, N, K, H, W))
    w_dw = rnd(3, K, 9, scale=0.3)
    sc, sh = torch.rand(Cin, device=DEV) + 0.5, rnd(4, Cin, scale=0.3)
    mean, invstd = rnd(5, Cin, scale=0.2), torch.rand(Cin, device=DEV) + 0.5
    rows = L.smaat_dw3x3_bwd_ws_rows(N, Cin, H, W)
    if not L.smaat_dw3x3_strip_ok(kpl, H, W):
        pytest.skip("shape not taken by the fused kernels")

    def run(typed):
        ws = torch.empty(rows, K, 10, device=DEV)
        dw, db = torch.empty(K, 9, device=DEV), torch.empty(K, device=DEV)
        rpart = torch.empty(2, rows - 1, Cin, device=DEV)
        if not typed:
            dx = torch.empty(N, Cin, H, W, device=DEV)
            rc = L.smaat_dw3x3_bwd_bnred(P(z32), Cin * H * W, P(sc), P(sh), P(dy32), K * H * W, P(w_dw), P(dx), Cin * H * W, P(ws),
                                         P(dw), P(db), P(mean), P(invstd), P(rpart), N, Cin, kpl, H, W, S())
        else:
            dx = torch.empty(N, Cin, H, W, dtype=torch.bfloat16, device=DEV)
            zb, dyb = z32.to(torch.bfloat16), dy32.to(torch.bfloat16)  # (named: a temporary would be freed before the launch)
            rc = L.smaat_dw3x3_bwd_t(P(zb), BF16, Cin * H * W, P(sc), P(sh), P(dyb), BF16,
                                     K * H * W, P(w_dw), P(dx), BF16, Cin * H * W, P(ws), P(dw), P(db), P(mean), P(invstd),
                                     P(rpart), N, Cin, kpl, H, W, S())
        assert rc == 0
        torch.cuda.synchronize()
        return dx, dw, db, rpart.sum(1)

    dx0, dw0, db0, r0 = run(False)
    dx1, dw1, db1, r1 = run(True)
    same_as_rounded(dx1, dx0, "bnred dx")
    assert torch.equal(dw1, dw0) and torch.equal(db1, db0)
    # the reduction is taken over dX as stored (bf16): compare with that sum formed from the f32 kernel's dX
    y = torch.relu(z32 * sc[None, :, None, None] + sh[None, :, None, None])
    g = torch.where(y > 0, dx1.float(), torch.zeros_like(y)).double()
    zh = ((z32 - mean[None, :, None, None]) * invstd[None, :, None, None]).double()
    assert rel(r1[0], g.sum((0, 2, 3))) < 1e-5
    assert rel(r1[1], (g * zh).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("shape", [(2, 7, 16, 24), (3, 5, 18, 18), (1, 3, 9, 11)])
def test_bn_kernels_t(shape):
    L = _lib.get()
    N, C, H, W = shape
    p = H * W
    z32, dy32 = r16(rnd(1, N, C, H, W)), r16(rnd(2, N, C, H, W))
    z, dy = z32.to(torch.bfloat16), dy32.to(torch.bfloat16)
    sc, sh = torch.rand(C, device=DEV) + 0.5, rnd(3, C, scale=0.3)
    mean, invstd = rnd(4, C, scale=0.2), torch.rand(C, device=DEV) + 0.5
    # affine_act
    y0 = torch.empty(N, C, H, W, device=DEV)
    assert L.smaat_affine_act(P(z32), C * p, P(sc), P(sh), P(y0), C * p, N, C, p, 1, S()) == 0
    y1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_affine_act_t(P(z), BF16, C * p, P(sc), P(sh), P(y1), BF16, C * p, N, C, p, 1, S()) == 0
    torch.cuda.synchronize()
    same_as_rounded(y1, y0, "affine_act_t")
    # reduce
    slots = L.smaat_plane_num_slots(N, p)
    p0, p1 = torch.empty(2, slots, C, device=DEV), torch.empty(2, slots, C, device=DEV)
    assert L.smaat_bn_bwd_reduce(P(dy32), C * p, P(z32), C * p, P(sc), P(sh), P(mean), P(invstd), P(p0), N, C, p, 1, S()) == 0
    assert L.smaat_bn_bwd_reduce_t(P(dy), BF16, C * p, P(z), BF16, C * p, P(sc), P(sh), P(mean), P(invstd), P(p1), N, C, p, 1,
                                   None, S()) == 0
    torch.cuda.synchronize()
    assert torch.equal(p0, p1)
    # apply
    coef = torch.stack([torch.rand(C, device=DEV) + 0.5, rnd(5, C, scale=0.1), rnd(6, C, scale=0.1)]).contiguous()
    d0 = torch.empty(N, C, H, W, device=DEV)
    assert L.smaat_bn_bwd_apply(P(dy32), C * p, P(z32), C * p, P(sc), P(sh), P(mean), P(invstd), P(coef), P(d0), C * p, N, C, p,
                                1, S()) == 0
    d1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_bn_bwd_apply_t(P(dy), BF16, C * p, P(z), BF16, C * p, P(sc), P(sh), P(mean), P(invstd), P(coef), P(d1), BF16,
                                  C * p, N, C, p, 1, None, S()) == 0
    torch.cuda.synchronize()
    same_as_rounded(d1, d0, "bn_bwd_apply_t")
    # head forms: f32 dlog, bf16 z / dz
    dlog, hw = rnd(7, N, 1, H, W), rnd(8, C)
    h0, h1 = torch.empty(3, slots, C, device=DEV), torch.empty(3, slots, C, device=DEV)
    assert L.smaat_bn_bwd_reduce_head(P(dlog), p, P(hw), P(z32), C * p, P(sc), P(sh), P(mean), P(invstd), P(h0), N, C, p, S()) == 0
    assert L.smaat_bn_bwd_reduce_t(P(dlog), F32, p, P(z), BF16, C * p, P(sc), P(sh), P(mean), P(invstd), P(h1), N, C, p, 1, P(hw),
                                   S()) == 0
    e0 = torch.empty(N, C, H, W, device=DEV)
    assert L.smaat_bn_bwd_apply_head(P(dlog), p, P(hw), P(z32), C * p, P(sc), P(sh), P(mean), P(invstd), P(coef), P(e0), C * p, N,
                                     C, p, S()) == 0
    e1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_bn_bwd_apply_t(P(dlog), F32, p, P(z), BF16, C * p, P(sc), P(sh), P(mean), P(invstd), P(coef), P(e1), BF16,
                                  C * p, N, C, p, 1, P(hw), S()) == 0
    torch.cuda.synchronize()
    assert torch.equal(h0, h1)
    same_as_rounded(e1, e0, "bn_bwd_apply_t head")
    # outconv1 + channel sum
    w1, b1 = rnd(9, C), rnd(10, 1)
    o0, o1 = torch.empty(N, 1, H, W, device=DEV), torch.empty(N, 1, H, W, device=DEV)
    assert L.smaat_outconv1_fwd(P(z32), C * p, P(sc), P(sh), P(w1), P(b1), P(o0), p, N, C, p, S()) == 0
    assert L.smaat_outconv1_fwd_t(P(z), BF16, C * p, P(sc), P(sh), P(w1), P(b1), P(o1), p, N, C, p, S()) == 0
    ws = torch.empty(slots, C, device=DEV)
    s0, s1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    assert L.smaat_channel_sum(P(z32), C * p, N, C, p, P(ws), P(s0), S()) == 0
    assert L.smaat_channel_sum_t(P(z), BF16, C * p, N, C, p, P(ws), P(s1), S()) == 0
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(s0, s1)


@pytest.mark.parametrize("shape", [(2, 5, 16, 24), (2, 3, 18, 18), (1, 4, 9, 11)])
def test_maxpool_t(shape):
    L = _lib.get()
    N, C, H, W = shape
    x32 = r16(rnd(1, N, C, H, W))
    Ho, Wo = H // 2, W // 2
    g32 = r16(rnd(2, N, C, Ho, Wo))
    xb, gb = x32.to(torch.bfloat16), g32.to(torch.bfloat16)
    y0 = torch.empty(N, C, Ho, Wo, device=DEV)
    y1 = torch.empty(N, C, Ho, Wo, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_maxpool2_fwd(P(x32), C * H * W, P(y0), C * Ho * Wo, N, C, H, W, S()) == 0
    assert L.smaat_maxpool2_fwd_t(P(xb), C * H * W, P(y1), C * Ho * Wo, N, C, H, W, BF16, S()) == 0
    d0 = torch.empty(N, C, H, W, device=DEV)
    d1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_maxpool2_bwd(P(x32), C * H * W, P(g32), C * Ho * Wo, P(d0), C * H * W, N, C, H, W, 0, S()) == 0
    assert L.smaat_maxpool2_bwd_t(P(xb), C * H * W, P(gb), C * Ho * Wo, P(d1), C * H * W, N, C, H, W, 0, BF16, S()) == 0
    torch.cuda.synchronize()
    same_as_rounded(y1, y0, "maxpool fwd")
    same_as_rounded(d1, d0, "maxpool bwd")


@pytest.mark.parametrize("shape", [(2, 5, 8, 12, 16, 24, 0, 0), (2, 3, 9, 10, 20, 24, 1, 0), (1, 4, 18, 18, 36, 36, 0, 0)])
def test_upsample_t(shape):
    L = _lib.get()
    N, C, H, W, Ho, Wo, pt, pl = shape
    x32 = r16(rnd(1, N, C, H, W))
    g32 = r16(rnd(2, N, C, Ho, Wo))
    xb, gb = x32.to(torch.bfloat16), g32.to(torch.bfloat16)
    o0 = torch.empty(N, C, Ho, Wo, device=DEV)
    o1 = torch.empty(N, C, Ho, Wo, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_upsample2x_fwd(P(x32), C * H * W, P(o0), C * Ho * Wo, N, C, H, W, Ho, Wo, pt, pl, S()) == 0
    assert L.smaat_upsample2x_fwd_t(P(xb), C * H * W, P(o1), C * Ho * Wo, N, C, H, W, Ho, Wo, pt, pl, BF16, S()) == 0
    d0 = torch.empty(N, C, H, W, device=DEV)
    d1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_upsample2x_bwd(P(g32), C * Ho * Wo, P(d0), C * H * W, N, C, H, W, Ho, Wo, pt, pl, S()) == 0
    rc = L.smaat_upsample2x_bwd_t(P(gb), C * Ho * Wo, P(d1), C * H * W, N, C, H, W, Ho, Wo, pt, pl, BF16, S())
    torch.cuda.synchronize()
    same_as_rounded(o1, o0, "upsample fwd")
    if rc == -2:
        pytest.skip("backward shape not taken by the row kernels (bf16 storage has no fallback)")
    assert rc == 0
    same_as_rounded(d1, d0, "upsample bwd")


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 16, 16, 24, 0), (3, 5, 9, 12, 2), (2, 64, 144, 144, 0), (1, 4, 36, 36, 3), (2, 3, 2, 4, 0)])
def test_cbam_chpool_pool_is_the_two_kernels(shape, dt):
    """channel pooling + MaxPool2d(2) in one pass == smaat_cbam_chpool_t followed by smaat_maxpool2_fwd_t: maxima, argmax, activation and
    pooled map bit for bit, the mean up to the f32 summation order (odd H: the last row is pooled over but belongs to no window; pad: a channel slice of a wider buffer)"""
    L = _lib.get()
    N, C, H, W, pad = shape
    p, Ho, Wo = H * W, H // 2, W // 2
    td = torch.bfloat16 if dt == BF16 else torch.float32
    xw = rnd(1, N, C + pad, H, W).to(td)
    x_bs = (C + pad) * p
    sc, sh = torch.rand(C, device=DEV) + 0.5, rnd(3, C, scale=0.3)
    for act in (False, True):
        a0, m0, i0 = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV), torch.empty(N, C, dtype=torch.int32, device=DEV)
        a1, m1, i1 = torch.empty_like(a0), torch.empty_like(m0), torch.empty_like(i0)
        y0 = torch.full((N, C, H, W), 7.0, dtype=td, device=DEV)
        y1 = torch.full((N, C, H, W), 9.0, dtype=td, device=DEV)
        q0 = torch.full((N, C, Ho, Wo), 7.0, dtype=td, device=DEV)
        q1 = torch.full((N, C, Ho, Wo), 9.0, dtype=td, device=DEV)
        if act:
            assert L.smaat_cbam_chpool_t(P(xw), x_bs, P(sc), P(sh), P(y0), C * p, N, C, p, P(a0), P(m0), P(i0), dt, S()) == 0
            assert L.smaat_maxpool2_fwd_t(P(y0), C * p, P(q0), C * Ho * Wo, N, C, H, W, dt, S()) == 0
            assert L.smaat_cbam_chpool_pool_t(P(xw), x_bs, P(sc), P(sh), P(y1), C * p, P(q1), C * Ho * Wo, N, C, H, W, P(a1),
                                              P(m1), P(i1), dt, S()) == 0
        else:
            assert L.smaat_cbam_chpool_t(P(xw), x_bs, None, None, None, 0, N, C, p, P(a0), P(m0), P(i0), dt, S()) == 0
            assert L.smaat_maxpool2_fwd_t(P(xw), x_bs, P(q0), C * Ho * Wo, N, C, H, W, dt, S()) == 0
            assert L.smaat_cbam_chpool_pool_t(P(xw), x_bs, None, None, None, 0, P(q1), C * Ho * Wo, N, C, H, W, P(a1), P(m1),
                                              P(i1), dt, S()) == 0
        torch.cuda.synchronize()
        assert torch.equal(m0, m1) and torch.equal(i0, i1), act
        # the mean is summed patch-major instead of row-major: same terms, another f32 summation order
        assert float((a0 - a1).abs().max()) <= 4e-6 * float(xw.float().abs().max()), act
        assert torch.equal(q0, q1), act
        if act:
            assert torch.equal(y0, y1)
    # a width the kernel does not take: the caller is told to run the two kernels
    xb = rnd(2, 1, 2, 6, 6).to(td)
    q = torch.empty(1, 2, 3, 3, dtype=td, device=DEV)
    assert L.smaat_cbam_chpool_pool_t(P(xb), 72, None, None, None, 0, P(q), 18, 1, 2, 6, 6, P(a1), P(m1), P(i1), dt, S()) == -2


@pytest.mark.parametrize("shape", [(2, 16, 16, 24), (2, 33, 18, 20), (1, 256, 72, 72), (1, 5, 2, 4), (2, 512, 6, 8)])
@pytest.mark.parametrize("pool", [True, False])
def test_cbam_three_pass_backward_t(shape, pool):
    """bf16 storage through the three-pass attention backward == the f32 kernels on the same (bf16-representable) values: dbn, the
    BatchNorm partials and both sets of ds partials bit for bit, dx = the f32 dx rounded once"""
    L = _lib.get()
    N, C, H, W = shape
    p, Ho, Wo = H * W, H // 2, W // 2
    x32 = r16(torch.relu(rnd(1, N, C, H, W)))
    dout32 = r16(rnd(4, N, C, H, W))
    dpool32 = r16(rnd(10, N, C, Ho, Wo))
    x, dout, dpool = x32.to(torch.bfloat16), dout32.to(torch.bfloat16), dpool32.to(torch.bfloat16)
    s = torch.rand(N, C, device=DEV) * 0.7 + 0.2
    gate = torch.rand(N, 1, H, W, device=DEV)
    conv = rnd(5, N, 1, H, W)
    mean, invstd = rnd(6, 1, scale=0.1), torch.rand(1, device=DEV) + 0.5
    dmaps = rnd(7, N, 2, H, W, scale=0.1)
    davg, dmx = rnd(8, N, C, scale=0.1), rnd(9, N, C, scale=0.1)
    amax = torch.randint(0, p, (N, C), dtype=torch.int32, device=DEV)
    nbp = L.smaat_cbam_pix_blocks(N, p)
    per = nbp // N
    out = []
    for dt, xx, gg, pp, td in ((F32, x32, dout32, dpool32, torch.float32), (BF16, x, dout, dpool, torch.bfloat16)):
        maps = torch.full((N, 2, H, W), float("nan"), device=DEV)
        amaxc = torch.full((N, H, W), -1, dtype=torch.int32, device=DEV)
        assert L.smaat_cbam_sppool_idx_t(P(xx), C * p, P(s), N, C, p, P(maps), P(amaxc), dt, S()) == 0
        dbn, pa = torch.empty(N, p, device=DEV), torch.empty(2, nbp, 1, device=DEV)
        dsp = torch.full((2 * per, N, C), float("nan"), device=DEV)
        dx = torch.full((N, C, H, W), float("nan"), dtype=td, device=DEV)
        assert L.smaat_cbam_bwd3_ok(P(xx), C * p, P(gg), C * p, P(pp) if pool else None, C * Ho * Wo if pool else 0, N, C, H, W, dt) == 1
        assert L.smaat_cbam_bwd_gate_ds_t(P(gg), C * p, P(xx), C * p, P(s), P(gate), P(conv), P(mean), P(invstd), N, C, p, P(dbn),
                                          P(pa), P(dsp), dt, S()) == 0
        assert L.smaat_cbam_bwd_ds2_t(P(xx), C * p, P(dmaps), P(amaxc), N, C, p, dsp.data_ptr() + 4 * per * N * C, dt, S()) == 0
        assert L.smaat_cbam_bwd_apply_t(P(gg), C * p, P(xx), C * p, P(s), P(gate), P(dmaps), P(amaxc), P(davg), P(dmx), P(amax),
                                        P(pp) if pool else None, C * Ho * Wo if pool else 0, N, C, H, W, P(dx), C * p, dt, S()) == 0
        out.append((dbn, pa, dsp, dx, maps, amaxc))
    torch.cuda.synchronize()
    (dbn0, pa0, dsp0, dx0, mp0, ix0), (dbn1, pa1, dsp1, dx1, mp1, ix1) = out
    assert torch.equal(mp0, mp1) and torch.equal(ix0, ix1) and int(ix0.min()) >= 0
    assert not bool(torch.isnan(dsp0).any()) and not bool(torch.isnan(dsp1).any())
    assert torch.equal(dbn0, dbn1) and torch.equal(pa0, pa1)
    assert torch.equal(dsp0, dsp1), (float((dsp0 - dsp1).abs().max()), float(dsp0.abs().max()))
    same_as_rounded(dx1, dx0, "cbam_bwd_apply_pool_t")


@pytest.mark.parametrize("shape", [(2, 16, 16, 24), (2, 32, 18, 18), (1, 8, 9, 11)])
def test_cbam_kernels_t(shape):
    L = _lib.get()
    N, C, H, W = shape
    p = H * W
    x32 = r16(rnd(1, N, C, H, W))
    x = x32.to(torch.bfloat16)
    # channel pooling, plain and with the activation applied on load
    sc, sh = torch.rand(C, device=DEV) + 0.5, rnd(3, C, scale=0.3)
    for act in (False, True):
        a0, m0, i0 = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV), torch.empty(N, C, dtype=torch.int32, device=DEV)
        a1, m1, i1 = torch.empty_like(a0), torch.empty_like(m0), torch.empty_like(i0)
        y1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
        if act:
            # f32 twin on the values the bf16 kernel pools over: the activation rounded to bf16
            y0 = torch.relu(torch.addcmul(sh[None, :, None, None], x32, sc[None, :, None, None]))  # fma(x, sc, sh)
            assert L.smaat_cbam_chpool_t(P(x), C * p, P(sc), P(sh), P(y1), C * p, N, C, p, P(a1), P(m1), P(i1), BF16, S()) == 0
            torch.cuda.synchronize()
            yr = y1.float()
            assert rel(yr, y0) < 3e-3
            assert L.smaat_cbam_chpool(P(yr), C * p, N, C, p, P(a0), P(m0), P(i0), S()) == 0
        else:
            assert L.smaat_cbam_chpool(P(x32), C * p, N, C, p, P(a0), P(m0), P(i0), S()) == 0
            assert L.smaat_cbam_chpool_t(P(x), C * p, None, None, None, 0, N, C, p, P(a1), P(m1), P(i1), BF16, S()) == 0
        torch.cuda.synchronize()
        assert torch.equal(a0, a1) and torch.equal(m0, m1) and torch.equal(i0, i1), act
    s = torch.rand(N, C, device=DEV)
    maps0, maps1 = torch.empty(N, 2, H, W, device=DEV), torch.empty(N, 2, H, W, device=DEV)
    assert L.smaat_cbam_sppool(P(x32), C * p, P(s), N, C, p, P(maps0), S()) == 0
    assert L.smaat_cbam_sppool_t(P(x), C * p, P(s), N, C, p, P(maps1), BF16, S()) == 0
    gate = torch.rand(N, 1, H, W, device=DEV)
    o0 = torch.empty(N, C, H, W, device=DEV)
    o1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    assert L.smaat_cbam_apply(P(x32), C * p, P(s), P(gate), P(o0), C * p, N, C, p, S()) == 0
    assert L.smaat_cbam_apply_t(P(x), C * p, P(s), P(gate), P(o1), C * p, N, C, p, BF16, S()) == 0
    torch.cuda.synchronize()
    assert torch.equal(maps0, maps1)
    same_as_rounded(o1, o0, "cbam_apply_t")
    # backward passes
    dout32 = r16(rnd(4, N, C, H, W))
    dout = dout32.to(torch.bfloat16)
    conv = rnd(5, N, 1, H, W)
    mean, invstd = rnd(6, 1, scale=0.1), torch.rand(1, device=DEV) + 0.5
    nbp = L.smaat_cbam_pix_blocks(N, p)
    dbn0, dbn1 = torch.empty(N, p, device=DEV), torch.empty(N, p, device=DEV)
    pa0, pa1 = torch.empty(2, nbp, 1, device=DEV), torch.empty(2, nbp, 1, device=DEV)
    assert L.smaat_cbam_bwd_gate(P(dout32), C * p, P(x32), C * p, P(s), P(gate), P(conv), P(mean), P(invstd), N, C, p, P(dbn0),
                                 P(pa0), S()) == 0
    assert L.smaat_cbam_bwd_gate_t(P(dout), C * p, P(x), C * p, P(s), P(gate), P(conv), P(mean), P(invstd), N, C, p, P(dbn1),
                                   P(pa1), BF16, S()) == 0
    dmaps = rnd(7, N, 2, H, W, scale=0.1)
    dx0 = torch.empty(N, C, H, W, device=DEV)
    dx1 = torch.empty(N, C, H, W, dtype=torch.bfloat16, device=DEV)
    ds0, ds1 = torch.empty(nbp, C, device=DEV), torch.empty(nbp, C, device=DEV)
    assert L.smaat_cbam_bwd_main(P(dout32), C * p, P(x32), C * p, P(s), P(gate), P(maps0), P(dmaps), N, C, p, P(dx0), C * p,
                                 P(ds0), S()) == 0
    assert L.smaat_cbam_bwd_main_t(P(dout), C * p, P(x), C * p, P(s), P(gate), P(maps0), P(dmaps), N, C, p, P(dx1), C * p,
                                   P(ds1), BF16, S()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dbn0, dbn1) and torch.equal(pa0, pa1)
    same_as_rounded(dx1, dx0, "cbam_bwd_main_t")
    assert torch.equal(ds0, ds1)
    # final passes (read-modify-write of dx): start both from the same bf16-representable dx
    davg, dmx = rnd(8, N, C, scale=0.1), rnd(9, N, C, scale=0.1)
    amax = torch.randint(0, p, (N, C), dtype=torch.int32, device=DEV)
    f0 = dx1.float().clone()
    f1 = dx1.clone()
    assert L.smaat_cbam_bwd_final(P(f0), C * p, P(davg), P(dmx), P(amax), N, C, p, S()) == 0
    assert L.smaat_cbam_bwd_final_t(P(f1), C * p, P(davg), P(dmx), P(amax), N, C, p, BF16, S()) == 0
    torch.cuda.synchronize()
    same_as_rounded(f1, f0, "cbam_bwd_final_t")
    if W % 4 == 0 and H >= 2:
        Ho, Wo = H // 2, W // 2
        dpool32 = r16(rnd(10, N, C, Ho, Wo))
        dpoolb = dpool32.to(torch.bfloat16)
        g0 = dx1.float().clone()
        g1 = dx1.clone()
        assert L.smaat_cbam_bwd_final_pool(P(g0), C * p, P(davg), P(dmx), P(amax), P(x32), C * p, P(dpool32), C * Ho * Wo, N, C, H,
                                           W, S()) == 0
        assert L.smaat_cbam_bwd_final_pool_t(P(g1), C * p, P(davg), P(dmx), P(amax), P(x), C * p, P(dpoolb),
                                             C * Ho * Wo, N, C, H, W, BF16, S()) == 0
        torch.cuda.synchronize()
        same_as_rounded(g1, g0, "cbam_bwd_final_pool_t")
