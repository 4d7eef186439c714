#!/usr/bin/env python
"""CPU study for the next round (no GPU): how accurate would a TWO-term fp16 operand split (3 MFMAs per product) be next to the
exact THREE-term bf16 split (6 MFMAs) the f32 GEMMs use now, and next to plain f32 accumulation?
Products are formed exactly (float64) from the rounded terms and accumulated in float32 in chunks of 16, as the MFMA does.
Operands: pointwise weights ~ N(0, 0.05); activations = ReLU(N(0,1)) (post-BatchNorm); gradients = heavy-tailed
(log-normal magnitudes over ~6 decades) at 1e-5 scale -- the case that decides whether fp16's 5-bit exponent is usable.
fp16 terms are scaled per tensor by a power of two so that max|x| = 2^14 (exact), results scaled back."""
import numpy as np

rng = np.random.default_rng(0)


def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16_3(x):
    p1 = bf16_trunc(x)
    r1 = (x - p1).astype(np.float32)
    p2 = bf16_trunc(r1)
    p3 = (r1 - p2).astype(np.float32)
    return [p1, p2, p3]


def split_f16_2(x):
    s = 2.0 ** (14 - np.ceil(np.log2(np.abs(x).max())))
    xs = (x * np.float32(s)).astype(np.float32)
    p1 = xs.astype(np.float16).astype(np.float32)
    p2 = (xs - p1).astype(np.float16).astype(np.float32)
    return [p1, p2], s


def acc_f32(terms_a, terms_b, pairs, K):
    """sum over the listed (i, j) term pairs of A_i @ B_j, float32 accumulation in chunks of 16 along K"""
    M, N = terms_a[0].shape[0], terms_b[0].shape[1]
    out = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        for i, j in pairs:
            out += (terms_a[i][:, k0:k0 + 16].astype(np.float64) @ terms_b[j][k0:k0 + 16].astype(np.float64)).astype(np.float32)
    return out


def study(name, a, b):
    K = a.shape[1]
    ref = a.astype(np.float64) @ b.astype(np.float64)
    nrm = np.linalg.norm(ref)
    f32 = acc_f32([a], [b], [(0, 0)], K)
    ta, tb = split_bf16_3(a), split_bf16_3(b)
    s6 = acc_f32(ta, tb, [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)], K)
    (fa, sa), (fb, sb) = split_f16_2(a), split_f16_2(b)
    h3 = acc_f32(fa, fb, [(1, 0), (0, 1), (0, 0)], K) / np.float32(sa * sb)
    h4 = acc_f32(fa, fb, [(1, 1), (1, 0), (0, 1), (0, 0)], K) / np.float32(sa * sb)
    e = lambda x: np.linalg.norm(x.astype(np.float64) - ref) / nrm  # noqa: E731
    # worst rows: relative error of each output row (a row = one output channel / one pixel set)
    w = lambda x: (np.linalg.norm(x.astype(np.float64) - ref, axis=1) / (np.linalg.norm(ref, axis=1) + 1e-300)).max()  # noqa: E731
    print(f"{name:34s} rel-L2: f32 {e(f32):.2e} | bf16x3 (6 MFMA) {e(s6):.2e} | f16x2 (3 MFMA) {e(h3):.2e} | f16x2 (4 MFMA) {e(h4):.2e}"
          f"   worst row: f32 {w(f32):.1e} bf16x3 {w(s6):.1e} f16x2/3 {w(h3):.1e}")
    # per-element view: error of every output element relative to the magnitude sum_k |a||b| of ITS OWN dot product (what an
    # f32 dot product guarantees, ~1e-7, however small the element is against the rest of the tensor)
    mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64) + 1e-300
    q = lambda x: np.quantile(np.abs(x.astype(np.float64) - ref) / mag, [0.5, 0.999, 1.0])  # noqa: E731
    for lab, x in (("f32", f32), ("bf16x3", s6), ("f16x2/3", h3)):
        m, hi, mx = q(x)
        print(f"      per-element |err| / sum|a||b|   {lab:8s} median {m:.1e}  99.9 % {hi:.1e}  max {mx:.1e}")


M, K, N = 64, 512, 2048
wgt = (rng.standard_normal((M, K)) * 0.05).astype(np.float32)
act = np.maximum(rng.standard_normal((K, N)), 0).astype(np.float32)
study("forward: W[64x512] . relu(N(0,1))", wgt, act)
grad = (np.exp(rng.standard_normal((K, N)) * 2.5) * 1e-5 * rng.choice([-1, 1], (K, N))).astype(np.float32)
study("dgrad: W^T . heavy-tailed 1e-5 dz", wgt, grad)
# weight gradient: contraction over pixels, both operands data-dependent; a few huge outliers in dz
dz = grad[:M, :].copy()
dz[0, :8] *= 1e4
a2 = dz                      # [M][N] pixels as contraction
b2 = act.T.copy()            # [N][K]
study("wgrad: dz (outliers x1e4) . act^T", a2, b2)
tiny = (grad * 1e-3).astype(np.float32)
tiny[:, 0] = 1.0             # one pixel 1e8 above the rest
study("dgrad: one column 1e8 above the rest", wgt, tiny)
