// HBM-bound spatial operators of the SmaAt-UNet path:
//   MaxPool2d(2)            reference models/unet_parts_depthwise_separable.py:48
//   Upsample(x2, bilinear, align_corners=True) + F.pad + cat slice write   :64, :78-85
//   depthwise 3x3 backward (dX, dW, db)   (nn.Conv2d groups=Cin, models/layers.py:38-44)
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------
// MaxPool 2x2 (floor mode).  grid: (N*C planes, segments of output pixels)
// ---------------------------------------------------------------------------------
// T: element type of x, y (and of dy, dx in the backward): f32 or bf16 storage
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool2_fwd(const T* __restrict__ x, long x_bs, T* __restrict__ y,
                                                      long y_bs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const T* xp = x + (long)n * x_bs + (long)c * H * W;
    T* yp = y + (long)n * y_bs + (long)c * Ho * Wo;
    const int Po = Ho * Wo;
    for (int o = blockIdx.y * 256 + threadIdx.x; o < Po; o += gridDim.y * 256) {
        const int i = o / Wo, j = o - i * Wo;
        const T* p = xp + (long)(2 * i) * W + 2 * j;
        const float2 a = ld2(p);  // 2*j even; row start parity handled below
        const float2 b2 = ld2(p + W);
        st1(yp + o, fmaxf(fmaxf(a.x, a.y), fmaxf(b2.x, b2.y)));
    }
}
// scalar variant for odd W (two-element loads would be misaligned)
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool2_fwd_s(const T* __restrict__ x, long x_bs, T* __restrict__ y,
                                                        long y_bs, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const T* xp = x + (long)n * x_bs + (long)c * H * W;
    T* yp = y + (long)n * y_bs + (long)c * Ho * Wo;
    const int Po = Ho * Wo;
    for (int o = blockIdx.y * 256 + threadIdx.x; o < Po; o += gridDim.y * 256) {
        const int i = o / Wo, j = o - i * Wo;
        const T* p = xp + (long)(2 * i) * W + 2 * j;
        st1(yp + o, fmaxf(fmaxf(ld1(p), ld1(p + 1)), fmaxf(ld1(p + W), ld1(p + W + 1))));
    }
}

// dx: gradient goes to the FIRST maximum in scan order (0,0),(0,1),(1,0),(1,1); rows/cols
// dropped by floor mode get zero.  One thread per INPUT pixel pair row -> full coverage.
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool2_bwd(const T* __restrict__ x, long x_bs,
                                                      const T* __restrict__ dy, long dy_bs,
                                                      T* __restrict__ dx, long dx_bs, int C, int H, int W,
                                                      int accum) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const T* xp = x + (long)n * x_bs + (long)c * H * W;
    const T* gp = dy + (long)n * dy_bs + (long)c * Ho * Wo;
    T* dp = dx + (long)n * dx_bs + (long)c * H * W;
    const int P = H * W;
    for (int p = blockIdx.y * 256 + threadIdx.x; p < P; p += gridDim.y * 256) {
        const int r = p / W, cc = p - r * W;
        const int i = r >> 1, j = cc >> 1;
        float g = 0.f;
        if (i < Ho && j < Wo) {
            const T* q = xp + (long)(2 * i) * W + 2 * j;
            const float v0 = ld1(q), v1 = ld1(q + 1), v2 = ld1(q + W), v3 = ld1(q + W + 1);
            int am = 0;
            float m = v0;
            if (v1 > m) { m = v1; am = 1; }
            if (v2 > m) { m = v2; am = 2; }
            if (v3 > m) { m = v3; am = 3; }
            const int me = ((r & 1) << 1) | (cc & 1);
            if (me == am) g = ld1(gp + i * Wo + j);
        }
        st1(dp + p, accum ? ld1(dp + p) + g : g);
    }
}

// ---------------------------------------------------------------------------------
// bilinear x2, align_corners=True, written into a (possibly padded) slice of a cat buffer
// out plane is Ho x Wo (skip-connection size); the 2H x 2W image sits at (pad_t, pad_l);
// the rest of the plane is zero (F.pad).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void ac_coef(int o, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
    const float src = (float)o * scale;
    i0 = (int)floorf(src);
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void k_upsample2x_fwd(const float* __restrict__ x, long x_bs,
                                                        float* __restrict__ out, long out_bs, int C, int H, int W,
                                                        int Ho, int Wo, int pad_t, int pad_l) {
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float* xp = x + (long)n * x_bs + (long)c * H * W;
    float* op = out + (long)n * out_bs + (long)c * Ho * Wo;
    const int H2 = 2 * H, W2 = 2 * W;
    const float sh = H2 > 1 ? (float)(H - 1) / (float)(H2 - 1) : 0.f;
    const float sw = W2 > 1 ? (float)(W - 1) / (float)(W2 - 1) : 0.f;
    const int Po = Ho * Wo;
    for (int o = blockIdx.y * 256 + threadIdx.x; o < Po; o += gridDim.y * 256) {
        const int r = o / Wo, cc = o - r * Wo;
        const int ur = r - pad_t, uc = cc - pad_l;
        float v = 0.f;
        if (ur >= 0 && ur < H2 && uc >= 0 && uc < W2) {
            int r0, r1, c0, c1;
            float a0, a1, b0, b1;
            ac_coef(ur, sh, H, r0, r1, a0, a1);
            ac_coef(uc, sw, W, c0, c1, b0, b1);
            const float tl = xp[r0 * W + c0], tr = xp[r0 * W + c1];
            const float bl = xp[r1 * W + c0], br = xp[r1 * W + c1];
            v = bilerp_v(a0, bilerp_h(b0, tl, b1, tr), a1, bilerp_h(b0, bl, b1, br));
        }
        op[o] = v;
    }
}

// gather form of the transpose (deterministic): each input pixel sums the output pixels
// that referenced it, recomputing the forward coefficients exactly.
__global__ __launch_bounds__(256) void k_upsample2x_bwd(const float* __restrict__ dout, long dout_bs,
                                                        float* __restrict__ dx, long dx_bs, int C, int H, int W,
                                                        int Ho, int Wo, int pad_t, int pad_l) {
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float* gp = dout + (long)n * dout_bs + (long)c * Ho * Wo;
    float* dp = dx + (long)n * dx_bs + (long)c * H * W;
    const int H2 = 2 * H, W2 = 2 * W;
    const float sh = H2 > 1 ? (float)(H - 1) / (float)(H2 - 1) : 0.f;
    const float sw = W2 > 1 ? (float)(W - 1) / (float)(W2 - 1) : 0.f;
    const int P = H * W;
    for (int p = blockIdx.y * 256 + threadIdx.x; p < P; p += gridDim.y * 256) {
        const int h = p / W, w = p - h * W;
        // candidate output rows: src in (h-1, h+1)  ->  o in [2h-2, 2h+3] is a safe superset
        int olo = 2 * h - 2, ohi = 2 * h + 3;
        if (olo < 0) olo = 0;
        if (ohi > H2 - 1) ohi = H2 - 1;
        int plo = 2 * w - 2, phi = 2 * w + 3;
        if (plo < 0) plo = 0;
        if (phi > W2 - 1) phi = W2 - 1;
        // column taps of this input pixel, computed once (not once per candidate row)
        float wcv[6];
        int gcv[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int ocol = plo + t;
            int c0, c1;
            float b0, b1;
            ac_coef(ocol < W2 ? ocol : W2 - 1, sw, W, c0, c1, b0, b1);
            float wc = 0.f;
            if (c0 == w) wc += b0;
            if (c1 == w) wc += b1;
            const int gc = ocol + pad_l;
            const bool v = ocol <= phi && gc >= 0 && gc < Wo;
            wcv[t] = v ? wc : 0.f;
            gcv[t] = v ? gc : 0;
        }
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int orow = olo + u;
            int r0, r1;
            float a0, a1;
            ac_coef(orow < H2 ? orow : H2 - 1, sh, H, r0, r1, a0, a1);
            float wr = 0.f;
            if (r0 == h) wr += a0;
            if (r1 == h) wr += a1;
            const int gr = orow + pad_t;
            const bool rv = orow <= ohi && gr >= 0 && gr < Ho;
            wr = rv ? wr : 0.f;
            if (wr != 0.f) {
                const float* grow = gp + (long)gr * Wo;
                float rowacc = 0.f;
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    if (wcv[t] != 0.f) rowacc = fmaf(wcv[t], grow[gcv[t]], rowacc);
                acc = fmaf(wr, rowacc, acc);
            }
        }
        dp[p] = acc;
    }
}

// ---------------------------------------------------------------------------------
// ConvTranspose2d(C, Co, kernel_size=2, stride=2) = a pointwise GEMM with 4 Co rows (row (a*2+b)*Co + co holds
// w[:, co, a, b]) followed by this 2x2 pixel shuffle (reference UpDS with bilinear=False,
// models/unet_parts_depthwise_separable.py:72-73).  Forward: the shuffled image + bias is written at (pad_t, pad_l) of an
// [Ho][Wo] plane of the concatenation buffer, zeros elsewhere (F.pad, :78-81).  Backward: the inverse gather.
//   out[n][co][2i+a+pad_t][2j+b+pad_l] = t[n][(a*2+b)*Co + co][i][j] + bias[co]
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pixel_shuffle2_fwd(const float* __restrict__ t, long t_bs,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            long out_bs, int Co, int H, int W, int Ho, int Wo, int pad_t,
                                                            int pad_l) {
    const int plane = blockIdx.x, n = plane / Co, co = plane - n * Co;
    const float* tp = t + (long)n * t_bs;
    float* op = out + (long)n * out_bs + (long)co * Ho * Wo;
    const float bv = bias ? bias[co] : 0.f;
    const int P = H * W, Po = Ho * Wo;
    for (int o = blockIdx.y * 256 + threadIdx.x; o < Po; o += gridDim.y * 256) {
        const int r = o / Wo, c = o - r * Wo;
        const int ur = r - pad_t, uc = c - pad_l;
        float v = 0.f;
        if (ur >= 0 && ur < 2 * H && uc >= 0 && uc < 2 * W) {
            const int ab = ((ur & 1) << 1) | (uc & 1);
            v = tp[(long)(ab * Co + co) * P + (ur >> 1) * W + (uc >> 1)] + bv;
        }
        op[o] = v;
    }
}

__global__ __launch_bounds__(256) void k_pixel_shuffle2_bwd(const float* __restrict__ dout, long dout_bs,
                                                            float* __restrict__ dt, long dt_bs, int Co, int H, int W,
                                                            int Ho, int Wo, int pad_t, int pad_l) {
    const int plane = blockIdx.x, n = plane / (4 * Co), m = plane - n * (4 * Co);
    const int ab = m / Co, co = m - ab * Co, a = ab >> 1, b = ab & 1;
    const float* gp = dout + (long)n * dout_bs + (long)co * Ho * Wo;
    float* dp = dt + (long)n * dt_bs + (long)m * H * W;
    const int P = H * W;
    for (int p = blockIdx.y * 256 + threadIdx.x; p < P; p += gridDim.y * 256) {
        const int i = p / W, j = p - i * W;
        const int r = 2 * i + a + pad_t, c = 2 * j + b + pad_l;
        dp[p] = (r >= 0 && r < Ho && c >= 0 && c < Wo) ? gp[r * Wo + c] : 0.f;
    }
}

// ---------------------------------------------------------------------------------
// bilinear x2 (align_corners) backward, separable two-pass form for W <= 256:
//   T[o][w]  = sum_p wc(w, p) * g[o][p]      (column contraction, <= 4 non-zero taps, per output row)
//   dx[h][w] = sum_o wr(h, o) * T[o][w]      (row contraction from LDS)
// A thread owns one input column (its 6 candidate column taps are computed once); a block covers
// `PB` planes x `UTH` input rows.  Same arithmetic (coefficients recomputed exactly as the forward) as
// the gather kernel above, ~5x fewer instructions per pixel.
// ---------------------------------------------------------------------------------
#define UTH 8
#define UROWS (2 * UTH + 6)

__global__ __launch_bounds__(256) void k_upsample2x_bwd_sep(const float* __restrict__ dout, long dout_bs,
                                                            float* __restrict__ dx, long dx_bs, int NC, int C,
                                                            int H, int W, int Ho, int Wo, int pad_t, int pad_l,
                                                            int PB) {
    extern __shared__ float usm[];
    float* Tl = usm;                       // [PB][UROWS][W]
    float* wrow = usm + PB * UROWS * W;    // [UTH][6]
    const int tid = threadIdx.x;
    const int pl = tid / W, w = tid - pl * W;
    const int plane = blockIdx.x * PB + pl;
    const bool act = pl < PB && plane < NC;
    const int n = act ? plane / C : 0, c = act ? plane - n * C : 0;
    const float* gp = dout + (long)n * dout_bs + (long)c * Ho * Wo;
    float* dp = dx + (long)n * dx_bs + (long)c * H * W;
    const int H2 = 2 * H, W2 = 2 * W;
    const float sh = H2 > 1 ? (float)(H - 1) / (float)(H2 - 1) : 0.f;
    const float sw = W2 > 1 ? (float)(W - 1) / (float)(W2 - 1) : 0.f;
    // column taps of this thread's input column
    float wcv[6];
    int gcv[6];
    {
        int plo = 2 * w - 2, phi = 2 * w + 3;
        if (plo < 0) plo = 0;
        if (phi > W2 - 1) phi = W2 - 1;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int ocol = plo + t;
            int c0, c1;
            float b0, b1;
            ac_coef(ocol < W2 ? ocol : W2 - 1, sw, W, c0, c1, b0, b1);
            float wc = 0.f;
            if (c0 == w) wc += b0;
            if (c1 == w) wc += b1;
            const int gc = ocol + pad_l;
            const bool v = ocol <= phi && gc >= 0 && gc < Wo;
            wcv[t] = v ? wc : 0.f;
            gcv[t] = v ? gc : 0;
        }
    }
    for (int h0 = blockIdx.y * UTH; h0 < H; h0 += gridDim.y * UTH) {
        int olo = 2 * h0 - 2;
        if (olo < 0) olo = 0;
        __syncthreads();  // previous tile's pass 2 is done with Tl / wrow
        if (tid < UTH * 6) {  // row taps: wrow[i][u] pairs with output row (2*(h0+i) - 2 clamped) + u
            const int i = tid / 6, u = tid - i * 6;
            const int h = h0 + i;
            int lo = 2 * h - 2, hi = 2 * h + 3;
            if (lo < 0) lo = 0;
            if (hi > H2 - 1) hi = H2 - 1;
            const int orow = lo + u;
            int r0, r1;
            float a0, a1;
            ac_coef(orow < H2 ? orow : H2 - 1, sh, H, r0, r1, a0, a1);
            float wr = 0.f;
            if (r0 == h) wr += a0;
            if (r1 == h) wr += a1;
            const int gr = orow + pad_t;
            wrow[tid] = (h < H && orow <= hi && gr >= 0 && gr < Ho) ? wr : 0.f;
        }
        if (act) {  // pass 1: UROWS output rows starting at olo
#pragma unroll 2
            for (int rr = 0; rr < UROWS; ++rr) {
                const int orow = olo + rr;
                const int gr = orow + pad_t;
                float acc = 0.f;
                if (orow < H2 && gr >= 0 && gr < Ho) {
                    const float* grow = gp + (long)gr * Wo;
#pragma unroll
                    for (int t = 0; t < 6; ++t) acc = fmaf(wcv[t], grow[gcv[t]], acc);
                }
                Tl[(pl * UROWS + rr) * W + w] = acc;
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < UTH; ++i) {
                const int h = h0 + i;
                if (h < H) {
                    int lo = 2 * h - 2;
                    if (lo < 0) lo = 0;
                    const int rbase = lo - olo;  // 0 .. 2*UTH
                    float acc = 0.f;
#pragma unroll
                    for (int u = 0; u < 6; ++u) acc = fmaf(wrow[i * 6 + u], Tl[(pl * UROWS + rbase + u) * W + w], acc);
                    dp[h * W + w] = acc;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// depthwise 3x3 backward.  One block per (n, input channel); loops over the pixel tiles
// of the plane.  dY is staged with a halo in LDS; every dY neighbour read feeds BOTH the
// data gradient (x w) and the 9-tap weight gradient (x x[centre]).
//   dX[ci][q]      = sum_j sum_tap w[ci*kpl+j][tap] * dY[ci*kpl+j][q - tap]
//   dW[o][tap]     = sum_q x[ci][q] * dY[o][q - tap]
//   db[o]          = sum_q dY[o][q]
// part: [N][Cdw][10]  (9 taps + bias), finished by k_reduce_rows over N.
// ---------------------------------------------------------------------------------
#define DWB_SMAX 768
#define DWB_KPL_MAX 4

// TX / TG / TD: element types of x, dY, dX (f32 | bf16 storage)
template <typename TX, typename TG, typename TD>
__global__ __launch_bounds__(256) void k_dw3x3_bwd(const TX* __restrict__ x, long x_bs,
                                                   const TG* __restrict__ dy, long dy_bs,
                                                   const float* __restrict__ w_dw, TD* __restrict__ dx,
                                                   long dx_bs, float* __restrict__ part, int Cin, int kpl,
                                                   TileGeom g) {
    // grid: (N*Cin planes, tile groups).  part: [N*groups][Cdw][10]
    __shared__ float S[DWB_KPL_MAX * DWB_SMAX];
    __shared__ float red[4 * 10 * DWB_KPL_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int plane = blockIdx.x, n = plane / Cin, ci = plane - n * Cin;
    const int Cdw = Cin * kpl;
    const TX* xp = x + (long)n * x_bs + (long)ci * g.P;
    const TG* dyp = dy + (long)n * dy_bs + (long)(ci * kpl) * g.P;
    TD* dxp = dx ? dx + (long)n * dx_bs + (long)ci * g.P : nullptr;

    float accw[DWB_KPL_MAX][10];
#pragma unroll
    for (int j = 0; j < DWB_KPL_MAX; ++j)
#pragma unroll
        for (int t = 0; t < 10; ++t) accw[j][t] = 0.f;

    // per-thread constants of the 2D tile mode (identical for every tile)
    int sr2[3], sc2[3], tr2 = 0, tc2 = 0;
    if (g.mode == 1) {
        const int SW = g.TW + 2;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int e = tid + 256 * jj;
            sr2[jj] = e / SW;
            sc2[jj] = e - sr2[jj] * SW;
        }
        tr2 = tid / g.TW;
        tc2 = tid - tr2 * g.TW;
    }

    // software pipeline over the tiles of this group: the loads of tile t+1 (dY halo patch and
    // the x centre value) are in flight while tile t is being computed.
    float sv[3][DWB_KPL_MAX];
    bool sin[3];
    float xnext = 0.f;
    int pnext = 0;
    bool pvnext = false;
    auto prefetch = [&](int tl) {
        const StageRegion rg = stage_region(g, tl);
        const int rsize = rg.nrows * rg.SW;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int e = tid + 256 * jj;
            int sr, sc;
            if (g.mode == 1) {
                sr = sr2[jj];
                sc = sc2[jj];
            } else {
                sr = e / rg.SW;
                sc = e - sr * rg.SW;
            }
            const int gr = rg.row_lo + sr, gc = rg.col_lo + sc;
            sin[jj] = (e < rsize) && (gr >= 0 && gr < g.H && gc >= 0 && gc < g.W);
            const int go = sin[jj] ? gr * g.W + gc : 0;
#pragma unroll
            for (int j = 0; j < DWB_KPL_MAX; ++j) sv[jj][j] = ld1(dyp + (long)(j < kpl ? j : 0) * g.P + go);
        }
        int r, c;
        if (g.mode == 1) {
            r = rg.row_lo + 1 + tr2;
            c = rg.col_lo + 1 + tc2;
            pvnext = (r < g.H) && (c < g.W);
        } else {
            pvnext = tile_pixel(g, tl, tid, r, c);
        }
        pnext = pvnext ? r * g.W + c : 0;
        xnext = ld1(xp + pnext);
    };

    int tl = blockIdx.y;
    if (tl < g.tiles_per_img) prefetch(tl);
    for (; tl < g.tiles_per_img; tl += gridDim.y) {
        const StageRegion rg = stage_region(g, tl);
        const int rsize = rg.nrows * rg.SW;
        __syncthreads();  // previous tile's reads are done
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int e = tid + 256 * jj;
            if (e < rsize) {
#pragma unroll
                for (int j = 0; j < DWB_KPL_MAX; ++j)
                    if (j < kpl) S[j * DWB_SMAX + e] = sin[jj] ? sv[jj][j] : 0.f;
            }
        }
        const bool pv = pvnext;
        const int po = pnext;
        const float xv = xnext;
        __syncthreads();
        {
            const int tn = tl + gridDim.y;
            prefetch(tn < g.tiles_per_img ? tn : tl);
        }
        if (pv) {
            const int r = po / g.W, c = po - r * g.W;
            const int sb = (r - rg.row_lo) * rg.SW + (c - rg.col_lo);
            const int SW = rg.SW;
            float dxa = 0.f;
#pragma unroll
            for (int j = 0; j < DWB_KPL_MAX; ++j) {
                if (j < kpl) {
                    const float* sp = S + j * DWB_SMAX + sb;
                    const float* w = w_dw + (ci * kpl + j) * 9;
                    // tap (dr,dc) pairs with dY at q - (dr,dc)
                    const float d00 = sp[SW + 1], d01 = sp[SW], d02 = sp[SW - 1];
                    const float d10 = sp[1], d11 = sp[0], d12 = sp[-1];
                    const float d20 = sp[-SW + 1], d21 = sp[-SW], d22 = sp[-SW - 1];
                    dxa = fmaf(w[0], d00, dxa);
                    dxa = fmaf(w[1], d01, dxa);
                    dxa = fmaf(w[2], d02, dxa);
                    dxa = fmaf(w[3], d10, dxa);
                    dxa = fmaf(w[4], d11, dxa);
                    dxa = fmaf(w[5], d12, dxa);
                    dxa = fmaf(w[6], d20, dxa);
                    dxa = fmaf(w[7], d21, dxa);
                    dxa = fmaf(w[8], d22, dxa);
                    accw[j][0] = fmaf(xv, d00, accw[j][0]);
                    accw[j][1] = fmaf(xv, d01, accw[j][1]);
                    accw[j][2] = fmaf(xv, d02, accw[j][2]);
                    accw[j][3] = fmaf(xv, d10, accw[j][3]);
                    accw[j][4] = fmaf(xv, d11, accw[j][4]);
                    accw[j][5] = fmaf(xv, d12, accw[j][5]);
                    accw[j][6] = fmaf(xv, d20, accw[j][6]);
                    accw[j][7] = fmaf(xv, d21, accw[j][7]);
                    accw[j][8] = fmaf(xv, d22, accw[j][8]);
                    accw[j][9] += d11;
                }
            }
            if (dxp) st1(dxp + po, dxa);
        }
    }
    // block reduction of the (kpl x 10) accumulators
#pragma unroll
    for (int j = 0; j < DWB_KPL_MAX; ++j) {
        if (j < kpl) {
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const float v = wave_sum_l63(accw[j][t]);
                if (lane == 63) red[(wave * DWB_KPL_MAX + j) * 10 + t] = v;
            }
        }
    }
    __syncthreads();
    if (tid < kpl * 10) {
        const int j = tid / 10, t = tid - j * 10;
        const float v = red[(0 * DWB_KPL_MAX + j) * 10 + t] + red[(1 * DWB_KPL_MAX + j) * 10 + t] +
                        red[(2 * DWB_KPL_MAX + j) * 10 + t] + red[(3 * DWB_KPL_MAX + j) * 10 + t];
        part[(((long)n * gridDim.y + blockIdx.y) * Cdw + ci * kpl + j) * 10 + t] = v;
    }
}

// ---------------------------------------------------------------------------------
// depthwise 3x3 backward, strip form (W % 4 == 0, 16-B aligned planes).
// The kernel above spends its time issuing instructions (one pixel per thread: 9 LDS reads and
// 18 FMAs per pixel and output channel, dword staging).  Here
//   * the dY halo tile is staged as aligned float4 columns [c0-4, c0+TW+4) x rows [r0-1, r0+TH]
//     (one global_load_dwordx4 + one ds_write_b128 per 4 floats);
//   * a thread owns a STRIP of 4 vertically adjacent pixels: 18 LDS reads (6 rows x 3 columns)
//     per output channel feed 4 pixels of dX and the 9+1 weight/bias accumulators.
// Same partial-result layout as k_dw3x3_bwd: part[(n*groups + group)][Cdw][10].
// ---------------------------------------------------------------------------------
struct DwbGeom {
    int H, W, P, TH, TW, tiles_x, tiles, stride, nrow, ncol4, ssz;
};

#ifndef DWB_MINW
#define DWB_MINW 1
#endif
template <int KPL>
__global__ __launch_bounds__(256, DWB_MINW) void k_dw3x3_bwd_strip(const float* __restrict__ x, long x_bs,
                                                         const float* __restrict__ dy, long dy_bs,
                                                         const float* __restrict__ w_dw, float* __restrict__ dx,
                                                         long dx_bs, float* __restrict__ part, int Cin,
                                                         const DwbGeom g, const float* __restrict__ bn_g,
                                                         const float* __restrict__ bn_b, float* __restrict__ rpart,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift) {
    // rpart (nullable; needs in_scale/in_shift, i.e. x = the PRE-BatchNorm tensor z of the previous half-block):
    // the kernel also emits that BatchNorm's backward reduction over its planes,
    //   rpart[0][row][ci] = sum g,  rpart[1][row][ci] = sum g * zhat,   g = dX * [y > 0],  zhat = (z - mean) * invstd
    // with (bn_g, bn_b) = (mean, invstd) of that BatchNorm -- the same expression ATen evaluates, valid for any
    // gamma (0 and tiny values included) -- so the separate pass over (dy, z) of smaat_bn_bwd_reduce is not needed.
    constexpr int NSL = 6;  // float4 staging slots per thread: KPL * nrow * ncol4 <= 1536
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* S = dsm;                   // [KPL][ssz]
    float* red = dsm + KPL * g.ssz;   // [4][KPL][10]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int plane = blockIdx.x, n = plane / Cin, ci = plane - n * Cin;
    const int Cdw = Cin * KPL;
    const float* xp = x + (long)n * x_bs + (long)ci * g.P;
    const float* dyp = dy + (long)n * dy_bs + (long)(ci * KPL) * g.P;
    float* dxp = dx ? dx + (long)n * dx_bs + (long)ci * g.P : nullptr;

    float accw[KPL][10];
#pragma unroll
    for (int j = 0; j < KPL; ++j)
#pragma unroll
        for (int t = 0; t < 10; ++t) accw[j][t] = 0.f;
    float wt[KPL][9];
#pragma unroll
    for (int j = 0; j < KPL; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[j][t] = w_dw[(ci * KPL + j) * 9 + t];
    float r1 = 0.f, r2 = 0.f;
    // in_scale/in_shift (nullable): x holds the PRE-BatchNorm tensor z of the previous half-block and the
    // activation is recomputed on load, y = relu(z * in_scale[ci] + in_shift[ci])
    const bool aff = in_scale != nullptr;
    const float asc = aff ? in_scale[ci] : 1.f, ash = aff ? in_shift[ci] : 0.f;
    const float rmean = rpart ? bn_g[ci] : 0.f;
    const float rinvstd = rpart ? bn_b[ci] : 0.f;

    // tile-independent thread constants
    const int per = g.nrow * g.ncol4, F = KPL * per;
    int s_j[NSL], s_rr[NSL], s_q[NSL], s_lo[NSL];
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int f = (tid + 256 * k) % F;  // surplus slots re-stage a valid element
        const int jj = f / per, rem = f - jj * per;
        s_j[k] = jj;
        s_rr[k] = rem / g.ncol4;
        s_q[k] = rem - s_rr[k] * g.ncol4;
        s_lo[k] = jj * g.ssz + s_rr[k] * g.stride + 4 * s_q[k];
    }
    const int nstrips = (g.TH >> 2) * g.TW;
    const int nit = (nstrips + 255) >> 8;  // strips per thread (tiles are as wide as the plane when it fits:
                                           // every staged row is a run of full cache lines)
    float4 sv[NSL];
    bool sok[NSL];
    auto prefetch = [&](int tl) {
        const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
        const int r0 = ty * g.TH, c0 = tx * g.TW;
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int gr = r0 - 1 + s_rr[k], gc = c0 - 4 + 4 * s_q[k];
            sok[k] = gr >= 0 && gr < g.H && gc >= 0 && gc < g.W;
            sv[k] = *(const float4*)(dyp + (long)s_j[k] * g.P + (sok[k] ? gr * g.W + gc : 0));
        }
    };

    int tl = blockIdx.y;
    if (tl < g.tiles) prefetch(tl);
    for (; tl < g.tiles; tl += gridDim.y) {
        const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
        const int r0 = ty * g.TH, c0 = tx * g.TW;
        __syncthreads();  // previous tile's reads are done
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            float4 v = sv[k];
            v.x = sok[k] ? v.x : 0.f;
            v.y = sok[k] ? v.y : 0.f;
            v.z = sok[k] ? v.z : 0.f;
            v.w = sok[k] ? v.w : 0.f;
            *(float4*)(S + s_lo[k]) = v;
        }
        __syncthreads();
        {
            const int tn = tl + gridDim.y;
            prefetch(tn < g.tiles ? tn : tl);  // in flight during the compute below
        }
#pragma unroll 1
        for (int it = 0; it < nit; ++it) {
            const int sidx = tid + (it << 8);
            if (sidx < nstrips) {
                const int srg = sidx / g.TW, sc = sidx - srg * g.TW;
                const int sb = (srg * 4) * g.stride + sc + 3;
                const int prow = r0 + srg * 4;
                const bool pcol = (c0 + sc) < g.W;
                const int po = prow * g.W + c0 + sc;
                float xc[4], zr[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ok[i] = pcol && (prow + i) < g.H;
                    zr[i] = xp[ok[i] ? po + i * g.W : 0];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = zr[i];
                    if (aff) v = fmaxf(fmaf(v, asc, ash), 0.f);
                    xc[i] = ok[i] ? v : 0.f;
                }
                float dxa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const float* sp = S + j * g.ssz + sb;
                    float d[6][3];
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
                        for (int dc = 0; dc < 3; ++dc) d[rr][dc] = sp[rr * g.stride + dc];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                            for (int tc = 0; tc < 3; ++tc) {
                                const float dv = d[i + 2 - tr][2 - tc];  // dY at q - tap offset
                                dxa[i] = fmaf(wt[j][tr * 3 + tc], dv, dxa[i]);
                                accw[j][tr * 3 + tc] = fmaf(xc[i], dv, accw[j][tr * 3 + tc]);
                            }
                        accw[j][9] += d[i + 1][1];
                    }
                }
                if (dxp) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ok[i]) dxp[po + i * g.W] = dxa[i];
                }
                if (rpart) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float gg = (xc[i] > 0.f) ? dxa[i] : 0.f;  // xc is 0 outside the image
                        r1 += gg;
                        r2 = fmaf(gg, (zr[i] - rmean) * rinvstd, r2);
                    }
                }
            }
        }
    }
    if (rpart) {
        const float v1 = wave_sum_l63(r1), v2 = wave_sum_l63(r2);
        if (lane == 63) {
            red[4 * KPL * 10 + wave * 2 + 0] = v1;
            red[4 * KPL * 10 + wave * 2 + 1] = v2;
        }
    }
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const float v = wave_sum_l63(accw[j][t]);
            if (lane == 63) red[(wave * KPL + j) * 10 + t] = v;
        }
    }
    __syncthreads();
    if (rpart && tid < 2) {
        const float* rr_ = red + 4 * KPL * 10;
        const float v = rr_[tid] + rr_[2 + tid] + rr_[4 + tid] + rr_[6 + tid];
        const long rows = (long)gridDim.y * (gridDim.x / Cin);
        rpart[((long)tid * rows + (long)n * gridDim.y + blockIdx.y) * Cin + ci] = v;
    }
    if (tid < KPL * 10) {
        const int j = tid / 10, t = tid - j * 10;
        const float v = red[(0 * KPL + j) * 10 + t] + red[(1 * KPL + j) * 10 + t] + red[(2 * KPL + j) * 10 + t] +
                        red[(3 * KPL + j) * 10 + t];
        part[(((long)n * gridDim.y + blockIdx.y) * Cdw + ci * KPL + j) * 10 + t] = v;
    }
}

static inline int cdivs(long a, long b) { return (int)((a + b - 1) / b); }

int launch_maxpool2_fwd(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, hipStream_t st, int dt) {
    const int Po = (H / 2) * (W / 2);
    if (Po == 0) return -1;
    int gy = cdivs(Po, 1024);
    if (gy > 64) gy = 64;
    dim3 grid(N * C, gy);
    SMAAT_DISPATCH_ET(dt, T,
        if ((W & 1) == 0 && (x_bs & 1) == 0 && ((((uintptr_t)x) & (2 * sizeof(T) - 1)) == 0))
            hipLaunchKernelGGL(k_maxpool2_fwd<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, (T*)y, y_bs, C, H, W);
        else
            hipLaunchKernelGGL(k_maxpool2_fwd_s<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, (T*)y, y_bs, C, H, W););
    return (int)hipGetLastError();
}

int launch_maxpool2_bwd(const void* x, long x_bs, const void* dy, long dy_bs, void* dx, long dx_bs, int N, int C,
                        int H, int W, int accum, hipStream_t st, int dt) {
    int gy = cdivs((long)H * W, 2048);
    if (gy > 64) gy = 64;
    dim3 grid(N * C, gy);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_maxpool2_bwd<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, (const T*)dy, dy_bs, (T*)dx, dx_bs, C,
                           H, W, accum););
    return (int)hipGetLastError();
}

// uprows.hip: row-walking kernels (float4 rows); -2 = shape / alignment not handled there
int launch_upsample2x_fwd_rows(const void*, long, void*, long, int, int, int, int, int, int, int, int, hipStream_t, int, unsigned*);
int launch_upsample2x_bwd_rows(const void*, long, void*, long, int, int, int, int, int, int, int, int, hipStream_t, int);

// dt: SMAAT_F32 | SMAAT_BF16 (bf16 storage: the row-walking kernels only, -2 when they do not take the shape)
int launch_upsample2x_fwd(const void* xv, long x_bs, void* outv, long out_bs, int N, int C, int H, int W, int Ho,
                          int Wo, int pad_t, int pad_l, hipStream_t st, int dt) {
    {
        const int rc = launch_upsample2x_fwd_rows(xv, x_bs, outv, out_bs, N, C, H, W, Ho, Wo, pad_t, pad_l, st, dt, nullptr);
        if (rc != -2 || dt != SMAAT_F32) return rc;
    }
    const float* x = (const float*)xv;
    float* out = (float*)outv;
    int gy = cdivs((long)Ho * Wo, 2048);
    if (gy > 64) gy = 64;
    dim3 grid(N * C, gy);
    hipLaunchKernelGGL(k_upsample2x_fwd, grid, dim3(256), 0, st, x, x_bs, out, out_bs, C, H, W, Ho, Wo, pad_t, pad_l);
    return (int)hipGetLastError();
}

int launch_upsample2x_bwd(const void* doutv, long dout_bs, void* dxv, long dx_bs, int N, int C, int H, int W, int Ho,
                          int Wo, int pad_t, int pad_l, hipStream_t st, int dt) {
    {
        const int rc = launch_upsample2x_bwd_rows(doutv, dout_bs, dxv, dx_bs, N, C, H, W, Ho, Wo, pad_t, pad_l, st, dt);
        if (rc != -2 || dt != SMAAT_F32) return rc;
    }
    const float* dout = (const float*)doutv;
    float* dx = (float*)dxv;
    if (W <= 256) {
        const int PB = 256 / W;
        const int NC = N * C;
        int gy = (H + UTH - 1) / UTH;
        const long blocks_x = (NC + PB - 1) / PB;
        while (gy > 1 && blocks_x * gy > 16384) gy = (gy + 1) / 2;
        const size_t lds = sizeof(float) * ((size_t)PB * UROWS * W + UTH * 6);
        hipLaunchKernelGGL(k_upsample2x_bwd_sep, dim3((unsigned)blocks_x, gy), dim3(256), lds, st, dout, dout_bs, dx,
                           dx_bs, NC, C, H, W, Ho, Wo, pad_t, pad_l, PB);
        return (int)hipGetLastError();
    }
    int gy = cdivs((long)H * W, 1024);
    if (gy > 64) gy = 64;
    dim3 grid(N * C, gy);
    hipLaunchKernelGGL(k_upsample2x_bwd, grid, dim3(256), 0, st, dout, dout_bs, dx, dx_bs, C, H, W, Ho, Wo, pad_t,
                       pad_l);
    return (int)hipGetLastError();
}

int launch_pixel_shuffle2_fwd(const float* t, long t_bs, const float* bias, float* out, long out_bs, int N, int Co,
                              int H, int W, int Ho, int Wo, int pad_t, int pad_l, hipStream_t st) {
    int gy = cdivs((long)Ho * Wo, 2048);
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(k_pixel_shuffle2_fwd, dim3(N * Co, gy), dim3(256), 0, st, t, t_bs, bias, out, out_bs, Co, H, W, Ho,
                       Wo, pad_t, pad_l);
    return (int)hipGetLastError();
}

int launch_pixel_shuffle2_bwd(const float* dout, long dout_bs, float* dt, long dt_bs, int N, int Co, int H, int W, int Ho,
                              int Wo, int pad_t, int pad_l, hipStream_t st) {
    int gy = cdivs((long)H * W, 2048);
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(k_pixel_shuffle2_bwd, dim3(N * 4 * Co, gy), dim3(256), 0, st, dout, dout_bs, dt, dt_bs, Co, H, W,
                       Ho, Wo, pad_t, pad_l);
    return (int)hipGetLastError();
}

void choose_geom_pub(int N, int H, int W, int PT, int smax, TileGeom* g);  // pwgemm.hip

// strip-kernel tile geometry: widest tile (preferably the whole plane width) whose staged halo of
// `nplanes` channels fits the 6 float4 staging slots of a 256-thread block
static DwbGeom strip_geom(int H, int W, int nplanes) {
    DwbGeom sg;
    sg.H = H;
    sg.W = W;
    sg.P = H * W;
    const int cand[4] = {W <= 288 ? W : 0, 96, 48, 32};
    const int hq = (H + 3) / 4;
    sg.TW = 0;
    int th0 = 0;
    for (int c = 0; c < 4 && sg.TW == 0; ++c) {
        const int tw = cand[c];
        if (tw <= 0 || tw > W) continue;
        const int rows = (1536 / nplanes) / ((tw + 8) / 4) - 2;
        th0 = 4 * (rows / 4);
        if (th0 > 4 * hq) th0 = 4 * hq;
        if (tw <= 72 && th0 > 4 * (256 / tw)) th0 = 4 * (256 / tw);  // narrow planes: one strip per thread
        if (th0 >= 4) sg.TW = tw;
    }
    if (sg.TW == 0) {
        sg.TW = W < 32 ? W : 32;
        th0 = 4;
    }
    const int nty = (hq + th0 / 4 - 1) / (th0 / 4);
    sg.TH = 4 * ((hq + nty - 1) / nty);
    sg.tiles_x = (W + sg.TW - 1) / sg.TW;
    sg.tiles = sg.tiles_x * nty;
    sg.stride = sg.TW + 8;
    if (((4 * sg.stride) & 31) == 0) sg.stride += 4;  // row groups of one wave on different banks
    sg.nrow = sg.TH + 2;
    sg.ncol4 = (sg.TW + 8) / 4;
    sg.ssz = sg.nrow * sg.stride;
    return sg;
}

// ---------------------------------------------------------------------------------
// standalone depthwise 3x3 forward, strip form (used by the bf16-split matrix path, where the
// pointwise GEMM reads the depthwise output from HBM; the f32-MFMA path fuses this stage instead).
//   y[ci*kpl + j][q] = b[j] + sum_tap w[j][tap] * x[ci][q + tap offset]
// reference: models/layers.py:38-44,48
// ---------------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void k_dw3x3_fwd_strip(const float* __restrict__ x, long x_bs,
                                                         const float* __restrict__ w_dw,
                                                         const float* __restrict__ b_dw, float* __restrict__ y,
                                                         long y_bs, int Cin, const DwbGeom g,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift) {
    // in_scale/in_shift (nullable): the input is read as relu(x * in_scale[ci] + in_shift[ci]) -- the
    // BatchNorm-apply + ReLU of the previous half-block fused into this load (zero padding stays zero)
    constexpr int NSL = 6;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* S = dsm;  // [ssz]
    const int tid = threadIdx.x;
    const int plane = blockIdx.x, n = plane / Cin, ci = plane - n * Cin;
    const float* xp = x + (long)n * x_bs + (long)ci * g.P;
    float* yp = y + (long)n * y_bs + (long)(ci * KPL) * g.P;
    float wt[KPL][9], bs[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[j][t] = w_dw[(ci * KPL + j) * 9 + t];
        bs[j] = b_dw ? b_dw[ci * KPL + j] : 0.f;
    }
    const bool aff = in_scale != nullptr;
    const float asc = aff ? in_scale[ci] : 1.f, ash = aff ? in_shift[ci] : 0.f;
    const int F = g.nrow * g.ncol4;
    int s_rr[NSL], s_q[NSL], s_lo[NSL];
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int f = (tid + 256 * k) % F;
        s_rr[k] = f / g.ncol4;
        s_q[k] = f - s_rr[k] * g.ncol4;
        s_lo[k] = s_rr[k] * g.stride + 4 * s_q[k];
    }
    const int nstrips = (g.TH >> 2) * g.TW;
    const int nit = (nstrips + 255) >> 8;
    float4 sv[NSL];
    bool sok[NSL];
    auto prefetch = [&](int tl) {
        const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
        const int r0 = ty * g.TH, c0 = tx * g.TW;
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int gr = r0 - 1 + s_rr[k], gc = c0 - 4 + 4 * s_q[k];
            sok[k] = gr >= 0 && gr < g.H && gc >= 0 && gc < g.W;
            sv[k] = *(const float4*)(xp + (sok[k] ? gr * g.W + gc : 0));
        }
    };
    int tl = blockIdx.y;
    if (tl < g.tiles) prefetch(tl);
    for (; tl < g.tiles; tl += gridDim.y) {
        const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
        const int r0 = ty * g.TH, c0 = tx * g.TW;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            float4 v = sv[k];
            if (aff) {  // uniform per block
                v.x = fmaxf(fmaf(v.x, asc, ash), 0.f);
                v.y = fmaxf(fmaf(v.y, asc, ash), 0.f);
                v.z = fmaxf(fmaf(v.z, asc, ash), 0.f);
                v.w = fmaxf(fmaf(v.w, asc, ash), 0.f);
            }
            v.x = sok[k] ? v.x : 0.f;
            v.y = sok[k] ? v.y : 0.f;
            v.z = sok[k] ? v.z : 0.f;
            v.w = sok[k] ? v.w : 0.f;
            *(float4*)(S + s_lo[k]) = v;
        }
        __syncthreads();
        {
            const int tn = tl + gridDim.y;
            prefetch(tn < g.tiles ? tn : tl);
        }
#pragma unroll 1
        for (int it = 0; it < nit; ++it) {
            const int sidx = tid + (it << 8);
            if (sidx < nstrips) {
                const int srg = sidx / g.TW, sc = sidx - srg * g.TW;
                const float* sp = S + (srg * 4) * g.stride + sc + 3;
                const int prow = r0 + srg * 4;
                const bool pcol = (c0 + sc) < g.W;
                const int po = prow * g.W + c0 + sc;
                float v[6][3];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) v[rr][dc] = sp[rr * g.stride + dc];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float acc = bs[j];
#pragma unroll
                        for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                            for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][tr * 3 + tc], v[i + tr][tc], acc);
                        if (pcol && (prow + i) < g.H) yp[(long)j * g.P + po + i * g.W] = acc;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Small planes (any H, W with H*W <= DWS_PMAX, e.g. the 18 x 18 maps of the deepest level whose rows are not
// 16-byte aligned): whole planes are copied flat into LDS (no alignment requirement), PPB planes per workgroup so that
// all 256 threads have work, and the 3x3 window is read with boundary predicates instead of a zero halo.
// Workgroups walk over plane groups with the next group's loads in flight.
// ---------------------------------------------------------------------------------
#define DWS_PMAX 1600
#define DWS_LDS_FLOATS 8192
#define DWS_NLD 24  // staged values per thread

template <int KPL, typename TX, typename TY>
__global__ __launch_bounds__(256) void k_dw3x3_fwd_small(const TX* __restrict__ x, long x_bs,
                                                         const float* __restrict__ w_dw, const float* __restrict__ b_dw,
                                                         TY* __restrict__ y, long y_bs, int NC, int Cin, int H, int W,
                                                         int PPB, const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift) {
    // PPB (a power of two <= 16) planes per group; the 256 / PPB threads of a plane keep its taps in registers
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int P = H * W, tid = threadIdx.x;
    const int TPP = 256 / PPB, pl = tid / TPP, lt = tid - pl * TPP;
    const int ngroups = (NC + PPB - 1) / PPB;
    const float invW = 1.0f / (float)W;
    const int nld = (P + TPP - 1) / TPP;  // elements of its plane each thread stages (<= DWS_NLD)
    float sv[DWS_NLD];
    float wt[KPL][9], bs[KPL];
    auto prefetch = [&](int g) {
        const int pi = g * PPB + pl;
        const int pic = pi < NC ? pi : NC - 1;
        const int n = pic / Cin, ci = pic - n * Cin;
        const TX* xp = x + (long)n * x_bs + (long)ci * P;
        const float sc = in_scale ? in_scale[ci] : 1.f, sh = in_scale ? in_shift[ci] : 0.f;
#pragma unroll
        for (int k = 0; k < DWS_NLD; ++k) {
            const int p = lt + TPP * k;
            float v = ld1(xp + (p < P ? p : P - 1));
            if (in_scale) v = fmaxf(fmaf(v, sc, sh), 0.f);
            sv[k] = v;
        }
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[j][t] = w_dw[(ci * KPL + j) * 9 + t];
            bs[j] = b_dw ? b_dw[ci * KPL + j] : 0.f;
        }
    };
    int g = blockIdx.x;
    if (g < ngroups) prefetch(g);
    for (; g < ngroups; g += gridDim.x) {
        float w0[KPL][9], b0[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
#pragma unroll
            for (int t = 0; t < 9; ++t) w0[j][t] = wt[j][t];
            b0[j] = bs[j];
        }
        __syncthreads();  // the previous group's reads are done
#pragma unroll
        for (int k = 0; k < DWS_NLD; ++k) {
            const int p = lt + TPP * k;
            if (k < nld && p < P) dsm[pl * P + p] = sv[k];
        }
        __syncthreads();
        {
            const int gn = g + gridDim.x;
            prefetch(gn < ngroups ? gn : g);  // in flight during the compute below
        }
        const int pi = g * PPB + pl;
        if (pi < NC) {
            const int n = pi / Cin, ci = pi - n * Cin;
            const float* sb = dsm + pl * P;
            TY* yb = y + (long)n * y_bs + (long)(ci * KPL) * P;
            for (int p = lt; p < P; p += TPP) {
                const int r = (int)(((float)p + 0.5f) * invW), c = p - r * W;
                const float* sp = sb + p;
                float v[3][3];
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) {
                        const bool in = (r + dr - 1) >= 0 && (r + dr - 1) < H && (c + dc - 1) >= 0 && (c + dc - 1) < W;
                        v[dr][dc] = in ? sp[(dr - 1) * W + (dc - 1)] : 0.f;
                    }
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    float acc = b0[j];
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc = fmaf(w0[j][t], v[t / 3][t % 3], acc);
                    st1(yb + (long)j * P + p, acc);
                }
            }
        }
    }
}

static int launch_dw3x3_fwd_small(const void* x, int x_dt, long x_bs, const float* w_dw, const float* b_dw, void* y,
                                  int y_dt, long y_bs, int N, int Cin, int kpl, int H, int W, hipStream_t st,
                                  const float* in_scale, const float* in_shift) {
    const int P = H * W;
    int PPB = 16;  // planes per workgroup: a power of two, 256 / PPB threads per plane, <= DWS_NLD staged values per thread
    while (PPB > 1 && ((P + 256 / PPB - 1) / (256 / PPB) > DWS_NLD || PPB * P > DWS_LDS_FLOATS)) PPB >>= 1;
    if ((P + 256 / PPB - 1) / (256 / PPB) > DWS_NLD || PPB * P > DWS_LDS_FLOATS) return -2;
    const int NC = N * Cin;
    int grid = (NC + PPB - 1) / PPB;
    if (grid > 2048) grid = 2048;
    const size_t lds = sizeof(float) * (size_t)PPB * P;
#define DWS_LAUNCH(K, TX, TY)                                                                                          \
    hipLaunchKernelGGL((k_dw3x3_fwd_small<K, TX, TY>), dim3(grid), dim3(256), lds, st, (const TX*)x, x_bs, w_dw, b_dw,  \
                       (TY*)y, y_bs, NC, Cin, H, W, PPB, in_scale, in_shift)
#define DWS_K(TX, TY)                          \
    do {                                       \
        if (kpl == 1) DWS_LAUNCH(1, TX, TY);   \
        else if (kpl == 2) DWS_LAUNCH(2, TX, TY); \
        else DWS_LAUNCH(4, TX, TY);            \
    } while (0)
    if (x_dt == SMAAT_F32 && y_dt == SMAAT_F32) DWS_K(float, float);
    else if (x_dt == SMAAT_F32 && y_dt == SMAAT_BF16) DWS_K(float, bf16_t);
    else if (x_dt == SMAAT_BF16 && y_dt == SMAAT_BF16) DWS_K(bf16_t, bf16_t);
    else return -2;
#undef DWS_K
#undef DWS_LAUNCH
    return (int)hipGetLastError();
}

// dwrows.hip: register row-streaming kernels (W % 4 == 0)
int dw_rows_ok(int kpl, int H, int W);
int dw_rows_wpp(int N, int Cin, int H, int W);
int launch_dw3x3_fwd_rows(const void*, int, long, const float*, const float*, void*, int, long, int, int, int, int, int,
                          hipStream_t, const float*, const float*, unsigned* amax);
int launch_dw3x3_bwd_rows(const void*, int, long, const void*, int, long, const float*, void*, int, long, float*, int, int,
                          int, int, int, hipStream_t, const float*, const float*, float*, const float*, const float*);

// x_dt / y_dt: SMAAT_F32 | SMAAT_BF16 (bf16 storage: the row-streaming and the small-plane kernels only)
int launch_dw3x3_fwd(const void* xv, int x_dt, long x_bs, const float* w_dw, const float* b_dw, void* yv, int y_dt,
                     long y_bs, int N, int Cin, int kpl, int H, int W, hipStream_t st, const float* in_scale,
                     const float* in_shift, unsigned* amax) {
    // amax (nullable): device word that receives max |y| (bit pattern; must hold 0 on entry) -- only the row-streaming
    // kernels produce it: -2 when another kernel would take the shape
    const unsigned xm = x_dt == SMAAT_BF16 ? 7u : 15u, ym = y_dt == SMAAT_BF16 ? 7u : 15u;
    const bool aligned = ((W & 3) == 0) && ((x_bs & 3) == 0) && ((((uintptr_t)xv) & xm) == 0) && H >= 1 &&
                         (kpl == 1 || kpl == 2 || kpl == 4);
    if (!(kpl == 1 || kpl == 2 || kpl == 4)) return -2;
    // W % 4 == 2 (18 x 18 ...): the row kernels with a two-column last group; rows are 8- (f32) / 4-byte (bf16) aligned
    if (!aligned && (W & 3) == 2 && dw_rows_ok(kpl, H, W) && (x_bs & 1) == 0 && (y_bs & 1) == 0 &&
        ((((uintptr_t)xv) & (xm >> 1)) == 0) && ((((uintptr_t)yv) & (ym >> 1)) == 0))
        return launch_dw3x3_fwd_rows(xv, x_dt, x_bs, w_dw, b_dw, yv, y_dt, y_bs, N, Cin, kpl, H, W, st, in_scale, in_shift, amax);
    if (amax && !(aligned && dw_rows_ok(kpl, H, W) && (y_bs & 3) == 0 && ((((uintptr_t)yv) & ym) == 0))) return -2;
    if (!aligned && H * W <= DWS_PMAX)  // small planes with unaligned rows (18 x 18 ...): the flat-copy kernel
        return launch_dw3x3_fwd_small(xv, x_dt, x_bs, w_dw, b_dw, yv, y_dt, y_bs, N, Cin, kpl, H, W, st, in_scale, in_shift);
    if (!aligned) return -2;  // caller falls back to the fused f32 kernel
    if (dw_rows_ok(kpl, H, W) && (y_bs & 3) == 0 && ((((uintptr_t)yv) & ym) == 0))
        return launch_dw3x3_fwd_rows(xv, x_dt, x_bs, w_dw, b_dw, yv, y_dt, y_bs, N, Cin, kpl, H, W, st, in_scale, in_shift, amax);
    if (x_dt != SMAAT_F32 || y_dt != SMAAT_F32) return -2;
    const float* x = (const float*)xv;
    float* y = (float*)yv;
    const DwbGeom sg = strip_geom(H, W, 1);
    if (sg.nrow * sg.ncol4 > 1536) return -2;
    long planes = (long)N * Cin;
    int groups = (int)((8192 + planes - 1) / planes);
    if (groups > sg.tiles) groups = sg.tiles;
    if (groups > 64) groups = 64;
    if (groups < 1) groups = 1;
    const size_t lds = sizeof(float) * (size_t)sg.ssz;
    dim3 grid(N * Cin, groups);
    if (kpl == 1)
        hipLaunchKernelGGL(k_dw3x3_fwd_strip<1>, grid, dim3(256), lds, st, x, x_bs, w_dw, b_dw, y, y_bs, Cin, sg,
                           in_scale, in_shift);
    else if (kpl == 2)
        hipLaunchKernelGGL(k_dw3x3_fwd_strip<2>, grid, dim3(256), lds, st, x, x_bs, w_dw, b_dw, y, y_bs, Cin, sg,
                           in_scale, in_shift);
    else
        hipLaunchKernelGGL(k_dw3x3_fwd_strip<4>, grid, dim3(256), lds, st, x, x_bs, w_dw, b_dw, y, y_bs, Cin, sg,
                           in_scale, in_shift);
    return (int)hipGetLastError();
}

// tile groups per plane: enough workgroups to fill the chip, few enough to keep the partials small
int dw_bwd_groups(int N, int Cin, int H, int W) {
    if (dw_rows_ok(1, H, W)) return dw_rows_wpp(N, Cin, H, W);  // the row kernels: one partial row per wave of a plane
    TileGeom g;
    choose_geom_pub(N, H, W, 256, DWB_SMAX, &g);
    long planes = (long)N * Cin;
    int groups = (int)((8192 + planes - 1) / planes);
    if (groups > g.tiles_per_img) groups = g.tiles_per_img;
    if (groups > 64) groups = 64;
    if (groups < 1) groups = 1;
    return groups;
}

// x_dt / dy_dt / dx_dt: SMAAT_F32 | SMAAT_BF16 (bf16 storage: the row-streaming and the generic kernels only)
int launch_dw3x3_bwd(const void* xv, int x_dt, long x_bs, const void* dyv, int dy_dt, long dy_bs, const float* w_dw,
                     void* dxv, int dx_dt, long dx_bs, float* part, int N, int Cin, int kpl, int H, int W, hipStream_t st,
                     const float* bn_g, const float* bn_b, float* rpart, const float* in_scale, const float* in_shift) {
    if (kpl < 1 || kpl > DWB_KPL_MAX) return -1;
    const int groups = dw_bwd_groups(N, Cin, H, W);
    static int use_strip = -1;
    if (use_strip < 0) {
        const char* e = getenv("SMAAT_DWB_STRIP");
        use_strip = e ? atoi(e) : 1;
    }
    const bool all32 = x_dt == SMAAT_F32 && dy_dt == SMAAT_F32 && dx_dt == SMAAT_F32;
    const unsigned xm = x_dt == SMAAT_BF16 ? 7u : 15u, gm = dy_dt == SMAAT_BF16 ? 7u : 15u, dm = dx_dt == SMAAT_BF16 ? 7u : 15u;
    const bool aligned = ((W & 3) == 0) && ((x_bs & 3) == 0) && ((dy_bs & 3) == 0) && ((dx_bs & 3) == 0) &&
                         ((((uintptr_t)xv) & xm) == 0) && ((((uintptr_t)dyv) & gm) == 0) &&
                         ((((uintptr_t)dxv) & dm) == 0) && (kpl == 1 || kpl == 2 || kpl == 4) && H >= 4;
    if (rpart && !in_scale) return -2;  // the fused reduction needs the pre-BatchNorm tensor (zhat = (z - mean) * invstd)
    const bool aligned2 = ((W & 3) == 2) && ((x_bs & 1) == 0) && ((dy_bs & 1) == 0) && ((dx_bs & 1) == 0) &&
                          ((((uintptr_t)xv) & (xm >> 1)) == 0) && ((((uintptr_t)dyv) & (gm >> 1)) == 0) &&
                          ((((uintptr_t)dxv) & (dm >> 1)) == 0) && H >= 4;  // two-column last group, see dwrows.hip
    if ((aligned || aligned2) && kpl <= 2 && dw_rows_ok(kpl, H, W))
        return launch_dw3x3_bwd_rows(xv, x_dt, x_bs, dyv, dy_dt, dy_bs, w_dw, dxv, dx_dt, dx_bs, part, N, Cin, kpl, H, W, st,
                                     bn_g, bn_b, rpart, in_scale, in_shift);
    if (use_strip && aligned && all32) {
        const float* x = (const float*)xv;
        const float* dy = (const float*)dyv;
        float* dx = (float*)dxv;
        const DwbGeom sg = strip_geom(H, W, kpl);
        if (kpl * sg.nrow * sg.ncol4 <= 1536) {
            const size_t lds = sizeof(float) * ((size_t)kpl * sg.ssz + 4 * kpl * 10 + 8);
            dim3 grid(N * Cin, groups);
            if (kpl == 1)
                hipLaunchKernelGGL(k_dw3x3_bwd_strip<1>, grid, dim3(256), lds, st, x, x_bs, dy, dy_bs, w_dw, dx, dx_bs,
                                   part, Cin, sg, bn_g, bn_b, rpart, in_scale, in_shift);
            else if (kpl == 2)
                hipLaunchKernelGGL(k_dw3x3_bwd_strip<2>, grid, dim3(256), lds, st, x, x_bs, dy, dy_bs, w_dw, dx, dx_bs,
                                   part, Cin, sg, bn_g, bn_b, rpart, in_scale, in_shift);
            else
                hipLaunchKernelGGL(k_dw3x3_bwd_strip<4>, grid, dim3(256), lds, st, x, x_bs, dy, dy_bs, w_dw, dx, dx_bs,
                                   part, Cin, sg, bn_g, bn_b, rpart, in_scale, in_shift);
            return (int)hipGetLastError();
        }
    }
    if (rpart || in_scale) return -2;  // the fused BatchNorm pieces exist in the row / strip kernels only
    TileGeom g;
    choose_geom_pub(N, H, W, 256, DWB_SMAX, &g);
    if (g.mode < 0) return -1;
#define DWG_GO(TX, TG, TD)                                                                                               \
    hipLaunchKernelGGL((k_dw3x3_bwd<TX, TG, TD>), dim3(N * Cin, groups), dim3(256), 0, st, (const TX*)xv, x_bs, (const TG*)dyv, \
                       dy_bs, w_dw, (TD*)dxv, dx_bs, part, Cin, kpl, g)
    if (all32) DWG_GO(float, float, float);
    else if (x_dt == SMAAT_BF16 && dy_dt == SMAAT_BF16 && dx_dt == SMAAT_BF16) DWG_GO(bf16_t, bf16_t, bf16_t);
    else if (x_dt == SMAAT_F32 && dy_dt == SMAAT_BF16 && dx_dt == SMAAT_F32) DWG_GO(float, bf16_t, float);
    else return -2;
#undef DWG_GO
    return (int)hipGetLastError();
}

// 1 when BOTH strip kernels (forward with the activation applied on load, backward with the fused BatchNorm
// reduction) take planes of this shape, given dense 16-byte aligned tensors: the host uses it to decide whether the
// first activation of a DoubleConvDS can stay unmaterialised (ops.py FUSE_FIRST_ACTIVATION)
int dw3x3_strip_ok(int kpl, int H, int W) {
    const char* e = getenv("SMAAT_DWB_STRIP");
    if (e && atoi(e) == 0) return 0;
    if ((W & 3) == 2)  // the row-streaming kernels with a two-column last group (18 x 18 ...): kernels_per_layer 1, 2
        return ((kpl == 1 || kpl == 2) && H >= 4 && dw_rows_ok(kpl, H, W)) ? 1 : 0;
    if (!(kpl == 1 || kpl == 2 || kpl == 4) || (W & 3) != 0 || H < 4) return 0;
    const DwbGeom f = strip_geom(H, W, 1), b = strip_geom(H, W, kpl);
    return (f.nrow * f.ncol4 <= 1536 && kpl * b.nrow * b.ncol4 <= 1536) ? 1 : 0;
}

// part[rows][Cdw][10] -> dW[Cdw][9], db[Cdw]: the fixed-order fp64 row sum and the split in one launch.
// A block owns 64 consecutive columns; its 16 row groups sum rows rg, rg + 16, ... (independent coalesced loads) and
// are then added in a fixed order.
__global__ __launch_bounds__(1024) void k_dw_reduce_split(const float* __restrict__ part, int rows, int Cdw,
                                                          float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int len = Cdw * 10;
    const int i = blockIdx.x * 64 + col;
    double s = 0.0;
    if (i < len) {
#pragma unroll 4
        for (int r = rg; r < rows; r += 16) s += (double)part[(long)r * len + i];
    }
    red[rg][col] = s;
    __syncthreads();
    if (rg == 0 && i < len) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        const int k = i / 10, tt = i - k * 10;
        if (tt < 9)
            dw[k * 9 + tt] = (float)t;
        else
            db[k] = (float)t;
    }
}

int launch_dw_reduce_split(const float* part, int rows, int Cdw, float* dw, float* db, hipStream_t st) {
    hipLaunchKernelGGL(k_dw_reduce_split, dim3(cdivs((long)Cdw * 10, 64)), dim3(1024), 0, st, part, rows, Cdw, dw, db);
    return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// Depthwise convolution of ANY geometry the reference's DepthwiseSeparableConv can be constructed with
// (models/layers.py:35-45: nn.Conv2d(Cin, Cin * kpl, kernel_size, padding=padding, groups=Cin), stride 1, dilation 1;
// any kernels_per_layer).  The 3x3 / padding-1 / kpl in {1, 2, 4} configuration of the network has its own kernels
// above and in dwrows.hip; this is the general form behind the same module, written for coverage, not for the roofline:
// direct gather kernels, one thread per output element, weights through the scalar cache.
//   Ho = H + 2 ph - KH + 1, Wo = W + 2 pw - KW + 1
//   y[n][k][oh][ow]  = b[k] + sum_{th,tw} w[k][th][tw] * x[n][k / kpl][oh + th - ph][ow + tw - pw]
//   dx[n][c][h][w]   = sum_{j<kpl} sum_{th,tw} w[c kpl + j][th][tw] * dy[n][c kpl + j][h - th + ph][w - tw + pw]
//   dw[k][th][tw]    = sum_{n,oh,ow} dy[n][k][oh][ow] * x[n][k / kpl][oh + th - ph][ow + tw - pw],   db[k] = sum dy[n][k]
// The weight gradient runs one workgroup per (output channel, tap) -- tap KH*KW is the bias -- and reduces over the batch
// and the plane in a fixed order (thread-strided partial sums, wave sums, fp64 across the four waves): deterministic.
// ---------------------------------------------------------------------------------------------------------------
struct DwgGeom {
    int N, Cin, kpl, H, W, KH, KW, ph, pw, Ho, Wo;
};

__global__ __launch_bounds__(256) void k_dwconv_fwd_any(const float* __restrict__ x, long x_bs, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, long y_bs,
                                                        const DwgGeom g) {
    const int k = blockIdx.y, n = blockIdx.z, c = k / g.kpl;
    const int po = blockIdx.x * 256 + threadIdx.x;
    if (po >= g.Ho * g.Wo) return;
    const int oh = po / g.Wo, ow = po - oh * g.Wo;
    const float* xp = x + (long)n * x_bs + (long)c * g.H * g.W;
    const float* wk = w + (long)k * g.KH * g.KW;
    float acc = b ? b[k] : 0.f;
    for (int th = 0; th < g.KH; ++th) {
        const int ih = oh + th - g.ph;
        if (ih < 0 || ih >= g.H) continue;
        for (int tw = 0; tw < g.KW; ++tw) {
            const int iw = ow + tw - g.pw;
            if (iw >= 0 && iw < g.W) acc = fmaf(wk[th * g.KW + tw], xp[(long)ih * g.W + iw], acc);
        }
    }
    y[(long)n * y_bs + (long)k * g.Ho * g.Wo + po] = acc;
}

__global__ __launch_bounds__(256) void k_dwconv_bwd_dx_any(const float* __restrict__ dy, long dy_bs,
                                                           const float* __restrict__ w, float* __restrict__ dx, long dx_bs,
                                                           const DwgGeom g) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= g.H * g.W) return;
    const int h = pi / g.W, wc = pi - h * g.W;
    float acc = 0.f;
    for (int j = 0; j < g.kpl; ++j) {
        const int k = c * g.kpl + j;
        const float* gp = dy + (long)n * dy_bs + (long)k * g.Ho * g.Wo;
        const float* wk = w + (long)k * g.KH * g.KW;
        for (int th = 0; th < g.KH; ++th) {
            const int oh = h - th + g.ph;
            if (oh < 0 || oh >= g.Ho) continue;
            for (int tw = 0; tw < g.KW; ++tw) {
                const int ow = wc - tw + g.pw;
                if (ow >= 0 && ow < g.Wo) acc = fmaf(wk[th * g.KW + tw], gp[(long)oh * g.Wo + ow], acc);
            }
        }
    }
    dx[(long)n * dx_bs + (long)c * g.H * g.W + pi] = acc;
}

__global__ __launch_bounds__(256) void k_dwconv_bwd_w_any(const float* __restrict__ x, long x_bs, const float* __restrict__ dy,
                                                          long dy_bs, float* __restrict__ dw, float* __restrict__ db,
                                                          const DwgGeom g) {
    __shared__ float red[4];
    const int k = blockIdx.y, tap = blockIdx.x, c = k / g.kpl, KK = g.KH * g.KW;
    const bool bias = tap == KK;
    const int th = bias ? 0 : tap / g.KW, tw = bias ? 0 : tap - th * g.KW;
    const int Po = g.Ho * g.Wo;
    float s = 0.f;
    for (int n = 0; n < g.N; ++n) {
        const float* gp = dy + (long)n * dy_bs + (long)k * Po;
        const float* xp = x + (long)n * x_bs + (long)c * g.H * g.W;
        for (int po = threadIdx.x; po < Po; po += 256) {
            const float gv = gp[po];
            if (bias) {
                s += gv;
            } else {
                const int oh = po / g.Wo, ow = po - oh * g.Wo;
                const int ih = oh + th - g.ph, iw = ow + tw - g.pw;
                if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) s = fmaf(gv, xp[(long)ih * g.W + iw], s);
            }
        }
    }
    s = wave_sum_l63(s);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (float)(((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]));
        if (bias) {
            if (db) db[k] = t;
        } else {
            dw[(long)k * KK + tap] = t;
        }
    }
}

static int dwg_geom(DwgGeom& g, int N, int Cin, int kpl, int H, int W, int KH, int KW, int ph, int pw) {
    if (N < 1 || Cin < 1 || kpl < 1 || H < 1 || W < 1 || KH < 1 || KW < 1 || ph < 0 || pw < 0) return -1;
    g = DwgGeom{N, Cin, kpl, H, W, KH, KW, ph, pw, H + 2 * ph - KH + 1, W + 2 * pw - KW + 1};
    if (g.Ho < 1 || g.Wo < 1) return -1;
    if ((long)Cin * kpl > 65535 || N > 65535 || (long)H * W >= (1L << 31) || (long)g.Ho * g.Wo >= (1L << 31)) return -2;
    return 0;
}

int launch_dwconv_fwd_any(const float* x, long x_bs, const float* w, const float* b, float* y, long y_bs, int N, int Cin,
                          int kpl, int H, int W, int KH, int KW, int ph, int pw, hipStream_t st) {
    DwgGeom g;
    const int rc = dwg_geom(g, N, Cin, kpl, H, W, KH, KW, ph, pw);
    if (rc) return rc;
    hipLaunchKernelGGL(k_dwconv_fwd_any, dim3(cdivs((long)g.Ho * g.Wo, 256), Cin * kpl, N), dim3(256), 0, st, x, x_bs, w, b, y,
                       y_bs, g);
    return (int)hipGetLastError();
}

int launch_dwconv_bwd_any(const float* x, long x_bs, const float* dy, long dy_bs, const float* w, float* dx, long dx_bs,
                          float* dw, float* db, int N, int Cin, int kpl, int H, int W, int KH, int KW, int ph, int pw,
                          hipStream_t st) {
    DwgGeom g;
    const int rc = dwg_geom(g, N, Cin, kpl, H, W, KH, KW, ph, pw);
    if (rc) return rc;
    if (dx)
        hipLaunchKernelGGL(k_dwconv_bwd_dx_any, dim3(cdivs((long)H * W, 256), Cin, N), dim3(256), 0, st, dy, dy_bs, w, dx, dx_bs, g);
    if (dw)
        hipLaunchKernelGGL(k_dwconv_bwd_w_any, dim3(KH * KW + 1, Cin * kpl), dim3(256), 0, st, x, x_bs, dy, dy_bs, dw, db, g);
    return (int)hipGetLastError();
}
