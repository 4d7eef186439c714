// Bilinear 2x upsampling (align_corners=True) + F.pad, forward and backward, as row-walking kernels (gfx950).
//
// reference: nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) followed by F.pad and torch.cat
// (models/unet_parts_depthwise_separable.py:64,76-85).  Semantics and coefficient arithmetic are those of the
// element-per-thread kernels in spatial.hip (ac_coef); what changes is the access pattern:
//   forward : a thread owns FOUR adjacent output columns and walks down a band of output rows; the two input rows an
//             output row blends are interpolated horizontally once, kept in registers and re-used by the (usually two)
//             output rows that share them; one global_store_dwordx4 per output row instead of four dword stores, a
//             quarter of the loads, a third of the arithmetic (the element-per-thread form is VALU bound).
//   backward: a thread owns the input columns (2m + 1, 2m + 2) -- every upsampled column that touches them lies in the
//             8 aligned columns [4m, 4m + 8), i.e. two float4 loads per gradient row -- and walks down the gradient
//             rows of a band, scattering each column-reduced row into the accumulators of the two input rows it was
//             blended from; an input row is stored once the walk has passed it.  No LDS, no barrier, deterministic.
// Measured (MI355X, batch 32, the four decoder levels of config 2): see profiles/r2.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "common.h"

// the coefficient rule shared with spatial.hip (PyTorch's area_pixel_compute_source_index, align_corners=True)
__device__ __forceinline__ void upr_coef(int o, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
    const float src = (float)o * scale;
    i0 = (int)floorf(src);
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

struct UprGeom {
    int C, H, W, Ho, Wo, pad_t, pad_l;
    int ntr;     // threads per band row
    int nbands;  // bands per plane
    int BH;      // rows per band (forward: output rows, backward: input rows)
    long total;  // planes * nbands * ntr
};

// ---------------------------------------------------------------------------------------------------------------
// forward (Wo % 4 == 0, 16-byte aligned output planes)
// ---------------------------------------------------------------------------------------------------------------
// T: element type of x and out (f32 | bf16 storage)
template <typename T>
__global__ __launch_bounds__(256) void k_upsample2x_fwd_rows(const T* __restrict__ x, long x_bs,
                                                             T* __restrict__ out, long out_bs, const UprGeom g,
                                                             unsigned* __restrict__ amax) {
    __shared__ float amred[4];
    // (threads beyond the list repeat its last entry -- the same values to the same addresses -- instead of returning: every
    // lane is alive for the block-wide maximum below, and no store is divergent)
    const long gid_ = (long)blockIdx.x * 256 + threadIdx.x;
    const long gid = gid_ < g.total ? gid_ : g.total - 1;
    float am = 0.f;
    const int per = g.nbands * g.ntr;
    const int plane = (int)(gid / per), rem = (int)(gid - (long)plane * per);
    const int band = rem / g.ntr, q = rem - band * g.ntr;
    const int n = plane / g.C, c = plane - n * g.C;
    const T* xp = x + (long)n * x_bs + (long)c * g.H * g.W;
    T* op = out + (long)n * out_bs + (long)c * g.Ho * g.Wo + 4 * q;
    const int H2 = 2 * g.H, W2 = 2 * g.W;
    const float sh = H2 > 1 ? (float)(g.H - 1) / (float)(H2 - 1) : 0.f;
    const float sw = W2 > 1 ? (float)(g.W - 1) / (float)(W2 - 1) : 0.f;
    // column side: fixed per thread
    int c0[4], c1[4];
    float b0[4], b1[4];
    bool cv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int uc = 4 * q + i - g.pad_l;
        cv[i] = uc >= 0 && uc < W2;
        upr_coef(cv[i] ? uc : 0, sw, g.W, c0[i], c1[i], b0[i], b1[i]);
    }
    const int ro0 = band * g.BH;
    const int ro1 = ro0 + g.BH < g.Ho ? ro0 + g.BH : g.Ho;
    // horizontally interpolated input rows cur and cur + 1 (clamped to H - 1): 8 loads + 4 blends per INPUT row, shared by
    // the (usually two) output rows between them -- the vertical blend is all that is left per output row
    float htop[4], hbot[4];
    int cur = -2;
    auto hrow = [&](int row, float (&hv)[4]) {
        const T* p = xp + (long)(row < g.H - 1 ? row : g.H - 1) * g.W;
#pragma unroll
        for (int i = 0; i < 4; ++i) hv[i] = bilerp_h(b0[i], ld1(p + c0[i]), b1[i], ld1(p + c1[i]));
    };
    for (int r = ro0; r < ro1; ++r) {
        const int ur = r - g.pad_t;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ur >= 0 && ur < H2) {
            int r0, r1;
            float a0, a1;
            upr_coef(ur, sh, g.H, r0, r1, a0, a1);
            if (r0 != cur) {
                if (r0 == cur + 1) {  // the usual step: one row down
#pragma unroll
                    for (int i = 0; i < 4; ++i) htop[i] = hbot[i];
                } else {  // first row of the band
                    hrow(r0, htop);
                }
                hrow(r0 + 1, hbot);
                cur = r0;
            }
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = cv[i] ? bilerp_v(a0, htop[i], a1, hbot[i]) : 0.f;
            o = make_float4(v[0], v[1], v[2], v[3]);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
        st4(op + (long)r * g.Wo, o);
    }
    // nullable amax buffer (common.h): max |out| -- the scale bound of a row-walking fused forward that reads the concatenation
    // buffer this kernel fills (dsrows.hip, NT == 2; f32 storage -- the bf16 store would round the maximum up)
    if (amax) amax_publish_block256(amax, am, blockIdx.x, amred);  // (block-uniform branch; every thread is alive)
}

// ---------------------------------------------------------------------------------------------------------------
// backward (W even, Wo % 4 == 0, pad_l % 4 == 0, 16-byte aligned gradient planes)
// thread mm = m + 1 of a band row owns the input columns wA = 2m + 1 and wB = 2m + 2, m = -1 .. W/2 - 1
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_upsample2x_bwd_rows(const T* __restrict__ dout, long dout_bs,
                                                             T* __restrict__ dx, long dx_bs, const UprGeom g) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= g.total) return;
    const int per = g.nbands * g.ntr;
    const int plane = (int)(gid / per), rem = (int)(gid - (long)plane * per);
    const int band = rem / g.ntr, mm = rem - band * g.ntr;
    const int m = mm - 1;
    const int n = plane / g.C, c = plane - n * g.C;
    const T* gp = dout + (long)n * dout_bs + (long)c * g.Ho * g.Wo;
    T* dp = dx + (long)n * dx_bs + (long)c * g.H * g.W;
    const int H2 = 2 * g.H, W2 = 2 * g.W;
    const float sh = H2 > 1 ? (float)(g.H - 1) / (float)(H2 - 1) : 0.f;
    const float sw = W2 > 1 ? (float)(g.W - 1) / (float)(W2 - 1) : 0.f;
    const int wA = 2 * m + 1, wB = 2 * m + 2;
    const bool vA = wA >= 0 && wA < g.W, vB = wB >= 0 && wB < g.W;
    // column weights of the 8 upsampled columns 4m .. 4m + 7 onto wA and wB (the forward's coefficients, exactly)
    float kA[8], kB[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int uc = 4 * m + t;
        const bool v = uc >= 0 && uc < W2;
        int c0, c1;
        float b0, b1;
        upr_coef(v ? uc : 0, sw, g.W, c0, c1, b0, b1);
        float a = 0.f, b = 0.f;
        if (c0 == wA) a += b0;
        if (c1 == wA) a += b1;
        if (c0 == wB) b += b0;
        if (c1 == wB) b += b1;
        kA[t] = (v && vA) ? a : 0.f;
        kB[t] = (v && vB) ? b : 0.f;
    }
    // the two float4 of a gradient row: columns 4m + pad_l .. (whole float4 inside or outside the upsampled image)
    const int g0 = 4 * m + g.pad_l, g1 = g0 + 4;
    const bool l0 = m >= 0 && 4 * m + 3 < W2, l1 = 4 * m + 4 >= 0 && 4 * m + 7 < W2;
    const int o0 = l0 ? g0 : (l1 ? g1 : 0), o1 = l1 ? g1 : o0;  // clamped addresses (masked by the zero weights)
    const int h0 = band * g.BH;
    const int h1 = h0 + g.BH < g.H ? h0 + g.BH : g.H;
    int olo = 2 * h0 - 2, ohi = 2 * h1 + 1;
    if (olo < 0) olo = 0;
    if (ohi > H2 - 1) ohi = H2 - 1;
    float tA = 0.f, tB = 0.f, uA = 0.f, uB = 0.f;  // accumulators of input rows cur and cur + 1
    int cur;
    {
        int r0, r1;
        float a0, a1;
        upr_coef(olo, sh, g.H, r0, r1, a0, a1);
        cur = r0;
    }
    auto flush = [&](int row, float a, float b) {
        if (row >= h0 && row < h1) {
            if (vA) st1(dp + (long)row * g.W + wA, a);
            if (vB) st1(dp + (long)row * g.W + wB, b);
        }
    };
    for (int orow = olo; orow <= ohi; ++orow) {
        int r0, r1;
        float a0, a1;
        upr_coef(orow, sh, g.H, r0, r1, a0, a1);
        if (r0 != cur) {  // (advances by exactly one row)
            flush(cur, tA, tB);
            tA = uA;
            tB = uB;
            uA = uB = 0.f;
            cur = r0;
        }
        const int gr = orow + g.pad_t;  // pad_t >= 0 and 2H + pad_t <= Ho: always inside the gradient plane
        const T* grow = gp + (long)gr * g.Wo;
        const float4 f0 = ld4(grow + o0), f1 = ld4(grow + o1);
        const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        float cA = 0.f, cB = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            cA = fmaf(kA[t], v[t], cA);
            cB = fmaf(kB[t], v[t], cB);
        }
        tA = fmaf(a0, cA, tA);
        tB = fmaf(a0, cB, tB);
        if (r1 == r0) {  // clamped last row: both coefficients land on it
            tA = fmaf(a1, cA, tA);
            tB = fmaf(a1, cB, tB);
        } else {
            uA = fmaf(a1, cA, uA);
            uB = fmaf(a1, cB, uB);
        }
    }
    flush(cur, tA, tB);
    flush(cur + 1, uA, uB);
}

static int upr_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMAAT_UP_ROWS");
        v = e ? atoi(e) : 1;
    }
    return v;
}

// target_bh / min_threads: measured per direction (profiles/r3/up_bench_geometry_r3.txt): the forward (a pure store stream per
// thread) prefers long bands and fewer threads, the backward the opposite
static UprGeom upr_geom(int N, int C, int H, int W, int Ho, int Wo, int pad_t, int pad_l, int ntr, int rows,
                        int target_bh, long min_threads) {
    UprGeom g;
    g.C = C;
    g.H = H;
    g.W = W;
    g.Ho = Ho;
    g.Wo = Wo;
    g.pad_t = pad_t;
    g.pad_l = pad_l;
    g.ntr = ntr;
    int nb = (rows + target_bh - 1) / target_bh;
    // enough threads to fill the chip (256 CUs x 2048 lanes) a few times over
    while ((long)N * C * nb * ntr < min_threads && nb < rows / 4) ++nb;
    if (nb < 1) nb = 1;
    g.BH = (rows + nb - 1) / nb;
    g.nbands = (rows + g.BH - 1) / g.BH;
    g.total = (long)N * C * g.nbands * ntr;
    return g;
}

// return -2: shape / alignment not handled here (the caller uses the element-per-thread kernels)
// dt: SMAAT_F32 | SMAAT_BF16 element type of both tensors
int launch_upsample2x_fwd_rows(const void* x, long x_bs, void* out, long out_bs, int N, int C, int H, int W, int Ho,
                               int Wo, int pad_t, int pad_l, hipStream_t st, int dt, unsigned* amax) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if (amax && dt != SMAAT_F32) return -2;
    if (!upr_enabled() || (Wo & 3) != 0 || (out_bs & 3) != 0 || ((((uintptr_t)out) & am) != 0) || H < 1 || W < 1)
        return -2;
    const UprGeom g = upr_geom(N, C, H, W, Ho, Wo, pad_t, pad_l, Wo / 4, Ho, 64, 500000L);
    if (g.total > (1L << 31) * 256L) return -2;
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_upsample2x_fwd_rows<T>, dim3((unsigned)((g.total + 255) / 256)), dim3(256), 0, st, (const T*)x,
                           x_bs, (T*)out, out_bs, g, amax););
    return (int)hipGetLastError();
}

int launch_upsample2x_bwd_rows(const void* dout, long dout_bs, void* dx, long dx_bs, int N, int C, int H, int W,
                               int Ho, int Wo, int pad_t, int pad_l, hipStream_t st, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if (!upr_enabled() || (W & 1) != 0 || (Wo & 3) != 0 || (pad_l & 3) != 0 || (dout_bs & 3) != 0 ||
        ((((uintptr_t)dout) & am) != 0) || pad_t < 0 || pad_l < 0 || 2 * H + pad_t > Ho || 2 * W + pad_l > Wo)
        return -2;
    const UprGeom g = upr_geom(N, C, H, W, Ho, Wo, pad_t, pad_l, W / 2 + 1, H, 16, 2000000L);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_upsample2x_bwd_rows<T>, dim3((unsigned)((g.total + 255) / 256)), dim3(256), 0, st, (const T*)dout,
                           dout_bs, (T*)dx, dx_bs, g););
    return (int)hipGetLastError();
}
