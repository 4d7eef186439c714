// Fused DepthwiseSeparableConv forward as a ROW-WALKING kernel (round 4; the plane-dominated layers, Cout <= 64):
//
//   z[n][m][p] = b_pw[m] + sum_k W_pw[m][k] * y[n][k][p],     y = depthwise3x3(act(x)) (+ b_dw),   k = 2 ci + j
//   (reference models/layers.py:47-50; act = the previous BatchNorm + ReLU applied on load, optional)
//   + the per-channel BatchNorm partials (mean, M2, count) of z - b_pw for the train-mode BatchNorm behind it
//     (models/unet_parts_depthwise_separable.py:25,34)
//
// k_dsconv_split (dsconv_split.hip) walks 4 x 32 / 8 x 16 pixel tiles: every tile stages its own halo (1.9x the input
// through L2), runs a prologue and an epilogue, and takes ~18 us for an 8-chunk tile (profiles/r2).  Here the structure of
// the recompute weight gradient (dswgrad.hip) is used for the forward:
//   * a workgroup walks DOWN a band of rows of one 32-column strip of one image, one row (32 pixels, ALL K) per iteration;
//   * 8 producer waves: thread (channel ci [+ 64], 4 pixels) keeps its 3 x 6 window of act(x) in registers (one
//     global_load_dwordx4 / dwordx2 + one edge element per row, neighbours' columns by DPP), forms y for its 2 (4) k-rows
//     x 4 pixels (tap order of k_dw3x3_fwd_rows), splits exactly into three bf16 terms (f32 storage) or rounds to bf16
//     (bf16 storage: one MFMA per product) and writes the B image [plane][pixel][K] -- conflict-free (swizzled);
//   * 4 consumer waves: the pointwise weight is CONSTANT over the walk, so its MFMA A fragments live in REGISTERS for the
//     whole kernel (loaded once from the pre-split planes): no A traffic at all.  Wave (wm, wkh) multiplies 32-row tile wm
//     over half wkh of the contraction; the two halves are added through a 16 KB LDS exchange that is read one iteration
//     later (no extra barrier), each wave finishing 16 of its tile's 32 rows: bias, row stores (32 lanes = 128 / 64
//     contiguous bytes), BatchNorm partials accumulated in registers over the whole band (one (mean, M2, count) slot per
//     band x strip x image; sums about the lane's first value, merged across lanes with the equal-count Chan update);
//   * all global loads inline asm with counted s_waitcnt, ONE barrier per row.
// HBM traffic: x once (+ 1/4 for the strip's edge columns, L2) and z once -- 4 (Cin + Cout) HW per image, the north-star's
// fused minimum -- and no depthwise tensor: the weight gradient recomputes it (dswgrad.hip).
// TX / TZ: storage types of x and z (float | bf16); NT = 3 (exact three-term bf16 split, f32-class error), 1 (bf16 operands) or
// 2 (round 6: two-term fp16 split, three MFMAs per product, as the other split GEMMs of the step).  The fp16 split needs a
// power-of-two scale that brings y into fp16's range BEFORE y exists; it comes from an a-priori bound
//     |y[k]| <= sum_taps |w_dw[k][tap]| * A[k / 2] + |b_dw[k]|,   A[c] = max |act(x[c])| <= max(0, |in_scale[c]| max|x| + in_shift[c])
// with max |x| taken from the amax buffer(s) the kernels that wrote x left (common.h), or -- x the output of the previous
// pointwise convolution, x = W' u + b' -- bounded per channel by sum_k |W'[c][k]| max|u| + |b'[c]| from the maximum of ITS operand.  The split is a FLOATING-point one
// (h = rn16(t), g = rn16(t - h)): a bound that is 2^L too large costs nothing until t falls below fp16's normal range,
// i.e. for values more than 2^(28 - L) below the bound -- their absolute error is then 2^-25 of the scaled unit, 2^-39 of the bound.
#include "common.h"
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
int split_mode();  // splitmma.hip

// timing ablations for experiment builds (hipcc -DDSR_DBG=<bits>; results are wrong, only the time means something; compile-time
// for the reason given in dswgrad.hip): 1 no LDS reads + MFMA, 2 no B-image writes, 4 no depthwise math, 8 no global loads,
// 16 no per-chunk barrier, 32 no output stores, 64 no BatchNorm partials
#ifndef DSR_DBG
#define DSR_DBG 0
#endif
#define DSR_CW 32  // pixels per row segment = width of a column strip

#include "rows_args.h"

__device__ __forceinline__ unsigned dsr_fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float dsr_bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ unsigned dsr_pack_hi16(float lo, float hi) {
    return __builtin_amdgcn_perm(dsr_fbits(hi), dsr_fbits(lo), 0x07060302u);  // {hi.hi16, lo.hi16}
}
__device__ __forceinline__ const void* dsr_uniform_ptr(const void* p) {  // (an "s" asm constraint does not make a value uniform)
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
typedef float dsr_f32x2_t __attribute__((ext_vector_type(2)));
// one value pair (k = 2 ci, 2 ci + 1 of one pixel) -> NT dwords {hi: k + 1, lo: k}
typedef _Float16 dsr_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dsr_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dsr_pack_f16(float a, float b) {  // one v_cvt_pk_f16_f32 (round to nearest even)
    const dsr_f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dsr_f16x2));
}
template <int NT>
__device__ __forceinline__ void dsr_split_pair(float a, float b, unsigned (&out)[NT], float sc) {
    if constexpr (NT == 2) {  // two-term fp16 split of the scaled values (splitmma.hip split2_f16)
        const float ta = a * sc, tb = b * sc;
        out[0] = dsr_pack_f16(ta, tb);
        const dsr_f16x2 hv = __builtin_bit_cast(dsr_f16x2, out[0]);
        out[1] = dsr_pack_f16(ta - (float)hv.x, tb - (float)hv.y);  // (the residual of a rounding to fewer bits is exact in f32)
    } else if (NT == 1) {
        out[0] = pack_bf16x2(a, b);  // round to nearest even
    } else {
        const float a1 = dsr_bitsf(dsr_fbits(a) & 0xFFFF0000u), b1 = dsr_bitsf(dsr_fbits(b) & 0xFFFF0000u);
        const float ra = a - a1, rb = b - b1;  // exact
        const float a2 = dsr_bitsf(dsr_fbits(ra) & 0xFFFF0000u), b2 = dsr_bitsf(dsr_fbits(rb) & 0xFFFF0000u);
        out[0] = dsr_pack_hi16(a1, b1);
        if constexpr (NT == 3) {
            out[1] = dsr_pack_hi16(a2, b2);
            out[2] = dsr_pack_hi16(ra - a2, rb - b2);  // exact, <= 8 significant bits
        }
    }
}
// raw registers of one 4-pixel row piece (ext vector types: inline-asm outputs) and their f32 values
typedef unsigned dsr_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct DsrRaw;
template <> struct DsrRaw<float> { typedef f32x4 type; };
template <> struct DsrRaw<bf16_t> { typedef dsr_u32x2 type; };
__device__ __forceinline__ void dsr_vals(const f32x4 v, float (&m)[4]) {
    m[0] = v[0]; m[1] = v[1]; m[2] = v[2]; m[3] = v[3];
}
__device__ __forceinline__ void dsr_vals(const dsr_u32x2 v, float (&m)[4]) {
    m[0] = bf16_lo(v[0]); m[1] = bf16_hi(v[0]); m[2] = bf16_lo(v[1]); m[3] = bf16_hi(v[1]);
}
__device__ __forceinline__ void dsr_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void dsr_store(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu); }

// CPT: channels per producer thread (1: Cin <= 64, 2: Cin <= 128); KS = K / 16 contraction steps (<= 8 * CPT)
typedef float dsr_f32x2 __attribute__((ext_vector_type(2)));

// PK: depthwise stage on packed f32 math (experiment switch SMAAT_DWG_PK=1, bf16-storage instantiations only)
template <int NT, bool AFF, int CPT, typename TX, typename TZ, bool PK = false>
__global__ __launch_bounds__(768) void k_dsconv_rows_fwd(const DsRowsArgs a) {
    constexpr int KMAX = 128 * CPT;          // k rows the B image holds
    constexpr int ROWB = KMAX * 2 + 16;      // bytes per pixel row of the B image (dword stride = 4 mod 64: conflict-free b128)
    constexpr int BPL = DSR_CW * ROWB;       // bytes per plane
    constexpr int BUFSZ = NT * BPL;
    constexpr int KSH = 4 * CPT;             // contraction steps per consumer wave (half of KMAX / 16)
    constexpr bool A3L = NT == 3 && CPT == 2;  // third weight plane in LDS (see the consumer prologue)
    // rows in flight per producer thread: bf16 halves the bytes per row (dswgrad.hip), two channels per thread double them again.
    // (Eight sets of two channels are 48 VGPRs of destinations: hipcc then SPILLS prefetched registers -- a scratch store of a
    // register whose load has not landed, reloaded later as if it held the row: caught by scripts/isa_hazards.py, round 6)
    constexpr int PD = (sizeof(TX) == 2 && CPT == 1) ? 8 : ((NT != 3 && CPT == 2 && sizeof(TX) == 4) ? 3 : 4);  // (bf16- / fp16-operand f32-x builds with two channels: a fourth set spills)
    constexpr int LPG = 2 * CPT;             // loads per group and producer thread: row piece + edge element per channel
    static_assert((PD - 1) * LPG <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* X = (float*)(lds + 2 * BUFSZ);    // [2][4 waves][8][64]: partner halves of the accumulators
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int l31 = lane & 31, half = lane >> 5;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int split = xcd * (a.nsplit >> 3) + idx;  // contiguous item ranges per XCD (neighbouring strips share an L2)
    // item j of this workgroup = it_lo + j * it_st (j < nitems): contiguous ranges, or (ilv, the default) the items of the XCD's
    // range dealt round-robin to its workgroups -- neighbouring strips then run at the same time on one XCD, so the halves of a
    // bf16 line / the neighbouring lines of a DRAM page are touched together (dswgrad.hip, measured there)
    const int s8 = a.nsplit >> 3;
    int it_lo, it_st, nitems;
    if (a.ilv) {
        const int xlo = xcd * s8 * a.ips;
        int xhi = xlo + s8 * a.ips;
        if (xhi > a.items) xhi = a.items;
        it_lo = xlo + idx;
        it_st = s8;
        nitems = it_lo < xhi ? (xhi - it_lo + s8 - 1) / s8 : 0;
    } else {
        it_lo = split * a.ips;
        it_st = 1;
        int hi = it_lo + a.ips;
        if (hi > a.items) hi = a.items;
        nitems = hi > it_lo ? hi - it_lo : 0;
    }
    const int it_hi = it_lo + nitems * it_st;  // (exclusive bound of the walk)
    const int bps = a.bands * a.strips;
    auto item_rows = [&](int item) {
        const int band = (item % bps) / a.strips;
        const int r0 = band * a.RB;
        return (a.H - r0 < a.RB ? a.H - r0 : a.RB);
    };
    int total = 0;
    for (int i = 0; i < nitems; ++i) total += item_rows(it_lo + i * it_st) + 2;  // rows + 2 priming iterations per item
    const int total_pad = (total + PD - 1) / PD * PD;

    // NT == 2: scale exponents of the two operands.  ka: the fp16 weight image's (its trailer); ky: from the a-priori bound of
    // |y| (file comment), formed by every wave for itself -- lane l takes the k-rows l, l + 64, ... -- so that the producers'
    // scale and the consumers' epilogue factor are the same number.  1.0001: the f32 roundings of the
    // bound and of the FMAs that form y (y <= bound (1 + 11 eps)); a scaled bound below 2^15 leaves fp16 a factor 2 beyond that.
    int ky = 0, ka = 0;
    if constexpr (NT == 2) {
        unsigned am = amax_read(a.x_amax);
        if (a.x_amax2) {
            const unsigned am2 = amax_read(a.x_amax2);
            am = am2 > am ? am2 : am;
        }
        const float xmax = dsr_bitsf(am);
        // x given as the output of the previous pointwise convolution (a.zb_w): per-channel bounds of |x| through LDS -- the 12
        // waves share the rows of zb_w (coalesced reads, one wave sum per row); the B image's first buffer is free until the
        // first commit, hence the second barrier.  Both barriers are executed by every wave (wave-uniform condition).
        float* zbnd = (float*)lds;
        if (a.zb_w) {
            for (int c = wv; c < a.Cin; c += 12) {
                float sw = 0.f;
                for (int k = lane; k < a.zb_K; k += 64) sw += fabsf(a.zb_w[(long)c * a.zb_K + k]);
                sw = wave_sum_all(sw);
                if (lane == 0) zbnd[c] = fmaf(sw, xmax, a.zb_b ? fabsf(a.zb_b[c]) : 0.f) * 1.0001f;
            }
            __syncthreads();
        }
        float bnd = 0.f;
        for (int k = lane; k < a.K; k += 64) {
            const int c = k >> 1;
            const float xb = a.zb_w ? zbnd[c] : xmax;
            const float A = AFF ? fmaxf(fmaf(fabsf(a.in_scale[c]), xb, a.in_shift[c]), 0.f) : xb;
            float sw = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) sw += fabsf(a.w_dw[k * 9 + t]);
            bnd = fmaxf(bnd, fmaf(sw, A, a.b_dw ? fabsf(a.b_dw[k]) : 0.f));
        }
        if (a.zb_w) __syncthreads();
        bnd = wave_max_all(bnd) * 1.0001f;
        ky = __builtin_amdgcn_readfirstlane(f16_kexp(dsr_fbits(bnd)));
        ka = __builtin_amdgcn_readfirstlane(*a.a_kexp);
    }

    if (producer) {
        const int ptid = tid - 256;
        const float ysc = pow2i(ky);
        const int ci = ptid >> 3, g = ptid & 7;
        bool cv[CPT];
        int cgc[CPT];
        float wt[CPT][2][9], bs[CPT][2], asc[CPT], ash[CPT];
        unsigned vo_x[CPT], vo_e[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int cg = ci + 64 * u;
            cv[u] = cg < a.Cin;
            cgc[u] = cv[u] ? cg : a.Cin - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int k = 0; k < 9; ++k) wt[u][j][k] = cv[u] ? a.w_dw[(cgc[u] * 2 + j) * 9 + k] : 0.f;
                bs[u][j] = (cv[u] && a.b_dw) ? a.b_dw[cgc[u] * 2 + j] : 0.f;  // channels beyond Cin: y = 0 exactly
            }
            asc[u] = AFF ? a.in_scale[cgc[u]] : 1.f;
            ash[u] = AFF ? a.in_shift[cgc[u]] : 0.f;
            vo_x[u] = (unsigned)(cgc[u] * a.P + 4 * g) * (unsigned)sizeof(TX);
            vo_e[u] = vo_x[u];
        }
        // (complete the compiler-visible loads before the first inline-asm load: a load still pending at the loop entry makes
        // hipcc drain vmcnt to 0 at the value's first use inside the loop, every iteration -- see dswgrad.hip)
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(wt[u][j][k]));
                asm volatile("" : "+v"(bs[u][j]));
            }
            asm volatile("" : "+v"(asc[u]), "+v"(ash[u]));
        }
        // issue cursor (wave-uniform)
        int w_item = it_lo - it_st, w_j = 0, w_len = 0, w_r0 = 0, w_eback = 0;
        const TX* w_xb = (const TX*)a.x;
        bool w_lok = false, w_rok = false;
        auto advance = [&]() __attribute__((always_inline)) {
            ++w_j;
            if (w_j >= w_len) {
                if (w_item + it_st < it_hi) {
                    w_item += it_st;
                    w_j = 0;
                    const int n = w_item / bps, rem = w_item - n * bps;
                    const int band = rem / a.strips, st_ = rem - band * a.strips;
                    w_r0 = band * a.RB;
                    const int c0 = st_ * DSR_CW;
                    w_xb = (const TX*)a.x + (long)n * a.x_bs + c0;
                    w_len = (a.H - w_r0 < a.RB ? a.H - w_r0 : a.RB) + 2;
                    w_eback = __builtin_amdgcn_readfirstlane(c0 > 0 ? 1 : 0);
                    w_lok = w_eback != 0;
                    w_rok = c0 + DSR_CW < a.W;
                    // edge element: scalar base one element back when a column exists left of the strip (lane offsets of a
                    // scalar-base load are unsigned); g = 0 reads column c0 - 1, g = 7 column c0 + 32 (or its own last one)
                    const unsigned eo = (g == 0 ? 0u : ((w_lok ? 1u : 0u) + (g == 7 ? (w_rok ? 4u : 3u) : 0u))) * (unsigned)sizeof(TX);
#pragma unroll
                    for (int u = 0; u < CPT; ++u) vo_e[u] = vo_x[u] + eo;
                } else {
                    w_j = w_len - 1;  // past the end: keep re-loading the last row (never consumed)
                }
            }
        };
        typename DsrRaw<TX>::type sx[PD][CPT];
        unsigned se[PD][CPT];
        int srow[PD];
        bool slok[PD], srok[PD];
        auto issue = [&](int set) __attribute__((always_inline)) {
            advance();
            const int xr = w_r0 - 1 + w_j;
            const int xrc = xr < 0 ? 0 : (xr >= a.H ? a.H - 1 : xr);
            const TX* xrow = (const TX*)dsr_uniform_ptr(w_xb + (long)xrc * a.W);
            const TX* erow = (const TX*)dsr_uniform_ptr(xrow - w_eback);
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                if constexpr ((DSR_DBG & 8) != 0) {
                    asm volatile("" : "+s"(xrow), "+s"(erow));
                } else if (sizeof(TX) == 4) {
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sx[set][u]) : "v"(vo_x[u]), "s"(xrow));
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(se[set][u]) : "v"(vo_e[u]), "s"(erow));
                } else {
                    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(sx[set][u]) : "v"(vo_x[u]), "s"(xrow));
                    asm volatile("global_load_ushort %0, %1, %2" : "=v"(se[set][u]) : "v"(vo_e[u]), "s"(erow));
                }
            }
            srow[set] = xr;
            slok[set] = w_lok;
            srok[set] = w_rok;
        };
        auto wait_set = [&](int set) __attribute__((always_inline)) {
            if constexpr ((DSR_DBG & 8) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * LPG) : "memory");
#pragma unroll
            for (int u = 0; u < CPT; ++u) asm volatile("" : "+v"(sx[set][u]), "+v"(se[set][u]));
        };
        float win[CPT][3][6];
#pragma unroll
        for (int u = 0; u < CPT; ++u)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) win[u][r][c] = 0.f;
        int c_j = 0, c_len = 0, c_item = it_lo - it_st;
        float yam = 0.f;  // running max |y| of this thread (published once, at the end of the walk, when a.y_amax is given)
        // B image address of this thread: pixel 4g + i, dword (ci + 64 u) ^ 8 (g >> 2)  (swizzle: conflict-free writes)
        const int bsw = (g >> 2) << 3;
        auto commit = [&](int set, int buf, bool live) __attribute__((always_inline)) {
            ++c_j;
            if (c_j >= c_len) {
                c_item += it_st;
                c_j = 0;
                c_len = item_rows(c_item) + 2;
            }
            const int xr = srow[set];
            const bool rin = xr >= 0 && xr < a.H;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                float m[4];
                dsr_vals(sx[set][u], m);
                float e = sizeof(TX) == 4 ? dsr_bitsf(se[set][u]) : bf16_lo(se[set][u]);
                if (AFF) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) m[c] = fmaxf(fmaf(m[c], asc[u], ash[u]), 0.f);
                    e = fmaxf(fmaf(e, asc[u], ash[u]), 0.f);
                }
                float l = dpp_src<0x111, 0xF>(m[3]);  // row_shr:1 (lane i <- lane i - 1)
                float r = dpp_src<0x101, 0xF>(m[0]);  // row_shl:1 (lane i <- lane i + 1)
                l = g == 0 ? e : l;
                r = g == 7 ? e : r;
                const bool rv = cv[u] && rin;
                const bool lv = rv && (g > 0 || slok[set]), rvv = rv && (g < 7 || srok[set]);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    win[u][0][c] = win[u][1][c];
                    win[u][1][c] = win[u][2][c];
                }
                win[u][2][0] = lv ? l : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) win[u][2][1 + c] = rv ? m[c] : 0.f;
                win[u][2][5] = rvv ? r : 0.f;
            }
            if (c_j < 2 || !live) return;  // priming iteration / surplus iteration of the padded walk (no chunk, no maximum)
            unsigned char* base = lds + buf * BUFSZ;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                float yy[2][4];
                if constexpr ((DSR_DBG & 4) != 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) yy[j][c] = win[u][1][1 + c] + bs[u][j];
                } else if (PK) {  // both k-rows of the channel in one v_pk_fma_f32 (same fma per component, same tap order)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dsr_f32x2 acc = {bs[u][0], bs[u][1]};
#pragma unroll
                        for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                            for (int tc = 0; tc < 3; ++tc) {
                                const dsr_f32x2 w2 = {wt[u][0][tr * 3 + tc], wt[u][1][tr * 3 + tc]};
                                const dsr_f32x2 x2 = {win[u][tr][c + tc], win[u][tr][c + tc]};
                                acc = __builtin_elementwise_fma(w2, x2, acc);
                            }
                        yy[0][c] = acc[0];
                        yy[1][c] = acc[1];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float acc = bs[u][j];  // tap order of k_dw3x3_fwd_rows: bit-identical y
#pragma unroll
                            for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                                for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[u][j][tr * 3 + tc], win[u][tr][c + tc], acc);
                            yy[j][c] = acc;
                        }
                }
                yam = fmaxf(yam, fmaxf(fmaxf(fmaxf(fabsf(yy[0][0]), fabsf(yy[0][1])), fmaxf(fabsf(yy[0][2]), fabsf(yy[0][3]))),
                                       fmaxf(fmaxf(fabsf(yy[1][0]), fabsf(yy[1][1])), fmaxf(fabsf(yy[1][2]), fabsf(yy[1][3])))));
                const int dw_ = ((ci + 64 * u) ^ bsw) * 4;
                if constexpr ((DSR_DBG & 2) != 0) {
                    asm volatile("" ::"v"(yy[0][0]), "v"(yy[0][1]), "v"(yy[0][2]), "v"(yy[0][3]), "v"(yy[1][0]), "v"(yy[1][1]), "v"(yy[1][2]), "v"(yy[1][3]));
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned pl[NT];
                    dsr_split_pair<NT>(yy[0][c], yy[1][c], pl, ysc);
#pragma unroll
                    for (int t = 0; t < NT; ++t) *(unsigned*)(base + t * BPL + (4 * g + c) * ROWB + dw_) = pl[t];
                }
            }
        };
        if (total > 0) {
#pragma unroll
            for (int s_ = 0; s_ < PD; ++s_) issue(s_);
            wait_set(0);
            commit(0, 0, true);
            issue(0);
        }
        __syncthreads();
        // padded walk, every slot unconditional (dswgrad.hip: the shape scripts/isa_hazards.py can prove)
        for (int t0 = 0; t0 < total_pad; t0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                wait_set((u + 1) % PD);
                commit((u + 1) % PD, (t0 + u + 1) & 1, t0 + u + 1 < total);
                issue((u + 1) % PD);
                if constexpr ((DSR_DBG & 16) == 0) __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.y_amax) amax_publish_wave(a.y_amax, yam, (unsigned)(blockIdx.x * 8 + (wv - 4)));  // (wave-uniform branch)
    } else {
        const int wave = wv & 3;
        const int wm = wave & 1, wkh = wave >> 1;
        const int partner = wave ^ 2;
        const int KS = a.K >> 4;  // contraction steps present (K % 16 == 0)
        // ---- A fragments of this wave: rows wm * 32 + l31, steps wkh * KSH .. + KSH, resident for the whole walk ----
        // A3L (exact split with two channels per producer thread, K = 256): 96 registers of fragments beside the accumulators,
        // the statistics and the epilogue values do not fit the 168 VGPRs of three waves per SIMD (the build spilled 30).  The
        // third plane -- the smallest term of the weight split, used by ONE of the six MFMAs of a step -- lives in LDS instead
        // (32 KB, written once by the wave that reads it, conflict-free 16-byte lanes).
        bf16x8 af[KSH][A3L ? 2 : NT];
        unsigned char* a3p = A3L ? lds + 2 * BUFSZ + 2 * 4 * 8 * 64 * 4 + (wave * KSH * 64 + lane) * 16 : nullptr;  // (after X)
        {
            const int m = wm * 32 + l31;
            const int mc = m < a.M ? m : a.M - 1;
#pragma unroll
            for (int s = 0; s < KSH; ++s) {
                const int ks = wkh * KSH + s;
                const int ksc = ks < KS ? ks : KS - 1;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 v = *(const bf16x8*)(a.planes + (((long)ksc * a.npl + t) * a.M + mc) * 16 + half * 8);
                    bf16x8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = 0;
                    if constexpr (A3L) {
                        const bf16x8 fv = (ks < KS && m < a.M) ? v : z;
                        if (t == 2)
                            *(bf16x8*)(a3p + s * 64 * 16) = fv;
                        else
                            af[s][t < 2 ? t : 0] = fv;
                    } else {
                        af[s][t] = (ks < KS && m < a.M) ? v : z;
                    }
                }
            }
        }
        // rows this wave finishes: registers r = 8 wkh .. 8 wkh + 7 of its tile  ->  m = wm * 32 + (r & 3) + 8 (r >> 2) + 4 half
        float bias[8];
        int mrow[8];
        // output addressing: row mrow[i] = (wm * 32 + 4 half) + rsub(i), rsub(i) = (r & 3) + 8 (r >> 2) wave-uniform.  The lane part
        // is ONE 32-bit element offset, the row part a scalar pointer per store (SALU) -- eight per-lane 64-bit addresses cost
        // 16 VGPRs and a VALU add each
        const unsigned loff0 = (unsigned)(wm * 32 + 4 * half) * (unsigned)a.P + (unsigned)l31;
        auto rsub = [&](int i) { const int r = 8 * wkh + i; return (r & 3) + 8 * (r >> 2); };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mrow[i] = wm * 32 + rsub(i) + 4 * half;
            bias[i] = (a.bias && mrow[i] < a.M) ? a.bias[mrow[i] < a.M ? mrow[i] : 0] : 0.f;
        }
        f32x16 acc;
        float keep[8];
        float zam = 0.f;
        const float e1 = pow2i(-((ka + ky) / 2)), e2 = pow2i(-((ka + ky) - (ka + ky) / 2));
        float s1[8], s2[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s1[i] = s2[i] = sh[i] = 0.f;
        TZ* c_ob = (TZ*)a.out;  // output element (channel 0, first row of the band, first column of the strip) of the open item
        TZ* p_op = (TZ*)a.out;  // ... of the pending chunk's row
        int nrows = 0;            // rows accumulated into the statistics of the open item
        // consume cursor + the chunk whose result is pending (finished one iteration later)
        int c_item = it_lo - it_st, c_j = 0, c_len = 0;
        bool pend = false;
        int p_buf = 0;
        const int swz = ((l31 >> 4) & 1) << 5;  // byte XOR of the B image's k index for pixels 16 .. 31 (8 dwords)
        auto flush_stats = [&](int item) __attribute__((always_inline)) {
            if ((DSR_DBG & 64) != 0 || !a.part || nrows == 0) return;
            const float fn = (float)nrows, inv = 1.f / fn;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // this lane: nrows samples of row mrow[i] in column l31, about the shift sh[i]
                const float mean_l = fmaf(s1[i], inv, sh[i]);
                const float m2_l = fmaxf(fmaf(-s1[i] * inv, s1[i], s2[i]), 0.f);
                float sm = mean_l, sq = m2_l;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) {  // the 32 lanes of a half hold the 32 columns (equal counts)
                    sm += __shfl_xor(sm, o, 64);
                    sq += __shfl_xor(sq, o, 64);
                }
                const float mean = sm * (1.f / 32.f);
                const float d = mean_l - mean;
                float sd = d * d;
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) sd += __shfl_xor(sd, o, 64);
                if (l31 == 0 && mrow[i] < a.M) {
                    a.part[((long)0 * a.items + item) * a.M + mrow[i]] = mean;
                    a.part[((long)1 * a.items + item) * a.M + mrow[i]] = fmaf(fn, sd, sq);
                    a.part[((long)2 * a.items + item) * a.M + mrow[i]] = fn * 32.f;
                }
            }
            nrows = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
        };
        // Round 4 (ablation builds): the epilogue was the critical path of the whole workgroup -- eight LDS reads each followed by
        // lgkmcnt(0) and an exec-masked store (the row guard made every row its own basic block), and three scalar divisions per
        // chunk for the output address.  Now: the eight partner values are read first (one wait), the row guard is a wave-uniform
        // branch around the whole store group (M == 64: no guard), the band's output pointer is cached per item.
        auto finish = [&]() __attribute__((always_inline)) {  // the pending chunk: add the partner half, store, statistics
            const float* xp = X + ((p_buf * 4 + partner) * 8) * 64 + lane;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = xp[i * 64];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += keep[i];
            if constexpr (NT == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * e1 * e2;  // (2^-(ka + ky) in two normal factors)
            }
            if (a.z_amax) {  // (wave-uniform) max |z| as stored: the scale bound of a row-walking forward that reads this z
#pragma unroll
                for (int i = 0; i < 8; ++i) zam = fmaxf(zam, mrow[i] < a.M ? fabsf(v[i] + bias[i]) : 0.f);
            }
            if constexpr ((DSR_DBG & 32) == 0) {
                if (a.M == 64 && !a.relu) {  // (the training form: no guard, no clamp)
#pragma unroll
                    for (int i = 0; i < 8; ++i) dsr_store(p_op + (long)rsub(i) * a.P + loff0, v[i] + bias[i]);
                } else if (a.M == 64) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float t = v[i] + bias[i];
                        dsr_store(p_op + (long)rsub(i) * a.P + loff0, t < 0.f ? 0.f : t);
                    }
                } else {
                    const float fl = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (mrow[i] < a.M) {  // (a select, not fmaxf: a NaN must come out as a NaN for every channel count)
                            const float t = v[i] + bias[i];
                            dsr_store(p_op + (long)rsub(i) * a.P + loff0, t < fl ? fl : t);
                        }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(v[i]));
            }
            if ((DSR_DBG & 64) == 0 && a.part) {
                const bool first = nrows == 0;  // (s1 = s2 = 0 then: flush_stats)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float sv = first ? v[i] : sh[i];
                    sh[i] = sv;
                    const float d = v[i] - sv;
                    s1[i] += d;
                    s2[i] = fmaf(d, d, s2[i]);
                }
            }
            ++nrows;
        };
        __syncthreads();
        for (int t = 0; t < total; ++t) {
            ++c_j;
            bool newitem = false;
            if (c_j >= c_len) {
                c_item += it_st;
                c_j = 0;
                const int n = c_item / bps, rem = c_item - n * bps;
                const int band = rem / a.strips, st_ = rem - band * a.strips;
                const int r0 = band * a.RB;
                c_len = (a.H - r0 < a.RB ? a.H - r0 : a.RB) + 2;
                c_ob = (TZ*)a.out + (long)n * a.out_bs + (long)r0 * a.W + st_ * DSR_CW;
                newitem = true;
            }
            if (pend) {
                finish();
                pend = false;
            }
            if (newitem && c_item > it_lo) flush_stats(c_item - it_st);  // (its last chunk was finished just above)
            if (c_j >= 2 && (DSR_DBG & 1) != 0) {
                pend = true;
                p_buf = t & 1;
                p_op = c_ob + (long)(c_j - 2) * a.W;
            } else if (c_j >= 2) {
                const unsigned char* base = lds + (t & 1) * BUFSZ + l31 * ROWB;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KSH; ++s) {
                    const int ks = wkh * KSH + s;
                    bf16x8 bf[NT];
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        bf[tt] = *(const bf16x8*)(base + tt * BPL + ((ks * 32 + half * 16) ^ swz));
                    if constexpr (A3L) {  // (the same six products in the same order; the third plane comes from LDS)
                        const bf16x8 a_lo = *(const bf16x8*)(a3p + s * 64 * 16);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], bf[2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bf[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][1], bf[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], bf[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][1], bf[0], acc, 0, 0, 0);
                    } else if constexpr (NT == 2) {  // two-term fp16 split: h g' + g h' (+ h h' below); the g g' term is dropped
                        const dsr_f16x8 a0 = __builtin_bit_cast(dsr_f16x8, af[s][0]), a1 = __builtin_bit_cast(dsr_f16x8, af[s][1]);
                        const dsr_f16x8 b0 = __builtin_bit_cast(dsr_f16x8, bf[0]), b1 = __builtin_bit_cast(dsr_f16x8, bf[1]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
                    } else if (NT == 3) {  // smallest terms first (the order of the other split GEMMs)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], bf[NT - 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][NT - 1], bf[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][NT / 2], bf[NT / 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], bf[NT / 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][NT / 2], bf[0], acc, 0, 0, 0);
                    }
                    if constexpr (NT != 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], bf[0], acc, 0, 0, 0);
                }
                // hand the partner its half, keep ours
                float* xw = X + (((t & 1) * 4 + wave) * 8) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xw[i * 64] = acc[8 * (1 - wkh) + i];
                    keep[i] = acc[8 * wkh + i];
                }
                pend = true;
                p_buf = t & 1;
                p_op = c_ob + (long)(c_j - 2) * a.W;
            }
            if constexpr ((DSR_DBG & 16) == 0) __syncthreads();
        }
        if (pend) finish();
        if (nitems > 0) flush_stats(it_hi - it_st);
        if (a.z_amax) amax_publish_wave(a.z_amax, zam, (unsigned)(blockIdx.x * 4 + wave));
        if constexpr ((DSR_DBG & 16) == 0)
            for (int t = total; t < total_pad; ++t) __syncthreads();  // the barriers of the producers' surplus iterations
    }
}

// ---------------------------------------------------------------------------------------------------------------
static void dsr_geom(DsRowsArgs& a) {
    a.P = a.H * a.W;
    a.strips = a.W / DSR_CW;
    // band length: every band costs two priming iterations, and a workgroup walks ceil(items / workgroups) items -- pick the
    // divisor of H (12 .. 64 rows) that minimises the iterations of the busiest workgroup (288 rows at batch 32: 36-row
    // bands = 9 items per workgroup exactly, 342 iterations; 48-row bands would leave 7 / 6 items, 350)
    const int wgs = 256;
    int rb = a.H > 64 ? 32 : a.H;
    long best = -1;
    for (int cand = 64; cand >= 12; --cand) {
        if (a.H % cand) continue;
        const long items = (long)a.N * (a.H / cand) * a.strips;
        const long cost = ((items + wgs - 1) / wgs) * (cand + 2);
        if (best < 0 || cost < best) {
            best = cost;
            rb = cand;
        }
    }
    a.RB = rb;
    a.bands = (a.H + rb - 1) / rb;
    a.items = a.N * a.bands * a.strips;
    int ns = 256;  // one workgroup per CU
    while (ns > 8 && ns > a.items) ns -= 8;
    a.nsplit = ns;
    a.ips = (a.items + ns - 1) / ns;
}

// shapes this kernel takes: kernels_per_layer 2, W % 32 == 0, Cout <= 64, Cin % 8 == 0 and Cin <= 128
int dsconv_rows_ok(int kpl, int Cin, int M, int H, int W) {
    return kpl == 2 && (W % DSR_CW) == 0 && M <= 64 && M >= 1 && (Cin & 7) == 0 && Cin >= 8 && Cin <= 128 && H >= 1;
}
// BatchNorm partial slots ( = items: band x strip x image)
int dsconv_rows_num_slots(int N, int H, int W) {
    DsRowsArgs a{};
    a.N = N; a.H = H; a.W = W;
    if (W % DSR_CW) return 0;
    dsr_geom(a);
    return a.items;
}

template <int NT, bool AFF, int CPT, typename TX, typename TZ, bool PK = false>
static int launch_dsr_cfg(const DsRowsArgs& a, hipStream_t st) {
    constexpr int ROWB = 128 * CPT * 2 + 16;
    const size_t lds = (size_t)2 * NT * DSR_CW * ROWB + (size_t)2 * 4 * 8 * 64 * sizeof(float) +
                       ((NT == 3 && CPT == 2) ? (size_t)4 * (4 * CPT) * 64 * 16 : 0);  // + the third weight plane (A3L)
    constexpr auto kern = k_dsconv_rows_fwd<NT, AFF, CPT, TX, TZ, PK>;
    static size_t granted = 0;
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    hipLaunchKernelGGL(kern, dim3(a.nsplit), dim3(768), lds, st, a);
    return (int)hipGetLastError();
}

template <int NT, typename TX, typename TZ, bool PK = false>
static int launch_dsr_sel(const DsRowsArgs& a, hipStream_t st) {
    const bool aff = a.in_scale != nullptr;
    if (a.Cin <= 64) return aff ? launch_dsr_cfg<NT, true, 1, TX, TZ, PK>(a, st) : launch_dsr_cfg<NT, false, 1, TX, TZ, PK>(a, st);
    return aff ? launch_dsr_cfg<NT, true, 2, TX, TZ, PK>(a, st) : launch_dsr_cfg<NT, false, 2, TX, TZ, PK>(a, st);
}

// x_dt / z_dt: SMAAT_F32 | SMAAT_BF16.  f32 storage: planes = the three split planes (or plane 0 only in bf16-operand
// mode), npl = 3 -- or, with a.x_amax / a.a_kexp, the fp16 image (npl = 2, two-term split); bf16 storage: planes = the bf16 image of smaat_bf16_planes ([K/16][M][16]), npl = 1, one MFMA per product.
// -2: shape / alignment / type combination not handled.
int launch_dsconv_rows(DsRowsArgs& a, int kpl, int x_dt, int z_dt, hipStream_t st) {
    if (!dsconv_rows_ok(kpl, a.Cin, a.M, a.H, a.W) || a.K != 2 * a.Cin) return -2;
    const int xe = x_dt == SMAAT_BF16 ? 2 : 4;
    if ((a.x_bs & 3) || ((((uintptr_t)a.x) * 1) & (4 * xe - 1)) || (((uintptr_t)a.planes) & 15)) return -2;
    if ((long)a.Cin * a.H * a.W * xe >= (1L << 32) || (long)a.M * a.H * a.W >= (1L << 31)) return -2;  // (32-bit element offsets)
    dsr_geom(a);
    {
        static int ilv = -1;  // SMAAT_ROWS_ILV=0: contiguous item ranges (A/B timing)
        if (ilv < 0) {
            const char* e = getenv("SMAAT_ROWS_ILV");
            ilv = e ? atoi(e) : 1;
        }
        a.ilv = ilv;
    }
    if (z_dt == SMAAT_BF16) {  // mixed precision: bf16 z, bf16 operands; x f32 (the stem) or bf16
        a.npl = 1;
        static int pk = -1;
        if (pk < 0) {
            const char* e = getenv("SMAAT_DWG_PK");
            pk = (e && e[0] == '1') ? 1 : 0;
        }
        if (x_dt == SMAAT_BF16) return pk ? launch_dsr_sel<1, bf16_t, bf16_t, true>(a, st) : launch_dsr_sel<1, bf16_t, bf16_t>(a, st);
        return launch_dsr_sel<1, float, bf16_t>(a, st);
    }
    if (x_dt != SMAAT_F32) return -2;
    if (a.x_amax || a.a_kexp) {  // two-term fp16 split: an fp16 weight image + the maximum of x
        if (!a.x_amax || !a.a_kexp) return -1;
        a.npl = 2;
        return launch_dsr_sel<2, float, float>(a, st);  // (hipcc packs the depthwise FMAs of this build into v_pk_fma_f32 by itself)
    }
    a.npl = 3;
    if (split_mode() == 1) return launch_dsr_sel<1, float, float>(a, st);
    return launch_dsr_sel<3, float, float>(a, st);
}
