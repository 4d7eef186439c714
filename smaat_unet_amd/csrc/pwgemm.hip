// Pointwise (1x1) convolution family on the f32 MFMA pipe of gfx950, with the
// depthwise 3x3 stage fused in front of it through LDS.
//
//   k_pwgemm<DW=true>   z[n][m][p] = sum_k wt[k][m] * dw3x3(x)[n][k][p] + bias[m]
//                       (DepthwiseSeparableConv.forward, reference models/layers.py:47-50)
//                       + per-tile BatchNorm partial statistics of the accumulators.
//   k_pwgemm<DW=false>  out[n][m][p] = sum_c wt[c][m] * in[n][c][p] + bias[m]
//                       (OutConv, reference models/unet_parts.py:72; and the data-gradient
//                        dY = W^T dZ of the pointwise conv, with wt = w_pw in natural layout)
//   k_wgrad<DW=true>    dW_pw[m][k] = sum_{n,p} dz[n][m][p] * dw3x3(x)[n][k][p]   (Y recomputed)
//   k_wgrad<DW=false>   dW[m][k]    = sum_{n,p} dz[n][m][p] * in[n][k][p]
//
// Mapping: v_mfma_f32_32x32x2_f32, A = weights (rows = output channel), B = pixels
// (cols), so that for a fixed accumulator register the 32 lanes of a half-wave hold 32
// consecutive pixels of one output channel -> every global store is a full 128 B row.
#include "common.h"
#include <cstdlib>
#include <stdlib.h>

#define KC 16       // contraction rows staged per chunk
#define SMAX 768    // staged floats per input channel (3 per thread)
#define SMAXW 512   // same for the weight-gradient kernel (64-pixel sub tiles, 2 per thread)
#define PSW 64      // pixels per sub tile in the weight-gradient kernel
#ifndef DWUNROLL
#define DWUNROLL 2
#endif
#ifndef TAPS_SMEM
#define TAPS_SMEM 0  // depthwise taps: 1 = scalar loads (constant address space), 0 = staged through LDS (measured faster)
#endif
typedef const float __attribute__((address_space(4))) cfloat;
#define SMAX_WS 512 // staged floats per input channel in the wave-specialised kernel (fixed stride)



// MODE 0: B operand rows are loaded straight from global (plain pointwise conv / dgrad)
// MODE 1,2,4: B operand rows are produced by the depthwise 3x3 stage, kpl = MODE
//
// Software pipeline per 16-row chunk (T14 "issue early / write late"): the global loads of
// chunk c+1 (input halo tile or B rows, and the weight slab) are issued into registers
// right before the MFMA block of chunk c and written to LDS after it.
// TR (plain pointwise, flattened pixel tiles, P % 4 == 0, no partial statistics): the MFMA operands are
// swapped (A = pixels, B = output channels), so that a lane's accumulator registers hold 4 CONSECUTIVE
// pixels of one channel and the epilogue is 16-byte stores (4x fewer store instructions: the short-
// contraction data-gradient GEMMs of the plane-dominated layers are epilogue/store-issue bound).
template <int WCO, int CT, int WPX, int PXT, int MODE, bool TR = false>
__global__ __launch_bounds__(SMAAT_THREADS, 2) void k_pwgemm(const PwArgs a) {
    constexpr bool DW = MODE > 0;
    constexpr int KPL = DW ? MODE : 1;
    constexpr int KCI = KC / KPL;  // input channels per chunk (DW)
    constexpr int COT = WCO * CT * 32;
    constexpr int PT = WPX * PXT * 32;
    constexpr int G = SMAAT_THREADS / PT;  // channel sub-groups
    constexpr int NW = KC * COT / SMAAT_THREADS;
    constexpr int NY = KC / G;
    extern __shared__ float smem[];
    float* Yl = smem;                   // [KC][PT]
    float* Wl = Yl + KC * PT;           // [KC][COT]
    float* stat = Wl + KC * COT;        // [WPX][3][COT] + [8] wave pixel counts (BN_STAT_FLOATS)
    int* pixoff = (int*)(stat + BN_STAT_FLOATS(WPX, COT));  // [PT]
    int* sidx = pixoff + PT;            // [PT]
    float* biasl = (float*)(sidx + PT); // [COT]
    float* S = biasl + COT;             // [KCI][sstride]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int l31 = lane & 31, half = lane >> 5;
    const TileGeom& g = a.g;

    // XCD-aware block map: the co tiles of one pixel tile run back to back on ONE XCD.
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int cot = idx % a.nco;
    // each XCD owns a CONTIGUOUS range of pixel tiles: neighbouring tiles (shared halo lines, shared
    // weight slabs) meet in one L2 instead of being dealt round-robin over the eight L2s
    const int ptg = (a.dbg & 16) ? (idx / a.nco) * 8 + xcd : xcd * ((g.T + 7) >> 3) + idx / a.nco;
    if (ptg >= g.T) return;
    const int n = ptg / g.tiles_per_img, tl = ptg - n * g.tiles_per_img;
    const int co0 = cot * COT;
    const StageRegion rg = stage_region(g, tl);
    const int sstride = a.sstride;
    if (tid < COT) {  // bias slab -> LDS (unconditional clamped load + select: no branch around a load)
        const int m = co0 + tid;
        const float* bp = a.bias ? a.bias : a.wt;
        const float v = bp[m < a.M ? m : 0];
        biasl[tid] = (a.bias && m < a.M) ? v : 0.f;
    }

    for (int i = tid; i < PT; i += SMAAT_THREADS) {
        int r, c;
        const bool v = tile_pixel(g, tl, i, r, c);
        pixoff[i] = v ? r * g.W + c : -1;
        sidx[i] = v ? (r - rg.row_lo) * rg.SW + (c - rg.col_lo) : (rg.SW + 1);
    }
    int goff[3] = {-2, -2, -2};
    if (DW) {
        const int rsize = rg.nrows * rg.SW;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = tid + SMAAT_THREADS * j;
            if (e < rsize) {
                const int sr = e / rg.SW, sc = e - sr * rg.SW;
                const int gr = rg.row_lo + sr, gc = rg.col_lo + sc;
                goff[j] = (gr >= 0 && gr < g.H && gc >= 0 && gc < g.W) ? gr * g.W + gc : -1;
            }
        }
    }

    f32x16 acc[CT][PXT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    const int nchunks = (a.Kdim + KC - 1) / KC;
    const int pi = tid % PT;
    const int g0 = __builtin_amdgcn_readfirstlane(tid / PT);
    __syncthreads();
    const int po = pixoff[pi];
    const bool aff = DW && (a.in_scale != nullptr);
    const float* xn = a.x + (long)n * a.x_bs;

    float sreg[DW ? KCI : 1][3];
    float yreg[DW ? 1 : NY];
    float wreg[NW];

    // All prefetch loads are UNCONDITIONAL (clamped, always-valid addresses) and the validity is
    // applied with a select afterwards: a branch around a load makes hipcc serialise the loads
    // behind per-element vmcnt(0) waits.
    int gsafe[3];
    bool gm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gm[j] = goff[j] >= 0;
        gsafe[j] = gm[j] ? goff[j] : 0;
    }
    const int po_s = po >= 0 ? po : 0;
    int wk_[NW], wm_[NW];
    bool wmv[NW];
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        const int e = tid + SMAAT_THREADS * r;
        wk_[r] = e / COT;
        const int m = co0 + (e - wk_[r] * COT);
        wmv[r] = m < a.M;
        wm_[r] = wmv[r] ? m : a.M - 1;
    }
    // prefetch = address arithmetic + loads ONLY (no consumer of the loaded values, no branch):
    // the masks / input affine are applied when the registers are committed to LDS.
    auto prefetch = [&](int ch) {
        const int k0 = ch * KC;
        if (DW) {
            const int ci0 = k0 / KPL;
#pragma unroll
            for (int cl = 0; cl < KCI; ++cl) {
                const int ci = ci0 + cl;
                const float* plane = xn + (long)(ci < a.Cin ? ci : a.Cin - 1) * g.P;
#pragma unroll
                for (int j = 0; j < 3; ++j) sreg[cl][j] = plane[gsafe[j]];
            }
        } else {
#pragma unroll
            for (int r = 0; r < NY; ++r) {
                const int kg = k0 + g0 + r * G;
                yreg[r] = xn[(long)(kg < a.Kdim ? kg : a.Kdim - 1) * g.P + po_s];
            }
        }
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int kg = k0 + wk_[r];
            wreg[r] = a.wt[(long)(kg < a.Kdim ? kg : a.Kdim - 1) * a.M + wm_[r]];
        }
    };

    prefetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int k0 = ch * KC;
        if (DW) {
            // S is free: its readers (the depthwise stage of chunk ch-1) finished before that chunk's 2nd barrier
            const int ci0 = k0 / KPL;
#pragma unroll
            for (int cl = 0; cl < KCI; ++cl) {
                const int ci = ci0 + cl;
                const bool cv = ci < a.Cin;
                float sc_ = 1.f, sh_ = 0.f;
                if (aff) {
                    sc_ = a.in_scale[cv ? ci : a.Cin - 1];
                    sh_ = a.in_shift[cv ? ci : a.Cin - 1];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float v = sreg[cl][j];
                    if (aff) v = fmaxf(fmaf(v, sc_, sh_), 0.f);
                    v = (cv && gm[j]) ? v : 0.f;
                    if (goff[j] != -2) S[cl * sstride + tid + SMAAT_THREADS * j] = v;
                }
            }
        }
        __syncthreads();  // B1: S visible; every wave is past the MFMA block of chunk ch-1
        if (DW) {
            const int sb = sidx[pi];
            const int SW = rg.SW;
            const bool wy = (a.y_out != nullptr) && (cot == 0) && (po >= 0);
            const float* bdw = a.b_dw ? a.b_dw : a.w_dw;
            float yv[(KCI / G) * KPL];
#pragma unroll
            for (int it = 0; it < KCI / G; ++it) {
                const int cl = g0 + it * G;
                const float* sp = S + cl * sstride + sb;
                const float s00 = sp[-SW - 1], s01 = sp[-SW], s02 = sp[-SW + 1];
                const float s10 = sp[-1], s11 = sp[0], s12 = sp[1];
                const float s20 = sp[SW - 1], s21 = sp[SW], s22 = sp[SW + 1];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int k = cl * KPL + j, kg = k0 + k;
                    const bool kv = kg < a.Kdim;
                    const int kgs = kv ? kg : a.Kdim - 1;  // clamped: weight loads are unconditional
                    const float* w = a.w_dw + kgs * 9;
                    float y = bdw[kgs];
                    y = a.b_dw ? y : 0.f;
                    y = fmaf(w[0], s00, y);
                    y = fmaf(w[1], s01, y);
                    y = fmaf(w[2], s02, y);
                    y = fmaf(w[3], s10, y);
                    y = fmaf(w[4], s11, y);
                    y = fmaf(w[5], s12, y);
                    y = fmaf(w[6], s20, y);
                    y = fmaf(w[7], s21, y);
                    y = fmaf(w[8], s22, y);
                    y = kv ? y : 0.f;
                    yv[it * KPL + j] = y;
                    Yl[k * PT + pi] = y;
                }
            }
            if (wy) {  // side output, kept out of the compute loop so that the loop stays branch-free
#pragma unroll
                for (int it = 0; it < KCI / G; ++it)
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const int kg = k0 + (g0 + it * G) * KPL + j;
                        if (kg < a.Kdim) a.y_out[((long)n * a.Kdim + kg) * g.P + po] = yv[it * KPL + j];
                    }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NY; ++r) {
                const int kg = k0 + g0 + r * G;
                Yl[(g0 + r * G) * PT + pi] = (kg < a.Kdim && po >= 0) ? yreg[r] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < NW; ++r)
            Wl[tid + SMAAT_THREADS * r] = ((k0 + wk_[r]) < a.Kdim && wmv[r]) ? wreg[r] : 0.f;
        __syncthreads();  // B2
        prefetch(ch + 1 < nchunks ? ch + 1 : ch);  // in flight during the MFMA block (unconditional: no phi copies)
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int krow = 2 * kk + half;
            float av[CT], bv[PXT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[ct] = Wl[krow * COT + (wco * CT + ct) * 32 + l31];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) bv[pt] = Yl[krow * PT + (wpx * PXT + pt) * 32 + l31];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
                    acc[ct][pt] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[pt], av[ct], acc[ct][pt], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct], bv[pt], acc[ct][pt], 0, 0, 0);
        }
    }

    if (TR) {
        // lane -> channel (wco*CT+ct)*32 + l31; register r -> pixel (r&3) + 8*(r>>2) + 4*half of the 32-pixel tile
        float* obase = a.out + (long)n * a.out_bs;
        const int p0 = tl * PT;  // flattened tiles: tile pixel i is plane pixel p0 + i
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = (wco * CT + ct) * 32 + l31;
            const int m = co0 + col;
            if (m < a.M) {
                const float bvv = biasl[col];
                float* rowp = obase + (long)m * g.P + p0;
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i0 = (wpx * PXT + pt) * 32 + 8 * q + 4 * half;
                        if (p0 + i0 < g.P)  // P % 4 == 0: a group of 4 is entirely inside or outside
                            *(float4*)(rowp + i0) = make_float4(fmaxf(acc[ct][pt][4 * q] + bvv, a.out_floor),
                                                                fmaxf(acc[ct][pt][4 * q + 1] + bvv, a.out_floor),
                                                                fmaxf(acc[ct][pt][4 * q + 2] + bvv, a.out_floor),
                                                                fmaxf(acc[ct][pt][4 * q + 3] + bvv, a.out_floor));
                    }
            }
        }
        return;
    }

    // ---- epilogue: bias + coalesced row stores --------------------------------------
    int off[PXT];
#pragma unroll
    for (int pt = 0; pt < PXT; ++pt) off[pt] = pixoff[(wpx * PXT + pt) * 32 + l31];
    float* obase = a.out + (long)n * a.out_bs;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int m = co0 + col;
            if (m < a.M) {
                const float bvv = biasl[col];
                float* rowp = obase + (long)m * g.P;
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
                    if (off[pt] >= 0) rowp[off[pt]] = fmaxf(acc[ct][pt][r] + bvv, a.out_floor);
            }
        }
    }
    // ---- BatchNorm partial statistics of the raw accumulators (z - bias) ---------------
    if (a.part) {
        bool pval[PXT];
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt) pval[pt] = off[pt] >= 0;
        int fl;
        const int nw = bn_wave_count<PXT>(pval, fl);
        bn_wave_partials<CT, PXT>(acc, pval, l31, half, stat + wpx * 3 * COT + wco * CT * 32, COT, fl);
        if (lane == 0) ((int*)(stat + WPX * 3 * COT))[wpx] = nw;
        __syncthreads();
        for (int col = tid; col < COT; col += SMAAT_THREADS) {
            float mean, m2, cnt;
            bn_tile_combine<WPX>(stat, (const int*)(stat + WPX * 3 * COT), COT, col, mean, m2, cnt);
            const int m = co0 + col;
            if (m < a.M) {
                a.part[((long)0 * g.T + ptg) * a.M + m] = mean;
                a.part[((long)1 * g.T + ptg) * a.M + m] = m2;
                a.part[((long)2 * g.T + ptg) * a.M + m] = cnt;
            }
        }
    }
}

// =====================================================================================
// Wave-specialised variant of k_pwgemm: 4 CONSUMER waves (one per SIMD) do nothing but
// ds_read + v_mfma_f32_32x32x2_f32 on chunk i, while NPT/64 PRODUCER waves stage chunk i+1
// (halo tile -> LDS -> depthwise 3x3 in VALU -> B-operand rows, weight slab) and keep the
// global loads of chunks i+2 / i+3 in flight.  All LDS buffers are double buffered, so there
// is exactly ONE workgroup barrier per 16-row chunk and the matrix pipe and the VALU/LDS/VMEM
// pipes of a SIMD work on different chunks at the same time (CDNA4 issues MFMA and VALU of
// different waves concurrently).
//   iteration i:   consumers  M(i)      : Yl[i&1], Wl[i&1] -> MFMA
//                  producers  D(i+1)    : S[(i+1)&1] -> depthwise -> Yl[(i+1)&1]
//                             Wc(i+1)   : weight registers -> Wl[(i+1)&1];  issue loads W(i+2)
//                             Sc(i+2)   : halo registers   -> S[i&1];       issue loads S(i+3)
//                  barrier
// =====================================================================================
template <int WCO, int CT, int WPX, int PXT, int MODE, int NPT, bool AFF = false>
__global__ __launch_bounds__(256 + NPT) void k_pwgemm_ws(const PwArgs a) {
    constexpr bool DW = MODE > 0;
    constexpr int KPL = DW ? MODE : 1;
    constexpr int KCI = KC / KPL;  // input channels per chunk (DW)
    constexpr int COT = WCO * CT * 32;
    constexpr int PT = WPX * PXT * 32;
    constexpr int NTH = 256 + NPT;
    constexpr int G = NPT / PT;         // producer channel sub-groups
    constexpr int NW = KC * COT / NPT;  // weight-slab elements per producer thread
    constexpr int NY = KC / G;          // MODE 0: B rows per producer thread
    constexpr int SST = SMAX_WS;                // staged floats per input channel (fixed LDS stride)
    constexpr int NS = SST / NPT;               // staged halo elements per producer thread and channel
    static_assert(WCO * WPX == 4, "4 consumer waves");
    static_assert(NPT % PT == 0 && KC % G == 0 && (KC * COT) % NPT == 0 && SST % NPT == 0, "producer mapping");
    static_assert(!DW || (KCI % G == 0), "depthwise mapping");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Yl = smem;                   // [2][KC][PT]
    float* Wl = Yl + 2 * KC * PT;       // [2][KC][COT]
    float* DWl = Wl + 2 * KC * COT;     // [2][256]: per chunk [KC][12] = 9 depthwise taps, bias, 2 pad
    float* stat = DWl + 2 * 256;        // [WPX][3][COT] + [8] wave pixel counts (BN_STAT_FLOATS)
    int* pixoff = (int*)(stat + BN_STAT_FLOATS(WPX, COT));  // [PT]
    int* sidx = pixoff + PT;            // [PT]
    float* biasl = (float*)(sidx + PT); // [COT]
    float* S = biasl + COT;             // [2][KCI][sstride]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int wave = wv & 3;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int l31 = lane & 31, half = lane >> 5;
    const int ptid = producer ? tid - 256 : 0;  // producer thread index
    const TileGeom& g = a.g;

    // XCD-aware block map: the co tiles of one pixel tile run back to back on ONE XCD.
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int cot = idx % a.nco;
    // each XCD owns a CONTIGUOUS range of pixel tiles: neighbouring tiles (shared halo lines, shared
    // weight slabs) meet in one L2 instead of being dealt round-robin over the eight L2s
    const int ptg = (a.dbg & 16) ? (idx / a.nco) * 8 + xcd : xcd * ((g.T + 7) >> 3) + idx / a.nco;
    if (ptg >= g.T) return;
    const int n = ptg / g.tiles_per_img, tl = ptg - n * g.tiles_per_img;
    const int co0 = cot * COT;
    const StageRegion rg = stage_region(g, tl);
    constexpr int sstride = SST;
    if (tid < COT) {
        const int m = co0 + tid;
        const float* bp = a.bias ? a.bias : a.wt;
        const float v = bp[m < a.M ? m : 0];
        biasl[tid] = (a.bias && m < a.M) ? v : 0.f;
    }
    for (int i = tid; i < PT; i += NTH) {
        int r, c;
        const bool v = tile_pixel(g, tl, i, r, c);
        pixoff[i] = v ? r * g.W + c : -1;
        sidx[i] = v ? (r - rg.row_lo) * rg.SW + (c - rg.col_lo) : (rg.SW + 1);
    }
    int goff[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) goff[j] = -2;
    if (DW) {
        const int rsize = rg.nrows * rg.SW;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int e = ptid + NPT * j;
            if (e < rsize) {
                const int sr = e / rg.SW, sc = e - sr * rg.SW;
                const int gr = rg.row_lo + sr, gc = rg.col_lo + sc;
                goff[j] = (gr >= 0 && gr < g.H && gc >= 0 && gc < g.W) ? gr * g.W + gc : -1;
            }
        }
    }

    const int nchunks = (a.Kdim + KC - 1) / KC;
    const int pi = ptid % PT;
    const int g0 = __builtin_amdgcn_readfirstlane(ptid / PT);
    __syncthreads();  // pixoff / sidx / biasl visible
    const int po = pixoff[pi];
    const float* xn = a.x + (long)n * a.x_bs;

    float sreg[DW ? KCI : 1][NS];
    float ascr[(DW && AFF) ? KCI : 1], ashr[(DW && AFF) ? KCI : 1];  // input affine of the chunk's channels
    float yreg[DW ? 1 : NY];
    float wreg[NW];
    int gsafe[NS];
    bool gm[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        gm[j] = goff[j] >= 0;
        gsafe[j] = gm[j] ? goff[j] : 0;
    }
    const int po_s = po >= 0 ? po : 0;
    int wk_[NW], wm_[NW];
    bool wmv[NW];
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        const int e = ptid + NPT * r;
        wk_[r] = e / COT;
        const int m = co0 + (e - wk_[r] * COT);
        wmv[r] = m < a.M;
        wm_[r] = wmv[r] ? m : a.M - 1;
    }
    // depthwise taps + bias of the chunk travel with the halo tile (same pipeline stage as S:
    // committed one barrier before the depthwise stage that reads them): slot (k, t) of [KC][12],
    // t < 9 tap, t == 9 bias.  (Read from global inside the depthwise stage they would be
    // per-channel dependent L2 round trips: the pointers are not provably read-only, so hipcc
    // cannot use scalar loads for them.)
    const int dwi = ptid & 255;
    const int dwk = dwi / 12, dwt = dwi - dwk * 12;
    const float* dwsrc = (dwt < 9) ? a.w_dw : (a.b_dw ? a.b_dw : a.w_dw);
    const int dwmul = (dwt < 9) ? 9 : 1, dwadd = (dwt < 9) ? dwt : 0;
    const bool dwvalid = DW && (dwk < KC) && (dwt < 9 || (dwt == 9 && a.b_dw != nullptr));
    float dwreg = 0.f;
    // ---- producer stages (loads are unconditional on clamped addresses; masks at commit) ----
    auto clampc = [&](int ch) { return ch < nchunks ? ch : nchunks - 1; };
    auto prefetch_b = [&](int ch_) {  // halo tile (DW) or B rows (plain) of chunk ch -> registers
        const int k0 = clampc(ch_) * KC;
        if (DW) {
            const int ci0 = k0 / KPL;
#pragma unroll
            for (int cl = 0; cl < KCI; ++cl) {
                const int ci = ci0 + cl;
                const float* plane = xn + (long)(ci < a.Cin ? ci : a.Cin - 1) * g.P;
#pragma unroll
                for (int j = 0; j < NS; ++j) sreg[cl][j] = plane[gsafe[j]];
                if (AFF) {
                    ascr[cl] = a.in_scale[ci < a.Cin ? ci : a.Cin - 1];
                    ashr[cl] = a.in_shift[ci < a.Cin ? ci : a.Cin - 1];
                }
            }
            if (!TAPS_SMEM) {
                const int kg = k0 + (dwk < KC ? dwk : 0);
                dwreg = dwsrc[(kg < a.Kdim ? kg : a.Kdim - 1) * dwmul + dwadd];
            }
        } else {
#pragma unroll
            for (int r = 0; r < NY; ++r) {
                const int kg = k0 + g0 + r * G;
                yreg[r] = xn[(long)(kg < a.Kdim ? kg : a.Kdim - 1) * g.P + po_s];
            }
        }
    };
    auto prefetch_w = [&](int ch_) {
        const int k0 = clampc(ch_) * KC;
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int kg = k0 + wk_[r];
            wreg[r] = a.wt[(long)(kg < a.Kdim ? kg : a.Kdim - 1) * a.M + wm_[r]];
        }
    };
    auto commit_w = [&](int ch_, int buf) {
        const int k0 = clampc(ch_) * KC;
        float* Wb = Wl + buf * (KC * COT);
#pragma unroll
        for (int r = 0; r < NW; ++r) Wb[ptid + NPT * r] = ((k0 + wk_[r]) < a.Kdim && wmv[r]) ? wreg[r] : 0.f;
    };
    auto commit_b = [&](int ch_, int buf) {  // registers -> S[buf] (DW) or Yl[buf] (plain)
        const int k0 = clampc(ch_) * KC;
        if (DW) {
            float* Sb = S + buf * (KCI * sstride);
            const int ci0 = k0 / KPL;
#pragma unroll
            for (int cl = 0; cl < KCI; ++cl) {
                const int ci = ci0 + cl;
                const bool cv = ci < a.Cin;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    float v = sreg[cl][j];
                    if (AFF) v = fmaxf(fmaf(v, ascr[cl], ashr[cl]), 0.f);  // relu(bn(z)) of the previous half on load
                    v = (cv && gm[j]) ? v : 0.f;                           // zero padding stays zero
                    Sb[cl * sstride + ptid + NPT * j] = v;  // every slot of the stripe is written: no divergent branch
                }
            }
            if (!TAPS_SMEM) DWl[buf * 256 + dwi] = (dwvalid && (k0 + dwk) < a.Kdim) ? dwreg : 0.f;
        } else {
            float* Yb = Yl + buf * (KC * PT);
#pragma unroll
            for (int r = 0; r < NY; ++r) {
                const int kg = k0 + g0 + r * G;
                Yb[(g0 + r * G) * PT + pi] = (kg < a.Kdim && po >= 0) ? yreg[r] : 0.f;
            }
        }
    };
    auto dwstage = [&](int ch_, int buf) {  // S[buf] -> depthwise 3x3 -> Yl[buf] (+ side output)
        const int k0 = ch_ * KC;
        const float* Sb = S + buf * (KCI * sstride);
        float* Yb = Yl + buf * (KC * PT);
        const int sb = sidx[pi];
        const int SW = rg.SW;
        const bool wy = (a.y_out != nullptr) && (cot == 0) && (po >= 0);
        const float4* DWb = (const float4*)(DWl + buf * 256);
        const cfloat* wdw_c = (const cfloat*)a.w_dw;
        const cfloat* bdw_c = (const cfloat*)(a.b_dw ? a.b_dw : a.w_dw);
        float* yo = a.y_out + ((long)n * a.Kdim) * g.P + po_s;
        // partially unrolled on purpose: the accumulators of the consumer role stay allocated
        // in this role too, so the depthwise stage must stay small in registers
#pragma unroll DWUNROLL
        for (int it = 0; it < KCI / G; ++it) {
            const int cl = g0 + it * G;
            const float* sp = Sb + cl * sstride + sb;
            const float s00 = sp[-SW - 1], s01 = sp[-SW], s02 = sp[-SW + 1];
            const float s10 = sp[-1], s11 = sp[0], s12 = sp[1];
            const float s20 = sp[SW - 1], s21 = sp[SW], s22 = sp[SW + 1];
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int k = cl * KPL + j, kg = k0 + k;
                const bool kv = kg < a.Kdim;
                float y;
                if (TAPS_SMEM) {
                    // wave-uniform taps through the scalar cache (constant address space -> s_load)
                    const int kgs = kv ? kg : a.Kdim - 1;
                    const cfloat* w = wdw_c + kgs * 9;
                    y = a.b_dw ? bdw_c[kgs] : 0.f;
                    y = fmaf(w[0], s00, y);
                    y = fmaf(w[1], s01, y);
                    y = fmaf(w[2], s02, y);
                    y = fmaf(w[3], s10, y);
                    y = fmaf(w[4], s11, y);
                    y = fmaf(w[5], s12, y);
                    y = fmaf(w[6], s20, y);
                    y = fmaf(w[7], s21, y);
                    y = fmaf(w[8], s22, y);
                    y = kv ? y : 0.f;
                } else {
                    const float4 wa = DWb[k * 3], wb = DWb[k * 3 + 1], wc = DWb[k * 3 + 2];  // zeros when kg >= Kdim
                    y = wc.y;
                    y = fmaf(wa.x, s00, y);
                    y = fmaf(wa.y, s01, y);
                    y = fmaf(wa.z, s02, y);
                    y = fmaf(wa.w, s10, y);
                    y = fmaf(wb.x, s11, y);
                    y = fmaf(wb.y, s12, y);
                    y = fmaf(wb.z, s20, y);
                    y = fmaf(wb.w, s21, y);
                    y = fmaf(wc.x, s22, y);
                }
                Yb[k * PT + pi] = y;
                if (wy && kv) yo[(long)kg * g.P] = y;
            }
        }
    };
    // The two roles run SEPARATE loops (wave-uniform scalar branch on `producer`): the accumulators
    // exist only on the consumer side, so the register allocation is max(role), not the sum, and the
    // MFMA chain carries no phi copies.  Every wave executes the same number of barriers.
    if (producer) {
        // ---- prologue ----
        prefetch_b(0);
        prefetch_w(0);
        commit_b(0, 0);
        commit_w(0, 0);
        prefetch_b(1);
        prefetch_w(1);
        __syncthreads();
        if (DW) {
            dwstage(0, 0);
            commit_b(1, 1);
            prefetch_b(2);
        }
        __syncthreads();
        // ---- main loop: one barrier per chunk ----
        for (int i = 0; i < nchunks; ++i) {
            if (i + 1 < nchunks && (a.dbg & 3) != 2) {
                const int nb = (i + 1) & 1;
                if (DW) {
                    // commits first (their loads were issued one iteration ago), then the new loads,
                    // then the LDS-only depthwise stage: nothing waits on a freshly issued load
                    commit_w(i + 1, nb);
                    commit_b(i + 2, i & 1);
                    prefetch_w(i + 2);
                    prefetch_b(i + 3);
                    dwstage(i + 1, nb);
                } else {
                    commit_b(i + 1, nb);
                    commit_w(i + 1, nb);
                    prefetch_b(i + 2);
                    prefetch_w(i + 2);
                }
            }
            __syncthreads();
        }
    } else {
        f32x16 acc[CT][PXT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
        __syncthreads();
        __syncthreads();
        if (a.dbg & 8) __builtin_amdgcn_s_setprio(3);  // experiment: matrix waves win issue arbitration
        for (int i = 0; i < nchunks; ++i) {
            const float* Wb = Wl + (i & 1) * (KC * COT);
            const float* Yb = Yl + (i & 1) * (KC * PT);
            // fragments of step kk+1 are read while the MFMAs of step kk run (two register sets)
            float av[2][CT], bv[2][PXT];
            const float* wp = Wb + half * COT + wco * CT * 32 + l31;
            const float* yp = Yb + half * PT + wpx * PXT * 32 + l31;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[0][ct] = wp[ct * 32];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) bv[0][pt] = yp[pt * 32];
            if ((a.dbg & 3) != 1)
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                if (kk + 1 < KC / 2) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) av[(kk + 1) & 1][ct] = wp[(2 * kk + 2) * COT + ct * 32];
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt) bv[(kk + 1) & 1][pt] = yp[(2 * kk + 2) * PT + pt * 32];
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        acc[ct][pt] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][ct], bv[kk & 1][pt], acc[ct][pt], 0, 0, 0);
            }
            __syncthreads();
        }

        // ---- epilogue (consumer waves): bias + coalesced row stores, BatchNorm partials ----
        int off[PXT];
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt) off[pt] = pixoff[(wpx * PXT + pt) * 32 + l31];
        float* obase = a.out + (long)n * a.out_bs;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int m = co0 + col;
                if (m < a.M) {
                    const float bvv = biasl[col];
                    float* rowp = obase + (long)m * g.P;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        if (off[pt] >= 0) rowp[off[pt]] = fmaxf(acc[ct][pt][r] + bvv, a.out_floor);
                }
            }
        }
        if (a.part) {
            bool pval[PXT];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) pval[pt] = off[pt] >= 0;
            int fl;
            const int nw = bn_wave_count<PXT>(pval, fl);
            bn_wave_partials<CT, PXT>(acc, pval, l31, half, stat + wpx * 3 * COT + wco * CT * 32, COT, fl);
            if (lane == 0) ((int*)(stat + WPX * 3 * COT))[wpx] = nw;
        }
    }
    if (a.part) {
        __syncthreads();
        for (int col = tid; col < COT; col += NTH) {
            float mean, m2, cnt;
            bn_tile_combine<WPX>(stat, (const int*)(stat + WPX * 3 * COT), COT, col, mean, m2, cnt);
            const int m = co0 + col;
            if (m < a.M) {
                a.part[((long)0 * g.T + ptg) * a.M + m] = mean;
                a.part[((long)1 * g.T + ptg) * a.M + m] = m2;
                a.part[((long)2 * g.T + ptg) * a.M + m] = cnt;
            }
        }
    }
}

// =====================================================================================
// k_dsconv_strip: wave-specialised fused depthwise->pointwise forward for 2-D pixel tiles
// (TH x TW pixels, TH % 4 == 0, W % 4 == 0, 16-B aligned planes).  Same pipeline and consumer
// as k_pwgemm_ws; the PRODUCER is rebuilt around instruction count, its real limit:
//   * the halo tile is staged as aligned float4 columns [c0-4, c0+TW+4) x rows [r0-1, r0+TH]:
//     one global_load_dwordx4 + one ds_write_b128 per 4 staged floats;
//   * the depthwise stage works on STRIPS of 4 vertically adjacent pixels of one input channel:
//     18 LDS reads (6 rows x 3 columns) feed 4 pixels x kpl outputs (2.25 reads per output
//     instead of 9), and the 9 taps + bias of an output channel are read once per strip.
// =====================================================================================
template <int WCO, int CT, int WPX, int PXT, int KPL, int NPT>
__global__ __launch_bounds__(256 + NPT) void k_dsconv_strip(const PwArgs a) {
    constexpr int KCI = KC / KPL;
    constexpr int COT = WCO * CT * 32;
    constexpr int PT = WPX * PXT * 32;
    constexpr int NTH = 256 + NPT;
    constexpr int NW = KC * COT / NPT;
    constexpr int SST = SMAX_WS;                      // floats per staged channel
    constexpr int NSL = (KCI * 108 + NPT - 1) / NPT;  // float4 staging slots per producer thread
    constexpr int SPC = PT / 4;                       // strips per input channel
    constexpr int TASKS = KCI * SPC;                  // strip tasks per chunk
    constexpr int NTK = (TASKS + NPT - 1) / NPT;      // strip tasks per producer thread
    static_assert(WCO * WPX == 4, "4 consumer waves");
    static_assert((KC * COT) % NPT == 0 && TASKS % 64 == 0, "producer mapping");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Yl = smem;                   // [2][KC][PT]
    float* Wl = Yl + 2 * KC * PT;       // [2][KC][COT]
    float* DWl = Wl + 2 * KC * COT;     // [2][256]: per chunk [KC][12] = 9 taps, bias, 2 pad
    float* stat = DWl + 2 * 256;        // [WPX][3][COT] + [8] wave pixel counts (BN_STAT_FLOATS)
    int* pixoff = (int*)(stat + BN_STAT_FLOATS(WPX, COT));  // [PT]
    float* biasl = (float*)(pixoff + PT);        // [COT]
    float* S = biasl + COT;             // [2][KCI][SST]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int wave = wv & 3;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int l31 = lane & 31, half = lane >> 5;
    const int ptid = producer ? tid - 256 : 0;
    const TileGeom& g = a.g;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int cot = idx % a.nco;
    // each XCD owns a CONTIGUOUS range of pixel tiles: neighbouring tiles (shared halo lines, shared
    // weight slabs) meet in one L2 instead of being dealt round-robin over the eight L2s
    const int ptg = (a.dbg & 16) ? (idx / a.nco) * 8 + xcd : xcd * ((g.T + 7) >> 3) + idx / a.nco;
    if (ptg >= g.T) return;
    const int n = ptg / g.tiles_per_img, tl = ptg - n * g.tiles_per_img;
    const int co0 = cot * COT;
    const int TW = g.TW, TH = g.TH;
    const int twl = __builtin_ctz(TW);
    const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
    const int r0 = ty * TH, c0 = tx * TW;
    const int stride = a.sstride;       // staged row stride (floats, multiple of 4)
    const int nrow = TH + 2, ncol4 = (TW + 8) >> 2;
    if (tid < COT) {
        const int m = co0 + tid;
        const float* bp = a.bias ? a.bias : a.wt;
        const float v = bp[m < a.M ? m : 0];
        biasl[tid] = (a.bias && m < a.M) ? v : 0.f;
    }
    for (int i = tid; i < PT; i += NTH) {
        const int r = r0 + (i >> twl), c = c0 + (i & (TW - 1));
        pixoff[i] = (r < g.H && c < g.W) ? r * g.W + c : -1;
    }
    const int nchunks = (a.Kdim + KC - 1) / KC;
    const float* xn = a.x + (long)n * a.x_bs;

    // ---- producer-side invariants ----
    int s_cl[NSL], s_in[NSL], s_lo[NSL];
    bool s_ok[NSL];
    {
        const int per = nrow * ncol4, F = KCI * per;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int f = (ptid + NPT * j) % F;  // surplus slots re-stage a valid element: no branches
            const int cl = f / per, rem = f - cl * per;
            const int rr = rem / ncol4, q = rem - rr * ncol4;
            const int gr = r0 - 1 + rr, gc = c0 - 4 + 4 * q;
            const bool ok = gr >= 0 && gr < g.H && gc >= 0 && gc < g.W;
            s_cl[j] = cl;
            s_in[j] = ok ? gr * g.W + gc : 0;
            s_lo[j] = cl * SST + rr * stride + 4 * q;
            s_ok[j] = ok;
        }
    }
    int t_cl[NTK], t_sb[NTK], t_px[NTK], t_go[NTK], t_gr[NTK];
    bool t_cok[NTK];
#pragma unroll
    for (int u = 0; u < NTK; ++u) {
        const int t = ptid + NPT * u;
        const int cl = (t / SPC) % KCI, sidx_ = t % SPC;  // (% KCI keeps surplus tasks in range; they are skipped)
        const int rgp = sidx_ >> twl, c = sidx_ & (TW - 1);
        t_cl[u] = cl;
        t_sb[u] = cl * SST + (rgp * 4) * stride + c + 3;
        t_px[u] = (rgp * 4) * TW + c;
        t_gr[u] = r0 + rgp * 4;
        t_go[u] = (r0 + rgp * 4) * g.W + c0 + c;
        t_cok[u] = (c0 + c) < g.W;
    }
    int wk_[NW], wm_[NW];
    bool wmv[NW];
#pragma unroll
    for (int r = 0; r < NW; ++r) {
        const int e = ptid + NPT * r;
        wk_[r] = e / COT;
        const int m = co0 + (e - wk_[r] * COT);
        wmv[r] = m < a.M;
        wm_[r] = wmv[r] ? m : a.M - 1;
    }
    const int dwi = ptid & 255;
    const int dwk = dwi / 12, dwt = dwi - dwk * 12;
    const float* dwsrc = (dwt < 9) ? a.w_dw : (a.b_dw ? a.b_dw : a.w_dw);
    const int dwmul = (dwt < 9) ? 9 : 1, dwadd = (dwt < 9) ? dwt : 0;
    const bool dwvalid = (dwk < KC) && (dwt < 9 || (dwt == 9 && a.b_dw != nullptr));
    float dwreg = 0.f;
    float4 sreg[NSL];
    float wreg[NW];

    auto clampc = [&](int ch) { return ch < nchunks ? ch : nchunks - 1; };
    auto prefetch_b = [&](int ch_) {
        const int ci0 = clampc(ch_) * KCI;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int ci = ci0 + s_cl[j];
            sreg[j] = *(const float4*)(xn + (long)(ci < a.Cin ? ci : a.Cin - 1) * g.P + s_in[j]);
        }
        const int kg = clampc(ch_) * KC + (dwk < KC ? dwk : 0);
        dwreg = dwsrc[(kg < a.Kdim ? kg : a.Kdim - 1) * dwmul + dwadd];
    };
    auto commit_b = [&](int ch_, int buf) {
        const int ci0 = clampc(ch_) * KCI;
        float* Sb = S + buf * (KCI * SST);
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const bool ok = s_ok[j] && (ci0 + s_cl[j]) < a.Cin;
            float4 v = sreg[j];
            v.x = ok ? v.x : 0.f;
            v.y = ok ? v.y : 0.f;
            v.z = ok ? v.z : 0.f;
            v.w = ok ? v.w : 0.f;
            *(float4*)(Sb + s_lo[j]) = v;
        }
        DWl[buf * 256 + dwi] = (dwvalid && (clampc(ch_) * KC + dwk) < a.Kdim) ? dwreg : 0.f;
    };
    auto prefetch_w = [&](int ch_) {
        const int k0 = clampc(ch_) * KC;
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int kg = k0 + wk_[r];
            wreg[r] = a.wt[(long)(kg < a.Kdim ? kg : a.Kdim - 1) * a.M + wm_[r]];
        }
    };
    auto commit_w = [&](int ch_, int buf) {
        const int k0 = clampc(ch_) * KC;
        float* Wb = Wl + buf * (KC * COT);
#pragma unroll
        for (int r = 0; r < NW; ++r) Wb[ptid + NPT * r] = ((k0 + wk_[r]) < a.Kdim && wmv[r]) ? wreg[r] : 0.f;
    };
    auto dwstage = [&](int ch_, int buf) {
        const int k0 = ch_ * KC;
        const float* Sb = S + buf * (KCI * SST);
        float* Yb = Yl + buf * (KC * PT);
        const float4* DWb = (const float4*)(DWl + buf * 256);
        const bool wy = (a.y_out != nullptr) && (cot == 0);
        float* yo = a.y_out + ((long)n * a.Kdim) * g.P;
#pragma unroll
        for (int u = 0; u < NTK; ++u) {
            if (NTK * NPT == TASKS || (ptid + NPT * u) < TASKS) {  // wave-uniform (TASKS % 64 == 0)
                const float* sp = Sb + t_sb[u];
                float v[6][3];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) v[rr][dc] = sp[rr * stride + dc];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int k = t_cl[u] * KPL + j, kg = k0 + k;
                    const float4 wa = DWb[k * 3], wb = DWb[k * 3 + 1], wc = DWb[k * 3 + 2];  // zeros when kg >= Kdim
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float y = wc.y;
                        y = fmaf(wa.x, v[i][0], y);
                        y = fmaf(wa.y, v[i][1], y);
                        y = fmaf(wa.z, v[i][2], y);
                        y = fmaf(wa.w, v[i + 1][0], y);
                        y = fmaf(wb.x, v[i + 1][1], y);
                        y = fmaf(wb.y, v[i + 1][2], y);
                        y = fmaf(wb.z, v[i + 2][0], y);
                        y = fmaf(wb.w, v[i + 2][1], y);
                        y = fmaf(wc.x, v[i + 2][2], y);
                        Yb[k * PT + t_px[u] + i * TW] = y;
                        if (wy && kg < a.Kdim && t_cok[u] && (t_gr[u] + i) < g.H)
                            yo[(long)kg * g.P + t_go[u] + i * g.W] = y;
                    }
                }
            }
        }
    };

    if (producer) {
        prefetch_b(0);
        prefetch_w(0);
        commit_b(0, 0);
        commit_w(0, 0);
        prefetch_b(1);
        prefetch_w(1);
        __syncthreads();
        dwstage(0, 0);
        commit_b(1, 1);
        prefetch_b(2);
        __syncthreads();
        for (int i = 0; i < nchunks; ++i) {
            if (i + 1 < nchunks && (a.dbg & 3) != 2) {
                const int nb = (i + 1) & 1;
                commit_w(i + 1, nb);
                commit_b(i + 2, i & 1);
                prefetch_w(i + 2);
                prefetch_b(i + 3);
                dwstage(i + 1, nb);
            }
            __syncthreads();
        }
    } else {
        f32x16 acc[CT][PXT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
        __syncthreads();
        __syncthreads();
        for (int i = 0; i < nchunks; ++i) {
            const float* Wb = Wl + (i & 1) * (KC * COT);
            const float* Yb = Yl + (i & 1) * (KC * PT);
            float av[2][CT], bv[2][PXT];
            const float* wp = Wb + half * COT + wco * CT * 32 + l31;
            const float* yp = Yb + half * PT + wpx * PXT * 32 + l31;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[0][ct] = wp[ct * 32];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) bv[0][pt] = yp[pt * 32];
            if ((a.dbg & 3) != 1)
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                if (kk + 1 < KC / 2) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) av[(kk + 1) & 1][ct] = wp[(2 * kk + 2) * COT + ct * 32];
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt) bv[(kk + 1) & 1][pt] = yp[(2 * kk + 2) * PT + pt * 32];
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        acc[ct][pt] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][ct], bv[kk & 1][pt], acc[ct][pt], 0, 0, 0);
            }
            __syncthreads();
        }
        int off[PXT];
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt) off[pt] = pixoff[(wpx * PXT + pt) * 32 + l31];
        float* obase = a.out + (long)n * a.out_bs;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int m = co0 + col;
                if (m < a.M) {
                    const float bvv = biasl[col];
                    float* rowp = obase + (long)m * g.P;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        if (off[pt] >= 0) rowp[off[pt]] = fmaxf(acc[ct][pt][r] + bvv, a.out_floor);
                }
            }
        }
        if (a.part) {
            bool pval[PXT];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) pval[pt] = off[pt] >= 0;
            int fl;
            const int nw = bn_wave_count<PXT>(pval, fl);
            bn_wave_partials<CT, PXT>(acc, pval, l31, half, stat + wpx * 3 * COT + wco * CT * 32, COT, fl);
            if (lane == 0) ((int*)(stat + WPX * 3 * COT))[wpx] = nw;
        }
    }
    if (a.part) {
        __syncthreads();
        for (int col = tid; col < COT; col += NTH) {
            float mean, m2, cnt;
            bn_tile_combine<WPX>(stat, (const int*)(stat + WPX * 3 * COT), COT, col, mean, m2, cnt);
            const int m = co0 + col;
            if (m < a.M) {
                a.part[((long)0 * g.T + ptg) * a.M + m] = mean;
                a.part[((long)1 * g.T + ptg) * a.M + m] = m2;
                a.part[((long)2 * g.T + ptg) * a.M + m] = cnt;
            }
        }
    }
}

// =====================================================================================
// streamed weight gradient  dW[m][k] = sum_{n,p} dz[n][m][p] * y[n][k][p]
//   output-stationary 64*CT x 128 tile per workgroup, contraction over 64-pixel chunks,
//   both operands staged row-major with an odd LDS stride (conflict-free column reads),
//   next chunk prefetched into registers during the MFMA block.
// =====================================================================================


template <int CT, bool VEC>
__global__ __launch_bounds__(SMAAT_THREADS, 2) void k_wgrad2(const Wg2Args a) {
    constexpr int MT = 64 * CT, KT = 128, PS = 64, LS = PS + 1;
    constexpr int NPF = (MT + KT) / 16;
    extern __shared__ float smem[];
    float* Zs = smem;            // [MT][LS]
    float* Ys = Zs + MT * LS;    // [KT][LS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wk = wave >> 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int r16 = tid >> 4, px4 = (tid & 15) * 4;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int ntile = a.nmt * a.nkt;
    const int rest = idx % ntile;
    const int split = (idx / ntile) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int mt = rest % a.nmt, kt = rest / a.nmt;
    const int m0 = mt * MT, k0 = kt * KT;
    f32x16 acc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int kw = 0; kw < 2; ++kw)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][kw][r] = 0.f;

    float4 pf[NPF];
    // row pointers with clamped (always valid) row indices; validity applied by select
    const float* rowp[NPF];
    bool rowv[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int row = r16 + 16 * j;
        if (row < MT) {
            rowv[j] = (m0 + row) < a.M;
            rowp[j] = a.dz + (long)(rowv[j] ? m0 + row : a.M - 1) * a.P;
        } else {
            rowv[j] = (k0 + row - MT) < a.K;
            rowp[j] = a.y + (long)(rowv[j] ? k0 + row - MT : a.K - 1) * a.P;
        }
    }
    // ONE load path per kernel instantiation (a runtime choice between two paths would make
    // the compiler copy the loaded registers at the merge point, i.e. wait for them at once).
    auto prefetch = [&](int c) {
        const int n = c / a.nchunk_img;
        const int p0 = (c - n * a.nchunk_img) * PS + px4;
        if (VEC) {  // P % 4 == 0: a lane is entirely inside or entirely outside the plane
            const int p0c = p0 + 3 < a.P ? p0 : a.P - 4;
#pragma unroll
            for (int j = 0; j < NPF; ++j)
                pf[j] = *(const float4*)(rowp[j] + (long)n * ((r16 + 16 * j) < MT ? a.dz_bs : a.y_bs) + p0c);
        } else {
            const int q0 = p0 + 0 < a.P ? p0 + 0 : a.P - 1, q1 = p0 + 1 < a.P ? p0 + 1 : a.P - 1;
            const int q2 = p0 + 2 < a.P ? p0 + 2 : a.P - 1, q3 = p0 + 3 < a.P ? p0 + 3 : a.P - 1;
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const float* src = rowp[j] + (long)n * ((r16 + 16 * j) < MT ? a.dz_bs : a.y_bs);
                pf[j] = make_float4(src[q0], src[q1], src[q2], src[q3]);
            }
        }
    };

    // chunk i of this split is chunk (split + i*nsplit): the workgroups that run at the same time
    // stream adjacent 256 B pieces of every row (DRAM page / L2 locality)
    const int c_begin = split, c_end = a.total_chunks, c_step = a.nsplit;
    if (c_begin < c_end) prefetch(c_begin);
    for (int c = c_begin; c < c_end; c += c_step) {
        __syncthreads();  // previous MFMA block done with Zs / Ys
        {
            const int pc = (c % a.nchunk_img) * PS + px4;
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int row = r16 + 16 * j;
                float* dst = (row < MT ? Zs + row * LS : Ys + (row - MT) * LS) + px4;
                dst[0] = (rowv[j] && pc + 0 < a.P) ? pf[j].x : 0.f;
                dst[1] = (rowv[j] && pc + 1 < a.P) ? pf[j].y : 0.f;
                dst[2] = (rowv[j] && pc + 2 < a.P) ? pf[j].z : 0.f;
                dst[3] = (rowv[j] && pc + 3 < a.P) ? pf[j].w : 0.f;
            }
        }
        __syncthreads();
        prefetch(c + c_step < c_end ? c + c_step : c);
#pragma unroll 8
        for (int s = 0; s < PS / 2; ++s) {
            const int px = 2 * s + half;
            float av[CT], bv[2];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[ct] = Zs[((wm * CT + ct) * 32 + l31) * LS + px];
#pragma unroll
            for (int kw = 0; kw < 2; ++kw) bv[kw] = Ys[((wk * 2 + kw) * 32 + l31) * LS + px];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int kw = 0; kw < 2; ++kw)
                    acc[ct][kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct], bv[kw], acc[ct][kw], 0, 0, 0);
        }
    }
    float* ob = a.part + (long)split * a.M * a.K;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < a.M) {
#pragma unroll
                for (int kw = 0; kw < 2; ++kw) {
                    const int kg = k0 + (wk * 2 + kw) * 32 + l31;
                    if (kg < a.K) ob[(long)m * a.K + kg] = acc[ct][kw][r];
                }
            }
        }
}

// =====================================================================================
// weight gradient with the depthwise stage recomputed in the kernel (memory-lean variant)
// =====================================================================================


template <int WCO, int CT, int WK, int KW, bool DW, bool AFF>
__global__ __launch_bounds__(SMAAT_THREADS) void k_wgrad(const WgArgs a) {
    constexpr int COT = WCO * CT * 32;
    constexpr int KT = WK * KW * 32;
    constexpr int ZS = COT + 1;  // odd strides: conflict-free column reads
    constexpr int YS = KT + 1;
    extern __shared__ float smem[];
    float* Zt = smem;             // [PSW][ZS]   dz tile, pixel-major
    float* Yt = Zt + PSW * ZS;    // [PSW][YS]   dw output tile, pixel-major
    float* S = Yt + PSW * YS;     // [KT/kpl][SMAXW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave % WCO, wk = wave / WCO;
    const int l31 = lane & 31, half = lane >> 5;
    const TileGeom& g = a.g;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int ntile = a.nco * a.nkt;
    const int rest = idx % ntile;
    const int split = (idx / ntile) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int cot = rest % a.nco, kt = rest / a.nco;
    const int co0 = cot * COT, kt0 = kt * KT;
    const int kci = KT / a.kpl;
    const int ci0 = kt0 / a.kpl;

    f32x16 acc[CT][KW];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][kw][r] = 0.f;

    int t_end = (split + 1) * a.tiles_per_split;
    if (t_end > g.T) t_end = g.T;
    for (int t = split * a.tiles_per_split; t < t_end; ++t) {
        const int n = t / g.tiles_per_img, tl = t - n * g.tiles_per_img;
        const StageRegion rg = stage_region(g, tl);
        int pr, pc;
        const bool pv = tile_pixel(g, tl, lane, pr, pc);  // this lane's pixel
        const int po = pv ? pr * g.W + pc : -1;
        // dz tile -> Zt (transposed)
        {
            const float* zb = a.dz + (long)n * a.dz_bs;
            for (int co = wave; co < COT; co += 4) {
                const int m = co0 + co;
                float v = 0.f;
                if (pv && m < a.M) v = zb[(long)m * g.P + po];
                Zt[lane * ZS + co] = v;
            }
        }
        if (DW) {
            const int rsize = rg.nrows * rg.SW;
            int goff[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = tid + SMAAT_THREADS * j;
                if (e < rsize) {
                    const int sr = e / rg.SW, sc = e - sr * rg.SW;
                    const int gr = rg.row_lo + sr, gc = rg.col_lo + sc;
                    goff[j] = (gr >= 0 && gr < g.H && gc >= 0 && gc < g.W) ? gr * g.W + gc : -1;
                } else {
                    goff[j] = -2;
                }
            }
            for (int cl = 0; cl < kci; ++cl) {
                const int ci = ci0 + cl;
                const bool cv = ci < a.Cin;
                const float* plane = a.x + (long)n * a.x_bs + (long)ci * g.P;
                float sc_ = 1.f, sh_ = 0.f;
                if (AFF && cv) {
                    sc_ = a.in_scale[ci];
                    sh_ = a.in_shift[ci];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (goff[j] != -2) {
                        float v = 0.f;
                        if (cv && goff[j] >= 0) {
                            v = plane[goff[j]];
                            if (AFF) v = fmaxf(fmaf(v, sc_, sh_), 0.f);
                        }
                        S[cl * a.sstride + tid + SMAAT_THREADS * j] = v;
                    }
                }
            }
            __syncthreads();
            const int SW = rg.SW;
            const int sb = pv ? (pr - rg.row_lo) * SW + (pc - rg.col_lo) : (SW + 1);
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            for (int cl = wv; cl < kci; cl += 4) {
                const float* sp = S + cl * a.sstride + sb;
                const float s00 = sp[-SW - 1], s01 = sp[-SW], s02 = sp[-SW + 1];
                const float s10 = sp[-1], s11 = sp[0], s12 = sp[1];
                const float s20 = sp[SW - 1], s21 = sp[SW], s22 = sp[SW + 1];
                for (int j = 0; j < a.kpl; ++j) {
                    const int k = cl * a.kpl + j, kg = kt0 + k;
                    float y = 0.f;
                    if (pv && kg < a.Kdim) {
                        const float* w = a.w_dw + kg * 9;
                        y = a.b_dw ? a.b_dw[kg] : 0.f;
                        y = fmaf(w[0], s00, y);
                        y = fmaf(w[1], s01, y);
                        y = fmaf(w[2], s02, y);
                        y = fmaf(w[3], s10, y);
                        y = fmaf(w[4], s11, y);
                        y = fmaf(w[5], s12, y);
                        y = fmaf(w[6], s20, y);
                        y = fmaf(w[7], s21, y);
                        y = fmaf(w[8], s22, y);
                    }
                    Yt[lane * YS + k] = y;
                }
            }
        } else {
            const float* xb = a.x + (long)n * a.x_bs;
            for (int k = wave; k < KT; k += 4) {
                const int kg = kt0 + k;
                float v = 0.f;
                if (pv && kg < a.Kdim) v = xb[(long)kg * g.P + po];
                Yt[lane * YS + k] = v;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < PSW / 2; ++s) {
            const int px = 2 * s + half;
            float av[CT], bv[KW];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[ct] = Zt[px * ZS + (wco * CT + ct) * 32 + l31];
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) bv[kw] = Yt[px * YS + (wk * KW + kw) * 32 + l31];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw)
                    acc[ct][kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct], bv[kw], acc[ct][kw], 0, 0, 0);
        }
        __syncthreads();
    }
    float* ob = a.dwpart + (long)split * a.M * a.Kdim;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = co0 + (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < a.M) {
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const int kg = kt0 + (wk * KW + kw) * 32 + l31;
                    if (kg < a.Kdim) ob[(long)m * a.Kdim + kg] = acc[ct][kw][r];
                }
            }
        }
}

// =====================================================================================
// host-side launchers (C ABI wrappers live in capi.cpp)
// =====================================================================================
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// choose the pixel-tile geometry for a (H, W) plane and PT pixels per tile
static void choose_geom(int N, int H, int W, int PT, int smax, TileGeom* g) {
    g->H = H;
    g->W = W;
    g->P = H * W;
    g->PT = PT;
    g->mode = -1;
    const int P = H * W;
    double best = -1.0;
    // score = tile utilisation, discounted by the halo over-read, with a small bonus for
    // layouts whose rows are >= 128 B contiguous
    {   // candidate 0: PT consecutive pixels of the flattened plane
        const int rows = (PT + W - 2) / W + 3;  // rows spanned by PT consecutive pixels (worst start column) + 2 halo rows
        const int staged = rows * (W + 2);
        if (staged <= smax) {
            const int tiles = ceil_div(P, PT);
            best = (double)P / ((double)tiles * PT) * (1.0 - 0.08 * staged / PT) + 0.02;
            g->mode = 0;
            g->TH = 0;
            g->TW = 0;
            g->tiles_x = 0;
            g->tiles_per_img = tiles;
        }
    }
    const int tws[3] = {32, 16, 8};
    for (int c = 0; c < 3; ++c) {
        const int TW = tws[c], TH = PT / TW;
        const int staged = (TH + 2) * (TW + 2);
        if (TH < 1 || staged > smax) continue;
        const int tx = ceil_div(W, TW), ty = ceil_div(H, TH);
        const double sc = (double)P / ((double)tx * ty * PT) * (1.0 - 0.08 * staged / PT) + (TW == 32 ? 0.02 : 0.0);
        if (sc > best) {
            best = sc;
            g->mode = 1;
            g->TH = TH;
            g->TW = TW;
            g->tiles_x = tx;
            g->tiles_per_img = tx * ty;
        }
    }
    g->T = N * g->tiles_per_img;
}

// cached per-kernel dynamic-LDS opt-in (hipFuncSetAttribute is not free)
template <auto KERN>
static int ensure_lds(size_t lds) {
    static size_t granted = 0;  // one per kernel instantiation
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    return 0;
}

static int pw_impl();
static void choose_geom_ws(int N, int H, int W, int PT, TileGeom* g);

template <int WCO, int CT, int WPX, int PXT, int MODE>
static int launch_pwgemm_mode(PwArgs& a, hipStream_t st) {
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32;
    constexpr bool DW = MODE > 0;
    if (MODE == 0 && a.part == nullptr && (a.g.H * a.g.W) % 4 == 0 && (a.out_bs & 3) == 0 &&
        ((((uintptr_t)a.out) & 15) == 0) && !(a.dbg & 32)) {
        // transposed-accumulator form on flattened tiles
        a.g.P = a.g.H * a.g.W;
        a.g.PT = PT;
        a.g.mode = 0;
        a.g.TH = a.g.TW = a.g.tiles_x = 0;
        a.g.tiles_per_img = ceil_div(a.g.P, PT);
        a.g.T = a.N * a.g.tiles_per_img;
        a.nco = ceil_div(a.M, COT);
        a.sstride = 0;
        const size_t lds = sizeof(float) * (size_t)(KC * PT + KC * COT + BN_STAT_FLOATS(WPX, COT) + 2 * PT + COT);
        const int grid = ceil_div(a.g.T, 8) * 8 * a.nco;
        constexpr auto kern = k_pwgemm<WCO, CT, WPX, PXT, 0, true>;
        int rc = ensure_lds<kern>(lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SMAAT_THREADS), lds, st, a);
        return (int)hipGetLastError();
    }
    if (DW && pw_impl() >= 1)
        choose_geom_ws(a.N, a.g.H, a.g.W, PT, &a.g);  // same tiles (= partial-statistics slots) as the ws family
    else
        choose_geom(a.N, a.g.H, a.g.W, PT, SMAX, &a.g);
    if (a.g.mode < 0) return -1;
    a.nco = ceil_div(a.M, COT);
    // staged floats per input channel: the exact region size, padded to a multiple of 32
    int sstride = 0;
    if (DW) {
        const int rs = (a.g.mode == 0) ? ((PT + a.g.W - 2) / a.g.W + 3) * (a.g.W + 2) : (a.g.TH + 2) * (a.g.TW + 2);
        sstride = (rs + 31) & ~31;
    }
    a.sstride = sstride;
    const int kci = KC / (DW ? MODE : 1);
    const size_t lds = sizeof(float) * (size_t)(KC * PT + KC * COT + BN_STAT_FLOATS(WPX, COT) + 2 * PT + COT + (DW ? kci * sstride : 0));
    const int grid = ceil_div(a.g.T, 8) * 8 * a.nco;
    constexpr auto kern = k_pwgemm<WCO, CT, WPX, PXT, MODE>;
    int rc = ensure_lds<kern>(lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SMAAT_THREADS), lds, st, a);
    return (int)hipGetLastError();
}

template <int WCO, int CT, int WPX, int PXT>
static int launch_pwgemm_cfg(PwArgs& a, bool dw, hipStream_t st) {
    if (!dw) return launch_pwgemm_mode<WCO, CT, WPX, PXT, 0>(a, st);
    switch (a.kpl) {
        case 1: return launch_pwgemm_mode<WCO, CT, WPX, PXT, 1>(a, st);
        case 2: return launch_pwgemm_mode<WCO, CT, WPX, PXT, 2>(a, st);
        case 4: return launch_pwgemm_mode<WCO, CT, WPX, PXT, 4>(a, st);
    }
    return -1;
}

// Geometry of the wave-specialised family: a 2-D tile that the strip producer can use when one
// covers the plane well (>= 85 % of the tile pixels inside the image), else the generic choice.
static int strip_row_stride(int TW) { return TW + 8 + (TW == 16 ? 4 : 0); }  // 16-wide: +4 spreads the 4 row groups of a wave over the banks
static void choose_geom_ws(int N, int H, int W, int PT, TileGeom* g) {
    const int P = H * W;
    double best = -1.0;
    int bTW = 0;
    const int tws[3] = {32, 16, 8};
    for (int c = 0; c < 3; ++c) {
        const int TW = tws[c], TH = PT / TW;
        if (TH < 4 || (TH & 3)) continue;
        if ((TH + 2) * strip_row_stride(TW) > SMAX_WS) continue;
        if ((TH + 2) * ((TW + 8) / 4) > 108) continue;
        const int tx = ceil_div(W, TW), ty = ceil_div(H, TH);
        const double util = (double)P / ((double)tx * ty * PT);
        const double sc = util + (TW == 32 ? 0.02 : TW == 16 ? 0.01 : 0.0);
        if (util >= 0.85 && sc > best) {
            best = sc;
            bTW = TW;
        }
    }
    if (bTW == 0) {
        choose_geom(N, H, W, PT, SMAX_WS, g);
        return;
    }
    g->H = H;
    g->W = W;
    g->P = P;
    g->PT = PT;
    g->mode = 1;
    g->TW = bTW;
    g->TH = PT / bTW;
    g->tiles_x = ceil_div(W, bTW);
    g->tiles_per_img = g->tiles_x * ceil_div(H, g->TH);
    g->T = N * g->tiles_per_img;
}

static bool strip_ok(const PwArgs& a) {
    const TileGeom& g = a.g;
    if (g.mode != 1 || (g.TH & 3) || g.TH < 4 || (g.W & 3)) return false;
    if ((g.TH + 2) * strip_row_stride(g.TW) > SMAX_WS || (g.TH + 2) * ((g.TW + 8) / 4) > 108) return false;
    if ((((uintptr_t)a.x) & 15) || (a.x_bs & 3)) return false;
    return a.in_scale == nullptr;
}

template <int WCO, int CT, int WPX, int PXT, int KPL, int NPT>
static int launch_dsconv_strip_kpl(PwArgs& a, hipStream_t st) {
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32;
    constexpr int kci = KC / KPL;
    a.nco = ceil_div(a.M, COT);
    a.sstride = strip_row_stride(a.g.TW);
    const size_t lds = sizeof(float) * (size_t)(2 * KC * PT + 2 * KC * COT + 2 * 256 + BN_STAT_FLOATS(WPX, COT) + PT + COT +
                                                2 * kci * SMAX_WS);
    const int grid = ceil_div(a.g.T, 8) * 8 * a.nco;
    constexpr auto kern = k_dsconv_strip<WCO, CT, WPX, PXT, KPL, NPT>;
    int rc = ensure_lds<kern>(lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 + NPT), lds, st, a);
    return (int)hipGetLastError();
}

template <int WCO, int CT, int WPX, int PXT, int NPT>
static int launch_dsconv_strip(PwArgs& a, hipStream_t st) {
    switch (a.kpl) {
        case 1: return launch_dsconv_strip_kpl<WCO, CT, WPX, PXT, 1, NPT>(a, st);
        case 2: return launch_dsconv_strip_kpl<WCO, CT, WPX, PXT, 2, NPT>(a, st);
        case 4: return launch_dsconv_strip_kpl<WCO, CT, WPX, PXT, 4, NPT>(a, st);
    }
    return -1;
}

template <int WCO, int CT, int WPX, int PXT, int MODE, int NPT, bool AFF = false>
static int launch_pwgemm_ws_mode(PwArgs& a, hipStream_t st) {
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32;
    constexpr bool DW = MODE > 0;
    choose_geom_ws(a.N, a.g.H, a.g.W, PT, &a.g);
    if (a.g.mode < 0) return -1;
    a.nco = ceil_div(a.M, COT);
    a.sstride = SMAX_WS;
    const int kci = KC / (DW ? MODE : 1);
    const size_t lds = sizeof(float) * (size_t)(2 * KC * PT + 2 * KC * COT + 2 * 256 + BN_STAT_FLOATS(WPX, COT) + 2 * PT + COT +
                                                (DW ? 2 * kci * SMAX_WS : 0));
    const int grid = ceil_div(a.g.T, 8) * 8 * a.nco;
    constexpr auto kern = k_pwgemm_ws<WCO, CT, WPX, PXT, MODE, NPT, AFF>;
    int rc = ensure_lds<kern>(lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 + NPT), lds, st, a);
    return (int)hipGetLastError();
}

template <int WCO, int CT, int WPX, int PXT, int NPT>
static int launch_pwgemm_ws_cfg(PwArgs& a, bool dw, hipStream_t st) {
    if (!dw) return launch_pwgemm_ws_mode<WCO, CT, WPX, PXT, 0, NPT>(a, st);
    switch (a.kpl) {
        case 1: return launch_pwgemm_ws_mode<WCO, CT, WPX, PXT, 1, NPT>(a, st);
        case 2: return launch_pwgemm_ws_mode<WCO, CT, WPX, PXT, 2, NPT>(a, st);
        case 4: return launch_pwgemm_ws_mode<WCO, CT, WPX, PXT, 4, NPT>(a, st);
    }
    return -1;
}

// SMAAT_PW_IMPL: 0 = single-role kernel (k_pwgemm), 1 = wave-specialised, 4 producer waves,
// 2 = wave-specialised with 8 producer waves where the accumulator tile leaves room for them
static int pw_impl() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMAAT_PW_IMPL");
        v = e ? atoi(e) : 5;
    }
    return v;
}

static bool pw_big(int N, int H, int W) { return (long)N * H * W >= 256L * 1024; }

int smaat_pw_num_slots_impl(int N, int H, int W, int M) {
    // must mirror the tile choice of launch_pwgemm()
    TileGeom g;
    if (pw_impl() >= 1)
        choose_geom_ws(N, H, W, pw_big(N, H, W) ? 256 : 128, &g);
    else
        choose_geom(N, H, W, pw_big(N, H, W) ? 256 : 128, SMAX, &g);
    (void)M;
    return g.T;
}

int launch_pwgemm(PwArgs& a, bool dw, hipStream_t st) {
    if (a.kpl != 1 && a.kpl != 2 && a.kpl != 4) return -1;
    const bool big = pw_big(a.N, a.g.H, a.g.W);
    {
        static int abl = -1;
        if (abl < 0) {
            const char* e = getenv("SMAAT_PW_ABLATE");
            abl = e ? atoi(e) : 0;
        }
        a.dbg = abl;
    }
    if (dw && a.in_scale != nullptr && a.M <= 64 && pw_impl() >= 1) {
        // input-affine form (second half of a DoubleConvDS reading the first half's pre-BN tensor) on the
        // wave-specialised kernel for the 64-wide co tile; wider tiles use the single-role kernel below
        switch (a.kpl) {
            case 1: return big ? launch_pwgemm_ws_mode<1, 2, 4, 2, 1, 512, true>(a, st) : launch_pwgemm_ws_mode<1, 2, 4, 1, 1, 512, true>(a, st);
            case 2: return big ? launch_pwgemm_ws_mode<1, 2, 4, 2, 2, 512, true>(a, st) : launch_pwgemm_ws_mode<1, 2, 4, 1, 2, 512, true>(a, st);
            case 4: return big ? launch_pwgemm_ws_mode<1, 2, 4, 2, 4, 512, true>(a, st) : launch_pwgemm_ws_mode<1, 2, 4, 1, 4, 512, true>(a, st);
        }
    }
    const int impl = (dw && a.in_scale != nullptr) ? 0 : pw_impl();  // other input-affine shapes: single-role kernel
    // impl 5 (default): per-shape choice measured on MI355X (profiles/r1): the strip producer for
    // wide co tiles on 2-D pixel tiles, the generic wave-specialised kernel with 8 producer waves
    // for everything else that has a depthwise stage, and the single-role kernel for the plain
    // pointwise / data-gradient GEMM (its producer is trivial; 2 co-resident blocks overlap better).
    if (impl >= 4 && dw) {
        choose_geom_ws(a.N, a.g.H, a.g.W, big ? 256 : 128, &a.g);
        if (strip_ok(a) && (a.M > 64 || impl == 4)) {
            if (a.M > 64) {
                if (big) return launch_dsconv_strip<2, 2, 2, 4, 256>(a, st);  // 128 x 256
                return launch_dsconv_strip<2, 2, 2, 2, 256>(a, st);           // 128 x 128
            }
            if (big) return launch_dsconv_strip<1, 2, 4, 2, 512>(a, st);  // 64 x 256, 8 producer waves
            return launch_dsconv_strip<1, 2, 4, 1, 256>(a, st);           // 64 x 128
        }
    }
    if (impl >= 5 && !dw && a.part == nullptr) {  // (with partial statistics the tile geometry must match smaat_pw_num_slots)
        if (a.M > 64) {
            if (big) return launch_pwgemm_cfg<2, 2, 2, 4>(a, dw, st);
            return launch_pwgemm_cfg<2, 2, 2, 2>(a, dw, st);
        }
        if (big) return launch_pwgemm_cfg<1, 2, 4, 2>(a, dw, st);
        return launch_pwgemm_cfg<1, 2, 4, 1>(a, dw, st);
    }
    if (impl >= 1) {
        if (a.M > 64) {
            if (impl >= 3) {  // 8 producer waves everywhere: more waves to hide the producer's latency chains
                if (big) return launch_pwgemm_ws_cfg<2, 2, 2, 4, 512>(a, dw, st);
                return launch_pwgemm_ws_cfg<2, 2, 2, 2, 512>(a, dw, st);
            }
            if (big) return launch_pwgemm_ws_cfg<2, 2, 2, 4, 256>(a, dw, st);  // 128 x 256
            return launch_pwgemm_ws_cfg<2, 2, 2, 2, 256>(a, dw, st);           // 128 x 128
        }
        if (impl >= 2) {
            if (big) return launch_pwgemm_ws_cfg<1, 2, 4, 2, 512>(a, dw, st);  // 64 x 256, 8 producer waves
            if (impl >= 3) return launch_pwgemm_ws_cfg<1, 2, 4, 1, 512>(a, dw, st);
            return launch_pwgemm_ws_cfg<1, 2, 4, 1, 256>(a, dw, st);           // 64 x 128
        }
        if (big) return launch_pwgemm_ws_cfg<1, 2, 4, 2, 256>(a, dw, st);  // 64 x 256
        return launch_pwgemm_ws_cfg<1, 2, 4, 1, 256>(a, dw, st);           // 64 x 128
    }
    if (a.M > 64) {
        if (big) return launch_pwgemm_cfg<2, 2, 2, 4>(a, dw, st);  // 128 x 256
        return launch_pwgemm_cfg<2, 2, 2, 2>(a, dw, st);           // 128 x 128
    }
    if (big) return launch_pwgemm_cfg<1, 2, 4, 2>(a, dw, st);  // 64 x 256
    return launch_pwgemm_cfg<1, 2, 4, 1>(a, dw, st);           // 64 x 128
}

// ---- streamed weight gradient ------------------------------------------------------------
int smaat_wgrad_num_splits_impl(int N, int P, int M, int K) {
    const int total = N * ceil_div(P, 64);
    const int mt = (M > 64) ? 128 : 64;
    const int ntile = ceil_div(M, mt) * ceil_div(K, 128);
    // Workgroups per launch = ns * ntile; every workgroup writes one partial tile (32-64 KB) that the row reduction reads
    // back, so fewer, longer workgroups are cheaper as long as the chip stays full: one workgroup per CU (256), and at
    // least 8 pixel splits per tile (measured per layer, profiles/r2/wgrad_splits_r2u.txt: 5.73 -> 5.10 ms per step
    // against the round-1 target of 1024 workgroups).
    static int target = -1;
    if (target < 0) {
        const char* e = getenv("SMAAT_WGRAD_WGS");
        target = e ? atoi(e) : 256;
        if (target < 8) target = 8;
    }
    int ns = ceil_div(target, ntile);
    if (ns < 8) ns = 8;
    if (ns > total) ns = total;
    if (ns < 1) ns = 1;
    return ns;
}

template <int CT, bool VEC>
static int launch_wgrad2_cfg(Wg2Args& a, hipStream_t st) {
    constexpr int MT = 64 * CT;
    a.nmt = ceil_div(a.M, MT);
    const size_t lds = sizeof(float) * (size_t)((MT + 128) * 65);
    constexpr auto kern = k_wgrad2<CT, VEC>;
    int rc = ensure_lds<kern>(lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.nsplit, 8) * 8 * a.nmt * a.nkt), dim3(SMAAT_THREADS), lds, st, a);
    return (int)hipGetLastError();
}

int split_mode();                                             // splitmma.hip
int launch_wgrad_split(Wg2Args& a, int nt, hipStream_t st);   // splitmma.hip

int launch_wgrad2(Wg2Args& a, hipStream_t st) {
    a.nchunk_img = ceil_div(a.P, 64);
    a.total_chunks = a.N * a.nchunk_img;
    a.nsplit = smaat_wgrad_num_splits_impl(a.N, a.P, a.M, a.K);
    a.chunks_per_split = ceil_div(a.total_chunks, a.nsplit);
    a.nkt = ceil_div(a.K, 128);
    if (split_mode() >= 1) {  // bf16 matrix path (splitmma.hip); -2 = shape not handled there
        // both operand maxima given (smaat_pointwise_wgrad_h): the two-term fp16 split, unless plain-bf16 mode is on
        const int nt = (a.dz_amax && a.y_amax && split_mode() != 1) ? 2 : (split_mode() >= 2 ? 3 : 1);
        const int rc = launch_wgrad_split(a, nt, st);
        if (rc != -2) return rc;
        a.nchunk_img = ceil_div(a.P, 64);
        a.total_chunks = a.N * a.nchunk_img;
    }
    const bool vec = ((a.P & 3) == 0) && (a.P >= 4) && ((a.dz_bs & 3) == 0) && ((a.y_bs & 3) == 0) &&
                     ((((uintptr_t)a.dz) & 15) == 0) && ((((uintptr_t)a.y) & 15) == 0);
    if (a.M > 64) return vec ? launch_wgrad2_cfg<2, true>(a, st) : launch_wgrad2_cfg<2, false>(a, st);
    return vec ? launch_wgrad2_cfg<1, true>(a, st) : launch_wgrad2_cfg<1, false>(a, st);
}

static int wgrad_smax(int M, int kpl, bool dw) {
    // staged floats per input channel that fit next to the dz / Y tiles in 160 KiB of LDS
    if (!dw) return SMAXW;
    const int cot = (M > 64) ? 128 : 64;
    const int fixed = PSW * (cot + 1) + PSW * (64 + 1);
    const int kci = 64 / kpl;
    int smax = (38 * 1024 - fixed) / kci;  // 152 KiB budget in floats
    if (smax > SMAXW) smax = SMAXW;
    return smax;
}

template <int WCO, int CT, int WK, int KW>
static int launch_wgrad_cfg(WgArgs& a, bool dw, bool aff, hipStream_t st) {
    constexpr int COT = WCO * CT * 32, KT = WK * KW * 32;
    a.nco = ceil_div(a.M, COT);
    a.nkt = ceil_div(a.Kdim, KT);
    const int kci = KT / a.kpl;
    size_t lds = sizeof(float) * (size_t)(PSW * (COT + 1) + PSW * (KT + 1) + (dw ? kci * a.sstride : 0));
    const int grid = ceil_div(a.nsplit, 8) * 8 * a.nco * a.nkt;
#define LAUNCH(DWF, AFFF)                                                                                 \
    do {                                                                                                  \
        auto kern = k_wgrad<WCO, CT, WK, KW, DWF, AFFF>;                                                     \
        HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SMAAT_THREADS), lds, st, a);                             \
    } while (0)
    if (aff)
        LAUNCH(true, true);
    else
        LAUNCH(true, false);
#undef LAUNCH
    (void)dw;
    return (int)hipGetLastError();
}

// number of pixel splits used by the weight-gradient kernel (size of the partial buffer)
int smaat_dsconv_wgrad_num_splits_impl(int N, int H, int W, int M, int Kdim) {
    // independent of the tile mode: bounded by the minimal tile count N*ceil(P/PSW)
    const int tmin = N * ceil_div(H * W, PSW);
    const int cot = (M > 64) ? 128 : 64;
    const int ntile = ceil_div(M, cot) * ceil_div(Kdim, 64);
    int ns = ceil_div(2048, ntile);
    if (ns > tmin) ns = tmin;
    if (ns < 1) ns = 1;
    return ns;
}

int launch_wgrad(WgArgs& a, bool dw, hipStream_t st) {
    if (a.kpl != 1 && a.kpl != 2 && a.kpl != 4) return -1;
    const bool aff = dw && a.in_scale != nullptr;
    if (!dw) return -1;
    const int smax = wgrad_smax(a.M, a.kpl, dw);
    choose_geom(a.N, a.g.H, a.g.W, PSW, smax, &a.g);
    if (a.g.mode < 0) return -1;
    a.sstride = smax;
    a.nsplit = smaat_dsconv_wgrad_num_splits_impl(a.N, a.g.H, a.g.W, a.M, a.Kdim);
    a.tiles_per_split = ceil_div(a.g.T, a.nsplit);
    if (a.M > 64) return launch_wgrad_cfg<2, 2, 2, 1>(a, dw, aff, st);  // 128 x 64
    return launch_wgrad_cfg<2, 1, 2, 1>(a, dw, aff, st);                // 64 x 64
}

// public wrapper of the geometry chooser (used by spatial.hip)
void choose_geom_pub(int N, int H, int W, int PT, int smax, TileGeom* g) { choose_geom(N, H, W, PT, smax, g); }
