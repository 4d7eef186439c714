// Pointwise weight gradient of a DepthwiseSeparableConv with the depthwise output RECOMPUTED on the fly
// (bf16-split matrix path; round 4):
//
//   dW[m][k] = sum_{n,p} dz[n][m][p] * y[n][k][p],     y = depthwise3x3(act(x)) (+ b_dw),   k = 2 ci + j
//   (reference models/layers.py:45,47-50 backward; act = the previous BatchNorm + ReLU applied on load, optional)
//
// k_wgrad_split (splitmma.hip) streams y from HBM: 4 (M + K) HW bytes per image.  Here the producer waves stream x
// (Cin = K / 2 channels) instead and form y in registers, so the 2x-expanded depthwise tensor is neither read here nor
// written by the forward (k_dsconv_split runs without its side output): 4 (M + K / 2) HW bytes, and the forward of the
// layer loses a 4 K HW write.  The consumer side is k_wgrad_split's: 4 waves, ds_read_b128 + six v_mfma_f32_32x32x16_bf16
// per 16-pixel step on [plane][row][32 px] bf16 images, output-stationary 64 (dz rows) x 128 (y rows) tile.
//
// Producers (4 waves = 256 threads): thread (ci = t >> 2, g = t & 3) owns channel ci of the 64-channel K tile and the 8
// pixels 8g .. 8g + 7 of a 32-pixel row segment, and WALKS DOWN a band of rows of one column strip (32 columns) of one
// image: the 3 x 10 window of act(x) it needs lives in registers and slides by one row per chunk (as in dwrows.hip), one
// row = two global_load_dwordx4 + the two edge dwords (L1 / L2 hits: the neighbour thread's float4).  No LDS staging of
// x, no halo re-reads inside a band; a band costs two extra (MFMA-free) priming iterations.  Per chunk a thread computes
// y for its 2 k-rows x 8 pixels (tap order of k_dw3x3_fwd_rows: bit-identical y), splits the 16 values exactly into
// three bf16 terms and writes three 16-byte pieces per k-row into the B image; the same threads split their share of the
// dz chunk (64 rows x 32 px) into the A image.  All global loads are inline asm with counted s_waitcnt (PD row/chunk
// groups in flight), one barrier per chunk.
//
// Work list: items (image, row band, column strip), strips innermost (neighbouring strips share their halo columns and
// run on the same XCD); a workgroup (split, K tile) walks a contiguous range of items.  Partials part[split][M][K] are
// summed in fixed order in fp64 by the existing reduction (capi: launch_reduce_rows): deterministic, no atomics.
#include "common.h"
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dwg_u32x4 __attribute__((ext_vector_type(4)));
int split_mode();  // splitmma.hip

// timing ablations for experiment builds (make EXTRA=-DDWG_DBG=<bits> OUT=...; the results are wrong, only the time means
// something): 1 no LDS reads + MFMA, 2 no LDS writes, 4 no depthwise math, 8 no global loads, 16 no per-chunk barrier.
// Compile-time on purpose: the same switches as run-time flags changed the code of the normal path (hipcc drained the
// loads in flight with vmcnt(0) at the head of the branch targets).
#ifndef DWG_DBG
#define DWG_DBG 0
#endif
#define DWG_SROW 80  // bytes per LDS row: 32 bf16 + 16 B pad (conflict-free ds_read_b128, as k_wgrad_split)
#define DWG_CW 32    // pixels per chunk = width of a column strip

#include "rows_args.h"

__device__ __forceinline__ unsigned dwg_fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float dwg_bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ unsigned dwg_pack_hi16(float lo, float hi) {
    return __builtin_amdgcn_perm(dwg_fbits(hi), dwg_fbits(lo), 0x07060302u);  // {hi.hi16, lo.hi16}
}
// 4 consecutive f32 -> NT planes of 4 bf16 (one 8-byte LDS piece per plane); NT = 3: exact three-term split
// (a = p1 + p2 + p3, truncations, residuals exact), NT = 1: round to nearest even (plain bf16 operands)
typedef _Float16 dwg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dwg_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dwg_pack_f16(float a, float b) {  // one v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2_native v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dwg_f16x2));
}
__device__ __forceinline__ void dwg_split2_f16(float t0, float t1, unsigned& h, unsigned& g) {  // as splitmma.hip split2_f16
    h = dwg_pack_f16(t0, t1);
    const dwg_f16x2 hv = __builtin_bit_cast(dwg_f16x2, h);
    g = dwg_pack_f16(t0 - (float)hv.x, t1 - (float)hv.y);
}
// NT = 2 (round 5): two fp16 terms of v * sc (sc = the operand tensor's power-of-two scale, splitmma.hip)
template <int NT>
__device__ __forceinline__ void dwg_split4(const float (&v)[4], uint2 (&out)[NT], float sc = 1.f) {
    if constexpr (NT == 2) {
        dwg_split2_f16(v[0] * sc, v[1] * sc, out[0].x, out[1].x);
        dwg_split2_f16(v[2] * sc, v[3] * sc, out[0].y, out[1].y);
        return;
    }
    float p1[4], p2[4], p3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (NT == 1) {
            p1[i] = p2[i] = p3[i] = 0.f;  // (rounded below, two values per v_cvt_pk_bf16_f32)
        } else {
            p1[i] = dwg_bitsf(dwg_fbits(v[i]) & 0xFFFF0000u);
            const float r1 = v[i] - p1[i];  // exact
            p2[i] = dwg_bitsf(dwg_fbits(r1) & 0xFFFF0000u);
            p3[i] = r1 - p2[i];             // exact, <= 8 significant bits
        }
    }
    if constexpr (NT == 1)
        out[0] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));  // round to nearest even
    else
        out[0] = make_uint2(dwg_pack_hi16(p1[0], p1[1]), dwg_pack_hi16(p1[2], p1[3]));
    if constexpr (NT == 3) {
        out[1] = make_uint2(dwg_pack_hi16(p2[0], p2[1]), dwg_pack_hi16(p2[2], p2[3]));
        out[2] = make_uint2(dwg_pack_hi16(p3[0], p3[1]), dwg_pack_hi16(p3[2], p3[3]));
    }
}

typedef float dwg_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const void* dwg_uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}

// PK: the depthwise stage on packed f32 math (v_pk_fma_f32 over the two k-rows of a channel); experiment switch
// SMAAT_DWG_PK=1 (the guide prices packed VALU beside MFMAs as an anti-lever: measured, profiles/r4)
// raw registers of a 4-element row piece (ext vector types: inline-asm outputs)
typedef unsigned dwg_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct DwgRaw;
template <> struct DwgRaw<float> { typedef f32x4 type; };
template <> struct DwgRaw<bf16_t> { typedef dwg_u32x2 type; };
__device__ __forceinline__ void dwg_vals(const f32x4 v, float (&m)[4]) {
    m[0] = v[0]; m[1] = v[1]; m[2] = v[2]; m[3] = v[3];
}
__device__ __forceinline__ void dwg_vals(const dwg_u32x2 v, float (&m)[4]) {
    m[0] = bf16_lo(v[0]); m[1] = bf16_hi(v[0]); m[2] = bf16_lo(v[1]); m[3] = bf16_hi(v[1]);
}

// TX / TG: storage types of x and dz (float | bf16).  bf16 dz (mixed precision, NT = 1) is the MFMA operand as it lies in
// memory: its pieces go to the A image unconverted.
template <int NT, bool AFF, bool PK, typename TX, typename TG>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_dsconv_wgrad_split(const DsWgArgs a) {
    static_assert(sizeof(TG) == 4 || NT == 1, "bf16 gradients are plain bf16 operands");
    constexpr int MT = 64, KT = 128, ROWS = MT + KT;
    constexpr int PLSZ = ROWS * DWG_SROW, BUFSZ = NT * PLSZ;
    // load groups (rows) in flight per producer thread.  bf16 storage halves the bytes per row AND the iteration time
    // (one MFMA per product, no operand split), so four rows ahead are only ~9 MB in flight on the chip: measured latency-
    // bound (2.2 TB/s); eight rows restore the f32 build's bytes in flight
    constexpr int PD = sizeof(TX) == 2 ? 8 : 4;
    constexpr int LPG = 3;       // loads per group: x row = dwordx4 + one edge dword, dz = dwordx4
    static_assert((PD - 1) * LPG <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int wave = wv & 3;
    const int wm = wave & 1, wk = wave >> 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int kt = idx % a.nkt;
    const int split = xcd * (a.nsplit >> 3) + idx / a.nkt;  // contiguous split ranges per XCD (halo columns meet in one L2)
    // item j of this workgroup = it_lo + j * it_st (j < nitems).  Contiguous ranges (it_st = 1), or -- ilv -- the items of the
    // XCD's range dealt round-robin to its workgroups: the strips of a band then run AT THE SAME TIME on neighbouring
    // workgroups of one XCD, so the two 64-byte halves of a bf16 line (and neighbouring lines of a DRAM page) are requested
    // together instead of one band walk (~40 us) apart.
    const int s8 = a.nsplit >> 3;
    int it_lo, it_st, nitems;
    if (a.ilv) {
        const int xlo = xcd * s8 * a.ips;
        int xhi = xlo + s8 * a.ips;
        if (xhi > a.items) xhi = a.items;
        it_lo = xlo + idx / a.nkt;
        it_st = s8;
        nitems = it_lo < xhi ? (xhi - it_lo + s8 - 1) / s8 : 0;
    } else {
        it_lo = split * a.ips;
        it_st = 1;
        int it_hi = it_lo + a.ips;
        if (it_hi > a.items) it_hi = a.items;
        nitems = it_hi > it_lo ? it_hi - it_lo : 0;
    }
    const int it_hi = it_lo + nitems * it_st;  // (exclusive bound of the walk)
    // flattened iteration space: every item contributes (rows of its band) + 2 priming iterations
    const int bps = a.bands * a.strips;  // items per image
    auto item_rows = [&](int item) {
        const int band = (item % bps) / a.strips;
        const int r0 = band * a.RB;
        return (a.H - r0 < a.RB ? a.H - r0 : a.RB);
    };
    int total = 0;
    for (int i = 0; i < nitems; ++i) total += item_rows(it_lo + i * it_st) + 2;  // (wave-uniform scalar loop, <= a few dozen items)
    const int total_pad = (total + PD - 1) / PD * PD;
    int ky = 0, kdz = 0;  // NT == 2: power-of-two scale exponents of y (its maximum was left by the forward) and dz
    if constexpr (NT == 2) {
        ky = f16_kexp(amax_read(a.y_amax));
        kdz = f16_kexp(amax_read(a.dz_amax));
        asm volatile("" : "+s"(ky), "+s"(kdz));
    }

    if (producer) {
        // 8 producer waves (two per SIMD: one's VALU chain covers the other's waits).  Thread (ci, g): channel ci of the
        // 64-channel K tile, the 4 pixels 4g .. 4g + 3 of the 32-pixel row segment; the 8 threads of a channel are 8
        // consecutive lanes, so the two edge columns of a thread's 3 x 6 window are its neighbour lanes' float4 ends (DPP
        // row shifts); only g = 0 / g = 7 read theirs from memory (ONE dword load per row whose per-lane offset is fixed
        // per item: -1, +4, or 0 where there is no such column).
        const int ptid = tid - 256;
        const int ci = ptid >> 3, g = ptid & 7;
        const int cg = kt * 64 + ci;
        const bool cv = cg < a.Cin;
        const int cgc = cv ? cg : a.Cin - 1;
        float wt[2][9], bs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wt[j][k] = cv ? a.w_dw[(cgc * 2 + j) * 9 + k] : 0.f;
            bs[j] = (cv && a.b_dw) ? a.b_dw[cgc * 2 + j] : 0.f;  // channels beyond Cin: zero window, zero taps -> y = 0
        }
        float asc = AFF ? a.in_scale[cgc] : 1.f, ash = AFF ? a.in_shift[cgc] : 0.f;
        const float sy = pow2i(ky), sdz = pow2i(kdz);
        // hipcc does not see the inline-asm loads below.  A compiler-visible load that is still pending at the loop entry makes it
        // drain the vector-memory counter (s_waitcnt vmcnt(0)) at the value's first use INSIDE the loop -- once per iteration,
        // which empties the prefetch queue every chunk (found in round 4 with the ablation builds: all but two instantiations
        // did this).  Using the values here completes those loads once, before the first asm load is issued.
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(wt[j][k]));
            asm volatile("" : "+v"(bs[j]));
        }
        asm volatile("" : "+v"(asc), "+v"(ash));
        // dz share: 64 rows x 8 float4 columns = 512 pieces, one per thread; within a group of 8 rows the row order is
        // 0,4,1,5,2,6,3,7 (k_wgrad_split: the rows a 16-lane group writes with one ds_write_b64 tile the banks)
        const int q = ptid & 7, g8 = ptid >> 3;
        const int zrow = (g8 & ~7) | ((g8 & 1) << 2) | ((g8 >> 1) & 3);
        const bool zv = zrow < a.M;
        // per-lane byte offsets from the wave-uniform (scalar) row bases: no vector address arithmetic per iteration
        const unsigned vo_x = (unsigned)(cgc * a.P + 4 * g) * (unsigned)sizeof(TX);
        const unsigned vo_z = (unsigned)((zv ? zrow : 0) * a.P + 4 * q) * (unsigned)sizeof(TG);
        unsigned vo_e = vo_x;  // edge load: set per item (strip position decides whether the neighbour column exists)
        // ---- issue cursor (wave-uniform: SGPRs) ----
        int w_item = it_lo - it_st, w_j = 0, w_len = 0;
        int w_r0 = 0;
        const TX* w_xb = (const TX*)a.x;   // x + n * x_bs + c0           (row 0 of the strip, channel 0)
        const TG* w_zb = (const TG*)a.dz;  // dz + n * dz_bs + c0
        bool w_lok = false, w_rok = false;
        int w_eback = 0;  // 1 when a column exists left of the strip: the edge load's scalar base is one element back
        auto advance = [&]() __attribute__((always_inline)) {
            ++w_j;
            if (w_j >= w_len) {
                if (w_item + it_st < it_hi) {
                    w_item += it_st;
                    w_j = 0;
                    const int n = w_item / bps, rem = w_item - n * bps;
                    const int band = rem / a.strips, st_ = rem - band * a.strips;
                    w_r0 = band * a.RB;
                    const int c0 = st_ * DWG_CW;
                    w_xb = (const TX*)a.x + (long)n * a.x_bs + c0;
                    w_zb = (const TG*)a.dz + (long)n * a.dz_bs + c0;
                    w_len = (a.H - w_r0 < a.RB ? a.H - w_r0 : a.RB) + 2;
                    w_eback = __builtin_amdgcn_readfirstlane(c0 > 0 ? 1 : 0);
                    w_lok = w_eback != 0;
                    w_rok = c0 + DWG_CW < a.W;
                    // edge load: scalar base = row start - 1 element when a column exists left of the strip (the per-lane
                    // offset of a scalar-base load is UNSIGNED: it cannot reach backwards); lane g = 0 then reads column
                    // c0 - 1, lane g = 7 column c0 + 32 (or its own last column when there is none), the others their own
                    vo_e = vo_x + (g == 0 ? 0u : ((w_lok ? 1u : 0u) + (g == 7 ? (w_rok ? 4u : 3u) : 0u))) * (unsigned)sizeof(TX);
                } else {
                    w_j = w_len - 1;  // past the end: keep re-loading the last row (never consumed)
                }
            }
        };
        typename DwgRaw<TX>::type sx[PD];
        typename DwgRaw<TG>::type sz[PD];
        unsigned se[PD];
        int srow[PD];        // x row index of the set (zero padding above / below the plane)
        bool slok[PD], srok[PD];
        auto issue = [&](int set) __attribute__((always_inline)) {
            advance();
            if constexpr ((DWG_DBG & 8) != 0) {
                srow[set] = w_r0 - 1 + w_j;
                slok[set] = w_lok;
                srok[set] = w_rok;
                return;
            }
            const int xr = w_r0 - 1 + w_j;  // x row delivered by this iteration
            const int xrc = xr < 0 ? 0 : (xr >= a.H ? a.H - 1 : xr);
            const TX* xrow = (const TX*)dwg_uniform_ptr(w_xb + (long)xrc * a.W);
            if (sizeof(TX) == 4)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sx[set]) : "v"(vo_x), "s"(xrow));
            else
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(sx[set]) : "v"(vo_x), "s"(xrow));
            // (the "s" constraint does not make a value uniform: hipcc kept this pointer in VGPRs when the -1 came from a
            // select on the bool and emitted a VGPR pair in the scalar-base slot -> memory fault; built from an integer
            // that is provably wave-uniform instead)
            const TX* erow = (const TX*)dwg_uniform_ptr(xrow - w_eback);
            if (sizeof(TX) == 4)
                asm volatile("global_load_dword %0, %1, %2" : "=v"(se[set]) : "v"(vo_e), "s"(erow));
            else
                asm volatile("global_load_ushort %0, %1, %2" : "=v"(se[set]) : "v"(vo_e), "s"(erow));
            srow[set] = xr;
            slok[set] = w_lok;
            srok[set] = w_rok;
            const int zr = w_r0 + w_j - 2;  // the chunk whose window this x row completes
            const TG* zrowp = (const TG*)dwg_uniform_ptr(w_zb + (long)(zr < 0 ? 0 : zr) * a.W);
            if (sizeof(TG) == 4)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sz[set]) : "v"(vo_z), "s"(zrowp));
            else
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(sz[set]) : "v"(vo_z), "s"(zrowp));
        };
        auto wait_set = [&](int set) __attribute__((always_inline)) {
            if constexpr ((DWG_DBG & 8) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * LPG) : "memory");
            asm volatile("" : "+v"(sx[set]), "+v"(se[set]), "+v"(sz[set]));
        };
        float win[3][6];  // act(x) rows r - 1, r, r + 1 of the chunk being formed; cols 4g - 1 .. 4g + 4
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) win[r][c] = 0.f;
        int c_j = 0, c_len = 0, c_item = it_lo - it_st;  // consume cursor
        auto commit = [&](int set, int buf, bool live) __attribute__((always_inline)) {
            ++c_j;
            if (c_j >= c_len) {
                c_item += it_st;
                c_j = 0;
                c_len = item_rows(c_item) + 2;
            }
            const int xr = srow[set];
            const bool rv = cv && xr >= 0 && xr < a.H;
            float m[4];
            dwg_vals(sx[set], m);
            float e = sizeof(TX) == 4 ? dwg_bitsf(se[set]) : bf16_lo(se[set]);
            if (AFF) {  // previous BatchNorm + ReLU on load (before the exchange: neighbours hand over activated values)
#pragma unroll
                for (int c = 0; c < 4; ++c) m[c] = fmaxf(fmaf(m[c], asc, ash), 0.f);
                e = fmaxf(fmaf(e, asc, ash), 0.f);
            }
            float l = dpp_src<0x111, 0xF>(m[3]);  // row_shr:1  (lane i <- lane i - 1, inside its 16-lane row)
            float r = dpp_src<0x101, 0xF>(m[0]);  // row_shl:1  (lane i <- lane i + 1)
            l = g == 0 ? e : l;
            r = g == 7 ? e : r;
            const bool lv = rv && (g > 0 || slok[set]), rvv = rv && (g < 7 || srok[set]);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                win[0][c] = win[1][c];
                win[1][c] = win[2][c];
            }
            win[2][0] = lv ? l : 0.f;  // zero padding (after the activation)
#pragma unroll
            for (int c = 0; c < 4; ++c) win[2][1 + c] = rv ? m[c] : 0.f;
            win[2][5] = rvv ? r : 0.f;
            if (c_j < 2 || !live) return;  // priming iteration / surplus iteration of the padded walk: no chunk
            unsigned char* base = lds + buf * BUFSZ;
            // y rows k = 2 ci + j of the chunk: tap order of k_dw3x3_fwd_rows (bias, then row-major taps): bit-identical y
            float yy[2][4];
            if constexpr ((DWG_DBG & 4) != 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) yy[j][c] = win[1][1 + c] + bs[j];
            } else if (PK) {
                // both k-rows of the channel share the window value: one v_pk_fma_f32 forms {y[0][c], y[1][c]} (the same
                // fma per component, same tap order: bit-identical to the scalar form)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dwg_f32x2 acc = {bs[0], bs[1]};
#pragma unroll
                    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                        for (int tc = 0; tc < 3; ++tc) {
                            const dwg_f32x2 w2 = {wt[0][tr * 3 + tc], wt[1][tr * 3 + tc]};
                            const dwg_f32x2 x2 = {win[tr][c + tc], win[tr][c + tc]};
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
                        float acc = bs[j];
#pragma unroll
                        for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                            for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][tr * 3 + tc], win[tr][c + tc], acc);
                        yy[j][c] = acc;  // (channels beyond Cin in the last K tile: zero taps and bias -> zero rows)
                    }
            }
            if constexpr ((DWG_DBG & 2) != 0) {  // (keep the values alive)
                asm volatile("" ::"v"(yy[0][0]), "v"(yy[0][1]), "v"(yy[0][2]), "v"(yy[0][3]), "v"(yy[1][0]), "v"(yy[1][1]), "v"(yy[1][2]), "v"(yy[1][3]));
                return;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint2 pl[NT];
                dwg_split4<NT>(yy[j], pl, sy);
#pragma unroll
                for (int t = 0; t < NT; ++t) *(uint2*)(base + t * PLSZ + (MT + 2 * ci + j) * DWG_SROW + g * 8) = pl[t];
            }
            if constexpr (sizeof(TG) == 4) {  // dz share: split into the A image
                const float v4[4] = {zv ? sz[set][0] : 0.f, zv ? sz[set][1] : 0.f, zv ? sz[set][2] : 0.f, zv ? sz[set][3] : 0.f};
                uint2 pl[NT];
                dwg_split4<NT>(v4, pl, sdz);
#pragma unroll
                for (int t = 0; t < NT; ++t) *(uint2*)(base + t * PLSZ + zrow * DWG_SROW + q * 8) = pl[t];
            } else {  // bf16 gradients are the operand as stored
                *(uint2*)(base + zrow * DWG_SROW + q * 8) = zv ? make_uint2(sz[set][0], sz[set][1]) : make_uint2(0u, 0u);
            }
        };
        if (total > 0) {
#pragma unroll
            for (int s_ = 0; s_ < PD; ++s_) issue(s_);
            wait_set(0);
            commit(0, 0, true);  // iteration 0 -> buffer 0
            issue(0);
        }
        __syncthreads();
        // The walk runs over total rounded up to a multiple of PD, every slot of the unrolled body unconditional: the surplus
        // iterations wait for and re-issue rows `advance` keeps re-loading and commit nothing (live = false: the LDS writes are
        // skipped, the loads are not).  No slot's loads can be skipped, so on EVERY path of the control-flow graph a set is
        // waited for exactly PD issues after it was issued -- which is what scripts/isa_hazards.py proves on the generated ISA
        // (a conditional tail makes paths on which a slot's commit follows a skipped issue: infeasible at run time, but a
        // path-insensitive proof cannot know, and round 5 shipped an instantiation nobody could vouch for).
        for (int t0 = 0; t0 < total_pad; t0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                wait_set((u + 1) % PD);              // iteration t0 + u + 1: issued PD iterations ago
                commit((u + 1) % PD, (t0 + u + 1) & 1, t0 + u + 1 < total);
                issue((u + 1) % PD);
                if constexpr ((DWG_DBG & 16) == 0) __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus loads of the tail still target live registers
    } else {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // which flattened iterations carry a chunk: the same walk, scalar
        int c_item = it_lo - it_st, c_j = 0, c_len = 0;
        __syncthreads();
        for (int t = 0; t < total_pad; ++t) {  // (the producers' barrier count: total rounded up to the unroll depth)
            ++c_j;
            if (c_j >= c_len) {
                c_item += it_st;
                c_j = 0;
                c_len = item_rows(c_item) + 2;
            }
            if (t < total && c_j >= 2 && (DWG_DBG & 1) == 0) {
                const unsigned char* base = lds + (t & 1) * BUFSZ;
                const unsigned char* ap = base + (wm * 32 + l31) * DWG_SROW + half * 16;
                const unsigned char* bp = base + (MT + (wk * 2) * 32 + l31) * DWG_SROW + half * 16;
#pragma unroll
                for (int s = 0; s < DWG_CW / 16; ++s) {
                    bf16x8 af[NT], bf[2][NT];
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) af[tt] = *(const bf16x8*)(ap + tt * PLSZ + s * 32);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) bf[j][tt] = *(const bf16x8*)(bp + tt * PLSZ + j * 32 * DWG_SROW + s * 32);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (NT == 2) {  // two-term fp16 split: h g' + g h' + h h'
                            const dwg_f16x8 a0 = __builtin_bit_cast(dwg_f16x8, af[0]), a1 = __builtin_bit_cast(dwg_f16x8, af[1]);
                            const dwg_f16x8 b0 = __builtin_bit_cast(dwg_f16x8, bf[j][0]), b1 = __builtin_bit_cast(dwg_f16x8, bf[j][1]);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[j], 0, 0, 0);
                            continue;
                        }
                        if constexpr (NT == 3) {  // smallest terms first (the order of k_wgrad_split)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][2], acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[j][0], acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[j][1], acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][1], acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[j][0], acc[j], 0, 0, 0);
                        }
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][0], acc[j], 0, 0, 0);
                    }
                }
            }
            if constexpr ((DWG_DBG & 16) == 0) __syncthreads();
        }
        float* ob = a.part + (long)split * a.M * a.K;
        const float e1 = pow2i(-((ky + kdz) / 2)), e2 = pow2i(-((ky + kdz) - (ky + kdz) / 2));  // (NT == 2: exponent split evenly)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < a.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kg = kt * KT + (wk * 2 + j) * 32 + l31;
                    if (kg < a.K) ob[(long)m * a.K + kg] = NT == 2 ? acc[j][r] * e1 * e2 : acc[j][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
static void dswg_geom(DsWgArgs& a) {
    const int slots = 256;  // one workgroup per CU
    a.P = a.H * a.W;
    a.strips = a.W / DWG_CW;
    a.nkt = (a.K + 127) / 128;
    // band length: every band costs two priming iterations, and a workgroup walks ceil(items / workgroups) items -- pick the
    // divisor of H (12 .. 64 rows) that minimises the iterations of the busiest workgroup (288 rows at batch 32: 36-row
    // bands = 9 items per workgroup exactly, 342 iterations; 48-row bands would leave 7 / 6 items, 350)
    const int wgs = (slots / a.nkt) & ~7 ? (slots / a.nkt) & ~7 : 8;
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
    int ns = (slots / a.nkt) & ~7;  // one workgroup per CU (92 KB of LDS), split ranges contiguous per XCD
    if (ns < 8) ns = 8;
    while (ns > 8 && ns > a.items) ns -= 8;
    a.nsplit = ns;
    a.ips = (a.items + ns - 1) / ns;
}

// shapes this kernel takes: kernels_per_layer = 2, W a multiple of 32, at most 64 output channels, 16-byte aligned rows
int dsconv_wgrad_split_ok(int kpl, int M, int H, int W) { return kpl == 2 && (W % DWG_CW) == 0 && M <= 64 && H >= 1; }

int dsconv_wgrad_split_num_splits(int N, int Cin, int M, int H, int W) {
    DsWgArgs a{};
    a.N = N; a.Cin = Cin; a.K = 2 * Cin; a.M = M; a.H = H; a.W = W;
    dswg_geom(a);
    return a.nsplit;
}

template <int NT, bool AFF, bool PK, typename TX, typename TG>
static int launch_dswg_cfg(const DsWgArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * NT * (64 + 128) * DWG_SROW;
    constexpr auto kern = k_dsconv_wgrad_split<NT, AFF, PK, TX, TG>;
    static size_t granted = 0;
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    hipLaunchKernelGGL(kern, dim3(a.nsplit * a.nkt), dim3(768), lds, st, a);
    return (int)hipGetLastError();
}

// x_dt / dz_dt: SMAAT_F32 | SMAAT_BF16 (built: everything f32; bf16 dz with x bf16 or f32 -- mixed precision, plain bf16
// operands).  returns -2 when the shape / alignment / type combination is not handled (the caller keeps the depthwise
// output and streams it)
int launch_dsconv_wgrad_split(DsWgArgs& a, int kpl, int x_dt, int dz_dt, hipStream_t st) {
    if (!dsconv_wgrad_split_ok(kpl, a.M, a.H, a.W) || a.K != 2 * a.Cin) return -2;
    const int xe = x_dt == SMAAT_BF16 ? 2 : 4, ze = dz_dt == SMAAT_BF16 ? 2 : 4;
    if ((a.x_bs & 3) || (a.dz_bs & 3) || (((uintptr_t)a.x) & (4 * xe - 1)) || (((uintptr_t)a.dz) & (4 * ze - 1))) return -2;
    if ((long)a.Cin * a.H * a.W * xe >= (1L << 32) || (long)a.M * a.H * a.W * ze >= (1L << 32)) return -2;
    dswg_geom(a);
    {
        static int ilv = -1;  // SMAAT_ROWS_ILV=0: contiguous item ranges (A/B timing)
        if (ilv < 0) {
            const char* e = getenv("SMAAT_ROWS_ILV");
            ilv = e ? atoi(e) : 1;
        }
        a.ilv = ilv;
    }
    const bool aff = a.in_scale != nullptr;
    static int pk = -1;
    if (pk < 0) {
        const char* e = getenv("SMAAT_DWG_PK");
        pk = (e && e[0] == '1') ? 1 : 0;
    }
    if (dz_dt == SMAAT_BF16) {
        if (x_dt == SMAAT_BF16) {
            if (pk) return aff ? launch_dswg_cfg<1, true, true, bf16_t, bf16_t>(a, st) : launch_dswg_cfg<1, false, true, bf16_t, bf16_t>(a, st);
            return aff ? launch_dswg_cfg<1, true, false, bf16_t, bf16_t>(a, st) : launch_dswg_cfg<1, false, false, bf16_t, bf16_t>(a, st);
        }
        return aff ? launch_dswg_cfg<1, true, false, float, bf16_t>(a, st) : launch_dswg_cfg<1, false, false, float, bf16_t>(a, st);
    }
    if (x_dt != SMAAT_F32) return -2;
    if (split_mode() == 1)
        return aff ? launch_dswg_cfg<1, true, false, float, float>(a, st) : launch_dswg_cfg<1, false, false, float, float>(a, st);
    if (a.y_amax && a.dz_amax) {  // both operand maxima at hand: two-term fp16 split (three MFMAs per product)
        // AFF: the PACKED-math build (v_pk_fma_f32 over the two k-rows of a channel; bit-identical y, 100 VGPRs).  The scalar-math
        // AFF build of this instantiation comes out of hipcc 7.2 at 132 VGPRs and is WRONG on the GPU: results differ from call to
        // call by 1e-3 (scripts/probes/dswgrad_h_debug.py, profiles/r5/dswgrad_h_aff_scalar_build_nondeterministic.txt) while the
        // ISA shows the same loads, waits and barriers as the correct builds -- not understood, not shipped (never instantiated).
#ifdef DWG_SCALAR_AFF  // experiment builds only (profiles/r6/): the scalar-math AFF instantiation
        if (aff) return launch_dswg_cfg<2, true, false, float, float>(a, st);
#endif
        if (aff) return launch_dswg_cfg<2, true, true, float, float>(a, st);
        return launch_dswg_cfg<2, false, false, float, float>(a, st);
    }
    if (pk) return aff ? launch_dswg_cfg<3, true, true, float, float>(a, st) : launch_dswg_cfg<3, false, true, float, float>(a, st);
    return aff ? launch_dswg_cfg<3, true, false, float, float>(a, st) : launch_dswg_cfg<3, false, false, float, float>(a, st);
}
