// Depthwise 3x3 forward / backward as register row-streaming kernels (gfx950).
//
// reference: models/layers.py:38-44,48 (nn.Conv2d(in, in * kernels_per_layer, 3, padding=1, groups=in)) and its ATen
// backward (input gradient = transposed depthwise convolution summed over the kernels_per_layer outputs of a group,
// weight gradient = correlation of the input with dY, bias gradient = sum of dY).
//
// The strip kernels of spatial.hip stage a halo tile through LDS between two barriers and give one thread a 1 x 4
// pixel strip: dword global accesses, 18 ds_read_b32 per strip and channel, 133 VGPRs in the backward (3 waves per
// SIMD) with the loads of x exposed inside the tile loop -- 3.5 TB/s (backward) and 4.3 TB/s (forward) of algorithmic
// traffic on MI355X (profiles/r2).  These kernels need no LDS staging and no barrier in the main loop:
//   * a thread owns FOUR adjacent columns (one float4) of a band of BH rows of one plane and walks down the band;
//   * the 3 x 6 window it needs (its float4 and the two neighbouring columns, three rows) lives in registers and
//     slides by one row per step: per row one global_load_dwordx4 + two edge dwords (L1 hits: the neighbour lane's
//     float4) per channel, one global_store_dwordx4 per output channel;
//   * forward: the two rows after next are always in flight (three rotating windows + three raw load sets, unroll by 3);
//   * a wave never straddles two planes, so the per-channel weights live in SGPRs; workgroups are just four
//     consecutive waves of the (plane, wave-in-plane) list -- no LDS, no barrier anywhere;
//   * the weight / bias gradient partials reduce with DPP wave sums: one partial row per (image, wave of the plane),
//     part[n * wpp + w][Cdw][10], rpart[2][n * wpp + w][Cin]  (dw_bwd_groups() == wpp).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "common.h"

#define DW_LIN_MIN_PLANE 5184  // (72 x 72; see launch_dw3x3_fwd_rows)

struct DwrGeom {
    int H, W, P, ncol4, nbands, BH, wpp;
    int seg, ppw;  // plane packing (small planes): a wave holds ppw = 64 / seg planes, one per segment of seg lanes
};

static int dwr_pack_enabled() {  // SMAAT_DW_PACK=0: A/B timing of the plane packing
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMAAT_DW_PACK");
        v = e ? atoi(e) : 1;
    }
    return v;
}

// the (image, channel, lane-in-plane index) of a lane.  PACK: the wave's ppw planes are the SAME channel of ppw consecutive
// images (the channel, hence the weights and the BatchNorm coefficients, stay wave-uniform: SGPRs); lanes of images beyond
// the batch walk the last image with every output masked.  Returns false when the whole wave has nothing to do.
template <bool PACK>
__device__ __forceinline__ bool dwr_place(const DwrGeom& g, int gw, int lane, int N, int Cin, int& n, int& ci, int& wip, int& t,
                                          bool& lane_on) {
    lane_on = true;
    if constexpr (PACK) {
        const int grp = gw / Cin;
        if (grp * g.ppw >= N) return false;
        ci = gw - grp * Cin;
        const int pi = lane / g.seg;
        t = lane - pi * g.seg;
        wip = 0;
        n = grp * g.ppw + pi;
        if (n >= N) {
            n = N - 1;
            lane_on = false;
        }
    } else {
        const int plane = gw / g.wpp;
        wip = gw - plane * g.wpp;
        if (plane >= N * Cin) return false;
        n = plane / Cin;
        ci = plane - n * Cin;
        t = wip * 64 + lane;
    }
    return true;
}

// band / wave decomposition of a plane: maximise lane utilisation x (BH / (BH + 2)) x chip fill
DwrGeom dw_rows_geom(int N, int Cin, int H, int W) {
    const long planes = (long)N * Cin;
    DwrGeom g;
    g.H = H;
    g.W = W;
    g.P = H * W;
    g.ncol4 = (W + 3) / 4;  // W % 4 == 2: the last column group of a row holds two columns (DwrLane::part)
    double best = -1.0;
    g.nbands = 1;
    g.BH = H;
    g.wpp = 0;
    g.seg = 64;
    g.ppw = 1;
    for (int nb = 1; nb <= H && (long)nb * g.ncol4 <= 64 * 64; ++nb) {
        const int bh = (H + nb - 1) / nb;
        if ((nb - 1) * bh >= H) continue;  // an empty trailing band: a smaller nb gives the same BH
        if (bh < 4 && nb > 1) break;
        const int T = g.ncol4 * nb;
        const int wpp = (T + 63) / 64;
        const double util = (double)T / (64.0 * wpp);
        const double halo = (double)bh / (bh + 2.0);
        double fill = (double)planes * wpp / 8192.0;
        if (fill > 1.0) fill = 1.0;
        const double score = util * halo * (0.25 + 0.75 * fill);
        if (score > best) {
            best = score;
            g.nbands = nb;
            g.BH = bh;
            g.wpp = wpp;
        }
    }
    // Plane packing (round 4): a plane whose (band, column group) list fills at most half a wave -- the 18 x 18 planes of
    // the bottleneck: 20 of 64 lanes, one short wave per plane, 0.8 TB/s -- shares the wave with the same channel's plane
    // of the next image(s) (dwr_place).  A plane takes one DPP row (16 lanes) or two (32), so that the per-plane sums of
    // the backward are the row / half-wave DPP sums.
    for (int nb = 1; nb <= H && nb * g.ncol4 <= 32; ++nb) {
        const int bh = (H + nb - 1) / nb;
        if ((nb - 1) * bh >= H) continue;
        if (bh < 4 && nb > 1) break;
        const int T = g.ncol4 * nb;
        const int seg = T <= 16 ? 16 : 32, ppw = 64 / seg;
        const long waves = (long)((N + ppw - 1) / ppw) * Cin;
        const double util = (double)T * (double)planes / (64.0 * (double)waves);
        const double halo = (double)bh / (bh + 2.0);
        double fill = (double)waves / 8192.0;
        if (fill > 1.0) fill = 1.0;
        const double score = util * halo * (0.25 + 0.75 * fill);
        if (score > best && dwr_pack_enabled()) {
            best = score;
            g.nbands = nb;
            g.BH = bh;
            g.wpp = 1;
            g.seg = seg;
            g.ppw = ppw;
        }
    }
    return g;
}

int dw_rows_wpp(int N, int Cin, int H, int W) { return dw_rows_geom(N, Cin, H, W).wpp; }

__device__ __forceinline__ float dwr_act(float v, bool aff, float sc, float sh) {
    return aff ? fmaxf(fmaf(v, sc, sh), 0.f) : v;
}

// One row of the 6-wide window of a float4 column group: cols 4q-1 .. 4q+4 (zero outside the plane).
// The two edge columns are the neighbour lanes' float4 ends (DPP wave shifts by one lane: the neighbour lane holds
// the neighbour column group of the SAME band, hence the same row); only lane 0 / lane 63 read theirs from memory,
// both with one dword load whose per-thread offset `eo` (-1, +4 or 0) is fixed.
// W % 4 == 2 (the 18 x 18 planes of the bottleneck): the LAST group of a row has only the columns 4q, 4q + 1.  Its
// 4-element access is shifted two columns to the left (cols 4q - 2 .. 4q + 1: never beyond the row, hence never beyond the
// tensor), the two valid values are moved to positions 0, 1 of the group and positions 2, 3 are the zero padding right of
// the plane; its store writes two elements.  Rows are then only 8-byte (f32) / 4-byte (bf16) aligned: the hardware takes
// unaligned dwordx4 / dwordx2 global accesses.
struct DwrLane {
    int eo;         // edge element offset of this lane's extra load
    bool l0, l63;   // lane 0 / lane 63 of the wave
    bool lok, rok;  // a column exists to the left / right of the group
    bool part;      // the two-column last group of a row (W % 4 == 2)
    int lo;         // element offset of the group's 4-element access: -2 for a partial group, else 0
};
// PART (template parameter of the kernels): W % 4 == 2.  The W % 4 == 0 instantiations contain none of the partial-group
// selects and no branch around their stores (a divergent store makes hipcc drain the loads in flight with vmcnt(0): the
// forward kernel lost a third of its bandwidth to that when the distinction was a run-time flag).
template <bool PART>
__device__ __forceinline__ DwrLane dwr_lane(int lane, int q, int ncol4) {
    DwrLane ln;
    ln.l0 = lane == 0;
    ln.l63 = lane == 63;
    ln.lok = q > 0;
    ln.rok = q < ncol4 - 1;
    ln.part = PART && q == ncol4 - 1;
    ln.lo = (PART && ln.part) ? -2 : 0;
    ln.eo = (ln.l0 && ln.lok) ? -1 : ((ln.l63 && ln.rok) ? 4 : 0);
    return ln;
}
// the four values of a group from its (possibly shifted) access: partial group -> (v.z, v.w, 0, 0)
template <bool PART>
__device__ __forceinline__ float4 dwr_group(const float4 v, const DwrLane& ln) {
    if (!PART) return v;
    return ln.part ? make_float4(v.z, v.w, 0.f, 0.f) : v;
}
// T = element type of the streamed tensor (float or bf16_t, common.h "element types"): a row is ONE 16- or 8-byte load
// per lane plus the edge element; the raw registers are converted to f32 in dwr_finish, after the pin.
template <typename T>
struct DwrRaw {
    typename Elem<T>::raw4 m;
    typename Elem<T>::raw1 e;
};
// issue the two loads of a window row (p = plane + 4q; nothing waits here)
template <typename T>
__device__ __forceinline__ DwrRaw<T> dwr_issue(const T* __restrict__ p, int r, int H, int W, const DwrLane& ln) {
    const int rc = min(max(r, 0), H - 1);
    const T* pr = p + (long)rc * W;
    DwrRaw<T> v;
    v.m = ldraw4(pr + ln.lo);
    v.e = ldraw1(pr + ln.eo);
    return v;
}
template <typename T>
__device__ __forceinline__ typename Elem<T>::raw4 dwr_issue4(const T* __restrict__ p, int r, int H, int W, const DwrLane& ln) {
    const int rc = min(max(r, 0), H - 1);
    return ldraw4(p + (long)rc * W + ln.lo);
}
// store the group's values of one row (two of them for a partial group)
template <bool PART, typename T>
__device__ __forceinline__ void dwr_store(T* p, const float4 v, const DwrLane& ln) {
    if (PART && ln.part)
        st2(p, v.x, v.y);
    else
        st4(p, v);
}
// pin the loaded registers: keeps hipcc from sinking parts of a 16-byte load into the row-validity select (it splits
// the load into dword loads plus a branch otherwise).  Call after ALL loads of a step have been issued.
__device__ __forceinline__ void dwr_pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void dwr_pin(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void dwr_pin(DwrRaw<float>& v) {
    asm volatile("" : "+v"(v.m.x), "+v"(v.m.y), "+v"(v.m.z), "+v"(v.m.w), "+v"(v.e));
}
__device__ __forceinline__ void dwr_pin(DwrRaw<bf16_t>& v) { asm volatile("" : "+v"(v.m.x), "+v"(v.m.y), "+v"(v.e)); }
// raw row -> window row: edge exchange, activation (previous BatchNorm + ReLU) on load, zero padding
template <bool PART, typename T>
__device__ __forceinline__ void dwr_finish(float (&w)[6], const DwrRaw<T>& vr, int r, int H, const DwrLane& ln, bool aff,
                                           float sc, float sh) {
    const bool rv = r >= 0 && r < H;
    const float4 m = dwr_group<PART>(cvt4(vr.m), ln);
    const float e = cvt1(vr.e);
    float l = dpp_src<0x138, 0xF>(m.w);   // wave_shr:1  (lane i <- lane i - 1)
    float rr = dpp_src<0x130, 0xF>(m.x);  // wave_shl:1  (lane i <- lane i + 1)
    l = ln.l0 ? e : l;
    rr = ln.l63 ? e : rr;
    w[0] = (rv && ln.lok) ? dwr_act(l, aff, sc, sh) : 0.f;
    w[1] = rv ? dwr_act(m.x, aff, sc, sh) : 0.f;
    w[2] = rv ? dwr_act(m.y, aff, sc, sh) : 0.f;
    const bool rv2 = PART ? (rv && !ln.part) : rv;  // (zero padding right of the plane: after the activation)
    w[3] = rv2 ? dwr_act(m.z, aff, sc, sh) : 0.f;
    w[4] = rv2 ? dwr_act(m.w, aff, sc, sh) : 0.f;
    w[5] = (rv && ln.rok) ? dwr_act(rr, aff, sc, sh) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// forward:  y[ci*KPL + j][r][c] = b[j] + sum_{tr,tc} w[j][tr][tc] * act(x)[ci][r + tr - 1][c + tc - 1]
// ---------------------------------------------------------------------------------------------------------------
// TX / TY: element types of x and y (f32 | bf16 storage; the arithmetic is f32 either way)
template <int KPL, typename TX, typename TY, bool PART, bool PACK = false>
__global__ __launch_bounds__(256) void k_dw3x3_fwd_rows(const TX* __restrict__ x, long x_bs,
                                                         const float* __restrict__ w_dw,
                                                         const float* __restrict__ b_dw, TY* __restrict__ y,
                                                         long y_bs, int Cin, int nplanes, const DwrGeom g,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift,
                                                         unsigned* __restrict__ amax) {
    // amax (nullable): max |y| over the whole tensor, for the two-term fp16 split GEMMs that read y (common.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (tid >> 6));  // (plane, wave of the plane) list
    int n, ci, wip, t;
    bool lane_on;
    if (!dwr_place<PACK>(g, gw, lane, nplanes / Cin, Cin, n, ci, wip, t, lane_on)) return;  // (whole wave)
    float am = 0.f;
    const int band_ = t / g.ncol4, q = t - band_ * g.ncol4;
    const bool active = lane_on && band_ < g.nbands;  // surplus lanes walk the last band again (stores masked)
    const int band = active ? band_ : g.nbands - 1;
    const int r0 = band * g.BH;
    const TX* xp = x + (long)n * x_bs + (long)ci * g.P + 4 * q;
    TY* yp = y + (long)n * y_bs + (long)(ci * KPL) * g.P + 4 * q;
    float wt[KPL][9], bs[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[j][k] = w_dw[(ci * KPL + j) * 9 + k];
        bs[j] = b_dw ? b_dw[ci * KPL + j] : 0.f;
    }
    const bool aff = in_scale != nullptr;
    const float asc = aff ? in_scale[ci] : 1.f, ash = aff ? in_shift[ci] : 0.f;
    const DwrLane ln = dwr_lane<PART>(lane, q, g.ncol4);

    // before step r (u = (r - r0) % 3): Wn[u] = row r - 1, Wn[u + 1] = row r, raw[u] = row r + 1 and raw[u + 1] = row
    // r + 2 in flight; the step issues row r + 3 into raw[u + 2], then finishes row r + 1 into Wn[u + 2] (the slot of the
    // dead row r - 2)   (indices mod 3: two rows of loads stay in flight behind the one being consumed)
    float Wn[3][6];
    DwrRaw<TX> raw[3];
    auto compute = [&](int r, const float (&R0)[6], const float (&R1)[6], const float (&R2)[6]) {
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = bs[j];  // tap order = the strip kernel's and the oracle's (row-major): same rounding
#pragma unroll
                for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][tc], R0[c + tc], acc);
#pragma unroll
                for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][3 + tc], R1[c + tc], acc);
#pragma unroll
                for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][6 + tc], R2[c + tc], acc);
                o[c] = acc;
            }
            if (active && r < g.H) {
                dwr_store<PART>(yp + (long)j * g.P + (long)r * g.W, make_float4(o[0], o[1], o[2], o[3]), ln);
                float m = fmaxf(fabsf(o[0]), fabsf(o[1]));
                if (!(PART && ln.part)) m = fmaxf(m, fmaxf(fabsf(o[2]), fabsf(o[3])));  // (a 2-wide last group stores o[0..1])
                am = fmaxf(am, m);
            }
        }
    };
    {
        DwrRaw<TX> a = dwr_issue(xp, r0 - 1, g.H, g.W, ln), b = dwr_issue(xp, r0, g.H, g.W, ln);
        raw[0] = dwr_issue(xp, r0 + 1, g.H, g.W, ln);
        raw[1] = dwr_issue(xp, r0 + 2, g.H, g.W, ln);
        dwr_pin(a);
        dwr_pin(b);
        dwr_finish<PART>(Wn[0], a, r0 - 1, g.H, ln, aff, asc, ash);
        dwr_finish<PART>(Wn[1], b, r0, g.H, ln, aff, asc, ash);
    }
    for (int i = 0; i < g.BH; i += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (i + u < g.BH) {
                const int r = r0 + i + u;
                raw[(u + 2) % 3] = dwr_issue(xp, r + 3, g.H, g.W, ln);
                dwr_pin(raw[u]);  // row r + 1 (issued two steps ago)
                dwr_finish<PART>(Wn[(u + 2) % 3], raw[u], r + 1, g.H, ln, aff, asc, ash);
                compute(r, Wn[u], Wn[(u + 1) % 3], Wn[(u + 2) % 3]);
            }
        }
    }
    dwr_pin(raw[0]);  // (loads still in flight target live registers)
    dwr_pin(raw[1]);
    dwr_pin(raw[2]);
    if (amax) amax_publish_wave(amax, am, (unsigned)gw);  // (waves are independent here: some have returned already)
}

// ---------------------------------------------------------------------------------------------------------------
// forward, NOT walking (round 5): a lane owns ONE output position (row r, column group q) of its input channel, loads the
// three window rows r - 1, r, r + 1 itself (float4 + the edge element each; the vertical re-use the walker keeps in
// registers is left to L1 / L2) and stores its KPL output groups; waves in (plane, row, column group) order, so that what
// the memory system sees is a window sweeping through x and y in address order instead of one 36-row stream per resident
// wave.  As pure data movement that is 6.06 against 4.86 TB/s on the 64 x 288 x 288 shape (scripts/probes/
// dw_shape_probe.hip).  Same window construction (dwr_finish) and the same tap order as k_dw3x3_fwd_rows: bit-identical y.
// W % 4 == 0; a plane takes a whole number of waves (the channel, hence weights and coefficients, stay wave-uniform).
// ---------------------------------------------------------------------------------------------------------------
template <int KPL, typename TX, typename TY>
__global__ __launch_bounds__(256) void k_dw3x3_fwd_lin(const TX* __restrict__ x, long x_bs, const float* __restrict__ w_dw,
                                                        const float* __restrict__ b_dw, TY* __restrict__ y, long y_bs, int Cin,
                                                        int nplanes, int H, int W, int wpp, const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, unsigned* __restrict__ amax) {
    __shared__ float amred[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const long gw = (long)blockIdx.x * 4 + (tid >> 6);
    const int plane_ = (int)(gw / wpp);
    const bool wave_on = plane_ < nplanes;
    const int plane = __builtin_amdgcn_readfirstlane(wave_on ? plane_ : nplanes - 1);
    const int wip = (int)(gw - (long)plane_ * wpp);
    const int ncol4 = W >> 2, per = H * ncol4, P = H * W;
    const int t_ = wip * 64 + lane;
    const bool on = wave_on && t_ < per;
    const int t = t_ < per ? t_ : per - 1;
    const int r = t / ncol4, q = t - r * ncol4;
    const int n = plane / Cin, ci = plane - n * Cin;
    const TX* xp = x + (long)n * x_bs + (long)ci * P + 4 * q;
    TY* yp = y + (long)n * y_bs + (long)(ci * KPL) * P + (long)r * W + 4 * q;
    float wt[KPL][9], bs[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[j][k] = w_dw[(ci * KPL + j) * 9 + k];
        bs[j] = b_dw ? b_dw[ci * KPL + j] : 0.f;
    }
    const bool aff = in_scale != nullptr;
    const float asc = aff ? in_scale[ci] : 1.f, ash = aff ? in_shift[ci] : 0.f;
    const DwrLane ln = dwr_lane<false>(lane, q, ncol4);
    DwrRaw<TX> ra = dwr_issue(xp, r - 1, H, W, ln), rb = dwr_issue(xp, r, H, W, ln), rc = dwr_issue(xp, r + 1, H, W, ln);
    dwr_pin(ra);
    dwr_pin(rb);
    dwr_pin(rc);
    float R0[6], R1[6], R2[6];
    dwr_finish<false>(R0, ra, r - 1, H, ln, aff, asc, ash);
    dwr_finish<false>(R1, rb, r, H, ln, aff, asc, ash);
    dwr_finish<false>(R2, rc, r + 1, H, ln, aff, asc, ash);
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = bs[j];  // (tap order of k_dw3x3_fwd_rows)
#pragma unroll
            for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][tc], R0[c + tc], acc);
#pragma unroll
            for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][3 + tc], R1[c + tc], acc);
#pragma unroll
            for (int tc = 0; tc < 3; ++tc) acc = fmaf(wt[j][6 + tc], R2[c + tc], acc);
            o[c] = acc;
        }
        if (on) {
            st4(yp + (long)j * P, make_float4(o[0], o[1], o[2], o[3]));
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (amax) amax_publish_block256(amax, am, blockIdx.x, amred);  // (block-uniform branch; no wave has returned)
}

// ---------------------------------------------------------------------------------------------------------------
// backward:
//   dX[ci][q]        = sum_j sum_tap w[j][tap] * dY[ci*KPL + j][q - off(tap)]
//   part[n][k][tap]  = sum_q xact[ci][q] * dY[k][q - off(tap)]      (tap < 9),   part[n][k][9] = sum_q dY[k][q]
//   rpart (nullable; needs in_scale / in_shift, x = the PRE-BatchNorm tensor z of the previous half-block):
//     rpart[0][n][ci] = sum g,  rpart[1][n][ci] = sum g * (z - mean) * invstd,  g = dX * [xact > 0]
// Scatter form: the dY row rho that arrives at a step contributes to the dX rows rho - 1, rho, rho + 1 (tap rows 0, 1,
// 2) and, with the x rows of the same three lines, to the weight gradient.  Live state: three dX accumulator rows,
// three activated x rows (zero outside the band, so that neighbouring bands do not count a product twice), ONE dY
// window row per channel.  A dX row is complete after the step of the row below it.
// The loads of a step are issued at its start and consumed in the same step: the latency is covered by occupancy (a
// variant that kept the next row in flight needed 128 VGPRs + spills and ran 1.4x slower; profiles/r2/dw_bench_r2m).
// ---------------------------------------------------------------------------------------------------------------
#ifndef DWR_BWD_PREFETCH
#define DWR_BWD_PREFETCH 1
#endif
#ifndef DWR_BWD_WAVES
#define DWR_BWD_WAVES 4  // 5 needs spills (18 VGPRs) and runs 1.3x slower; profiles/r2/dw_bench_r2n
#endif
// RP: the variant that also emits rpart keeps the raw (pre-BatchNorm) rows of the three open lines in registers: the
// counter passes showed the re-load of the completed row as +25 % HBM fetch (it had left the L2 two steps later).
// TX / TG / TD: element types of x (or z), dY and dX
// PACK: plane packing (dw_rows_geom, dwr_place)
template <int KPL, bool RP, typename TX, typename TG, typename TD, bool PART, bool PACK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DWR_BWD_WAVES, DWR_BWD_WAVES))) void k_dw3x3_bwd_rows(const TX* __restrict__ x, long x_bs,
                                                         const TG* __restrict__ dy, long dy_bs,
                                                         const float* __restrict__ w_dw, TD* __restrict__ dx,
                                                         long dx_bs, float* __restrict__ part, int Cin, int nplanes,
                                                         int N, const DwrGeom g, const float* __restrict__ bn_mean,
                                                         const float* __restrict__ bn_invstd,
                                                         float* __restrict__ rpart,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (tid >> 6));  // (plane, wave of the plane) list
    int n, ci, wip, t;
    bool lane_on;
    if (!dwr_place<PACK>(g, gw, lane, N, Cin, n, ci, wip, t, lane_on)) return;  // (whole wave; no barrier below)
    const int band_ = t / g.ncol4, q = t - band_ * g.ncol4;
    const bool active = lane_on && band_ < g.nbands;  // the other lanes walk a valid band with every output masked
    const int band = band_ < g.nbands ? band_ : g.nbands - 1;
    const int r0 = band * g.BH;
    const TX* xp = x + (long)n * x_bs + (long)ci * g.P + 4 * q;
    const TG* dyp = dy + (long)n * dy_bs + (long)(ci * KPL) * g.P + 4 * q;
    TD* dxp = dx ? dx + (long)n * dx_bs + (long)ci * g.P + 4 * q : nullptr;
    float wt[KPL][9];
#pragma unroll
    for (int j = 0; j < KPL; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[j][k] = w_dw[(ci * KPL + j) * 9 + k];
    const bool aff = in_scale != nullptr;
    const float asc = aff ? in_scale[ci] : 1.f, ash = aff ? in_shift[ci] : 0.f;
    const float rmean = rpart ? bn_mean[ci] : 0.f, rinvstd = rpart ? bn_invstd[ci] : 0.f;
    const DwrLane ln = dwr_lane<PART>(lane, q, g.ncol4);

    float accw[KPL][10];
#pragma unroll
    for (int j = 0; j < KPL; ++j)
#pragma unroll
        for (int k = 0; k < 10; ++k) accw[j][k] = 0.f;
    float r1s = 0.f, r2s = 0.f;
    {
        const int r1 = !active ? r0 : ((r0 + g.BH < g.H) ? r0 + g.BH : g.H);  // band = rows [r0, r1); empty when masked
        float dxa[3][4], xc[3][4];
        float zraw[RP ? 3 : 1][4];
        float d[KPL][6];
        DwrRaw<TG> raw[KPL];
        typename Elem<TX>::raw4 xn;
        // PF (bf16 gradients): the loads of step k + 1 are issued at the start of step k.  The raw rows of a bf16 step are
        // 8 registers, so the second set fits the 128-VGPR budget of four waves per SIMD; with f32 rows it spilled
        // (profiles/r2/dw_bench_rows_prefetch_r2m.txt).
        constexpr bool PF = DWR_BWD_PREFETCH && sizeof(TG) == 2 && sizeof(TX) == 2;
        DwrRaw<TG> nraw[PF ? KPL : 1];
        typename Elem<TX>::raw4 nxn;
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) nraw[j] = dwr_issue(dyp + (long)j * g.P, r0 - 1, g.H, g.W, ln);
            nxn = dwr_issue4(xp, r0, g.H, g.W, ln);
        }
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dxa[s_][c] = xc[s_][c] = 0.f;
                if (RP) zraw[s_][c] = 0.f;  // (a slot that was never opened still enters 0 * (z - mean) * invstd)
            }
        // step k: rho = r0 - 1 + k;  rows rho - 1, rho, rho + 1 live in slots k % 3, (k + 1) % 3, (k + 2) % 3
        constexpr int UN = 3;
        for (int k0 = 0; k0 < g.BH + 2; k0 += UN) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = k0 + u;
                if (k < g.BH + 2) {
                    const int rho = r0 - 1 + k;
                    const int sa = u % 3, sb = (u + 1) % 3, sc_ = (u + 2) % 3;
                    if constexpr (PF) {  // consume the rows issued one step ago, issue the next step's (rows are clamped)
#pragma unroll
                        for (int j = 0; j < KPL; ++j) raw[j] = nraw[j];
                        xn = nxn;
#pragma unroll
                        for (int j = 0; j < KPL; ++j) nraw[j] = dwr_issue(dyp + (long)j * g.P, rho + 1, g.H, g.W, ln);
                        nxn = dwr_issue4(xp, rho + 2, g.H, g.W, ln);
                    } else {  // issue: dY row rho, x row rho + 1
#pragma unroll
                        for (int j = 0; j < KPL; ++j) raw[j] = dwr_issue(dyp + (long)j * g.P, rho, g.H, g.W, ln);
                        xn = dwr_issue4(xp, rho + 1, g.H, g.W, ln);
                    }
#pragma unroll
                    for (int j = 0; j < KPL; ++j) dwr_pin(raw[j]);
                    dwr_pin(xn);
#pragma unroll
                    for (int j = 0; j < KPL; ++j) dwr_finish<PART>(d[j], raw[j], rho, g.H, ln, false, 1.f, 0.f);
                    {  // open the slot of row rho + 1
                        const bool in = (rho + 1) >= r0 && (rho + 1) < r1;
                        const float4 zv = dwr_group<PART>(cvt4(xn), ln);
                        const bool in2 = PART ? (in && !ln.part) : in;  // (columns beyond the plane)
                        xc[sc_][0] = in ? dwr_act(zv.x, aff, asc, ash) : 0.f;
                        xc[sc_][1] = in ? dwr_act(zv.y, aff, asc, ash) : 0.f;
                        xc[sc_][2] = in2 ? dwr_act(zv.z, aff, asc, ash) : 0.f;
                        xc[sc_][3] = in2 ? dwr_act(zv.w, aff, asc, ash) : 0.f;
                        if (RP) {
                            zraw[sc_][0] = zv.x;
                            zraw[sc_][1] = zv.y;
                            zraw[sc_][2] = zv.z;
                            zraw[sc_][3] = zv.w;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) dxa[sc_][c] = 0.f;
                    }
                    const bool inb = rho >= r0 && rho < r1;
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const float(&dv)[6] = d[j];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
#pragma unroll
                            for (int tc = 0; tc < 3; ++tc) {
                                const float e = dv[c + 2 - tc];
                                dxa[sa][c] = fmaf(wt[j][tc], e, dxa[sa][c]);        // row rho - 1: tap row 0
                                dxa[sb][c] = fmaf(wt[j][3 + tc], e, dxa[sb][c]);    // row rho    : tap row 1
                                dxa[sc_][c] = fmaf(wt[j][6 + tc], e, dxa[sc_][c]);  // row rho + 1: tap row 2
                                accw[j][tc] = fmaf(xc[sa][c], e, accw[j][tc]);
                                accw[j][3 + tc] = fmaf(xc[sb][c], e, accw[j][3 + tc]);
                                accw[j][6 + tc] = fmaf(xc[sc_][c], e, accw[j][6 + tc]);
                            }
                        }
                        const float bsum = (dv[1] + dv[2]) + (dv[3] + dv[4]);
                        accw[j][9] += inb ? bsum : 0.f;
                    }
                    // row rho - 1 is complete
                    const int rd = rho - 1;
                    const bool fin = rd >= r0 && rd < r1;
                    if (dxp && fin)
                        dwr_store<PART>(dxp + (long)rd * g.W, make_float4(dxa[sa][0], dxa[sa][1], dxa[sa][2], dxa[sa][3]), ln);
                    if (RP) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            // (the value the BatchNorm backward will read back from dX)
                            const float gg = (fin && xc[sa][c] > 0.f) ? as_stored(dxp, dxa[sa][c]) : 0.f;
                            r1s += gg;
                            r2s = fmaf(gg, (zraw[sa][c] - rmean) * rinvstd, r2s);
                        }
                    }
                }
            }
        }
    }
    // wave sums -> one partial row per (image, wave of the plane).  PACK: one sum per plane = per segment of the wave (a DPP
    // row or a half-wave; masked lanes hold zeros), written by the segment's last lane.
    const long row = (long)n * g.wpp + wip;
    const bool seg16 = PACK && g.seg == 16;
    auto psum = [&](float v) { return !PACK ? wave_sum_l63(v) : (seg16 ? row16_sum(v) : half32_sum_hi(v)); };
    const bool writer = !PACK ? lane == 63 : (lane_on && (lane & (g.seg - 1)) == g.seg - 1);
#pragma unroll
    for (int j = 0; j < KPL; ++j)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = psum(accw[j][k]);
            if (writer) part[(row * Cin * KPL + ci * KPL + j) * 10 + k] = v;
        }
    if (RP) {
        const float v1 = psum(r1s), v2 = psum(r2s);
        const long rows = (long)N * g.wpp;
        if (writer) {
            rpart[row * Cin + ci] = v1;
            rpart[(rows + row) * Cin + ci] = v2;
        }
    }
}

static int dwr_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMAAT_DW_ROWS");
        v = e ? atoi(e) : 1;
    }
    return v;
}

// 1 when the row kernels take this shape (W even: W % 4 == 2 runs with a two-column last group; the caller checks pointer /
// stride alignment)
int dw_rows_ok(int kpl, int H, int W) {
    return dwr_enabled() && (kpl == 1 || kpl == 2 || kpl == 4) && (W & 1) == 0 && W >= 4 && W <= 4096 && H >= 1;
}

// x_dt / y_dt (/ dy_dt / dx_dt): SMAAT_F32 or SMAAT_BF16.  Built combinations: everything f32; bf16 outputs from an f32
// or bf16 input (forward), bf16 dY with x and dX both f32 or both bf16 (backward).  -2 otherwise.
int launch_dw3x3_fwd_rows(const void* x, int x_dt, long x_bs, const float* w_dw, const float* b_dw, void* y, int y_dt,
                          long y_bs, int N, int Cin, int kpl, int H, int W, hipStream_t st, const float* in_scale,
                          const float* in_shift, unsigned* amax) {
    const int nplanes = N * Cin;
    const DwrGeom g = dw_rows_geom(N, Cin, H, W);
    if (g.wpp == 0) return -2;
    const bool pack = g.ppw > 1;
    {
        // the non-walking form on the planes where it was measured faster (profiles/r5/dw_fwd_lin_r5o.txt); SMAAT_DW_LIN=0 / 1:
        // never / wherever it applies (read at every call: A/B and bit-equality tests in one process)
        const char* e = getenv("SMAAT_DW_LIN");
        const int mode = e ? atoi(e) : 2;
        const bool can = (W & 3) == 0 && !pack && (kpl == 1 || kpl == 2) && H >= 2;
        // (default: f32 storage from 72 x 72 up -- step 29.00 -> 28.66 ms; mixed precision at batch 64 measured 0.8 % SLOWER
        // with it, 32.04 -> 32.31 ms, and keeps the walker: profiles/r5/bench_ab_dw_fwd_lin_r5p.txt)
        const bool want = mode == 1 || (mode == 2 && (long)H * W >= DW_LIN_MIN_PLANE && x_dt == SMAAT_F32 && y_dt == SMAAT_F32);
        if (can && want) {
            const int wpp = (H * (W >> 2) + 63) / 64;
            const long nw = (long)nplanes * wpp;
            const dim3 grid((unsigned)((nw + 3) / 4)), blk(256);
#define DWL_GO(K, TX, TY)                                                                                                  \
    hipLaunchKernelGGL((k_dw3x3_fwd_lin<K, TX, TY>), grid, blk, 0, st, (const TX*)x, x_bs, w_dw, b_dw, (TY*)y, y_bs, Cin, nplanes, \
                       H, W, wpp, in_scale, in_shift, amax)
#define DWL_K(TX, TY)                       \
    do {                                    \
        if (kpl == 1) DWL_GO(1, TX, TY);    \
        else DWL_GO(2, TX, TY);             \
    } while (0)
            if (x_dt == SMAAT_F32 && y_dt == SMAAT_F32) DWL_K(float, float);
            else if (x_dt == SMAAT_F32 && y_dt == SMAAT_BF16) DWL_K(float, bf16_t);
            else if (x_dt == SMAAT_BF16 && y_dt == SMAAT_BF16) DWL_K(bf16_t, bf16_t);
            else return -2;
#undef DWL_K
#undef DWL_GO
            return (int)hipGetLastError();
        }
    }
    const long nwaves = pack ? (long)((N + g.ppw - 1) / g.ppw) * Cin : (long)nplanes * g.wpp;
    const dim3 grid((unsigned)((nwaves + 3) / 4)), blk(256);
    const bool part = (W & 3) != 0;
#define DWF_GO1(K, TX, TY, PT, PK)                                                                                          \
    hipLaunchKernelGGL((k_dw3x3_fwd_rows<K, TX, TY, PT, PK>), grid, blk, 0, st, (const TX*)x, x_bs, w_dw, b_dw, (TY*)y, y_bs, \
                       Cin, nplanes, g, in_scale, in_shift, amax)
#define DWF_GO(K, TX, TY)                                   \
    do {                                                    \
        if (pack) {                                         \
            if (part) DWF_GO1(K, TX, TY, true, true);       \
            else DWF_GO1(K, TX, TY, false, true);           \
        } else if (part) DWF_GO1(K, TX, TY, true, false);   \
        else DWF_GO1(K, TX, TY, false, false);              \
    } while (0)
#define DWF_K(TX, TY)                        \
    do {                                     \
        if (kpl == 1) DWF_GO(1, TX, TY);     \
        else if (kpl == 2) DWF_GO(2, TX, TY); \
        else DWF_GO(4, TX, TY);              \
    } while (0)
    if (x_dt == SMAAT_F32 && y_dt == SMAAT_F32) DWF_K(float, float);
    else if (x_dt == SMAAT_F32 && y_dt == SMAAT_BF16) DWF_K(float, bf16_t);
    else if (x_dt == SMAAT_BF16 && y_dt == SMAAT_BF16) DWF_K(bf16_t, bf16_t);
    else return -2;
#undef DWF_K
#undef DWF_GO
#undef DWF_GO1
    return (int)hipGetLastError();
}

int launch_dw3x3_bwd_rows(const void* x, int x_dt, long x_bs, const void* dy, int dy_dt, long dy_bs, const float* w_dw,
                          void* dx, int dx_dt, long dx_bs, float* part, int N, int Cin, int kpl, int H, int W,
                          hipStream_t st, const float* bn_mean, const float* bn_invstd, float* rpart,
                          const float* in_scale, const float* in_shift) {
    const int nplanes = N * Cin;
    const DwrGeom g = dw_rows_geom(N, Cin, H, W);
    if (g.wpp == 0) return -2;
    if (kpl != 1 && kpl != 2) return -2;
    const bool pack = g.ppw > 1;
    const long nwaves = pack ? (long)((N + g.ppw - 1) / g.ppw) * Cin : (long)nplanes * g.wpp;
    const dim3 grid((unsigned)((nwaves + 3) / 4)), blk(256);
    const bool pgrp = (W & 3) != 0;
#define DWR_GO1(K, R, TX, TG, TD, PT, PK)                                                                                      \
    hipLaunchKernelGGL((k_dw3x3_bwd_rows<K, R, TX, TG, TD, PT, PK>), grid, blk, 0, st, (const TX*)x, x_bs, (const TG*)dy, dy_bs, \
                       w_dw, (TD*)dx, dx_bs, part, Cin, nplanes, N, g, bn_mean, bn_invstd, rpart, in_scale, in_shift)
#define DWR_GO(K, R, TX, TG, TD)                                  \
    do {                                                          \
        if (pack) {                                               \
            if (pgrp) DWR_GO1(K, R, TX, TG, TD, true, true);      \
            else DWR_GO1(K, R, TX, TG, TD, false, true);          \
        } else if (pgrp) DWR_GO1(K, R, TX, TG, TD, true, false);  \
        else DWR_GO1(K, R, TX, TG, TD, false, false);             \
    } while (0)
#define DWR_K(TX, TG, TD)                                                  \
    do {                                                                   \
        if (kpl == 1) {                                                    \
            if (rpart) DWR_GO(1, true, TX, TG, TD); else DWR_GO(1, false, TX, TG, TD); \
        } else {                                                           \
            if (rpart) DWR_GO(2, true, TX, TG, TD); else DWR_GO(2, false, TX, TG, TD); \
        }                                                                  \
    } while (0)
    if (x_dt == SMAAT_F32 && dy_dt == SMAAT_F32 && dx_dt == SMAAT_F32) DWR_K(float, float, float);
    else if (x_dt == SMAAT_BF16 && dy_dt == SMAAT_BF16 && dx_dt == SMAAT_BF16) DWR_K(bf16_t, bf16_t, bf16_t);
    else if (x_dt == SMAAT_F32 && dy_dt == SMAAT_BF16 && dx_dt == SMAAT_F32) DWR_K(float, bf16_t, float);
    else return -2;
#undef DWR_K
#undef DWR_GO
#undef DWR_GO1
    return (int)hipGetLastError();
}
