// Fused DepthwiseSeparableConv forward on the bf16 matrix pipe (training path of the plane-dominated layers):
//
//   z[n][m][p] = b_pw[m] + sum_k W_pw[m][k] * y[n][k][p],     y = depthwise3x3(act(x)) (+ b_dw)
//   (reference models/layers.py:47-50; act = the previous BatchNorm + ReLU applied on load, optional)
//
// The depthwise output y NEVER goes through HBM: producer waves stage the input halo tile in LDS (aligned float4
// columns), run the 3x3 stage in VALU on 4-pixel vertical strips, split every value exactly into three bf16 terms and
// write them straight into the B operand image of the split GEMM ([plane][pixel][16 channels]); consumer waves do
// nothing but ds_read_b128 + six v_mfma_f32_32x32x16_bf16 per 16-deep chunk (splitmma.hip explains the split).
// HBM traffic = x (+ halo re-reads served by L2) + z: 4 (Cin + Cout) HW per image instead of 4 (Cin + 2 K + Cout) HW
// for the depthwise kernel + GEMM pair (K = 2 Cin).
//
// Geometry: 64 output channels x 128 pixels per workgroup (2 workgroups per CU), pixel tile = 4 x 32 or 8 x 16
// (template TWL), kernels_per_layer = 2, one barrier per chunk, every LDS buffer double buffered:
//   iteration i:  consumers  M(i)     : A[i&1], B[i&1] -> MFMA
//                 producers  wait for load group i (issued PD = 3 iterations earlier: counted s_waitcnt)
//                            Ac(i+1)  : weight-plane registers -> A[(i+1)&1]
//                            Sc(i+2)  : halo registers -> S[i&1] (act + zero pad)
//                            issue load group i + PD = {A(i+1+PD), S(i+2+PD)}
//                            D(i+1)   : S[(i+1)&1] -> depthwise -> split -> B[(i+1)&1]
// The producers' global loads go through inline asm with explicit counted waits: hipcc's own bookkeeping drains to
// vmcnt(0) at every commit, which leaves ONE group in flight and makes an iteration last one memory latency (~1.5 us
// measured: 31 us for a 16-chunk tile; profiles/r2).
// Task map of the depthwise stage (one task per producer thread and chunk): lane -> (column c = lane & 7 + 8 wave,
// channel cl = (lane >> 3) & 3 + 4 (lane >> 5)): the 32 lanes of a half-wave read 8 columns x 4 channels of the
// staging buffer (channel stride = 8 mod 32 banks) and write 8 pixels x 4 channel pairs of the B image (pixel
// stride 12 words) -- both conflict-free.
#include "common.h"
#include <cstdlib>
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define DSS_BROW 48  // bytes per LDS row of a [row][16 bf16] image (32 + 16 pad: conflict-free ds_read_b128)
// staged floats per input channel (240 used: 6 x 40 or 10 x 24).  The stride decides the LDS banks of the depthwise
// stage's reads: lane map 0 (8 columns x 8 channels per wave) wants = 8 mod 64, lane map 1 (a full tile row per half
// wave, see ROWMAP) wants = 32 (4 x 32 tile) or = 16 (8 x 16 tile) mod 64.
#define DSS_CSTRIDE_OF(TWL, ROWMAP) ((ROWMAP) ? ((TWL) == 5 ? 288 : 272) : 264)

__device__ __forceinline__ unsigned dss_fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float dss_bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ unsigned dss_pack_hi16(float lo, float hi) {
    return __builtin_amdgcn_perm(dss_fbits(hi), dss_fbits(lo), 0x07060302u);  // {hi.hi16, lo.hi16}
}

struct DsSplitArgs {
    const float* x;
    long x_bs;
    const float* in_scale;
    const float* in_shift;
    const float* w_dw;             // [K][9]
    const float* b_dw;             // [K] or null
    const unsigned short* planes;  // pointwise weight, split planes, chunk-major [Kp/16][3][M][16]
    const float* bias;             // [M] or null
    float* out;
    long out_bs;
    float* part;   // [3][T][M] or null
    float* y_out;  // [N][K][P] or null (depthwise output as a side product)
    int N, Cin, Kdim, M, nco, H, W, P, tiles_x, tiles_per_img, T;
    float out_floor;  // epilogue: out = max(acc + bias, out_floor); -inf = plain, 0 = fused ReLU
};

// TWL: log2 of the tile width (5: 4 x 32 tile, 4: 8 x 16 tile).  NT: 3 = exact split, 1 = plain bf16 operands.
typedef float dss_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dss_u32x4 __attribute__((ext_vector_type(4)));

// ROWMAP: the depthwise stage's lane -> (column, channel) map.  0: a wave covers 8 columns x 8 channels (bank-conflict
// free on both LDS sides).  1: a half wave covers one full tile row of one channel, so that the side output y_out is
// written as whole 128-B (4 x 32 tile) / 64-B (8 x 16 tile) runs; the bf16 image writes then conflict 2-way.
template <int TWL, int NT, bool AFF, bool YOUT, bool ROWMAP>
// (bf16-operand builds with the activation on load: at 128 VGPRs hipcc spills PREFETCHED registers -- scratch stores of load
// destinations still in flight, scripts/isa_hazards.py; they get the 256-register budget: two workgroups per CU instead of four)
__global__ __launch_bounds__(512, (NT == 1 && AFF) ? 2 : 4) void k_dsconv_split(const DsSplitArgs a) {
    constexpr int DSS_CSTRIDE = DSS_CSTRIDE_OF(TWL, ROWMAP);
    constexpr int KPL = 2, KC = 16, KCI = KC / KPL;
    constexpr int CT = 2, WPX = 4, PXT = 1;
    constexpr int COT = 64, PT = 128, NPT = 256;
    constexpr int TW = 1 << TWL, TH = PT / TW;
    constexpr int STRIDE = TW + 8, NROW = TH + 2, NCOL4 = (TW + 8) / 4, PER = NROW * NCOL4;  // PER = 60 in both shapes
    static_assert(PER == 60 && NROW * STRIDE <= DSS_CSTRIDE, "staging geometry");
    constexpr int F = KCI * PER;                  // float4 staging slots per chunk (480)
    constexpr int NSL = (F + NPT - 1) / NPT;      // per producer thread (2)
    constexpr int APL = COT * DSS_BROW, BPL = PT * DSS_BROW;
    constexpr int BUFSZ = NT * (APL + BPL);       // bytes: NT planes of A then NT planes of B
    constexpr int NAP = COT * 2 * NT;             // 16-byte pieces of the A planes per chunk
    constexpr int NAT = (NAP + NPT - 1) / NPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* S = (float*)(lds + 2 * BUFSZ);         // [2][KCI][DSS_CSTRIDE]
    float* DWl = S + 2 * KCI * DSS_CSTRIDE;       // [2][256]: per chunk [KC][12] = 9 taps, bias, 2 pad
    float* stat = DWl + 2 * 256;                  // BN_STAT_FLOATS(WPX, COT)
    int* pixoff = (int*)(stat + BN_STAT_FLOATS(WPX, COT));  // [PT]
    float* biasl = (float*)(pixoff + PT);         // [COT]
    float* affl = biasl + COT;                    // AFF: [2][Cin] scale / shift of the activation applied on load

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int wpx = wv & 3;       // consumer wave -> 32 pixels of the tile
    const int pw = wv & 3;        // producer wave index
    const int l31 = lane & 31, half = lane >> 5;
    const int ptid = producer ? tid - 256 : 0;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int cot = idx % a.nco;
    const int ptg = xcd * ((a.T + 7) >> 3) + idx / a.nco;  // contiguous tile range per XCD (halo lines meet in one L2)
    if (ptg >= a.T) return;
    const int n = ptg / a.tiles_per_img, tl = ptg - n * a.tiles_per_img;
    const int ty = tl / a.tiles_x, tx = tl - ty * a.tiles_x;
    const int r0 = ty * TH, c0 = tx * TW;
    const int co0 = cot * COT;
    const int nchunks = (a.Kdim + KC - 1) / KC;
    constexpr int PD = 3;                                   // load groups in flight per producer thread
    const int nch_pad = (nchunks + PD - 1) / PD * PD;       // barriers of the chunk loop (producers and consumers alike)
    const float* xn = a.x + (long)n * a.x_bs;
    if (tid < COT) {
        const int m = co0 + tid;
        const float* bp = a.bias ? a.bias : (const float*)a.planes;
        const float v = bp[m < a.M ? m : 0];
        biasl[tid] = (a.bias && m < a.M) ? v : 0.f;
    }
    for (int i = tid; i < PT; i += 512) {
        const int r = r0 + (i >> TWL), c = c0 + (i & (TW - 1));
        pixoff[i] = (r < a.H && c < a.W) ? r * a.W + c : -1;
    }
    if (AFF) {
        for (int i = tid; i < a.Cin; i += 512) {
            affl[i] = a.in_scale[i];
            affl[a.Cin + i] = a.in_shift[i];
        }
        __syncthreads();
    }

    if (producer) {
        // ---- staging slots: (channel of the chunk, halo row, float4 column) ----
        int s_cl[NSL], s_in[NSL], s_lo[NSL];
        bool s_ok[NSL];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int f = (ptid + NPT * j) % F;  // surplus slots of the last group re-stage a valid element
            const int cl = f / PER, rem = f - cl * PER;
            const int rr = rem / NCOL4, q = rem - rr * NCOL4;
            const int gr = r0 - 1 + rr, gc = c0 - 4 + 4 * q;
            const bool ok = gr >= 0 && gr < a.H && gc >= 0 && gc < a.W;  // W % 4 == 0: a float4 is inside or outside
            s_cl[j] = cl;
            s_in[j] = ok ? gr * a.W + gc : 0;
            s_lo[j] = cl * DSS_CSTRIDE + rr * STRIDE + 4 * q;
            s_ok[j] = ok;
        }
        // ---- depthwise task of this thread: channel t_cl of the chunk, strip (t_rg, t_c) ----
        int t_cl, t_c, t_rg;
        if (ROWMAP) {
            t_cl = half + 2 * pw;
            t_c = lane & (TW - 1);
            t_rg = TWL == 5 ? 0 : (lane >> 4) & 1;
        } else {
            t_cl = ((lane >> 3) & 3) + 4 * half;
            if (TWL == 5) {
                t_c = (lane & 7) + 8 * pw;
                t_rg = 0;
            } else {
                t_c = (lane & 7) + 8 * (pw & 1);
                t_rg = pw >> 1;
            }
        }
        const int t_sb = t_cl * DSS_CSTRIDE + (t_rg * 4) * STRIDE + t_c + 3;   // S index of (row - 1, col - 1)
        const int t_px = (t_rg * 4) * TW + t_c;                                // tile pixel of the strip's first row
        const int t_gr = r0 + t_rg * 4;
        const int t_go = t_gr * a.W + c0 + t_c;
        const bool t_cok = (c0 + t_c) < a.W;
        const unsigned t_yo = (unsigned)(t_cl * 2 * a.P + t_go) * 4u;
        // ---- A-plane pieces ----
        int a_src[NAT], a_dst[NAT];
#pragma unroll
        for (int u = 0; u < NAT; ++u) {
            const int id = (ptid + NPT * u) % NAP;
            const int pl = id / (COT * 2), rem = id - pl * (COT * 2);
            const int row = rem >> 1, h = rem & 1;
            const int m = co0 + row;
            a_src[u] = (pl * a.M + (m < a.M ? m : a.M - 1)) * 16 + h * 8;  // + chunk * 3 * M * 16 (ushort units)
            a_dst[u] = pl * APL + row * DSS_BROW + h * 16;
        }
        const int dwk = ptid / 12, dwt = ptid - dwk * 12;
        const float* dwsrc = (dwt < 9) ? a.w_dw : (a.b_dw ? a.b_dw : a.w_dw);
        const int dwmul = (dwt < 9) ? 9 : 1, dwadd = (dwt < 9) ? dwt : 0;
        const bool dwvalid = (dwk < KC) && (dwt < 9 || (dwt == 9 && a.b_dw != nullptr));
        // ---- load groups: PD register sets in flight, inline-asm loads, counted waits ----
        constexpr int LPC = NSL + 1 + NAT;                       // loads per group
        constexpr int SPC = YOUT ? 8 : 0;                        // side-output stores per iteration (also count in vmcnt)
        constexpr int WAITN = SPC + (PD - 1) * (LPC + SPC);      // younger operations when a group is consumed
        static_assert(WAITN <= 63, "vmcnt is a 6-bit counter");
        dss_f32x4 sx[PD][NSL];
        float sdw[PD];
        dss_u32x4 sa[PD][NAT];
        auto clampc = [&](int ch) { return ch < nchunks ? ch : nchunks - 1; };
        // group g = {weight planes of chunk g + 1, halo + taps of chunk g + 2}
        auto issue = [&](int g, int set) __attribute__((always_inline)) {
            const int cx = clampc(g + 2), ca = clampc(g + 1);
            const int ci0 = cx * KCI;
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const int ci = ci0 + s_cl[j];
                const float* src = xn + (long)(ci < a.Cin ? ci : a.Cin - 1) * a.P + s_in[j];
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sx[set][j]) : "v"(src));
            }
            {
                const int kg = cx * KC + (dwk < KC ? dwk : 0);
                const float* src = dwsrc + (kg < a.Kdim ? kg : a.Kdim - 1) * dwmul + dwadd;
                asm volatile("global_load_dword %0, %1, off" : "=v"(sdw[set]) : "v"(src));
            }
            const unsigned short* ab = a.planes + (long)ca * 3 * a.M * 16;
#pragma unroll
            for (int u = 0; u < NAT; ++u) {
                const unsigned short* src = ab + a_src[u];
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sa[set][u]) : "v"(src));
            }
        };
        auto wait_set = [&](int set) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
#pragma unroll
            for (int j = 0; j < NSL; ++j) asm volatile("" : "+v"(sx[set][j]));  // uses stay behind the wait
            asm volatile("" : "+v"(sdw[set]));
#pragma unroll
            for (int u = 0; u < NAT; ++u) asm volatile("" : "+v"(sa[set][u]));
        };
        auto store_b = [&](int ch_, int buf, const dss_f32x4 (&xr)[NSL], float dwv) __attribute__((always_inline)) {
            const int ci0 = clampc(ch_) * KCI;
            float* Sb = S + buf * (KCI * DSS_CSTRIDE);
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const int ci = ci0 + s_cl[j];
                const bool ok = s_ok[j] && ci < a.Cin;
                float4 v = make_float4(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);
                if (AFF) {  // the previous BatchNorm + ReLU applied on load; the zero padding stays zero
                    const int cic = ci < a.Cin ? ci : a.Cin - 1;
                    const float sc = affl[cic], sh = affl[a.Cin + cic];
                    v.x = fmaxf(fmaf(v.x, sc, sh), 0.f);
                    v.y = fmaxf(fmaf(v.y, sc, sh), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc, sh), 0.f);
                    v.w = fmaxf(fmaf(v.w, sc, sh), 0.f);
                }
                v.x = ok ? v.x : 0.f;
                v.y = ok ? v.y : 0.f;
                v.z = ok ? v.z : 0.f;
                v.w = ok ? v.w : 0.f;
                *(float4*)(Sb + s_lo[j]) = v;
            }
            DWl[buf * 256 + ptid] = (dwvalid && (clampc(ch_) * KC + dwk) < a.Kdim) ? dwv : 0.f;
        };
        auto store_a = [&](int buf, const dss_u32x4 (&ar)[NAT]) __attribute__((always_inline)) {
            unsigned char* base = lds + buf * BUFSZ;
#pragma unroll
            for (int u = 0; u < NAT; ++u) *(dss_u32x4*)(base + a_dst[u]) = ar[u];
        };
        // side output through a buffer descriptor: always issued (fixed operation count), dropped by the range check
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            YOUT ? a.y_out + (long)n * a.Kdim * a.P : (float*)nullptr, 0, YOUT ? a.Kdim * a.P * 4 : 0, 0x00020000);
        auto dwstage = [&](int ch_, int buf) {
            const int k0 = ch_ * KC;
            const float* sp = S + buf * (KCI * DSS_CSTRIDE) + t_sb;
            const float4* DWb = (const float4*)(DWl + buf * 256);
            unsigned char* Bb = lds + buf * BUFSZ + NT * APL + t_px * DSS_BROW + t_cl * 4;
            float v[6][3];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) v[rr][dc] = sp[rr * STRIDE + dc];
            float y[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = t_cl * 2 + j;
                const float4 wa = DWb[k * 3], wb = DWb[k * 3 + 1], wc = DWb[k * 3 + 2];  // zeros when k0 + k >= Kdim
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float acc = wc.y;
                    acc = fmaf(wa.x, v[i][0], acc);
                    acc = fmaf(wa.y, v[i][1], acc);
                    acc = fmaf(wa.z, v[i][2], acc);
                    acc = fmaf(wa.w, v[i + 1][0], acc);
                    acc = fmaf(wb.x, v[i + 1][1], acc);
                    acc = fmaf(wb.y, v[i + 1][2], acc);
                    acc = fmaf(wb.z, v[i + 2][0], acc);
                    acc = fmaf(wb.w, v[i + 2][1], acc);
                    acc = fmaf(wc.x, v[i + 2][2], acc);
                    y[j][i] = acc;
                }
            }
            if (YOUT) {
                // byte offset = per-thread part (channel pair, strip origin) + per-chunk / per-store scalar part
                const bool pok = cot == 0 && t_cok && (k0 + t_cl * 2) < a.Kdim;  // Kdim is even: a pair is in or out
                unsigned vo = pok ? t_yo : 0x80000000u;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned voff = (t_gr + i) < a.H ? vo : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[j][i]), yrs, voff,
                                                              ((k0 + j) * a.P + i * a.W) * 4, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float p1[2], p2[2], p3[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float x = y[j][i];
                    if (NT == 1) {
                        const unsigned bb = dss_fbits(x);
                        p1[j] = dss_bitsf((bb + 0x7FFFu + ((bb >> 16) & 1u)) & 0xFFFF0000u);  // round to nearest even
                        p2[j] = p3[j] = 0.f;
                    } else {
                        p1[j] = dss_bitsf(dss_fbits(x) & 0xFFFF0000u);
                        const float r1 = x - p1[j];  // exact
                        p2[j] = dss_bitsf(dss_fbits(r1) & 0xFFFF0000u);
                        p3[j] = r1 - p2[j];          // exact, <= 8 significant bits
                    }
                }
                unsigned char* dst = Bb + i * (TW * DSS_BROW);
                *(unsigned*)(dst) = dss_pack_hi16(p1[0], p1[1]);
                if (NT == 3) {
                    *(unsigned*)(dst + BPL) = dss_pack_hi16(p2[0], p2[1]);
                    *(unsigned*)(dst + 2 * BPL) = dss_pack_hi16(p3[0], p3[1]);
                }
            }
        };

        // The PD groups that the loop consumes first are issued before the (compiler-tracked) prologue loads: the
        // prologue's wait covers all of them, i.e. one memory latency for the whole start-up.
#pragma unroll
        for (int g = 0; g < PD; ++g) issue(g, g);
        {
            dss_f32x4 x0[NSL], x1[NSL];
            dss_u32x4 a0[NAT];
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const int c0_ = s_cl[j], c1_ = clampc(1) * KCI + s_cl[j];
                x0[j] = *(const dss_f32x4*)(xn + (long)(c0_ < a.Cin ? c0_ : a.Cin - 1) * a.P + s_in[j]);
                x1[j] = *(const dss_f32x4*)(xn + (long)(c1_ < a.Cin ? c1_ : a.Cin - 1) * a.P + s_in[j]);
            }
            const int kg0 = dwk < KC ? dwk : 0, kg1 = clampc(1) * KC + kg0;
            const float d0 = dwsrc[(kg0 < a.Kdim ? kg0 : a.Kdim - 1) * dwmul + dwadd];
            const float d1 = dwsrc[(kg1 < a.Kdim ? kg1 : a.Kdim - 1) * dwmul + dwadd];
#pragma unroll
            for (int u = 0; u < NAT; ++u) a0[u] = *(const dss_u32x4*)(a.planes + a_src[u]);
            store_b(0, 0, x0, d0);
            store_a(0, a0);
            __syncthreads();
            dwstage(0, 0);
            store_b(1, 1, x1, d1);
            __syncthreads();
        }
        // The chunk loop runs over nchunks rounded up to a multiple of PD with every slot of the unrolled body unconditional:
        // a surplus iteration waits for its group, stages clamped (valid) data into buffers nobody reads any more and issues its
        // side-output stores out of range (dropped by the descriptor) -- the operation counts the hand-counted waits rely on
        // hold on EVERY path of the control-flow graph, which scripts/isa_hazards.py proves on the generated ISA (round 6).
        for (int i0 = 0; i0 < nch_pad; i0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int i = i0 + u;
                const int nb = (i + 1) & 1;
                wait_set(u);                       // group i: its loads were issued PD iterations ago
                store_a(nb, sa[u]);                // A(i + 1)
                store_b(i + 2, i & 1, sx[u], sdw[u]);  // S(i + 2)
                issue(i + PD, u);
                dwstage(i + 1, nb);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);  // slots stay apart (interleaved, the unconditional body spilled 6 registers)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus groups of the tail still target live registers
    } else {
        f32x16 acc[CT][PXT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][0][r] = 0.f;
        const int aoff = l31 * DSS_BROW + half * 16;
        const int boff = NT * APL + (wpx * 32 + l31) * DSS_BROW + half * 16;
        __syncthreads();
        __syncthreads();
        for (int i = 0; i < nchunks; ++i) {
            const unsigned char* base = lds + (i & 1) * BUFSZ;
            bf16x8 af[CT][NT], bf[NT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < NT; ++t) af[ct][t] = *(const bf16x8*)(base + aoff + t * APL + ct * 32 * DSS_BROW);
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = *(const bf16x8*)(base + boff + t * BPL);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (NT == 3) {  // smallest terms first
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ct][0], bf[NT - 1], acc[ct][0], 0, 0, 0);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ct][NT - 1], bf[0], acc[ct][0], 0, 0, 0);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ct][NT / 2], bf[NT / 2], acc[ct][0], 0, 0, 0);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ct][0], bf[NT / 2], acc[ct][0], 0, 0, 0);
                    acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ct][NT / 2], bf[0], acc[ct][0], 0, 0, 0);
                }
                acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ct][0], bf[0], acc[ct][0], 0, 0, 0);
            }
            __syncthreads();
        }
        for (int i = nchunks; i < nch_pad; ++i) __syncthreads();  // the producers' surplus iterations
        // ---- epilogue: bias + row stores (a wave's 32 pixels are one or two full tile rows), BatchNorm partials ----
        const int off = pixoff[wpx * 32 + l31];
        float* obase = a.out + (long)n * a.out_bs;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int m = co0 + col;
                if (m < a.M && off >= 0) obase[(long)m * a.P + off] = fmaxf(acc[ct][0][r] + biasl[col], a.out_floor);
            }
        }
        if (a.part) {
            bool pval[PXT];
            pval[0] = off >= 0;
            int fl;
            const int nw = bn_wave_count<PXT>(pval, fl);
            bn_wave_partials<CT, PXT>(acc, pval, l31, half, stat + wpx * 3 * COT, COT, fl);
            if (lane == 0) ((int*)(stat + WPX * 3 * COT))[wpx] = nw;
        }
    }
    if (a.part) {
        __syncthreads();
        for (int col = tid; col < COT; col += 512) {
            float mean, m2, cnt;
            bn_tile_combine<WPX>(stat, (const int*)(stat + WPX * 3 * COT), COT, col, mean, m2, cnt);
            const int m = co0 + col;
            if (m < a.M) {
                a.part[((long)0 * a.T + ptg) * a.M + m] = mean;
                a.part[((long)1 * a.T + ptg) * a.M + m] = m2;
                a.part[((long)2 * a.T + ptg) * a.M + m] = cnt;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
int split_mode();  // splitmma.hip

static int dss_twl(int H, int W) {
    if ((W & 31) == 0 && H >= 4) return 5;
    if ((W & 15) == 0 && H >= 8) return 4;
    return 0;
}

// tiles of the fused kernel for an N x H x W problem; 0 when the shape is not handled
int dsconv_split_num_slots(int N, int H, int W) {
    const int twl = dss_twl(H, W);
    if (!twl) return 0;
    const int TW = 1 << twl, TH = 128 / TW;
    return N * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
}

template <int TWL, int NT, bool AFF, bool YOUT, bool ROWMAP>
static int launch_dss_cfg(DsSplitArgs& a, hipStream_t st) {
    constexpr int COT = 64, PT = 128, KCI = 8, DSS_CSTRIDE = DSS_CSTRIDE_OF(TWL, ROWMAP);
    const size_t lds = (size_t)2 * NT * (COT + PT) * DSS_BROW +
                       sizeof(float) * (size_t)(2 * KCI * DSS_CSTRIDE + 2 * 256 + BN_STAT_FLOATS(4, COT) + PT + COT +
                                                (AFF ? 2 * a.Cin : 0));
    constexpr auto kern = k_dsconv_split<TWL, NT, AFF, YOUT, ROWMAP>;
    static size_t granted = 0;
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    const int grid = ((a.T + 7) / 8) * 8 * a.nco;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    return (int)hipGetLastError();
}

static int dss_rowmap_noy() {  // experiment switch: lane map 1 also without the side output
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMAAT_DSS_ROWMAP");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v;
}

template <int TWL, int NT>
static int launch_dss_sel(DsSplitArgs& a, hipStream_t st) {
    const bool aff = a.in_scale != nullptr, yo = a.y_out != nullptr;
    if (yo)
        return aff ? launch_dss_cfg<TWL, NT, true, true, true>(a, st) : launch_dss_cfg<TWL, NT, false, true, true>(a, st);
    if (dss_rowmap_noy())
        return aff ? launch_dss_cfg<TWL, NT, true, false, true>(a, st) : launch_dss_cfg<TWL, NT, false, false, true>(a, st);
    return aff ? launch_dss_cfg<TWL, NT, true, false, false>(a, st) : launch_dss_cfg<TWL, NT, false, false, false>(a, st);
}

// returns -2 when the shape / alignment is not handled (kernels_per_layer != 2, W % 16 != 0, unaligned planes)
int launch_dsconv_split(DsSplitArgs& a, int kpl, hipStream_t st) {
    const int twl = dss_twl(a.H, a.W);
    if (kpl != 2 || !twl || (a.x_bs & 3) || (((uintptr_t)a.x) & 15) || a.Kdim != 2 * a.Cin) return -2;
    a.P = a.H * a.W;
    if ((long)a.M * a.P >= (1L << 31) || (long)a.Kdim * a.P >= (1L << 31)) return -2;
    const int TW = 1 << twl, TH = 128 / TW;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_per_img = a.tiles_x * ((a.H + TH - 1) / TH);
    a.T = a.N * a.tiles_per_img;
    a.nco = (a.M + 63) / 64;
    if (a.in_scale != nullptr && a.Cin > 1024) return -2;  // (the activation table lives in LDS)
    const int nt = split_mode() == 1 ? 1 : 3;
    if (twl == 5) return nt == 1 ? launch_dss_sel<5, 1>(a, st) : launch_dss_sel<5, 3>(a, st);
    return nt == 1 ? launch_dss_sel<4, 1>(a, st) : launch_dss_sel<4, 3>(a, st);
}
