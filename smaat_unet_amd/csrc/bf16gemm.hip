// Mixed-precision matrix path (BASELINE configs[3]): bf16 activations in HBM, bf16 weight images, ONE
// v_mfma_f32_32x32x16_bf16 per product, f32 accumulation, f32 BatchNorm partials, bf16 (or f32) output.
//
//   k_bf16_planes   f32 weight matrix -> bf16 chunk-major image [Cp/16][R][16]          (once per weight per step)
//   k_pw_bf16       out[n][m][p] = sum_c A[m][c] * x[n][c][p] + bias[m]   (+ BatchNorm partials)
//                   pointwise conv of the forward pass (reference models/layers.py:45,49) and its data gradient
//   k_wgrad_bf16    dW[m][k] = sum_{n,p} dz[n][m][p] * y[n][k][p]         (pointwise weight gradient)
//
// In bf16 every layer of the network is HBM-bound (SURVEY 8(d) ridge analysis), so these kernels are built around the
// memory pipeline, not the matrix pipe:
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no VALU, no ds_write; three or
//     four LDS stages, two or three chunks of loads in flight per workgroup behind the one being consumed, counted s_waitcnt vmcnt and
//     raw s_barrier (a __syncthreads() or a compiler-tracked LDS read would drain the DMA queue to vmcnt(0));
//   * the activation tile is stored in LDS exactly as it lies in memory ([channel][pixel], pixels contiguous); the MFMA B
//     operand needs 8 consecutive CHANNELS of one pixel per lane: ds_read_b64_tr_b16 (the gfx950 transpose read,
//     mapping verified by scripts/probes/lds_tr_probe.hip) delivers exactly that from the row-major image.  LDS-DMA
//     writes lane-linearly, so the bank swizzle (16-byte chunk index ^ 4 * (row & 3)) is applied to the SOURCE address;
//   * the weight image is our own layout: [row][16 k] 32-byte rows, k halves swapped on odd 8-row groups so that the
//     four 16-lane groups of a ds_read_b128 each cover all 64 banks;
//   * the weight gradient contracts over pixels, which are contiguous for both operands: no transpose at all, fragments
//     are plain ds_read_b128 of the DMA image (chunk swizzle c ^ ((row >> 2) & 3)).
// All LDS reads are inline asm (hipcc would otherwise wait vmcnt(0) before any LDS read that may alias a DMA write).
#include <utility>

#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// zero source for the out-of-plane lanes of the weight gradient's loads
__device__ __attribute__((aligned(64))) unsigned g_bf16_zero[16] = {0};

template <int OFF>
__device__ __forceinline__ bf16x8 lds_rd128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_rd_tr(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ float2 lds_rd64(unsigned addr) {
    float2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int GW>
__device__ __forceinline__ void glds(const void* src, void* lds_dst) {
    static_assert(GW == 16 || GW == 4, "LDS-DMA width");
    if constexpr (GW == 16)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)lds_dst, 4, 0, 0);
}

// =====================================================================================
// weights: f32 [R][C] (or, src_t, stored [C][R]) -> bf16 chunk-major [Cp/16][R][16], Cp = C rounded up to 32, zero padded
// =====================================================================================
__global__ __launch_bounds__(256) void k_bf16_planes(const float* __restrict__ w, int R, int C, int Cp,
                                                     bf16_t* __restrict__ out, int src_t) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)R * Cp) return;
    // consecutive threads walk the contiguous dimension of the source
    const int r = src_t ? (int)(i % R) : (int)(i / Cp), c = src_t ? (int)(i / R) : (int)(i - (long)r * Cp);
    const float x = c < C ? (src_t ? w[(long)c * R + r] : w[(long)r * C + c]) : 0.f;
    out[((long)(c >> 4) * R + r) * 16 + (c & 15)] = (bf16_t)(pack_bf16x2(x, 0.f) & 0xFFFFu);
}

int launch_bf16_planes(const float* w, int R, int C, bf16_t* out, int src_t, hipStream_t st) {
    const int Cp = (C + 31) & ~31;
    const long n = (long)R * Cp;
    hipLaunchKernelGGL(k_bf16_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, R, C, Cp, out, src_t);
    return (int)hipGetLastError();
}

// =====================================================================================
// k_pw_bf16
// =====================================================================================
#define BF_KC 32  // channels per stage
// LDS stages: NST - 1 chunks of loads are in flight behind the one being consumed.  Four where three workgroups of four
// stages fit a CU's 160 KB (the 64-channel tiles: the 288^2 layers, which are pure streaming), three otherwise.
template <int COT>
struct BfStages {
    static constexpr int value = COT <= 64 ? 4 : 3;
};

// GW: bytes per lane of an activation LDS-DMA (16: P % 8 == 0; 4: P % 2 == 0 -- the 18 x 18 planes)
//
// PERSISTENT over (pixel tile, channel tile) items: a workgroup walks the items idx0, idx0 + gstep, ... of its XCD and
// issues its LDS-DMA as ONE flattened chunk stream across item boundaries -- the first two chunks of the next tile are in
// flight while the store epilogue of the current tile runs.  The short-contraction GEMMs (data gradients of the 288^2 /
// 144^2 layers: two to four chunks per tile, then 32-64 KB of output) would otherwise serialise load latency, MFMAs and
// stores per workgroup.  Everything between the first DMA and the last store touches LDS through inline asm only (bias
// slots included): a compiler-visible LDS access would be preceded by s_waitcnt vmcnt(0) and drain the prefetch.  The
// BatchNorm partials (forward GEMMs only) use the shared helpers and accept that drain once per tile, after the stores.
template <int WCO, int CT, int WPX, int PXT, int GW, typename TO>
__global__ __launch_bounds__(WCO * WPX * 64, 3) void k_pw_bf16(const PwBfArgs a) {
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32, NW = WCO * WPX, NTH = NW * 64;
    static_assert(PT == 128 && NW == 4, "128-pixel tiles (256-byte LDS rows), four waves");
    constexpr int XB = BF_KC * PT * 2;          // X stage: [KC][PT] bf16
    constexpr int AB = (BF_KC / 16) * COT * 32;  // A stage: [KC/16][COT][16] bf16
    constexpr int STG = XB + AB;
    constexpr int BF_NST = BfStages<COT>::value;
    constexpr int NXP = GW == 16 ? XB / 1024 : BF_KC;  // X pieces (one wave-instruction each) per stage
    constexpr int NAP = AB / 1024;
    static_assert(NXP % NW == 0 && NAP % NW == 0, "pieces divide evenly among the waves");
    constexpr int XPW = NXP / NW, APW = NAP / NW, PPW = XPW + APW;
    static_assert((BF_NST - 1) * PPW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;  // asm reads take LDS byte addresses
    float* stat = (float*)(lds + BF_NST * STG);            // [WPX][3][COT] + [8]
    constexpr int BIAS_OFF = BF_NST * STG + 4 * BN_STAT_FLOATS(WPX, COT);  // [2][COT] floats: bias of the item, by item parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wv % WCO, wpx = wv / WCO;
    const int l31 = lane & 31, half = lane >> 5;

    // items of this workgroup: idx = idx0 + k * gstep within the XCD's range; idx -> (pixel tile idx / nco, channel tile idx % nco)
    const int b = blockIdx.x, xcd = b & 7, idx0 = b >> 3;
    const int gstep = gridDim.x >> 3;
    const int tpx = (a.T + 7) >> 3;  // contiguous tile range per XCD (all channel tiles of a pixel tile share an L2)
    int lim_t = a.T - xcd * tpx;
    lim_t = lim_t < tpx ? lim_t : tpx;
    const int lim = lim_t > 0 ? lim_t * a.nco : 0;
    if (idx0 >= lim) return;
    const int nitems = (lim - 1 - idx0) / gstep + 1;
    const int nchunks = a.Cp / BF_KC;
    const unsigned rowbytes = (unsigned)a.P * 2u;

    // ---- LDS-DMA cursor: (item, chunk) + the per-item source addresses, fixed per lane within an item ---------------
    int pf_item = 0, pf_ch = 0;
    const unsigned char* pf_x = nullptr;  // image base
    unsigned pf_xcol = 0;                 // byte offset of this lane's pixels within a row
    int pf_co0 = 0;
    const int xr = lane >> 4;                                    // GW 16: row within a 4-row piece
    const int xc16 = (lane & 15) ^ (4 * (xr & 3));               // GW 16: source chunk of LDS chunk (lane & 15)
    const int xc4 = (lane >> 2) ^ (4 * (wv & 3));                // GW 4: piece q = wv + NW * u is row q; q & 3 == wv & 3
    const unsigned xrow0 = GW == 16 ? (unsigned)(4 * wv + xr) : (unsigned)wv;
    const int arow = lane >> 1, ah = (lane & 1) ^ ((lane >> 4) & 1);
    // Per-lane byte offsets of the pieces, fixed within an item: the per-chunk address is a wave-uniform 64-bit base (scalar
    // arithmetic) + this 32-bit offset -- the saddr form of global_load_lds, no vector address arithmetic per chunk.  (The
    // first version recomputed clamp, 32-bit multiply and 64-bit add per piece and chunk: 26 VALU + a v_mad_i64 pair per
    // 8 MFMAs; the counters showed 16 VALU instructions per MFMA for this kernel.)
    unsigned pf_xo[XPW], pf_ao[APW];
    auto pf_setup = [&]() __attribute__((always_inline)) {
        const int it = pf_item < nitems ? pf_item : nitems - 1;  // surplus issues re-load the last item into a dead stage
        const int idx = idx0 + it * gstep;
        const int j = idx / a.nco, cot = idx - j * a.nco;
        const int ptg = xcd * tpx + j;
        const int n = ptg / a.tiles_per_img, tl = ptg - n * a.tiles_per_img;
        pf_x = (const unsigned char*)(a.x + (long)n * a.x_bs);
        const int px = tl * PT + (GW == 16 ? 8 * xc16 : 8 * xc4 + 2 * (lane & 3));
        pf_xcol = (unsigned)(px < a.P ? px : 0) * 2u;
        pf_co0 = cot * COT;
#pragma unroll
        for (int u = 0; u < XPW; ++u) {
            const unsigned row0 = GW == 16 ? xrow0 + 16u * u : (unsigned)(wv + NW * u);
            pf_xo[u] = row0 * rowbytes + pf_xcol;
        }
#pragma unroll
        for (int u = 0; u < APW; ++u) {
            const int q = wv + NW * u;
            const int jj = q / (COT / 32), rb = q - jj * (COT / 32);
            int m = pf_co0 + rb * 32 + arow;
            m = m < a.M ? m : a.M - 1;
            pf_ao[u] = (unsigned)((jj * a.M + m) * 16 + ah * 8) * 2u;
        }
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
        const int k0 = pf_ch * BF_KC;
        unsigned char* sb = lds + stage * STG;
        if (k0 + BF_KC <= a.Cin) {  // wave-uniform: every channel row of the chunk exists
            const unsigned char* xs = pf_x + (long)k0 * rowbytes;
#pragma unroll
            for (int u = 0; u < XPW; ++u) glds<GW>(xs + pf_xo[u], sb + (wv + NW * u) * (GW == 16 ? 1024 : 256));
        } else {  // last chunk of a contraction that is not a multiple of 32 (the 24-channel stem): clamp the row per lane
#pragma unroll
            for (int u = 0; u < XPW; ++u) {
                const int q = wv + NW * u;
                int row = k0 + (GW == 16 ? (int)xrow0 + 16 * u : q);
                row = row < a.Cin ? row : a.Cin - 1;  // (the weight image is zero there)
                glds<GW>(pf_x + (unsigned)row * rowbytes + pf_xcol, sb + q * (GW == 16 ? 1024 : 256));
            }
        }
        const unsigned char* as = (const unsigned char*)a.planes + (long)(k0 >> 4) * a.M * 32;
#pragma unroll
        for (int u = 0; u < APW; ++u) {
            const int q = wv + NW * u;
            const int j = q / (COT / 32), rb = q - j * (COT / 32);
            glds<16>(as + pf_ao[u], sb + XB + (j * COT + rb * 32) * 32);
        }
        if (++pf_ch == nchunks) {
            pf_ch = 0;
            ++pf_item;
            pf_setup();
        }
    };

    // ---- fragment addresses -------------------------------------------------------------------------------------------
    // A: row m = (wco * CT + ct) * 32 + l31 of sub-chunk j at ((j * COT + m) * 2 + (half ^ ((m >> 3) & 1))) * 16
    const unsigned a_addr = lds0 + (unsigned)(XB + ((wco * CT * 32 + l31) * 2 + (half ^ ((l31 >> 3) & 1))) * 16);
    // B: lane i of a 16-lane group addresses row 8 * half + 4 t + (i >> 2) (+ 16 j), pixels wave_px + 16 * g1 + 4 * (i & 3)
    //    .. + 3 and receives channel rows 8 * half + 4 t + 0..3 of pixel wave_px + 16 * g1 + i
    unsigned b_addr[PXT];
    {
        const int i = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt) {
            const int wpxl = (wpx * PXT + pt) * 32;  // first pixel of the wave's tile within the 128-pixel block tile
            const int chunk = (((wpxl >> 5) ^ (i >> 2)) << 2) + 2 * g1 + ((i & 3) >> 1);
            b_addr[pt] = lds0 + (unsigned)((8 * half + (i >> 2)) * 256 + chunk * 16 + (i & 1) * 8);
        }
    }
    // bias slot addresses: the pair (col, col + 1) of register pair (r, r + 1), r even
    const unsigned bias_rd = lds0 + (unsigned)(BIAS_OFF + (wco * CT * 32 + 4 * half) * 4);

    pf_setup();
#pragma unroll
    for (int s = 0; s < BF_NST - 1; ++s) issue(s);
    int stage = 0;
    for (int k = 0; k < nitems; ++k) {
        const int idx = idx0 + k * gstep;
        const int jt = idx / a.nco, cot = idx - jt * a.nco;
        const int ptg = xcd * tpx + jt;
        const int n = ptg / a.tiles_per_img, tl = ptg - n * a.tiles_per_img;
        const int co0 = cot * COT, p0 = tl * PT;
        if (tid < COT) {  // (read two barriers later at the earliest; slot k & 1 was last read two items ago)
            const int m = co0 + tid;
            const float bv = (a.bias && m < a.M) ? a.bias[m] : 0.f;
            asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + (unsigned)(BIAS_OFF + ((k & 1) * COT + tid) * 4)), "v"(bv) : "memory");
        }
        f32x16 acc[CT][PXT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
        for (int i = 0; i < nchunks; ++i) {
            // the chunk at the head of the stream has landed once at most the loads of the NST - 2 chunks after it are
            // outstanding (this wave's pieces); the barrier extends that to every wave's pieces and says that everybody is done
            // reading the stage that the next issue overwrites
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((BF_NST - 2) * PPW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int s2 = stage + BF_NST - 1;
            s2 = s2 >= BF_NST ? s2 - BF_NST : s2;
            issue(s2);
            const unsigned sbase = (unsigned)(stage * STG);
            bf16x8 af[BF_KC / 16][CT];
            s16x4 bq[BF_KC / 16][PXT][2];
            static_for<BF_KC / 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<CT>([&](auto cc) {
                    constexpr int ct = decltype(cc)::value;
                    af[j][ct] = lds_rd128<(j * COT + ct * 32) * 32>(sbase + a_addr);
                });
                static_for<PXT>([&](auto pc) {
                    constexpr int pt = decltype(pc)::value;
                    bq[j][pt][0] = lds_rd_tr<j * 16 * 256>(sbase + b_addr[pt]);
                    bq[j][pt][1] = lds_rd_tr<j * 16 * 256 + 4 * 256>(sbase + b_addr[pt]);
                });
            });
            if constexpr (CT == 2 && PXT == 2) {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bq[0][0][0]), "+v"(bq[0][0][1]),
                               "+v"(bq[0][1][0]), "+v"(bq[0][1][1]), "+v"(bq[1][0][0]), "+v"(bq[1][0][1]), "+v"(bq[1][1][0]),
                               "+v"(bq[1][1][1])::"memory");
            } else {
                static_assert(CT == 2 && PXT == 1, "tile configurations: 2x2 or 2x1 MFMA tiles per wave");
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bq[0][0][0]), "+v"(bq[0][0][1]),
                               "+v"(bq[1][0][0]), "+v"(bq[1][0][1])::"memory");
            }
#pragma unroll
            for (int j = 0; j < BF_KC / 16; ++j)
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt) {
                    bf16x8 bf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bf[e] = bq[j][pt][0][e];
                        bf[4 + e] = bq[j][pt][1][e];
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j][ct], bf, acc[ct][pt], 0, 0, 0);
                }
            stage = stage + 1 >= BF_NST ? 0 : stage + 1;
        }

        // ---- epilogue: bias, floor, stores (the DMA of the next item's first chunks is in flight) ----
        float2 bia[CT][8];
        const unsigned badr = bias_rd + (unsigned)((k & 1) * COT * 4);
        static_for<CT * 8>([&](auto ic) {
            constexpr int ct = decltype(ic)::value / 8, e = decltype(ic)::value % 8;
            // register pair (2e, 2e + 1): rows ct * 32 + (2e & 3) + 8 * (2e >> 2) (+ 4 * half) and the next one
            constexpr int col = ct * 32 + ((2 * e) & 3) + 8 * ((2 * e) >> 2);
            bia[ct][e] = lds_rd64<col * 4>(badr);
        });
        if constexpr (CT == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(bia[0][0]), "+v"(bia[0][1]), "+v"(bia[0][2]), "+v"(bia[0][3]), "+v"(bia[0][4]), "+v"(bia[0][5]),
                           "+v"(bia[0][6]), "+v"(bia[0][7]), "+v"(bia[1][0]), "+v"(bia[1][1]), "+v"(bia[1][2]), "+v"(bia[1][3]),
                           "+v"(bia[1][4]), "+v"(bia[1][5]), "+v"(bia[1][6]), "+v"(bia[1][7])::"memory");
        }
        TO* obase = (TO*)a.out + (long)n * a.out_bs;
        bool pval[PXT];
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt) pval[pt] = p0 + (wpx * PXT + pt) * 32 + l31 < a.P;
        if constexpr (sizeof(TO) == 4) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int m = co0 + col;
                    const float bvv = (r & 1) ? bia[ct][r >> 1].y : bia[ct][r >> 1].x;
                    float* rowp = (float*)obase + (long)m * a.P + p0 + wpx * PXT * 32 + l31;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        if (pval[pt] && m < a.M) rowp[pt * 32] = fmaxf(acc[ct][pt][r] + bvv, a.out_floor);
                }
        } else {
            // bf16 rows: lanes (2e, 2e + 1) hold adjacent pixels; for the register pair (r, r + 1) = rows (m, m + 1) the even
            // lane stores row m, the odd lane row m + 1, each ONE dword = two pixels (one DPP exchange per pair)
            const bool odd = lane & 1;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int col = (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // row of register r
                    const float b0 = bia[ct][r >> 1].x, b1 = bia[ct][r >> 1].y;
                    const int m = co0 + col + (odd ? 1 : 0);
                    bf16_t* rowp = (bf16_t*)obase + (long)m * a.P + p0 + wpx * PXT * 32 + (l31 & ~1);
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt) {
                        const float v0 = fmaxf(acc[ct][pt][r] + b0, a.out_floor), v1 = fmaxf(acc[ct][pt][r + 1] + b1, a.out_floor);
                        const float send = odd ? v0 : v1;
                        const float recv = dpp_src<0xB1, 0xF>(send);  // quad_perm [1,0,3,2]: the neighbour lane of the pair
                        const unsigned pk = odd ? pack_bf16x2(recv, v1) : pack_bf16x2(v0, recv);
                        if (pval[pt] && m < a.M) *(unsigned*)(rowp + pt * 32) = pk;  // P is even: a pair is valid or not as a whole
                    }
                }
        }
        if (a.part) {  // BatchNorm partials of the raw accumulators (compiler-visible LDS: waits for the DMA in flight)
            int nw = a.P - (p0 + wpx * PXT * 32);
            nw = nw < 0 ? 0 : (nw > PXT * 32 ? PXT * 32 : nw);
            bn_wave_partials<CT, PXT>(acc, pval, l31, half, stat + wpx * 3 * COT + wco * CT * 32, COT, 0);
            if (lane == 0) ((int*)(stat + WPX * 3 * COT))[wpx] = nw;
            __syncthreads();
            for (int col = tid; col < COT; col += NTH) {
                float mean, m2, cnt;
                bn_tile_combine<WPX>(stat, (const int*)(stat + WPX * 3 * COT), COT, col, mean, m2, cnt);
                const int m = co0 + col;
                if (m < a.M) {
                    a.part[((long)0 * a.slots + ptg) * a.M + m] = mean;
                    a.part[((long)1 * a.slots + ptg) * a.M + m] = m2;
                    a.part[((long)2 * a.slots + ptg) * a.M + m] = cnt;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMA of the tail must not outlive the workgroup's LDS
}

template <auto KERN>
static int ensure_lds_b(size_t lds) {
    static size_t granted = 0;
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    return 0;
}

int pw_split_num_slots(int N, int P);  // splitmma.hip: N * ceil(P / 128)

template <int WCO, int CT, int WPX, int PXT, int GW, typename TO>
static int launch_pw_bf16_cfg(PwBfArgs& a, hipStream_t st) {
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32;
    a.nco = (a.M + COT - 1) / COT;
    a.tiles_per_img = (a.P + PT - 1) / PT;
    a.T = a.N * a.tiles_per_img;
    a.slots = pw_split_num_slots(a.N, a.P);
    const int items = ((a.T + 7) / 8) * 8 * a.nco;
    constexpr int BF_NST = BfStages<COT>::value;
    const size_t lds = (size_t)BF_NST * (BF_KC * PT * 2 + (BF_KC / 16) * COT * 32) + sizeof(float) * (BN_STAT_FLOATS(WPX, COT) + 2 * COT);
    constexpr auto kern = k_pw_bf16<WCO, CT, WPX, PXT, GW, TO>;
    int rc = ensure_lds_b<kern>(lds);
    if (rc) return rc;
    static_assert((size_t)BF_NST * (BF_KC * PT * 2 + (BF_KC / 16) * COT * 32) + sizeof(float) * (BN_STAT_FLOATS(WPX, COT) + 2 * COT) <=
                      160 * 1024 / 3, "k_pw_bf16: three workgroups per CU");
    const int grid = items < 768 ? items : 768;  // persistent: 3 workgroups per CU walk the items of their XCD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WCO * WPX * 64), lds, st, a);
    return (int)hipGetLastError();
}

// -2: shape not handled (odd plane size, image too large for 32-bit offsets)
int launch_pw_bf16(PwBfArgs& a, int out_dt, hipStream_t st) {
    if ((a.P & 1) != 0 || (long)a.Cin * a.P * 2 >= (1L << 31) || (a.x_bs & 1) != 0 || (a.out_bs & 1) != 0 ||
        ((((uintptr_t)a.x) & 3) != 0) || ((((uintptr_t)a.out) & 3) != 0) || ((((uintptr_t)a.planes) & 15) != 0))
        return -2;
    a.Cp = (a.Cin + 31) & ~31;
    const bool g16 = (a.P & 7) == 0 && (a.x_bs & 7) == 0 && ((((uintptr_t)a.x) & 15) == 0);
#define PWBF_GO(TO)                                                                                        \
    do {                                                                                                   \
        if (a.M > 64) {                                                                                    \
            if (g16) return launch_pw_bf16_cfg<2, 2, 2, 2, 16, TO>(a, st);                                 \
            return launch_pw_bf16_cfg<2, 2, 2, 2, 4, TO>(a, st);                                           \
        }                                                                                                  \
        if (g16) return launch_pw_bf16_cfg<1, 2, 4, 1, 16, TO>(a, st);                                     \
        return launch_pw_bf16_cfg<1, 2, 4, 1, 4, TO>(a, st);                                               \
    } while (0)
    if (out_dt == SMAAT_BF16) PWBF_GO(bf16_t);
    PWBF_GO(float);
#undef PWBF_GO
}

// =====================================================================================
// k_wgrad_bf16:  part[split][m][k] = sum over the split's 32-pixel chunks of dz[n][m][p] * y[n][k][p]
// =====================================================================================
#define WB_SPS 32  // pixels per stage (64-byte LDS rows)

template <int MTT, int GW>
__global__ __launch_bounds__(256) void k_wgrad_bf16(const WgBfArgs a) {
    constexpr int MT = 64 * MTT, KT = 128, ROWS = MT + KT;
    constexpr int STG = ROWS * 64;                       // bytes per stage
    constexpr int RPP = GW == 16 ? 16 : 4;               // rows per piece
    constexpr int NP = ROWS / RPP, PPW = NP / 4;
    static_assert(NP % 4 == 0, "pieces divide evenly among the four waves");
    static_assert(2 * PPW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 1, wk = wv >> 1;  // wave -> (m half, k half) of the block tile
    const int l31 = lane & 31, half = lane >> 5;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int ntile = a.nmt * a.nkt;
    const int rest = idx % ntile;
    const int split = (idx / ntile) * 8 + xcd;  // every tile of a pixel split runs on ONE XCD: dz / y tiles re-read from its L2
    if (split >= a.nsplit) return;
    const int mt = rest % a.nmt, kt = rest / a.nmt;
    const int m0 = mt * MT, k0 = kt * KT;
    // A 32-pixel chunk of a bf16 row is HALF a 128-byte line: a workgroup takes chunk PAIRS (2j, 2j + 1), j = split,
    // split + nsplit, ... so that the second half of every line it touches is an L2 hit on its own XCD a moment later
    // (with single chunks strided over the splits the two halves were fetched by workgroups on two different XCDs).
    const int npairs = (a.total_chunks + 1) >> 1;
    const int np_mine = split < npairs ? (npairs - split + a.nsplit - 1) / a.nsplit : 0;
    int nit = 2 * np_mine;
    if (nit > 0 && 2 * (split + (np_mine - 1) * a.nsplit) + 1 >= a.total_chunks) --nit;  // (odd total: the last pair is half)

    // lane -> (row within a piece, source pixel offset within the chunk); LDS chunk c' of a row holds source chunk
    // c' ^ ((row >> 2) & 3)
    int prow, pxo;
    if (GW == 16) {
        prow = lane >> 2;
        pxo = 8 * ((lane & 3) ^ ((prow >> 2) & 3));
    } else {
        prow = lane >> 4;
        pxo = 0;  // + the piece's (row >> 2) & 3, see below
    }
    auto issue = [&](int it_, int stage) __attribute__((always_inline)) {
        const int it = it_ < nit ? it_ : nit - 1;
        const int c = 2 * (split + (it >> 1) * a.nsplit) + (it & 1);
        const int n = c / a.nchunk_img;
        const int pc0 = (c - n * a.nchunk_img) * WB_SPS;
        unsigned char* sb = lds + stage * STG;
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int q = wv + 4 * u;       // piece: rows q * RPP .. + RPP - 1
            const int row = q * RPP + prow;
            int px;
            if (GW == 16) {
                px = pc0 + pxo;
            } else {
                const int d = lane & 15;
                px = pc0 + 8 * ((d >> 2) ^ ((row >> 2) & 3)) + 2 * (d & 3);
            }
            const bf16_t* src;
            if (row < MT) {
                const int m = m0 + row;
                src = a.dz + (long)n * a.dz_bs + (long)(m < a.M ? m : a.M - 1) * a.P + px;
            } else {
                const int k = k0 + row - MT;
                src = a.y + (long)n * a.y_bs + (long)(k < a.K ? k : a.K - 1) * a.P + px;
            }
            if (px >= a.P) src = (const bf16_t*)g_bf16_zero;  // the contraction runs over these positions: they must be 0
            glds<GW>(src, sb + q * (RPP * 64));
        }
    };
    // fragment address: row r, 8 pixels at 16 * s + 8 * half -> chunk c = 2 s + half at r * 64 + ((c ^ ((r >> 2) & 3)) * 16)
    unsigned fa[2], fb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int sw = ((2 * s + half) ^ ((l31 >> 2) & 3)) * 16;
        fa[s] = lds0 + (unsigned)(((wm * MTT) * 32 + l31) * 64 + sw);
        fb[s] = lds0 + (unsigned)((MT + (wk * 2) * 32 + l31) * 64 + sw);
    }
    f32x16 acc[MTT][2];
#pragma unroll
    for (int i = 0; i < MTT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nit > 0) {
        issue(0, 0);
        issue(1, 1);
    }
    int stage = 0;
    for (int it = 0; it < nit; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int s2 = stage + 2;
        s2 = s2 >= 3 ? s2 - 3 : s2;
        issue(it + 2, s2);
        const unsigned sbase = (unsigned)(stage * STG);
        bf16x8 af[2][MTT], bf[2][2];
        static_for<2>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            static_for<MTT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                af[s][i] = lds_rd128<i * 32 * 64>(sbase + fa[s]);
            });
            static_for<2>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                bf[s][j] = lds_rd128<j * 32 * 64>(sbase + fb[s]);
            });
        });
        if constexpr (MTT == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bf[0][0]), "+v"(bf[0][1]),
                           "+v"(bf[1][0]), "+v"(bf[1][1])::"memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0][0]), "+v"(af[1][0]), "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[1][0]), "+v"(bf[1][1])::"memory");
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < MTT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
        stage = stage + 1 >= 3 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* ob = a.part + (long)split * a.M * a.K;
#pragma unroll
    for (int i = 0; i < MTT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MTT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < a.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kg = k0 + (wk * 2 + j) * 32 + l31;
                    if (kg < a.K) ob[(long)m * a.K + kg] = acc[i][j][r];
                }
            }
        }
}

template <int MTT, int GW>
static int launch_wgrad_bf16_cfg(WgBfArgs& a, hipStream_t st) {
    constexpr int MT = 64 * MTT;
    a.nmt = (a.M + MT - 1) / MT;
    a.nkt = (a.K + 127) / 128;
    const size_t lds = (size_t)3 * (MT + 128) * 64;
    constexpr auto kern = k_wgrad_bf16<MTT, GW>;
    int rc = ensure_lds_b<kern>(lds);
    if (rc) return rc;
    const int grid = ((a.nsplit + 7) / 8) * 8 * a.nmt * a.nkt;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

// a.nsplit on entry: the number of partial tiles the caller's buffer holds (smaat_wgrad_num_splits); every one of them
// is written (a split without chunks writes zeros)
int launch_wgrad_bf16(WgBfArgs& a, hipStream_t st) {
    if ((a.P & 1) != 0 || (a.dz_bs & 1) != 0 || (a.y_bs & 1) != 0 || ((((uintptr_t)a.dz) & 3) != 0) ||
        ((((uintptr_t)a.y) & 3) != 0))
        return -2;
    a.nchunk_img = (a.P + WB_SPS - 1) / WB_SPS;
    a.total_chunks = a.N * a.nchunk_img;
    const bool g16 = (a.P & 7) == 0 && (a.dz_bs & 7) == 0 && (a.y_bs & 7) == 0 && ((((uintptr_t)a.dz) & 15) == 0) &&
                     ((((uintptr_t)a.y) & 15) == 0);
    if (a.M > 64) return g16 ? launch_wgrad_bf16_cfg<2, 16>(a, st) : launch_wgrad_bf16_cfg<2, 4>(a, st);
    return g16 ? launch_wgrad_bf16_cfg<1, 16>(a, st) : launch_wgrad_bf16_cfg<1, 4>(a, st);
}
