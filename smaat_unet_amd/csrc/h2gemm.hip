// PROTOTYPE (round 6): the pointwise GEMM on PRE-SPLIT fp16 operand planes -- no VALU in the loop.
//
//   out[n][m][p] = 2^-ksum * sum_c (Ah Xg + Ag Xh + Ah Xh)[m][c][p] + bias[m]
//
// Both operands arrive as two fp16 planes (h = rn16(t), g = rn16(t - h) of the scaled value t): the weight as two chunk-major
// images [2][Cp/16][M][16], the activation as [N][2][C][P] -- the SAME bytes as the f32 tensor it replaces.  Structure of
// k_pw_bf16 (bf16gemm.hip): persistent over (pixel tile, channel tile) items, operands global -> LDS by LDS-DMA, transposed LDS
// reads (ds_read_b64_tr_b16) for the activation fragments, counted waits, raw barriers; per 16-channel sub-chunk a wave reads
// 4 + 4 fragments (two planes each of 2 weight and 2 activation tiles) and issues 12 MFMAs (k_pw_bf16: 2 + 2 for 4).
// Not selected by ops.py: nothing in the step PRODUCES activation planes (that would take an a-priori bound of every GEMM operand
// in its producer: NOTEBOOK 4.9).  It exists to MEASURE what such a GEMM would reach on the deep layers
// (scripts/probes/h2_gemm_probe.py, profiles/r6/h2_gemm_probe_*.txt).
#include <utility>

#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));

struct PwH2Args {
    const unsigned short* x;       // [N][2][Cin][P] fp16 planes h, g
    long x_bs;                     // elements between images (2 Cin P)
    long xp_bs;                    // elements between the planes of an image (Cin P)
    const unsigned short* planes;  // [2][Cp/16][M][16] fp16 planes h, g of the weight
    long ap_bs;                    // elements between the weight planes
    const float* bias;
    float* out;
    long out_bs;
    float* part;  // [3][slots][M] or null
    int N, Cin, Cp, M, P, nco, tiles_per_img, T, slots;
    int ksum;     // ka + kx: the two power-of-two scale exponents
    float out_floor;
};

template <typename F, int... Is>
__device__ __forceinline__ void h2_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    h2_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
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
    static_assert(GW == 16, "LDS-DMA width");
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

template <int WCO, int CT, int WPX, int PXT, int KC, int NST, int WGS>
__global__ __launch_bounds__(WCO * WPX * 64, WGS) void k_pw_h2(const PwH2Args a) {
    constexpr int GW = 16;
    typedef float TO;
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32, NW = WCO * WPX, NTH = NW * 64;
    static_assert(PT == 128 && NW == 4, "128-pixel tiles (256-byte LDS rows), four waves");
    constexpr int XB1 = KC * PT * 2;            // one X plane of a stage: [KC][PT] fp16
    constexpr int AB1 = (KC / 16) * COT * 32;    // one A plane of a stage: [KC/16][COT][16] fp16
    constexpr int XB = 2 * XB1, AB = 2 * AB1;    // planes h, g
    constexpr int STG = XB + AB;
    constexpr int BF_NST = NST;
    constexpr int NXP = XB1 / 1024;  // X pieces (one wave-instruction each) per stage and plane
    constexpr int NAP = AB1 / 1024;
    static_assert(NXP % NW == 0 && NAP % NW == 0, "pieces divide evenly among the waves");
    constexpr int XPW = NXP / NW, APW = NAP / NW, PPW = 2 * (XPW + APW);
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
    const int nchunks = a.Cp / KC;
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
        const int k0 = pf_ch * KC;  // (Cin % KC == 0: checked by the launcher -- a prototype takes no ragged contraction)
        unsigned char* sb = lds + stage * STG;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const unsigned char* xs = pf_x + (long)pl * a.xp_bs * 2 + (long)k0 * rowbytes;
#pragma unroll
            for (int u = 0; u < XPW; ++u) glds<16>(xs + pf_xo[u], sb + pl * XB1 + (wv + NW * u) * 1024);
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const unsigned char* as = (const unsigned char*)a.planes + (long)pl * a.ap_bs * 2 + (long)(k0 >> 4) * a.M * 32;
#pragma unroll
            for (int u = 0; u < APW; ++u) {
                const int q = wv + NW * u;
                const int j = q / (COT / 32), rb = q - j * (COT / 32);
                glds<16>(as + pf_ao[u], sb + XB + pl * AB1 + (j * COT + rb * 32) * 32);
            }
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

    const float e1 = pow2i(-(a.ksum / 2)), e2 = pow2i(-(a.ksum - a.ksum / 2));
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
            bf16x8 af[KC / 16][CT][2];
            s16x4 bq[KC / 16][PXT][2][2];
            static_for<KC / 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<2>([&](auto plc) {
                    constexpr int pl = decltype(plc)::value;
                    static_for<CT>([&](auto cc) {
                        constexpr int ct = decltype(cc)::value;
                        af[j][ct][pl] = lds_rd128<pl * AB1 + (j * COT + ct * 32) * 32>(sbase + a_addr);
                    });
                    static_for<PXT>([&](auto pc) {
                        constexpr int pt = decltype(pc)::value;
                        bq[j][pt][pl][0] = lds_rd_tr<pl * XB1 + j * 16 * 256>(sbase + b_addr[pt]);
                        bq[j][pt][pl][1] = lds_rd_tr<pl * XB1 + j * 16 * 256 + 4 * 256>(sbase + b_addr[pt]);
                    });
                });
            });
            // (one wait per 16-channel sub-chunk: the first covers everything -- lgkmcnt counts in order -- and ties the registers)
            static_assert(CT == 2 && PXT == 2 && (KC == 16 || KC == 32), "prototype: 2 x 2 MFMA tiles per wave");
#define H2_TIE(j)                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                        \
                 : "+v"(af[j][0][0]), "+v"(af[j][0][1]), "+v"(af[j][1][0]), "+v"(af[j][1][1]), "+v"(bq[j][0][0][0]),           \
                   "+v"(bq[j][0][0][1]), "+v"(bq[j][0][1][0]), "+v"(bq[j][0][1][1]), "+v"(bq[j][1][0][0]), "+v"(bq[j][1][0][1]), \
                   "+v"(bq[j][1][1][0]), "+v"(bq[j][1][1][1])::"memory")
            H2_TIE(0);
            if constexpr (KC == 32) H2_TIE(KC / 16 - 1);
#undef H2_TIE
#pragma unroll
            for (int j = 0; j < KC / 16; ++j)
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt) {
                    // planes h, g of the activation fragment: the two transposed reads side by side (8 consecutive channels).
                    // (One shuffle + one bit_cast of the whole vector.  Element by element --
                    //  `bh[e] = __builtin_bit_cast(_Float16, bq[...][e])` -- hipcc 7.2 broadcast element 0 of each read into all four
                    //  positions: every group of four channels read as its first one, found with identity weights.)
                    const h2_f16x8 bh = __builtin_bit_cast(h2_f16x8, __builtin_shufflevector(bq[j][pt][0][0], bq[j][pt][0][1], 0, 1, 2, 3, 4, 5, 6, 7));
                    const h2_f16x8 bg = __builtin_bit_cast(h2_f16x8, __builtin_shufflevector(bq[j][pt][1][0], bq[j][pt][1][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {  // h g' + g h' + h h' (the order of the other two-term GEMMs)
                        const h2_f16x8 ah = __builtin_bit_cast(h2_f16x8, af[j][ct][0]), ag = __builtin_bit_cast(h2_f16x8, af[j][ct][1]);
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bg, acc[ct][pt], 0, 0, 0);
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, bh, acc[ct][pt], 0, 0, 0);
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ct][pt], 0, 0, 0);
                    }
                }
            stage = stage + 1 >= BF_NST ? 0 : stage + 1;
        }

        // ---- scale back by 2^-(ka + kx) (two exact factors), then the epilogue of k_pw_bf16: bias, floor, stores ----
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][pt][r] = acc[ct][pt][r] * e1 * e2;
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


int pw_split_num_slots(int N, int P);  // splitmma.hip: N * ceil(P / 128)

template <int KC, int NST, int WGS>
static int launch_pw_h2_cfg(PwH2Args& a, hipStream_t st) {
    constexpr int WCO = 2, CT = 2, WPX = 2, PXT = 2;
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32;
    a.nco = (a.M + COT - 1) / COT;
    a.tiles_per_img = (a.P + PT - 1) / PT;
    a.T = a.N * a.tiles_per_img;
    a.slots = pw_split_num_slots(a.N, a.P);
    const int items = ((a.T + 7) / 8) * 8 * a.nco;
    constexpr size_t lds = (size_t)NST * 2 * (KC * PT * 2 + (KC / 16) * COT * 32) + sizeof(float) * (BN_STAT_FLOATS(WPX, COT) + 2 * COT);
    static_assert(lds * WGS <= 160 * 1024, "LDS of the resident workgroups");
    constexpr auto kern = k_pw_h2<WCO, CT, WPX, PXT, KC, NST, WGS>;
    static size_t granted = 0;
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    const int cap = 256 * WGS;
    const int grid = items < cap ? items : cap;  // persistent
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

// cfg: 0 = 32-channel stages x 3, one workgroup per CU; 1 = 16-channel stages x 4, two workgroups per CU; 2 = 16 x 3, three
// -2: shape not handled by the prototype (M <= 64, Cin % 32, P % 8, alignment)
int launch_pw_h2(PwH2Args& a, int cfg, hipStream_t st) {
    if ((a.P & 7) != 0 || (a.Cin & 31) != 0 || a.M <= 64 || (a.x_bs & 7) != 0 || (a.xp_bs & 7) != 0 || ((((uintptr_t)a.x) & 15) != 0) ||
        ((((uintptr_t)a.out) & 3) != 0) || ((((uintptr_t)a.planes) & 15) != 0) || (long)a.Cin * a.P * 2 >= (1L << 31))
        return -2;
    a.Cp = a.Cin;
    if (cfg == 0) return launch_pw_h2_cfg<32, 3, 1>(a, st);
    if (cfg == 1) return launch_pw_h2_cfg<16, 4, 2>(a, st);
    if (cfg == 2) return launch_pw_h2_cfg<16, 3, 3>(a, st);
    return -1;
}

// C-side entry of the prototype (capi.hip declares the extern "C" wrapper)
int pw_h2_proto(const void* x, long x_bs, long xp_bs, const void* planes, long ap_bs, const float* bias, float* out, long out_bs,
                float* part, int N, int Cin, int M, int P, int ksum, int cfg, hipStream_t st) {
    if (N < 1 || Cin < 1 || M < 1 || P < 1 || !x || !planes || !out) return -1;
    PwH2Args a{};
    a.x = (const unsigned short*)x; a.x_bs = x_bs; a.xp_bs = xp_bs; a.planes = (const unsigned short*)planes; a.ap_bs = ap_bs;
    a.bias = bias; a.out = out; a.out_bs = out_bs; a.part = part; a.N = N; a.Cin = Cin; a.M = M; a.P = P; a.ksum = ksum;
    a.out_floor = -__builtin_inff();
    return launch_pw_h2(a, cfg, st);
}
