// Backward of a DepthwiseSeparableConv with the data gradient of the pointwise convolution NEVER in HBM (round 6):
//
//   dY[n][k][p]  = sum_m w_pw[m][k] * dz[n][m][p]                       (pointwise data gradient, k = 2 ci + j)
//   dX[n][ci][q] = sum_j sum_tap w_dw[2 ci + j][tap] * dY[n][2 ci + j][q - off(tap)]
//   dW_dw[k][tap] = sum_{n,q} act(x)[n][ci][q] * dY[n][k][q - off(tap)],   db_dw[k] = sum dY[n][k]
//   (+ the backward sums of the PREVIOUS BatchNorm when x is its input and the activation is applied on load)
//   reference: autograd of models/layers.py:47-50 (depthwise -> pointwise), unet_parts_depthwise_separable.py:17-36
//
// The two-kernel form writes dY (the 2x-expanded tensor: 4 K HW bytes per image) from k_pw_split_p and reads it back in
// k_dw3x3_bwd_rows: on the 288 x 288 layers 10.9 GB of the step's 119 GB.  Here a workgroup walks DOWN a band of rows of one
// 30-column strip of one image:
//   * 4 MFMA waves stream dz by rows (inline-asm loads, counted waits, two rows ahead), split the row into the two fp16 planes
//     of the B image in LDS ([pixel][m], 144-byte rows), hold the fp16 image of w_pw^T for their 32 of the tile's 128 k-rows in
//     registers and form ONE dY row per step (32 pixels x 128 k, contraction over the 64 m: 12 v_mfma_f32_32x32x16_f16)
//     into a two-row ring in LDS;
//   * 8 VALU waves (thread = channel ci of the 64-channel tile x 4 columns) bring the x rows in by LDS-DMA (six-row ring, three
//     steps ahead),
//     read the dY row of the previous step back (two ds_read_b128 per thread) and run the scatter form of
//     k_dw3x3_bwd_rows on it: three open dX rows, three activated x rows, the 2 x 10 weight-gradient sums -- in registers.
//   The horizontal halo of a 3x3 needs dY one column left and right of the columns whose dX a strip owns: MFMA N-tiles are 32
//   pixels wide at a STRIDE OF 30 (tile s = columns 30 s - 1 .. 30 s + 30, owns dX columns 30 s .. 30 s + 29): 6.7 % more MFMA
//   work, and row accesses that start at 4 (30 s - 1) bytes -- dword accesses through buffer descriptors (out-of-range = 0 /
//   dropped), whole-wave coalesced.
// ARITHMETIC: dY is formed by the SAME sequence of MFMAs as k_pw_split_p<NT = 2> forms it (chunks of 16 m ascending, terms
// h g', g h', h h', result * e1 * e2 + 0) and the depthwise part adds in the order of k_dw3x3_bwd_rows: dX is bit-identical to the
// two-kernel form (tests/test_gpu_kernels.py); the weight-gradient and BatchNorm sums are partitioned differently (per
// workgroup instead of per wave of a plane) and merged in fp64 by the same reducers.
// One pipeline slot per row, every slot unconditional, walks padded to the unroll depth (dswgrad.hip: the form
// scripts/isa_hazards.py can prove).
#include "common.h"
#include "rows_args.h"

typedef short dbw_s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dbw_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dbw_f16x2 __attribute__((ext_vector_type(2)));

// timing ablations for experiment builds (make EXTRA=-DDBW_DBG=<bits>; results are wrong, only the time means something):
// 1 no MFMAs / dY row writes, 2 no depthwise FMA block, 4 no dX stores, 8 no x DMA, 16 no dz image writes, 32 no per-step barrier
#ifndef DBW_DBG
#define DBW_DBG 0
#endif
#define DBW_TW 32     // MFMA N tile: columns per strip tile
#define DBW_VW 30     // dX columns a strip owns
#define DBW_YROW 48   // dwords per k-row of the dY ring (32 + 16 pad: the two k-rows of a channel start 32 banks apart)
#define DBW_BROW 144  // bytes per pixel row of the dz image (64 m x 2 B + 16: conflict-free ds_read_b128)
#define DBW_PD 3      // load sets per VALU thread (two rows ahead)
#define DBW_LPG 8     // loads per set of an MFMA-wave thread: 8 dwords of dz (a VALU-wave thread: 4 dwords of x)

__device__ __forceinline__ unsigned dbw_pack_f16(float a, float b) {
    const f32x2_native v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dbw_f16x2));
}
// two-term fp16 split of one value (splitmma.hip split2_f16, one lane of it): h = rn16(t), g = rn16(t - h)
__device__ __forceinline__ void dbw_split1(float t, unsigned short& h, unsigned short& g) {
    const unsigned hh = dbw_pack_f16(t, 0.f);
    const dbw_f16x2 hv = __builtin_bit_cast(dbw_f16x2, hh);
    h = (unsigned short)(hh & 0xFFFFu);
    g = (unsigned short)(dbw_pack_f16(t - (float)hv.x, 0.f) & 0xFFFFu);
}
typedef unsigned dbw_u32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (stride 0, num_records bytes) in SGPRs, for the inline-asm buffer loads
__device__ __forceinline__ dbw_u32x4 dbw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    dbw_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// LDS reads of the VALU waves are inline asm: beside LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read
// it can see (it cannot tell which bytes the DMA writes), i.e. the x rows would be waited for one step after their issue at the
// latest -- measured: 3 us per step instead of 0.8.  The waits are counted by hand (vmcnt: every step issues exactly 4 stores and
// 4 DMA loads per wave; lgkmcnt(0) with the destinations tied).
__device__ __forceinline__ f32x4 dbw_lds_rd128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ float dbw_sum8(float v) {  // sum over the 8 lanes of a channel (aligned 8-lane group)
    v += dpp_src<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_src<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_src<0x141, 0xF>(v);  // row_half_mirror
    return v;
}

__device__ __forceinline__ int b_hf(const DsBwArgs& a, int b) {  // channel half of workgroup b (see the kernel)
    const int wg = (b & 7) * (256 >> 3) + (b >> 3);
    return wg / a.wgh;
}

template <bool AFF>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_dsconv_bwd_rows(const DsBwArgs a) {
    constexpr int PD = DBW_PD;
    constexpr int BIMG = 2 * DBW_TW * DBW_BROW;        // bytes of one dz image (two planes)
    constexpr int YBUF = 128 * DBW_YROW * 4;           // bytes of one dY row (128 k)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* bimg = lds;                         // [2][plane][pixel][DBW_BROW]
    float* ybuf = (float*)(lds + 2 * BIMG);            // [2][128][DBW_YROW]
    float* xring = (float*)(lds + 2 * BIMG + 2 * YBUF);  // [6][64][32]: x rows of the tile by LDS-DMA (issued three steps ahead, read until two steps after)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool valu = wv >= 4;
    const int l31 = lane & 31, half = lane >> 5;

    // workgroup -> (channel half, contiguous item range); items = (image, band, strip), strips innermost
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int wg = xcd * (gridDim.x >> 3) + idx;       // contiguous ranges per XCD: neighbouring strips share dz lines in one L2
    const int hf = wg / a.wgh, wgi = wg - hf * a.wgh;  // channel half, workgroup within the half
    const int it_lo = wgi * a.ips;
    int it_hi = it_lo + a.ips;
    it_hi = it_hi > a.items ? a.items : it_hi;
    const int bps = a.bands * a.strips;
    const int kdz = f16_kexp(amax_read(a.dz_amax)), ka = *a.a_kexp;
    int kdz_s = kdz, ka_s = ka;
    asm volatile("" : "+s"(kdz_s), "+s"(ka_s));  // (the scalar loads are waited for here, not inside the walk)

    // the dY ring starts as zeros (the first two steps of the walk read it before any row was formed)
    for (int i = tid; i < 2 * 128 * DBW_YROW; i += 768) ybuf[i] = 0.f;
    float4* coef = (float4*)(xring + 6 * 64 * 32);  // AFF: [64] {scale, shift, mean, invstd} of the previous BatchNorm, by tile channel
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;  // asm reads take LDS byte addresses
    if (tid >= 256 && (tid & 7) == 0) {               // (read back per step: four VGPRs the AFF build does not have)
        const int c_ = (b_hf(a, blockIdx.x)) * 64 + ((tid - 256) >> 3);
        coef[(tid - 256) >> 3] = AFF ? make_float4(a.in_scale[c_], a.in_shift[c_], a.bn_mean[c_], a.bn_invstd[c_]) : make_float4(1.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    if (valu) {
        const int ptid = tid - 256;
        const int ci = ptid >> 3, g = ptid & 7;     // consume role: channel of the tile, 4-column group of the strip tile
        const int cg = hf * 64 + ci;                // channel of the layer
        float wt[2][9];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) wt[j][k] = a.w_dw[(cg * 2 + j) * 9 + k];
        // (complete the compiler-visible loads before the first inline-asm load: dswgrad.hip)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(wt[j][k]));
        // buffer descriptors: range-checked dword accesses (a strip tile starts at column 30 s - 1: out of the tensor for the
        // first element of the very first plane, and not 16-byte aligned anywhere)
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, a.dx_bytes, 0x00020000);

        float accw[2][10];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 10; ++k) accw[j][k] = 0.f;
        float r1s = 0.f, r2s = 0.f;

        for (int item = it_lo; item < it_hi; ++item) {
            const int n = item / bps, rem = item - n * bps;
            const int band = rem / a.strips, st_ = rem - band * a.strips;
            const int r0 = band * a.RB;
            const int r1 = (r0 + a.RB < a.H) ? r0 + a.RB : a.H;
            const int nsteps = (r1 - r0) + 4;                       // rows + halo rows + pipeline fill
            const int nsteps_pad = (nsteps + PD - 1) / PD * PD;
            const int c0 = st_ * DBW_VW - 1;                        // plane column of tile column 0
            // per-lane column state (fixed within the item)
            bool own[4];
            const unsigned vx = (unsigned)(cg * a.P + c0 + 4 * g) * 4u;  // dX byte offset of the thread's 4 columns (+ image / row: scalar)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tc = 4 * g + i, pc = c0 + tc;
                own[i] = tc >= 1 && tc <= DBW_VW && pc >= 0 && pc < a.W;
            }
            // x rows go global -> LDS by LDS-DMA (no destination registers: nothing for hipcc to copy or spill early, and 12 VGPRs
            // this kernel does not have).  One wave-instruction = two channels x 32 columns = 256 contiguous LDS bytes; a wave
            // loads the 8 channels ITS threads consume, so the data needs no barrier, only this wave's own counted vmcnt.
            // Row tau is issued at the end of step tau - 3; columns outside the plane are clamped here and deselected at use.
            const unsigned dimg = (unsigned)((long)n * a.dx_bs * 4);
            int xcol = c0 + (lane & 31);
            xcol = xcol < 0 ? 0 : (xcol >= a.W ? a.W - 1 : xcol);
            const float* xlane = a.x + (long)n * a.x_bs + (long)(hf * 64 + (wv - 4) * 8 + (lane >> 5)) * a.P + xcol;
            auto issue = [&](int tau) __attribute__((always_inline)) {  // x row r0 - 2 + tau -> ring slot tau mod 6
                int xr = r0 - 2 + tau;
                xr = xr < 0 ? 0 : (xr >= a.H ? a.H - 1 : xr);       // (rows outside the plane: a valid row, deselected at use)
                const float* src = xlane + (long)xr * a.W;
                float* dst = xring + ((tau + 12) % 6) * 2048 + (wv - 4) * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (long)(2 * j) * a.P),
                                                     (void __attribute__((address_space(3)))*)(dst + j * 64), 4, 0, 0);
            };
            float dxa[3][4], xc[3][4];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int c = 0; c < 4; ++c) dxa[s_][c] = xc[s_][c] = 0.f;

            // The walk starts PD steps early (the MFMA waves' reason, below; here the early steps issue the first two x rows): what
            // a step t < 2 "consumes" is outside every band -- selected to zero, never multiplied.
            for (int t0 = -PD; t0 < nsteps_pad; t0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int t = t0 + u;
                    // ---- consume dY row rho = r0 - 3 + t from ring slot t & 1 (written by the MFMA waves at step t - 1) ----
                    // (steps 0 and 1 of an item read a row of the previous item / the zeros of the kernel's start: rho < r0 - 1, every
                    // x row they meet is outside the band, i.e. zero, and no dX row they touch is stored -- no branch around the block)
                    {
                        const int rho = r0 - 3 + t;
                        const int sa = (u + 1) % 3, sb = (u + 2) % 3, sc_ = u % 3;  // rows rho - 1, rho, rho + 1  ((t - 2) % 3 == (u + 1) % 3)
                        // this wave's x rows up to row t have landed once at most the 16 operations of the last two steps are
                        // outstanding (4 stores + 4 DMA loads per step, in that order)
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        const unsigned ya = lds0 + (unsigned)(2 * BIMG + ((t & 1) * (128 * DBW_YROW) + (2 * ci) * DBW_YROW + 4 * g) * 4);
                        const unsigned xa = lds0 + (unsigned)(2 * BIMG + 2 * YBUF + (ci * 32 + 4 * g) * 4);
                        f32x4 y0 = dbw_lds_rd128(ya), y1 = dbw_lds_rd128(ya + DBW_YROW * 4);
                        f32x4 xo = dbw_lds_rd128(xa + (unsigned)(((t + 12) % 6) * 8192));   // x row of this step
                        f32x4 zr4 = dbw_lds_rd128(xa + (unsigned)(((t + 10) % 6) * 8192));  // ... of two steps ago (AFF: the raw row of the dX row completed now)
                        f32x4 cf = dbw_lds_rd128(lds0 + (unsigned)(2 * BIMG + 2 * YBUF + 6 * 8192 + ci * 16));  // AFF: {scale, shift, mean, invstd}
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y0), "+v"(y1), "+v"(xo), "+v"(zr4), "+v"(cf)::"memory");
                        float d[2][6];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 v = j ? y1 : y0;
                            const float l = dpp_src<0x111, 0xF>(v[3]);  // row_shr:1: lane - 1 (its last column)
                            const float r = dpp_src<0x101, 0xF>(v[0]);  // row_shl:1: lane + 1 (its first column)
                            d[j][0] = g == 0 ? 0.f : l;                 // (tile column -1: only ever meets the unowned column 0)
                            d[j][1] = v[0];
                            d[j][2] = v[1];
                            d[j][3] = v[2];
                            d[j][4] = v[3];
                            d[j][5] = g == 7 ? 0.f : r;
                        }
                        {  // open the slot of row rho + 1
                            const bool in = (rho + 1) >= r0 && (rho + 1) < r1;
                            const float xo_[4] = {xo[0], xo[1], xo[2], xo[3]};
                            const float asc = AFF ? cf[0] : 1.f, ash = AFF ? cf[1] : 0.f;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float zv = xo_[c];
                                const float av = AFF ? fmaxf(fmaf(zv, asc, ash), 0.f) : zv;
                                xc[sc_][c] = (in && own[c]) ? av : 0.f;
                                dxa[sc_][c] = 0.f;
                            }
                        }
                        const bool inb = rho >= r0 && rho < r1;
#pragma unroll
                        for (int j = 0; j < ((DBW_DBG & 2) ? 0 : 2); ++j) {
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
                            // bias gradient: dY over the OWNED columns of the band's rows
                            const float bsum = ((own[0] ? dv[1] : 0.f) + (own[1] ? dv[2] : 0.f)) + ((own[2] ? dv[3] : 0.f) + (own[3] ? dv[4] : 0.f));
                            accw[j][9] += inb ? bsum : 0.f;
                        }
                        const int rd = rho - 1;  // row rho - 1 is complete
                        const bool fin = rd >= r0 && rd < r1;
                        {   // ALWAYS four stores (a fixed operation count for the counted waits): rows that are not complete and
                            // unowned columns go out of range and are dropped by the descriptor
                            const unsigned ds_ = dimg + (unsigned)((fin ? rd : 0) * a.W) * 4u;
#pragma unroll
                            for (int c = 0; c < ((DBW_DBG & 4) ? 0 : 4); ++c)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dxa[sa][c]), rs_d,
                                                                      (fin && own[c]) ? vx + 4u * c : 0x80000000u, ds_, 0);
                            if (DBW_DBG & 4) asm volatile("" ::"v"(dxa[sa][0]), "v"(dxa[sa][1]), "v"(dxa[sa][2]), "v"(dxa[sa][3]), "s"(ds_));
                        }
                        if (AFF) {
                            const float zr_[4] = {zr4[0], zr4[1], zr4[2], zr4[3]};
                            const float rmean = cf[2], rinv = cf[3];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const bool ok = fin && xc[sa][c] > 0.f;
                                const float gg = ok ? dxa[sa][c] : 0.f;
                                r1s += gg;
                                r2s = fmaf(gg, ok ? (zr_[c] - rmean) * rinv : 0.f, r2s);  // (a select: the raw row of an early step is not data)
                            }
                        }
                    }
                    if (!(DBW_DBG & 8)) issue(t + 3);  // (behind the step's four stores: the order the counted wait assumes)
                    if (!(DBW_DBG & 32)) __syncthreads();
                }
            }
        }
        // workgroup partials: one row per (workgroup of the half), summed over the channel's 8 lanes
        const long prow = wgi;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float v = dbw_sum8(accw[j][k]);
                if (g == 7) a.part[(prow * a.K + cg * 2 + j) * 10 + k] = v;
            }
        if (AFF) {
            const float v1 = dbw_sum8(r1s), v2 = dbw_sum8(r2s);
            if (g == 7) {
                a.rpart[prow * a.Cin + cg] = v1;
                a.rpart[((long)a.wgh + prow) * a.Cin + cg] = v2;
            }
        }
    } else {
        // ---- MFMA waves: k-rows 32 wv .. 32 wv + 31 of the tile; A = fp16 image of w_pw^T, [M/16][2][K][16], resident ----
        const int kt = wv;
        dbw_s16x8 af[4][2];
        {
            const int krow = hf * 128 + kt * 32 + l31;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    af[s][p] = *(const dbw_s16x8*)(a.planes_t + (((long)s * 2 + p) * a.K + krow) * 16 + half * 8);
            // (complete these compiler-visible loads HERE: pending at the loop entry, hipcc waits for them with vmcnt(0) at their
            // first use inside the walk -- every step, draining the dz rows in flight: dswgrad.hip, round 4)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(af[s][p]));
        }
        const float e1 = pow2i(-((kdz_s + ka_s) / 2)), e2 = pow2i(-((kdz_s + ka_s) - (kdz_s + ka_s) / 2));
        const float sdz = pow2i(kdz_s);
        const dbw_u32x4 rs_z = dbw_rsrc(a.dz, a.dz_bytes);
        // staging role of these 256 threads: dz row m = tid >> 2, 8 pixels 8 (tid & 3) .. + 7 of the 32-pixel tile
        const int zm = tid >> 2, zq = tid & 3;
        for (int item = it_lo; item < it_hi; ++item) {
            const int n = item / bps, rem = item - n * bps;
            const int band = rem / a.strips, st_ = rem - band * a.strips;
            const int r0 = band * a.RB;
            const int r1 = (r0 + a.RB < a.H) ? r0 + a.RB : a.H;
            const int nsteps = (r1 - r0) + 4;
            const int nsteps_pad = (nsteps + PD - 1) / PD * PD;
            const int c0 = st_ * DBW_VW - 1;
            const int pz = c0 + 8 * zq;
            const unsigned vz = (unsigned)(zm * a.P + pz) * 4u;
            bool zok[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) zok[i] = (pz + i) >= 0 && (pz + i) < a.W;
            const unsigned zimg = (unsigned)((long)n * a.dz_bs * 4);
            float sz[PD][8];
            // set tau: dz row r0 - 1 + tau (staged at step tau)
            auto issue = [&](int set, int tau) __attribute__((always_inline)) {
                int zr = r0 - 1 + tau;
                zr = zr < 0 ? 0 : (zr >= a.H ? a.H - 1 : zr);
                const unsigned zs = zimg + (unsigned)(zr * a.W) * 4u;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(sz[set][0]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:4" : "=v"(sz[set][1]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:8" : "=v"(sz[set][2]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:12" : "=v"(sz[set][3]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:16" : "=v"(sz[set][4]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:20" : "=v"(sz[set][5]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:24" : "=v"(sz[set][6]) : "v"(vz), "s"(rs_z), "s"(zs));
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:28" : "=v"(sz[set][7]) : "v"(vz), "s"(rs_z), "s"(zs));
            };
            auto wait_set = [&](int set) __attribute__((always_inline)) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 2) * DBW_LPG) : "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(sz[set][i]));
            };
            // The walk starts PD steps EARLY instead of issuing the first sets in a prologue: a register set that is defined both
            // in front of the loop and inside it reaches the loop through a phi, and hipcc resolved such a phi with v_mov copies
            // of the just-issued (not yet landed) registers in front of the first wait -- scripts/isa_hazards.py caught it in the
            // first build of this kernel.  Steps t < 0 only issue (what they stage is overwritten before any MFMA reads it).
            for (int t0 = -PD; t0 < nsteps_pad; t0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int t = t0 + u;
                    wait_set(u);
                    {   // ---- stage dz row r0 - 1 + t into image t & 1 (rows / columns outside the plane: zeros) ----
                        const int zr = r0 - 1 + t;
                        const bool zin = zr >= 0 && zr < a.H && t >= 0 && t < (r1 - r0) + 2;
                        unsigned char* bb = bimg + (t & 1) * BIMG + (8 * zq) * DBW_BROW + zm * 2;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float v = (zin && zok[i]) ? sz[u][i] : 0.f;
                            unsigned short h, gq;
                            dbw_split1(v * sdz, h, gq);
                            if (DBW_DBG & 16) {
                                asm volatile("" ::"v"((unsigned)h), "v"((unsigned)gq));
                                continue;
                            }
                            *(unsigned short*)(bb + i * DBW_BROW) = h;
                            *(unsigned short*)(bb + i * DBW_BROW + DBW_TW * DBW_BROW) = gq;
                        }
                    }
                    if (!(DBW_DBG & 1) && t >= 1 && t < (r1 - r0) + 3) {  // dY row r0 - 2 + t from the dz image of step t - 1
                        const unsigned char* bb = bimg + ((t - 1) & 1) * BIMG + l31 * DBW_BROW + half * 16;
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const dbw_f16x8 b0 = *(const dbw_f16x8*)(bb + s * 32);
                            const dbw_f16x8 b1 = *(const dbw_f16x8*)(bb + s * 32 + DBW_TW * DBW_BROW);
                            const dbw_f16x8 a0 = __builtin_bit_cast(dbw_f16x8, af[s][0]), a1 = __builtin_bit_cast(dbw_f16x8, af[s][1]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);  // (the term order of mma_terms<2>)
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
                        }
                        float* yb = ybuf + ((t - 1) & 1) * (128 * DBW_YROW) + (kt * 32 + 4 * half) * DBW_YROW + l31;
#pragma unroll
                        for (int r = 0; r < 16; ++r) yb[((r & 3) + 8 * (r >> 2)) * DBW_YROW] = acc[r] * e1 * e2 + 0.f;  // (k_pw_split_p's epilogue)
                    }
                    issue((u + 2) % PD, t + 2);
                    if (!(DBW_DBG & 32)) __syncthreads();
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus sets of the item's tail
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
int dsconv_bwd_rows_ok(int kpl, int Cin, int M, int H, int W) {
    return kpl == 2 && M == 64 && (Cin == 64 || Cin == 128) && H >= 8 && W >= 32;
}

static void dsbw_geom(DsBwArgs& a) {
    a.P = a.H * a.W;
    a.strips = (a.W + DBW_VW - 1) / DBW_VW;
    a.nhalf = a.Cin / 64;
    a.wgh = 256 / a.nhalf;  // workgroups per channel half (one workgroup per CU)
    // band length: every item costs 4 extra steps; pick the divisor-free cut that minimises the steps of the busiest workgroup
    int rb = a.H;
    long best = -1;
    for (int cand = 72; cand >= 12; --cand) {
        if (cand > a.H) continue;
        const int bands = (a.H + cand - 1) / cand;
        const long items = (long)a.N * bands * a.strips;
        const long cost = ((items + a.wgh - 1) / a.wgh) * (cand + 4 + DBW_PD);
        if (best < 0 || cost < best) {
            best = cost;
            rb = cand;
        }
    }
    a.RB = rb;
    a.bands = (a.H + rb - 1) / rb;
    a.items = a.N * a.bands * a.strips;
    a.ips = (a.items + a.wgh - 1) / a.wgh;
}

int dsconv_bwd_rows_num_rows(int N, int Cin, int H, int W) {
    (void)N; (void)H; (void)W;
    return 256 / (Cin / 64);
}

// -2: shape / alignment not handled (the caller keeps the two-kernel form)
int launch_dsconv_bwd_rows(DsBwArgs& a, int kpl, hipStream_t st) {
    if (!dsconv_bwd_rows_ok(kpl, a.Cin, a.M, a.H, a.W) || a.K != 2 * a.Cin) return -2;
    if (!a.dz_amax || !a.a_kexp || !a.planes_t || !a.dx) return -2;
    const long xb = ((long)(a.N - 1) * a.x_bs + (long)a.Cin * a.H * a.W) * 4, zb = ((long)(a.N - 1) * a.dz_bs + (long)a.M * a.H * a.W) * 4;
    const long db = ((long)(a.N - 1) * a.dx_bs + (long)a.Cin * a.H * a.W) * 4;
    if (xb >= (1L << 31) || zb >= (1L << 31) || db >= (1L << 31)) return -2;  // (32-bit byte offsets, bit 31 = "drop")
    a.x_bytes = (unsigned)xb;
    a.dz_bytes = (unsigned)zb;
    a.dx_bytes = (unsigned)db;
    dsbw_geom(a);
    const bool aff = a.in_scale != nullptr;
    if (aff && (!a.in_shift || !a.bn_mean || !a.bn_invstd || !a.rpart)) return -1;
    const size_t lds = (size_t)2 * 2 * DBW_TW * DBW_BROW + (size_t)2 * 128 * DBW_YROW * 4 + (size_t)6 * 64 * 32 * 4 + 64 * 16;
    if (aff) {
        static size_t granted = 0;
        if (lds > granted) {
            HIP_RET(hipFuncSetAttribute((const void*)k_dsconv_bwd_rows<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            granted = lds;
        }
        hipLaunchKernelGGL(k_dsconv_bwd_rows<true>, dim3(256), dim3(768), lds, st, a);
    } else {
        static size_t granted = 0;
        if (lds > granted) {
            HIP_RET(hipFuncSetAttribute((const void*)k_dsconv_bwd_rows<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            granted = lds;
        }
        hipLaunchKernelGGL(k_dsconv_bwd_rows<false>, dim3(256), dim3(768), lds, st, a);
    }
    return (int)hipGetLastError();
}
