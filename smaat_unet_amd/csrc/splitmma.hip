// f32 GEMMs on the bf16 matrix pipe by operand splitting (gfx950 has no xf32/TF32 MFMA and its
// f32-input MFMA runs at 1/16 of the bf16 rate).
//
// Every f32 operand a is split EXACTLY into three bf16 terms  a = a1 + a2 + a3  (8 significant bits
// each; truncation splits, residuals computed in f32 are exact).  A product a*b is then
//     a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)  +  O(2^-24 |a b|)
// i.e. SIX v_mfma_f32_32x32x16_bf16 per 16-deep step, each term exact in the f32 accumulator input
// (8 bit x 8 bit products), f32 accumulation as in the f32 MFMA.  The dropped terms are below one f32
// ulp of the product, so the result has f32-class error (tests compare against the fp64 oracle next
// to the f32-MFMA kernels).
//
// NT = 2 (round 5) is the TWO-term fp16 split:  a * 2^ka = h1 + h2 + O(2^-22 |a| 2^ka)  with h1 = fp16(a 2^ka),
// h2 = fp16(a 2^ka - h1) (round to nearest; 11 significant bits each) and a power-of-two scale 2^ka PER OPERAND TENSOR that
// puts the tensor's largest magnitude into [2^14, 2^15) (fp16 overflows at 65504; its normal range reaches 28 binades
// below that).  A product is  h1 g1 + h1 g2 + h2 g1  -- THREE v_mfma_f32_32x32x16_f16, each term an exact 22-bit product,
// f32 accumulation; the dropped term h2 g2 is 2^-22 of the product -- and the accumulator is scaled back by the exact
// 2^-(ka+kb).  Elements more than 2^17 below the tensor's maximum lose relative precision (their second term goes
// subnormal), with an absolute error still <= 2^-39 of the maximum: f32-class in every norm a GEMM result is judged in
// (profiles/r4/split_formats_*_study.txt; tests/test_gpu_kernels.py compares with fp64 next to the three-term split).
// The maxima come from the kernels that WRITE the operands (amax_publish, common.h): depthwise forward (y), BatchNorm
// backward apply (dz), the weight-image kernel (weights); a call without them runs the three-term bf16 split.
//
//   k_wgrad_split   dW[m][k] = sum_{n,p} dz[n][m][p] * y[n][k][p]      (pointwise weight gradient)
//
// Structure = the wave-specialised pipeline of pwgemm.hip: 4 consumer waves (ds_read_b128 + MFMA
// only), NPW producer waves (global float4 loads -> split -> ds_write_b64 into [plane][row][32 px]
// bf16 images), double-buffered LDS, one barrier per 32-pixel chunk.
#include "common.h"
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static int pws_cfg();
int split_mode();

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// low half <- bf16 of x, high half <- bf16 of y (both already have zero low 16 bits or are truncated here)
__device__ __forceinline__ unsigned pack_hi16(float x, float y) {
    return __builtin_amdgcn_perm(fbits(y), fbits(x), 0x07060302u);  // one v_perm_b32: {y.hi16, x.hi16}
}
__device__ __forceinline__ float rne_bf16(float x) {  // round-to-nearest-even to 8 significant bits, kept as f32
    const unsigned b = fbits(x);
    return bitsf((b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u);
}

// out = max(v, floor) that PROPAGATES a NaN in v (fmaxf returns the other operand: a NaN accumulator used to come out as the
// floor, -inf when there is no ReLU).  floor itself is never NaN.
__device__ __forceinline__ float floor_nan(float v, float fl) { return v < fl ? fl : v; }

// ---- two-term fp16 split of a pair (already scaled): h = {fp16(t0), fp16(t1)}, g = fp16 of the exact residuals
__device__ __forceinline__ unsigned pack_f16(float a, float b) {  // one v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2_native v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split2_f16(float t0, float t1, unsigned& h, unsigned& g) {
    h = pack_f16(t0, t1);
    const f16x2 hv = __builtin_bit_cast(f16x2, h);
    g = pack_f16(t0 - (float)hv.x, t1 - (float)hv.y);  // the residual of a rounding to fewer bits is exact in f32
}

// NT MFMA terms of one 32x32x16 block product, smallest terms first (NT = 3 / 1: bf16 planes, NT = 2: fp16 planes)
template <int NT>
__device__ __forceinline__ void mma_terms(f32x16& acc, const bf16x8 (&a)[NT], const bf16x8 (&b)[NT]) {
    if constexpr (NT == 2) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
        const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    } else {
        if constexpr (NT == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
}

// split 4 consecutive f32 into NT planes; plane t of the 4 values -> uint2.  NT = 3 / 1: bf16 terms; NT = 2: fp16 terms
// of v * sc (sc = the operand's power-of-two scale)
template <int NT>
__device__ __forceinline__ void split4(const float4 v, uint2 (&out)[NT], float sc) {
    if constexpr (NT == 2) {
        split2_f16(v.x * sc, v.y * sc, out[0].x, out[1].x);
        split2_f16(v.z * sc, v.w * sc, out[0].y, out[1].y);
    } else {
        const float x[4] = {v.x, v.y, v.z, v.w};
        float p1[4], p2[4], p3[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p1[i] = NT == 1 ? rne_bf16(x[i]) : bitsf(fbits(x[i]) & 0xFFFF0000u);
            const float r1 = x[i] - p1[i];  // exact
            if (NT == 1) {
                p2[i] = p3[i] = 0.f;
            } else {
                p2[i] = bitsf(fbits(r1) & 0xFFFF0000u);
                p3[i] = r1 - p2[i];  // exact, <= 8 significant bits: its truncation to bf16 is exact
            }
        }
        out[0] = make_uint2(pack_hi16(p1[0], p1[1]), pack_hi16(p1[2], p1[3]));
        if constexpr (NT == 3) {
            out[1] = make_uint2(pack_hi16(p2[0], p2[1]), pack_hi16(p2[2], p2[3]));
            out[2] = make_uint2(pack_hi16(p3[0], p3[1]), pack_hi16(p3[2], p3[3]));
        }
    }
}

// 8 f32 values of one pixel (8 consecutive channels) -> NT planes of a [pixel][16 ch] image, one 16-byte store per plane
template <int NT>
__device__ __forceinline__ void split8_store(const float (&x)[8], float sc, unsigned char* dst, int plane_bytes) {
    if constexpr (NT == 2) {
        unsigned h[4], g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2_f16(x[2 * e] * sc, x[2 * e + 1] * sc, h[e], g[e]);
        *(uint4*)(dst) = make_uint4(h[0], h[1], h[2], h[3]);
        *(uint4*)(dst + plane_bytes) = make_uint4(g[0], g[1], g[2], g[3]);
    } else {
        float p1[8], p2[8], p3[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p1[e] = NT == 1 ? rne_bf16(x[e]) : bitsf(fbits(x[e]) & 0xFFFF0000u);
            const float r1 = x[e] - p1[e];
            p2[e] = bitsf(fbits(r1) & 0xFFFF0000u);
            p3[e] = r1 - p2[e];
        }
        *(uint4*)(dst) = make_uint4(pack_hi16(p1[0], p1[1]), pack_hi16(p1[2], p1[3]), pack_hi16(p1[4], p1[5]),
                                    pack_hi16(p1[6], p1[7]));
        if constexpr (NT == 3) {
            *(uint4*)(dst + plane_bytes) = make_uint4(pack_hi16(p2[0], p2[1]), pack_hi16(p2[2], p2[3]),
                                                      pack_hi16(p2[4], p2[5]), pack_hi16(p2[6], p2[7]));
            *(uint4*)(dst + 2 * plane_bytes) = make_uint4(pack_hi16(p3[0], p3[1]), pack_hi16(p3[2], p3[3]),
                                                          pack_hi16(p3[4], p3[5]), pack_hi16(p3[6], p3[7]));
        }
    }
}

#define SPS 32       // pixels per chunk
#define SROW 80      // bytes per LDS row (32 bf16 + 16 B pad: conflict-free ds_read_b128 / ds_write_b64)

// MTT: 32-row m tiles per consumer wave (2 -> 128-row block tile, 1 -> 64-row block tile); k tile = 128
template <int NT, int MTT, int NPT>
__global__ __launch_bounds__(256 + NPT) void k_wgrad_split(const Wg2Args a) {
    constexpr int MT = 64 * MTT, KT = 128, ROWS = MT + KT;
    constexpr int PLSZ = ROWS * SROW;          // bytes per plane
    constexpr int BUFSZ = NT * PLSZ;           // bytes per buffer
    constexpr int NF4 = ROWS * (SPS / 4) / NPT;  // float4 loads per producer thread per chunk
    static_assert((ROWS * (SPS / 4)) % NPT == 0, "producer mapping");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;
    const int wave = wv & 3;
    const int wm = wave & 1, wk = wave >> 1;  // consumer wave -> (m half, k half)
    const int l31 = lane & 31, half = lane >> 5;
    const int ptid = producer ? tid - 256 : 0;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int ntile = a.nmt * a.nkt;
    const int rest = idx % ntile;
    const int split = (idx / ntile) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int mt = rest % a.nmt, kt = rest / a.nmt;
    const int m0 = mt * MT, k0 = kt * KT;
    const int c_begin = split, c_end = a.total_chunks, c_step = a.nsplit;
    int nit = 0;
    if (c_begin < c_end) nit = (c_end - c_begin + c_step - 1) / c_step;
    constexpr int PD = 4;                             // register sets of loads in flight per producer thread
    const int nit_pad = (nit + PD - 1) / PD * PD;     // barriers of the walk (producers and consumers alike)
    // NT == 2: per-tensor power-of-two scales of the two operands (wave-uniform scalar loads)
    int kdz = 0, ky = 0;
    if constexpr (NT == 2) {
        kdz = f16_kexp(amax_read(a.dz_amax));
        ky = f16_kexp(amax_read(a.y_amax));
        asm volatile("" : "+s"(kdz), "+s"(ky));  // (the scalar loads are waited for HERE, not at a first use inside the loop)
    }

    if (producer) {
        const float sdz = pow2i(kdz), sy = pow2i(ky);
        // thread -> (row group, float4 column): 8 lanes cover one 128-byte row segment
        // 8 lanes cover one 128-byte row segment (32 pixels).  Within a group of 8 rows the row order is
        // 0,4,1,5,2,6,3,7: the two rows a 16-lane group writes with one ds_write_b64 are 4 rows (320 B = 16
        // banks mod 32) apart, so their 64-byte pieces tile the 32 banks (consecutive rows, 20 banks apart,
        // overlap on a quarter of the banks).
        const int q = ptid & 7, g8 = ptid >> 3;
        const int rbase = (g8 & ~7) | ((g8 & 1) << 2) | ((g8 >> 1) & 3);
        constexpr int RSTEP = NPT / 8;
        const float* rowp[NF4];
        bool rowv[NF4];
        int lofs[NF4];
#pragma unroll
        for (int j = 0; j < NF4; ++j) {
            const int row = rbase + RSTEP * j;
            if (row < MT) {
                rowv[j] = (m0 + row) < a.M;
                rowp[j] = a.dz + (long)(rowv[j] ? m0 + row : a.M - 1) * a.P;
            } else {
                rowv[j] = (k0 + row - MT) < a.K;
                rowp[j] = a.y + (long)(rowv[j] ? k0 + row - MT : a.K - 1) * a.P;
            }
            lofs[j] = row * SROW + q * 8;
        }
        // PD register sets of loads in flight.  The loads are issued through inline asm and waited for with an
        // explicit counted s_waitcnt (hipcc's own bookkeeping drains to vmcnt(0) at every commit, which would
        // leave ONE chunk in flight per CU: the iteration time was the memory latency, ~1.7 us, against
        // ~0.65 us of MFMA work).  Invariant: between the prefetch of a set and its commit exactly PD - 1
        // other prefetches (NF4 loads each) are issued.
        static_assert((PD - 1) * NF4 <= 63, "vmcnt is a 6-bit counter");
        f32x4 pf[PD][NF4];
        bool pin[PD];
        int pf_it = 0;  // next chunk to prefetch
        auto prefetch = [&](int set) __attribute__((always_inline)) {
            const int it = pf_it < nit ? pf_it : nit - 1;
            ++pf_it;
            const int c = c_begin + it * c_step;
            const int n = c / a.nchunk_img;
            const int p0 = (c - n * a.nchunk_img) * SPS + q * 4;
            const bool in = p0 < a.P;  // P % 4 == 0: a float4 is entirely inside or outside the plane
            const int p0c = in ? p0 : 0;
            pin[set] = in;
#pragma unroll
            for (int j = 0; j < NF4; ++j) {
                const int row = rbase + RSTEP * j;
                const float* src = rowp[j] + (long)n * (row < MT ? a.dz_bs : a.y_bs) + p0c;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pf[set][j]) : "v"(src));
            }
        };
        auto commit = [&](int buf, int set, bool live) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NF4) : "memory");
#pragma unroll
            for (int j = 0; j < NF4; ++j) asm volatile("" : "+v"(pf[set][j]));  // uses stay behind the wait
            if (!live) return;  // surplus iteration of the padded walk: waited for, not written
            unsigned char* base = lds + buf * BUFSZ;
#pragma unroll
            for (int j = 0; j < NF4; ++j) {
                const bool ok = pin[set] && rowv[j];
                float4 v;
                v.x = ok ? pf[set][j][0] : 0.f;
                v.y = ok ? pf[set][j][1] : 0.f;
                v.z = ok ? pf[set][j][2] : 0.f;
                v.w = ok ? pf[set][j][3] : 0.f;
                uint2 pl[NT];
                split4<NT>(v, pl, (rbase + RSTEP * j) < MT ? sdz : sy);
#pragma unroll
                for (int t = 0; t < NT; ++t) *(uint2*)(base + t * PLSZ + lofs[j]) = pl[t];
            }
        };
        if (nit > 0) {
#pragma unroll
            for (int j = 0; j < PD; ++j) prefetch(j);
            commit(0, 0, true);
            prefetch(0);
        }
        __syncthreads();
        // The walk runs over nit rounded up to a multiple of PD with every slot of the unrolled body unconditional (the surplus
        // iterations wait, re-issue the last chunk's loads and write nothing): on every path of the control-flow graph a set is
        // committed exactly PD prefetches after it was issued, which scripts/isa_hazards.py proves on the generated ISA
        // (dswgrad.hip, round 6).
        for (int it0 = 0; it0 < nit_pad; it0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                commit((it0 + u + 1) & 1, (u + 1) % PD, it0 + u + 1 < nit);  // chunk it+1: its loads were issued PD chunks ago
                prefetch((u + 1) % PD);
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus prefetches of the tail
    } else {
        f32x16 acc[MTT][2];
#pragma unroll
        for (int i = 0; i < MTT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const unsigned char* base = lds + (it & 1) * BUFSZ;
            // A rows = dz rows of this wave's m tiles, B rows = y rows of its k tiles; lane -> row l31,
            // 8 consecutive pixels at 16*step + 8*half
            const unsigned char* ap = base + ((wm * MTT) * 32 + l31) * SROW + half * 16;
            const unsigned char* bp = base + (MT + (wk * 2) * 32 + l31) * SROW + half * 16;
#pragma unroll
            for (int s = 0; s < SPS / 16; ++s) {
                bf16x8 af[MTT][NT], bf[2][NT];
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) af[i][t] = *(const bf16x8*)(ap + t * PLSZ + i * 32 * SROW + s * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) bf[j][t] = *(const bf16x8*)(bp + t * PLSZ + j * 32 * SROW + s * 32);
#pragma unroll
                for (int i = 0; i < MTT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma_terms<NT>(acc[i][j], af[i], bf[j]);  // smallest terms first
            }
            __syncthreads();
        }
        for (int it = nit; it < nit_pad; ++it) __syncthreads();  // the producers' surplus iterations
        float* ob = a.part + (long)split * a.M * a.K;
        // two exact factors, the exponent split evenly: 2^-(kdz + ky) itself may be out of range, and so may the product of
        // the accumulator with 2^-kdz alone while the final result is not
        const float e1 = pow2i(-((kdz + ky) / 2)), e2 = pow2i(-((kdz + ky) - (kdz + ky) / 2));
#pragma unroll
        for (int i = 0; i < MTT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MTT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int kg = k0 + (wk * 2 + j) * 32 + l31;
                        if (kg < a.K) ob[(long)m * a.K + kg] = NT == 2 ? acc[i][j][r] * e1 * e2 : acc[i][j][r];
                    }
                }
            }
    }
}

template <auto KERN>
static int ensure_lds_s(size_t lds) {
    static size_t granted = 0;
    if (lds > granted) {
        HIP_RET(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    return 0;
}

template <int NT, int MTT, int NPT>
static int launch_wgrad_split_cfg(Wg2Args& a, hipStream_t st) {
    constexpr int MT = 64 * MTT;
    a.nmt = (a.M + MT - 1) / MT;
    a.nkt = (a.K + 127) / 128;
    const size_t lds = (size_t)2 * NT * (MT + 128) * SROW;
    constexpr auto kern = k_wgrad_split<NT, MTT, NPT>;
    int rc = ensure_lds_s<kern>(lds);
    if (rc) return rc;
    const int grid = ((a.nsplit + 7) / 8) * 8 * a.nmt * a.nkt;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 + NPT), lds, st, a);
    return (int)hipGetLastError();
}

// SMAAT_SPLIT / smaat_set_split_mode: 0 = f32 MFMA kernels only, 3 (default) = exact three-term split (six
// bf16 MFMAs per product, f32-class error), 2 = two-term split (three MFMAs, ~1e-5), 1 = operands rounded to
// bf16 (ONE MFMA: the bf16 mixed-precision mode of BASELINE configs[3], ~1e-2)
static int g_split_mode = -1;
int split_mode() {
    if (g_split_mode < 0) {
        const char* e = getenv("SMAAT_SPLIT");
        g_split_mode = e ? atoi(e) : 3;
    }
    return g_split_mode;
}
int set_split_mode(int m) {
    const int prev = split_mode();
    g_split_mode = m;
    return prev;
}

// returns -2 when the shape / alignment is not handled here (caller falls back to the f32 MFMA kernel)
int launch_wgrad_split(Wg2Args& a, int nt, hipStream_t st) {
    const bool vec = ((a.P & 3) == 0) && (a.P >= 4) && ((a.dz_bs & 3) == 0) && ((a.y_bs & 3) == 0) &&
                     ((((uintptr_t)a.dz) & 15) == 0) && ((((uintptr_t)a.y) & 15) == 0);
    if (!vec) return -2;
    a.nchunk_img = (a.P + SPS - 1) / SPS;
    a.total_chunks = a.N * a.nchunk_img;
    if (a.nsplit > a.total_chunks) a.nsplit = a.total_chunks;
    if (nt == 1) {  // plain bf16 operands (mixed-precision mode)
        if (a.M > 64) return launch_wgrad_split_cfg<1, 2, 256>(a, st);
        return launch_wgrad_split_cfg<1, 1, 256>(a, st);
    }
    if (nt == 3) {
        if (a.M > 64 && (pws_cfg() & 8)) return launch_wgrad_split_cfg<3, 2, 512>(a, st);
        if (a.M > 64) return launch_wgrad_split_cfg<3, 2, 256>(a, st);
        return launch_wgrad_split_cfg<3, 1, 256>(a, st);
    }
    if (!a.dz_amax || !a.y_amax) return -1;  // nt == 2: the two-term fp16 split needs both operand maxima
    if (a.M > 64) return launch_wgrad_split_cfg<2, 2, 256>(a, st);
    return launch_wgrad_split_cfg<2, 1, 256>(a, st);
}

// =====================================================================================
// k_split_planes: f32 matrix [R][C] -> three bf16 planes in CHUNK-MAJOR order [Cp/16][3][R][16], Cp = C
// rounded up to 16 (zero padded): the A operand of one 16-deep contraction chunk for a block of rows is
// one contiguous run per plane (rows x 32 B), so the producer waves fetch it with fully coalesced
// 16-byte loads.  (Row-major planes [3][R][Cp] made every chunk load touch one 32-byte piece of `rows`
// different cache lines, and the next chunks re-touch those lines while they are still pending: the
// vector L1 spent ~70% of the short-contraction GEMMs stalled on pending lines.)  Run once per weight
// tensor per step.
// =====================================================================================
// src_t: the planes of the TRANSPOSE are wanted, w is stored [C][R] (the data gradient dY = W^T dZ takes the planes of
// pointwise.weight^T straight from the weight, without a transposed copy)
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ w, int R, int C, int Cp,
                                                      unsigned short* __restrict__ out, int bf16_only, int src_t) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)R * Cp) return;
    // consecutive threads walk the CONTIGUOUS dimension of the source (coalesced reads; the 2-byte scattered writes of
    // the transposed case go to a tensor of at most a few MB)
    const int r = src_t ? (int)(i % R) : (int)(i / Cp), c = src_t ? (int)(i / R) : (int)(i - (long)r * Cp);
    const float x = c < C ? (src_t ? w[(long)c * R + r] : w[(long)r * C + c]) : 0.f;
    const float p1 = bf16_only ? rne_bf16(x) : bitsf(fbits(x) & 0xFFFF0000u);
    const float r1 = bf16_only ? 0.f : x - p1;
    const float p2 = bitsf(fbits(r1) & 0xFFFF0000u);
    const float p3 = r1 - p2;
    // chunk-major image: element (chunk c / 16, plane t, row r, c % 16)
    const long o = ((long)(c >> 4) * 3 * R + r) * 16 + (c & 15);
    const long plane = (long)R * 16;
    out[o] = (unsigned short)(fbits(p1) >> 16);
    out[o + plane] = (unsigned short)(fbits(p2) >> 16);
    out[o + 2 * plane] = (unsigned short)(fbits(p3) >> 16);
}

int launch_split_planes(const float* w, int R, int C, unsigned short* out, hipStream_t st, int src_t) {
    const int Cp = (C + 15) & ~15;
    const long n = (long)R * Cp;
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, R, C, Cp, out,
                       split_mode() == 1 ? 1 : 0, src_t);
    return (int)hipGetLastError();
}

// ---- fp16 two-term weight images (NT == 2 GEMMs) ---------------------------------------------------------------------
// image of A * 2^kexp: [Cp/16][2][R][16] halves (h, g: A 2^kexp = h + g + O(2^-22)), followed by a trailer
//   { int kexp; 12 bytes pad; float pmax[npart] }   (smaat_split_planes_h_bytes)
// kexp = f16_kexp(max |A|): the maximum is taken in two levels without atomics or zero-filled state -- a first launch leaves
// the maxima of 4096-element pieces of the SOURCE in pmax, every block of the image launch reduces those (<= a few hundred
// values) for itself.  Bit-reproducible; the trailer's kexp is what the GEMM reads (PwSplitArgs::a_kexp).
#define HPIECE 4096
__host__ __device__ static inline long h_image_elems(int R, int Cp) { return (long)(Cp >> 4) * 2 * R * 16; }
__host__ __device__ static inline int h_npart(int R, int C) { return (int)(((long)R * C + HPIECE - 1) / HPIECE); }
long split_planes_h_bytes(int R, int C) {
    const int Cp = (C + 15) & ~15;
    return h_image_elems(R, Cp) * 2 + 16 + (long)h_npart(R, C) * 4;
}
long split_planes_h_kexp_offset(int R, int C) { return h_image_elems(R, (C + 15) & ~15) * 2; }  // bytes from the image start

__device__ __forceinline__ float block256_max(float m, float* red) {  // all 256 threads take part; result in every thread
    m = wave_max_all(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// piece `p` of a source of n floats -> pmax[p]
__device__ __forceinline__ void amax_piece(const float* __restrict__ w, long n, int p, float* __restrict__ pmax, float* red) {
    const long b = (long)p * HPIECE;
    float m = 0.f;
#pragma unroll 4
    for (int t = threadIdx.x; t < HPIECE; t += 256)
        if (b + t < n) m = fmaxf(m, fabsf(w[b + t]));
    m = block256_max(m, red);
    if (threadIdx.x == 0) pmax[p] = m;
}
// one element of the image; i = linear index of the [R][Cp] domain as in k_split_planes
__device__ __forceinline__ void h_image_elem(const float* __restrict__ w, int R, int C, int Cp, int src_t, long i, float sc,
                                             unsigned short* __restrict__ out) {
    const int r = src_t ? (int)(i % R) : (int)(i / Cp), c = src_t ? (int)(i / R) : (int)(i - (long)r * Cp);
    const float x = c < C ? (src_t ? w[(long)c * R + r] : w[(long)r * C + c]) : 0.f;
    unsigned h, g;
    split2_f16(x * sc, 0.f, h, g);
    const long o = ((long)(c >> 4) * 2 * R + r) * 16 + (c & 15);
    out[o] = (unsigned short)(h & 0xFFFFu);
    out[o + (long)R * 16] = (unsigned short)(g & 0xFFFFu);
}
__global__ __launch_bounds__(256) void k_amax_pieces(const float* __restrict__ w, long n, float* __restrict__ pmax) {
    __shared__ float red[4];
    amax_piece(w, n, blockIdx.x, pmax, red);
}
__global__ __launch_bounds__(256) void k_split_planes_h(const float* __restrict__ w, int R, int C, int Cp,
                                                        unsigned short* __restrict__ out, int src_t) {
    __shared__ float red[4];
    int* trailer = (int*)(out + h_image_elems(R, Cp));
    const float* pmax = (const float*)(trailer + 4);
    const int np = h_npart(R, C);
    float m = 0.f;
    for (int t = threadIdx.x; t < np; t += 256) m = fmaxf(m, pmax[t]);
    m = block256_max(m, red);
    const int kexp = f16_kexp(__builtin_bit_cast(unsigned, m));
    if (blockIdx.x == 0 && threadIdx.x == 0) trailer[0] = kexp;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)R * Cp) h_image_elem(w, R, C, Cp, src_t, i, pow2i(kexp), out);
}
int launch_split_planes_h(const float* w, int R, int C, unsigned short* out, int src_t, hipStream_t st) {
    const int Cp = (C + 15) & ~15;
    float* pmax = (float*)((int*)(out + h_image_elems(R, Cp)) + 4);
    hipLaunchKernelGGL(k_amax_pieces, dim3((unsigned)h_npart(R, C)), dim3(256), 0, st, w, (long)R * C, pmax);
    const long n = (long)R * Cp;
    hipLaunchKernelGGL(k_split_planes_h, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, R, C, Cp, out, src_t);
    return (int)hipGetLastError();
}
// first launch of a multi-image refresh that contains kind-3 rows: block b -> the b-th piece over all kind-3 matrices
__global__ __launch_bounds__(256) void k_weight_amax_multi(const long long* __restrict__ desc, int nd) {
    __shared__ float red[4];
    int cum = 0;
    for (int j = 0; j < nd; ++j) {  // (wave-uniform walk over <= a few dozen descriptors)
        const long long* d = desc + (long)j * 8;
        if ((int)d[4] != 3) continue;
        const int R = (int)d[2], C = (int)d[3], np = h_npart(R, C);
        if ((int)blockIdx.x < cum + np) {
            unsigned short* out = (unsigned short*)d[1];
            float* pmax = (float*)((int*)(out + h_image_elems(R, (C + 15) & ~15)) + 4);
            amax_piece((const float*)d[0], (long)R * C, (int)blockIdx.x - cum, pmax, red);
            return;
        }
        cum += np;
    }
}

// =====================================================================================
// k_weight_planes_multi: the operand images of MANY weight matrices in ONE launch (round 4).  A training step needs the
// images of 18 pointwise weights and of 17 transposes -- 35 launches of ~5 us each doing a few KB of work, one after the
// other in the stream.  All of them become stale at the same moment (the optimizer step), so the host refreshes every
// registered image at the first use after it (smaat_unet_amd/ops.py: _weight_planes).
// desc [n][8] int64 in device memory: { src, dst, R, C, kind, src_t, first block, blocks }
//   kind 0: split planes as k_split_planes (bf16_only = the current split mode, passed by the host)
//   kind 2: the single bf16 image of the mixed-precision kernels (k_bf16_planes in bf16gemm.hip: Cp = C rounded up to 32)
// The element arithmetic is that of the single-matrix kernels: bit-identical images.
// =====================================================================================
__global__ __launch_bounds__(256) void k_weight_planes_multi(const long long* __restrict__ desc, int nd, int bf16_only) {
    __shared__ int sel;
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < nd; j += 256) {
        const long long b0 = desc[j * 8 + 6], nb = desc[j * 8 + 7];
        if (b >= b0 && b < b0 + nb) sel = j;  // (block ranges are disjoint: one writer)
    }
    __syncthreads();
    const long long* d = desc + (long)sel * 8;
    const float* __restrict__ w = (const float*)d[0];
    unsigned short* __restrict__ out = (unsigned short*)d[1];
    const int R = (int)d[2], C = (int)d[3], kind = (int)d[4], src_t = (int)d[5];
    const int Cp = kind == 2 ? ((C + 31) & ~31) : ((C + 15) & ~15);
    const long i = (long)(b - (int)d[6]) * 256 + threadIdx.x;
    if (kind == 3) {  // fp16 two-term image (block-uniform branch: the whole block belongs to one matrix)
        __shared__ float red[4];
        int* trailer = (int*)(out + h_image_elems(R, Cp));
        const float* pmax = (const float*)(trailer + 4);
        const int np = h_npart(R, C);
        float m = 0.f;
        for (int t = threadIdx.x; t < np; t += 256) m = fmaxf(m, pmax[t]);
        m = block256_max(m, red);
        const int kexp = f16_kexp(__builtin_bit_cast(unsigned, m));
        if (b == (int)d[6] && threadIdx.x == 0) trailer[0] = kexp;
        if (i < (long)R * Cp) h_image_elem(w, R, C, Cp, src_t, i, pow2i(kexp), out);
        return;
    }
    if (i >= (long)R * Cp) return;
    const int r = src_t ? (int)(i % R) : (int)(i / Cp), c = src_t ? (int)(i / R) : (int)(i - (long)r * Cp);
    const float x = c < C ? (src_t ? w[(long)c * R + r] : w[(long)r * C + c]) : 0.f;
    if (kind == 2) {
        out[((long)(c >> 4) * R + r) * 16 + (c & 15)] = (unsigned short)(pack_bf16x2(x, 0.f) & 0xFFFFu);
        return;
    }
    const float p1 = bf16_only ? rne_bf16(x) : bitsf(fbits(x) & 0xFFFF0000u);
    const float r1 = bf16_only ? 0.f : x - p1;
    const float p2 = bitsf(fbits(r1) & 0xFFFF0000u);
    const float p3 = r1 - p2;
    const long o = ((long)(c >> 4) * 3 * R + r) * 16 + (c & 15);
    const long plane = (long)R * 16;
    out[o] = (unsigned short)(fbits(p1) >> 16);
    out[o + plane] = (unsigned short)(fbits(p2) >> 16);
    out[o + 2 * plane] = (unsigned short)(fbits(p3) >> 16);
}

int launch_weight_planes_multi(const long long* desc, int nd, int total_blocks, hipStream_t st, int h_pieces) {
    // h_pieces > 0: the table holds kind-3 rows with that many 4096-element pieces in total (maxima first)
    if (h_pieces > 0) hipLaunchKernelGGL(k_weight_amax_multi, dim3((unsigned)h_pieces), dim3(256), 0, st, desc, nd);
    hipLaunchKernelGGL(k_weight_planes_multi, dim3((unsigned)total_blocks), dim3(256), 0, st, desc, nd,
                       split_mode() == 1 ? 1 : 0);
    return (int)hipGetLastError();
}

// =====================================================================================
// k_pw_split:  out[n][m][p] = sum_c A[m][c] * x[n][c][p] + bias[m]   (+ BatchNorm partials)
//   A = pre-split planes, chunk-major [Cp/16][3][M][16] (k_split_planes); x split on the fly by the producer waves.
//   Used for the pointwise conv of the forward pass (x = depthwise output) and for its data gradient
//   (x = dZ, A = transposed weight).  Pixel tiles are PT consecutive pixels of the flattened plane.
// =====================================================================================


#define BROW 48   // bytes per LDS row of a [row][16 bf16] image (32 + 16 pad: conflict-free b128)

template <int WCO, int CT, int WPX, int PXT, int NPT, int NT>
__global__ __launch_bounds__(WCO * WPX * 64 + NPT) void k_pw_split(const PwSplitArgs a) {
    constexpr int COT = WCO * CT * 32;
    constexpr int PT = WPX * PXT * 32;
    constexpr int NCW = WCO * WPX;      // consumer waves (4 or 8)
    constexpr int NCT = NCW * 64;       // consumer threads
    constexpr int NTH = NCT + NPT;
    constexpr int APL = COT * BROW, BPL = PT * BROW;  // bytes per plane
    constexpr int BUFSZ = NT * (APL + BPL);  // NT planes of A then NT planes of B
    constexpr int NBT = PT * 2 / NPT;    // B tasks (pixel, k half) per producer thread
    constexpr int NAT = (COT * 2 * NT + NPT - 1) / NPT;  // A copy tasks per producer thread
    constexpr int NPL = NT == 2 ? 2 : 3;  // planes per chunk of the weight image (the bf16 image always has three)
    static_assert(NCW == 4 || NCW == 8, "4 or 8 consumer waves");
    static_assert((PT * 2) % NPT == 0, "producer mapping");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* stat = (float*)(lds + 2 * BUFSZ);  // [WPX][3][COT] + [8] wave pixel counts (BN_STAT_FLOATS)
    float* biasl = stat + BN_STAT_FLOATS(WPX, COT);  // [COT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= NCW;
    const int wave = producer ? 0 : wv;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int l31 = lane & 31, half = lane >> 5;
    const int ptid = producer ? tid - NCT : 0;

    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int cot = idx % a.nco;
    const int ptg = xcd * ((a.T + 7) >> 3) + idx / a.nco;  // contiguous tile range per XCD (L2 locality)
    if (ptg >= a.T) return;
    const int n = ptg / a.tiles_per_img, tl = ptg - n * a.tiles_per_img;
    const int co0 = cot * COT, p0 = tl * PT;
    const int nchunks = (a.Cin + 15) >> 4;
    const float* xn = a.x + (long)n * a.x_bs;
    int kx = 0, ka = 0;  // NT == 2: power-of-two scale exponents of x (from its maximum) and of the weight image
    if constexpr (NT == 2) {
        kx = f16_kexp(amax_read(a.x_amax));
        ka = *a.a_kexp;
        asm volatile("" : "+s"(kx), "+s"(ka));
    }
    const float sx = pow2i(kx);
    if (tid < COT) {  // bias through LDS (visible after the first barrier): a global load in the epilogue would
        const int m = co0 + tid;  // serialise the stores behind vmcnt(0)
        const float* bp = a.bias ? a.bias : (const float*)a.planes;
        const float v = bp[m < a.M ? m : 0];
        biasl[tid] = (a.bias && m < a.M) ? v : 0.f;
    }

    if (producer) {
        // B tasks: task u of this thread -> (pixel, k half); 8 channel values of one pixel per task
        int bpix[NBT], bhalf[NBT], bp[NBT];
        bool bv[NBT];
#pragma unroll
        for (int u = 0; u < NBT; ++u) {
            const int t = ptid + NPT * u;
            bpix[u] = t % PT;
            bhalf[u] = t / PT;
            const int p = p0 + bpix[u];
            bv[u] = p < a.P;
            bp[u] = bv[u] ? p : 0;
        }
        // A tasks: (plane, row, half)
        int arow[NAT], aofs[NAT];
        long asrc[NAT];
        bool av[NAT];
#pragma unroll
        for (int u = 0; u < NAT; ++u) {
            // surplus tasks wrap around and re-copy a valid piece: every task stores unconditionally (a
            // divergent branch around the store makes hipcc drain ALL outstanding loads with vmcnt(0))
            const int t = (ptid + NPT * u) % (COT * 2 * NT);
            const int pl = t / (COT * 2), rem = t - pl * (COT * 2);
            const int row = rem >> 1, h = rem & 1;
            arow[u] = row;
            av[u] = (co0 + row) < a.M;
            asrc[u] = ((long)pl * a.M + (av[u] ? co0 + row : 0)) * 16 + h * 8;  // + chunk * NPL * M * 16
            aofs[u] = pl * APL + row * BROW + h * 16;
        }
        // PD register sets: the global loads of PD chunks are in flight at any time (an iteration lasts
        // ~0.8-1.5k cycles, an L2/HBM round trip under load longer than that)
        constexpr int PD = 4;
        float breg[PD][NBT][8];
        uint4 areg[PD][NAT];
        auto prefetch = [&](int ch_, int set) {
            const int ch = ch_ < nchunks ? ch_ : nchunks - 1;
            const int k0 = ch * 16;
#pragma unroll
            for (int u = 0; u < NBT; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = k0 + bhalf[u] * 8 + e;
                    breg[set][u][e] = xn[(long)(c < a.Cin ? c : a.Cin - 1) * a.P + bp[u]];
                }
#pragma unroll
            for (int u = 0; u < NAT; ++u) areg[set][u] = *(const uint4*)(a.planes + asrc[u] + (long)k0 * NPL * a.M);
        };
        auto commit = [&](int ch_, int buf, int set) {
            const int ch = ch_ < nchunks ? ch_ : nchunks - 1;
            const int k0 = ch * 16;
            unsigned char* base = lds + buf * BUFSZ;
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                float xv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = k0 + bhalf[u] * 8 + e;
                    xv[e] = (bv[u] && c < a.Cin) ? breg[set][u][e] : 0.f;
                }
                split8_store<NT>(xv, sx, base + NT * APL + bpix[u] * BROW + bhalf[u] * 16, BPL);
            }
#pragma unroll
            for (int u = 0; u < NAT; ++u) {
                uint4 v = areg[set][u];
                v.x = av[u] ? v.x : 0u;
                v.y = av[u] ? v.y : 0u;
                v.z = av[u] ? v.z : 0u;
                v.w = av[u] ? v.w : 0u;
                *(uint4*)(base + aofs[u]) = v;
            }
        };
#pragma unroll
        for (int j = 0; j < PD; ++j) prefetch(j, j);
        commit(0, 0, 0);
        prefetch(PD, 0);
        __syncthreads();
        for (int i0 = 0; i0 < nchunks; i0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int i = i0 + u;
                if (i < nchunks) {
                    if (i + 1 < nchunks) {
                        commit(i + 1, (i + 1) & 1, (u + 1) % PD);   // loads issued PD iterations ago
                        prefetch(i + 1 + PD, (u + 1) % PD);
                    }
                    __syncthreads();
                }
            }
        }
    } else {
        f32x16 acc[CT][PXT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
        const int aoff = ((wco * CT) * 32 + l31) * BROW + half * 16;
        const int boff = NT * APL + ((wpx * PXT) * 32 + l31) * BROW + half * 16;
        auto load = [&](bf16x8 (&af)[CT][NT], bf16x8 (&bf)[PXT][NT], int buf) {
            const unsigned char* base = lds + buf * BUFSZ;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < NT; ++t) af[ct][t] = *(const bf16x8*)(base + aoff + t * APL + ct * 32 * BROW);
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                for (int t = 0; t < NT; ++t) bf[pt][t] = *(const bf16x8*)(base + boff + t * BPL + pt * 32 * BROW);
        };
        auto mma = [&](const bf16x8 (&af)[CT][NT], const bf16x8 (&bf)[PXT][NT]) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt) mma_terms<NT>(acc[ct][pt], af[ct], bf[pt]);
        };
        __syncthreads();
        for (int i = 0; i < nchunks; ++i) {
            bf16x8 af[CT][NT], bf[PXT][NT];
            load(af, bf, i & 1);
            mma(af, bf);
            __syncthreads();
        }
        if constexpr (NT == 2) {  // back to the operands' own scale (exact powers of two)
            const float e1 = pow2i(-((kx + ka) / 2)), e2 = pow2i(-((kx + ka) - (kx + ka) / 2));  // (exponent split evenly)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][pt][r] = acc[ct][pt][r] * e1 * e2;
        }
        // ---- epilogue: bias + coalesced row stores, BatchNorm partials of the raw accumulators ----
        int off[PXT];
#pragma unroll
        for (int pt = 0; pt < PXT; ++pt) {
            const int p = p0 + (wpx * PXT + pt) * 32 + l31;
            off[pt] = p < a.P ? p : -1;
        }
        float* obase = a.out + (long)n * a.out_bs;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (wco * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int m = co0 + col;
                if (m < a.M) {
                    const float bvv = biasl[col];
                    float* rowp = obase + (long)m * a.P;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        if (off[pt] >= 0) rowp[off[pt]] = floor_nan(acc[ct][pt][r] + bvv, a.out_floor);
                }
            }
        }
        if (a.part) {
            bool pval[PXT];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) pval[pt] = off[pt] >= 0;
            // flattened tiles: the valid pixels of the wave are a prefix of its PXT x 32 pixels
            int nw = a.P - (p0 + wpx * PXT * 32);
            nw = nw < 0 ? 0 : (nw > PXT * 32 ? PXT * 32 : nw);
            bn_wave_partials<CT, PXT>(acc, pval, l31, half, stat + wpx * 3 * COT + wco * CT * 32, COT, 0);
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
                a.part[((long)0 * a.slots + ptg) * a.M + m] = mean;
                a.part[((long)1 * a.slots + ptg) * a.M + m] = m2;
                a.part[((long)2 * a.slots + ptg) * a.M + m] = cnt;
            }
        }
    }
}


// -------------------------------------------------------------------------------------
// k_pw_split_p: the same GEMM as k_pw_split, PERSISTENT over (pixel tile, channel tile) items.
// A workgroup walks items  idx0, idx0 + gstep, ...  of its XCD; the producer waves run a single
// flattened chunk stream across item boundaries, so the global loads of the next PD chunks (= the whole
// next tile when the contraction is short) are in flight while the consumer waves run the epilogue of
// the current tile.  For the short-contraction GEMMs (data gradients of the plane-dominated layers:
// 4-8 chunks per tile, 64 KB of output per tile) this is what keeps HBM busy: in the one-tile-per-
// workgroup form fill, MFMAs and the store epilogue of a tile are serial and only two workgroups per
// CU overlap them.  Out-of-range rows/pixels are CLAMPED to valid addresses rather than zeroed: their
// accumulator rows/columns are never stored and are masked out of the statistics.
// BatchNorm partials: each consumer wave leaves its per-channel sums in stat[item parity]; they are
// combined and written one item later (all waves have passed a barrier in between).
// -------------------------------------------------------------------------------------
template <int WCO, int CT, int WPX, int PXT, int NPT, int NT>
__global__ __launch_bounds__(WCO * WPX * 64 + NPT, (WCO * WPX == 4) ? 4 : 2) void k_pw_split_p(const PwSplitArgs a) {
    constexpr int COT = WCO * CT * 32;
    constexpr int PT = WPX * PXT * 32;
    constexpr int NCW = WCO * WPX;
    constexpr int NCT = NCW * 64;
    constexpr int APL = COT * BROW, BPL = PT * BROW;
    constexpr int BUFSZ = NT * (APL + BPL);
    constexpr int NBT = PT * 2 / NPT;
    constexpr int NAT = (COT * 2 * NT + NPT - 1) / NPT;
    constexpr int NPL = NT == 2 ? 2 : 3;  // planes per chunk of the weight image
    static_assert(NCW == 4 || NCW == 8, "4 or 8 consumer waves");
    static_assert((PT * 2) % NPT == 0, "producer mapping");
    static_assert(PT % 64 == 0 && NPT % 64 == 0, "the channel half of a producer thread's B task is wave-uniform");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // LDS budget: two workgroups per CU need <= 80 KB each.  The 128 x 128 configuration is at 73.7 KB of operand
    // buffers + 2 KB of bias slots; the statistics area is [2][WPX][3][COT] floats = 6 KB and nothing else (the wave
    // pixel counts are recomputed from the tile index instead of being exchanged through LDS: with them the block
    // was 64 bytes over the limit and ran alone on its CU), and it is only allocated when partials are requested.
    constexpr int STSZ = WPX * 3 * COT;
    float* stat = (float*)(lds + 2 * BUFSZ);  // [2][STSZ]: per item parity, [WPX][3][COT] wave partials
    float* biasl = stat + (a.part ? 2 * STSZ : 0);  // [4][COT]: bias of the channel tile of item k in slot k & 3

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= NCW;
    const int wave = producer ? 0 : wv;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int l31 = lane & 31, half = lane >> 5;
    const int ptid = producer ? tid - NCT : 0;

    // items of this workgroup: idx = idx0 + k * gstep on XCD xcd; idx -> (pixel tile idx / nco of the XCD's
    // contiguous tile range, channel tile idx % nco)
    const int b = blockIdx.x, xcd = b & 7, idx0 = b >> 3;
    const int gstep = gridDim.x >> 3;
    const int tpx = (a.T + 7) >> 3;
    int lim_t = a.T - xcd * tpx;
    lim_t = lim_t < tpx ? lim_t : tpx;
    const int lim = lim_t > 0 ? lim_t * a.nco : 0;
    if (idx0 >= lim) return;
    const int nitems = (lim - 1 - idx0) / gstep + 1;
    const int nchunks = (a.Cin + 15) >> 4;
    const int G = nitems * nchunks;  // chunks of this workgroup
    constexpr int PD = 4;                       // register sets of loads in flight per producer thread
    const int G_pad = (G + PD - 1) / PD * PD;   // barriers of the walk (producers and consumers alike)
    int kx = 0, ka = 0;  // NT == 2: power-of-two scale exponents of x (from its maximum) and of the weight image
    if constexpr (NT == 2) {
        kx = f16_kexp(amax_read(a.x_amax));
        ka = *a.a_kexp;
        asm volatile("" : "+s"(kx), "+s"(ka));  // (waited for here, not inside the chunk loop)
    }

    if (producer) {
        int bpix[NBT], bhalf[NBT];
#pragma unroll
        for (int u = 0; u < NBT; ++u) {
            const int t = ptid + NPT * u;
            bpix[u] = t % PT;
            bhalf[u] = t / PT;
        }
        // A tasks: (plane, row, k half).  When a plane is exactly NPT tasks, task u of a thread is plane u of
        // ONE (row, half): the per-task state collapses to a base + compile-time offsets.
        constexpr bool ASIMPLE = (COT * 2 == NPT);
        int apl[NAT], arow[NAT], ah8[NAT], aofs[NAT];
#pragma unroll
        for (int u = 0; u < NAT; ++u) {
            const int t = ASIMPLE ? ptid + NPT * u : (ptid + NPT * u) % (COT * 2 * NT);  // surplus tasks re-copy a valid piece
            const int pl = ASIMPLE ? u : t / (COT * 2), rem = ASIMPLE ? ptid : t - pl * (COT * 2);
            // (row, half) of a task: consecutive lanes take consecutive ROWS of one half.  With 48-byte rows the 16 lanes of
            // a ds_write_b128 group then hit 16 disjoint 4-bank windows; the earlier map (lanes 2i, 2i + 1 = the two halves
            // of row i) put rows i and i + 5 on the same banks: SQ_LDS_BANK_CONFLICT was 19 % of the LDS-active cycles of
            // this kernel (profiles/r3/pmc_lds_f32_before.txt), all of it from these writes.
            apl[u] = pl;
            arow[u] = rem % COT;
            ah8[u] = (rem / COT) * 8;
            aofs[u] = pl * APL + arow[u] * BROW + (rem / COT) * 16;
        }
        const float sx = pow2i(kx);
        float breg[PD][NBT][8];
        u32x4 areg[PD][NAT];
        float bias_reg[PD];  // bias of channel (ptid % COT) of the chunk's item: reaches the consumers through LDS
        const float* biasp = a.bias ? a.bias : (const float*)a.planes;  // any readable address when there is no bias  // native vector: a struct copy global -> private -> LDS would stay a memcpy through scratch
        // prefetch cursor: (item, chunk) + the per-item address bases
        int pf_item = 0, pf_ch = 0;
        const float* pf_x = a.x;
        const unsigned short* pf_pl = a.planes;  // weight planes of the item's K slice
        int pf_bp[NBT], pf_bidx = 0;
        unsigned pf_bvo[NBT], pf_avo[NAT];  // per-lane byte offsets of the scalar-base (saddr) loads
        long pf_as[NAT];
        auto pf_setup = [&]() __attribute__((always_inline)) {
            const int it = pf_item < nitems ? pf_item : nitems - 1;
            const int idx = idx0 + it * gstep;
            const int j = idx / a.nco, cot = idx - j * a.nco;
            const int ptg = xcd * tpx + j;
            const int n = ptg / a.tiles_per_img, tl = ptg - n * a.tiles_per_img;
            pf_x = a.x + (long)n * a.x_bs;
            pf_pl = a.planes + (a.ksplit > 1 ? (long)(n % a.ksplit) * a.planes_bs : 0L);
            pf_bidx = cot * COT + ptid % COT;
            pf_bidx = pf_bidx < a.M ? pf_bidx : a.M - 1;
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                const int p = tl * PT + bpix[u];
                pf_bp[u] = p < a.P ? p : a.P - 1;
                pf_bvo[u] = (unsigned)pf_bp[u] * 4u;  // (the half's 8 channels are part of the scalar base)
            }
#pragma unroll
            for (int u = 0; u < NAT; ++u) {
                const int r = cot * COT + arow[ASIMPLE ? 0 : u];
                const long o = (long)(r < a.M ? r : a.M - 1) * 16 + ah8[ASIMPLE ? 0 : u];
                pf_as[u] = ASIMPLE ? o : o + (long)apl[u] * a.M * 16;
                pf_avo[u] = (unsigned)pf_as[u] * 2u;
            }
        };
        const long aplane = (long)a.M * 16;  // planes are chunk-major [Cp/16][3][M][16]
        // The global loads of the producers are issued through inline asm and waited for with an explicit
        // counted s_waitcnt: hipcc's own bookkeeping drains to vmcnt(0) at every commit of this loop (the
        // conditionals and the rotating register sets defeat its counting), which caps the prefetch depth at
        // ONE chunk however many register sets exist.  With LPC loads per chunk and PD sets, the loads of
        // the chunk being committed are complete when at most (PD - 1) * LPC younger ones are outstanding.
        constexpr int LPC = NBT * 8 + NAT + 1;
        static_assert((PD - 1) * LPC <= 63, "vmcnt is a 6-bit counter");
        // Addresses: a wave-uniform 64-bit base in SGPRs (image, chunk, channel: scalar arithmetic) plus a 32-bit
        // per-lane byte offset that only changes with the item -> no vector address arithmetic per load.
        auto prefetch = [&](int set) __attribute__((always_inline)) {
            const int k0 = pf_ch * 16;
            // ONE form for full and partial chunks (round 6): the channel of element e is clamped in scalar arithmetic -- the
            // half a producer thread serves is wave-uniform (PT is a multiple of 64) -- and the values of the channels past Cin
            // are zeroed at the commit.  The earlier two-armed form (scalar bases | per-lane clamped addresses) came out of hipcc
            // as two flag-guarded blocks, i.e. a graph with paths that issue both or neither: the load count per prefetch must
            // be the same on every path for the counted waits to be provable (scripts/isa_hazards.py).
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                const int hb = k0 + __builtin_amdgcn_readfirstlane(bhalf[u]) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = hb + e < a.Cin ? hb + e : a.Cin - 1;
                    const float* sbe = pf_x + (long)c * a.P;
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(breg[set][u][e]) : "v"(pf_bvo[u]), "s"(sbe));
                }
            }
            {
                const unsigned short* sa = pf_pl + (long)k0 * NPL * a.M;
#pragma unroll
                for (int u = 0; u < NAT; ++u) {
                    const unsigned short* sau = sa + (ASIMPLE ? u * aplane : 0);
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(areg[set][u]) : "v"(pf_avo[ASIMPLE ? 0 : u]), "s"(sau));
                }
                const unsigned bvo = (unsigned)pf_bidx * 4u;
                asm volatile("global_load_dword %0, %1, %2" : "=v"(bias_reg[set]) : "v"(bvo), "s"(biasp));
            }
            if (++pf_ch == nchunks) {  // wave-uniform: address arithmetic only
                pf_ch = 0;
                ++pf_item;
                pf_setup();
            }
        };
        auto commit = [&](int ch, int buf, int set, int slot, bool live) __attribute__((always_inline)) {
            const int k0 = ch * 16;
            // wait for this set's loads; the "+v" ties keep every use of the set's registers behind the wait
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * LPC) : "memory");
#pragma unroll
            for (int u = 0; u < NBT; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(breg[set][u][e]));
#pragma unroll
            for (int u = 0; u < NAT; ++u) asm volatile("" : "+v"(areg[set][u]));
            asm volatile("" : "+v"(bias_reg[set]));
            if (!live) return;  // surplus iteration of the padded walk: waited for, not written
            const bool partial = k0 + 16 > a.Cin;  // wave-uniform
            biasl[slot * COT + ptid % COT] = a.bias ? bias_reg[set] : 0.f;  // every chunk of the item rewrites the same values
            unsigned char* base = lds + buf * BUFSZ;
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                float xv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = breg[set][u][e];
                    if (partial) x = (k0 + bhalf[u] * 8 + e < a.Cin) ? x : 0.f;
                    xv[e] = x;
                }
                split8_store<NT>(xv, sx, base + NT * APL + bpix[u] * BROW + bhalf[u] * 16, BPL);
            }
#pragma unroll
            for (int u = 0; u < NAT; ++u) *(u32x4*)(base + (ASIMPLE ? aofs[0] + u * APL : aofs[u])) = areg[set][u];
        };
        pf_setup();
#pragma unroll
        for (int j = 0; j < PD; ++j) prefetch(j);
        commit(0, 0, 0, 0, true);
        prefetch(0);
        int cm_ch = nchunks > 1 ? 1 : 0;  // chunk (within its item) of the NEXT commit
        int cm_slot = nchunks > 1 ? 0 : 1;  // ... and its item & 3
        __syncthreads();
        // padded walk, every slot unconditional (k_wgrad_split above: the shape scripts/isa_hazards.py can prove)
        for (int g0 = 0; g0 < G_pad; g0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                commit(cm_ch, (g0 + u + 1) & 1, (u + 1) % PD, cm_slot, g0 + u + 1 < G);  // loads issued PD chunks ago
                if (++cm_ch == nchunks) {
                    cm_ch = 0;
                    cm_slot = (cm_slot + 1) & 3;
                }
                prefetch((u + 1) % PD);
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus prefetches of the tail still target live registers
        if (a.part) __syncthreads();
    } else {
        const int aoff = ((wco * CT) * 32 + l31) * BROW + half * 16;
        const int boff = NT * APL + ((wpx * PXT) * 32 + l31) * BROW + half * 16;
        auto flush = [&](int par, int ptg, int co0) __attribute__((always_inline)) {  // merge the WPX waves' partials of a finished item
            const float* sp = stat + par * STSZ;
            int nwv[WPX];  // valid pixels per wave column of that tile (flattened tiles: a prefix)
            {
                const int tl_ = ptg % a.tiles_per_img;
#pragma unroll
                for (int w = 0; w < WPX; ++w) {
                    const int v = a.P - (tl_ * PT + w * PXT * 32);
                    nwv[w] = v < 0 ? 0 : (v > PXT * 32 ? PXT * 32 : v);
                }
            }
            for (int col = tid; col < COT; col += NCT) {
                float mean, m2, cnt;
                bn_tile_combine<WPX>(sp, nwv, COT, col, mean, m2, cnt);
                const int m = co0 + col;
                if (m < a.M) {
                    a.part[((long)0 * a.slots + ptg) * a.M + m] = mean;
                    a.part[((long)1 * a.slots + ptg) * a.M + m] = m2;
                    a.part[((long)2 * a.slots + ptg) * a.M + m] = cnt;
                }
            }
        };
        __syncthreads();
        int g = 0, prev_ptg = 0, prev_co0 = 0;
        for (int k = 0; k < nitems; ++k) {
            const int idx = idx0 + k * gstep;
            const int j = idx / a.nco, cot = idx - j * a.nco;
            const int ptg = xcd * tpx + j;
            const int n = ptg / a.tiles_per_img, tl = ptg - n * a.tiles_per_img;
            const int co0 = cot * COT, p0 = tl * PT;
            f32x16 acc[CT][PXT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;
            for (int i = 0; i < nchunks; ++i, ++g) {
                const unsigned char* base = lds + (g & 1) * BUFSZ;
                bf16x8 af[CT][NT], bf[PXT][NT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int t = 0; t < NT; ++t) af[ct][t] = *(const bf16x8*)(base + aoff + t * APL + ct * 32 * BROW);
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) bf[pt][t] = *(const bf16x8*)(base + boff + t * BPL + pt * 32 * BROW);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt) mma_terms<NT>(acc[ct][pt], af[ct], bf[pt]);
                __syncthreads();
            }
            // ---- epilogue (no barriers: the producers keep streaming the next item meanwhile) ----
            // Branch-free and load-free: a global load (bias) or a divergent branch in here makes hipcc put an
            // s_waitcnt vmcnt(0) in front of every store, i.e. each store waits for the previous one to be
            // acknowledged (that, not bandwidth, was the cost of the store tail).  Stores go through a buffer
            // descriptor of the output image: rows >= M fall beyond num_records and pixels >= P get an offset
            // with bit 31 set, so the hardware range check drops them.
            if (a.part && k > 0) flush((k - 1) & 1, prev_ptg, prev_co0);
            if constexpr (NT == 2) {  // back to the operands' own scale (exact powers of two; two EVEN factors: 2^-(kx+ka) may
                const float e1 = pow2i(-((kx + ka) / 2)), e2 = pow2i(-((kx + ka) - (kx + ka) / 2));  // be out of range
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ct][pt][r] = acc[ct][pt][r] * e1 * e2;
            }
            unsigned pvo[PXT];
            bool pval[PXT];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) {
                const int p = p0 + (wpx * PXT + pt) * 32 + l31;
                pval[pt] = p < a.P;
                pvo[pt] = pval[pt] ? (unsigned)p * 4u : 0x80000000u;
            }
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(a.out + (long)n * a.out_bs, 0, a.M * a.P * 4, 0x00020000);
            const unsigned rowb = (unsigned)(co0 + wco * CT * 32 + 4 * half) * (unsigned)a.P * 4u;
            const float* bl = biasl + (k & 3) * COT + wco * CT * 32 + 4 * half;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rc = ct * 32 + (r & 3) + 8 * (r >> 2);  // row of the wave tile (+ 4 * half): constant
                    const float bvv = bl[rc];
                    const unsigned ro = rowb + (unsigned)rc * (unsigned)a.P * 4u;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, floor_nan(acc[ct][pt][r] + bvv, a.out_floor)), rs,
                                                              ro + pvo[pt], 0, 0);
                }
            }
            if (a.part) {
                float* sb = stat + (k & 1) * STSZ;
                bn_wave_partials<CT, PXT>(acc, pval, l31, half, sb + wpx * 3 * COT + wco * CT * 32, COT, 0);
                prev_ptg = ptg;
                prev_co0 = co0;
            }
        }
        for (int gg = G; gg < G_pad; ++gg) __syncthreads();  // the producers' surplus iterations
        if (a.part) {
            __syncthreads();
            flush((nitems - 1) & 1, prev_ptg, prev_co0);
        }
    }
}

int pw_split_num_slots(int N, int P) {
    // upper bound over the tile choices of launch_pw_split (unused slots are never written: see below)
    const int PT = 128;
    return N * ((P + PT - 1) / PT);
}

template <int WCO, int CT, int WPX, int PXT, int NPT, int NT = 3>
static int launch_pw_split_cfg(PwSplitArgs& a, hipStream_t st) {
    constexpr int COT = WCO * CT * 32, PT = WPX * PXT * 32;
    a.nco = (a.M + COT - 1) / COT;
    a.tiles_per_img = (a.P + PT - 1) / PT;
    a.T = a.N * a.tiles_per_img;
    a.slots = pw_split_num_slots(a.N, a.P);
    a.dbg = 0;
    const int items = ((a.T + 7) / 8) * 8 * a.nco;
    // persistent form: 2 workgroups per CU walk the items.  Its store tail addresses an output image through a
    // 32-bit buffer offset with bit 31 as the "dropped" marker: images of 2 GiB and more (and SMAAT_PWS_CFG=16,
    // for A/B timing) take the one-tile-per-workgroup kernel.
    if (!(pws_cfg() & 16) && (long)(a.M + COT) * a.P * 4 < (1L << 31)) {
        static_assert((size_t)2 * NT * (COT + PT) * BROW + sizeof(float) * (2 * WPX * 3 * COT + 4 * COT) <= 80 * 1024,
                      "k_pw_split_p: two workgroups per CU (160 KB of LDS) -- round 2 lost 1.5 ms per step to 64 bytes");
        const size_t lds = (size_t)2 * NT * (COT + PT) * BROW + sizeof(float) * ((a.part ? 2 * WPX * 3 * COT : 0) + 4 * COT);
        constexpr auto kern = k_pw_split_p<WCO, CT, WPX, PXT, NPT, NT>;
        int rc = ensure_lds_s<kern>(lds);
        if (rc) return rc;
        const int grid = items < 512 ? items : 512;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WCO * WPX * 64 + NPT), lds, st, a);
    } else {
        const size_t lds = (size_t)2 * NT * (COT + PT) * BROW + sizeof(float) * (BN_STAT_FLOATS(WPX, COT) + COT);
        constexpr auto kern = k_pw_split<WCO, CT, WPX, PXT, NPT, NT>;
        int rc = ensure_lds_s<kern>(lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(items), dim3(WCO * WPX * 64 + NPT), lds, st, a);
    }
    if (a.part && a.T < a.slots) {  // partial-statistics rows this tile choice does not use
        for (int w = 0; w < 3; ++w)
            HIP_RET(hipMemsetAsync(a.part + ((long)w * a.slots + a.T) * a.M, 0, sizeof(float) * (size_t)(a.slots - a.T) * a.M, st));
    }
    return (int)hipGetLastError();
}

static int pws_cfg() {  // SMAAT_PWS_CFG: tuning experiments (0 = default)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SMAAT_PWS_CFG");
        v = e ? atoi(e) : 0;
    }
    return v;
}

// 1 when launch_pw_split takes the persistent kernel for this shape (the only one that knows about K slices)
int pws_persistent_ok(int M, int P) {
    const int COT = M > 64 ? 128 : 64;
    return (!(pws_cfg() & 16) && (long)(M + COT) * P * 4 < (1L << 31)) ? 1 : 0;
}

int launch_pw_split(PwSplitArgs& a, hipStream_t st) {
    // 128-pixel tiles everywhere: two workgroups per CU overlap each other's fill/drain and barriers
    // (measured faster than 256-pixel tiles with one workgroup per CU, than 256 x 128 tiles with eight
    // consumer waves, and than three LDS buffers with a fragment prefetch, on every layer shape)
    if (split_mode() == 1) {  // plain bf16 operands, one MFMA per product
        if (a.M > 64) return launch_pw_split_cfg<2, 2, 2, 2, 256, 1>(a, st);
        return launch_pw_split_cfg<1, 2, 4, 1, 256, 1>(a, st);
    }
    if (a.x_amax || a.a_kexp) {  // two-term fp16 split: an fp16 weight image + the maximum of x
        if (!a.x_amax || !a.a_kexp) return -1;
        if (a.M > 64) return launch_pw_split_cfg<2, 2, 2, 2, 256, 2>(a, st);
        return launch_pw_split_cfg<1, 2, 4, 1, 256, 2>(a, st);
    }
    if (a.M > 64) return launch_pw_split_cfg<2, 2, 2, 2, 256>(a, st);  // 128 x 128
    return launch_pw_split_cfg<1, 2, 4, 1, 256>(a, st);                // 64 x 128
}

// ---- split-K for the inference GEMMs at small batch --------------------------------------------------------------
// A deep layer at batch 1 has a few dozen (pixel tile, channel tile) items, each a serial chain of up to 128
// contraction chunks: the chip idles and the kernel time is the chain's latency (batch-1 trace: 10 launches of 37 us).
// The contraction is cut into S slices that run as S "virtual images" of the same persistent kernel (x slice = a
// channel offset, weight-plane slice = planes_bs) into a partial buffer [N][S][M][P]; this kernel adds the slices in a
// fixed order, the bias and the ReLU.
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                       float* __restrict__ out, long out_bs, int M, int P, float out_floor,
                                                       long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const long per_img = (long)M * P / 4;
    const int n = (int)(i / per_img);
    const long e = (i - (long)n * per_img) * 4;  // element of the [M][P] image
    const int m = (int)(e / P);
    const float* w0 = ws + ((long)n * S) * M * P + e;
    float4 acc = *(const float4*)w0;
    for (int s = 1; s < S; ++s) {
        const float4 v = *(const float4*)(w0 + (long)s * M * P);
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
    }
    const float b = bias ? bias[m] : 0.f;
    acc.x = floor_nan(acc.x + b, out_floor);
    acc.y = floor_nan(acc.y + b, out_floor);
    acc.z = floor_nan(acc.z + b, out_floor);
    acc.w = floor_nan(acc.w + b, out_floor);
    *(float4*)(out + (long)n * out_bs + e) = acc;
}

// number of K slices for a [N][Cin][P] -> [N][M][P] GEMM (1 = no split): fill ~`budget` workgroup slots (512 = the
// persistent kernel's grid: inference at small batch), keep >= 8 chunks of 16 channels per slice, at most 8 slices.
// Training passes a larger budget for the layers that leave the chip under-filled (18 x 18 planes at batch 32: 384
// items, each a serial chain of 64 chunks): a few waves of short chains beat one wave of long ones.
int pw_splitk_slices(int N, int Cin, int M, int P, int budget) {
    if ((Cin & 15) != 0 || (P & 3) != 0) return 1;
    const int cot = M > 64 ? 128 : 64;
    const long items = (long)N * ((P + 127) / 128) * ((M + cot - 1) / cot);
    if (items >= 512 && budget > 512) return 1;  // the chip is full without slicing
    int s = 1;
    while (s < 8 && items * (s * 2) <= budget && (Cin / 16) % (s * 2) == 0 && Cin / 16 / (s * 2) >= 8) s *= 2;
    return s;
}

// the slice reduction of a TRAINING forward GEMM: one workgroup per (image, output channel) plane adds the S partial
// planes in a fixed order, writes z = max(sum + bias, floor) and the plane's BatchNorm partial (mean, M2, count) of the
// raw sums (without the bias, the contract of the GEMM epilogues) about a shift that is a sample of the plane.
// part [3][slots][M], slot = n * tiles_per_img (the image's other slots get count 0).
__global__ __launch_bounds__(256) void k_splitk_reduce_stats(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                             float* __restrict__ out, long out_bs, int M, int P,
                                                             float out_floor, float* __restrict__ part, int slots,
                                                             int tiles_per_img) {
    __shared__ float red[9];
    const int plane = blockIdx.x, n = plane / M, m = plane - n * M;
    const float* w0 = ws + ((long)n * S * M + m) * P;
    const long sstride = (long)M * P;
    float* op = out + (long)n * out_bs + (long)m * P;
    const float b = bias ? bias[m] : 0.f;
    if (threadIdx.x == 0) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += w0[s * sstride];
        red[8] = v;
    }
    __syncthreads();
    const float sh = red[8];
    float s1 = 0.f, s2 = 0.f;
    for (int p = threadIdx.x * 4; p < P; p += 1024) {  // P % 4 == 0
        float4 acc = *(const float4*)(w0 + p);
        for (int s = 1; s < S; ++s) {
            const float4 v = *(const float4*)(w0 + s * sstride + p);
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
        }
        const float d0 = acc.x - sh, d1 = acc.y - sh, d2 = acc.z - sh, d3 = acc.w - sh;
        s1 += (d0 + d1) + (d2 + d3);
        s2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, s2))));
        *(float4*)(op + p) = make_float4(floor_nan(acc.x + b, out_floor), floor_nan(acc.y + b, out_floor), floor_nan(acc.z + b, out_floor),
                                         floor_nan(acc.w + b, out_floor));
    }
    const float t1 = block_sum_t0(s1, red);
    const float t2 = block_sum_t0(s2, red + 4);
    if (threadIdx.x == 0) {
        const float cnt = (float)P, mean = sh + t1 / cnt;
        const float m2 = fmaxf(t2 - t1 * t1 / cnt, 0.f);
        const long slot = (long)n * tiles_per_img;
        part[(0L * slots + slot) * M + m] = mean;
        part[(1L * slots + slot) * M + m] = m2;
        part[(2L * slots + slot) * M + m] = cnt;
        for (int t = 1; t < tiles_per_img; ++t) {
            part[(0L * slots + slot + t) * M + m] = 0.f;
            part[(1L * slots + slot + t) * M + m] = 0.f;
            part[(2L * slots + slot + t) * M + m] = 0.f;
        }
    }
}

int launch_pw_split_k(PwSplitArgs& a, float* ws, int S, hipStream_t st) {
    // a: the un-split problem (N, Cin, dense x: x_bs == Cin * P); out / out_bs / bias / out_floor of the final result;
    // a.part (nullable): BatchNorm partials of the final result, [3][pw_split_num_slots(N, P)][M]
    float* out = a.out;
    float* part = a.part;
    const long out_bs = a.out_bs;
    const float* bias = a.bias;
    const float floor_ = a.out_floor;
    const int N = a.N;
    a.ksplit = S;
    a.planes_bs = (long)(a.Cin / S / 16) * (a.x_amax ? 2 : 3) * a.M * 16;
    a.x_bs = (long)(a.Cin / S) * a.P;
    a.Cin = a.Cin / S;
    a.Cp = a.Cin;
    a.N = N * S;
    a.out = ws;
    a.out_bs = (long)a.M * a.P;
    a.bias = nullptr;
    a.part = nullptr;
    a.out_floor = -__builtin_inff();
    int rc = launch_pw_split(a, st);
    if (rc) return rc;
    if (part) {
        hipLaunchKernelGGL(k_splitk_reduce_stats, dim3((unsigned)(N * a.M)), dim3(256), 0, st, ws, S, bias, out, out_bs, a.M, a.P,
                           floor_, part, pw_split_num_slots(N, a.P), (a.P + 127) / 128);
        return (int)hipGetLastError();
    }
    const long total4 = (long)N * a.M * a.P / 4;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, ws, S, bias, out, out_bs,
                       a.M, a.P, floor_, total4);
    return (int)hipGetLastError();
}
