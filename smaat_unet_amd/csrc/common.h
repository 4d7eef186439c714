// Shared device helpers for the SmaAt-UNet gfx950 kernels.
// Written for CDNA4 only: 64-lane wavefronts, DPP row ops, f32 MFMA 32x32x2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMAAT_THREADS 256

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- element types of the activation tensors ------------------------------------------------------------------
// f32 (default) or bf16 storage (mixed precision: BASELINE configs[3] -- activations and their gradients 2 bytes per
// element in HBM, f32 arithmetic / accumulation / BatchNorm statistics in registers, f32 master weights).  A kernel
// that exists for both is a template over the element type of each tensor and touches memory only through these
// accessors, so the f32 instantiation is the code it was before.  SMAAT_F32 / SMAAT_BF16 are the dtype codes of the C ABI.
#define SMAAT_F32 0
#define SMAAT_BF16 1
typedef unsigned short bf16_t;  // bit pattern of a bf16 value
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2_native v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_native));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    typedef float4 raw4;    // registers of one 4-element load
    typedef float raw1;
    static constexpr int code = SMAAT_F32;
    static constexpr unsigned vmask = 15;  // byte alignment mask of a 4-element access
    static constexpr int bytes = 4;
};
template <> struct Elem<bf16_t> {
    typedef uint2 raw4;
    typedef unsigned raw1;
    static constexpr int code = SMAAT_BF16;
    static constexpr unsigned vmask = 7;
    static constexpr int bytes = 2;
};
// SMAAT_NT (compile-time, experiment builds: scripts/build_nt_libs.sh): bit 0 = the 4-element streaming loads below carry the
// non-temporal hint (`global_load_dwordx4 ... nt`), bit 1 = the 4-element streaming stores do
#ifndef SMAAT_NT
#define SMAAT_NT 0
#endif
typedef float smaat_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned smaat_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldraw4(const float* p) {
#if SMAAT_NT & 1
    const smaat_f32x4 v = __builtin_nontemporal_load((const smaat_f32x4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *(const float4*)p;
#endif
}
__device__ __forceinline__ uint2 ldraw4(const bf16_t* p) {
#if SMAAT_NT & 1
    const smaat_u32x2 v = __builtin_nontemporal_load((const smaat_u32x2*)p);
    return make_uint2(v.x, v.y);
#else
    return *(const uint2*)p;
#endif
}
__device__ __forceinline__ float ldraw1(const float* p) { return *p; }
__device__ __forceinline__ unsigned ldraw1(const bf16_t* p) { return *p; }
__device__ __forceinline__ float4 cvt4(const float4 v) { return v; }
__device__ __forceinline__ float4 cvt4(const uint2 v) { return make_float4(bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y)); }
__device__ __forceinline__ float cvt1(const float v) { return v; }
__device__ __forceinline__ float cvt1(const unsigned v) { return bf16_lo(v); }
template <typename T> __device__ __forceinline__ float4 ld4(const T* p) { return cvt4(ldraw4(p)); }
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return cvt1(ldraw1(p)); }
__device__ __forceinline__ void st4(float* p, const float4 v) {
#if SMAAT_NT & 2
    const smaat_f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (smaat_f32x4*)p);
#else
    *(float4*)p = v;
#endif
}
__device__ __forceinline__ void st4(bf16_t* p, const float4 v) {
#if SMAAT_NT & 2
    const smaat_u32x2 t = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    __builtin_nontemporal_store(t, (smaat_u32x2*)p);
#else
    *(uint2*)p = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
#endif
}
__device__ __forceinline__ void st1(float* p, const float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, const float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xFFFFu); }
__device__ __forceinline__ void st2(float* p, const float a, const float b) { *(float2*)p = make_float2(a, b); }
__device__ __forceinline__ void st2(bf16_t* p, const float a, const float b) { *(unsigned*)p = pack_bf16x2(a, b); }
__device__ __forceinline__ float2 ld2(const float* p) { return *(const float2*)p; }
__device__ __forceinline__ float2 ld2(const bf16_t* p) {
    const unsigned u = *(const unsigned*)p;
    return make_float2(bf16_lo(u), bf16_hi(u));
}
// the value a store of v to a tensor of type T leaves there (so that a kernel which both stores and keeps using a value
// -- deferred activation + pooling -- sees what its consumers will read)
__device__ __forceinline__ float as_stored(const float*, float v) { return v; }
__device__ __forceinline__ float as_stored(const bf16_t*, float v) { return bf16_lo(pack_bf16x2(v, 0.f)); }
// host side: run `body` with ET bound to the element type of dtype code `dt`
#define SMAAT_DISPATCH_ET(dt, ET, ...)                     \
    do {                                                   \
        if ((dt) == SMAAT_BF16) {                          \
            typedef bf16_t ET;                             \
            __VA_ARGS__                                    \
        } else {                                           \
            typedef float ET;                              \
            __VA_ARGS__                                    \
        }                                                  \
    } while (0)

// ---- bilinear blend, one fixed evaluation order for every upsample kernel (element-per-thread and row-walking forms
// give bit-identical results):  v = a0 * (b0 * tl + b1 * tr) + a1 * (b0 * bl + b1 * br)
__device__ __forceinline__ float bilerp_h(float b0, float l, float b1, float r) { return fmaf(b0, l, b1 * r); }
__device__ __forceinline__ float bilerp_v(float a0, float top, float a1, float bot) { return fmaf(a0, top, a1 * bot); }

// ---- DPP cross-lane adds (pure VALU, no LDS crossbar) -------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_src(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane of the row ends with the row sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_src<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_src<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_src<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_src<0x140, 0xF>(v);  // row_mirror
    return v;
}
// sum over each 32-lane half of the wave; valid in lanes 16..31 (half 0) and 48..63 (half 1)
__device__ __forceinline__ float half32_sum_hi(float v) {
    v = row16_sum(v);
    v += dpp_src<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    return v;
}
// full 64-lane sum, valid in lane 63
__device__ __forceinline__ float wave_sum_l63(float v) {
    v = half32_sum_hi(v);
    v += dpp_src<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_src<0xB1, 0xF>(v));
    v = fmaxf(v, dpp_src<0x4E, 0xF>(v));
    v = fmaxf(v, dpp_src<0x141, 0xF>(v));
    v = fmaxf(v, dpp_src<0x140, 0xF>(v));
    return v;
}
// butterfly shuffles for the places where every lane needs the result
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// ---- operand maxima for the two-term fp16 split GEMMs (splitmma.hip, NT == 2) ---------------------------------
// A kernel that WRITES a GEMM operand (depthwise output, BatchNorm-backward dz) also leaves max |v| over the whole tensor in
// an "amax buffer": SMAAT_AMAX_WORDS zero-initialised uint32 words in device memory; the maximum is the maximum over the
// buffer (bit patterns of non-negative floats: unsigned compare = float compare; order-independent, i.e. bit-reproducible).
// Producers scatter their partial maxima over 32 words that lie in 32 different 128-byte lines (word 32 * (key % 32)) with
// fire-and-forget atomics; consumers take the maximum of those 32 words with scalar loads.  Why not one word: 20,000-80,000
// waves of a streaming kernel publishing to ONE address serialise at the memory side (~2.6 ns each, measured: +0.2 ms on a
// 0.13 ms BatchNorm-apply launch), and reading the word first to skip redundant atomics puts a ~2 us round trip in front
// of every wave's retirement (+40 %); profiles/r5/amax_publish_variants_r5.txt.  fmaxf drops NaNs: a NaN element stays a
// NaN in the scaled operand.
#define SMAAT_AMAX_WORDS 1024
#define SMAAT_AMAX_SLOTS 32
#define SMAAT_AMAX_STRIDE 32  // words between slots: one 128-byte line each
__device__ __forceinline__ float wave_max_l63(float v) {  // v >= 0; valid in lane 63 (masked-out DPP rows read 0)
    v = row16_max(v);
    v = fmaxf(v, dpp_src<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, dpp_src<0x143, 0xC>(v));  // row_bcast:31 into rows 2, 3
    return v;
}
// one atomic per WAVE (kernels whose waves are independent / may have returned early)
__device__ __forceinline__ void amax_publish_wave(unsigned* buf, float m, unsigned key) {
    m = wave_max_l63(m);
    if ((threadIdx.x & 63) == 63)
        (void)__hip_atomic_fetch_max(buf + (key % SMAAT_AMAX_SLOTS) * SMAAT_AMAX_STRIDE, __builtin_bit_cast(unsigned, m),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one atomic per 256-thread BLOCK (every thread of the block must call this); red = 4 floats of LDS
__device__ __forceinline__ void amax_publish_block256(unsigned* buf, float m, unsigned key, float* red) {
    m = wave_max_l63(m);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        (void)__hip_atomic_fetch_max(buf + (key % SMAAT_AMAX_SLOTS) * SMAAT_AMAX_STRIDE, __builtin_bit_cast(unsigned, t),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the maximum a consumer derives its scale from (wave-uniform: 32 scalar loads)
__device__ __forceinline__ unsigned amax_read(const unsigned* buf) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < SMAAT_AMAX_SLOTS; ++i) {
        const unsigned v = buf[i * SMAAT_AMAX_STRIDE];
        m = v > m ? v : m;
    }
    return m;
}
// scale exponent of an operand whose max |x| has the bit pattern `am`: x * 2^k has its maximum in [2^14, 2^15), inside
// fp16's range (65504) with 28 binades of normal range below it.  inf / nan / 0 -> 0; |k| <= 126 so that 2^k and 2^-k
// are both normal floats.
__host__ __device__ __forceinline__ int f16_kexp(unsigned am) {
    const int e = (int)((am >> 23) & 0xFFu);
    if (e == 255 || (am & 0x7FFFFFFFu) == 0u) return 0;
    const int k = 141 - e;
    return k > 126 ? 126 : (k < -126 ? -126 : k);
}
__device__ __forceinline__ float pow2i(int k) { return __builtin_bit_cast(float, (unsigned)(k + 127) << 23); }  // |k| <= 126

// block-wide sum for 256-thread blocks; result valid in thread 0. `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_t0(float v, float* red) {
    v = wave_sum_l63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- BatchNorm partial statistics of an MFMA accumulator tile ---------------------------------------
// The train-mode BatchNorm that follows a pointwise GEMM (reference unet_parts_depthwise_separable.py:25,34)
// needs mean and biased variance over (N, H, W).  Plain f32 sums of z and z^2 lose the variance when
// |mean| >> std (E[z^2] - E[z]^2 cancels), so every tile reports
//     part[0][tile][c] = mean_t,   part[1][tile][c] = M2_t = sum (z - mean_t)^2,   part[2][tile][c] = n_t
// and k_bn_finalize merges the tiles with the pairwise (Chan) update in fp64.  Inside a tile the sums are
// taken about a SHIFT that is itself a sample of the row (the value of its first valid pixel): with the shift
// inside the data range, S2 - S1^2/n has no cancellation beyond the spread of the data itself.
//
// bn_wave_partials: the 32 lanes of a half-wave hold PXT x 32 pixels of row (r & 3) + 8 (r >> 2) + 4 half of
// each 32-row tile ct (the 32x32 MFMA C layout).  Writes, for every row of the wave tile, (S1, S2, shift)
// about the shift (the row's value in lane `fl` of the half) to sp[0 / COT / 2 COT + row].
#define BN_STAT_FLOATS(WPX, COT) ((WPX) * 3 * (COT) + 8)  // LDS floats: [WPX][3][COT] wave partials + [8] pixel counts
// pixel count of a wave tile and its first valid pixel of pt = 0 from the validity masks (2-D tiles; lanes l and
// l + 32 hold the same pixel).  Kernels with flattened tiles compute both arithmetically instead.
template <int PXT>
__device__ __forceinline__ int bn_wave_count(const bool (&pval)[PXT], int& fl) {
    const unsigned lo = (unsigned)__ballot(pval[0]);
    fl = lo ? __builtin_ctz(lo) : 0;
    int n = __builtin_popcount(lo);
#pragma unroll
    for (int pt = 1; pt < PXT; ++pt) n += __builtin_popcount((unsigned)__ballot(pval[pt]));
    return n;
}

template <int CT, int PXT>
__device__ __forceinline__ void bn_wave_partials(const f32x16 (&acc)[CT][PXT], const bool (&pval)[PXT], int l31,
                                                 int half, float* sp, int COT, int fl) {
    // The shift of a row = its value at the first valid pixel (lane fl of each half).  It is handed to the other
    // lanes through the wave's own shift row of the stat buffer (written by two lanes, read back by all 64: LDS
    // operations of ONE wave execute in issue order, so no barrier is involved).  v_readlane would park 2 x 32
    // values in SGPRs (the persistent GEMM is at the scalar-register limit: spills), and ds_bpermute of consecutive
    // accumulator registers was miscompiled by hipcc 7.2 (every row got the shift of row 0).
    // Register r of tile ct is row ct*32 + (r & 3) + 8 (r >> 2) + 4 half: registers 4g .. 4g+3 are four consecutive
    // rows -> 16-byte LDS accesses.
    float* shp = sp + 2 * COT + 4 * half;
    if (l31 == fl) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(shp + ct * 32 + 8 * g) = make_float4(acc[ct][0][4 * g], acc[ct][0][4 * g + 1],
                                                                acc[ct][0][4 * g + 2], acc[ct][0][4 * g + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int l15 = l31 & 15;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        float sh[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *(const float4*)(shp + ct * 32 + 8 * g);
            sh[4 * g] = v.x;
            sh[4 * g + 1] = v.y;
            sh[4 * g + 2] = v.z;
            sh[4 * g + 3] = v.w;
        }
        // 16-lane (DPP row) sums of every register; lane (16 k + j) keeps the sum of register j: after the loop the
        // two DPP rows of a half hold the two halves of each row's total, combined by ONE xor-16 swizzle.
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s, q;
            {
                const float d = pval[0] ? acc[ct][0][r] - sh[r] : 0.f;
                s = d;
                q = d * d;
            }
#pragma unroll
            for (int pt = 1; pt < PXT; ++pt) {
                const float d = pval[pt] ? acc[ct][pt][r] - sh[r] : 0.f;
                s += d;
                q = fmaf(d, d, q);
            }
            s = row16_sum(s);
            q = row16_sum(q);
            int lr = l15;
            asm volatile("" : "+v"(lr));  // a fresh compare per row: 16 hoisted lane masks would cost 32 SGPRs (spills)
            const bool mine = lr == r;
            ts = mine ? s : ts;
            tq = mine ? q : tq;
        }
        ts += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, ts), 0x401F));  // lane ^ 16
        tq += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, tq), 0x401F));
        if (l31 < 16) {
            const int rc = ct * 32 + (l31 & 3) + 8 * (l31 >> 2) + 4 * half;
            sp[rc] = ts;
            sp[COT + rc] = tq;
        }
    }
}

// merge the WPX wave partials of one row into the tile's (mean, M2, n); stat = [WPX][3][COT], cnt = [WPX].
// Two passes over <= 4 waves; v_rcp_f32 (1 ulp) is ample: it only scales deviations from a shift that is itself a
// sample of the row.
template <int WPX>
__device__ __forceinline__ void bn_tile_combine(const float* stat, const int* cnt, int COT, int col, float& mean,
                                                float& m2, float& n) {
    // branch-free (no per-wave predicates kept in scalar registers): an empty wave has s1 = s2 = 0 and a finite
    // shift (its accumulators come from clamped, valid loads), and enters every sum with weight n_w = 0
    float nw[WPX], mw[WPX], qw[WPX];
    float nt = 0.f, sm = 0.f;
#pragma unroll
    for (int w = 0; w < WPX; ++w) {
        nw[w] = (float)cnt[w];
        const float* sp = stat + w * 3 * COT + col;
        const float s1 = sp[0], s2 = sp[COT], sh = sp[2 * COT];
        const float inv = __builtin_amdgcn_rcpf(fmaxf(nw[w], 1.f));
        mw[w] = fmaf(s1, inv, sh);
        qw[w] = fmaxf(fmaf(-s1 * inv, s1, s2), 0.f);
        nt += nw[w];
        sm = fmaf(nw[w], mw[w] - mw[0], sm);  // about the first wave's mean (wave 0 holds the tile's first pixel)
    }
    mean = fmaf(sm, __builtin_amdgcn_rcpf(fmaxf(nt, 1.f)), mw[0]);
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < WPX; ++w) {
        const float d = mw[w] - mean;
        q += fmaf(nw[w] * d, d, qw[w]);
    }
    m2 = q;
    n = nt;
}

// ---- pixel-tile geometry ---------------------------------------------------------
// A "pixel tile" is PT pixels of ONE image.  mode 0: PT consecutive pixels of the
// flattened H*W plane (small maps).  mode 1: a TH x TW patch (TH*TW == PT).
struct TileGeom {
    int H, W, P;
    int mode, TH, TW;
    int tiles_x;        // mode 1: tiles along W
    int tiles_per_img;  // per image
    int T;              // N * tiles_per_img
    int PT;
};

struct StageRegion {
    int row_lo, col_lo, nrows, SW;  // region = nrows x SW staged floats per channel
};

__device__ __forceinline__ StageRegion stage_region(const TileGeom& g, int tl) {
    StageRegion r;
    if (g.mode == 0) {
        const int p0 = tl * g.PT;
        int p1 = p0 + g.PT;
        if (p1 > g.P) p1 = g.P;
        r.row_lo = p0 / g.W - 1;
        r.nrows = (p1 - 1) / g.W - r.row_lo + 2;
        r.SW = g.W + 2;
        r.col_lo = -1;
    } else {
        const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
        r.row_lo = ty * g.TH - 1;
        r.nrows = g.TH + 2;
        r.SW = g.TW + 2;
        r.col_lo = tx * g.TW - 1;
    }
    return r;
}

// tile pixel i -> (row, col); returns false when the pixel is outside the image
__device__ __forceinline__ bool tile_pixel(const TileGeom& g, int tl, int i, int& r, int& c) {
    if (g.mode == 0) {
        const int p = tl * g.PT + i;
        r = p / g.W;
        c = p - r * g.W;
        return p < g.P;
    }
    const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
    const int tr = i / g.TW, tc = i - tr * g.TW;
    r = ty * g.TH + tr;
    c = tx * g.TW + tc;
    return (r < g.H) && (c < g.W);
}

// ---- launch-argument blocks of the GEMM family (shared by the kernels and the C ABI) ----
struct PwArgs {
    const float* x;
    long x_bs;
    const float* in_scale;
    const float* in_shift;
    const float* w_dw;
    const float* b_dw;
    const float* wt;    // [Kdim][M]
    const float* bias;  // [M] or null
    float* out;
    long out_bs;
    float* part;   // [2][T][M] or null
    float* y_out;  // [N][Kdim][P] or null: depthwise output side product (kept for the weight gradient)
    int N, Cin, kpl, Kdim, M, nco, sstride;
    TileGeom g;
    int dbg;  // timing ablations only (SMAAT_PW_ABLATE): 1 = consumers skip the MFMAs, 2 = producers idle
    float out_floor;  // epilogue: out = max(acc + bias, out_floor); -inf = plain, 0 = fused ReLU (inference path)
};

struct Wg2Args {
    const float* dz;
    long dz_bs;
    const float* y;
    long y_bs;
    float* part;  // [nsplit][M][K]
    int N, M, K, P, nmt, nkt, nsplit, chunks_per_split, nchunk_img, total_chunks;
    // two-term fp16 split (splitmma.hip, NT == 2): bit patterns of max |dz| and max |y| over the whole operand tensors
    // (device memory, written by the kernels that produce the operands); null selects the exact three-term bf16 split
    const unsigned* dz_amax;
    const unsigned* y_amax;
};

struct WgArgs {
    const float* x;
    long x_bs;
    const float* in_scale;
    const float* in_shift;
    const float* w_dw;
    const float* b_dw;
    const float* dz;
    long dz_bs;
    float* dwpart;  // [nsplit][M][Kdim]
    int N, Cin, kpl, Kdim, M, nco, nkt, nsplit, tiles_per_split, sstride;
    TileGeom g;  // PT == PSW
};

struct PwSplitArgs {
    const float* x;
    long x_bs;
    const unsigned short* planes;  // [Cp/16][3][M][16]
    const float* bias;             // [M] or null
    float* out;
    long out_bs;
    float* part;  // [2][T][M] or null
    int N, Cin, Cp, M, P, nco, tiles_per_img, T, slots;
    int dbg;  // reserved (0)
    float out_floor;  // epilogue: out = max(acc + bias, out_floor); -inf = plain, 0 = fused ReLU (inference path)
    // split-K (inference at small batch): image v = n * ksplit + s reads the s-th slice of Cin channels of x (x_bs = slice
    // stride) and of the weight planes (planes_bs elements per slice) and writes partial image v; ksplit = 1, planes_bs = 0
    // otherwise
    int ksplit;
    long planes_bs;
    // two-term fp16 split (NT == 2): planes = fp16 image [Cp/16][2][M][16] of A * 2^kexp (smaat_split_planes_h), a_kexp -> that
    // exponent (the image's trailer), x_amax -> bit pattern of max |x| over the whole operand tensor.  Both null otherwise.
    const unsigned* x_amax;
    const int* a_kexp;
};

// mixed-precision GEMMs (bf16gemm.hip)
struct PwBfArgs {
    const bf16_t* x;       // [N][Cin][P]
    long x_bs;
    const bf16_t* planes;  // [Cp/16][M][16], Cp = Cin rounded up to 32 (smaat_bf16_planes)
    const float* bias;     // [M] or null
    void* out;             // [N][M][P], f32 or bf16
    long out_bs;
    float* part;  // [3][slots][M] or null
    int N, Cin, Cp, M, P, nco, tiles_per_img, T, slots;
    float out_floor;
};
struct WgBfArgs {
    const bf16_t* dz;  // [N][M][P]
    long dz_bs;
    const bf16_t* y;  // [N][K][P]
    long y_bs;
    float* part;  // [nsplit][M][K]
    int N, M, K, P, nmt, nkt, nsplit, nchunk_img, total_chunks;
};

#define HIP_RET(expr)                          \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) return (int)_e;  \
    } while (0)
