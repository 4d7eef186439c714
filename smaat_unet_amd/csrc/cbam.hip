// CBAM = ChannelAttention -> SpatialAttention  (reference models/layers.py:90-141)
//
// forward:
//   k_cbam_chpool   avg / max (+ first argmax) over H*W per (n,c)            layers.py:107-108
//   k_cbam_mlp      shared MLP on both pooled vectors, sum, sigmoid -> s[n][c]   :98-103,109-110
//   k_cbam_sppool   mean / max over channels of x*s -> maps[n][2][p]          :123-125
//   k_cbam_spconv   Conv2d(2,1,k,pad k/2,no bias) + BN(1) partial stats          :126-127
//   (bn finalize from bn.hip, C = 1)
//   k_cbam_apply    out = x * s[n][c] * sigmoid(conv*scale+shift)[n][p]          :110,128
// backward: see the individual kernels.
#include "common.h"
#include <stdlib.h>

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// ---------------------------------------------------------------------------------
// ACT: x is the PRE-BatchNorm tensor of the block that feeds the attention (reference
// unet_parts_depthwise_separable.py:34-35 in front of layers.py:107-108): y = relu(x * scale[c] + shift[c]) is formed on
// load, pooled, and WRITTEN to y_out -- the block output is materialised by its first consumer, the separate
// BatchNorm-apply + ReLU pass over it disappears.
// T: element type of x and y_out (f32 | bf16 storage).  With bf16 storage the pools are taken over the values AS STORED
// (rounded to bf16), i.e. over the tensor every later consumer reads -- the argmax index stays exact.
template <bool ACT, typename T>
__global__ __launch_bounds__(256) void k_cbam_chpool(const T* __restrict__ x, long x_bs, int C, int P,
                                                     float* __restrict__ avg, float* __restrict__ mx,
                                                     int* __restrict__ amax, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, T* __restrict__ y_out,
                                                     long y_bs) {
    __shared__ float rf[8];
    __shared__ int ri[4];
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const T* xp = x + (long)n * x_bs + (long)c * P;
    T* yp = ACT ? y_out + (long)n * y_bs + (long)c * P : nullptr;
    const float asc = ACT ? scale[c] : 1.f, ash = ACT ? shift[c] : 0.f;
    auto act4 = [&](float4 v) {
        if (ACT) {  // the expression of k_affine_act: identical bits
            v.x = as_stored(yp, fmaxf(fmaf(v.x, asc, ash), 0.f));
            v.y = as_stored(yp, fmaxf(fmaf(v.y, asc, ash), 0.f));
            v.z = as_stored(yp, fmaxf(fmaf(v.z, asc, ash), 0.f));
            v.w = as_stored(yp, fmaxf(fmaf(v.w, asc, ash), 0.f));
        }
        return v;
    };
    float s = 0.f, m = -INFINITY;
    int mi = 0x7fffffff;
    // branch-free running maximum (strictly greater: the first index wins, positions are visited in increasing
    // order), 16-byte loads, four of them in flight per thread: with a divergent `if (v > m)` around the update
    // hipcc waits for every load before issuing the next one
    auto upd = [&](float v, int p) {
        s += v;
        const bool gt = v > m;
        m = gt ? v : m;
        mi = gt ? p : mi;
    };
    if ((P & 3) == 0 && (x_bs & 3) == 0 && ((((uintptr_t)x) & Elem<T>::vmask) == 0) &&
        (!ACT || ((y_bs & 3) == 0 && ((((uintptr_t)y_out) & Elem<T>::vmask) == 0)))) {
        const int P4 = P >> 2;
        int q = threadIdx.x;
        for (; q + 768 < P4; q += 1024) {
            typename Elem<T>::raw4 vr[4];
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) vr[u] = ldraw4(xp + 4 * (q + 256 * u));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = 4 * (q + 256 * u);
                v[u] = act4(cvt4(vr[u]));
                if (ACT) st4(yp + p, v[u]);
                upd(v[u].x, p);
                upd(v[u].y, p + 1);
                upd(v[u].z, p + 2);
                upd(v[u].w, p + 3);
            }
        }
        for (; q < P4; q += 256) {
            const float4 v = act4(ld4(xp + 4 * q));
            if (ACT) st4(yp + 4 * q, v);
            upd(v.x, 4 * q);
            upd(v.y, 4 * q + 1);
            upd(v.z, 4 * q + 2);
            upd(v.w, 4 * q + 3);
        }
    } else {
        for (int p = threadIdx.x; p < P; p += 256) {
            float v = ld1(xp + p);
            if (ACT) {
                v = as_stored(yp, fmaxf(fmaf(v, asc, ash), 0.f));
                st1(yp + p, v);
            }
            upd(v, p);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float ws = wave_sum_all(s);
    const float wm = wave_max_all(m);
    if (lane == 0) {
        rf[wave] = ws;
        rf[4 + wave] = wm;
    }
    __syncthreads();
    const float bm = fmaxf(fmaxf(rf[4], rf[5]), fmaxf(rf[6], rf[7]));
    int cand = (m == bm) ? mi : 0x7fffffff;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const int o = __shfl_xor(cand, k, 64);
        cand = o < cand ? o : cand;
    }
    if (lane == 0) ri[wave] = cand;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (double)rf[0] + (double)rf[1] + (double)rf[2] + (double)rf[3];
        avg[plane] = (float)(tot / (double)P);
        mx[plane] = bm;
        int a = ri[0];
        a = ri[1] < a ? ri[1] : a;
        a = ri[2] < a ? ri[2] : a;
        a = ri[3] < a ? ri[3] : a;
        amax[plane] = a;
    }
}

// Channel pooling + the MaxPool2d(2) that reads the same tensor (an encoder level of SmaAt_UNet.forward feeds both CBAM
// and the next DownDS: reference SmaAt_UNet.py:43-50, unet_parts_depthwise_separable.py:48), in ONE pass: a thread owns
// 2 x 4 patches (two window rows), so the pooled map comes from the registers the channel pools are taken from and the
// separate max-pool pass over the level output (one read of every encoder activation) disappears.  Same maxima and
// first-argmax as k_cbam_chpool (a thread still visits its positions in increasing order), same activation expression,
// same max order as k_maxpool2_fwd: those outputs are bit-identical to the two kernels it replaces; the mean adds the same
// terms patch-major instead of row-major (f32 partial sums per thread, f64 across waves).  W % 4 == 0.
template <bool ACT, typename T>
__global__ __launch_bounds__(256) void k_cbam_chpool_pool(const T* __restrict__ x, long x_bs, int C, int H, int W,
                                                          float* __restrict__ avg, float* __restrict__ mx,
                                                          int* __restrict__ amax, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ y_out,
                                                          long y_bs, T* __restrict__ pooled, long p_bs) {
    __shared__ float rf[8];
    __shared__ int ri[4];
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const int P = H * W, ncol4 = W >> 2, npair = (H + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
    const T* xp = x + (long)n * x_bs + (long)c * P;
    T* yp = ACT ? y_out + (long)n * y_bs + (long)c * P : nullptr;
    T* pp = pooled + (long)n * p_bs + (long)c * Ho * Wo;
    const float asc = ACT ? scale[c] : 1.f, ash = ACT ? shift[c] : 0.f;
    auto act4 = [&](float4 v) {
        if (ACT) {
            v.x = as_stored(yp, fmaxf(fmaf(v.x, asc, ash), 0.f));
            v.y = as_stored(yp, fmaxf(fmaf(v.y, asc, ash), 0.f));
            v.z = as_stored(yp, fmaxf(fmaf(v.z, asc, ash), 0.f));
            v.w = as_stored(yp, fmaxf(fmaf(v.w, asc, ash), 0.f));
        }
        return v;
    };
    float s = 0.f, m = -INFINITY;
    int mi = 0x7fffffff;
    auto upd = [&](float v, int p) {
        s += v;
        const bool gt = v > m;
        m = gt ? v : m;
        mi = gt ? p : mi;
    };
    auto upd4 = [&](const float4 v, int p) {
        upd(v.x, p);
        upd(v.y, p + 1);
        upd(v.z, p + 2);
        upd(v.w, p + 3);
    };
    const int npatch = npair * ncol4;
    typedef typename Elem<T>::raw4 R4;
    for (int idx = threadIdx.x; idx < npatch; idx += 512) {  // two patches per trip: four loads in flight
        const int idx2 = idx + 256;
        const bool has2 = idx2 < npatch;
        const int i0 = idx / ncol4, q0 = idx - i0 * ncol4;
        const int i1 = has2 ? idx2 / ncol4 : i0, q1 = has2 ? idx2 - i1 * ncol4 : q0;
        const int pa = 2 * i0 * W + 4 * q0, pb = 2 * i1 * W + 4 * q1;
        const bool two0 = 2 * i0 + 1 < H, two1 = 2 * i1 + 1 < H;
        const R4 ra0 = ldraw4(xp + pa), ra1 = ldraw4(xp + (two0 ? pa + W : pa));
        const R4 rb0 = ldraw4(xp + pb), rb1 = ldraw4(xp + (two1 ? pb + W : pb));
        {
            const float4 v0 = act4(cvt4(ra0)), v1 = act4(cvt4(ra1));
            if (ACT) st4(yp + pa, v0);
            upd4(v0, pa);
            if (two0) {
                if (ACT) st4(yp + pa + W, v1);
                upd4(v1, pa + W);
                st2(pp + (long)i0 * Wo + 2 * q0, fmaxf(fmaxf(v0.x, v0.y), fmaxf(v1.x, v1.y)), fmaxf(fmaxf(v0.z, v0.w), fmaxf(v1.z, v1.w)));
            }
        }
        if (has2) {
            const float4 v0 = act4(cvt4(rb0)), v1 = act4(cvt4(rb1));
            if (ACT) st4(yp + pb, v0);
            upd4(v0, pb);
            if (two1) {
                if (ACT) st4(yp + pb + W, v1);
                upd4(v1, pb + W);
                st2(pp + (long)i1 * Wo + 2 * q1, fmaxf(fmaxf(v0.x, v0.y), fmaxf(v1.x, v1.y)), fmaxf(fmaxf(v0.z, v0.w), fmaxf(v1.z, v1.w)));
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float ws = wave_sum_all(s);
    const float wm = wave_max_all(m);
    if (lane == 0) {
        rf[wave] = ws;
        rf[4 + wave] = wm;
    }
    __syncthreads();
    const float bm = fmaxf(fmaxf(rf[4], rf[5]), fmaxf(rf[6], rf[7]));
    int cand = (m == bm) ? mi : 0x7fffffff;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const int o = __shfl_xor(cand, k, 64);
        cand = o < cand ? o : cand;
    }
    if (lane == 0) ri[wave] = cand;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (double)rf[0] + (double)rf[1] + (double)rf[2] + (double)rf[3];
        avg[plane] = (float)(tot / (double)P);
        mx[plane] = bm;
        int a = ri[0];
        a = ri[1] < a ? ri[1] : a;
        a = ri[2] < a ? ri[2] : a;
        a = ri[3] < a ? ri[3] : a;
        amax[plane] = a;
    }
}

// one block per sample.  LDS: avg[C], mx[C], ha[Cr], hm[Cr]
__global__ __launch_bounds__(256) void k_cbam_mlp(const float* __restrict__ avg, const float* __restrict__ mx,
                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                  const float* __restrict__ w2, const float* __restrict__ b2, int C,
                                                  int Cr, float* __restrict__ ha_out, float* __restrict__ hm_out,
                                                  float* __restrict__ s_out) {
    extern __shared__ float sm[];
    float* la = sm;
    float* lm = la + C;
    float* ha = lm + C;
    float* hm = ha + Cr;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        la[c] = avg[(long)n * C + c];
        lm[c] = mx[(long)n * C + c];
    }
    __syncthreads();
    for (int j = wave; j < Cr; j += 4) {
        float pa = 0.f, pm = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float w = w1[(long)j * C + c];
            pa = fmaf(w, la[c], pa);
            pm = fmaf(w, lm[c], pm);
        }
        pa = wave_sum_all(pa);
        pm = wave_sum_all(pm);
        if (lane == 0) {
            const float a = fmaxf(pa + b1[j], 0.f), m = fmaxf(pm + b1[j], 0.f);
            ha[j] = a;
            hm[j] = m;
            ha_out[(long)n * Cr + j] = a;
            hm_out[(long)n * Cr + j] = m;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float oa = b2[c], om = b2[c];
        for (int j = 0; j < Cr; ++j) {
            const float w = w2[(long)c * Cr + j];
            oa = fmaf(w, ha[j], oa);
            om = fmaf(w, hm[j], om);
        }
        s_out[(long)n * C + c] = sigmoidf_(oa + om);
    }
}

// thread per pixel, loop over channels.  maps[n][0][p] = mean_c(x*s), maps[n][1][p] = max_c(x*s)
template <typename T>
__global__ __launch_bounds__(256) void k_cbam_sppool(const T* __restrict__ x, long x_bs,
                                                     const float* __restrict__ s, int C, int P,
                                                     float* __restrict__ maps) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const T* xp = x + (long)n * x_bs + p;
    const float* sp = s + (long)n * C;
    float sum = 0.f, m = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float v = ld1(xp + (long)c * P) * sp[c];
        sum += v;
        m = fmaxf(m, v);
    }
    maps[((long)n * 2 + 0) * P + p] = sum / (float)C;
    maps[((long)n * 2 + 1) * P + p] = m;
}

// k x k conv (k = 3 or 7) over the 2-channel maps, 16x16 pixel tiles, + BN(1) partials
// part[3][nblocks] = (mean, M2, n) per block
#define SPT 16
__global__ __launch_bounds__(256) void k_cbam_spconv(const float* __restrict__ maps, const float* __restrict__ wc,
                                                     int ks, int H, int W, float* __restrict__ conv,
                                                     float* __restrict__ part, int nblocks) {
    __shared__ float tile[2][SPT + 6][SPT + 6 + 1];
    __shared__ float wl[2 * 49];
    __shared__ float red[8];
    const int n = blockIdx.z, P = H * W;
    const int r0 = blockIdx.y * SPT, c0 = blockIdx.x * SPT;
    const int pd = ks >> 1, R = SPT + 2 * pd;
    const int tid = threadIdx.x;
    if (tid < 2 * ks * ks) wl[tid] = wc[tid];
    for (int e = tid; e < 2 * R * R; e += 256) {
        const int ch = e / (R * R), rem = e - ch * R * R;
        const int sr = rem / R, sc = rem - sr * R;
        const int gr = r0 - pd + sr, gc = c0 - pd + sc;
        float v = 0.f;
        if (gr >= 0 && gr < H && gc >= 0 && gc < W) v = maps[((long)n * 2 + ch) * P + gr * W + gc];
        tile[ch][sr][sc] = v;
    }
    __syncthreads();
    const int tr = tid / SPT, tc = tid - tr * SPT;
    const int r = r0 + tr, c = c0 + tc;
    float acc = 0.f;
    const bool valid = r < H && c < W;
    if (valid) {
        for (int ch = 0; ch < 2; ++ch)
            for (int i = 0; i < ks; ++i)
                for (int j = 0; j < ks; ++j) acc = fmaf(wl[(ch * ks + i) * ks + j], tile[ch][tr + i][tc + j], acc);
        conv[(long)n * P + r * W + c] = acc;
    }
    // BN(1) partial of this block as (mean, M2, n) about the block mean (common.h "BatchNorm partial statistics")
    const float v = valid ? acc : 0.f;
    const float cntf = block_sum_t0(valid ? 1.f : 0.f, red);
    const float t1 = block_sum_t0(v, red + 4);
    const float mb = cntf > 0.f ? t1 / cntf : 0.f;  // every thread sees the same four wave sums -> same value
    const float d = valid ? acc - mb : 0.f;
    const float t2 = block_sum_t0(d * d, red);
    if (tid == 0) {
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[blk] = mb;
        part[nblocks + blk] = t2;
        part[2 * nblocks + blk] = cntf;
    }
}

// m[n][p] = sigmoid(conv * scale + shift)   (BN(1) + sigmoid)
__global__ __launch_bounds__(256) void k_cbam_gate(const float* __restrict__ conv, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, long total,
                                                   float* __restrict__ gate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) gate[i] = sigmoidf_(fmaf(conv[i], scale[0], shift[0]));
}

// out[n][c][p] = x[n][c][p] * s[n][c] * gate[n][p];  grid (N*C planes, segments)
template <typename T>
__global__ __launch_bounds__(256) void k_cbam_apply(const T* __restrict__ x, long x_bs,
                                                    const float* __restrict__ s, const float* __restrict__ gate,
                                                    T* __restrict__ out, long out_bs, int C, int P, int seg_len,
                                                    unsigned* __restrict__ amax) {
    __shared__ float amred[4];
    float am = 0.f;
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float sv = s[plane];
    const T* xp = x + (long)n * x_bs + (long)c * P;
    const float* gp = gate + (long)n * P;
    T* op = out + (long)n * out_bs + (long)c * P;
    const int p0 = blockIdx.y * seg_len;
    int p1 = p0 + seg_len;
    if (p1 > P) p1 = P;
    const bool vec = ((P & 3) == 0) && ((x_bs & 3) == 0) && ((out_bs & 3) == 0) && ((seg_len & 3) == 0) &&
                     ((((uintptr_t)x) & Elem<T>::vmask) == 0) && ((((uintptr_t)out) & Elem<T>::vmask) == 0) &&
                     ((((uintptr_t)gate) & 15) == 0);
    if (vec) {
        auto one = [&](float4 v, const float4 g, int p) {
            v.x = v.x * sv * g.x;
            v.y = v.y * sv * g.y;
            v.z = v.z * sv * g.z;
            v.w = v.w * sv * g.w;
            am = fmaxf(am, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st4(op + p, v);
        };
        int p = p0 + threadIdx.x * 4;  // two positions per trip: four loads in flight
        for (; p + 1024 < p1; p += 2048) {
            const auto xa = ldraw4(xp + p);
            const float4 ga = *(const float4*)(gp + p);
            const auto xb = ldraw4(xp + p + 1024);
            const float4 gb = *(const float4*)(gp + p + 1024);
            one(cvt4(xa), ga, p);
            one(cvt4(xb), gb, p + 1024);
        }
        if (p < p1) one(ld4(xp + p), *(const float4*)(gp + p), p);
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float v = ld1(xp + p) * sv * gp[p];
            am = fmaxf(am, fabsf(v));
            st1(op + p, v);
        }
    }
    // nullable amax buffer (common.h): max |out| (f32 storage) -- the scale bound of a row-walking fused forward that reads the
    // concatenation buffer this kernel fills (dsrows.hip, NT == 2)
    if (amax) amax_publish_block256(amax, am, blockIdx.x + blockIdx.y * gridDim.x, amred);  // (block-uniform; no thread has returned)
}

// ===================================== inference (eval mode) ======================================
// Three launches for a whole CBAM (+ the MaxPool2d(2) that consumes the same tensor in SmaAt_UNet.forward):
//   k_cbam_chpool          avg / max over H*W per (n, c)                                        (as in training)
//   k_cbam_mlp_sppool      every block recomputes the tiny shared MLP -> s[C] in LDS, then mean / max over channels
//                          of x*s for its 256 pixels; the four waves split the channels (4x shorter load chains)
//   k_cbam_gate_apply      k x k conv on the 2-channel maps (halo tile in LDS) + BatchNorm(1) with RUNNING statistics
//                          (a fixed affine map: no grid-wide reduction) + sigmoid -> gate, then
//                          out = x * s * gate for the block's channel range, and pooled = maxpool2(x) from the same loads
// reference: models/layers.py:105-111, 122-129, 138-141; unet_parts_depthwise_separable.py:48
// NWV waves per block (4, or 16 when the plane gives too few blocks: the channel loop of a wave is then 4x shorter)
template <int NWV>
__global__ __launch_bounds__(NWV * 64) void k_cbam_mlp_sppool(const float* __restrict__ x, long x_bs,
                                                         const float* __restrict__ avg, const float* __restrict__ mx,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, int C,
                                                         int Cr, int P, float* __restrict__ s_out,
                                                         float* __restrict__ maps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* la = sm;            // [C]
    float* lm = la + C;        // [C]
    float* sl = lm + C;        // [C]
    float* ha = sl + C;        // [Cr]
    float* hm = ha + Cr;       // [Cr]
    constexpr int NT = NWV * 64;
    float4* red = (float4*)(sm + ((3 * C + 2 * Cr + 3) & ~3));  // [2][NWV waves][64 lanes], 16-byte aligned
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += NT) {
        la[c] = avg[(long)n * C + c];
        lm[c] = mx[(long)n * C + c];
    }
    __syncthreads();
    // Shared MLP, recomputed by every block: at batch 1 its latency chains ARE the kernel on the deep levels (C = 512,
    // Cr = 32: 128 KB of weights), so both layers are laid out for independent loads: a hidden unit is a dot product
    // spread over 16 lanes (quarter wave) with all its loads issued before the DPP row sum, an output channel reads its
    // Cr contiguous weights with the loop unrolled.
    {
        const int q16 = tid >> 4, l16 = tid & 15;  // NT / 16 groups of 16 lanes
        for (int j = q16; j < Cr; j += NT / 16) {
            float pa = 0.f, pm = 0.f;
            const float* wr = w1 + (long)j * C;
#pragma unroll 4
            for (int c = l16; c < C; c += 16) {
                const float w = wr[c];
                pa = fmaf(w, la[c], pa);
                pm = fmaf(w, lm[c], pm);
            }
            pa = row16_sum(pa);
            pm = row16_sum(pm);
            if (l16 == 0) {
                ha[j] = fmaxf(pa + b1[j], 0.f);
                hm[j] = fmaxf(pm + b1[j], 0.f);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        float oa = b2[c], om = b2[c];
        const float* wr = w2 + (long)c * Cr;
#pragma unroll 8
        for (int j = 0; j < Cr; ++j) {
            const float w = wr[j];
            oa = fmaf(w, ha[j], oa);
            om = fmaf(w, hm[j], om);
        }
        const float sv = sigmoidf_(oa + om);
        sl[c] = sv;
        if (blockIdx.x == 0) s_out[(long)n * C + c] = sv;
    }
    __syncthreads();
    // pixels p .. p+3 of this lane; wave w takes channels w, w+4, ...
    const int p = blockIdx.x * 256 + lane * 4;
    const bool vec = ((P & 3) == 0) && ((x_bs & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (p < P) {
        const float* xp = x + (long)n * x_bs + p;
        if (vec) {
            int c = wave;
            for (; c + 3 * NWV < C; c += 4 * NWV) {  // four loads in flight
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(xp + (long)(c + NWV * u) * P);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float sv = sl[c + NWV * u];
                    const float a = v[u].x * sv, b = v[u].y * sv, cc = v[u].z * sv, d = v[u].w * sv;
                    sum.x += a; sum.y += b; sum.z += cc; sum.w += d;
                    m.x = fmaxf(m.x, a); m.y = fmaxf(m.y, b); m.z = fmaxf(m.z, cc); m.w = fmaxf(m.w, d);
                }
            }
            for (; c < C; c += NWV) {
                const float4 v = *(const float4*)(xp + (long)c * P);
                const float sv = sl[c];
                const float a = v.x * sv, b = v.y * sv, cc = v.z * sv, d = v.w * sv;
                sum.x += a; sum.y += b; sum.z += cc; sum.w += d;
                m.x = fmaxf(m.x, a); m.y = fmaxf(m.y, b); m.z = fmaxf(m.z, cc); m.w = fmaxf(m.w, d);
            }
        } else {
            for (int c = wave; c < C; c += NWV) {
                const float sv = sl[c];
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = (p + q < P) ? xp[(long)c * P + q] * sv : 0.f;
                sum.x += e[0]; sum.y += e[1]; sum.z += e[2]; sum.w += e[3];
                m.x = fmaxf(m.x, e[0]); m.y = fmaxf(m.y, e[1]); m.z = fmaxf(m.z, e[2]); m.w = fmaxf(m.w, e[3]);
            }
        }
    }
    red[wave * 64 + lane] = sum;
    red[NT + wave * 64 + lane] = m;
    __syncthreads();
    if (wave == 0 && p < P) {
        float4 ts = red[lane], tm = red[NT + lane];
#pragma unroll
        for (int w = 1; w < NWV; ++w) {
            const float4 a = red[w * 64 + lane], b = red[NT + w * 64 + lane];
            ts.x += a.x; ts.y += a.y; ts.z += a.z; ts.w += a.w;
            tm.x = fmaxf(tm.x, b.x); tm.y = fmaxf(tm.y, b.y); tm.z = fmaxf(tm.z, b.z); tm.w = fmaxf(tm.w, b.w);
        }
        const float ic = 1.f / (float)C;
        float* m0 = maps + ((long)n * 2 + 0) * P + p;
        float* m1 = maps + ((long)n * 2 + 1) * P + p;
        const float so[4] = {ts.x * ic, ts.y * ic, ts.z * ic, ts.w * ic}, mo[4] = {tm.x, tm.y, tm.z, tm.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (p + q < P) {
                m0[q] = so[q];
                m1[q] = mo[q];
            }
    }
}

#define GAT_TH 8
#define GAT_TW 32
__global__ __launch_bounds__(256) void k_cbam_gate_apply(const float* __restrict__ x, long x_bs,
                                                         const float* __restrict__ s, const float* __restrict__ maps,
                                                         const float* __restrict__ wc, int ks,
                                                         const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                         const float* __restrict__ bn_rm, const float* __restrict__ bn_rv,
                                                         float eps, int C, int H, int W, int csplit,
                                                         float* __restrict__ out, long out_bs,
                                                         float* __restrict__ pooled, long pooled_bs) {
    __shared__ float tile[2][GAT_TH + 6][GAT_TW + 6 + 1];
    __shared__ float wl[2 * 49];
    const int n = blockIdx.z / csplit, cs = blockIdx.z - n * csplit;
    const int P = H * W;
    const int r0 = blockIdx.y * GAT_TH, c0 = blockIdx.x * GAT_TW;
    const int pd = ks >> 1, RH = GAT_TH + 2 * pd, RW = GAT_TW + 2 * pd;
    const int tid = threadIdx.x;
    if (tid < 2 * ks * ks) wl[tid] = wc[tid];
    for (int e = tid; e < 2 * RH * RW; e += 256) {
        const int ch = e / (RH * RW), rem = e - ch * RH * RW;
        const int sr = rem / RW, sc = rem - sr * RW;
        const int gr = r0 - pd + sr, gc = c0 - pd + sc;
        float v = 0.f;
        if (gr >= 0 && gr < H && gc >= 0 && gc < W) v = maps[((long)n * 2 + ch) * P + gr * W + gc];
        tile[ch][sr][sc] = v;
    }
    __syncthreads();
    const int tr = tid >> 5, tc = tid & 31;
    const int r = r0 + tr, c = c0 + tc;
    const bool valid = r < H && c < W;
    float acc = 0.f;
    for (int ch = 0; ch < 2; ++ch)
        for (int i = 0; i < ks; ++i)
            for (int j = 0; j < ks; ++j) acc = fmaf(wl[(ch * ks + i) * ks + j], tile[ch][tr + i][tc + j], acc);
    const float is = 1.f / sqrtf(bn_rv[0] + eps);
    const float sc_ = (bn_g ? bn_g[0] : 1.f) * is;
    const float sh_ = (bn_b ? bn_b[0] : 0.f) - bn_rm[0] * sc_;
    const float g = sigmoidf_(fmaf(acc, sc_, sh_));
    // channel range of this block
    const int per = (C + csplit - 1) / csplit;
    const int cb = cs * per;
    int ce = cb + per;
    if (ce > C) ce = C;
    const long po = (long)r * W + c;
    const float* xp = x + (long)n * x_bs + (valid ? po : 0);
    float* op = out + (long)n * out_bs + po;
    const float* sp = s + (long)n * C;
    const int Hp = H >> 1, Wp = W >> 1;
    const bool pw = pooled != nullptr && valid && !(tr & 1) && !(tc & 1) && (r >> 1) < Hp && (c >> 1) < Wp;
    float* pp = pooled ? pooled + (long)n * pooled_bs + (long)(r >> 1) * Wp + (c >> 1) : nullptr;
    int ch = cb;
    for (; ch + 3 < ce; ch += 4) {  // four loads in flight
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = xp[(long)(ch + u) * P];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xv = valid ? v[u] : -INFINITY;
            if (valid) op[(long)(ch + u) * P] = xv * sp[ch + u] * g;
            if (pooled) {  // uniform
                float mxv = fmaxf(xv, __shfl_xor(xv, 1, 64));   // the column neighbour
                mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));      // the row below (tile rows are 32 lanes apart)
                if (pw) pp[(long)(ch + u) * Hp * Wp] = mxv;
            }
        }
    }
    for (; ch < ce; ++ch) {
        const float xv = valid ? xp[(long)ch * P] : -INFINITY;
        if (valid) op[(long)ch * P] = xv * sp[ch] * g;
        if (pooled) {
            float mxv = fmaxf(xv, __shfl_xor(xv, 1, 64));
            mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
            if (pw) pp[(long)ch * Hp * Wp] = mxv;
        }
    }
}

// ===================================== backward ======================================
// B1: dgate[n][p] = sum_c dout*x*s ; dbn = dgate * m * (1-m); BN(1) backward partials
//     (sum dbn, sum dbn*xhat), xhat = (conv - mean) * invstd.   part[2][nblocks]
template <typename T>
__global__ __launch_bounds__(256) void k_cbam_bwd_gate(const T* __restrict__ dout, long dout_bs,
                                                       const T* __restrict__ x, long x_bs,
                                                       const float* __restrict__ s, const float* __restrict__ gate,
                                                       const float* __restrict__ conv,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, int C, int P,
                                                       float* __restrict__ dbn, float* __restrict__ part,
                                                       int nblocks) {
    __shared__ float red[8];
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float d = 0.f, dx = 0.f;
    if (p < P) {
        const T* xp = x + (long)n * x_bs + p;
        const T* gp = dout + (long)n * dout_bs + p;
        const float* sp = s + (long)n * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = fmaf(ld1(gp + (long)c * P) * ld1(xp + (long)c * P), sp[c], acc);
        const float m = gate[(long)n * P + p];
        d = acc * m * (1.f - m);
        dbn[(long)n * P + p] = d;
        dx = d * (conv[(long)n * P + p] - mean[0]) * invstd[0];
    }
    const float t1 = block_sum_t0(d, red);
    const float t2 = block_sum_t0(dx, red + 4);
    if (threadIdx.x == 0) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        part[blk] = t1;
        part[nblocks + blk] = t2;
    }
}

// B2: dconv = c1*(dbn - c2 - xhat*c3) on the fly (staged with halo); transposed conv ->
//     dmaps[n][2][p]; conv-weight gradient partials wpart[nblocks][2*ks*ks]
__global__ __launch_bounds__(256) void k_cbam_bwd_spconv(const float* __restrict__ dbn,
                                                         const float* __restrict__ conv,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ coef,
                                                         const float* __restrict__ maps,
                                                         const float* __restrict__ wc, int ks, int H, int W,
                                                         float* __restrict__ dmaps, float* __restrict__ wpart) {
    __shared__ float tile[SPT + 6][SPT + 6 + 1];
    __shared__ float wl[2 * 49];
    __shared__ float red[4 * 2 * 49];
    const int n = blockIdx.z, P = H * W;
    const int r0 = blockIdx.y * SPT, c0 = blockIdx.x * SPT;
    const int pd = ks >> 1, R = SPT + 2 * pd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float mu = mean[0], is = invstd[0], c1 = coef[0], c2 = coef[1], c3 = coef[2];
    if (tid < 2 * ks * ks) wl[tid] = wc[tid];
    for (int e = tid; e < R * R; e += 256) {
        const int sr = e / R, sc = e - sr * R;
        const int gr = r0 - pd + sr, gc = c0 - pd + sc;
        float v = 0.f;
        if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
            const long o = (long)n * P + gr * W + gc;
            v = c1 * (dbn[o] - c2 - (conv[o] - mu) * is * c3);
        }
        tile[sr][sc] = v;
    }
    __syncthreads();
    const int tr = tid / SPT, tc = tid - tr * SPT;
    const int r = r0 + tr, c = c0 + tc;
    const bool valid = r < H && c < W;
    float m0 = 0.f, m1 = 0.f;
    if (valid) {
        m0 = maps[((long)n * 2 + 0) * P + r * W + c];
        m1 = maps[((long)n * 2 + 1) * P + r * W + c];
    }
    float d0 = 0.f, d1 = 0.f;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    // tap (i,j): forward read maps at q+(i-pd, j-pd); so dconv at q' - (i-pd, j-pd) pairs with maps[q']
    for (int i = 0; i < ks; ++i) {
        for (int j = 0; j < ks; ++j) {
            const float d = valid ? tile[tr + 2 * pd - i][tc + 2 * pd - j] : 0.f;
            d0 = fmaf(wl[(0 * ks + i) * ks + j], d, d0);
            d1 = fmaf(wl[(1 * ks + i) * ks + j], d, d1);
            const float a0 = wave_sum_l63(m0 * d);
            const float a1 = wave_sum_l63(m1 * d);
            if (lane == 63) {
                red[(wave * 2 + 0) * 49 + i * ks + j] = a0;
                red[(wave * 2 + 1) * 49 + i * ks + j] = a1;
            }
        }
    }
    if (valid) {
        dmaps[((long)n * 2 + 0) * P + r * W + c] = d0;
        dmaps[((long)n * 2 + 1) * P + r * W + c] = d1;
    }
    __syncthreads();
    if (tid < 2 * ks * ks) {
        const int ch = tid / (ks * ks), t = tid - ch * ks * ks;
        const float v = red[(0 * 2 + ch) * 49 + t] + red[(1 * 2 + ch) * 49 + t] + red[(2 * 2 + ch) * 49 + t] +
                        red[(3 * 2 + ch) * 49 + t];
        wpart[(long)blk * (2 * ks * ks) + tid] = v;
    }
}

// B3: per pixel, loop over channels:
//     xs = x*s ; dxs = dout*gate + dmaps0/C + [c == first argmax_c xs] * dmaps1
//     t[n][c][p] = dxs * s         (main part of dx, written to dx)
//     dspart[blk][n][c] = sum_p dxs * x          (-> ds[n][c] after k_reduce_rows over blk)
template <typename T>
__global__ __launch_bounds__(256) void k_cbam_bwd_main(const T* __restrict__ dout, long dout_bs,
                                                       const T* __restrict__ x, long x_bs,
                                                       const float* __restrict__ s, const float* __restrict__ gate,
                                                       const float* __restrict__ maps,
                                                       const float* __restrict__ dmaps, int C, int P,
                                                       T* __restrict__ dx, long dx_bs,
                                                       float* __restrict__ dspart) {
    extern __shared__ float red[];  // [4][C]
    const int n = blockIdx.y, N = gridDim.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = p < P;
    const int pp = valid ? p : P - 1;
    const T* xp = x + (long)n * x_bs + pp;
    const T* gp = dout + (long)n * dout_bs + pp;
    T* dp = dx + (long)n * dx_bs + pp;
    const float* sp = s + (long)n * C;
    const float g = gate[(long)n * P + pp];
    const float mxv = maps[((long)n * 2 + 1) * P + pp];
    const float da = dmaps[((long)n * 2 + 0) * P + pp] / (float)C;
    const float dm = dmaps[((long)n * 2 + 1) * P + pp];
    bool found = false;
    for (int c = 0; c < C; ++c) {
        const float xv = ld1(xp + (long)c * P), sv = sp[c];
        const float xs = xv * sv;
        float dxs = fmaf(ld1(gp + (long)c * P), g, da);
        if (!found && xs == mxv) {
            dxs += dm;
            found = true;
        }
        if (valid) st1(dp + (long)c * P, dxs * sv);
        const float r = wave_sum_l63(valid ? dxs * xv : 0.f);
        if (lane == 63) red[wave * C + c] = r;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        dspart[((long)blockIdx.x * N + n) * C + c] = red[c] + red[C + c] + red[2 * C + c] + red[3 * C + c];
}

// ---------------------------------------------------------------------------------
// float4 forms of the three pixel-major CBAM passes (P % 4 == 0, 16-byte aligned planes): ONE wave per
// workgroup covers the same 256 pixels as a 256-thread block of the scalar kernels (same partial-result
// slots), each lane owns 4 consecutive pixels: 16-byte loads/stores and one wave reduction per channel
// instead of four.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_cbam_sppool_v4(const T* __restrict__ x, long x_bs,
                                                       const float* __restrict__ s, int C, int P,
                                                       float* __restrict__ maps) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x * 4;
    if (p >= P) return;
    const T* xp = x + (long)n * x_bs + p;
    const float* sp = s + (long)n * C;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float4 v = ld4(xp + (long)c * P);
        const float sv = sp[c];
        const float a = v.x * sv, b = v.y * sv, cc = v.z * sv, d = v.w * sv;
        sum.x += a; sum.y += b; sum.z += cc; sum.w += d;
        m.x = fmaxf(m.x, a); m.y = fmaxf(m.y, b); m.z = fmaxf(m.z, cc); m.w = fmaxf(m.w, d);
    }
    const float ic = (float)C;
    *(float4*)(maps + ((long)n * 2 + 0) * P + p) = make_float4(sum.x / ic, sum.y / ic, sum.z / ic, sum.w / ic);
    *(float4*)(maps + ((long)n * 2 + 1) * P + p) = m;
}

template <typename T>
__global__ __launch_bounds__(64) void k_cbam_bwd_gate_v4(const T* __restrict__ dout, long dout_bs,
                                                         const T* __restrict__ x, long x_bs,
                                                         const float* __restrict__ s, const float* __restrict__ gate,
                                                         const float* __restrict__ conv,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int C, int P,
                                                         float* __restrict__ dbn, float* __restrict__ part,
                                                         int nblocks) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x * 4;
    float t1 = 0.f, t2 = 0.f;
    if (p < P) {
        const T* xp = x + (long)n * x_bs + p;
        const T* gp = dout + (long)n * dout_bs + p;
        const float* sp = s + (long)n * C;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float4 g = ld4(gp + (long)c * P);
            const float4 v = ld4(xp + (long)c * P);
            const float sv = sp[c];
            acc.x = fmaf(g.x * v.x, sv, acc.x);
            acc.y = fmaf(g.y * v.y, sv, acc.y);
            acc.z = fmaf(g.z * v.z, sv, acc.z);
            acc.w = fmaf(g.w * v.w, sv, acc.w);
        }
        const float4 m = *(const float4*)(gate + (long)n * P + p);
        const float4 cv = *(const float4*)(conv + (long)n * P + p);
        const float mu = mean[0], is = invstd[0];
        float4 d;
        d.x = acc.x * m.x * (1.f - m.x);
        d.y = acc.y * m.y * (1.f - m.y);
        d.z = acc.z * m.z * (1.f - m.z);
        d.w = acc.w * m.w * (1.f - m.w);
        *(float4*)(dbn + (long)n * P + p) = d;
        t1 = (d.x + d.y) + (d.z + d.w);
        t2 = (d.x * (cv.x - mu) + d.y * (cv.y - mu) + d.z * (cv.z - mu) + d.w * (cv.w - mu)) * is;
    }
    t1 = wave_sum_l63(t1);
    t2 = wave_sum_l63(t2);
    if (threadIdx.x == 63) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        part[blk] = t1;
        part[nblocks + blk] = t2;
    }
}

template <typename T>
__global__ __launch_bounds__(64) void k_cbam_bwd_main_v4(const T* __restrict__ dout, long dout_bs,
                                                         const T* __restrict__ x, long x_bs,
                                                         const float* __restrict__ s, const float* __restrict__ gate,
                                                         const float* __restrict__ maps,
                                                         const float* __restrict__ dmaps, int C, int P,
                                                         T* __restrict__ dx, long dx_bs,
                                                         float* __restrict__ dspart) {
    const int n = blockIdx.y, N = gridDim.y;
    const int p = blockIdx.x * 256 + threadIdx.x * 4;
    const bool valid = p < P;
    const int pp = valid ? p : 0;
    constexpr unsigned EB = Elem<T>::bytes;
    const T* xp = x + (long)n * x_bs + pp;
    const T* gp = dout + (long)n * dout_bs + pp;
    const float* sp = s + (long)n * C;
    const float4 g = *(const float4*)(gate + (long)n * P + pp);
    const float4 mxv = *(const float4*)(maps + ((long)n * 2 + 1) * P + pp);
    float4 da = *(const float4*)(dmaps + ((long)n * 2 + 0) * P + pp);
    const float4 dm = *(const float4*)(dmaps + ((long)n * 2 + 1) * P + pp);
    const float ic = 1.f / (float)C;
    da.x /= (float)C; da.y /= (float)C; da.z /= (float)C; da.w /= (float)C;
    (void)ic;
    bool f0 = false, f1 = false, f2 = false, f3 = false;
    float* dsrow = dspart + ((long)blockIdx.x * N + n) * C;
    // The channel loop is a dependent chain per iteration (load x, dOut -> arithmetic -> store dX -> wave sum),
    // so it is software-pipelined by hand: the loads of channel c+1 are issued before channel c is processed,
    // and the body has no branches (the first-maximum rule as selects; the stores through buffer descriptors
    // whose range check drops the lanes that must not write) -- a divergent branch or a load between the
    // stores makes hipcc drain the prefetch with vmcnt(0).
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dx + (long)n * dx_bs, 0, C * P * (int)EB, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(dsrow, 0, C * 4, 0x00020000);
    const unsigned dvo = valid ? (unsigned)pp * EB : 0x80000000u;
    const unsigned svo = threadIdx.x == 63 ? 0u : 0x80000000u;
    // two channels per trip, the loads of the NEXT two in flight meanwhile (four 16-byte loads per lane); a channel
    // index beyond C reads clamped addresses and its stores fall outside the buffer ranges (dropped by the hardware)
    typedef typename Elem<T>::raw4 R4;
    auto ld = [&](int c, R4& xv, R4& gv, float& sv) {
        const int cc = c < C ? c : C - 1;
        xv = ldraw4(xp + (long)cc * P);
        gv = ldraw4(gp + (long)cc * P);
        sv = sp[cc];
    };
    auto process = [&](const R4 xr, const R4 gr, const float sv, int c) {
        const float4 xv = cvt4(xr), gv = cvt4(gr);
        float4 dxs;
        dxs.x = fmaf(gv.x, g.x, da.x);
        dxs.y = fmaf(gv.y, g.y, da.y);
        dxs.z = fmaf(gv.z, g.z, da.z);
        dxs.w = fmaf(gv.w, g.w, da.w);
        const bool h0 = !f0 && xv.x * sv == mxv.x, h1 = !f1 && xv.y * sv == mxv.y;
        const bool h2 = !f2 && xv.z * sv == mxv.z, h3 = !f3 && xv.w * sv == mxv.w;
        dxs.x += h0 ? dm.x : 0.f;
        dxs.y += h1 ? dm.y : 0.f;
        dxs.z += h2 ? dm.z : 0.f;
        dxs.w += h3 ? dm.w : 0.f;
        f0 |= h0;
        f1 |= h1;
        f2 |= h2;
        f3 |= h3;
        if constexpr (EB == 4) {
            u4 o;
            o.x = __builtin_bit_cast(unsigned, dxs.x * sv);
            o.y = __builtin_bit_cast(unsigned, dxs.y * sv);
            o.z = __builtin_bit_cast(unsigned, dxs.z * sv);
            o.w = __builtin_bit_cast(unsigned, dxs.w * sv);
            __builtin_amdgcn_raw_buffer_store_b128(o, drs, dvo + (unsigned)c * (unsigned)P * 4u, 0, 0);
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 o;
            o.x = pack_bf16x2(dxs.x * sv, dxs.y * sv);
            o.y = pack_bf16x2(dxs.z * sv, dxs.w * sv);
            __builtin_amdgcn_raw_buffer_store_b64(o, drs, dvo + (unsigned)c * (unsigned)P * 2u, 0, 0);
        }
        float r = valid ? (dxs.x * xv.x + dxs.y * xv.y) + (dxs.z * xv.z + dxs.w * xv.w) : 0.f;
        r = wave_sum_l63(r);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), srs, svo + (unsigned)c * 4u, 0, 0);
    };
    R4 xa, ga, xb, gb;
    float sa, sb;
    ld(0, xa, ga, sa);
    ld(1, xb, gb, sb);
    for (int c = 0; c < C; c += 2) {
        const R4 x0 = xa, g0 = ga, x1 = xb, g1 = gb;
        const float s0 = sa, s1 = sb;
        ld(c + 2, xa, ga, sa);
        ld(c + 3, xb, gb, sb);
        process(x0, g0, s0, c);
        // (c + 1 == C for an odd channel count: a clamped duplicate of the last channel whose stores are dropped; the
        // first-maximum flags it may set are not read again)
        process(x1, g1, s1, c + 1 < C ? c + 1 : C);
    }
}

// B4: MLP backward, one block per sample.  Per-sample parameter-gradient partials:
//   pg[n][ C*Cr (dW2) | C (db2) | Cr*C (dW1) | Cr (db1) ]  -> reduced over n by k_reduce_rows
//   davg[n][c], dmx[n][c]
__global__ __launch_bounds__(256) void k_cbam_bwd_mlp(const float* __restrict__ ds, const float* __restrict__ s,
                                                      const float* __restrict__ avg, const float* __restrict__ mx,
                                                      const float* __restrict__ ha, const float* __restrict__ hm,
                                                      const float* __restrict__ w1, const float* __restrict__ w2,
                                                      int C, int Cr, float* __restrict__ pg,
                                                      float* __restrict__ davg, float* __restrict__ dmx) {
    extern __shared__ float sm[];
    float* dout = sm;        // [C]
    float* dha = dout + C;   // [Cr]
    float* dhm = dha + Cr;   // [Cr]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long pgs = (long)C * Cr + C + (long)Cr * C + Cr;
    float* pgn = pg + (long)n * pgs;
    float* dW2 = pgn;
    float* db2 = dW2 + (long)C * Cr;
    float* dW1 = db2 + C;
    float* db1 = dW1 + (long)Cr * C;
    for (int c = tid; c < C; c += 256) {
        const float sv = s[(long)n * C + c];
        const float d = ds[(long)n * C + c] * sv * (1.f - sv);
        dout[c] = d;
        db2[c] = 2.f * d;
        for (int j = 0; j < Cr; ++j) dW2[(long)c * Cr + j] = d * (ha[(long)n * Cr + j] + hm[(long)n * Cr + j]);
    }
    __syncthreads();
    for (int j = wave; j < Cr; j += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(dout[c], w2[(long)c * Cr + j], a);
        a = wave_sum_all(a);
        if (lane == 0) {
            const float va = ha[(long)n * Cr + j] > 0.f ? a : 0.f;
            const float vm = hm[(long)n * Cr + j] > 0.f ? a : 0.f;
            dha[j] = va;
            dhm[j] = vm;
            db1[j] = va + vm;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float av = avg[(long)n * C + c], mv = mx[(long)n * C + c];
        float da = 0.f, dm = 0.f;
        for (int j = 0; j < Cr; ++j) {
            const float w = w1[(long)j * C + c];
            da = fmaf(dha[j], w, da);
            dm = fmaf(dhm[j], w, dm);
            dW1[(long)j * C + c] = dha[j] * av + dhm[j] * mv;
        }
        davg[(long)n * C + c] = da;
        dmx[(long)n * C + c] = dm;
    }
}

// B5: dx[n][c][p] += davg[n][c]/P ; dx[n][c][amax[n][c]] += dmx[n][c]     (in place)
template <typename T>
__global__ __launch_bounds__(256) void k_cbam_bwd_final(T* __restrict__ dx, long dx_bs,
                                                        const float* __restrict__ davg,
                                                        const float* __restrict__ dmx, const int* __restrict__ amax,
                                                        int C, int P, int seg_len) {
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float add = davg[plane] / (float)P;
    const float dm = dmx[plane];
    const int am = amax[plane];
    T* dp = dx + (long)n * dx_bs + (long)c * P;
    const int p0 = blockIdx.y * seg_len;
    int p1 = p0 + seg_len;
    if (p1 > P) p1 = P;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        float v = ld1(dp + p) + add;
        if (p == am) v += dm;
        st1(dp + p, v);
    }
}

// B5 + the backward of the MaxPool2d(2) that reads the same tensor (encoder levels: x -> CBAM(x) for the skip and
// x -> maxpool -> next block; reference SmaAt_UNet.py:43-50), in ONE read-modify-write pass over dX:
//   dx[n][c][p] += davg[n][c]/P + [p == amax[n][c]] * dmx[n][c] + [p is the first maximum of its 2x2 window] * dpool
// instead of two passes (k_cbam_bwd_final, k_maxpool2_bwd with accum = 1).  A thread owns a 2 x 4 patch (two windows):
// float4 accesses, the window maximum taken in the scan order of k_maxpool2_bwd.  W % 4 == 0; a last odd row has no
// window (floor mode) and only receives the channel-attention terms.
template <typename T>
__global__ __launch_bounds__(256) void k_cbam_final_pool_bwd(T* __restrict__ dx, long dx_bs,
                                                             const float* __restrict__ davg,
                                                             const float* __restrict__ dmx,
                                                             const int* __restrict__ amax,
                                                             const T* __restrict__ x, long x_bs,
                                                             const T* __restrict__ dpool, long dp_bs, int C, int H,
                                                             int W, long total) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ncol4 = W >> 2, npair = (H + 1) >> 1;
    const int per = ncol4 * npair;
    const int plane = (int)(gid / per), rem = (int)(gid - (long)plane * per);
    const int i = rem / ncol4, q = rem - i * ncol4;
    const int n = plane / C, c = plane - n * C;
    const int P = H * W, Wo = W >> 1, Ho = H >> 1;
    const float add = davg[plane] / (float)P;
    const float dm = dmx[plane];
    const int am = amax[plane];
    const T* xp = x + (long)n * x_bs + (long)c * P;
    T* dp = dx + (long)n * dx_bs + (long)c * P;
    const int r0 = 2 * i, p0 = r0 * W + 4 * q;
    const bool two = r0 + 1 < H;  // (i < Ho)
    float4 d0 = ld4(dp + p0);
    float a0[4] = {d0.x + add, d0.y + add, d0.z + add, d0.w + add};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (p0 + k == am) a0[k] += dm;
    if (two) {
        const float4 x0 = ld4(xp + p0), x1 = ld4(xp + p0 + W);
        float4 d1 = ld4(dp + p0 + W);
        const float2 g = ld2(dpool + (long)n * dp_bs + (long)c * Ho * Wo + (long)i * Wo + 2 * q);
        float a1[4] = {d1.x + add, d1.y + add, d1.z + add, d1.w + add};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (p0 + W + k == am) a1[k] += dm;
        const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
        const float gg[2] = {g.x, g.y};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const float v0 = xa[2 * w], v1 = xa[2 * w + 1], v2 = xb[2 * w], v3 = xb[2 * w + 1];
            int sel = 0;
            float m = v0;
            if (v1 > m) { m = v1; sel = 1; }
            if (v2 > m) { m = v2; sel = 2; }
            if (v3 > m) { m = v3; sel = 3; }
            a0[2 * w] += sel == 0 ? gg[w] : 0.f;
            a0[2 * w + 1] += sel == 1 ? gg[w] : 0.f;
            a1[2 * w] += sel == 2 ? gg[w] : 0.f;
            a1[2 * w + 1] += sel == 3 ? gg[w] : 0.f;
        }
        st4(dp + p0 + W, make_float4(a1[0], a1[1], a1[2], a1[3]));
    }
    st4(dp + p0, make_float4(a0[0], a0[1], a0[2], a0[3]));
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: the backward of a level's attention in THREE passes over the level's tensors instead of the gate / main / final
// passes above (2 + 3 + 3.25 tensor streams -> 2 + 1 + 3.25: `dx` is written ONCE, complete):
//   ds[n][c] = sum_p dxs * x,  dxs = dout * gate + dmaps0 / C + [c first argmax] dmaps1   (k_cbam_bwd_main)
//            = sum_p (dout * gate) * x              <- k_cbam_bwd_gate_ds_v4: the gate pass reads dout and x anyway
//            + sum_p (dmaps0 / C + [..] dmaps1) * x <- k_cbam_bwd_ds2_v4: x alone, after the spatial branch's backward
//   so the shared MLP's backward (davg, dmx) is known BEFORE dx is formed, and k_cbam_bwd_apply_v4 writes
//   dx = dxs * s + davg / P + [p == amax] dmx + [p first maximum of its 2 x 2 window] dpool
//   (the terms of k_cbam_bwd_main + k_cbam_final_pool_bwd in their order: f32 dx is bit-identical given the same ds;
//   bf16 storage rounds once instead of twice) without the read-modify-write pass over it.
// The pixel-major passes above walk ALL channels in one wave: a dependent chain of C / 2 trips of ~1.5 us each, which is what
// the deep levels cost (profiles/r5: 36 x 36 x 512 channels: 350 us for 0.28 GB; 72 x 72 x 256: 230 us for 0.55 GB) while the
// 288 x 288 and 144 x 144 levels stream at 5 TB/s.  Here a workgroup is CS = blockDim.x / 64 waves that share the block's 256
// pixels and SPLIT THE CHANNELS (contiguous ranges, in wave order).  That needs the "first argmax over channels" of the spatial
// max-pool as data instead of as a scan-order flag: k_cbam_sppool_idx_v4 (forward) leaves amaxc[n][p], the index
// k_cbam_bwd_main finds by its scan (the first channel whose x * s equals the maximum).
// reference: the autograd of models/layers.py:105-111, 122-129, 138-141 + the MaxPool2d of unet_parts_depthwise_separable.py:48
// ---------------------------------------------------------------------------------------------------------------
// channels [c_lo, c_hi) of this wave: even-sized contiguous chunks in wave order (possibly empty)
__device__ __forceinline__ void cbam_chan_range(int C, int& c_lo, int& c_hi) {
    const int cs = (int)(blockDim.x >> 6), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunk = (((C + cs - 1) / cs) + 1) & ~1;
    c_lo = w * chunk < C ? w * chunk : C;
    c_hi = c_lo + chunk < C ? c_lo + chunk : C;
}

// k_cbam_sppool_v4 + amaxc[n][p] = the first channel that attains the maximum.  CS == 1: maps bit-identical to
// k_cbam_sppool_v4; CS > 1: the mean adds the waves' partial sums in wave order (the maximum and its index do not depend on it).
template <typename T>
__global__ __launch_bounds__(512) void k_cbam_sppool_idx_v4(const T* __restrict__ x, long x_bs, const float* __restrict__ s,
                                                            int C, int P, float* __restrict__ maps, int* __restrict__ amaxc) {
    extern __shared__ float4 sp_red[];  // [3][CS][64] (sum, max, index) when CS > 1
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int cs = (int)(blockDim.x >> 6), w = (int)(threadIdx.x >> 6);
    const int p = blockIdx.x * 256 + lane * 4;
    const bool valid = p < P;
    const int pp = valid ? p : 0;
    const T* xp = x + (long)n * x_bs + pp;
    const float* sp = s + (long)n * C;
    int c_lo, c_hi;
    cbam_chan_range(C, c_lo, c_hi);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int i0 = c_lo, i1 = c_lo, i2 = c_lo, i3 = c_lo;
#pragma unroll 4
    for (int c = c_lo; c < c_hi; ++c) {
        const float4 v = ld4(xp + (long)c * P);
        const float sv = sp[c];
        const float a = v.x * sv, b = v.y * sv, cc = v.z * sv, d = v.w * sv;
        sum.x += a; sum.y += b; sum.z += cc; sum.w += d;
        i0 = a > m.x ? c : i0; m.x = a > m.x ? a : m.x;
        i1 = b > m.y ? c : i1; m.y = b > m.y ? b : m.y;
        i2 = cc > m.z ? c : i2; m.z = cc > m.z ? cc : m.z;
        i3 = d > m.w ? c : i3; m.w = d > m.w ? d : m.w;
    }
    if (cs > 1) {
        sp_red[(0 * cs + w) * 64 + lane] = sum;
        sp_red[(1 * cs + w) * 64 + lane] = m;
        sp_red[(2 * cs + w) * 64 + lane] = make_float4(__int_as_float(i0), __int_as_float(i1), __int_as_float(i2), __int_as_float(i3));
        __syncthreads();
        if (w != 0) return;
        for (int k = 1; k < cs; ++k) {
            const float4 s2 = sp_red[(0 * cs + k) * 64 + lane], m2 = sp_red[(1 * cs + k) * 64 + lane];
            const float4 j2 = sp_red[(2 * cs + k) * 64 + lane];
            sum.x += s2.x; sum.y += s2.y; sum.z += s2.z; sum.w += s2.w;
            i0 = m2.x > m.x ? __float_as_int(j2.x) : i0; m.x = m2.x > m.x ? m2.x : m.x;
            i1 = m2.y > m.y ? __float_as_int(j2.y) : i1; m.y = m2.y > m.y ? m2.y : m.y;
            i2 = m2.z > m.z ? __float_as_int(j2.z) : i2; m.z = m2.z > m.z ? m2.z : m.z;
            i3 = m2.w > m.w ? __float_as_int(j2.w) : i3; m.w = m2.w > m.w ? m2.w : m.w;
        }
    }
    if (!valid) return;
    const float ic = (float)C;
    *(float4*)(maps + ((long)n * 2 + 0) * P + p) = make_float4(sum.x / ic, sum.y / ic, sum.z / ic, sum.w / ic);
    *(float4*)(maps + ((long)n * 2 + 1) * P + p) = m;
    *(int4*)(amaxc + (long)n * P + p) = make_int4(i0, i1, i2, i3);
}

// k_cbam_bwd_gate_v4 + dspart[blockIdx.x][n][c] = sum over the block's 256 pixels of (dout * gate) * x.
// CS == 1: dbn and its partials bit-identical to k_cbam_bwd_gate_v4; CS > 1: the waves' channel-range sums are added in wave order.
template <typename T>
__global__ __launch_bounds__(512) void k_cbam_bwd_gate_ds_v4(const T* __restrict__ dout, long dout_bs,
                                                             const T* __restrict__ x, long x_bs,
                                                             const float* __restrict__ s, const float* __restrict__ gate,
                                                             const float* __restrict__ conv,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int C, int P,
                                                             float* __restrict__ dbn, float* __restrict__ part,
                                                             int nblocks, float* __restrict__ dspart) {
    extern __shared__ float4 gd_red[];  // [CS][64] when CS > 1
    const int n = blockIdx.y, N = gridDim.y, lane = threadIdx.x & 63;
    const int cs = (int)(blockDim.x >> 6), w = (int)(threadIdx.x >> 6);
    const int p = blockIdx.x * 256 + lane * 4;
    const bool valid = p < P;
    const int pp = valid ? p : 0;
    const T* xp = x + (long)n * x_bs + pp;
    const T* gp = dout + (long)n * dout_bs + pp;
    const float* sp = s + (long)n * C;
    const float4 m = *(const float4*)(gate + (long)n * P + pp);
    float* dsrow = dspart + ((long)blockIdx.x * N + n) * C;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(dsrow, 0, C * 4, 0x00020000);
    const unsigned svo = lane == 63 ? 0u : 0x80000000u;
    int c_lo, c_hi;
    cbam_chan_range(C, c_lo, c_hi);
    typedef typename Elem<T>::raw4 R4;
    auto ld = [&](int c, R4& xv, R4& gv, float& sv) {
        const int cc = c < C ? c : C - 1;
        xv = ldraw4(xp + (long)cc * P);
        gv = ldraw4(gp + (long)cc * P);
        sv = sp[cc];
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto process = [&](const R4 xr, const R4 gr, const float sv, int c) {  // (c == C: a clamped duplicate, nothing kept)
        const float4 v = cvt4(xr), g = cvt4(gr);
        const bool keep = c < C;
        const float a0 = fmaf(g.x * v.x, sv, acc.x), a1 = fmaf(g.y * v.y, sv, acc.y);
        const float a2 = fmaf(g.z * v.z, sv, acc.z), a3 = fmaf(g.w * v.w, sv, acc.w);
        acc.x = keep ? a0 : acc.x;
        acc.y = keep ? a1 : acc.y;
        acc.z = keep ? a2 : acc.z;
        acc.w = keep ? a3 : acc.w;
        // (explicit fma placement: left to the compiler, the contraction of a * b + c * d may differ between the f32 and the
        // bf16 instantiation, and the two are tested bit for bit against each other)
        float r = valid ? fmaf(g.x * m.x, v.x, (g.y * m.y) * v.y) + fmaf(g.z * m.z, v.z, (g.w * m.w) * v.w) : 0.f;
        r = wave_sum_l63(r);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), srs, svo + (unsigned)c * 4u, 0, 0);
    };
    R4 xa, ga, xb, gb;
    float sa, sb;
    ld(c_lo, xa, ga, sa);
    ld(c_lo + 1, xb, gb, sb);
    for (int c = c_lo; c < c_hi; c += 2) {
        const R4 x0 = xa, g0 = ga, x1 = xb, g1 = gb;
        const float s0 = sa, s1 = sb;
        ld(c + 2, xa, ga, sa);
        ld(c + 3, xb, gb, sb);
        process(x0, g0, s0, c);
        process(x1, g1, s1, c + 1 < c_hi ? c + 1 : C);
    }
    if (cs > 1) {
        gd_red[w * 64 + lane] = acc;
        __syncthreads();
        if (w != 0) return;
        for (int k = 1; k < cs; ++k) {
            const float4 a2 = gd_red[k * 64 + lane];
            acc.x += a2.x; acc.y += a2.y; acc.z += a2.z; acc.w += a2.w;
        }
    }
    float t1 = 0.f, t2 = 0.f;
    if (valid) {
        const float4 cv = *(const float4*)(conv + (long)n * P + p);
        const float mu = mean[0], is = invstd[0];
        float4 d;
        d.x = acc.x * m.x * (1.f - m.x);
        d.y = acc.y * m.y * (1.f - m.y);
        d.z = acc.z * m.z * (1.f - m.z);
        d.w = acc.w * m.w * (1.f - m.w);
        *(float4*)(dbn + (long)n * P + p) = d;
        t1 = (d.x + d.y) + (d.z + d.w);
        t2 = (d.x * (cv.x - mu) + d.y * (cv.y - mu) + d.z * (cv.z - mu) + d.w * (cv.w - mu)) * is;
    }
    t1 = wave_sum_l63(t1);
    t2 = wave_sum_l63(t2);
    if (lane == 63) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        part[blk] = t1;
        part[nblocks + blk] = t2;
    }
}

// dspart[blockIdx.x][n][c] = sum over the block's 256 pixels of (dmaps0 / C + [c == amaxc] dmaps1) * x
template <typename T>
__global__ __launch_bounds__(512) void k_cbam_bwd_ds2_v4(const T* __restrict__ x, long x_bs, const float* __restrict__ dmaps,
                                                         const int* __restrict__ amaxc, int C, int P,
                                                         float* __restrict__ dspart) {
    const int n = blockIdx.y, N = gridDim.y, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 256 + lane * 4;
    const bool valid = p < P;
    const int pp = valid ? p : 0;
    const T* xp = x + (long)n * x_bs + pp;
    const int4 am = *(const int4*)(amaxc + (long)n * P + pp);
    float4 da = *(const float4*)(dmaps + ((long)n * 2 + 0) * P + pp);
    const float4 dm = *(const float4*)(dmaps + ((long)n * 2 + 1) * P + pp);
    da.x /= (float)C; da.y /= (float)C; da.z /= (float)C; da.w /= (float)C;
    float* dsrow = dspart + ((long)blockIdx.x * N + n) * C;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(dsrow, 0, C * 4, 0x00020000);
    const unsigned svo = lane == 63 ? 0u : 0x80000000u;
    int c_lo, c_hi;
    cbam_chan_range(C, c_lo, c_hi);
    typedef typename Elem<T>::raw4 R4;
    auto ld = [&](int c, R4& xv) {
        const int cc = c < C ? c : C - 1;
        xv = ldraw4(xp + (long)cc * P);
    };
    auto process = [&](const R4 xr, int c) {
        const float4 xv = cvt4(xr);
        const float e0 = da.x + (am.x == c ? dm.x : 0.f), e1 = da.y + (am.y == c ? dm.y : 0.f);
        const float e2 = da.z + (am.z == c ? dm.z : 0.f), e3 = da.w + (am.w == c ? dm.w : 0.f);
        float r = valid ? fmaf(e0, xv.x, e1 * xv.y) + fmaf(e2, xv.z, e3 * xv.w) : 0.f;
        r = wave_sum_l63(r);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), srs, svo + (unsigned)c * 4u, 0, 0);
    };
    // four channels in flight (one 16-byte load each), two per trip
    R4 xa, xb, xc, xd;
    ld(c_lo, xa);
    ld(c_lo + 1, xb);
    ld(c_lo + 2, xc);
    ld(c_lo + 3, xd);
    for (int c = c_lo; c < c_hi; c += 2) {
        const R4 x0 = xa, x1 = xb;
        xa = xc;
        xb = xd;
        ld(c + 4, xc);
        ld(c + 5, xd);
        process(x0, c);
        process(x1, c + 1 < c_hi ? c + 1 : C);  // (an odd range: the duplicate's store is dropped)
    }
}

// dx, complete.  POOL: a lane owns a 2 x 4 patch (two pooling windows) of every channel of its wave's range; + the
// MaxPool2d(2) backward (window maximum in the scan order of k_maxpool2_bwd, on x as stored); H even, W % 4 == 0.
// !POOL (the last level: nothing pools it): a lane owns 4 consecutive pixels; P % 4 == 0.
// The per-pixel maps stay in registers over the channel loop, the per-channel scalars (s, davg, dmx, amax) come through the
// scalar cache.
template <typename T, bool POOL>
__global__ __launch_bounds__(512) void k_cbam_bwd_apply_v4(const T* __restrict__ dout, long dout_bs,
                                                           const T* __restrict__ x, long x_bs,
                                                           const float* __restrict__ s, const float* __restrict__ gate,
                                                           const float* __restrict__ dmaps, const int* __restrict__ amaxc,
                                                           const float* __restrict__ davg, const float* __restrict__ dmx,
                                                           const int* __restrict__ amax, const T* __restrict__ dpool,
                                                           long dp_bs, int C, int H, int W, T* __restrict__ dx, long dx_bs) {
    constexpr int R = POOL ? 2 : 1;
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int P = H * W, Wo = W >> 1, Po = (H >> 1) * Wo;
    int p0, ipl = 0;
    bool valid;
    if (POOL) {
        const int ncol4 = W >> 2, per = ncol4 * (H >> 1);
        const int idx = blockIdx.x * 64 + lane;
        valid = idx < per;
        const int idc = valid ? idx : 0;
        const int i = idc / ncol4, q = idc - i * ncol4;
        p0 = 2 * i * W + 4 * q;
        ipl = i * Wo + 2 * q;
    } else {
        const int p = blockIdx.x * 256 + lane * 4;
        valid = p < P;
        p0 = valid ? p : 0;
    }
    constexpr unsigned EB = Elem<T>::bytes;
    const T* xp = x + (long)n * x_bs + p0;
    const T* gp = dout + (long)n * dout_bs + p0;
    const T* pl = POOL ? dpool + (long)n * dp_bs + ipl : nullptr;
    const float* sp = s + (long)n * C;
    const float* dap = davg + (long)n * C;
    const float* dmp = dmx + (long)n * C;
    const int* amp = amax + (long)n * C;
    float g_[R][4], da_[R][4], dm_[R][4];
    int am_[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int pr = p0 + r * W;
        const float4 g4 = *(const float4*)(gate + (long)n * P + pr);
        const float4 a4 = *(const float4*)(dmaps + ((long)n * 2 + 0) * P + pr);
        const float4 m4 = *(const float4*)(dmaps + ((long)n * 2 + 1) * P + pr);
        const int4 i4 = *(const int4*)(amaxc + (long)n * P + pr);
        g_[r][0] = g4.x; g_[r][1] = g4.y; g_[r][2] = g4.z; g_[r][3] = g4.w;
        da_[r][0] = a4.x / (float)C; da_[r][1] = a4.y / (float)C; da_[r][2] = a4.z / (float)C; da_[r][3] = a4.w / (float)C;
        dm_[r][0] = m4.x; dm_[r][1] = m4.y; dm_[r][2] = m4.z; dm_[r][3] = m4.w;
        am_[r][0] = i4.x; am_[r][1] = i4.y; am_[r][2] = i4.z; am_[r][3] = i4.w;
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dx + (long)n * dx_bs, 0, C * P * (int)EB, 0x00020000);
    const unsigned dvo = valid ? (unsigned)p0 * EB : 0x80000000u;
    int c_lo, c_hi;
    cbam_chan_range(C, c_lo, c_hi);
    typedef typename Elem<T>::raw4 R4;
    struct Ch {
        R4 xr[R], gr[R];
        float2 gg;
        float sv, add, dmv;
        int am;
    };
    auto ld = [&](int c, Ch& k) {
        const int cc = c < C ? c : C - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            k.xr[r] = ldraw4(xp + (long)cc * P + r * W);
            k.gr[r] = ldraw4(gp + (long)cc * P + r * W);
        }
        if constexpr (POOL) k.gg = ld2(pl + (long)cc * Po);
        k.sv = sp[cc];
        k.add = dap[cc];
        k.dmv = dmp[cc];
        k.am = amp[cc];
    };
    auto process = [&](const Ch& k, int c) {
        float xv[R][4], a[R][4];
        const float add = k.add / (float)P;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 x4 = cvt4(k.xr[r]), g4 = cvt4(k.gr[r]);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
            xv[r][0] = x4.x; xv[r][1] = x4.y; xv[r][2] = x4.z; xv[r][3] = x4.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dxs = fmaf(gv[e], g_[r][e], da_[r][e]);
                dxs += am_[r][e] == c ? dm_[r][e] : 0.f;
                float t = dxs * k.sv;
                asm volatile("" : "+v"(t));  // (no contraction with the add: the rounding of k_cbam_bwd_main's stored product)
                float v = t + add;
                v += (p0 + r * W + e == k.am) ? k.dmv : 0.f;
                a[r][e] = v;
            }
        }
        if constexpr (POOL) {
            const float gg[2] = {k.gg.x, k.gg.y};
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const float v0 = xv[0][2 * w], v1 = xv[0][2 * w + 1], v2 = xv[R - 1][2 * w], v3 = xv[R - 1][2 * w + 1];
                int sel = 0;
                float mm = v0;
                if (v1 > mm) { mm = v1; sel = 1; }
                if (v2 > mm) { mm = v2; sel = 2; }
                if (v3 > mm) { mm = v3; sel = 3; }
                a[0][2 * w] += sel == 0 ? gg[w] : 0.f;
                a[0][2 * w + 1] += sel == 1 ? gg[w] : 0.f;
                a[R - 1][2 * w] += sel == 2 ? gg[w] : 0.f;
                a[R - 1][2 * w + 1] += sel == 3 ? gg[w] : 0.f;
            }
        }
        const unsigned off = dvo + (unsigned)c * (unsigned)P * EB;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (EB == 4) {
                u4 o;
                o.x = __builtin_bit_cast(unsigned, a[r][0]);
                o.y = __builtin_bit_cast(unsigned, a[r][1]);
                o.z = __builtin_bit_cast(unsigned, a[r][2]);
                o.w = __builtin_bit_cast(unsigned, a[r][3]);
                __builtin_amdgcn_raw_buffer_store_b128(o, drs, off + (unsigned)(r * W) * 4u, 0, 0);
            } else {
                u2 o;
                o.x = pack_bf16x2(a[r][0], a[r][1]);
                o.y = pack_bf16x2(a[r][2], a[r][3]);
                __builtin_amdgcn_raw_buffer_store_b64(o, drs, off + (unsigned)(r * W) * 2u, 0, 0);
            }
        }
    };
    // two channels per trip, the loads of the NEXT two issued before the first is processed and no load between the stores of a
    // trip (k_cbam_bwd_main_v4: a load between stores makes hipcc drain the prefetch with vmcnt(0) -- the first form of this loop
    // did, and ran at 3.4 TB/s); a channel index >= C reads clamped addresses and its stores fall outside the buffer range
    // (dropped by the hardware)
    Ch ka, kb;
    ld(c_lo, ka);
    ld(c_lo + 1, kb);
    for (int c = c_lo; c < c_hi; c += 2) {
        const Ch k0 = ka, k1 = kb;
        ld(c + 2, ka);
        ld(c + 3, kb);
        process(k0, c);
        process(k1, c + 1 < c_hi ? c + 1 : C);
    }
}

// =====================================================================================
static inline int cdivc(long a, long b) { return (int)((a + b - 1) / b); }
static int seg_len_c(int P) { return P <= 8192 ? ((P + 1023) / 1024) * 1024 : 8192; }

int smaat_cbam_spconv_blocks_impl(int N, int H, int W) { return N * cdivc(H, SPT) * cdivc(W, SPT); }
int smaat_cbam_pix_blocks_impl(int N, int P) { return N * cdivc(P, 256); }

// dt (here and below): SMAAT_F32 | SMAAT_BF16 element type of the activation-sized tensors (x, out, dout, dx, pooled)
int launch_cbam_chpool(const void* x, long x_bs, int N, int C, int P, float* avg, float* mx, int* amax,
                       hipStream_t st, const float* scale, const float* shift, void* y_out, long y_bs, int dt) {
    SMAAT_DISPATCH_ET(dt, T,
        if (scale)
            hipLaunchKernelGGL((k_cbam_chpool<true, T>), dim3(N * C), dim3(256), 0, st, (const T*)x, x_bs, C, P, avg, mx, amax,
                               scale, shift, (T*)y_out, y_bs);
        else
            hipLaunchKernelGGL((k_cbam_chpool<false, T>), dim3(N * C), dim3(256), 0, st, (const T*)x, x_bs, C, P, avg, mx, amax,
                               nullptr, nullptr, (T*)nullptr, 0L););
    return (int)hipGetLastError();
}
// -2: shape / alignment not handled (the caller runs smaat_cbam_chpool[_act] + smaat_maxpool2_fwd)
int launch_cbam_chpool_pool(const void* x, long x_bs, int N, int C, int H, int W, float* avg, float* mx, int* amax,
                            hipStream_t st, const float* scale, const float* shift, void* y_out, long y_bs, void* pooled,
                            long p_bs, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if ((W & 3) != 0 || H < 2 || (x_bs & 3) != 0 || (p_bs & 1) != 0 || ((((uintptr_t)x) & am) != 0) ||
        ((((uintptr_t)pooled) & (am >> 1)) != 0) || (scale && ((y_bs & 3) != 0 || ((((uintptr_t)y_out) & am) != 0))))
        return -2;
    SMAAT_DISPATCH_ET(dt, T,
        if (scale)
            hipLaunchKernelGGL((k_cbam_chpool_pool<true, T>), dim3(N * C), dim3(256), 0, st, (const T*)x, x_bs, C, H, W, avg, mx,
                               amax, scale, shift, (T*)y_out, y_bs, (T*)pooled, p_bs);
        else
            hipLaunchKernelGGL((k_cbam_chpool_pool<false, T>), dim3(N * C), dim3(256), 0, st, (const T*)x, x_bs, C, H, W, avg, mx,
                               amax, nullptr, nullptr, (T*)nullptr, 0L, (T*)pooled, p_bs););
    return (int)hipGetLastError();
}
int launch_cbam_mlp(const float* avg, const float* mx, const float* w1, const float* b1, const float* w2,
                    const float* b2, int N, int C, int Cr, float* ha, float* hm, float* s, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(2 * C + 2 * Cr);
    hipLaunchKernelGGL(k_cbam_mlp, dim3(N), dim3(256), lds, st, avg, mx, w1, b1, w2, b2, C, Cr, ha, hm, s);
    return (int)hipGetLastError();
}
int launch_cbam_sppool(const void* x, long x_bs, const float* s, int N, int C, int P, float* maps, hipStream_t st, int dt) {
    SMAAT_DISPATCH_ET(dt, T,
        if (((P & 3) == 0) && ((x_bs & 3) == 0) && ((((uintptr_t)x) & Elem<T>::vmask) == 0) && ((((uintptr_t)maps) & 15) == 0))
            hipLaunchKernelGGL(k_cbam_sppool_v4<T>, dim3(cdivc(P, 256), N), dim3(64), 0, st, (const T*)x, x_bs, s, C, P, maps);
        else
            hipLaunchKernelGGL(k_cbam_sppool<T>, dim3(cdivc(P, 256), N), dim3(256), 0, st, (const T*)x, x_bs, s, C, P, maps););
    return (int)hipGetLastError();
}
int launch_cbam_spconv(const float* maps, const float* wc, int ks, int N, int H, int W, float* conv, float* part,
                       hipStream_t st) {
    if (ks != 3 && ks != 7) return -1;
    dim3 grid(cdivc(W, SPT), cdivc(H, SPT), N);
    hipLaunchKernelGGL(k_cbam_spconv, grid, dim3(256), 0, st, maps, wc, ks, H, W, conv, part,
                       (int)(grid.x * grid.y * grid.z));
    return (int)hipGetLastError();
}
int launch_cbam_gate(const float* conv, const float* scale, const float* shift, long total, float* gate,
                     hipStream_t st) {
    hipLaunchKernelGGL(k_cbam_gate, dim3(cdivc(total, 256)), dim3(256), 0, st, conv, scale, shift, total, gate);
    return (int)hipGetLastError();
}
int launch_cbam_apply(const void* x, long x_bs, const float* s, const float* gate, void* out, long out_bs, int N,
                      int C, int P, hipStream_t st, int dt, unsigned* amax) {
    if (amax && dt != SMAAT_F32) return -2;
    const int seg = seg_len_c(P);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_cbam_apply<T>, dim3(N * C, cdivc(P, seg)), dim3(256), 0, st, (const T*)x, x_bs, s, gate, (T*)out,
                           out_bs, C, P, seg, amax););
    return (int)hipGetLastError();
}
int launch_cbam_eval_pool(const float* x, long x_bs, const float* avg, const float* mx, const float* w1, const float* b1,
                          const float* w2, const float* b2, int N, int C, int Cr, int P, float* s_out, float* maps,
                          hipStream_t st) {
    const int blocks = cdivc(P, 256) * N;
    if (blocks < 128 && C >= 64) {  // deep levels at small batch: 16 waves share the channels of a 256-pixel block
        const size_t lds = sizeof(float) * (size_t)((3 * C + 2 * Cr + 3) & ~3) + 2 * 1024 * sizeof(float4);
        hipLaunchKernelGGL(k_cbam_mlp_sppool<16>, dim3(cdivc(P, 256), N), dim3(1024), lds, st, x, x_bs, avg, mx, w1, b1, w2,
                           b2, C, Cr, P, s_out, maps);
    } else {
        const size_t lds = sizeof(float) * (size_t)((3 * C + 2 * Cr + 3) & ~3) + 2 * 256 * sizeof(float4);
        hipLaunchKernelGGL(k_cbam_mlp_sppool<4>, dim3(cdivc(P, 256), N), dim3(256), lds, st, x, x_bs, avg, mx, w1, b1, w2, b2,
                           C, Cr, P, s_out, maps);
    }
    return (int)hipGetLastError();
}
int launch_cbam_eval_apply(const float* x, long x_bs, const float* s, const float* maps, const float* wc, int ks,
                           const float* bn_g, const float* bn_b, const float* bn_rm, const float* bn_rv, float eps, int N,
                           int C, int H, int W, float* out, long out_bs, float* pooled, long pooled_bs, hipStream_t st) {
    if (ks != 3 && ks != 7) return -1;
    const int tx = cdivc(W, GAT_TW), ty = cdivc(H, GAT_TH);
    // enough blocks to fill the chip AND to keep a block's channel loop short: at batch 1 a block that walks all 64
    // channels of a 288^2 tile is a 64-step load chain (19 us); the gate is cheap to recompute per channel slice
    int csplit = 2048 / (tx * ty * N);
    if (csplit > C / 8) csplit = C / 8;
    if (csplit < 1) csplit = 1;
    if ((long)N * csplit > 65535) csplit = 1;
    hipLaunchKernelGGL(k_cbam_gate_apply, dim3(tx, ty, N * csplit), dim3(256), 0, st, x, x_bs, s, maps, wc, ks, bn_g, bn_b,
                       bn_rm, bn_rv, eps, C, H, W, csplit, out, out_bs, pooled, pooled_bs);
    return (int)hipGetLastError();
}
int launch_cbam_bwd_gate(const void* dout, long dout_bs, const void* x, long x_bs, const float* s,
                         const float* gate, const float* conv, const float* mean, const float* invstd, int N, int C,
                         int P, float* dbn, float* part, hipStream_t st, int dt) {
    dim3 grid(cdivc(P, 256), N);
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    const bool v4 = ((P & 3) == 0) && ((x_bs & 3) == 0) && ((dout_bs & 3) == 0) && ((((uintptr_t)x) & am) == 0) &&
                    ((((uintptr_t)dout) & am) == 0) && ((((uintptr_t)gate) & 15) == 0) &&
                    ((((uintptr_t)conv) & 15) == 0) && ((((uintptr_t)dbn) & 15) == 0);
    SMAAT_DISPATCH_ET(dt, T,
        if (v4)
            hipLaunchKernelGGL(k_cbam_bwd_gate_v4<T>, grid, dim3(64), 0, st, (const T*)dout, dout_bs, (const T*)x, x_bs, s, gate,
                               conv, mean, invstd, C, P, dbn, part, (int)(grid.x * grid.y));
        else
            hipLaunchKernelGGL(k_cbam_bwd_gate<T>, grid, dim3(256), 0, st, (const T*)dout, dout_bs, (const T*)x, x_bs, s, gate,
                               conv, mean, invstd, C, P, dbn, part, (int)(grid.x * grid.y)););
    return (int)hipGetLastError();
}
int launch_cbam_bwd_spconv(const float* dbn, const float* conv, const float* mean, const float* invstd,
                           const float* coef, const float* maps, const float* wc, int ks, int N, int H, int W,
                           float* dmaps, float* wpart, hipStream_t st) {
    if (ks != 3 && ks != 7) return -1;
    dim3 grid(cdivc(W, SPT), cdivc(H, SPT), N);
    hipLaunchKernelGGL(k_cbam_bwd_spconv, grid, dim3(256), 0, st, dbn, conv, mean, invstd, coef, maps, wc, ks, H, W,
                       dmaps, wpart);
    return (int)hipGetLastError();
}
int launch_cbam_bwd_main(const void* dout, long dout_bs, const void* x, long x_bs, const float* s,
                         const float* gate, const float* maps, const float* dmaps, int N, int C, int P, void* dx,
                         long dx_bs, float* dspart, hipStream_t st, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    const bool v4 = ((P & 3) == 0) && ((x_bs & 3) == 0) && ((dout_bs & 3) == 0) && ((dx_bs & 3) == 0) &&
                    ((((uintptr_t)x) & am) == 0) && ((((uintptr_t)dout) & am) == 0) &&
                    ((((uintptr_t)dx) & am) == 0) && ((((uintptr_t)gate) & 15) == 0) &&
                    ((((uintptr_t)maps) & 15) == 0) && ((((uintptr_t)dmaps) & 15) == 0) &&
                    ((long)C * P * 4 < (1L << 31));  // dX goes through a 32-bit buffer offset (bit 31 = dropped)
    SMAAT_DISPATCH_ET(dt, T,
        if (v4) {
            hipLaunchKernelGGL(k_cbam_bwd_main_v4<T>, dim3(cdivc(P, 256), N), dim3(64), 0, st, (const T*)dout, dout_bs,
                               (const T*)x, x_bs, s, gate, maps, dmaps, C, P, (T*)dx, dx_bs, dspart);
        } else {
            const size_t lds = sizeof(float) * (size_t)(4 * C);
            hipLaunchKernelGGL(k_cbam_bwd_main<T>, dim3(cdivc(P, 256), N), dim3(256), lds, st, (const T*)dout, dout_bs,
                               (const T*)x, x_bs, s, gate, maps, dmaps, C, P, (T*)dx, dx_bs, dspart);
        });
    return (int)hipGetLastError();
}
int launch_cbam_bwd_mlp(const float* ds, const float* s, const float* avg, const float* mx, const float* ha,
                        const float* hm, const float* w1, const float* w2, int N, int C, int Cr, float* pg,
                        float* davg, float* dmx, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(C + 2 * Cr);
    hipLaunchKernelGGL(k_cbam_bwd_mlp, dim3(N), dim3(256), lds, st, ds, s, avg, mx, ha, hm, w1, w2, C, Cr, pg, davg,
                       dmx);
    return (int)hipGetLastError();
}
int launch_cbam_bwd_final(void* dx, long dx_bs, const float* davg, const float* dmx, const int* amax, int N, int C,
                          int P, hipStream_t st, int dt) {
    const int seg = seg_len_c(P);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_cbam_bwd_final<T>, dim3(N * C, cdivc(P, seg)), dim3(256), 0, st, (T*)dx, dx_bs, davg, dmx, amax,
                           C, P, seg););
    return (int)hipGetLastError();
}

// -2: shape / alignment not handled (the caller runs smaat_cbam_bwd_final + smaat_maxpool2_bwd)
int launch_cbam_final_pool_bwd(void* dx, long dx_bs, const float* davg, const float* dmx, const int* amax,
                               const void* x, long x_bs, const void* dpool, long dp_bs, int N, int C, int H, int W,
                               hipStream_t st, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if ((W & 3) != 0 || (dx_bs & 3) != 0 || (x_bs & 3) != 0 || (dp_bs & 1) != 0 || ((((uintptr_t)dx) & am) != 0) ||
        ((((uintptr_t)x) & am) != 0) || ((((uintptr_t)dpool) & (am >> 1)) != 0) || H < 2)
        return -2;
    const long total = (long)N * C * (W >> 2) * ((H + 1) >> 1);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_cbam_final_pool_bwd<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (T*)dx, dx_bs, davg,
                           dmx, amax, (const T*)x, x_bs, (const T*)dpool, dp_bs, C, H, W, total););
    return (int)hipGetLastError();
}

// ---- the three-pass backward (see k_cbam_bwd_gate_ds_v4) --------------------------------------------------------------
// waves per workgroup that split the channels: 1 while the launch fills the chip with single-wave blocks or the channel loop
// is short; otherwise enough that a wave walks <= 64 channels (32 trips), at most 8
static int cbam_chan_split(int N, int C, int P) {
    static int forced = -1;  // SMAAT_CBAM_CS=<1|2|4|8>: experiment switch
    if (forced < 0) {
        const char* e = getenv("SMAAT_CBAM_CS");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    const long waves = (long)cdivc(P, 256) * N;
    if (waves >= 4096 || C <= 64) return 1;
    int cs = 1;
    while (cs < 8 && C / cs > 64) cs *= 2;
    return cs;
}
// 1: every pointer / stride the three kernels touch allows the 16-byte forms; with a pooled gradient H even and W % 4 == 0,
// without P % 4 == 0
int cbam_bwd3_ok(const void* x, long x_bs, const void* dout, long dout_bs, const void* dpool, long dp_bs, int N, int C, int H,
                 int W, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if (N < 1 || C < 1 || H < 1 || W < 1 || (((long)H * W) & 3)) return 0;
    if (dpool && (H < 2 || (H & 1) || (W & 3))) return 0;
    if ((x_bs & 3) || (dout_bs & 3) || (((uintptr_t)x) & am) || (((uintptr_t)dout) & am)) return 0;
    if (dpool && ((dp_bs & 1) || (((uintptr_t)dpool) & (am >> 1)))) return 0;
    if ((long)C * H * W * 4 >= (1L << 31)) return 0;  // dX goes through a 32-bit buffer offset (bit 31 = dropped)
    return 1;
}
int launch_cbam_sppool_idx(const void* x, long x_bs, const float* s, int N, int C, int P, float* maps, int* amaxc,
                           hipStream_t st, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if ((P & 3) || (x_bs & 3) || (((uintptr_t)x) & am) || (((uintptr_t)maps) & 15) || (((uintptr_t)amaxc) & 15)) return -2;
    const int cs = cbam_chan_split(N, C, P);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_cbam_sppool_idx_v4<T>, dim3(cdivc(P, 256), N), dim3(64 * cs), cs > 1 ? (size_t)3 * cs * 64 * 16 : 0, st,
                           (const T*)x, x_bs, s, C, P, maps, amaxc););
    return (int)hipGetLastError();
}
int launch_cbam_bwd_gate_ds(const void* dout, long dout_bs, const void* x, long x_bs, const float* s, const float* gate,
                            const float* conv, const float* mean, const float* invstd, int N, int C, int P, float* dbn,
                            float* part, float* dspart, hipStream_t st, int dt) {
    if ((P & 3) || (((uintptr_t)gate) & 15) || (((uintptr_t)conv) & 15) || (((uintptr_t)dbn) & 15)) return -2;
    dim3 grid(cdivc(P, 256), N);
    const int cs = cbam_chan_split(N, C, P);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_cbam_bwd_gate_ds_v4<T>, grid, dim3(64 * cs), cs > 1 ? (size_t)cs * 64 * 16 : 0, st, (const T*)dout,
                           dout_bs, (const T*)x, x_bs, s, gate, conv, mean, invstd, C, P, dbn, part, (int)(grid.x * grid.y),
                           dspart););
    return (int)hipGetLastError();
}
int launch_cbam_bwd_ds2(const void* x, long x_bs, const float* dmaps, const int* amaxc, int N, int C, int P, float* dspart,
                        hipStream_t st, int dt) {
    if ((P & 3) || (((uintptr_t)dmaps) & 15) || (((uintptr_t)amaxc) & 15)) return -2;
    const int cs = cbam_chan_split(N, C, P);
    SMAAT_DISPATCH_ET(dt, T,
        hipLaunchKernelGGL(k_cbam_bwd_ds2_v4<T>, dim3(cdivc(P, 256), N), dim3(64 * cs), 0, st, (const T*)x, x_bs, dmaps, amaxc, C, P,
                           dspart););
    return (int)hipGetLastError();
}
int launch_cbam_bwd_apply(const void* dout, long dout_bs, const void* x, long x_bs, const float* s, const float* gate,
                          const float* dmaps, const int* amaxc, const float* davg, const float* dmx, const int* amax,
                          const void* dpool, long dp_bs, int N, int C, int H, int W, void* dx, long dx_bs, hipStream_t st, int dt) {
    const unsigned am = dt == SMAAT_BF16 ? 7u : 15u;
    if (!cbam_bwd3_ok(x, x_bs, dout, dout_bs, dpool, dp_bs, N, C, H, W, dt) || (dx_bs & 3) || (((uintptr_t)dx) & am) ||
        (((uintptr_t)gate) & 15) || (((uintptr_t)dmaps) & 15) || (((uintptr_t)amaxc) & 15))
        return -2;
    const int P = H * W;
    const int cs = cbam_chan_split(N, C, P);
    SMAAT_DISPATCH_ET(dt, T,
        if (dpool)
            hipLaunchKernelGGL((k_cbam_bwd_apply_v4<T, true>), dim3(cdivc((W >> 2) * (H >> 1), 64), N), dim3(64 * cs), 0, st,
                               (const T*)dout, dout_bs, (const T*)x, x_bs, s, gate, dmaps, amaxc, davg, dmx, amax, (const T*)dpool,
                               dp_bs, C, H, W, (T*)dx, dx_bs);
        else
            hipLaunchKernelGGL((k_cbam_bwd_apply_v4<T, false>), dim3(cdivc(P, 256), N), dim3(64 * cs), 0, st, (const T*)dout,
                               dout_bs, (const T*)x, x_bs, s, gate, dmaps, amaxc, davg, dmx, amax, (const T*)nullptr, 0L, C, H, W,
                               (T*)dx, dx_bs););
    return (int)hipGetLastError();
}
