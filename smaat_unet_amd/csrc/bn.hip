// Train-mode BatchNorm2d (+ReLU) forward/backward passes around the pointwise GEMM.
// Reference: nn.BatchNorm2d in models/unet_parts_depthwise_separable.py:25,34 and
// models/layers.py:120,127 (biased variance for normalisation, unbiased into
// running_var, momentum/eps from the module), nn.ReLU(inplace=True) :26,35.
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------
// finalize forward statistics: per-tile partials part[3][T][C] = (mean_t, M2_t, n_t) of (z - shift_bias)
// (common.h "BatchNorm partial statistics") -> per channel.  The tiles are merged in fp64 with the pairwise
// update  M2 = sum_t M2_t + n_t (mean_t - mean)^2, evaluated in one pass about a shift taken from the data: no
// E[z^2] - E[z]^2 of raw values anywhere, so the variance keeps its accuracy when |mean| >> std.
// ---------------------------------------------------------------------------------
template <int NTH>
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int st = NTH / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// NTH threads per channel: 1024 for the layers with thousands of tiles (the reads of one channel are strided by C
// floats, so the time is set by the number of loads in flight), 256 otherwise.
// Level 1 for layers with thousands of tiles: merge the tiles of a slab of R rows into ONE partial of the same form,
// written over the slab's head row (the partial buffer is scratch once the GEMM has finished; same convention as
// k_reduce_rows_l1).  A block owns 16 adjacent channels (64-byte runs: full sectors) and 16 row lanes; fp64 one-pass
// sums about the head row's mean, row lanes added in a fixed order.  k_bn_finalize then reads the head rows only.
__global__ __launch_bounds__(256) void k_bn_merge_slabs(float* __restrict__ part, int T, int C, int R) {
    __shared__ double red[3][16][17];
    const int ch = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + ch;
    const int t0 = blockIdx.y * R;
    const int t1 = t0 + R < T ? t0 + R : T;
    const bool cv = c < C;
    const int cc = cv ? c : C - 1;
    float* pm = part + cc;
    float* pq = part + (long)T * C + cc;
    float* pn = part + 2L * T * C + cc;
    const double K = (double)pm[(long)t0 * C];
    double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int t = t0 + rl; t < t1; t += 16) {
        const double nt = (double)pn[(long)t * C];
        const double d = (double)pm[(long)t * C] - K;
        n += nt;
        s1 += nt * d;
        s2 += (double)pq[(long)t * C] + nt * d * d;
    }
    red[0][ch][rl] = n;
    red[1][ch][rl] = s1;
    red[2][ch][rl] = s2;
    __syncthreads();  // (also: every read of the head row is done before it is overwritten)
    if (rl == 0 && cv) {
        double nn = 0.0, a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            nn += red[0][ch][j];
            a += red[1][ch][j];
            b += red[2][ch][j];
        }
        const double mean = nn > 0.0 ? K + a / nn : 0.0;
        double m2 = nn > 0.0 ? b - a * a / nn : 0.0;
        if (m2 < 0.0) m2 = 0.0;
        pm[(long)t0 * C] = (float)mean;
        pq[(long)t0 * C] = (float)m2;
        pn[(long)t0 * C] = (float)nn;
    }
}

// RS: row stride (1, or the slab size after k_bn_merge_slabs: only the head rows are read)
template <int NTH>
__global__ __launch_bounds__(NTH) void k_bn_finalize(const float* __restrict__ part, int T, int RS, int C, double count,
                                                     const float* __restrict__ bias_shift,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float momentum, float* running_mean,
                                                     float* running_var, float* mean_out, float* invstd_out,
                                                     float* scale_out, float* shift_out) {
    __shared__ double red[NTH];
    const int c = blockIdx.x;
    const long RC = (long)RS * C;  // distance between the rows that are read
    const float* pm = part + c;
    const float* pq = part + (long)T * C + c;
    const float* pn = part + 2L * T * C + c;
    const int TR = (T + RS - 1) / RS;  // rows read
    // ONE pass, in fp64, about the shift K = mean of tile 0 (a value inside the data range, so the final
    // s2 - s1^2 / N has no cancellation beyond the spread of the tile means):
    //   N = sum n_t,  s1 = sum n_t (m_t - K),  s2 = sum [M2_t + n_t (m_t - K)^2]
    const double K = (double)pm[0];
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    int t = threadIdx.x;
    for (; t + 3 * NTH < TR; t += 4 * NTH) {  // four independent loads of each row in flight
        double nn[4], dd[4], qq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long o = (long)(t + NTH * u) * RC;
            nn[u] = (double)pn[o];
            dd[u] = (double)pm[o] - K;
            qq[u] = (double)pq[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            n += nn[u];
            s1 += nn[u] * dd[u];
            s2 += qq[u] + nn[u] * dd[u] * dd[u];
        }
    }
    for (; t < TR; t += NTH) {
        const double nt = (double)pn[(long)t * RC];
        const double d = (double)pm[(long)t * RC] - K;
        n += nt;
        s1 += nt * d;
        s2 += (double)pq[(long)t * RC] + nt * d * d;
    }
    const double ntot = block_sum_f64<NTH>(n, red);
    const double s1t = block_sum_f64<NTH>(s1, red);
    const double s2t = block_sum_f64<NTH>(s2, red);
    const double m0 = ntot > 0.0 ? K + s1t / ntot : 0.0;
    const double qtot = ntot > 0.0 ? s2t - s1t * s1t / ntot : 0.0;
    if (threadIdx.x == 0) {
        double var = ntot > 0.0 ? qtot / ntot : 0.0;
        if (var < 0.0) var = 0.0;
        const double mean = m0 + (bias_shift ? (double)bias_shift[c] : 0.0);
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float meanf = (float)mean, invf = (float)invstd;
        mean_out[c] = meanf;
        invstd_out[c] = invf;
        const float sc = g * invf;
        scale_out[c] = sc;
        shift_out[c] = bt - meanf * sc;
        if (running_mean) {
            const double unb = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
        }
    }
}

// ---------------------------------------------------------------------------------
// y = [relu](z * scale[c] + shift[c]) over [N][C][P] planes with batch strides
// grid: (N*C planes, segments)
// ---------------------------------------------------------------------------------
// TZ / TY: element types of z and y (f32 | bf16 storage, common.h)
template <bool RELU, typename TZ, typename TY>
__global__ __launch_bounds__(256) void k_affine_act(const TZ* __restrict__ z, long z_bs,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    TY* __restrict__ y, long y_bs, int C, int P, int seg_len) {
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float sc = scale[c], sh = shift[c];
    const TZ* zp = z + (long)n * z_bs + (long)c * P;
    TY* yp = y + (long)n * y_bs + (long)c * P;
    const int p0 = blockIdx.y * seg_len;
    int p1 = p0 + seg_len;
    if (p1 > P) p1 = P;
    const bool vec = ((P & 3) == 0) && ((z_bs & 3) == 0) && ((y_bs & 3) == 0) && ((seg_len & 3) == 0) &&
                     ((((uintptr_t)z) & Elem<TZ>::vmask) == 0) && ((((uintptr_t)y) & Elem<TY>::vmask) == 0);
    if (vec) {
        auto one = [&](float4 v, int p) {
            v.x = fmaf(v.x, sc, sh);
            v.y = fmaf(v.y, sc, sh);
            v.z = fmaf(v.z, sc, sh);
            v.w = fmaf(v.w, sc, sh);
            if (RELU) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            st4(yp + p, v);
        };
        int p = p0 + threadIdx.x * 4;  // two positions per trip: both loads issued before the first store
        for (; p + 1024 < p1; p += 2048) {
            const auto a = ldraw4(zp + p), b = ldraw4(zp + p + 1024);
            one(cvt4(a), p);
            one(cvt4(b), p + 1024);
        }
        if (p < p1) one(ld4(zp + p), p);
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            float v = fmaf(ld1(zp + p), sc, sh);
            if (RELU) v = fmaxf(v, 0.f);
            st1(yp + p, v);
        }
    }
}

// ---------------------------------------------------------------------------------
// backward pass 1: per (plane, segment) partial sums of g and g*xhat,
//   g = dy * [z*scale+shift > 0] (RELU) or dy;  xhat = (z - mean) * invstd
// part[2][slots][C], slot = n * nseg + seg
// ---------------------------------------------------------------------------------
// HEAD (hw != null): the consumer of y = relu(bn(z)) is a 1x1 convolution to ONE channel (OutConv with n_classes = 1,
// reference models/unet_parts.py:67-73) whose gradient dlog [N][P] is given instead of dy: dy[n][c][p] = hw[c] *
// dlog[n][p] is formed on the fly (the same single product ATen's conv backward stores), and the kernel also emits that
// convolution's weight gradient, part[2][slot][c] = sum dlog * y.  The 64-channel dy tensor is never written or read.
// TG / TZ: element types of dy (HEAD: of dlog) and z
template <bool RELU, bool HEAD, typename TG, typename TZ>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const TG* __restrict__ dy, long dy_bs,
                                                       const TZ* __restrict__ z, long z_bs,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ part,
                                                       int C, int P, int seg_len, int slots,
                                                       const float* __restrict__ hw) {
    __shared__ float red[12];
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const float wc = HEAD ? hw[c] : 1.f;
    const TZ* zp = z + (long)n * z_bs + (long)c * P;
    const TG* gp = dy + (long)n * dy_bs + (HEAD ? 0L : (long)c * P);
    const int p0 = blockIdx.y * seg_len;
    int p1 = p0 + seg_len;
    if (p1 > P) p1 = P;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const bool vec = ((P & 3) == 0) && ((z_bs & 3) == 0) && ((dy_bs & 3) == 0) && ((seg_len & 3) == 0) &&
                     ((((uintptr_t)z) & Elem<TZ>::vmask) == 0) && ((((uintptr_t)dy) & Elem<TG>::vmask) == 0);
    if (vec) {
        auto one = [&](const float4 zv, const float4 gv) {
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
            const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = fmaf(zz[j], sc, sh);
                float g = HEAD ? wc * gg[j] : gg[j];
                if (RELU && !(a > 0.f)) g = 0.f;
                s1 += g;
                s2 = fmaf(g, (zz[j] - mu) * is, s2);
                if (HEAD) s3 = fmaf(gg[j], RELU ? fmaxf(a, 0.f) : a, s3);
            }
        };
        int p = p0 + threadIdx.x * 4;  // two positions per trip: four loads in flight (same summation order)
        for (; p + 1024 < p1; p += 2048) {
            const auto za = ldraw4(zp + p);
            const auto ga = ldraw4(gp + p);
            const auto zb = ldraw4(zp + p + 1024);
            const auto gb = ldraw4(gp + p + 1024);
            one(cvt4(za), cvt4(ga));
            one(cvt4(zb), cvt4(gb));
        }
        if (p < p1) one(ld4(zp + p), ld4(gp + p));
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float zz = ld1(zp + p);
            const float a = fmaf(zz, sc, sh);
            const float g0 = ld1(gp + p);
            float g = HEAD ? wc * g0 : g0;
            if (RELU && !(a > 0.f)) g = 0.f;
            s1 += g;
            s2 = fmaf(g, (zz - mu) * is, s2);
            if (HEAD) s3 = fmaf(g0, RELU ? fmaxf(a, 0.f) : a, s3);
        }
    }
    const float t1 = block_sum_t0(s1, red);
    const float t2 = block_sum_t0(s2, red + 4);
    const float t3 = HEAD ? block_sum_t0(s3, red + 8) : 0.f;
    if (threadIdx.x == 0) {
        const int slot = n * gridDim.y + blockIdx.y;
        part[(long)slot * C + c] = t1;
        part[((long)slots + slot) * C + c] = t2;
        if (HEAD) part[(2L * slots + slot) * C + c] = t3;
    }
}

// finalize backward: dgamma = S2, dbeta = S1, coefficients for the apply pass
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ part, int slots, int C,
                                                         double count, const float* __restrict__ gamma,
                                                         const float* __restrict__ invstd, float* dgamma,
                                                         float* dbeta, float* coef /*[3][C]*/) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < slots; t += 256) {
        s += (double)part[(long)t * C + c];
        q += (double)part[((long)slots + t) * C + c];
    }
    __shared__ double rs[256], rq[256];
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            rs[threadIdx.x] += rs[threadIdx.x + st];
            rq[threadIdx.x] += rq[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)rs[0];
        if (dgamma) dgamma[c] = (float)rq[0];
        const float g = gamma ? gamma[c] : 1.f;
        coef[c] = g * invstd[c];
        coef[C + c] = (float)(rs[0] / count);
        coef[2 * C + c] = (float)(rq[0] / count);
    }
}

// backward pass 2: dz = c1 * (g - c2 - xhat * c3)
template <bool RELU, bool HEAD, typename TG, typename TZ, typename TD>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const TG* __restrict__ dy, long dy_bs,
                                                      const TZ* __restrict__ z, long z_bs,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* __restrict__ coef, TD* __restrict__ dz,
                                                      long dz_bs, int C, int P, int seg_len,
                                                      const float* __restrict__ hw, unsigned* __restrict__ amax, int lin_nseg) {
    // amax (nullable): max |dz| over the whole tensor, for the two-term fp16 split GEMMs that read dz (common.h)
    // lin_nseg > 0: one-dimensional grid, workgroup b = segment b % lin_nseg of plane b / lin_nseg -- consecutive workgroups
    // walk consecutive addresses (scripts/probes/stream_shape_probe.hip: 5.6-5.9 TB/s against 5.1 for the (planes, segments) grid)
    const int plane = lin_nseg > 0 ? (int)(blockIdx.x / (unsigned)lin_nseg) : (int)blockIdx.x;
    const int sgi = lin_nseg > 0 ? (int)(blockIdx.x - (unsigned)plane * (unsigned)lin_nseg) : (int)blockIdx.y;
    const int n = plane / C, c = plane - n * C;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const float c1 = coef[c], c2 = coef[C + c], c3 = coef[2 * C + c];
    float am = 0.f;
    __shared__ float amred[4];
    const float wc = HEAD ? hw[c] : 1.f;  // HEAD: dy[n][c][p] = hw[c] * dlog[n][p], see k_bn_bwd_reduce
    const TZ* zp = z + (long)n * z_bs + (long)c * P;
    const TG* gp = dy + (long)n * dy_bs + (HEAD ? 0L : (long)c * P);
    TD* op = dz + (long)n * dz_bs + (long)c * P;
    const int p0 = sgi * seg_len;
    int p1 = p0 + seg_len;
    if (p1 > P) p1 = P;
    const bool vec = ((P & 3) == 0) && ((z_bs & 3) == 0) && ((dy_bs & 3) == 0) && ((dz_bs & 3) == 0) &&
                     ((seg_len & 3) == 0) && ((((uintptr_t)z) & Elem<TZ>::vmask) == 0) &&
                     ((((uintptr_t)dy) & Elem<TG>::vmask) == 0) && ((((uintptr_t)dz) & Elem<TD>::vmask) == 0);
    if (vec) {
        auto one = [&](const float4 zv, const float4 gv, int p) {
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
            const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float g = HEAD ? wc * gg[j] : gg[j];
                if (RELU && !(fmaf(zz[j], sc, sh) > 0.f)) g = 0.f;
                o[j] = c1 * (g - c2 - (zz[j] - mu) * is * c3);
                am = fmaxf(am, fabsf(o[j]));
            }
            st4(op + p, make_float4(o[0], o[1], o[2], o[3]));
        };
        // two positions per trip, their four loads issued together (one position per trip waits for its two loads
        // with nothing else in flight: scripts/asm_lint.py DRAIN)
        int p = p0 + threadIdx.x * 4;
        for (; p + 1024 < p1; p += 2048) {
            const auto za = ldraw4(zp + p);
            const auto ga = ldraw4(gp + p);
            const auto zb = ldraw4(zp + p + 1024);
            const auto gb = ldraw4(gp + p + 1024);
            one(cvt4(za), cvt4(ga), p);
            one(cvt4(zb), cvt4(gb), p + 1024);
        }
        if (p < p1) one(ld4(zp + p), ld4(gp + p), p);
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            const float zz = ld1(zp + p);
            float g = HEAD ? wc * ld1(gp + p) : ld1(gp + p);
            if (RELU && !(fmaf(zz, sc, sh) > 0.f)) g = 0.f;
            const float o = c1 * (g - c2 - (zz - mu) * is * c3);
            am = fmaxf(am, fabsf(o));
            st1(op + p, o);
        }
    }
    if (amax) amax_publish_block256(amax, am, (unsigned)plane * 5u + (unsigned)sgi, amred);  // (block-uniform branch)
}

// ---------------------------------------------------------------------------------
// generic helpers
// ---------------------------------------------------------------------------------
// out[j] = alpha * sum_r part[r][j], fp64 accumulation, deterministic order.
// Two levels so that a tall skinny partial buffer (thousands of rows x a few thousand columns)
// still fills the chip: level 1 reduces row slices IN PLACE into the first row of each slice,
// level 2 sums the slice heads.
__global__ __launch_bounds__(256) void k_reduce_rows_l1(float* __restrict__ part, int rows, long len, int q) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= len) return;
    const int r0 = blockIdx.y * q;
    int r1 = r0 + q;
    if (r1 > rows) r1 = rows;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += (double)part[(long)(r + 0) * len + j];
        s1 += (double)part[(long)(r + 1) * len + j];
        s2 += (double)part[(long)(r + 2) * len + j];
        s3 += (double)part[(long)(r + 3) * len + j];
    }
    for (; r < r1; ++r) s0 += (double)part[(long)r * len + j];
    // NOTE: the slice head keeps a float; the final sum of <= 64 heads is again fp64
    if (r0 < rows) part[(long)r0 * len + j] = (float)((s0 + s1) + (s2 + s3));
}

__global__ __launch_bounds__(256) void k_reduce_rows(const float* __restrict__ part, int rows, long len, int stride,
                                                     float* __restrict__ out, float alpha) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= len) return;
    double s = 0.0;
    for (int r = 0; r < rows; r += stride) s += (double)part[(long)r * len + j];
    out[j] = (float)(s * alpha);
}

// part[slot][c] = sum over one plane segment of x[n][c][:]   (bias gradients; finished by k_reduce_rows)
template <typename T>
__global__ __launch_bounds__(256) void k_plane_sum(const T* __restrict__ x, long x_bs, int C, int P, int seg_len,
                                                   float* __restrict__ part) {
    __shared__ float red[4];
    const int plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const T* xp = x + (long)n * x_bs + (long)c * P;
    const int p0 = blockIdx.y * seg_len;
    int p1 = p0 + seg_len;
    if (p1 > P) p1 = P;
    float ls = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) ls += ld1(xp + p);
    const float t = block_sum_t0(ls, red);
    if (threadIdx.x == 0) part[(long)(n * gridDim.y + blockIdx.y) * C + c] = t;
}

// strided plane copy: dst[n][c][p] = src[n][c][p] with separate batch strides
__global__ __launch_bounds__(256) void k_copy_planes(const float* __restrict__ src, long s_bs,
                                                     float* __restrict__ dst, long d_bs, long plane_len, int accum) {
    const int n = blockIdx.y;
    const float* sp = src + (long)n * s_bs;
    float* dp = dst + (long)n * d_bs;
    const bool vec = ((plane_len & 3) == 0) && ((s_bs & 3) == 0) && ((d_bs & 3) == 0) &&
                     ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 15) == 0);
    if (vec) {
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < plane_len; i += (long)gridDim.x * 1024) {
            float4 v = *(const float4*)(sp + i);
            if (accum) {
                const float4 o = *(const float4*)(dp + i);
                v.x += o.x;
                v.y += o.y;
                v.z += o.z;
                v.w += o.w;
            }
            *(float4*)(dp + i) = v;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < plane_len; i += (long)gridDim.x * 256)
            dp[i] = accum ? dp[i] + sp[i] : sp[i];
    }
}

// =====================================================================================
// launchers
// =====================================================================================
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// segment length used by the plane-wise kernels: multiple of 1024, <= 8192 elements per block
static int plane_seg_len(int P) {
    if (P <= 8192) return ((P + 1023) / 1024) * 1024;
    return 8192;
}

int smaat_bn_bwd_num_slots_impl(int N, int P) { return N * cdiv(P, plane_seg_len(P)); }

int launch_bn_finalize(float* part, int T, int C, double count, const float* bias_shift, const float* gamma,
                       const float* beta, float eps, float momentum, float* rm, float* rv, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t st) {
    int RS = 1;
    if (T >= 2048) {  // two levels: 64 slabs merged in place, then the head rows
        RS = (T + 63) / 64;
        const int ns = (T + RS - 1) / RS;
        hipLaunchKernelGGL(k_bn_merge_slabs, dim3((C + 15) / 16, ns), dim3(256), 0, st, (float*)part, T, C, RS);
    }
    hipLaunchKernelGGL(k_bn_finalize<256>, dim3(C), dim3(256), 0, st, part, T, RS, C, count, bias_shift, gamma, beta, eps,
                       momentum, rm, rv, mean, invstd, scale, shift);
    return (int)hipGetLastError();
}

// *_dt: SMAAT_F32 | SMAAT_BF16 element type of the tensor argument in front of it.  Built combinations: all f32; all
// bf16; (HEAD) f32 dlog with bf16 z / dz.  -2 otherwise.
int launch_affine_act(const void* z, int z_dt, long z_bs, const float* scale, const float* shift, void* y, int y_dt,
                      long y_bs, int N, int C, int P, int relu, hipStream_t st) {
    const int seg = plane_seg_len(P);
    dim3 grid(N * C, cdiv(P, seg));
#define AFF_GO(R, TZ, TY)                                                                                          \
    hipLaunchKernelGGL((k_affine_act<R, TZ, TY>), grid, dim3(256), 0, st, (const TZ*)z, z_bs, scale, shift, (TY*)y, y_bs, C, \
                       P, seg)
#define AFF_T(TZ, TY)                              \
    do {                                           \
        if (relu) AFF_GO(true, TZ, TY); else AFF_GO(false, TZ, TY); \
    } while (0)
    if (z_dt == SMAAT_F32 && y_dt == SMAAT_F32) AFF_T(float, float);
    else if (z_dt == SMAAT_BF16 && y_dt == SMAAT_BF16) AFF_T(bf16_t, bf16_t);
    else if (z_dt == SMAAT_BF16 && y_dt == SMAAT_F32) AFF_T(bf16_t, float);
    else if (z_dt == SMAAT_F32 && y_dt == SMAAT_BF16) AFF_T(float, bf16_t);
    else return -2;
#undef AFF_T
#undef AFF_GO
    return (int)hipGetLastError();
}

int launch_bn_bwd_reduce(const void* dy, int dy_dt, long dy_bs, const void* z, int z_dt, long z_bs, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float* part, int N, int C, int P,
                         int relu, hipStream_t st, const float* hw) {
    const int seg = plane_seg_len(P);
    dim3 grid(N * C, cdiv(P, seg));
    const int slots = N * grid.y;
#define BNR_GO(R, H, TG, TZ)                                                                                         \
    hipLaunchKernelGGL((k_bn_bwd_reduce<R, H, TG, TZ>), grid, dim3(256), 0, st, (const TG*)dy, dy_bs, (const TZ*)z, z_bs,  \
                       scale, shift, mean, invstd, part, C, P, seg, slots, hw)
#define BNR_T(TG, TZ)                                                          \
    do {                                                                       \
        if (hw) {                                                              \
            if (relu) BNR_GO(true, true, TG, TZ); else BNR_GO(false, true, TG, TZ);   \
        } else {                                                               \
            if (relu) BNR_GO(true, false, TG, TZ); else BNR_GO(false, false, TG, TZ); \
        }                                                                      \
    } while (0)
    if (dy_dt == SMAAT_F32 && z_dt == SMAAT_F32) BNR_T(float, float);
    else if (dy_dt == SMAAT_BF16 && z_dt == SMAAT_BF16) BNR_T(bf16_t, bf16_t);
    else if (dy_dt == SMAAT_F32 && z_dt == SMAAT_BF16) BNR_T(float, bf16_t);
    else return -2;
#undef BNR_T
#undef BNR_GO
    return (int)hipGetLastError();
}

int launch_bn_bwd_finalize(const float* part, int slots, int C, double count, const float* gamma,
                           const float* invstd, float* dgamma, float* dbeta, float* coef, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C), dim3(256), 0, st, part, slots, C, count, gamma, invstd, dgamma,
                       dbeta, coef);
    return (int)hipGetLastError();
}

int launch_bn_bwd_apply(const void* dy, int dy_dt, long dy_bs, const void* z, int z_dt, long z_bs, const float* scale,
                        const float* shift, const float* mean, const float* invstd, const float* coef, void* dz, int dz_dt,
                        long dz_bs, int N, int C, int P, int relu, hipStream_t st, const float* hw, unsigned* amax) {
    int seg = plane_seg_len(P);
    dim3 grid(N * C, cdiv(P, seg));
    int lin = 0;
    {
        // Workgroups in ADDRESS ORDER, one float4 per thread, on the big f32 planes: 64 x 288^2 at batch 32 392 -> 340 us (5.2 ->
        // 6.0 TB/s), 128 x 144^2 188 -> 176 us; 72^2 and below unchanged or slower (profiles/r5/bn_apply_grid_r5l.txt).  What the
        // memory system sees is a window of a few MB sweeping through three tensors instead of ~16k concurrent 32 KB streams
        // (scripts/probes/stream_shape_probe.hip).  SMAAT_BN_LIN=<0|1024|2048|...>: experiment switch (0: always the 2-D grid)
        static int lin_seg = -1;
        if (lin_seg < 0) {
            const char* e = getenv("SMAAT_BN_LIN");
            lin_seg = e ? atoi(e) : 1024;
        }
        const bool big_f32 = P >= 16384 && dy_dt == SMAAT_F32 && z_dt == SMAAT_F32 && dz_dt == SMAAT_F32;
        if (big_f32 && lin_seg >= 1024 && (lin_seg & 1023) == 0 && (long)N * C * cdiv(P, lin_seg) < (1L << 31)) {
            seg = lin_seg;
            lin = cdiv(P, seg);
            grid = dim3((unsigned)((long)N * C * lin), 1);
        }
    }
#define BNA_GO(R, H, TG, TZ, TD)                                                                                        \
    hipLaunchKernelGGL((k_bn_bwd_apply<R, H, TG, TZ, TD>), grid, dim3(256), 0, st, (const TG*)dy, dy_bs, (const TZ*)z, z_bs,  \
                       scale, shift, mean, invstd, coef, (TD*)dz, dz_bs, C, P, seg, hw, amax, lin)
#define BNA_T(TG, TZ, TD)                                                              \
    do {                                                                               \
        if (hw) {                                                                      \
            if (relu) BNA_GO(true, true, TG, TZ, TD); else BNA_GO(false, true, TG, TZ, TD);   \
        } else {                                                                       \
            if (relu) BNA_GO(true, false, TG, TZ, TD); else BNA_GO(false, false, TG, TZ, TD); \
        }                                                                              \
    } while (0)
    if (dy_dt == SMAAT_F32 && z_dt == SMAAT_F32 && dz_dt == SMAAT_F32) BNA_T(float, float, float);
    else if (dy_dt == SMAAT_BF16 && z_dt == SMAAT_BF16 && dz_dt == SMAAT_BF16) BNA_T(bf16_t, bf16_t, bf16_t);
    else if (dy_dt == SMAAT_F32 && z_dt == SMAAT_BF16 && dz_dt == SMAAT_BF16) BNA_T(float, bf16_t, bf16_t);
    else return -2;
#undef BNA_T
#undef BNA_GO
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// OutConv with ONE output channel on an un-materialised activation (reference models/unet_parts.py:67-73 after
// unet_parts_depthwise_separable.py:34-35):  out[n][p] = b + sum_c w[c] * relu(z[n][c][p] * scale[c] + shift[c]).
// A thread owns four pixels and walks the channels (coalesced float4 rows); the block output of the last decoder level
// is never written.
// ---------------------------------------------------------------------------------
template <typename TZ>
__global__ __launch_bounds__(256) void k_outconv1_fwd(const TZ* __restrict__ z, long z_bs,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ out, long out_bs, int C, int P) {
    const int n = blockIdx.y;
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= P) return;
    const TZ* zp = z + (long)n * z_bs + p;
    const float b0 = b ? b[0] : 0.f;
    float4 acc = make_float4(b0, b0, b0, b0);
    const bool vec = ((P & 3) == 0) && ((z_bs & 3) == 0) && ((out_bs & 3) == 0) &&
                     ((((uintptr_t)z) & Elem<TZ>::vmask) == 0) && ((((uintptr_t)out) & 15) == 0);
    if (vec) {
        int c = 0;
        for (; c + 3 < C; c += 4) {  // four rows in flight
            typename Elem<TZ>::raw4 vr[4];
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) vr[u] = ldraw4(zp + (long)(c + u) * P);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = cvt4(vr[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sc = scale[c + u], sh = shift[c + u], wc = w[c + u];
                acc.x = fmaf(wc, fmaxf(fmaf(v[u].x, sc, sh), 0.f), acc.x);
                acc.y = fmaf(wc, fmaxf(fmaf(v[u].y, sc, sh), 0.f), acc.y);
                acc.z = fmaf(wc, fmaxf(fmaf(v[u].z, sc, sh), 0.f), acc.z);
                acc.w = fmaf(wc, fmaxf(fmaf(v[u].w, sc, sh), 0.f), acc.w);
            }
        }
        for (; c < C; ++c) {
            const float4 v = ld4(zp + (long)c * P);
            const float sc = scale[c], sh = shift[c], wc = w[c];
            acc.x = fmaf(wc, fmaxf(fmaf(v.x, sc, sh), 0.f), acc.x);
            acc.y = fmaf(wc, fmaxf(fmaf(v.y, sc, sh), 0.f), acc.y);
            acc.z = fmaf(wc, fmaxf(fmaf(v.z, sc, sh), 0.f), acc.z);
            acc.w = fmaf(wc, fmaxf(fmaf(v.w, sc, sh), 0.f), acc.w);
        }
        *(float4*)(out + (long)n * out_bs + p) = acc;
    } else {
        float a[4] = {b0, b0, b0, b0};
        for (int c = 0; c < C; ++c) {
            const float sc = scale[c], sh = shift[c], wc = w[c];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (p + q < P) a[q] = fmaf(wc, fmaxf(fmaf(ld1(zp + (long)c * P + q), sc, sh), 0.f), a[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (p + q < P) out[(long)n * out_bs + p + q] = a[q];
    }
}

int launch_outconv1_fwd(const void* z, int z_dt, long z_bs, const float* scale, const float* shift, const float* w,
                        const float* b, float* out, long out_bs, int N, int C, int P, hipStream_t st) {
    SMAAT_DISPATCH_ET(z_dt, TZ,
        hipLaunchKernelGGL(k_outconv1_fwd<TZ>, dim3(cdiv(P, 1024), N), dim3(256), 0, st, (const TZ*)z, z_bs, scale, shift, w,
                           b, out, out_bs, C, P););
    return (int)hipGetLastError();
}

int launch_reduce_rows(const float* part, int rows, long len, float* out, float alpha, hipStream_t st) {
    // NB: the partial buffer is scratch -- level 1 overwrites the head row of every slice.
    const int gx = cdiv(len, 256);
    int stride = 1;
    if (rows > 16 && (long)gx * 1 < 2048) {
        int slices = (2048 + gx - 1) / gx;   // aim at ~2048 blocks
        if (slices > 64) slices = 64;
        if (slices > rows / 4) slices = rows / 4;
        if (slices > 1) {
            stride = cdiv(rows, slices);
            const int ns = cdiv(rows, stride);
            hipLaunchKernelGGL(k_reduce_rows_l1, dim3(gx, ns), dim3(256), 0, st, (float*)part, rows, len, stride);
        }
    }
    hipLaunchKernelGGL(k_reduce_rows, dim3(gx), dim3(256), 0, st, part, rows, len, stride, out, alpha);
    return (int)hipGetLastError();
}

int launch_channel_sum(const void* x, int x_dt, long x_bs, int N, int C, int P, float* ws, float* out, hipStream_t st) {
    const int seg = plane_seg_len(P);
    dim3 grid(N * C, cdiv(P, seg));
    SMAAT_DISPATCH_ET(x_dt, T, hipLaunchKernelGGL(k_plane_sum<T>, grid, dim3(256), 0, st, (const T*)x, x_bs, C, P, seg, ws););
    const int slots = N * grid.y;
    hipLaunchKernelGGL(k_reduce_rows, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)ws, slots, (long)C, 1, out, 1.f);
    return (int)hipGetLastError();
}

int launch_copy_planes(const float* src, long s_bs, float* dst, long d_bs, int N, long plane_len, int accum,
                       hipStream_t st) {
    int gx = cdiv(plane_len, 4096);
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_copy_planes, dim3(gx, N), dim3(256), 0, st, src, s_bs, dst, d_bs, plane_len, accum);
    return (int)hipGetLastError();
}

// eval-mode BatchNorm coefficients from the running statistics (one launch instead of a chain of
// element-wise torch kernels: the batch-1 inference path is launch bound)
__global__ __launch_bounds__(256) void k_bn_eval_coefs(const float* __restrict__ rm, const float* __restrict__ rv,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int C,
                                                       float* __restrict__ st) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float m = rm[c];
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    st[c] = m;
    st[C + c] = is;
    st[2 * C + c] = sc;
    st[3 * C + c] = (beta ? beta[c] : 0.f) - m * sc;
}

int launch_bn_eval_coefs(const float* rm, const float* rv, const float* gamma, const float* beta, float eps, int C,
                         float* st, hipStream_t stream) {
    hipLaunchKernelGGL(k_bn_eval_coefs, dim3((C + 255) / 256), dim3(256), 0, stream, rm, rv, gamma, beta, eps, C, st);
    return (int)hipGetLastError();
}
