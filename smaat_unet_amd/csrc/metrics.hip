// On-device PrecipitationMetrics.update.
// reference: /root/reference/metric/precipitation_metrics.py:37-95 -- NaN check (:46-48), MSE sum of the normalised
// and of the de-normalised tensors divided by the batch size (:61-76), threshold -> 4-bin confusion counts
// (:78-95).  The reference does this with ~12 torch ops and a host sync (`.any()`) per step; here it is ONE
// streaming pass (two float4 loads per 4 pixels, nothing written but per-block partials) plus a one-block
// finish that adds the batch into the persistent metric state in a fixed order (deterministic, no atomics, no
// host synchronisation: a batch containing a NaN is skipped on the device and counted in state_i64[0]).
#include "common.h"

#define PM_THREADS 256
#define PM_MAX_BLOCKS 1024

struct PmAcc {
    float se, sed;        // sum (p - t)^2, sum (p*f - t*f)^2
    int tn, fp, fn, tp;   // confusion bins: target_mask * 2 + pred_mask
    int nan;
};

__device__ __forceinline__ void pm_one(PmAcc& a, float p, float t, float factor, float thr, int denorm) {
    a.nan |= (p != p) | (t != t);
    const float d = p - t;
    a.se = fmaf(d, d, a.se);
    // the reference multiplies both tensors by the factor first (:69-70) and by 12 next (:79-80)
    const float pu = denorm ? p * factor : p, tu = denorm ? t * factor : t;
    const float dd = pu - tu;
    a.sed = fmaf(dd, dd, a.sed);
    const bool pm = (pu * 12.f) > thr, tm = (tu * 12.f) > thr;
    a.tn += (!tm) & (!pm);
    a.fp += (!tm) & pm;
    a.fn += tm & (!pm);
    a.tp += tm & pm;
}

template <typename T>
__device__ __forceinline__ T pm_wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// partials: pf[block][2] (double), pi[block][5] (int): se, sed | tn, fp, fn, tp, nan
__global__ __launch_bounds__(PM_THREADS) void k_precip_metrics_partial(const float* __restrict__ preds,
                                                                       const float* __restrict__ target, long n,
                                                                       float factor, float thr, int denorm, int vec,
                                                                       double* __restrict__ pf, int* __restrict__ pi) {
    PmAcc a = {0.f, 0.f, 0, 0, 0, 0, 0};
    const long tid = (long)blockIdx.x * PM_THREADS + threadIdx.x, nthr = (long)gridDim.x * PM_THREADS;
    if (vec) {
        const long n4 = n >> 2;
        for (long q = tid; q < n4; q += nthr) {
            const float4 p = *(const float4*)(preds + 4 * q);
            const float4 t = *(const float4*)(target + 4 * q);
            pm_one(a, p.x, t.x, factor, thr, denorm);
            pm_one(a, p.y, t.y, factor, thr, denorm);
            pm_one(a, p.z, t.z, factor, thr, denorm);
            pm_one(a, p.w, t.w, factor, thr, denorm);
        }
        for (long i = 4 * n4 + tid; i < n; i += nthr) pm_one(a, preds[i], target[i], factor, thr, denorm);
    } else {
        for (long i = tid; i < n; i += nthr) pm_one(a, preds[i], target[i], factor, thr, denorm);
    }
    __shared__ double sd[2][PM_THREADS / 64];
    __shared__ int si[5][PM_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double se = pm_wave_sum((double)a.se), sed = pm_wave_sum((double)a.sed);
    const int tn = pm_wave_sum(a.tn), fp = pm_wave_sum(a.fp), fn = pm_wave_sum(a.fn), tp = pm_wave_sum(a.tp);
    const int nn = pm_wave_sum(a.nan);
    if (lane == 0) {
        sd[0][wave] = se;
        sd[1][wave] = sed;
        si[0][wave] = tn;
        si[1][wave] = fp;
        si[2][wave] = fn;
        si[3][wave] = tp;
        si[4][wave] = nn;
    }
    __syncthreads();
    if (threadIdx.x < 2) pf[(long)blockIdx.x * 2 + threadIdx.x] = (sd[threadIdx.x][0] + sd[threadIdx.x][1]) + (sd[threadIdx.x][2] + sd[threadIdx.x][3]);
    if (threadIdx.x >= 64 && threadIdx.x < 69) {
        const int k = threadIdx.x - 64;
        pi[(long)blockIdx.x * 5 + k] = (si[k][0] + si[k][1]) + (si[k][2] + si[k][3]);
    }
}

// state_f64: [0] total_loss, [1] total_loss_denorm;   state_i64: [0] batches skipped (NaN), [1] tn, [2] fp, [3] fn,
// [4] tp, [5] total_samples, [6] total_pixels        (reference add_state list :26-35)
__global__ __launch_bounds__(64) void k_precip_metrics_finish(const double* __restrict__ pf, const int* __restrict__ pi,
                                                              int nblocks, int batch, long n, int denorm,
                                                              double* __restrict__ state_f64,
                                                              long long* __restrict__ state_i64) {
    const int lane = threadIdx.x;
    double s[2] = {0.0, 0.0};
    long long c[5] = {0, 0, 0, 0, 0};
    for (int b = lane; b < nblocks; b += 64) {  // fixed assignment of partials to lanes, fixed shuffle tree
        s[0] += pf[(long)b * 2];
        s[1] += pf[(long)b * 2 + 1];
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] += pi[(long)b * 5 + k];
    }
    s[0] = pm_wave_sum(s[0]);
    s[1] = pm_wave_sum(s[1]);
#pragma unroll
    for (int k = 0; k < 5; ++k) c[k] = pm_wave_sum(c[k]);
    if (lane == 0) {
        if (c[4] != 0) {  // :46-48: a batch with a NaN leaves every state untouched
            state_i64[0] += 1;
        } else {
            state_f64[0] += s[0] / (double)batch;
            if (denorm) state_f64[1] += s[1] / (double)batch;
            state_i64[1] += c[0];
            state_i64[2] += c[1];
            state_i64[3] += c[2];
            state_i64[4] += c[3];
            state_i64[5] += batch;
            state_i64[6] += n;
        }
    }
}

static int pm_blocks(long n) {
    long b = (n + 4 * PM_THREADS * 4 - 1) / (4 * PM_THREADS * 4);  // >= 16 pixels per thread
    if (b < 1) b = 1;
    if (b > PM_MAX_BLOCKS) b = PM_MAX_BLOCKS;
    return (int)b;
}

long precip_metrics_ws_bytes(long n) { return (long)pm_blocks(n) * (2 * sizeof(double) + 5 * sizeof(int) + 4); }

int launch_precip_metrics_update(const float* preds, const float* target, long n, int batch, float factor, float thr,
                                 int denorm, void* ws, double* state_f64, long long* state_i64, hipStream_t st) {
    if (n <= 0 || batch <= 0 || ws == nullptr || state_f64 == nullptr || state_i64 == nullptr) return -1;
    if ((((uintptr_t)ws) & 7) != 0) return -1;
    const int nb = pm_blocks(n);
    double* pf = (double*)ws;
    int* pi = (int*)(pf + (long)nb * 2);
    const int vec = ((((uintptr_t)preds) & 15) == 0) && ((((uintptr_t)target) & 15) == 0);
    hipLaunchKernelGGL(k_precip_metrics_partial, dim3(nb), dim3(PM_THREADS), 0, st, preds, target, n, factor, thr, denorm,
                       vec, pf, pi);
    hipLaunchKernelGGL(k_precip_metrics_finish, dim3(1), dim3(64), 0, st, pf, pi, nb, batch, n, denorm, state_f64,
                       state_i64);
    return (int)hipGetLastError();
}
