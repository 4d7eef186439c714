// Adam over every parameter tensor of the network in ONE launch (round 6).
//
// reference: optim.Adam(self.parameters(), lr=...) -- /root/reference/models/regression_lightning.py:48 and
// train_SmaAtUNet.py:182 (default betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).  The arithmetic is that of
// torch.optim.Adam's multi-tensor path, operation for operation in f32 (torch/optim/adam.py _multi_tensor_adam):
//     m  = lerp(m, g, 1 - beta1)                    = m + (1 - beta1) (g - m)
//     v  = v * beta2;   v = v + (1 - beta2) g g
//     d  = sqrt(v) / sqrt(1 - beta2^t) + eps
//     p  = p + (-lr / (1 - beta1^t)) * (m / d)
// with the scalars formed in double on the host and rounded to f32 once, as torch's scalar arguments are.  torch runs this as
// ~20 launches per step over the 145 tensors (foreach kernels take a bounded tensor list per launch: 0.2-0.3 ms per step on this
// network, three to five passes over every state tensor); here every element is read once (p, g, m, v) and written once (p, m, v).
//
// The parameter / state pointers live in a device table built once; the GRADIENT pointers change from step to step (autograd
// hands over fresh tensors), so they travel in the kernel argument block (<= SMAAT_ADAM_MAX pointers per launch, 2 KB).
#include "common.h"

#define SMAAT_ADAM_MAX 256  // tensors per launch (kernel arguments: 4 KB limit)
#define ADAM_EPB 1024       // elements per block: 256 threads x 4

struct AdamRow {  // one parameter tensor (device table, 32 bytes)
    float* p;
    float* m;
    float* v;
    long numel;
};
struct AdamGrads {
    const float* g[SMAAT_ADAM_MAX];
};
struct AdamK {  // scalars: w1 = 1 - beta1, w2 = 1 - beta2, bc2s = sqrt(1 - beta2^t), step = -lr / (1 - beta1^t)
    float w1, beta2, w2, bc2s, eps, step;
    int variant;  // arithmetic variant bits (tests / probes: which contractions torch's kernels were compiled with); 0 = default
};

// one element.  Variant bits: 1 lerp as fma(w1, g - m, m); 2 second moment as fma(w2 g, g, v beta2); 4 update as fma(step, m / d, p)
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, const float g, const AdamK k) {
#pragma clang fp contract(off)  // (the variant bits decide every contraction, not the compiler)
    const float diff = g - m;
    m = (k.variant & 1) ? fmaf(k.w1, diff, m) : m + k.w1 * diff;
    const float vb = v * k.beta2;
    v = (k.variant & 2) ? fmaf(k.w2 * g, g, vb) : vb + (k.w2 * g) * g;
    const float d = sqrtf(v) / k.bc2s + k.eps;
    const float q = m / d;
    p = (k.variant & 4) ? fmaf(k.step, q, p) : p + k.step * q;
}

// grid: total blocks of the launch; blk2t[b] = tensor (row of this launch) of block b, blk0[t] = first block of tensor t
__global__ __launch_bounds__(256) void k_adam_multi(const AdamRow* __restrict__ rows, const AdamGrads G,
                                                    const int* __restrict__ blk2t, const int* __restrict__ blk0,
                                                    const AdamK k) {
    const int b = blockIdx.x;
    const int t = blk2t[b];
    const AdamRow r = rows[t];
    const float* __restrict__ g = G.g[t];
    const long e0 = (long)(b - blk0[t]) * ADAM_EPB + 4 * threadIdx.x;
    if (e0 >= r.numel) return;
    const bool vec = e0 + 4 <= r.numel && ((((uintptr_t)r.p) | ((uintptr_t)r.m) | ((uintptr_t)r.v) | ((uintptr_t)g)) & 15) == 0;
    if (vec) {
        float4 p = *(const float4*)(r.p + e0), m = *(const float4*)(r.m + e0), v = *(const float4*)(r.v + e0);
        const float4 gg = *(const float4*)(g + e0);
        adam_one(p.x, m.x, v.x, gg.x, k);
        adam_one(p.y, m.y, v.y, gg.y, k);
        adam_one(p.z, m.z, v.z, gg.z, k);
        adam_one(p.w, m.w, v.w, gg.w, k);
        *(float4*)(r.p + e0) = p;
        *(float4*)(r.m + e0) = m;
        *(float4*)(r.v + e0) = v;
    } else {
        const long e1 = e0 + 4 < r.numel ? e0 + 4 : r.numel;
        for (long e = e0; e < e1; ++e) {
            float p = r.p[e], m = r.m[e], v = r.v[e];
            adam_one(p, m, v, g[e], k);
            r.p[e] = p;
            r.m[e] = m;
            r.v[e] = v;
        }
    }
}

int adam_max_tensors() { return SMAAT_ADAM_MAX; }
int adam_block_elems() { return ADAM_EPB; }

// rows: device table of n rows; grads: HOST array of n device pointers; blk2t [total_blocks], blk0 [n]: device.
// Scalars as torch/optim/adam.py forms them in Python floats (the CALLER does: the same libm, the same expressions):
//   w1 = 1 - beta1, w2 = 1 - beta2, bc2_sqrt = (1 - beta2 ** step) ** 0.5, step_size = (lr / (1 - beta1 ** step)) * -1
// each rounded to f32 once here, as torch's scalar arguments of f32 foreach kernels are.
int launch_adam_multi(const void* rows, const void* const* grads, const int* blk2t, const int* blk0, int n, int total_blocks,
                      double w1, double beta2, double w2, double bc2_sqrt, double eps, double step_size, int variant,
                      hipStream_t st) {
    if (n < 1 || n > SMAAT_ADAM_MAX || total_blocks < 1 || !rows || !grads || !blk2t || !blk0 || !(bc2_sqrt > 0.0)) return -1;
    AdamGrads G;
    for (int i = 0; i < n; ++i) {
        if (!grads[i]) return -1;
        G.g[i] = (const float*)grads[i];
    }
    for (int i = n; i < SMAAT_ADAM_MAX; ++i) G.g[i] = nullptr;
    AdamK k;
    k.w1 = (float)w1;
    k.beta2 = (float)beta2;
    k.w2 = (float)w2;
    k.bc2s = (float)bc2_sqrt;
    k.eps = (float)eps;
    k.step = (float)step_size;
    k.variant = variant;
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)total_blocks), dim3(256), 0, st, (const AdamRow*)rows, G, blk2t, blk0, k);
    return (int)hipGetLastError();
}
