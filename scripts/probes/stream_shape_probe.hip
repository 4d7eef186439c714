// Why do the step's streaming kernels reach 5.0-5.6 TB/s where a linear triad reaches 6.0?  The access SHAPE of k_bn_bwd_apply (bn.hip)
// rebuilt step by step on 680 MB tensors [32][64][288 x 288] (2 reads + 1 write):
//   A  linear triad, one float4 per thread                                    (nt_stream_probe.hip: 6.0 TB/s)
//   B  grid (planes, segments of SEG elements), block walks its segment two positions per trip  (the kernel's loop)
//   C  B + the per-channel coefficients (7 scalar loads before the first vector load)
//   D  grid (segments, planes): consecutive workgroups walk consecutive addresses
//   E  D with the coefficients
// hipcc --offload-arch=gfx950 -O3 stream_shape_probe.hip -o /tmp/ssp && /tmp/ssp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_lin(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    o[i] = a[i] * 1.5f + b[i];
}

template <bool COEF, bool SEGX>
__global__ __launch_bounds__(256) void k_seg(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                             const float* __restrict__ co, int C, int P, int seg) {
    const int plane = SEGX ? blockIdx.y : blockIdx.x, sg = SEGX ? blockIdx.x : blockIdx.y;
    const int c = plane % C;
    float s0 = 1.5f, s1 = 0.f, s2 = 0.f, s3 = 1.f, s4 = 0.f, s5 = 0.f, s6 = 0.f;
    if (COEF) {
        s0 = co[c]; s1 = co[C + c]; s2 = co[2 * C + c]; s3 = co[3 * C + c]; s4 = co[4 * C + c]; s5 = co[5 * C + c]; s6 = co[6 * C + c];
    }
    const float* ap = a + (long)plane * P;
    const float* bp = b + (long)plane * P;
    float* op = o + (long)plane * P;
    const int p0 = sg * seg;
    int p1 = p0 + seg;
    if (p1 > P) p1 = P;
    auto one = [&](const f4 x, const f4 y, int p) {
        f4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g = y[j];
            if (!(fmaf(x[j], s0, s1) > s6)) g = 0.f;
            r[j] = s3 * (g - s4 - (x[j] - s2) * s5);
        }
        *(f4*)(op + p) = r;
    };
    int p = p0 + threadIdx.x * 4;
    for (; p + 1024 < p1; p += 2048) {
        const f4 xa = *(const f4*)(ap + p), ya = *(const f4*)(bp + p);
        const f4 xb = *(const f4*)(ap + p + 1024), yb = *(const f4*)(bp + p + 1024);
        one(xa, ya, p);
        one(xb, yb, p + 1024);
    }
    if (p < p1) one(*(const f4*)(ap + p), *(const f4*)(bp + p), p);
}

__global__ void k_fill(float* p, long n, unsigned seed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(int)(h & 0xFFFFFF) * (1.f / 8388608.f) - 1.f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <typename F>
static float timeit(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int N = 32, C = 64, P = 288 * 288;
    const long n = (long)N * C * P;
    const size_t bytes = (size_t)n * 4;
    float *a, *b, *o, *co;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&o, bytes));
    CK(hipMalloc(&co, 7 * C * 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, n, 1u);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, b, n, 2u);
    hipLaunchKernelGGL(k_fill, dim3(2), dim3(256), 0, 0, co, 7L * C, 3u);
    const double gb = 3.0 * bytes / 1e9;
    float t = timeit([&] { hipLaunchKernelGGL(k_lin, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (const f4*)a, (const f4*)b, (f4*)o, n / 4); }, 20);
    printf("A linear triad                                   : %7.1f us  %6.0f GB/s\n", t * 1e3, gb / t * 1e3);
    for (int seg : {2048, 8192, 20736, 82944}) {
        const int ns = (P + seg - 1) / seg;
        t = timeit([&] { hipLaunchKernelGGL((k_seg<false, false>), dim3(N * C, ns), dim3(256), 0, 0, a, b, o, co, C, P, seg); }, 20);
        printf("B grid (planes, segments)  seg %6d              : %7.1f us  %6.0f GB/s\n", seg, t * 1e3, gb / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_seg<true, false>), dim3(N * C, ns), dim3(256), 0, 0, a, b, o, co, C, P, seg); }, 20);
        printf("C   + per-channel coefficients                   : %7.1f us  %6.0f GB/s\n", t * 1e3, gb / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_seg<false, true>), dim3(ns, N * C), dim3(256), 0, 0, a, b, o, co, C, P, seg); }, 20);
        printf("D grid (segments, planes)  seg %6d              : %7.1f us  %6.0f GB/s\n", seg, t * 1e3, gb / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_seg<true, true>), dim3(ns, N * C), dim3(256), 0, 0, a, b, o, co, C, P, seg); }, 20);
        printf("E   + per-channel coefficients                   : %7.1f us  %6.0f GB/s\n", t * 1e3, gb / t * 1e3);
    }
    return 0;
}
