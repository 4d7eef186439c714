// Streaming-rate probe: do non-temporal hints change what a read/write stream reaches on MI355X?
// out = a * s + b over 680 MB tensors (the size of a 64-channel 288 x 288 activation at batch 32) with plain / non-temporal loads and
// stores, a copy and a two-stream reduction; 16 bytes per lane, 256-thread blocks, one float4 per thread (the shape of the streaming
// kernels of the step).  hipcc --offload-arch=gfx950 -O3 nt_stream_probe.hip -o /tmp/nt_probe && /tmp/nt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_triad(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o, float s, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f4 x = NTL ? __builtin_nontemporal_load(a + i) : a[i];
    const f4 y = NTL ? __builtin_nontemporal_load(b + i) : b[i];
    const f4 r = x * s + y;
    if (NTS)
        __builtin_nontemporal_store(r, o + i);
    else
        o[i] = r;
}
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f4 x = NTL ? __builtin_nontemporal_load(a + i) : a[i];
    if (NTS)
        __builtin_nontemporal_store(x, o + i);
    else
        o[i] = x;
}
// four float4 per thread, strided by the grid (more bytes in flight per wave)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_triad4(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o, float s, long n) {
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
    f4 x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = i0 + k * 256;
        if (i < n) {
            x[k] = NTL ? __builtin_nontemporal_load(a + i) : a[i];
            y[k] = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = i0 + k * 256;
        if (i < n) {
            const f4 r = x[k] * s + y[k];
            if (NTS)
                __builtin_nontemporal_store(r, o + i);
            else
                o[i] = r;
        }
    }
}

__global__ void k_fill(float* p, long n, unsigned seed) {  // pseudo-random finite values (the rates above are NOT those of zero pages)
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
    const long n4 = 64L * 82944 * 32 / 4;  // float4 elements of one tensor (680 MB)
    const size_t bytes = (size_t)n4 * 16;
    f4 *a, *b, *o, *o2;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&o, bytes));
    CK(hipMalloc(&o2, bytes));
    if (getenv("NT_PROBE_ZERO")) {
        CK(hipMemset(a, 0, bytes));
        CK(hipMemset(b, 0, bytes));
        printf("inputs: zero pages\n");
    } else {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n4 * 4 + 255) / 256)), dim3(256), 0, 0, (float*)a, n4 * 4, 1u);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n4 * 4 + 255) / 256)), dim3(256), 0, 0, (float*)b, n4 * 4, 2u);
        printf("inputs: pseudo-random values\n");
    }
    const int g1 = (int)((n4 + 255) / 256), g4 = (int)((n4 + 1023) / 1024);
    const double gb3 = 3.0 * bytes / 1e9, gb2 = 2.0 * bytes / 1e9;
    for (int round = 0; round < 1; ++round) {
        float t;
        t = timeit([&] { hipLaunchKernelGGL((k_triad<false, false>), dim3(g1), dim3(256), 0, 0, a, b, o, 1.5f, n4); }, 20);
        printf("triad  (2 reads + 1 write) plain loads,  plain stores : %7.1f us  %6.0f GB/s\n", t * 1e3, gb3 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_triad<true, false>), dim3(g1), dim3(256), 0, 0, a, b, o, 1.5f, n4); }, 20);
        printf("triad                      nt loads,     plain stores : %7.1f us  %6.0f GB/s\n", t * 1e3, gb3 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_triad<false, true>), dim3(g1), dim3(256), 0, 0, a, b, o, 1.5f, n4); }, 20);
        printf("triad                      plain loads,  nt stores    : %7.1f us  %6.0f GB/s\n", t * 1e3, gb3 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_triad<true, true>), dim3(g1), dim3(256), 0, 0, a, b, o, 1.5f, n4); }, 20);
        printf("triad                      nt loads,     nt stores    : %7.1f us  %6.0f GB/s\n", t * 1e3, gb3 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_triad4<false, false>), dim3(g4), dim3(256), 0, 0, a, b, o, 1.5f, n4); }, 20);
        printf("triad x4 per thread        plain loads,  plain stores : %7.1f us  %6.0f GB/s\n", t * 1e3, gb3 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_triad4<true, true>), dim3(g4), dim3(256), 0, 0, a, b, o, 1.5f, n4); }, 20);
        printf("triad x4 per thread        nt loads,     nt stores    : %7.1f us  %6.0f GB/s\n", t * 1e3, gb3 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(g1), dim3(256), 0, 0, a, o, n4); }, 20);
        printf("copy   (1 read + 1 write)  plain                      : %7.1f us  %6.0f GB/s\n", t * 1e3, gb2 / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_copy<true, true>), dim3(g1), dim3(256), 0, 0, a, o, n4); }, 20);
        printf("copy                       nt loads,     nt stores    : %7.1f us  %6.0f GB/s\n", t * 1e3, gb2 / t * 1e3);
        // producer -> consumer: the triad writes o, a second kernel reads o (and a) and writes o2 -- does an nt store hurt the reader?
        t = timeit([&] {
            hipLaunchKernelGGL((k_triad<false, false>), dim3(g1), dim3(256), 0, 0, a, b, o, 1.5f, n4);
            hipLaunchKernelGGL((k_triad<false, false>), dim3(g1), dim3(256), 0, 0, o, a, o2, 0.5f, n4);
        }, 10);
        printf("producer + consumer        plain / plain              : %7.1f us  %6.0f GB/s\n", t * 1e3, 2 * gb3 / t * 1e3);
        t = timeit([&] {
            hipLaunchKernelGGL((k_triad<true, true>), dim3(g1), dim3(256), 0, 0, a, b, o, 1.5f, n4);
            hipLaunchKernelGGL((k_triad<true, true>), dim3(g1), dim3(256), 0, 0, o, a, o2, 0.5f, n4);
        }, 10);
        printf("producer + consumer        nt / nt                    : %7.1f us  %6.0f GB/s\n", t * 1e3, 2 * gb3 / t * 1e3);
    }
    return 0;
}
