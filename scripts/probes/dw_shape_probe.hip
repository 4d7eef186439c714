// The memory shape of the depthwise row-walking forward against a non-walking one, as pure data movement
// (x [32][64][288][288] f32 -> y [32][128][288][288]: 1 read + 2 writes = 2.04 GB):
//   W  walker: a lane owns 4 columns of one band of BH rows and walks down, one float4 load per row (2 rows in flight), two float4
//      stores per row (the two k-rows of its channel); waves in (plane, band, column group) order            [k_dw3x3_fwd_rows]
//   L  linear: a thread owns ONE output float4 position; loads the three input rows r-1, r, r+1 (vertical re-use left to L1 / L2)
//      and stores two float4; workgroups in address order of the input plane
//   L2 linear, two rows per thread (4 loads, 4 stores)
// hipcc --offload-arch=gfx950 -O3 dw_shape_probe.hip -o /tmp/dsp && /tmp/dsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_walk(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, int BH, int nb,
                                              int wpp, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= nwaves) return;
    const int plane = (int)(gw / wpp), wip = (int)(gw - (long)plane * wpp);
    const int ncol4 = W / 4;
    const int t = wip * 64 + lane;
    int band = t / ncol4;
    const int q = t - band * ncol4;
    const bool active = band < nb;
    if (!active) band = nb - 1;
    const int r0 = band * BH;
    const long P = (long)H * W;
    const float* xp = x + (long)plane * P + 4 * q;
    const int n = plane / C, ci = plane - n * C;
    float* yp = y + ((long)n * 2 * C + 2 * ci) * P + 4 * q;
    auto ld = [&](int r) { const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r); return *(const f4*)(xp + (long)rc * W); };
    f4 a = ld(r0 - 1), b = ld(r0), c = ld(r0 + 1), d = ld(r0 + 2);
    for (int i = 0; i < BH; ++i) {
        const int r = r0 + i;
        const f4 e = ld(r + 3);
        const f4 s = a + b + c;
        if (active && r < H) {
            *(f4*)(yp + (long)r * W) = s;
            *(f4*)(yp + P + (long)r * W) = s * 2.f;
        }
        a = b; b = c; c = d; d = e;
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_lin(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, long total) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ncol4 = W / 4, nrg = H / R;
    const long per = (long)ncol4 * nrg;
    const int plane = (int)(gid / per);
    const int rem = (int)(gid - (long)plane * per);
    const int rg = rem / ncol4, q = rem - rg * ncol4;
    const long P = (long)H * W;
    const float* xp = x + (long)plane * P + 4 * q;
    const int n = plane / C, ci = plane - n * C;
    float* yp = y + ((long)n * 2 * C + 2 * ci) * P + 4 * q;
    auto ld = [&](int r) { const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r); return *(const f4*)(xp + (long)rc * W); };
    f4 v[R + 2];
#pragma unroll
    for (int k = 0; k < R + 2; ++k) v[k] = ld(rg * R - 1 + k);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const f4 s = v[k] + v[k + 1] + v[k + 2];
        const int r = rg * R + k;
        *(f4*)(yp + (long)r * W) = s;
        *(f4*)(yp + P + (long)r * W) = s * 2.f;
    }
}

// ---- the backward's shape: reads dY (two planes per input channel) and x, writes dX (3 reads + 1 write per input channel) ----
__global__ __launch_bounds__(256) void k_walk_b(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int C, int H,
                                                int W, int BH, int nb, int wpp, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= nwaves) return;
    const int plane = (int)(gw / wpp), wip = (int)(gw - (long)plane * wpp);
    const int ncol4 = W / 4;
    const int t = wip * 64 + lane;
    int band = t / ncol4;
    const int q = t - band * ncol4;
    const bool active = band < nb;
    if (!active) band = nb - 1;
    const int r0 = band * BH;
    const long P = (long)H * W;
    const int n = plane / C, ci = plane - n * C;
    const float* xp = x + (long)plane * P + 4 * q;
    const float* gp = dy + ((long)n * 2 * C + 2 * ci) * P + 4 * q;
    float* op = dx + (long)plane * P + 4 * q;
    auto cl = [&](int r) { return r < 0 ? 0 : (r >= H ? H - 1 : r); };
    f4 a0 = *(const f4*)(gp + (long)cl(r0 - 1) * W), a1 = *(const f4*)(gp + P + (long)cl(r0 - 1) * W);
    f4 b0 = *(const f4*)(gp + (long)cl(r0) * W), b1 = *(const f4*)(gp + P + (long)cl(r0) * W);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < BH; ++i) {
        const int r = r0 + i;
        const f4 c0 = *(const f4*)(gp + (long)cl(r + 1) * W), c1 = *(const f4*)(gp + P + (long)cl(r + 1) * W);
        const f4 xv = *(const f4*)(xp + (long)cl(r) * W);
        const f4 s = a0 + a1 + b0 + b1 + c0 + c1;
        acc += xv * b0;
        if (active && r < H) *(f4*)(op + (long)r * W) = s;
        a0 = b0; a1 = b1; b0 = c0; b1 = c1;
    }
    if (acc.x == 123.456f) op[0] = acc.y;  // (keeps the x loads alive)
}
template <int R>
__global__ __launch_bounds__(256) void k_lin_b(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int C, int H,
                                               int W, long total) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ncol4 = W / 4, nrg = H / R;
    const long per = (long)ncol4 * nrg;
    const int plane = (int)(gid / per);
    const int rem = (int)(gid - (long)plane * per);
    const int rg = rem / ncol4, q = rem - rg * ncol4;
    const long P = (long)H * W;
    const int n = plane / C, ci = plane - n * C;
    const float* xp = x + (long)plane * P + 4 * q;
    const float* gp = dy + ((long)n * 2 * C + 2 * ci) * P + 4 * q;
    float* op = dx + (long)plane * P + 4 * q;
    auto cl = [&](int r) { return r < 0 ? 0 : (r >= H ? H - 1 : r); };
    f4 g0[R + 2], g1[R + 2], xv[R + 2];
#pragma unroll
    for (int k = 0; k < R + 2; ++k) {
        g0[k] = *(const f4*)(gp + (long)cl(rg * R - 1 + k) * W);
        g1[k] = *(const f4*)(gp + P + (long)cl(rg * R - 1 + k) * W);
        xv[k] = *(const f4*)(xp + (long)cl(rg * R - 1 + k) * W);
    }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const f4 s = g0[k] + g1[k] + g0[k + 1] + g1[k + 1] + g0[k + 2] + g1[k + 2];
        acc += (xv[k] + xv[k + 1] + xv[k + 2]) * g0[k + 1];
        *(f4*)(op + (long)(rg * R + k) * W) = s;
    }
    if (acc.x == 123.456f) op[0] = acc.y;
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
    const int N = 32, C = 64, H = 288, W = 288;
    const long P = (long)H * W, nx = (long)N * C * P;
    float *x, *y;
    CK(hipMalloc(&x, nx * 4));
    CK(hipMalloc(&y, nx * 8));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, 0, x, nx, 1u);
    const double gb = 3.0 * nx * 4 / 1e9;
    for (int nb : {8, 16, 24, 4}) {
        const int BH = (H + nb - 1) / nb, T = (W / 4) * nb, wpp = (T + 63) / 64;
        const long nwaves = (long)N * C * wpp;
        float t = timeit([&] { hipLaunchKernelGGL(k_walk, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, 0, x, y, C, H, W, BH, nb, wpp, nwaves); }, 20);
        printf("W  walker, %2d bands of %3d rows (+2 halo rows read)  : %7.1f us  %6.0f GB/s algorithmic\n", nb, BH, t * 1e3, gb / t * 1e3);
    }
    {
        const long total = (long)N * C * (W / 4) * H;
        float t = timeit([&] { hipLaunchKernelGGL((k_lin<1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, x, y, C, H, W, total); }, 20);
        printf("L  linear, one output row per thread (3 loads)        : %7.1f us  %6.0f GB/s algorithmic\n", t * 1e3, gb / t * 1e3);
        const long total2 = total / 2;
        t = timeit([&] { hipLaunchKernelGGL((k_lin<2>), dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, 0, x, y, C, H, W, total2); }, 20);
        printf("L2 linear, two output rows per thread (4 loads)       : %7.1f us  %6.0f GB/s algorithmic\n", t * 1e3, gb / t * 1e3);
        const long total4 = total / 4;
        t = timeit([&] { hipLaunchKernelGGL((k_lin<4>), dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, 0, x, y, C, H, W, total4); }, 20);
        printf("L4 linear, four output rows per thread (6 loads)      : %7.1f us  %6.0f GB/s algorithmic\n", t * 1e3, gb / t * 1e3);
    }
    {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((2 * nx + 255) / 256)), dim3(256), 0, 0, y, 2 * nx, 5u);
        float* dx;
        CK(hipMalloc(&dx, nx * 4));
        const double gbb = 4.0 * nx * 4 / 1e9;
        for (int nb : {8, 24}) {
            const int BH = (H + nb - 1) / nb, T = (W / 4) * nb, wpp = (T + 63) / 64;
            const long nwaves = (long)N * C * wpp;
            float t = timeit([&] { hipLaunchKernelGGL(k_walk_b, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, 0, x, y, dx, C, H, W, BH, nb, wpp, nwaves); }, 20);
            printf("backward shape W  walker, %2d bands of %3d rows           : %7.1f us  %6.0f GB/s algorithmic\n", nb, BH, t * 1e3, gbb / t * 1e3);
        }
        const long total = (long)N * C * (W / 4) * H;
        float t = timeit([&] { hipLaunchKernelGGL((k_lin_b<1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, x, y, dx, C, H, W, total); }, 20);
        printf("backward shape L  linear, one row per thread (9 loads)  : %7.1f us  %6.0f GB/s algorithmic\n", t * 1e3, gbb / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_lin_b<2>), dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, 0, x, y, dx, C, H, W, total / 2); }, 20);
        printf("backward shape L2 linear, two rows per thread (12 loads): %7.1f us  %6.0f GB/s algorithmic\n", t * 1e3, gbb / t * 1e3);
        t = timeit([&] { hipLaunchKernelGGL((k_lin_b<4>), dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, 0, x, y, dx, C, H, W, total / 4); }, 20);
        printf("backward shape L4 linear, four rows per thread (18 loads): %7.1f us  %6.0f GB/s algorithmic\n", t * 1e3, gbb / t * 1e3);
    }
    return 0;
}
