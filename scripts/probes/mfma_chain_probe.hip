// gfx950 probe: throughput of v_mfma_f32_32x32x16_bf16 when consecutive MFMAs accumulate into the SAME accumulator
// (dependent chain, what the six-term split GEMMs issue per (channel tile, pixel tile)) against the same number of MFMAs
// interleaved over 2 / 4 independent accumulators.  One wave per SIMD (256 threads per CU), every CU busy.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_chain_probe.hip -o scripts/probes/mfma_chain_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)(0x3F80 + threadIdx.x % 3);
        b[e] = (short)(0x3F80 + threadIdx.x % 5);
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        // 24 MFMAs per iteration: round-robin over NACC accumulators (NACC = 1: one 24-deep dependent chain)
#pragma unroll
        for (int m = 0; m < 24; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the order the split GEMMs use today: 4 accumulators, SIX consecutive MFMAs into each before moving to the next
__global__ __launch_bounds__(256) void k_six(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)(0x3F80 + threadIdx.x % 3);
        b[e] = (short)(0x3F80 + threadIdx.x % 5);
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double run(F launch, int iters, int blocks, int threads) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 24.0 * iters * (double)blocks * (threads / 64);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    float* d;
    hipMalloc(&d, 1024 * 2048 * sizeof(float));
    const int iters = 20000, blocks = 256 * 4;
    printf("v_mfma_f32_32x32x16_bf16, 24 MFMAs per loop trip, %d blocks x 256 threads (one wave per SIMD per block)\n", blocks);
    printf("1 accumulator (24-deep dependent chain):      %7.1f TFLOP/s\n",
           run([&] { hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256), 0, 0, d, iters); }, iters, blocks, 256));
    printf("2 accumulators, alternating:                  %7.1f TFLOP/s\n",
           run([&] { hipLaunchKernelGGL((k<2, 1>), dim3(blocks), dim3(256), 0, 0, d, iters); }, iters, blocks, 256));
    printf("4 accumulators, round robin:                  %7.1f TFLOP/s\n",
           run([&] { hipLaunchKernelGGL((k<4, 1>), dim3(blocks), dim3(256), 0, 0, d, iters); }, iters, blocks, 256));
    printf("4 accumulators, six consecutive MFMAs each:   %7.1f TFLOP/s   (the split GEMMs' order)\n",
           run([&] { hipLaunchKernelGGL(k_six, dim3(blocks), dim3(256), 0, 0, d, iters); }, iters, blocks, 256));
    printf("1 accumulator, 2 waves per SIMD (512 threads): %7.1f TFLOP/s\n",
           run([&] { hipLaunchKernelGGL((k<1, 2>), dim3(blocks), dim3(512), 0, 0, d, iters); }, iters, blocks, 512));
    return 0;
}
