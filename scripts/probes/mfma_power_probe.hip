// gfx950 probe: what does v_mfma_f32_32x32x16_bf16 sustain for SECONDS with operands whose bits toggle like real data, and
// at which clock / power?  (mfma_chain_probe runs 26 ms on near-constant operands: 2.46 PFLOP/s.)  Run beside
// `rocm-smi --showclocks --showpower` samples (scripts/probes/mfma_power.sh).
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_power_probe.hip -o /tmp/mfma_power_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
// -DPROBE_F16=1 (round 4, for the two-term fp16 split studied in DESIGN 4.7): the same chains on v_mfma_f32_32x32x16_f16 with
// operands whose 10 mantissa bits are random -- does the fp16 pipe sustain what the bf16 pipe does under the power limit?
#ifndef PROBE_F16
#define PROBE_F16 0
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: constant operands; 1: pseudo-random operands, fixed per lane; 2: pseudo-random operands that change every trip
// (register-only xorshift: no memory traffic in any mode)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    auto rnd16 = [&]() {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        if (PROBE_F16) return (short)(((s >> 8) & 0x83FF) | 0x3C00);  // fp16: sign + 10 mantissa bits random, exponent of 1.0
        return (short)(((s >> 8) & 0x807F) | 0x3F00);  // sign + 7 mantissa bits random, exponent near 1: finite
    };
    bf16x8 a[3], b[3];
    for (int t = 0; t < 3; ++t)
        for (int e = 0; e < 8; ++e) {
            a[t][e] = MODE == 0 ? (short)(PROBE_F16 ? 0x3C00 : 0x3F80) : rnd16();
            b[t][e] = MODE == 0 ? (short)(PROBE_F16 ? 0x3C00 : 0x3F80) : rnd16();
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            for (int t = 0; t < 3; ++t) {
                a[t][it & 7] = rnd16();
                b[t][(it + 3) & 7] = rnd16();
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // the six-term order of the split GEMMs (fp16: the same six issue slots)
#if PROBE_F16
#define MF(A, B) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[i], 0, 0, 0)
#else
#define MF(A, B) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[i], 0, 0, 0)
#endif
            MF(a[0], b[2]);
            MF(a[2], b[0]);
            MF(a[1], b[1]);
            MF(a[0], b[1]);
            MF(a[1], b[0]);
            MF(a[0], b[0]);
#undef MF
        }
        if ((it & 63) == 63)  // keep the accumulators finite
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    float* d;
    hipMalloc(&d, 1024 * 256 * sizeof(float));
    const int iters = 20000, blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
        else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
    };
    launch();
    hipDeviceSynchronize();
    double total_ms = 0;
    int n = 0;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(e0);
        for (int j = 0; j < 10; ++j) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        n += 10;
    }
    const double flops = 2.0 * 32 * 32 * 16 * 24.0 * iters * (double)blocks * 4 * n;
    printf("mode %d: %.1f TFLOP/s sustained over %.1f s\n", mode, flops / (total_ms * 1e-3) / 1e12, total_ms * 1e-3);
    return 0;
}
