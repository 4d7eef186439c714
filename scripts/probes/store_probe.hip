// Store-pattern probe (gfx950): how fast can 256 CUs write a [N][C][P] f32 tensor when each workgroup owns
// a (channels x pixels) tile, as the epilogue of the pointwise GEMMs does?  Variants differ in tile shape,
// store width and in which rows one store instruction touches.  Build: hipcc -O3 --offload-arch=gfx950
// store_probe.hip -o store_probe ; run: ./store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1);} } while (0)

// V: 0 contiguous float4 | 1 MFMA-layout dword (2 rows x 128 B per instr) | 2 transposed float4 (32 rows x 32 B)
//    3 row-major dword (1 row x 256 B per instr) | 4 row-major float4 (1 row x 1 KB per instr)
template <int V, int CT_ROWS, int PT_PX>
__global__ __launch_bounds__(256) void k_store(float* out, int N, int C, int P, int tiles_per_img, int nco, int items, int persistent) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    for (int it = blockIdx.x; it < items; it += persistent ? gridDim.x : items) {
        const int cot = it % nco, ptg = it / nco;
        const int n = ptg / tiles_per_img, tl = ptg - n * tiles_per_img;
        float* base = out + ((long)n * C + (long)cot * CT_ROWS) * P + (long)tl * PT_PX;
        const float v = (float)it;
        if (V == 0) {
            // same bytes, contiguous: item = CT_ROWS*PT_PX floats in a row
            float* b2 = out + (long)it * CT_ROWS * PT_PX;
            for (int i = tid * 4; i < CT_ROWS * PT_PX; i += 1024) *(float4*)(b2 + i) = make_float4(v, v, v, v);
        } else if (V == 1) {
            // 4 waves as 2 (rows) x 2 (px): wave tile = CT_ROWS/2 rows x PT_PX/2 px; per instr: rows {r, r+4} x 32 px
            const int wco = wave & 1, wpx = wave >> 1;
            for (int ct = 0; ct < CT_ROWS / 64; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wco * (CT_ROWS / 2) + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float* rp = base + (long)row * P + wpx * (PT_PX / 2) + l31;
                    for (int pt = 0; pt < PT_PX / 64; ++pt) rp[pt * 32] = v;
                }
        } else if (V == 2) {
            // transposed accumulators: lane -> row (32 rows), regs -> 4 consecutive px; instr = 32 rows x (2 x 16 B)
            const int wco = wave & 1, wpx = wave >> 1;
            for (int ct = 0; ct < CT_ROWS / 64; ++ct) {
                const int row = wco * (CT_ROWS / 2) + ct * 32 + l31;
                float* rp = base + (long)row * P + wpx * (PT_PX / 2);
                for (int pt = 0; pt < PT_PX / 64; ++pt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(float4*)(rp + pt * 32 + 8 * q + 4 * half) = make_float4(v, v, v, v);
            }
        } else if (V == 3) {
            // row-major dword: wave w owns rows w, w+4, ...; instr = 1 row x 64 px
            for (int row = wave; row < CT_ROWS; row += 4) {
                float* rp = base + (long)row * P + lane;
                for (int c = 0; c < PT_PX; c += 64) rp[c] = v;
            }
        } else if (V == 4) {
            // row-major float4: instr = 1 row x 256 px (or the whole row piece when PT_PX < 256)
            constexpr int LPR = PT_PX / 4 < 64 ? PT_PX / 4 : 64;  // lanes per row
            constexpr int RPI = 64 / LPR;                         // rows per instruction
            const int lr = lane / LPR, lc = lane % LPR;
            for (int row = wave * RPI + lr; row < CT_ROWS; row += 4 * RPI) {
                float* rp = base + (long)row * P + lc * 4;
                for (int c = 0; c < PT_PX; c += LPR * 4) *(float4*)(rp + c) = make_float4(v, v, v, v);
            }
        }
    }
}

template <int V, int CT_ROWS, int PT_PX>
static void run(const char* name, float* out, int N, int C, int P, int persistent) {
    const int tiles = P / PT_PX, nco = C / CT_ROWS, items = N * tiles * nco;
    const int grid = persistent ? persistent : items;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_store<V, CT_ROWS, PT_PX>), dim3(grid), dim3(256), 0, 0, out, N, C, P, tiles, nco, items, persistent);
    CK(hipEventRecord(e0));
    const int it = 5;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_store<V, CT_ROWS, PT_PX>), dim3(grid), dim3(256), 0, 0, out, N, C, P, tiles, nco, items, persistent);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double gb = (double)N * C * P * 4 / 1e9;
    printf("%-58s grid %6d  %7.3f ms  %7.1f GB/s\n", name, grid, ms, gb / ms * 1e3);
}

int main() {
    const int N = 32, C = 256, P = 288 * 288;  // 82944 = 648 * 128 = 162 * 512
    float* out; CK(hipMalloc(&out, (size_t)N * C * P * 4));
    run<0, 128, 128>("warm-up", out, N, C, P, 0);
    const int grids[] = {0, 512, 256, 128};  // 0 = one workgroup per tile; else persistent with that many workgroups (4 waves each)
    for (int gi = 0; gi < 4; ++gi) {
        const int p = grids[gi];
        run<1, 128, 128>("128 rows x 128 px, MFMA-layout dword (2 rows x 128 B)", out, N, C, P, p);
        run<2, 128, 128>("128 rows x 128 px, transposed float4 (32 rows x 32 B)", out, N, C, P, p);
        run<3, 128, 128>("128 rows x 128 px, row-major dword (1 row x 256 B)", out, N, C, P, p);
        run<4, 128, 128>("128 rows x 128 px, row-major float4 (2 rows x 512 B)", out, N, C, P, p);
        run<4, 32, 512>("32 rows x 512 px, row-major float4 (1 row x 1 KB)", out, N, C, P, p);
    }
    return 0;
}
