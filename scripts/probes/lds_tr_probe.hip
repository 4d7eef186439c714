// gfx950 probe: (1) lane / element mapping of ds_read_b64_tr_b16, (2) global_load_lds_dwordx4 from a source that is only
// 8-byte aligned (bf16 planes with P % 8 != 0), (3) global_load_lds_dword lane-linear image.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/lds_tr_probe.hip -o scripts/probes/lds_tr_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef short s4 __attribute__((ext_vector_type(4)));

__global__ void k_tr(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + lane * 4));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)a[j];
}

// addresses as the GEMM builds them: lane i of a 16-lane group supplies row (i >> 2) (row stride RS elements),
// columns 4 * (i & 3) .. + 3 of a [4][16] block; expected result: lane i gets column i of the block, rows 0..3
__global__ void k_tr2(unsigned short* out, int RS) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const int e = (i >> 2) * RS + 4 * (i & 3) + g * 16;
    s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + e));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)a[j];
}

__global__ void k_glds16(const unsigned short* src, int elem_off, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[512];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + elem_off + lane * 8),
                                     (void __attribute__((address_space(3)))*)(lds), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

__global__ void k_glds4(const unsigned short* src, int elem_off, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[128];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + elem_off + lane * 2),
                                     (void __attribute__((address_space(3)))*)(lds), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 128; i += 64) out[i] = lds[i];
}

int main() {
    unsigned short *d, *s;
    hipMalloc(&d, 8192);
    hipMalloc(&s, 8192);
    std::vector<unsigned short> h(4096), src(4096);
    for (int i = 0; i < 4096; ++i) src[i] = (unsigned short)(i + 1000);
    hipMemcpy(s, src.data(), 8192, hipMemcpyHostToDevice);

    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("tr_b16, addr = lane*8 B (element index lane*4): result[lane][j]\n");
    int okA = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, i = l & 15;
            const int src_lane = g * 16 + 4 * j + (i >> 2);
            if (h[l * 4 + j] != src_lane * 4 + (i & 3)) okA = 0;
        }
    }
    printf("TR_MODEL_A %s  (result[j] of lane i = element (i&3) of the 8 bytes addressed by lane 4j + (i>>2) of its 16-lane group)\n",
           okA ? "OK" : "MISMATCH");
    for (int RS : {16, 128, 144}) {
        hipLaunchKernelGGL(k_tr2, dim3(1), dim3(64), 0, 0, d, RS);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j)
                if (h[l * 4 + j] != j * RS + (l >> 4) * 16 + (l & 15)) ok = 0;
        printf("TR_GEMM_ADDR RS=%d %s\n", RS, ok ? "OK" : "MISMATCH");
    }
    for (int off : {0, 4, 2}) {  // byte offsets 0, 8, 4
        hipMemset(d, 0, 1024);
        hipLaunchKernelGGL(k_glds16, dim3(1), dim3(64), 0, 0, s, off, d);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess;
        for (int i = 0; i < 512; ++i)
            if (h[i] != src[off + i]) ok = 0;
        printf("GLDS16 src byte offset %d: %s (err %d) first %d %d %d %d\n", off * 2, ok ? "OK" : "MISMATCH", (int)e, h[0], h[1],
               h[8], h[9]);
    }
    for (int off : {0, 2}) {
        hipMemset(d, 0, 256);
        hipLaunchKernelGGL(k_glds4, dim3(1), dim3(64), 0, 0, s, off, d);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess;
        for (int i = 0; i < 128; ++i)
            if (h[i] != src[off + i]) ok = 0;
        printf("GLDS4 src byte offset %d: %s\n", off * 2, ok ? "OK" : "MISMATCH");
    }
    return 0;
}
