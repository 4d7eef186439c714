// Argument blocks of the row-walking kernels (dsrows.hip, dswgrad.hip), shared with the C ABI (capi.hip): ONE declaration, so
// that a field added for a kernel cannot silently disagree with the caller's copy.
#pragma once

struct DsRowsArgs {
    const void* x;
    long x_bs;
    const float* in_scale;
    const float* in_shift;
    const float* w_dw;             // [K][9]
    const float* b_dw;             // [K] or null
    const unsigned short* planes;  // pointwise weight images, chunk-major [K/16][NPL][M][16] (NPL = 3 split planes | 1)
    const float* bias;             // [M] or null
    void* out;
    long out_bs;
    float* part;  // [3][items][M] or null
    int N, Cin, K, M, H, W, P;
    int nsplit, strips, bands, RB, items, ips, npl;
    int ilv;   // 1: the workgroups of an XCD take its items round-robin (set by the launcher)
    int relu;  // 1: the output is max(z + bias, 0) (inference: BatchNorm folded into the weights, ReLU in the epilogue)
    unsigned* y_amax;  // nullable amax buffer (common.h): receives max |y| of the depthwise output the producers form -- the scale
                       // of the two-term fp16 recompute weight gradient that re-forms the same y in the backward
    // two-term fp16 split (NT == 2; all null otherwise): planes = the fp16 image of smaat_split_planes_h ([K/16][2][M][16]),
    // a_kexp = its exponent (the image's trailer), x_amax (+ x_amax2: a second writer of x, nullable) = amax buffers holding
    // max |x| of the tensor x points into (the RAW tensor when in_scale is given)
    const unsigned* x_amax;
    const unsigned* x_amax2;
    // ... or, zb_w != null: x = zb_w [Cin][zb_K] . u + zb_b is the output of the previous pointwise convolution and x_amax holds
    // max |u| of ITS operand: the kernel bounds |x[c]| <= sum_k |zb_w[c][k]| max|u| + |zb_b[c]| per channel
    const float* zb_w;
    const float* zb_b;  // [Cin] or null
    int zb_K;
    const int* a_kexp;
    unsigned* z_amax;  // nullable amax buffer: receives max |out| (any NT)
};

struct DsWgArgs {
    const void* x;   // TX
    long x_bs;
    const float* in_scale;
    const float* in_shift;
    const float* w_dw;  // [K][9]
    const float* b_dw;  // [K] or null
    const void* dz;  // TG
    long dz_bs;
    float* part;  // [nsplit][M][K]
    int N, Cin, K, M, H, W, P;
    int nkt, nsplit, strips, bands, RB, items, ips;
    int ilv;  // 1: the workgroups of an XCD take its items round-robin (neighbouring strips run at the same time)
    const unsigned* y_amax;   // NT == 2 (two-term fp16 split): amax buffers of the depthwise output (from the forward) and of dz
    const unsigned* dz_amax;
};

// fused backward of a DepthwiseSeparableConv (dsbwd.hip): dz -> dY (MFMA, on chip) -> dX, depthwise weight / bias gradient
// partials, and the backward sums of the previous BatchNorm when its activation is applied on load
struct DsBwArgs {
    const float* x;   // [N][Cin][H][W]: the depthwise input (the PRE-BatchNorm tensor of the previous half when in_scale != null)
    long x_bs;
    const float* in_scale;  // [Cin] or null: previous BatchNorm + ReLU applied on load
    const float* in_shift;
    const float* bn_mean;    // with in_scale: the previous BatchNorm's batch statistics (for rpart)
    const float* bn_invstd;
    const float* dz;  // [N][M][H][W]
    long dz_bs;
    const unsigned* dz_amax;         // amax buffer of dz (common.h)
    const unsigned short* planes_t;  // fp16 image of w_pw^T: [M/16][2][K][16] (smaat_split_planes_h, src_t = 1)
    const int* a_kexp;               // its power-of-two exponent (the image's trailer)
    const float* w_dw;               // [K][9]
    float* dx;        // [N][Cin][H][W]
    long dx_bs;
    float* part;      // [rows][K][10]: depthwise weight (9 taps) + bias gradient partials, rows = dsconv_bwd_rows_num_rows()
    float* rpart;     // [2][rows][Cin] or null: sum g, sum g * xhat of the previous BatchNorm (g = dX * [act > 0])
    int N, Cin, K, M, H, W, P;
    int strips, bands, RB, items, ips, nhalf, wgh;  // set by the launcher
    unsigned x_bytes, dz_bytes, dx_bytes;           // buffer descriptor ranges (set by the launcher)
};
