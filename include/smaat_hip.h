/* smaat_hip.h -- C ABI of libsmaat_hip.so: the MI355X (gfx950) kernels behind the
 * SmaAt-UNet forward+backward hot path.
 *
 * The reference (HansBambel/SmaAt-UNet) has no FFI: the path sits behind torch.nn
 * modules.  Each entry point below replaces the torch.nn call(s) cited next to it; the
 * Python host (smaat_unet_amd/) binds them with ctypes and exposes them as
 * torch.ops.smaat.* custom ops behind module classes with the reference's constructor
 * signatures and state_dict keys.
 *
 * Conventions
 *  - all tensors are float32 (bf16 storage: the "mixed precision" section), NCHW, device pointers; "plane" = H*W contiguous floats;
 *    channel stride = H*W; batch stride is passed explicitly (`*_bs`, in elements) so
 *    that a tensor may be a channel slice of a larger concatenation buffer.
 *  - no allocation, no synchronisation inside: workspaces are passed in, kernels are enqueued on `stream` (a
 *    hipStream_t).  Element types are per-call arguments (dtype codes of the "mixed precision" section).  ONE process-wide
 *    setting exists: smaat_set_split_mode, the A/B switch of the f32-storage matrix path (exact three-term split | single
 *    bf16 term); hosts that need both in one process call it around the launches concerned (it is read at launch time).
 *    (Deliberately NOT thread-local: PyTorch's autograd engine launches the backward kernels from its own device thread,
 *    which must see the mode the forward ran under; tried and reverted in round 4.)
 *  - return value: 0 = ok, >0 = hipError_t, -1 = invalid argument, -2 = shape / alignment not taken by this entry point
 *    (the caller uses the general one; documented per entry point).
 */
#ifndef SMAAT_HIP_H
#define SMAAT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int smaat_abi_version(void);

/* ---- DepthwiseSeparableConv: depthwise 3x3 (pad 1, groups=Cin, kpl outputs per input
 *      channel) fused through LDS into the pointwise 1x1 on the f32 MFMA pipe.
 *      reference: models/layers.py:34-50 (depthwise :38-44, pointwise :45, forward :47-50)
 *   x        [N][Cin][H][W]            (optional in_scale/in_shift[Cin]: x := relu(x*sc+sh) on load)
 *   w_dw     [Cin*kpl][9], b_dw [Cin*kpl] (nullable)
 *   wt_pw    [Cin*kpl][Cout]  = pointwise.weight TRANSPOSED (k-major), b_pw [Cout] (nullable)
 *   z        [N][Cout][H][W]
 *   part     nullable; [3][slots][Cout] per-tile (mean, M2 = sum of squared deviations from that mean, pixel
 *            count) of (z - b_pw) for the following train-mode BatchNorm; slots = smaat_pw_num_slots(N,H,W,Cout).
 *            smaat_bn_finalize merges the tiles pairwise in fp64 (no E[z^2] - E[z]^2 cancellation).
 *   y_out    nullable; [N][Cin*kpl][H][W] depthwise output, written as a side product so that the
 *            backward pass can form the pointwise weight gradient as one streamed GEMM
 */
int smaat_pw_num_slots(int N, int H, int W, int M);
int smaat_dsconv_fwd(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                     const float* b_dw, const float* wt_pw, const float* b_pw, float* z, long z_bs, float* part,
                     float* y_out, int N, int Cin, int kpl, int Cout, int H, int W, void* stream);

/* ---- plain pointwise conv  out[n][m][p] = sum_c wt[c][m] * x[n][c][p] + bias[m]
 *      reference: OutConv models/unet_parts.py:67-73; also the data gradient of the
 *      pointwise conv (dY = W^T dZ: pass wt = pointwise.weight in its natural [Cout][K] layout,
 *      Cin := Cout, M := K).
 */
int smaat_pointwise_fwd(const float* x, long x_bs, const float* wt, const float* bias, float* out, long out_bs,
                        float* part, int N, int Cin, int M, int H, int W, void* stream);

/* ---- pointwise weight gradient  dW[m][k] = sum_{n,p} dz[n][m][p] * Y[n][k][p]   (dw_out [M][K])
 *      (autograd of nn.Conv2d(K, M, 1), models/layers.py:45 / models/unet_parts.py:70)
 *      smaat_pointwise_wgrad: Y given (the y_out of smaat_dsconv_fwd, or the OutConv input);
 *                             ws: [smaat_wgrad_num_splits(...)][M][K] floats.
 *      smaat_dsconv_wgrad:    memory-lean variant, Y = depthwise(x) recomputed in the kernel;
 *                             ws: [smaat_dsconv_wgrad_num_splits(...)][Cout][Cin*kpl] floats.
 */
int smaat_wgrad_num_splits(int N, int H, int W, int M, int K);
int smaat_dsconv_wgrad_num_splits(int N, int H, int W, int Cout, int K);
int smaat_dsconv_wgrad(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                       const float* b_dw, const float* dz, long dz_bs, float* ws, float* dw_out, int N, int Cin,
                       int kpl, int Cout, int H, int W, void* stream);
int smaat_pointwise_wgrad(const float* x, long x_bs, const float* dz, long dz_bs, float* ws, float* dw_out, int N,
                          int Cin, int M, int H, int W, void* stream);

/* ---- the same weight gradient on the bf16-split matrix path with Y = depthwise3x3(act(x)) RECOMPUTED by the producer
 *      waves from x (round 4; csrc/dswgrad.hip): reads Cin channels instead of the Cin*kpl channels of the kept
 *      depthwise output, and lets the forward of the layer (smaat_dsconv_fwd_split) run without its y_out side output,
 *      so the 2x-expanded tensor never exists in HBM.  in_scale / in_shift (nullable pair): x is the pre-BatchNorm
 *      tensor of the previous half block, relu(x * scale + shift) is applied on load.  Exact three-term operand split
 *      (f32-class error) or plain bf16 operands, following smaat_split_mode as the other split GEMMs.
 *      smaat_dsconv_wgrad_split_ok: 1 when the kernel takes the shape (kernels_per_layer 2, W % 32 == 0, Cout <= 64);
 *      otherwise smaat_dsconv_wgrad_split returns -2 and the caller streams a kept Y through smaat_pointwise_wgrad.
 *      ws: [smaat_dsconv_wgrad_split_num_splits(...)][Cout][Cin*kpl] floats; fixed-order fp64 reduction, deterministic.
 *      reference: autograd of nn.Conv2d(K, Cout, 1) behind the depthwise conv, models/layers.py:45,47-50.
 */
int smaat_dsconv_wgrad_split_ok(int kpl, int Cout, int H, int W);
int smaat_dsconv_wgrad_split_num_splits(int N, int Cin, int Cout, int H, int W);
int smaat_dsconv_wgrad_split(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                             const float* b_dw, const float* dz, long dz_bs, float* ws, float* dw_out, int N, int Cin,
                             int kpl, int Cout, int H, int W, void* stream);
/* typed form (mixed precision): x_dt / dz_dt = SMAAT_DT_F32 | SMAAT_DT_BF16.  Built: everything f32 (= the entry point
 * above); bf16 dz with bf16 x, or f32 x for the stem -- plain bf16 operands (the stored gradient IS the MFMA operand), f32
 * depthwise arithmetic, f32 accumulation and result.  -2 for any other combination. */
int smaat_dsconv_wgrad_split_t(const void* x, int x_dt, long x_bs, const float* in_scale, const float* in_shift,
                               const float* w_dw, const float* b_dw, const void* dz, int dz_dt, long dz_bs, float* ws,
                               float* dw_out, int N, int Cin, int kpl, int Cout, int H, int W, void* stream);

/* ---- depthwise 3x3 backward (autograd of nn.Conv2d(groups=Cin), models/layers.py:38-44)
 *   dy [N][Cin*kpl][H][W] -> dx [N][Cin][H][W] (nullable), dw_out [Cin*kpl][9], db_out [Cin*kpl] (nullable)
 *   ws: [smaat_dw3x3_bwd_ws_rows(N,Cin,H,W)][Cin*kpl][10] floats
 */
int smaat_dw3x3_bwd_ws_rows(int N, int Cin, int H, int W);
int smaat_dw3x3_bwd(const float* x, long x_bs, const float* dy, long dy_bs, const float* w_dw, float* dx, long dx_bs,
                    float* ws, float* dw_out, float* db_out, int N, int Cin, int kpl, int H, int W, void* stream);
/* same, fused with the backward reduction of the train-mode BatchNorm2d + ReLU in FRONT of this depthwise conv (the
 * first half of a DoubleConvDS, unet_parts_depthwise_separable.py:17-36).  x holds that BatchNorm's INPUT z (the
 * pre-BatchNorm tensor); the activation y = relu(z*in_scale + in_shift) is recomputed on load (it is never written to
 * memory) and the kernel additionally emits
 *   rpart[0][r][ci] = sum dX*[y>0],  rpart[1][r][ci] = sum dX*[y>0]*(z - bn_mean)*bn_invstd   (r < smaat_dw3x3_bwd_ws_rows - 1)
 * which smaat_bn_bwd_finalize consumes in place of the output of smaat_bn_bwd_reduce (one pass over dy and z saved).
 * bn_mean / bn_invstd [Cin]: the batch statistics of that BatchNorm (rows 0 and 1 of smaat_bn_finalize's output): the
 * normalised value is formed exactly as ATen does, for any gamma (zero included).
 * in_scale, in_shift, bn_mean, bn_invstd, dx, rpart are all required.
 * Returns -2 when the shape is not handled by the strip kernel (W % 4 != 0 ...): run the two kernels separately. */
/* 1 when the strip-form depthwise kernels (smaat_dw3x3_fwd with in_scale/in_shift and smaat_dw3x3_bwd_bnred) handle
 * planes of this shape given dense, 16-byte aligned tensors; 0 otherwise (they would return -2). */
int smaat_dw3x3_strip_ok(int kpl, int H, int W);
int smaat_dw3x3_bwd_bnred(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* dy,
                          long dy_bs, const float* w_dw, float* dx, long dx_bs, float* ws, float* dw_out, float* db_out,
                          const float* bn_mean, const float* bn_invstd, float* rpart, int N, int Cin, int kpl, int H,
                          int W, void* stream);

/* ---- BatchNorm2d (train) + ReLU   reference: unet_parts_depthwise_separable.py:25-26,34-35,
 *      layers.py:120,127.  Statistics arrive as partial sums (from smaat_dsconv_fwd etc.).
 *   finalize: part [3][T][C] = per-tile (mean, M2, count) as written by the GEMM kernels (tiles with count 0 are
 *             ignored), count = N*H*W; bias_shift[C] (nullable) is added to the mean (the partials are of z - bias).  Writes mean/invstd/scale/shift [C] and updates
 *             running_mean/var (nullable) with `momentum` and the unbiased variance.
 *   affine_act: y = relu?(z*scale[c] + shift[c])
 *   bwd: g = dy*[y>0]; part [2][slots][C] with slots = smaat_plane_num_slots(N,P);
 *        finalize -> dgamma, dbeta, coef[3][C]; apply -> dz.
 */
/* part is SCRATCH: with T >= 2048 tiles the rows are first merged in place (64 slabs, head rows overwritten) */
int smaat_bn_finalize(float* part, int T, int C, double count, const float* bias_shift, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                      float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: st[4][C] = running_mean, 1/sqrt(running_var+eps), gamma*invstd, beta - mean*gamma*invstd */
int smaat_bn_eval_coefs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                        float eps, int C, float* st, void* stream);
int smaat_affine_act(const float* z, long z_bs, const float* scale, const float* shift, float* y, long y_bs, int N,
                     int C, int P, int relu, void* stream);
int smaat_plane_num_slots(int N, int P);
/* ---- OutConv with ONE output channel fused with the BatchNorm2d + ReLU in front of it: the head of SmaAt_UNet(12, 1)
 *      (reference models/SmaAt_UNet.py:38,56: self.outc = OutConv(64, n_classes) on the output of up4, whose last two
 *      layers are unet_parts_depthwise_separable.py:34-35).  The 64-channel block output and its gradient are never
 *      materialised:
 *   smaat_outconv1_fwd        out[n][p] = b + sum_c w[c] * relu(z[n][c][p] * scale[c] + shift[c])      (out_bs = P)
 *   smaat_bn_bwd_reduce_head  like smaat_bn_bwd_reduce with dy[n][c][p] = w[c] * dlog[n][p] formed on the fly, plus the
 *                             convolution's weight gradient: part [3][slots][C], part[2] = sum dlog * relu(bn(z))
 *   smaat_bn_bwd_apply_head   like smaat_bn_bwd_apply with the same on-the-fly dy                                   */
int smaat_outconv1_fwd(const float* z, long z_bs, const float* scale, const float* shift, const float* w, const float* b,
                       float* out, long out_bs, int N, int C, int P, void* stream);
int smaat_bn_bwd_reduce_head(const float* dlog, long dlog_bs, const float* w, const float* z, long z_bs,
                             const float* scale, const float* shift, const float* mean, const float* invstd, float* part,
                             int N, int C, int P, void* stream);
int smaat_bn_bwd_apply_head(const float* dlog, long dlog_bs, const float* w, const float* z, long z_bs,
                            const float* scale, const float* shift, const float* mean, const float* invstd,
                            const float* coef, float* dz, long dz_bs, int N, int C, int P, void* stream);
int smaat_bn_bwd_reduce(const float* dy, long dy_bs, const float* z, long z_bs, const float* scale,
                        const float* shift, const float* mean, const float* invstd, float* part, int N, int C, int P,
                        int relu, void* stream);
int smaat_bn_bwd_finalize(const float* part, int slots, int C, double count, const float* gamma, const float* invstd,
                          float* dgamma, float* dbeta, float* coef, void* stream);
int smaat_bn_bwd_apply(const float* dy, long dy_bs, const float* z, long z_bs, const float* scale,
                       const float* shift, const float* mean, const float* invstd, const float* coef, float* dz,
                       long dz_bs, int N, int C, int P, int relu, void* stream);

/* ---- small helpers */
/* out[j] = alpha * sum_r part[r][j] (fp64 accumulation, fixed order); `part` is SCRATCH: rows of it
 * may be overwritten by the first level of the reduction. */
int smaat_reduce_rows(const float* part, int rows, long len, float* out, float alpha, void* stream);
/* out[c] = sum_{n,p} x[n][c][p]; ws: [smaat_plane_num_slots(N,P)][C] */
int smaat_channel_sum(const float* x, long x_bs, int N, int C, int P, float* ws, float* out, void* stream);
/* dst[n][0:plane_len] (+)= src[n][0:plane_len] with separate batch strides (torch.cat slices) */
int smaat_copy_planes(const float* src, long s_bs, float* dst, long d_bs, int N, long plane_len, int accum,
                      void* stream);

/* ---- MaxPool2d(2)   reference: unet_parts_depthwise_separable.py:48 */
int smaat_maxpool2_fwd(const float* x, long x_bs, float* y, long y_bs, int N, int C, int H, int W, void* stream);
int smaat_maxpool2_bwd(const float* x, long x_bs, const float* dy, long dy_bs, float* dx, long dx_bs, int N, int C,
                       int H, int W, int accum, void* stream);

/* ---- nn.Upsample(x2, bilinear, align_corners=True) + F.pad into an [Ho][Wo] plane of the
 *      concatenation buffer   reference: unet_parts_depthwise_separable.py:64,76-85 */
int smaat_upsample2x_fwd(const float* x, long x_bs, float* out, long out_bs, int N, int C, int H, int W, int Ho,
                         int Wo, int pad_t, int pad_l, void* stream);
int smaat_upsample2x_bwd(const float* dout, long dout_bs, float* dx, long dx_bs, int N, int C, int H, int W, int Ho,
                         int Wo, int pad_t, int pad_l, void* stream);

/* ---- nn.ConvTranspose2d(C, Cout, kernel_size=2, stride=2) + F.pad into an [Ho][Wo] plane of the concatenation buffer
 *      (UpDS with bilinear=False, reference unet_parts_depthwise_separable.py:72-73,78-85).  The transposed convolution
 *      is a pointwise GEMM with 4*Cout output rows -- row (a*2+b)*Cout + co holds weight[:, co, a, b] -- run with
 *      smaat_pointwise_fwd / smaat_pointwise_fwd_split into t [N][4*Cout][H][W], followed by this 2x2 pixel shuffle:
 *        out[n][co][2i+a+pad_t][2j+b+pad_l] = t[n][(a*2+b)*Cout + co][i][j] + bias[co],  zeros elsewhere.
 *      _bwd is the inverse gather dout -> dt; weight / input / bias gradients then come from smaat_pointwise_wgrad,
 *      the pointwise data-gradient GEMM and smaat_channel_sum. */
int smaat_pixel_shuffle2_fwd(const float* t, long t_bs, const float* bias, float* out, long out_bs, int N, int Cout,
                             int H, int W, int Ho, int Wo, int pad_t, int pad_l, void* stream);
int smaat_pixel_shuffle2_bwd(const float* dout, long dout_bs, float* dt, long dt_bs, int N, int Cout, int H, int W,
                             int Ho, int Wo, int pad_t, int pad_l, void* stream);

/* ---- CBAM   reference: models/layers.py:90-141 */
int smaat_cbam_spconv_blocks(int N, int H, int W);
int smaat_cbam_pix_blocks(int N, int P);
int smaat_cbam_chpool(const float* x, long x_bs, int N, int C, int P, float* avg, float* mx, int* amax, void* stream);
/* the same pooling over y = relu(z * scale[c] + shift[c]) formed on load from the PRE-BatchNorm tensor z of the block in
 * front of the attention (unet_parts_depthwise_separable.py:34-35), with y written out: the block output is
 * materialised by its first consumer instead of by a BatchNorm-apply pass of its own */
int smaat_cbam_chpool_act(const float* z, long z_bs, const float* scale, const float* shift, float* y, long y_bs, int N,
                          int C, int P, float* avg, float* mx, int* amax, void* stream);
int smaat_cbam_mlp(const float* avg, const float* mx, const float* w1, const float* b1, const float* w2,
                   const float* b2, int N, int C, int Cr, float* ha, float* hm, float* s, void* stream);
int smaat_cbam_sppool(const float* x, long x_bs, const float* s, int N, int C, int P, float* maps, void* stream);
int smaat_cbam_spconv(const float* maps, const float* wc, int ks, int N, int H, int W, float* conv, float* part,
                      void* stream);
int smaat_cbam_gate(const float* conv, const float* scale, const float* shift, long total, float* gate, void* stream);
int smaat_cbam_apply(const float* x, long x_bs, const float* s, const float* gate, float* out, long out_bs, int N,
                     int C, int P, void* stream);
/* inference (eval mode, running statistics in the spatial attention's BatchNorm2d(1): a fixed affine map, no grid-wide
 * reduction).  A whole CBAM -- and the MaxPool2d(2) that consumes the same tensor in SmaAt_UNet.forward -- is
 *   smaat_cbam_chpool -> smaat_cbam_eval_pool (shared MLP + sigmoid -> s [N][C], mean/max over channels of x*s -> maps)
 *   -> smaat_cbam_eval_apply (k x k conv on maps + BN(1) eval + sigmoid -> gate; out = x*s*gate; pooled (nullable) =
 *   maxpool2(x) from the same loads).  reference: models/layers.py:105-111,122-129,138-141. */
int smaat_cbam_eval_pool(const float* x, long x_bs, const float* avg, const float* mx, const float* w1, const float* b1,
                         const float* w2, const float* b2, int N, int C, int Cr, int P, float* s_out, float* maps,
                         void* stream);
int smaat_cbam_eval_apply(const float* x, long x_bs, const float* s, const float* maps, const float* wc, int ks,
                          const float* bn_gamma, const float* bn_beta, const float* bn_rm, const float* bn_rv, float eps,
                          int N, int C, int H, int W, float* out, long out_bs, float* pooled, long pooled_bs,
                          void* stream);
int smaat_cbam_bwd_gate(const float* dout, long dout_bs, const float* x, long x_bs, const float* s,
                        const float* gate, const float* conv, const float* mean, const float* invstd, int N, int C,
                        int P, float* dbn, float* part, void* stream);
int smaat_cbam_bwd_spconv(const float* dbn, const float* conv, const float* mean, const float* invstd,
                          const float* coef, const float* maps, const float* wc, int ks, int N, int H, int W,
                          float* dmaps, float* wpart, void* stream);
int smaat_cbam_bwd_main(const float* dout, long dout_bs, const float* x, long x_bs, const float* s,
                        const float* gate, const float* maps, const float* dmaps, int N, int C, int P, float* dx,
                        long dx_bs, float* dspart, void* stream);
int smaat_cbam_bwd_mlp(const float* ds, const float* s, const float* avg, const float* mx, const float* ha,
                       const float* hm, const float* w1, const float* w2, int N, int C, int Cr, float* pg,
                       float* davg, float* dmx, void* stream);
int smaat_cbam_bwd_final(float* dx, long dx_bs, const float* davg, const float* dmx, const int* amax, int N, int C,
                         int P, void* stream);
/* smaat_cbam_bwd_final + the backward of the MaxPool2d(2) that reads the same tensor (encoder levels, reference
 * models/SmaAt_UNet.py:43-50: x -> CBAM(x) for the skip, x -> DownDS -> next level), in one read-modify-write pass:
 *   dx += davg/P + [p == amax] dmx + maxpool2_backward(x, dpool).   Returns -2 when the shape / alignment is not taken
 *   (W % 4 != 0, unaligned planes): the caller then runs smaat_cbam_bwd_final and smaat_maxpool2_bwd(accum = 1). */
int smaat_cbam_bwd_final_pool(float* dx, long dx_bs, const float* davg, const float* dmx, const int* amax, const float* x,
                              long x_bs, const float* dpool, long dp_bs, int N, int C, int H, int W, void* stream);

/* ---- bf16-split matrix path (f32 operands split exactly into three bf16 terms, six bf16 MFMAs per
 *      product, f32 accumulation: f32-class error at 2.7x the f32-MFMA rate).  Same reference call
 *      sites as smaat_dsconv_fwd / smaat_pointwise_fwd; the depthwise stage runs as its own kernel
 *      and the pointwise GEMM reads its output.
 *   smaat_split_mode / smaat_set_split_mode (process-wide switch, initial value from env SMAAT_SPLIT, default 3):
 *        0 = f32-MFMA kernels only; 3 (and 2) = exact three-term split (f32-class error);
 *        1 = operands rounded to bf16, ONE MFMA per product = the bf16 mixed-precision mode of BASELINE
 *        configs[3] (f32 storage, f32 accumulation, ~1e-2 class).  set returns the previous mode, -1 on a bad argument.
 *   smaat_split_enabled: 1 when mode != 0
 *   smaat_split_planes:  w [R][C] f32 -> planes u16, chunk-major [Cp/16][3][R][16], Cp = C rounded up to 16 (zero padded);
 *                        R x C = Cout x K for the forward, K x Cout (the transposed weight) for dX
 *   smaat_dw3x3_fwd:     depthwise 3x3, pad 1 (models/layers.py:38-44,48): x [N][Cin][H][W] -> y [N][Cin*kpl][H][W];
 *                        optional in_scale/in_shift[Cin]: x := relu(x*sc+sh) on load (as smaat_dsconv_fwd);
 *                        returns -2 when the shape/alignment is not handled (W % 4 != 0): use smaat_dsconv_fwd
 *   smaat_pointwise_fwd_split: out[n][m][p] = sum_c A[m][c] x[n][c][p] + bias[m], A given as planes;
 *                        part: nullable [3][smaat_pw_split_num_slots(N,H,W)][M] BatchNorm partials (mean, M2, count per tile) of out - bias
 */
int smaat_split_enabled(void);
int smaat_split_mode(void);
int smaat_set_split_mode(int mode);
int smaat_split_planes(const float* w, int R, int C, void* planes, void* stream);
/* planes of the TRANSPOSE of a matrix stored [C][R] (same output layout as smaat_split_planes(w^T, R, C)): the data
 * gradient of a pointwise conv takes A = pointwise.weight^T without a transposed copy of the weight */
int smaat_split_planes_t(const float* w, int R, int C, void* planes, void* stream);
/* The operand images of several weight matrices in one launch (they all become stale together, at the optimizer step; the
 * reference has no counterpart: cuDNN / MKLDNN re-pack weights inside every convolution call).  desc: DEVICE array
 * [n_desc][8] of int64 { src (f32 matrix), dst (image), R, C, kind, src_t, first block, blocks }, blocks = ceil(R * Cp / 256),
 * kind 0 = smaat_split_planes / _t (Cp = C rounded up to 16; src_t = 1: the transpose of a matrix stored [C][R]),
 * kind 2 = smaat_bf16_planes (Cp = C rounded up to 32); total_blocks = the sum of blocks.  Images are bit-identical to the
 * single-matrix entry points'. */
int smaat_weight_planes_multi(const void* desc, int n_desc, int total_blocks, void* stream);
int smaat_pw_split_num_slots(int N, int H, int W);
int smaat_dw3x3_fwd(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                    const float* b_dw, float* y, long y_bs, int N, int Cin, int kpl, int H, int W, void* stream);
int smaat_pointwise_fwd_split(const float* x, long x_bs, const void* planes, const float* bias, float* out,
                              long out_bs, float* part, int N, int Cin, int M, int H, int W, void* stream);

/* ---- two-term fp16 split (round 5): the same GEMMs with THREE fp16 MFMAs per product instead of six bf16 ones.
 *      reference call sites: nn.Conv2d(K, Cout, 1) forward and its autograd (data and weight gradient),
 *      models/layers.py:45,49 -- the arithmetic the reference leaves to MKLDNN / cuDNN in f32.
 *   An f32 operand tensor T is used as  T * 2^k = h + g + O(2^-22 |T| 2^k),  h = fp16(T 2^k), g = fp16(T 2^k - h)  (round to
 *   nearest), with ONE power-of-two scale per operand tensor: k puts max |T| into [2^14, 2^15).  A product is evaluated as
 *   h_a h_b + h_a g_b + g_a h_b (each an exact 22-bit product, f32 accumulation) and the accumulator is scaled back by the exact
 *   2^-(k_a + k_b): f32-class error (tests/test_gpu_kernels.py: next to the three-term split against fp64, incl. operands with
 *   one channel 1e8 above the rest, gradients in the denormal range, all-zero planes).
 *   The maxima are produced by the kernels that WRITE the operands, as a side output `amax`: a buffer of SMAAT_AMAX_WORDS
 *   (1024) uint32 words in device memory that must hold zeros before the producing launch; afterwards the maximum over the
 *   buffer is the bit pattern of max |v| over the tensor (non-negative floats: unsigned order = float order; the result is
 *   order-independent, i.e. bit-reproducible).  The producers scatter partial maxima over words 0, 32, 64, ... (32 different
 *   128-byte lines: thousands of waves publishing to ONE address serialise at the memory side), the GEMMs read those 32 words:
 *     smaat_dw3x3_fwd_amax      = smaat_dw3x3_fwd + amax of y; -2 when the row-streaming kernel does not take the shape
 *                                 (then: smaat_dw3x3_fwd and the three-term GEMMs)
 *     smaat_bn_bwd_apply_amax   = smaat_bn_bwd_apply (head_w null) or smaat_bn_bwd_apply_head (head_w [C], dy = dlog) + amax of dz
 *   Weights:  smaat_split_planes_h (transposed = 1: the image of w^T, w stored [C][R]) writes the fp16 image
 *     [Cp/16][2][R][16] of w * 2^kexp followed by the trailer { int32 kexp; 12 bytes; scratch } into a buffer of
 *     smaat_split_planes_h_bytes(R, C) bytes (16-byte aligned); its maximum is taken by a first launch over
 *     smaat_split_planes_h_pieces(R, C) 4096-element pieces -- no atomics, no state.  In smaat_weight_planes_multi_h (the
 *     one-launch-per-step refresh) such an image is a descriptor of kind 3 and h_pieces = the sum of the pieces of all kind-3
 *     rows (0: exactly smaat_weight_planes_multi).
 *   GEMMs (arguments as the entry points without the suffix; x_amax / dz_amax = the operands' amax buffers, planes = an fp16 image):
 *     smaat_pointwise_fwd_split_h, smaat_pointwise_fwd_split_k_h, smaat_pointwise_wgrad_h (x = the kept depthwise output).
 *   A NaN / Inf in an operand gives k = 0 for that tensor and propagates through the fp16 terms.
 */
#define SMAAT_AMAX_WORDS 1024
int smaat_dw3x3_fwd_amax(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                         const float* b_dw, float* y, long y_bs, void* amax, int N, int Cin, int kpl, int H, int W,
                         void* stream);
int smaat_bn_bwd_apply_amax(const float* dy, long dy_bs, const float* head_w, const float* z, long z_bs, const float* scale,
                            const float* shift, const float* mean, const float* invstd, const float* coef, float* dz,
                            long dz_bs, void* amax, int N, int C, int P, int relu, void* stream);
int smaat_split_planes_h_bytes(int R, int C);
int smaat_split_planes_h_pieces(int R, int C);
int smaat_split_planes_h(const float* w, int R, int C, void* planes, int transposed, void* stream);
int smaat_weight_planes_multi_h(const void* desc, int n_desc, int total_blocks, int h_pieces, void* stream);
int smaat_pointwise_fwd_split_h(const float* x, long x_bs, const void* x_amax, const void* planes, const float* bias,
                                float* out, long out_bs, float* part, int N, int Cin, int M, int H, int W, void* stream);
int smaat_pointwise_fwd_split_k_h(const float* x, long x_bs, const void* x_amax, const void* planes, const float* bias,
                                  float* out, long out_bs, float* part, float* ws, int S, int N, int Cin, int M, int H, int W,
                                  void* stream);
int smaat_pointwise_wgrad_h(const float* x, long x_bs, const void* x_amax, const float* dz, long dz_bs, const void* dz_amax,
                            float* ws, float* dw_out, int N, int Cin, int M, int H, int W, void* stream);
/* ... and for the layers whose depthwise output never exists in memory (the row-walking pair of the 288^2 layers): the fused
 * forward smaat_dsconv_fwd_rows (f32 storage) as smaat_dsconv_fwd_rows_amax ALSO leaves max |y| of the depthwise output its
 * producer waves form -- the forward GEMM itself keeps the three-term split: its operand's maximum is not known before it
 * runs -- and the recompute weight gradient smaat_dsconv_wgrad_split_h, which re-forms the same y bit for bit in the
 * backward, runs the two-term fp16 split with that maximum and the one of dz.  Arguments otherwise as smaat_dsconv_fwd_rows /
 * smaat_dsconv_wgrad_split (reference: models/layers.py:47-50 forward, :45 weight gradient). */
int smaat_dsconv_fwd_rows_amax(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                               const float* b_dw, const void* planes, const float* b_pw, float* z, long z_bs, float* part,
                               void* y_amax, int N, int Cin, int kpl, int Cout, int H, int W, void* stream);
int smaat_dsconv_wgrad_split_h(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                               const float* b_dw, const void* y_amax, const float* dz, long dz_bs, const void* dz_amax, float* ws,
                               float* dw_out, int N, int Cin, int kpl, int Cout, int H, int W, void* stream);
/* Round 6: the fused forward itself on the two-term fp16 split (three MFMAs per product instead of six).  The scale of y has
 * to be fixed before y exists, so the kernel BOUNDS it:  |y[k]| <= sum_taps |w_dw[k][tap]| * A[k / kpl] + |b_dw[k]|,
 * A[c] = max |act(x[c])| <= max(0, |in_scale[c]| * X[c] + in_shift[c])  (A = X without in_scale), where X bounds |x|:
 *   prev_w == NULL: X = max|x| read from the amax buffer(s) of the kernels that wrote x: x_amax, and x_amax2 (nullable) when
 *                   x is a channel concatenation with two writers (smaat_cbam_apply_amax, smaat_upsample2x_fwd_amax);
 *   prev_w != NULL: x is the output of the previous pointwise convolution, x = prev_w [Cin][prev_K] . u + prev_b (prev_b
 *                   nullable), and x_amax holds max|u| of ITS operand (smaat_dw3x3_fwd_amax / the y_amax of this entry point):
 *                   X[c] = sum_k |prev_w[c][k]| * max|u| + |prev_b[c]|.  x_amax2 must be NULL.
 * The split is a floating-point one (h = rn16(t), g = rn16(t - h)), so a bound that is 2^L too large costs nothing for values
 * above 2^(L - 28) of the bound and an absolute 2^-39 of the bound below.  planes = the fp16 image of w_pw
 * (smaat_split_planes_h, not transposed).  y_amax (nullable) as smaat_dsconv_fwd_rows_amax (the TRUE maximum, for the recompute
 * weight gradient); z_amax (nullable) receives max |z|.  Shapes as smaat_dsconv_fwd_rows (f32 storage); -2 otherwise.
 * (reference: models/layers.py:47-50) */
int smaat_dsconv_fwd_rows_h(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                            const float* b_dw, const void* x_amax, const void* x_amax2, const float* prev_w, const float* prev_b,
                            int prev_K, const void* planes, const float* b_pw, float* z, long z_bs, float* part, void* y_amax,
                            void* z_amax, int N, int Cin, int kpl, int Cout, int H, int W, void* stream);
/* the two writers of a decoder concatenation buffer, leaving max |out| in an amax buffer (zero on entry, as every amax buffer):
 * smaat_cbam_apply / smaat_upsample2x_fwd otherwise (reference: models/layers.py:110,128; unet_parts_depthwise_separable.py:64).
 * smaat_upsample2x_fwd_amax: -2 where the row-walking kernel does not take the shape (Wo % 4, alignment). */
int smaat_cbam_apply_amax(const float* x, long x_bs, const float* s, const float* gate, float* out, long out_bs, void* amax,
                          int N, int C, int P, void* stream);
int smaat_upsample2x_fwd_amax(const float* x, long x_bs, float* out, long out_bs, void* amax, int N, int C, int H, int W, int Ho,
                              int Wo, int pad_t, int pad_l, void* stream);

/* ---- fused BACKWARD of a DepthwiseSeparableConv (round 6; reference: autograd of models/layers.py:47-50, the pointwise data
 *      gradient of :49 feeding the depthwise backward of :48).  The two-kernel form writes dY = W_pw^T dz (the 2x-expanded
 *      tensor) with smaat_pointwise_fwd_split_h and reads it back in smaat_dw3x3_bwd[_bnred]; here MFMA waves form one row of dY
 *      per step into LDS and the depthwise backward consumes it in the same kernel.  Results: dx bit-identical to the two-kernel
 *      form (same MFMA and FMA sequences), dw_dw / db_dw and the BatchNorm sums merged from per-workgroup partials.
 *      x: the depthwise input [N][Cin][H][W]; with in_scale / in_shift (both or neither) it is the PRE-BatchNorm tensor of the
 *      previous half, the activation is applied on load, bn_mean / bn_invstd are that BatchNorm's batch statistics and rpart
 *      [2][rows][Cin] receives its backward sums (sum g, sum g * xhat; g = dx * [act > 0]) as smaat_dw3x3_bwd_bnred leaves them.
 *      dz [N][Cout][H][W] with its amax buffer; planes_t = smaat_split_planes_h(pointwise.weight [Cout][K], src_t = 1) (the
 *      fp16 image of the transpose, exponent in its trailer); ws [rows][K][10] workspace, rows =
 *      smaat_dsconv_bwd_rows_num_rows; dw_out [K][9], db_out [K].  Returns -2 when smaat_dsconv_bwd_rows_ok says no
 *      (kernels_per_layer 2, Cout == 64, Cin 64 or 128, W >= 32, H >= 8, tensors < 2 GiB): keep the two-kernel form. */
int smaat_dsconv_bwd_rows_ok(int kpl, int Cin, int Cout, int H, int W);
int smaat_dsconv_bwd_rows_num_rows(int N, int Cin, int H, int W);
int smaat_dsconv_bwd_rows_h(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* bn_mean,
                            const float* bn_invstd, const float* dz, long dz_bs, const void* dz_amax, const void* planes_t,
                            const float* w_dw, float* dx, long dx_bs, float* ws, float* dw_out, float* db_out, float* rpart, int N,
                            int Cin, int kpl, int Cout, int H, int W, void* stream);

/* ---- fused DepthwiseSeparableConv forward on the bf16-split matrix pipe (training path of the plane-dominated
 *      layers; same reference call site as smaat_dsconv_fwd, models/layers.py:47-50).  The depthwise output never
 *      goes through HBM: producer waves stage the halo tile, run the 3x3 stage and write bf16 split planes straight
 *      into the GEMM's B operand image.  Arguments as smaat_dsconv_fwd, except that the pointwise weight is given as
 *      split planes (smaat_split_planes of pointwise.weight [Cout][K]); part: nullable
 *      [3][smaat_dsconv_split_num_slots(N,H,W)][Cout]; y_out: nullable side output (depthwise result, for the streamed
 *      weight gradient).  Returns -2 when the shape is not handled (kernels_per_layer != 2, W % 16 != 0, H < 4 ...):
 *      run smaat_dw3x3_fwd + smaat_pointwise_fwd_split instead.  smaat_dsconv_split_num_slots returns 0 for such shapes.
 */
/* smaat_pointwise_fwd_split_act with the contraction cut into K slices when the problem has too few tiles to fill the
 * chip (inference at batch 1: the deep layers are a few dozen serial chunk chains): the slices run as virtual images of
 * the same kernel into ws [N][S][M][P] and a second kernel adds them in a fixed order, the bias and the ReLU.
 * ws: smaat_pointwise_splitk_ws_floats(...) floats (0 = no split for this shape; ws may then be null).  Needs dense x
 * (x_bs == Cin*H*W); otherwise, and without ws, it is smaat_pointwise_fwd_split_act. */
int smaat_pointwise_splitk_ws_floats(int N, int Cin, int M, int H, int W);
int smaat_pointwise_fwd_split_act_k(const float* x, long x_bs, const void* planes, const float* bias, float* out,
                                    long out_bs, float* ws, int N, int Cin, int M, int H, int W, int relu_out,
                                    void* stream);
/* training form: the slice count is the caller's (smaat_pointwise_splitk_slices with a budget of workgroup items: more
 * than one only when the un-sliced launch leaves the chip under-filled -- the 18 x 18 layers at batch 32 are 384 serial
 * chains of 64 chunks), ws = N * S * M * H * W floats, and the slice reduction also emits the BatchNorm partials
 * part [3][smaat_pw_split_num_slots(N,H,W)][M] of the summed result (nullable), as smaat_pointwise_fwd_split does.
 * reference: the pointwise conv of models/layers.py:49 in front of the train-mode BatchNorm of
 * unet_parts_depthwise_separable.py:25,34.  -2: shape / alignment not handled (x must be dense, Cin / 16 divisible by S). */
int smaat_pointwise_splitk_slices(int N, int Cin, int M, int H, int W, int budget_items);
int smaat_pointwise_fwd_split_k(const float* x, long x_bs, const void* planes, const float* bias, float* out, long out_bs,
                                float* part, float* ws, int S, int N, int Cin, int M, int H, int W, int relu_out,
                                void* stream);

int smaat_dsconv_split_num_slots(int N, int H, int W);
int smaat_dsconv_fwd_split(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                           const float* b_dw, const void* planes, const float* b_pw, float* z, long z_bs, float* part,
                           float* y_out, int N, int Cin, int kpl, int Cout, int H, int W, void* stream);

/* ---- the same fused forward as a ROW-WALKING kernel (round 4; csrc/dsrows.hip): a workgroup walks down a band of rows of
 *      one 32-column strip, the depthwise window lives in the producer threads' registers (no halo re-reads, no staging),
 *      the pointwise weight's MFMA fragments live in the consumer waves' registers for the whole walk.  Same reference
 *      call site (models/layers.py:47-50 + the BatchNorm partials for unet_parts_depthwise_separable.py:25,34).
 *      x_dt / z_dt: SMAAT_DT_F32 | SMAAT_DT_BF16.  f32 storage (x f32, z f32): planes = smaat_split_planes of
 *      pointwise.weight [Cout][K] (exact three-term split, or one term in bf16-operand mode).  bf16 storage (z bf16; x bf16,
 *      or f32 for the stem): planes = smaat_bf16_planes of the weight, one MFMA per product, f32 accumulation.
 *      part: nullable [3][smaat_dsconv_rows_num_slots(N,H,W)][Cout] (mean, M2, count of z - b_pw per band x strip x image).
 *      There is no depthwise side output: the weight gradient recomputes it (smaat_dsconv_wgrad_split).
 *      smaat_dsconv_rows_ok: 1 when the kernel takes the shape (kernels_per_layer 2, W % 32 == 0, Cout <= 64, Cin % 8 == 0,
 *      Cin <= 128); otherwise smaat_dsconv_fwd_rows returns -2 (use smaat_dsconv_fwd_split / the unfused pair).
 */
int smaat_dsconv_rows_ok(int kpl, int Cin, int Cout, int H, int W);
int smaat_dsconv_rows_num_slots(int N, int H, int W);
int smaat_dsconv_fwd_rows(const void* x, int x_dt, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                          const float* b_dw, const void* planes, const float* b_pw, void* z, int z_dt, long z_bs, float* part,
                          int N, int Cin, int kpl, int Cout, int H, int W, void* stream);

/* ---- inference forms (reference call stack D, calc_metrics_test_set.py:119): the same three
 *      forward GEMM entry points without statistics / side outputs and with an optional fused ReLU epilogue
 *      (relu_out != 0: out = max(acc + bias, 0)).  With BatchNorm folded into the pointwise weights by the host
 *      (w' = w * gamma / sqrt(running_var + eps), b' = (b - running_mean) * gamma / sqrt(running_var + eps) + beta:
 *      eval-mode BatchNorm2d, unet_parts_depthwise_separable.py:25,34) one launch is a whole
 *      DepthwiseSeparableConv -> BatchNorm2d -> ReLU half block.
 */
int smaat_dsconv_fwd_act(const float* x, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                         const float* b_dw, const float* wt_pw, const float* b_pw, float* z, long z_bs, int N, int Cin,
                         int kpl, int Cout, int H, int W, int relu_out, void* stream);
int smaat_dsconv_fwd_split_act(const float* x, long x_bs, const float* in_scale, const float* in_shift,
                               const float* w_dw, const float* b_dw, const void* planes, const float* b_pw, float* z,
                               long z_bs, int N, int Cin, int kpl, int Cout, int H, int W, int relu_out, void* stream);
int smaat_pointwise_fwd_split_act(const float* x, long x_bs, const void* planes, const float* bias, float* out,
                                  long out_bs, int N, int Cin, int M, int H, int W, int relu_out, void* stream);

/* ================= mixed precision: bf16 activation storage (BASELINE.json configs[3]) =================================
 * The reference has no AMP switch (train_precip_lightning.py:71-72 leaves set_float32_matmul_precision commented out);
 * what a maintainer would turn on is Lightning's precision="bf16-mixed" = torch.autocast around the nn.Modules of
 * models/SmaAt_UNet.py:41-57.  This section is that mode built natively: every activation tensor of the network and its
 * gradient is stored as bf16 (2 bytes per element in HBM -- in bf16 every layer is HBM-bound), all arithmetic
 * is f32 in registers, the pointwise GEMMs run ONE v_mfma_f32_32x32x16_bf16 per product with f32 accumulation, weights
 * (f32 masters), weight gradients, BatchNorm statistics / partial sums and the small attention maps stay f32.
 *
 * Conventions: a tensor argument of variable element type is a `const void*` / `void*` followed (or described) by a dtype
 * code: SMAAT_DT_F32 = 0, SMAAT_DT_BF16 = 1; strides stay in ELEMENTS.  "-2" = this combination of shape / alignment /
 * dtypes is not built (the f32 entry points above cover everything in f32).  Entry points named *_t are the typed forms of
 * the f32 entry points of the same name: same arithmetic, same argument meaning, same reference call sites.
 */
#define SMAAT_DT_F32 0
#define SMAAT_DT_BF16 1

/* pointwise.weight (f32 master, [R][C] row-major; transposed != 0: the image of the TRANSPOSE of a matrix stored [C][R]) ->
 * bf16 image [ceil32(C)/16][R][16] (contraction chunk major, zero padded): the A operand of smaat_pointwise_fwd_bf16.
 * reference: the weight of nn.Conv2d(K, Cout, 1), models/layers.py:45 (forward: R x C = Cout x K; data gradient: K x Cout). */
int smaat_bf16_planes(const float* w, int R, int C, void* planes, int transposed, void* stream);
/* out[n][m][p] = sum_c A[m][c] x[n][c][p] + bias[m]  (x bf16 [N][Cin][P], out bf16 or f32 per out_dt; bias, part f32):
 * the pointwise conv of DepthwiseSeparableConv.forward (models/layers.py:49) on the depthwise output, and its data
 * gradient dY = W^T dZ (A = transposed image, bias = null).  part: nullable [3][smaat_pw_split_num_slots(N,H,W)][M]
 * BatchNorm partials (mean, M2, count per 128-pixel tile) of the f32 accumulators; relu_out: max(., 0) in the epilogue.
 * LDS-DMA operand staging + ds_read_b64_tr_b16 transposed fragment reads (csrc/bf16gemm.hip).  -2: odd H*W. */
int smaat_pointwise_fwd_bf16(const void* x, long x_bs, const void* planes, const float* bias, void* out, long out_bs,
                             int out_dt, float* part, int N, int Cin, int M, int H, int W, int relu_out, void* stream);
/* dW[m][k] = sum_{n,p} dz[n][m][p] * y[n][k][p] with bf16 dz, y and f32 accumulation / output (autograd of the same
 * nn.Conv2d); ws: [smaat_wgrad_num_splits(N,H,W,M,Cin)][M][Cin] floats, dw_out [M][Cin]. */
int smaat_pointwise_wgrad_bf16(const void* y, long y_bs, const void* dz, long dz_bs, float* ws, float* dw_out, int N,
                               int Cin, int M, int H, int W, void* stream);
/* ---- depthwise convolution of ANY geometry DepthwiseSeparableConv can be constructed with (models/layers.py:35-45:
 *      nn.Conv2d(Cin, Cin * kpl, kernel_size, padding=padding, groups=Cin); stride 1, dilation 1, any kpl >= 1), f32.
 *      The network's own 3 x 3 / padding 1 / kpl in {1, 2, 4} layers use smaat_dw3x3_*; this is the general form behind the
 *      same module (direct gather kernels, deterministic reductions; coverage, not a roofline kernel).
 *   x [N][Cin][H][W]; w_dw [Cin*kpl][KH][KW]; b_dw [Cin*kpl] (nullable); y, dy [N][Cin*kpl][Ho][Wo],
 *   Ho = H + 2 pad_h - KH + 1, Wo = W + 2 pad_w - KW + 1  (-1 when that is < 1)
 *   bwd: dx (nullable) [N][Cin][H][W]; dw_out (nullable) [Cin*kpl][KH][KW]; db_out (nullable, needs dw_out) [Cin*kpl] */
int smaat_dwconv_fwd_any(const float* x, long x_bs, const float* w_dw, const float* b_dw, float* y, long y_bs, int N, int Cin,
                         int kpl, int H, int W, int KH, int KW, int pad_h, int pad_w, void* stream);
int smaat_dwconv_bwd_any(const float* x, long x_bs, const float* dy, long dy_bs, const float* w_dw, float* dx, long dx_bs,
                         float* dw_out, float* db_out, int N, int Cin, int kpl, int H, int W, int KH, int KW, int pad_h,
                         int pad_w, void* stream);
/* typed smaat_dw3x3_fwd (models/layers.py:38-44,48): (x_dt, y_dt) in {(f32,f32), (f32,bf16), (bf16,bf16)} */
int smaat_dw3x3_fwd_t(const void* x, int x_dt, long x_bs, const float* in_scale, const float* in_shift, const float* w_dw,
                      const float* b_dw, void* y, int y_dt, long y_bs, int N, int Cin, int kpl, int H, int W,
                      void* stream);
/* typed smaat_dw3x3_bwd / smaat_dw3x3_bwd_bnred (rpart != null selects the fused BatchNorm reduction, which then needs
 * in_scale / in_shift / bn_mean / bn_invstd / dx): (x_dt, dy_dt, dx_dt) in {(f32,f32,f32), (bf16,bf16,bf16), (f32,bf16,f32)} */
int smaat_dw3x3_bwd_t(const void* x, int x_dt, long x_bs, const float* in_scale, const float* in_shift, const void* dy,
                      int dy_dt, long dy_bs, const float* w_dw, void* dx, int dx_dt, long dx_bs, float* ws, float* dw_out,
                      float* db_out, const float* bn_mean, const float* bn_invstd, float* rpart, int N, int Cin, int kpl,
                      int H, int W, void* stream);
/* typed smaat_affine_act (nn.BatchNorm2d + nn.ReLU apply, unet_parts_depthwise_separable.py:25-26,34-35) */
int smaat_affine_act_t(const void* z, int z_dt, long z_bs, const float* scale, const float* shift, void* y, int y_dt,
                       long y_bs, int N, int C, int P, int relu, void* stream);
/* typed smaat_bn_bwd_reduce / _apply; head_w != null: the *_head forms (dy is dlog [N][1][P], f32), see above.
 * (dy_dt, z_dt[, dz_dt]) in {all f32, all bf16, (f32, bf16[, bf16])} */
int smaat_bn_bwd_reduce_t(const void* dy, int dy_dt, long dy_bs, const void* z, int z_dt, long z_bs, const float* scale,
                          const float* shift, const float* mean, const float* invstd, float* part, int N, int C, int P,
                          int relu, const float* head_w, void* stream);
int smaat_bn_bwd_apply_t(const void* dy, int dy_dt, long dy_bs, const void* z, int z_dt, long z_bs, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const float* coef, void* dz,
                         int dz_dt, long dz_bs, int N, int C, int P, int relu, const float* head_w, void* stream);
/* typed smaat_outconv1_fwd (logits stay f32) and smaat_channel_sum */
int smaat_outconv1_fwd_t(const void* z, int z_dt, long z_bs, const float* scale, const float* shift, const float* w,
                         const float* b, float* out, long out_bs, int N, int C, int P, void* stream);
int smaat_channel_sum_t(const void* x, int x_dt, long x_bs, int N, int C, int P, float* ws, float* out, void* stream);
/* typed MaxPool2d(2) and bilinear Upsample + pad (unet_parts_depthwise_separable.py:48,64,76-85): one dtype `dt` for
 * every tensor argument.  The upsample forms exist for the row-walking kernels only (-2 otherwise in bf16). */
int smaat_maxpool2_fwd_t(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, int dt, void* stream);
int smaat_maxpool2_bwd_t(const void* x, long x_bs, const void* dy, long dy_bs, void* dx, long dx_bs, int N, int C, int H,
                         int W, int accum, int dt, void* stream);
int smaat_upsample2x_fwd_t(const void* x, long x_bs, void* out, long out_bs, int N, int C, int H, int W, int Ho, int Wo,
                           int pad_t, int pad_l, int dt, void* stream);
int smaat_upsample2x_bwd_t(const void* dout, long dout_bs, void* dx, long dx_bs, int N, int C, int H, int W, int Ho,
                           int Wo, int pad_t, int pad_l, int dt, void* stream);
/* typed CBAM passes over the activation-sized tensors (models/layers.py:105-141); x / y / out / dout / dx / dpool all have
 * element type `dt`, the per-(n, c) vectors, the 2-channel maps and the gate stay f32.  smaat_cbam_chpool_t with
 * scale != null is smaat_cbam_chpool_act (pools taken over the values as stored). */
int smaat_cbam_chpool_t(const void* x, long x_bs, const float* scale, const float* shift, void* y, long y_bs, int N,
                        int C, int P, float* avg, float* mx, int* amax, int dt, void* stream);
/* smaat_cbam_chpool_t + the MaxPool2d(2) of the same tensor (SmaAt_UNet.py:43-50: an encoder level feeds CBAM and the next
 * DownDS) in one pass: pooled [N][C][H/2][W/2] (floor mode) written from the registers the channel pools are taken from.
 * y, pooled, mx, amax bit-identical to smaat_cbam_chpool[_act] followed by smaat_maxpool2_fwd; avg up to the f32 summation
 * order.  Either dtype.  -2: W % 4 != 0 / alignment (run the two entry points instead). */
int smaat_cbam_chpool_pool_t(const void* x, long x_bs, const float* scale, const float* shift, void* y, long y_bs,
                             void* pooled, long pooled_bs, int N, int C, int H, int W, float* avg, float* mx, int* amax, int dt,
                             void* stream);
int smaat_cbam_sppool_t(const void* x, long x_bs, const float* s, int N, int C, int P, float* maps, int dt,
                        void* stream);
int smaat_cbam_apply_t(const void* x, long x_bs, const float* s, const float* gate, void* out, long out_bs, int N, int C,
                       int P, int dt, void* stream);
int smaat_cbam_bwd_gate_t(const void* dout, long dout_bs, const void* x, long x_bs, const float* s, const float* gate,
                          const float* conv, const float* mean, const float* invstd, int N, int C, int P, float* dbn,
                          float* part, int dt, void* stream);
int smaat_cbam_bwd_main_t(const void* dout, long dout_bs, const void* x, long x_bs, const float* s, const float* gate,
                          const float* maps, const float* dmaps, int N, int C, int P, void* dx, long dx_bs,
                          float* dspart, int dt, void* stream);
int smaat_cbam_bwd_final_t(void* dx, long dx_bs, const float* davg, const float* dmx, const int* amax, int N, int C,
                           int P, int dt, void* stream);
int smaat_cbam_bwd_final_pool_t(void* dx, long dx_bs, const float* davg, const float* dmx, const int* amax, const void* x,
                                long x_bs, const void* dpool, long dp_bs, int N, int C, int H, int W, int dt,
                                void* stream);

/* ---- the attention backward of a level in three passes (round 5) ---------------------------------------------------
 * Same gradients as smaat_cbam_bwd_gate_t + smaat_cbam_bwd_main_t + smaat_cbam_bwd_final[_pool]_t (the autograd of
 * models/layers.py:105-111, 122-129, 138-141 and of the MaxPool2d of unet_parts_depthwise_separable.py:48 that reads the same
 * tensor), with dx written once, and with the channels of a level split over the waves of a workgroup where one wave per 256
 * pixels would walk hundreds of channels in a dependent chain (the 72 x 72 ... 18 x 18 levels):
 *   smaat_cbam_sppool_idx_t   (forward) smaat_cbam_sppool_t + amaxc[n][p] = the FIRST channel that attains max_c x * s
 *                             (models/layers.py:123-125; what torch.max's backward routes the gradient to)
 *   smaat_cbam_bwd_gate_ds_t  the gate pass + dspart[blk][n][c] = sum_p (dout * gate) * x              (reads dout, x)
 *   smaat_cbam_bwd_ds2_t      dspart[blk][n][c] = sum_p (dmaps0 / C + [c == amaxc] dmaps1) * x         (reads x)
 *     -> ds = the sum of both partial sets (smaat_reduce_rows over 2 * blocks rows), then smaat_cbam_bwd_mlp
 *   smaat_cbam_bwd_apply_t    dx = (dout * gate + dmaps0 / C + [c == amaxc] dmaps1) * s + davg / P + [p == amax] dmx
 *                                  (+ [p first maximum of its 2 x 2 window] dpool when dpool != NULL)
 * blk as in smaat_cbam_pix_blocks (blocks of 256 pixels per image).  smaat_cbam_bwd3_ok: 1 when the shapes, strides and
 * alignments allow this route (H * W % 4 == 0, 4-element aligned planes; with dpool: H even and W % 4 == 0); otherwise the
 * entries return -2 and the caller keeps smaat_cbam_sppool_t and the gate / main / final sequence. */
int smaat_cbam_bwd3_ok(const void* x, long x_bs, const void* dout, long dout_bs, const void* dpool, long dp_bs, int N, int C,
                       int H, int W, int dt);
int smaat_cbam_sppool_idx_t(const void* x, long x_bs, const float* s, int N, int C, int P, float* maps, int* amaxc, int dt,
                            void* stream);
int smaat_cbam_bwd_gate_ds_t(const void* dout, long dout_bs, const void* x, long x_bs, const float* s, const float* gate,
                             const float* conv, const float* mean, const float* invstd, int N, int C, int P, float* dbn,
                             float* part, float* dspart, int dt, void* stream);
int smaat_cbam_bwd_ds2_t(const void* x, long x_bs, const float* dmaps, const int* amaxc, int N, int C, int P, float* dspart,
                         int dt, void* stream);
int smaat_cbam_bwd_apply_t(const void* dout, long dout_bs, const void* x, long x_bs, const float* s, const float* gate,
                           const float* dmaps, const int* amaxc, const float* davg, const float* dmx, const int* amax,
                           const void* dpool, long dp_bs, int N, int C, int H, int W, void* dx, long dx_bs, int dt,
                           void* stream);

/* ---- on-device PrecipitationMetrics.update ---------------------------------------------
 * replaces metric/precipitation_metrics.py:37-95 (called every train/val/test step, models/regression_lightning.py:
 * 75,86,94): NaN check, sum (p-t)^2 / batch, sum (p*f - t*f)^2 / batch, and the 4-bin confusion counts of
 * (x * f * 12 > threshold), accumulated INTO the persistent state without a host synchronisation:
 *   state_f64[2] = { total_loss, total_loss_denorm }
 *   state_i64[7] = { batches skipped because they contained a NaN, tn, fp, fn, tp, total_samples, total_pixels }
 * preds / target: n contiguous floats; batch = target.size(0); denormalize = 0 leaves the factor out (:77-78).
 * ws: smaat_precip_metrics_ws_bytes(n) bytes, 8-byte aligned.  A batch with a NaN changes only state_i64[0].
 */
int smaat_precip_metrics_ws_bytes(long n);
int smaat_precip_metrics_update(const float* preds, const float* target, long n, int batch, float factor,
                                float threshold, int denormalize, void* ws, double* state_f64, long long* state_i64,
                                void* stream);

/* ---- Adam in one launch (round 6) -------------------------------------------------------------------------------------------
 * reference: optim.Adam(self.parameters(), lr) -- models/regression_lightning.py:48, train_SmaAtUNet.py:182 (default betas, eps,
 * no weight decay, no amsgrad); arithmetic of torch.optim.Adam's multi-tensor path, operation for operation in f32:
 *   m = m + w1 (g - m);  v = v beta2 + w2 g g;  p = p + step_size * (m / (sqrt(v) / bc2_sqrt + eps))
 * rows: DEVICE table [n][4] of 64-bit words {p, m, v, numel} (f32 tensors, dense); grads: HOST array of n device pointers (this
 * step's gradients: autograd hands over fresh tensors each step, so they travel in the kernel arguments); blk2t [total_blocks],
 * blk0 [n]: device int32 -- the row of block b and the first block of row t, smaat_adam_block_elems() elements per block;
 * n <= smaat_adam_max_tensors() per call.  Scalars in double, formed by the caller exactly as torch/optim/adam.py forms them
 * (Python floats): w1 = 1 - beta1, w2 = 1 - beta2, bc2_sqrt = (1 - beta2 ** t) ** 0.5, step_size = (lr / (1 - beta1 ** t)) * -1,
 * rounded to f32 once.  variant: bits 1 / 2 / 4 = first moment / second moment / update contracted into one fma (which of them
 * torch's own kernels contract is a property of its build; smaat_unet_amd/optim.py uses the combination its test found). */
int smaat_adam_max_tensors(void);
int smaat_adam_block_elems(void);
int smaat_adam_step(const void* rows, const void* const* grads, const int* blk2t, const int* blk0, int n, int total_blocks, double w1,
                    double beta2, double w2, double bc2_sqrt, double eps, double step_size, int variant, void* stream);

/* ---- PROTOTYPE (round 6): pointwise GEMM on pre-split fp16 planes of both operands -- measurement only ---------------------
 * out[n][m][p] = 2^-ksum * sum_c (Ah Xg + Ag Xh + Ah Xh) + bias[m]: the arithmetic of smaat_pointwise_fwd_split_h (reference
 * models/layers.py:45,49) with NO operand split inside the kernel: x_planes [N][2][Cin][H*W] fp16 (h, g of x * 2^kx; x_bs / xp_bs =
 * elements between images / planes), a_planes [2][Cin/16][M][16] fp16 (h, g of w * 2^ka; ap_bs elements between the planes),
 * ksum = ka + kx.  LDS-DMA + transposed LDS reads, no VALU in the loop (csrc/h2gemm.hip).  cfg 0 / 1 / 2: stage depth x resident
 * workgroups.  Cin % 32 == 0, M > 64, (H*W) % 8 == 0; -2 otherwise.  Nothing in the training step produces activation planes, so
 * nothing selects this entry: scripts/probes/h2_gemm_probe.py measures what such a GEMM reaches (NOTEBOOK 4.9). */
int smaat_pointwise_fwd_h2_proto(const void* x_planes, long x_bs, long xp_bs, const void* a_planes, long ap_bs, const float* bias,
                                 float* out, long out_bs, float* part, int N, int Cin, int M, int H, int W, int ksum, int cfg,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMAAT_HIP_H */
