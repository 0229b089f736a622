/*
 * gan_amd.h -- C ABI of libgan_amd.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * StyleGAN2 training hot path of gau-nernst/gan.
 *
 * The reference has no FFI; its operator boundary is the Python module API.
 * Each entry point below is what the Python autograd.Functions of `gan_amd` bind through
 * ctypes, and cites the reference code (file:line under /root/reference) whose arithmetic it
 * replaces.
 *
 * Conventions
 *  - All tensors are device pointers.  Activations are channels-last: [N][H][W][C], C fastest.
 *  - `dtype` selects the activation element type: GAN_F32 (float) or GAN_BF16 (bfloat16).
 *    Accumulation is always fp32.  Per-channel / per-sample side vectors (bias, scales, FIR
 *    taps, reductions, weight gradients) are always fp32.
 *  - The caller allocates every output.  The library never allocates, frees or synchronises;
 *    kernels are enqueued on `stream` (a hipStream_t) and the call returns immediately.
 *  - Return value: 0 on success, negative on error; gan_amd_last_error() then describes it
 *    (thread-local).  Nothing is launched when an argument check fails.
 *  - One process per GPU; calls are re-entrant; there is no global mutable state (one exception, documented at
 *    gan_conv2d_wgrad_static: the built-in scratch buffer of the entry point that allocates nothing).
 */
#ifndef GAN_AMD_H
#define GAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAN_F32 0
#define GAN_BF16 1

#define GAN_OK 0
#define GAN_ERR_ARG (-1)
#define GAN_ERR_LAUNCH (-2)

typedef void* gan_stream_t; /* hipStream_t */

/* ---- probes ---------------------------------------------------------------------------- */
int gan_amd_version(void);            /* ABI version, bumped on any signature change          */
const char* gan_amd_arch(void);       /* "gfx950"                                             */
const char* gan_amd_last_error(void); /* message of the last failing call on this thread      */

/* ---- upfirdn2d -------------------------------------------------------------------------
 * replaces upfirdn2d()            modelling/nvidia_ops.py:18-34
 *          UpFIRDn2d fwd/bwd      modelling/nvidia_ops.py:40-54 (backward = same entry, swapped args)
 *          nn.Upsample(2) / F.upsample / nn.AvgPool2d(2) as special cases
 *                                 modelling/stylegan2.py:45,99  modelling/progressive_gan.py:48,50
 * y[n,oy,ox,c] = sum_{i,j} taps[i*kx+j] * P[n, oy*down+i, ox*down+j, c], P = zero-insert(x, up)
 * padded by (px1,px2,py1,py2) (negative pad crops); taps are NOT flipped (correlation).
 * x: [N][H][W][C], y: [N][Ho][Wo][C], taps: fp32 [ky*kx] on the device.
 * Ho/Wo are passed by the caller (reference formula (H*up+py1+py2-ky)/down+1; the backward
 * passes the forward's input size, see gan_amd/modelling/nvidia_ops.py). */
int gan_upfirdn2d(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx, int up,
                  int down, int px1, int py1, int Ho, int Wo, int dtype, gan_stream_t stream);

/* ---- fused channel-affine + noise + residual + leaky-ReLU ---------------------------------
 * replaces the unfused sequence  conv bias (stylegan2.py:35) -> Noise (stylegan.py:24-27) ->
 *          LeakyReLU(0.2) (stylegan2.py:49,56; progressive_gan.py:44,46,73), the demodulation
 *          scale of ModulatedConv2d (stylegan2.py:33) and the residual `shortcut + layers*scale`
 *          (progressive_gan.py:54).  No gain, no clamp.
 * y = lrelu_slope( x*a[n*a_stride_n + c] + b[c] + noise[n,hw]*ns[c] + r[n,hw,c] )
 * Any of a, b, noise/ns, r may be NULL (term dropped).  slope = 1 gives a linear op.
 * x, r, y: [N][HW][C]; noise: [N][HW] (same dtype as x); a, b, ns: fp32. */
int gan_affine_act_fwd(const void* x, const float* a, int64_t a_stride_n, const float* b, const void* noise,
                       const float* ns, const void* r, void* y, int N, int64_t HW, int C, float slope, int dtype,
                       gan_stream_t stream);

/* Backward building block of the op above (and, applied twice, its double backward):
 * out = g * m(ymask) * a[n*a_stride_n + c],  m(v) = v > 0 ? 1 : slope   (mask from the OUTPUT
 * of the activation: sign(y) == sign(pre-activation) for slope > 0).  ymask, a may be NULL. */
int gan_mask_scale(const void* g, const void* ymask, const float* a, int64_t a_stride_n, void* out, int N,
                   int64_t HW, int C, float slope, int dtype, gan_stream_t stream);

/* Channel reductions for bias / noise-scale / style / demodulation / D-block `scale` gradients:
 * out[n*out_stride_n + c] = sum_hw A[n,hw,c] * m(ymask[n,hw,c]) * Bterm,
 *   Bterm = B[n,hw,c] (b_mode 1), B[n,hw] broadcast over c (b_mode 2), or 1 (b_mode 0).
 * out_stride_n = C gives per-sample sums [N][C]; out_stride_n = 0 gives batch sums [C].
 * Two deterministic stages (wavefront shuffle + LDS reduction per workgroup, partial rows stored to `workspace`, a fold
 * kernel that sums them in a fixed order): no atomics, `out` need not be zeroed, every element of it is written.
 * accumulate = 1: out += the sums instead (a parameter's gradient buffer that other passes have already written to:
 * autograd's accumulation step, torch/csrc/autograd/functions/accumulate_grad.h, without its launch per tensor).
 * workspace: at least gan_chan_dot_workspace(N, HW, C, dtype) floats, 16-byte aligned, caller-owned (the library never
 * allocates). */
int64_t gan_chan_dot_workspace(int N, int64_t HW, int C, int dtype);
int gan_chan_dot(const void* A, const void* B, int b_mode, const void* ymask, float slope, float* out,
                 int64_t out_stride_n, int N, int64_t HW, int C, float* workspace, int64_t workspace_floats, int accumulate,
                 int dtype, gan_stream_t stream);

/* Fused passes of the modulated convolution's backward (autograd of ModulatedConv2d.forward,
 * modelling/stylegan2.py:29-36; they replace three gan_chan_dot calls + one gan_mask_scale, and one of each):
 *   gan_modconv_bwd_pre: gz = gy * m(y) * dinv[n,c]                     (y, noise, dinv may be NULL)
 *        sums[n][0][c] = sum_hw gy*m(y), sums[n][1][c] = sum_hw gy*m(y)*noise[n,hw], sums[n][2][c] = sum_hw gy*y
 *        sums: fp32 [N][3][C], zeroed here (rows 1 / 2 stay zero without noise / y).
 *   gan_scale_dot:       out = A * a[n,c],  dot[n][c] = sum_hw A*B       (dot: fp32 [N][C], zeroed here).
 *   gan_modconv_bwd_post: the parameter gradients from `sums` in one launch (any output may be NULL):
 *        gb[c] = sum_n sums[n][0][c],  gns[c] = sum_n sums[n][1][c],
 *        gdinv[n][c] = (sums[n][2][c] - sums[n][0][c]*bias[c] - sums[n][1][c]*ns[c]) / dinv[n][c]
 *        (bias / ns NULL = 0; all fp32). */
int gan_modconv_bwd_pre(const void* gy, const void* y, const void* noise, const float* dinv, float slope, void* gz,
                        float* sums, int sums_zeroed, int N, int64_t HW, int C, int dtype, gan_stream_t stream);
int gan_scale_dot(const void* A, const void* B, const float* a, void* out, float* dot, int dot_zeroed, int N, int64_t HW, int C,
                  int dtype, gan_stream_t stream);
/* (sums_zeroed / dot_zeroed != 0: the caller hands over a buffer that already holds zeros -- one memset for all the
 * layers of a backward pass instead of one per call; 0: the call clears it on `stream` first.) */
int gan_modconv_bwd_post(const float* sums, const float* bias, const float* ns, const float* dinv, float* gb, float* gns,
                         float* gdinv, int N, int C, gan_stream_t stream);

/* Demodulation factor of ModulatedConv2d: replaces F.normalize(weight * s, dim=(2,3,4), eps=1e-8)
 * (modelling/stylegan2.py:33) by dinv[n][co] = 1 / max(sqrt(sum_ci s[n,ci]^2 * wsq[co,ci]), 1e-8),
 * wsq[co][ci] = sum_taps W^2 (written here: a [Cout][Cin] scratch the caller owns and hands back to the
 * backward).  All fp32; s: [N][Cin], w: [Cout][Cin][taps].  Backward: gs [N][Cin] and / or gw [Cout][Cin][taps]
 * (either may be NULL) from g = d loss / d dinv; gw_accumulate != 0 adds into gw (a parameter's gradient buffer).
 * gan_demod_fwd with w == NULL takes wsq as given (gan_pack_weights_multi writes it with the packed images). */
int gan_demod_fwd(const float* s, const float* w, float* wsq, float* dinv, int N, int Cin, int Cout, int taps,
                  gan_stream_t stream);
int gan_demod_bwd(const float* g, const float* dinv, const float* s, const float* w, const float* wsq, float* gs,
                  float* gw, int gw_accumulate, int N, int Cin, int Cout, int taps, gan_stream_t stream);
/* The same three products for EVERY demodulating layer of a generator in one launch each (the reference evaluates
 * modelling/stylegan2.py:30-33 layer by layer; all layers see the same w, so their styles are the column blocks of one
 * (N, lds) matrix s_all and their factors / factor gradients the column blocks of one (N, ldd) matrix):
 *   mode 0: out_all = dinv_all (N, ldd);  mode 1: out_all = gs_all (N, lds) from g_all, dinv_all (N, ldd) and s_all;
 *   mode 2: every row's gw (+)= from the same.
 * `table`: device memory, `count` rows of 10 int64: [0] wsq, [1] w, [2] gw, [3] Cin, [4] Cout, [5] taps, [6] the layer's
 * first column in s_all, [7] its first column in dinv_all / g_all, [8] its first workgroup in THIS mode's launch
 * (ceil(Nn/16) * ceil(M/16) workgroups per row; (M, Nn) = (N, Cout), (N, Cin), (Cout, Cin) for modes 0, 1, 2),
 * [9] accumulate into gw.  `blocks`: the launch's workgroup count.  Columns [6], [7] multiples of 4. */
int gan_demod_multi(int mode, const void* table, int count, int64_t blocks, const float* s_all, int lds, const float* g_all,
                    const float* dinv_all, int ldd, float* out_all, int N, gan_stream_t stream);

/* ---- implicit-GEMM convolution (MFMA) ----------------------------------------------------
 * replaces ModulatedConv2d.forward / batched_conv2d   modelling/stylegan2.py:18-36
 *          nn.Conv2d of the discriminator             modelling/progressive_gan.py:45-50,61,72,74
 *          and, with transposed/flipped weights, their input gradients (autograd of the above).
 * y[n,oy,ox,co] = act( out_scale[n,co] * sum_{dy,dx,ci} X[n,oy+dy-py,ox+dx-px,ci] * in_scale[n,ci]
 *                       * Wt[co][(dy*kw+dx)*Cin+ci] + bias[co] + noise[n,oy,ox]*noise_scale[co] )
 * X = x, or nearest-neighbour 2x upsampling of x when up2 != 0 (stylegan2.py:45 fused into the gather).
 * Stride 1.  x: [N][H][W][Cin] with Cin % 8 == 0; wt: [Cout][Kpad] in the activation dtype,
 * K index = tap*Cin + ci, zero-padded to Kpad (multiple of 32, >= kh*kw*Cin);
 * y: [N][Ho][Wo][Cout]; in_scale [N][Cin], out_scale [N][Cout], bias/noise_scale [Cout] fp32 or NULL;
 * noise [N][Ho*Wo] in the activation dtype or NULL; slope = 1 for no activation.
 *
 * Stages of the discriminator block fused around the GEMM (progressive_gan.py:45-52: LeakyReLU -> Conv2d
 * [-> AvgPool2d], and the adjoints of those in the backward pass), so they cost no pass over HBM:
 *   in_slope != 1 : X = leaky_relu(x, in_slope) applied as the input is staged (before in_scale);
 *   pool = 1      : the convolution result is 2x2 average pooled (Ho, Wo even) BEFORE out_scale / bias /
 *                   noise / act; y, noise and aux are then at (Ho/2, Wo/2).  Ho, Wo stay the un-pooled size;
 *   aux != NULL   : y *= m(aux), m(v) = v > 0 ? 1 : aux_slope, aux shaped like y (the leaky-ReLU derivative
 *                   mask that follows an input gradient);
 *   gain          : scalar factor on the accumulated sum (0.25 for the adjoint of the pooling, whose
 *                   nearest-neighbour upsampling is the up2 gather);
 *   resid != NULL : y += resid (shaped like y) after everything else: the gradient that reaches the same
 *                   tensor along a second path (the D block's shortcut) without an add pass.
 * Pass 1, 0, NULL, 1, 1, NULL for none. */
int gan_conv2d_fwd(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                   const float* bias, const void* noise, const float* noise_scale, float slope, int N, int H, int W,
                   int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad,
                   float in_slope, int pool, const void* aux, float aux_slope, float gain, const void* resid, int dtype,
                   gan_stream_t stream);

/* gan_conv2d_fwd with an optional caller workspace.  A wide layer on a small grid (512 channels on 4x4 / 8x8 maps at a
 * per-GPU batch of 32..64: fewer 128 x 64 tiles than half the chip's CUs) is a serial chain of K steps per tile -- 72 for a
 * 512-channel 3x3 layer, ~42 us whatever the batch.  With gan_conv2d_fwd_workspace(...) floats of workspace (0: the shape does
 * not split) the kernel splits K over up to 8 workgroups per tile, writes fp32 partial tiles there and a second kernel folds
 * them in a fixed order and applies every output stage.  Same result up to fp32 summation order; workspace NULL or too
 * small = gan_conv2d_fwd.  Contents of the workspace are undefined on entry and exit. */
int64_t gan_conv2d_fwd_workspace(int N, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int Kpad, int has_in_scale, int dtype);
int gan_conv2d_fwd_ws(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale, const float* bias,
                      const void* noise, const float* noise_scale, float slope, int N, int H, int W, int Cin, int up2, int Ho, int Wo,
                      int Cout, int kh, int kw, int py, int px, int Kpad, float in_slope, int pool, const void* aux, float aux_slope,
                      float gain, const void* resid, float* workspace, int64_t workspace_floats, int dtype, gan_stream_t stream);

/* Weight gradient (autograd of the convolutions above; torch.nn.grad.conv2d_weight in the reference's autograd):
 * dw[co][ci][ky][kx] = gain * sum_{n,oy,ox} gy[n,oy,ox,co] * X[n,oy+ky-py,ox+kx-px,ci] * in_scale[n,ci]
 * dw: fp32 (Cout_out, Cin_out, kh, kw) contiguous -- the parameter's own layout; every element is written
 * (no need to zero it).  x / gy carry Cin / Cout channels, multiples of 16 (callers pad with zeros);
 * Cout_out <= Cout and Cin_out <= Cin are the un-padded counts.
 * The reduction over pixels is split over workgroups: partial tiles go to `workspace` (at least
 * gan_conv2d_wgrad_workspace(...) floats, 16-byte aligned, contents undefined on entry and exit) with plain
 * stores -- or, on the register-staged fallback kernel, with fp32 atomics into one zeroed slice of it -- and a
 * second kernel folds them into dw.  The library never allocates: the caller owns the workspace.
 * Fused stages (the weight gradient of the LeakyReLU -> Conv2d -> AvgPool2d chain, progressive_gan.py:45-50):
 *   x_slope != 1 : X = leaky_relu(x, x_slope) as it is staged (before in_scale);
 *   gy_up2 = 1   : gy is [N][Ho/2][Wo/2][Cout] and read through a nearest-neighbour 2x gather (the adjoint
 *                  of the pooling, together with gain = 0.25);
 *   gain         : scalar factor on dw.  Pass 1, 0, 1 for none.
 * accumulate = 1: dw += the gradient instead of dw = (as gan_chan_dot's flag: the fold kernel reads dw back). */
int64_t gan_conv2d_wgrad_workspace(int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw,
                                   int py, int px, int dtype);
int gan_conv2d_wgrad(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, const float* in_scale, int N, int H, int W, int Cin,
                     int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, float* workspace,
                     int64_t workspace_floats, float x_slope, int gy_up2, float gain, int accumulate, int dtype,
                     gan_stream_t stream);
/* The same with the layer's BIAS gradient from the same call: dbias[co] (+)= sum over (n, oy, ox) of gy (Cout floats;
 * needs Cout_out == Cout; dbias_accumulate as accumulate).  Replaces the `.sum((0, 2, 3))` autograd runs for the bias of
 * nn.Conv2d (progressive_gan.py:45-50) -- a second full read of gy.  On the thin kernel the sums are one more MFMA per
 * tile row (gy^T x ones) and ride in the split-K slices; other paths run the channel-sum kernel after the weight
 * gradient, in the same workspace: gan_conv2d_wgrad_bias_workspace(...) floats (>= gan_conv2d_wgrad_workspace). */
int64_t gan_conv2d_wgrad_bias_workspace(int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px,
                                        int gy_up2, int dtype);
int gan_conv2d_wgrad_bias(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, float* dbias, int dbias_accumulate,
                          const float* in_scale, int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py,
                          int px, float* workspace, int64_t workspace_floats, float x_slope, int gy_up2, float gain, int accumulate,
                          int dtype, gan_stream_t stream);

/* The same weight gradient for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO counterpart in the reference: the reference allocates freely through
 * PyTorch's caching allocator.  Arguments and result are those of gan_conv2d_wgrad without `workspace` /
 * `workspace_floats`: the split-K slices go to a buffer built into the library, a file-scope device array of
 * GAN_WGRAD_STATIC_MAX_FLOATS floats (48 MiB; every GPU the library runs on holds its own copy until the process
 * ends).  The call allocates and frees nothing (no hipMalloc / hipFree, no allocating library routine); shapes whose
 * gan_conv2d_wgrad_workspace(...) exceeds the buffer are rejected with GAN_ERR_ARG and nothing is launched.
 * Every call on one GPU shares that GPU's buffer: the kernels overwrite what they read back (the one slice that is
 * accumulated into is cleared on `stream` first), so consecutive calls on one stream are independent of each other,
 * but calls that OVERLAP on one GPU (two streams, two threads) race -- serialise them.
 * gan_amd_init() resolves the buffer's device address for the CURRENT device (hipGetSymbolAddress, once per device);
 * gan_conv2d_wgrad_static does the same on its first call on a device.  Call gan_amd_init() beforehand where that
 * first call would fall inside a stream capture. */
#define GAN_WGRAD_STATIC_MAX_FLOATS 12582912
int gan_amd_init(void);
int64_t gan_conv2d_wgrad_static_max_floats(void);
int gan_conv2d_wgrad_static(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, const float* in_scale, int N, int H,
                            int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, float x_slope,
                            int gy_up2, float gain, int dtype, gan_stream_t stream);

/* Packs a (Cout, Cin, kh, kw) weight tensor (w_dtype) into the GEMM-B image the convolutions above take:
 * out[row][Kpad] in out_dtype, k = tap*cols_p + c, zero padded.  transposed = 0: rows = Cout, c = ci;
 * transposed = 1: the flipped, channel-swapped weights of the input-gradient correlation (rows = Cin, c = co,
 * tap reversed).  Host-side glue of the reference (the weight is used as is by F.conv2d, stylegan2.py:36). */
int gan_pack_weight(const void* w, int w_dtype, void* out, int out_dtype, int Cout, int Cin, int kh, int kw, int cols_p,
                    int Kpad, int transposed, gan_stream_t stream);
/* Both images of one weight in one launch: `out` as gan_pack_weight(..., transposed = 0) with (cols_p, Kpad), `out_t`
 * as transposed = 1 with (cols_p_t, Kpad_t).  Every layer that trains uses both (forward + input gradient). */
int gan_pack_weight_pair(const void* w, int w_dtype, void* out, int cols_p, int Kpad, void* out_t, int cols_p_t, int Kpad_t,
                         int out_dtype, int Cout, int Cin, int kh, int kw, gan_stream_t stream);
/* The images of a LIST of weights in one launch (the reference re-reads its weights in every F.conv2d; here they are
 * re-packed once per optimizer step, and ~60 launches of 5-8 us per iteration become two).  `table`: device memory,
 * `count` rows of 12 int64: [0] w, [1] out or 0, [2] out_t or 0, [3] wsq (fp32 [Cout][Cin] sums over the taps of W^2,
 * the first factor of gan_demod_fwd) or 0, [4] first workgroup of the row, [5] Cout, [6] Cin, [7] kh*kw, [8] cols_p,
 * [9] Kpad, [10] cols_p_t, [11] Kpad_t (cols_p % 8 == 0, Kpad % 32 == 0, images 16-byte aligned; the padding of an
 * absent image 0).  A workgroup owns a 64 x 64 (co x ci) tile of one weight with all its taps, staged in LDS: a row owns
 * ceil(max(Cout, cols_p_t) / 64) * ceil(max(Cin, cols_p) / 64) workgroups and `blocks` is their total; max_taps = the
 * largest kh*kw of the table, 4096 * max_taps * sizeof(out element) <= 150 KB.  All rows share w_dtype and out_dtype. */
int gan_pack_weights_multi(const int64_t* table, int count, int64_t blocks, int max_taps, int w_dtype, int out_dtype, gan_stream_t stream);

/* Image-like tensors (C < 8: the RGB inputs / outputs of the from-RGB and to-RGB convolutions,
 * progressive_gan.py:61, stylegan2.py:124) -> channels-last, zero padded to Cp channels (multiple of 8), in
 * out_dtype: out[n][h][w][c] = x[n*sn + c*sc + h*sh + w*sw] (element strides; NCHW or channels-last, fp32 or
 * bf16).  Host-side glue of the reference (F.conv2d takes 3 channels as they are); one pass instead of
 * cast + layout copy + zero fill + strided copy. */
int gan_pad_channels(const void* x, int x_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, void* out, int out_dtype,
                     int N, int C, int H, int W, int Cp, gan_stream_t stream);

/* RGB skip of StyleGan2Generator.forward (stylegan2.py:99): out = nearest_up2(lo) + hi in one pass.
 * lo: [N][H][W][C], hi / out: [N][2H][2W][C], same dtype. */
int gan_up2_add(const void* lo, const void* hi, void* out, int N, int H, int W, int C, int dtype, gan_stream_t stream);

/* ---- MinibatchStdDev ---------------------------------------------------------------------
 * replaces MinibatchStdDev.forward   modelling/progressive_gan.py:20-36
 * stat[j] = mean_{c,h,w} sqrt( var_k x[j + k*M] + 1e-6 ),  M = N/G, population variance over the
 * G group members.  x: [N][HW][C]; stat: fp32 [M] (zeroed here). */
int gan_mbstd_fwd(const void* x, float* stat, int N, int G, int64_t HW, int C, int dtype, gan_stream_t stream);
/* autograd of the statistic: gx[j + k*M][hw][c] = gstat[j] * (x - mean_k x) / (G * std * HW*C), mean and std over the
 * G members recomputed per element.  gx: [N][HW][C] in x's dtype (every element written). */
int gan_mbstd_bwd(const void* x, const float* gstat, void* gx, int N, int G, int64_t HW, int C, int dtype,
                  gan_stream_t stream);

/* autograd of gan_mbstd_bwd (second order of the statistic: R1 differentiates the discriminator twice,
 * losses.py:100-103).  u: cotangent of gx, shaped like x.  Outputs (either may be NULL):
 *   gx2 [N][HW][C] (x's dtype) = d<u, gx>/dx,   ggstat fp32 [M] = d<u, gx>/dgstat (every element written). */
int gan_mbstd_bwd2(const void* x, const float* gstat, const void* u, void* gx2, float* ggstat, int N, int G, int64_t HW, int C,
                   int dtype, gan_stream_t stream);

/* The module's output (progressive_gan.py:31-36): out [N][HW][C+1] = x with the statistic of the sample's group column
 * (stat[n % M], fp32 [M]) appended as channel C -- the reference's repeat / view / expand / cat in one pass -- and the
 * adjoint of that linear map: gx [N][HW][C] = g[..., :C] (NULL: skipped), gstat[j] = sum_k sum_hw g[j + k*M][hw][C]. */
int gan_mbstd_concat(const void* x, const float* stat, void* out, int N, int G, int64_t HW, int C, int dtype, gan_stream_t stream);
int gan_mbstd_concat_adj(const void* g, void* gx, float* gstat, int N, int G, int64_t HW, int C, int dtype, gan_stream_t stream);

/* ---- the image boundary: 1x1 convolution from / to <= 4 channels ---------------------------------------------------
 * replaces nn.Conv2d(img_channels, base_dim, 1) of the discriminator (modelling/progressive_gan.py:61) and its two
 * adjoints (autograd of it; R1 differentiates with respect to the image, losses.py:100-103), reading / writing the
 * 3-channel tensor as it is (no 16-channel padded copy, no cast / layout pass in front):
 *   fwd  : y[n][p][co] = lrelu_slope( sum_c x[n*sn + c*sc + h*sh + w*sw] * w[co][c] + bias[co] ),  y: [N][HW][Cout] (y_dtype)
 *   dgrad: gx[n][p][c] = sum_co gy[n][p][co] * w[co][c],  gx: [N][HW][Cin] in gy's dtype
 *   wgrad: dw[co][c]   = sum_{n,p} gy[n][p][co] * x[n, c, p]  (fp32; two deterministic stages through `workspace`,
 *          at least gan_rgb_conv1x1_wgrad_workspace(Cin, Cout) floats, caller-owned); dbias != NULL (Cin <= 3): also
 *          dbias[co] = sum_{n,p} gy[n][p][co], the layer's bias gradient, from the same pass -- a constant 1 in the free
 *          channel slot
 * x: element strides, fp32 or bf16; w: fp32 [Cout][Cin] (the parameter itself); Cin <= 4, Cout % 8 == 0, Cout <= 64. */
int gan_rgb_conv1x1_fwd(const void* x, int x_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* w, const float* bias, void* y,
                        int y_dtype, int N, int Cin, int H, int W, int Cout, float slope, gan_stream_t stream);
int gan_rgb_conv1x1_dgrad(const void* gy, const float* w, void* gx, int64_t NP, int Cin, int Cout, int dtype, gan_stream_t stream);
int64_t gan_rgb_conv1x1_wgrad_workspace(int Cin, int Cout);
int gan_rgb_conv1x1_wgrad(const void* x, int x_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const void* gy, int gy_dtype, float* dw,
                          float* dbias, float* workspace, int64_t workspace_floats, int N, int Cin, int H, int W, int Cout, gan_stream_t stream);

/* DiffAugment (diff_augment.py:13-60; train_celeba.py:63 puts it in front of the discriminator): brightness,
 * saturation, contrast, translation and cut-out of an image batch as one affine map, applied in one pass after a
 * per-sample sum (the reference: eight launches and as many round trips).
 *   in   : [N, C, H, W] with element strides (sn, sc, sh, sw), C <= 4, fp32 or bf16 (`dtype`)
 *   out  : [N][H][W][C] (channels-last, contiguous), same dtype
 *   usk  : [N][3] device floats in [0, 1): the draws of rand_brightness / rand_saturation / rand_contrast (:14,19,24)
 *   shift: 4 device ints (tx, ty, cy, cx): translate(x, tx, ty) (:35-36) and RandomCutout's (offset_y, offset_x) (:48-49);
 *          the cut-out is H/2 x H/2 as in the reference (:50)
 *   mode : 0 = the augmentation y = A x + c;  1 = its linear part y = A x (the gradient of mode 2);
 *          2 = its adjoint y = A' x (the gradient of modes 0 and 1) -- closed under differentiation, as R1 needs.
 * Every draw is read on the device, so the call can be captured in a HIP graph (the reference's `.item()` cannot).
 * workspace: gan_diffaug_workspace(N) floats, caller-owned, contents undefined on entry and exit. */
/* The generator's mapping network (stylegan2.py:70-72,93): LayerNorm(dim) then L x [Linear(dim, dim), LeakyReLU(slope)],
 * dim = 128, L <= 8, fp32 -- one launch forward, two backward, instead of ~17 and ~50 library calls.
 *   fwd : acts [L+1][N][dim]: acts[0] = LayerNorm(z), acts[l+1] = lrelu(acts[l] W_l' + B_l) (the network's output is
 *         acts[L]); stats [N][2] = (mean, rstd) of LayerNorm.  W, B: host arrays of L device pointers (nn.Linear's
 *         (out, in) weights and biases where they are; B or its entries may be null); ln_w / ln_b may be null.
 *   bwd : from gout [N][dim] (the gradient at acts[L]) to `grads`, one flat fp32 buffer laid out
 *         dW[L][dim][dim], dB[L][dim], dln_w[dim], dln_b[dim]; no gradient for z (a random draw in every caller).
 *         workspace: gan_mapping_bwd_workspace(N, L) floats, caller-owned (per-workgroup partials, folded in a fixed
 *         order: deterministic).
 * gan_split_cols: [N][T] -> the concatenation of contiguous [N][C_l] blocks, block l at N * offsets[l] (offsets: nl + 1
 * device ints, offsets[nl] = T); reverse = 1 is the inverse.  One launch instead of a copy per ModulatedConv2d style. */
int gan_mapping_fwd(const float* z, const float* ln_w, const float* ln_b, const float* const* W, const float* const* B, int L, float* acts,
                    float* stats, int N, int dim, float slope, float eps, gan_stream_t stream);
int64_t gan_mapping_bwd_workspace(int N, int L);
int gan_mapping_bwd(const float* gout, const float* acts, const float* stats, const float* z, const float* ln_w, const float* const* W, int L,
                    float* grads, float* workspace, int64_t workspace_floats, int N, int dim, float slope, gan_stream_t stream);
int gan_split_cols(const float* in, float* out, const int* offsets, int nl, int N, int T, int reverse, gan_stream_t stream);

int64_t gan_diffaug_workspace(int N);
int gan_diffaug(const void* in, int64_t sn, int64_t sc, int64_t sh, int64_t sw, void* out, const float* usk, const int* shift, int mode, int N,
                int C, int H, int W, float* workspace, int64_t workspace_floats, int dtype, gan_stream_t stream);

/* ---- stride-2 convolutions as stride-1 convolutions ------------------------------------------------------------
 * replaces the stride of nn.Conv2d(k, 2, p) / nn.ConvTranspose2d(k, 2, p) with even k in DCGAN (modelling/dcgan.py:21-29,
 * 50-60 through conv_norm_act, modelling/common.py:46-70): Conv_s2(x; W) = ValidConv_{k/2}(S2D_p(x); W') with
 *   S2D_p (to_depth = 1): out [N][(H+2p)/2][(W+2p)/2][4C], channel (dy*2+dx)*C + c <- x[n][2by+dy-p][2bx+dx-p][c], 0 outside
 *   its adjoint (to_depth = 0): out [N][H][W][C] from in [N][(H+2p)/2][(W+2p)/2][4C]  (ConvTranspose2d's last stage).
 * H, W, C always describe the un-folded tensor. */
int gan_space_to_depth(const void* x, void* out, int N, int H, int W, int C, int pad, int to_depth, int dtype, gan_stream_t stream);

/* ---- AdaIN / LayerNorm2d (the StyleGAN-v1 and Progressive-GAN generators) ------------------------------------
 * replaces AdaIN.forward              modelling/common.py:13-16   F.instance_norm(x) * (style_w(w) + 1) + style_b(w)
 *          LayerNorm2d.forward        modelling/common.py:20-24   nn.LayerNorm over the channels of each pixel
 *          and the LeakyReLU(0.2) that follows them (stylegan.py:33-34, progressive_gan.py:89-90,124-125).
 * gan_chan_stats : mean[n][c], rstd[n][c] = 1 / sqrt(var + eps) of x [N][HW][C] over the pixels (biased variance).
 * gan_chan_affine: y = lrelu_slope(x * a[n][c] + b[n][c])  (AdaIN with a = rstd * gamma, b = beta - mean * a).
 * gan_chan_lin2  : out = g * m(ymask) * a1[n][c] + x * a2[n][c] + a3[n][c]  (its input gradient in one pass; ymask may be
 *                  NULL).  a, b, a1..a3, mean, rstd: fp32 [N][C].
 * gan_layernorm2d_fwd: y[p][c] = lrelu_slope((x[p][c] - mean_p) * rstd_p * w[c] + b[c]); stats: fp32 [NP][2] = (mean_p,
 *                  rstd_p), kept for the backward pass; NP = N*H*W pixels.
 * gan_layernorm2d_bwd: gx (NULL: skipped) and gwb fp32 [2][C] = (gradient of w, gradient of b), zeroed here; g is multiplied
 *                  by the leaky-ReLU mask of ymask (the forward's output; NULL = no activation). */
int gan_chan_stats(const void* x, float* mean, float* rstd, int N, int64_t HW, int C, float eps, int dtype, gan_stream_t stream);
int gan_chan_affine(const void* x, const float* a, const float* b, void* y, int N, int64_t HW, int C, float slope, int dtype,
                    gan_stream_t stream);
int gan_chan_lin2(const void* g, const void* ymask, const void* x, const float* a1, const float* a2, const float* a3, void* out, int N,
                  int64_t HW, int C, float slope, int dtype, gan_stream_t stream);
int gan_layernorm2d_fwd(const void* x, const float* w, const float* b, void* y, float* stats, int64_t NP, int C, float eps, float slope,
                        int dtype, gan_stream_t stream);
int gan_layernorm2d_bwd(const void* g, const void* x, const void* ymask, const float* w, const float* stats, void* gx, float* gwb,
                        int64_t NP, int C, float slope, int dtype, gan_stream_t stream);

/* ---- input gradient of a modulated convolution with its scale + dot pass on the accumulators ----
 * replaces the backward of `x * style` in front of the shared-weight convolution              modelling/stylegan2.py:30-35
 * With gxs = conv(gz, W^T flipped) * gain [2x2-pooled: the adjoint of the layer's nearest-2x upsampling], autograd needs
 *   gx[n][p][c] = gxs[n][p][c] * s[n][c]          (the input gradient)
 *   gs[n][c]    = sum_p gxs[n][p][c] * xin[n][p][c]   (the style gradient; xin = the layer's saved input)
 * i.e. a second pass that reads gxs and xin and writes gx (gan_scale_dot).  Here both ride on the convolution's
 * accumulators: y = gx, dot_out += gs (fp32 atomics, one per channel, wave and image: dot_out must hold zeros -- or a
 * running sum -- on entry).  x = gz (N, H, W, Cin), wt = the packed transposed / flipped weights (gan_pack_weight),
 * out_scale = s (N, Cout) fp32, dot_x = xin shaped like y.  Thin layers only (the layers whose activations are large):
 * gan_conv2d_fwd_dot_supported(...) == 1, else call gan_conv2d_fwd and gan_scale_dot. */
int gan_conv2d_fwd_dot_supported(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad, int pool,
                                 int dtype);
int gan_conv2d_fwd_dot(const void* x, const void* wt, void* y, const float* out_scale, const void* dot_x, float* dot_out, int N, int H, int W,
                       int Cin, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad, int pool, float gain, int dtype,
                       gan_stream_t stream);

/* ---- input gradient of a D block's two consumers of x in one kernel ----------------------------
 * replaces the backward of `self.layers[1](self.layers[0](x))` and `self.shortcut(x)` w.r.t. x    modelling/progressive_gan.py:45-52
 *   y = LeakyReLU'(aux) * (conv(x, wt) * gain) + sc_gain * NearestUp2( conv1x1(sc_x, sc_w) )
 * x = the gradient at conv1's output (N, H, W, Cin), wt = conv1's packed transposed / flipped weights, aux = the block input
 * (its sign decides the mask), sc_x = the gradient at the shortcut's output (N, H/2, W/2, sc_c = 2 Cout), sc_w = the shortcut's
 * packed transposed 1x1 weights [Cout][sc_c], sc_gain = 1/4 (the adjoint of AvgPool2d(2)).  The shortcut term is sc_c / 32 more
 * K steps of the same tile into a second set of accumulators -- without this entry it is a kernel of its own that writes a
 * full-resolution tensor which gan_conv2d_fwd(..., resid) reads back.  Thin layers with Cout 16 or 32 (64 with Cin 64):
 * gan_conv2d_fwd_short_supported(...) == 1. */
int gan_conv2d_fwd_short_supported(int N, int H, int W, int Cin, int Cout, int kh, int kw, int py, int px, int Kpad, int sc_c, int dtype);
int gan_conv2d_fwd_short(const void* x, const void* wt, void* y, const void* aux, float aux_slope, float gain, const void* sc_x,
                         const void* sc_w, int sc_c, float sc_gain, int N, int H, int W, int Cin, int Cout, int kh, int kw, int py, int px,
                         int Kpad, int dtype, gan_stream_t stream);

/* ---- convolution parameters scaled per output channel ------------------------------------
 * replaces the activation-sized multiply of `self.shortcut(x) + self.layers(x) * self.scale`   modelling/progressive_gan.py:54
 * conv(x, W) * scale + b * scale = conv(x, W * scale) + b * scale: the block's last convolution runs on
 *   ws[r][:] = w[r][:] * s[r],  bs[r] = b[r] * s[r]          (rows = Cout, cols = Cin*kh*kw; b / bs may both be NULL)
 * and adds the shortcut as its `resid` (gan_conv2d_fwd), so the merge costs no pass over the activations.
 * gan_scale_rows_bwd is the adjoint of that map (any of gws / gbs / gw / gb / gs may be NULL):
 *   gw[r][:] (+)= gws[r][:] * s[r],  gb[r] (+)= gbs[r] * s[r],  gs[r] (+)= sum_c gws[r][c] * w[r][c] + gbs[r] * b[r]
 * acc_* = 1 adds into the buffer (a parameter's gradient slice), 0 overwrites.  All fp32; one workgroup per row, fixed
 * summation order. */
int gan_scale_rows(const float* w, const float* b, const float* s, float* ws, float* bs, int rows, int64_t cols, gan_stream_t stream);
int gan_scale_rows_bwd(const float* gws, const float* gbs, const float* w, const float* b, const float* s, float* gw, int acc_w,
                       float* gb, int acc_b, float* gs, int acc_s, int rows, int64_t cols, gan_stream_t stream);

/* ---- EMA ---------------------------------------------------------------------------------
 * replaces the per-parameter lerp_ loop of EMA.step   utils.py:95-96
 * ema[i] += (p[i] - ema[i]) * weight over one flat fp32 arena. */
int gan_ema_lerp(float* ema, const float* p, int64_t n, float weight, gan_stream_t stream);

/* ---- Adam -------------------------------------------------------------------------------
 * replaces torch.optim.Adam / AdamW as built at train_celeba.py:75-77 (one launch per parameter tensor and scalar
 * there) by one pass over flat arenas: p, g (gradient), m (exp_avg), v (exp_avg_sq): fp32 [n], 16-byte aligned.
 *   t = step[0] (+ 1 first when advance != 0: the count lives on the device so the call can be graph-captured)
 *   g' = g + weight_decay * p (decoupled = 0)      or      p *= 1 - lr * weight_decay (decoupled = 1, AdamW)
 *   m += (g' - m) * (1 - beta1);  v = v * beta2 + g'^2 * (1 - beta2)
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * advance = 0 lets one logical step be issued as several calls over sub-ranges of the arena. */
int gan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float* step, int advance, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int decoupled, gan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAN_AMD_H */
