// Shared helpers for the gfx950 kernels of libgan_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gan_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing ---------------------------------------------------------------------
void gan_set_error(const char* fmt, ...);

#define GAN_CHECK_ARG(cond, ...)  \
    do {                          \
        if (!(cond)) {            \
            gan_set_error(__VA_ARGS__); \
            return GAN_ERR_ARG;   \
        }                         \
    } while (0)

#define GAN_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            gan_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return GAN_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

// ---- 16-byte vector access: V elements of T as floats ------------------------------------
template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
    static constexpr int V = 4;
};
template <>
struct VecOf<bf16> {
    static constexpr int V = 8;
};

template <typename T, int V>
__device__ __forceinline__ void load_f(const T* p, float (&out)[V]);

template <>
__device__ __forceinline__ void load_f<float, 4>(const float* p, float (&out)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load_f<float, 1>(const float* p, float (&out)[1]) {
    out[0] = *p;
}
template <>
__device__ __forceinline__ void load_f<bf16, 8>(const bf16* p, float (&out)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <>
__device__ __forceinline__ void load_f<bf16, 1>(const bf16* p, float (&out)[1]) {
    out[0] = (float)*p;
}

template <typename T, int V>
__device__ __forceinline__ void store_f(T* p, const float (&in)[V]);

template <>
__device__ __forceinline__ void store_f<float, 4>(float* p, const float (&in)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <>
__device__ __forceinline__ void store_f<float, 1>(float* p, const float (&in)[1]) {
    *p = in[0];
}
template <>
__device__ __forceinline__ void store_f<bf16, 8>(bf16* p, const float (&in)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)in[i];
    *reinterpret_cast<bf16x8*>(p) = v;
}
template <>
__device__ __forceinline__ void store_f<bf16, 1>(bf16* p, const float (&in)[1]) {
    *p = (bf16)in[0];
}

// 16-byte global load that is ALWAYS issued (invalid lanes read the buffer's first bytes and get zeros by
// a select afterwards).  A load inside `if (ok)` makes hipcc branch around it and wait for each one in
// turn, which serialises a batch of staging loads into that many memory latencies (measured: the wide
// weight-gradient producers waited 68 % of their cycles that way).
__device__ __forceinline__ uint4 ld16_or_zero(const void* base, int64_t elem_off_bytes, bool ok) {
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + (ok ? elem_off_bytes : 0));
    return ok ? v : make_uint4(0u, 0u, 0u, 0u);
}

// value of lane ^ 1 / lane ^ 2 within each quad, as a DPP move (no trip through the LDS crossbar like __shfl_xor)
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float lrelu_mask(float y, float slope) { return y > 0.f ? 1.f : slope; }

// leaky-ReLU of a 16-byte chunk on its way into LDS (the activation that precedes a convolution)
__device__ __forceinline__ uint4 lrelu_chunk(uint4 raw, float slope, bf16) {
    bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)lrelu_f((float)v[j], slope);
    return *reinterpret_cast<uint4*>(&v);
}
__device__ __forceinline__ uint4 lrelu_chunk(uint4 raw, float slope, float) {
    float4 v = *reinterpret_cast<float4*>(&raw);
    v.x = lrelu_f(v.x, slope); v.y = lrelu_f(v.y, slope); v.z = lrelu_f(v.z, slope); v.w = lrelu_f(v.w, slope);
    return *reinterpret_cast<uint4*>(&v);
}

// The fused stages around a convolution beyond modulation / bias / noise / activation (include/gan_amd.h):
// leaky-ReLU of the input as it is staged, 2x2 average pooling of the result, a final multiply by the
// leaky-ReLU derivative mask of `aux` (a tensor shaped like the output), a scalar gain, a final residual add.
struct ConvFuse {
    float in_slope;   // 1 = no input activation
    int pool;         // 1 = 2x2 average pooling; bias / noise / aux / y are at the pooled resolution
    const void* aux;  // nullptr = no mask
    float aux_slope;
    float gain;
    const void* resid;  // nullptr, or a tensor shaped like y added to the result last (a second gradient path)
    float* ws;          // optional caller workspace (gan_conv2d_fwd_ws): lets small-grid wide layers split K
    int64_t ws_floats;
    // gan_conv2d_fwd_dot (thin kernel only): dot_out[n][co] += sum over pixels of (result before out_scale) * dot_x[n][pixel][co]
    const void* dot_x;
    float* dot_out;
    // gan_conv2d_fwd_short (thin kernel only, with aux): the residual is formed in the kernel instead of read --
    // + sc_gain * NearestUp2( conv1x1(sc_x, sc_w) ), sc_x (N, Ho/2, Wo/2, sc_c) channels-last, sc_w packed [Cout][sc_c]
    const void* sc_x;
    const void* sc_w;
    int sc_c;
    float sc_gain;
};

// conv_thin.hip: direct convolution for thin layers (bf16).  1 = served, 0 = shape not eligible, < 0 = error.
int gan_conv2d_fwd_thin(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                        const float* bias, const void* noise, const float* noise_scale, float slope, int N, int H,
                        int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad,
                        const ConvFuse& fu, hipStream_t st);

// conv_wide.hip: LDS-DMA implicit GEMM for Cin % 64 == 0 (bf16, no input-side stages).  1 / 0 / < 0 as above.
int gan_conv2d_fwd_wide(const void* x, const void* wt, void* y, const float* out_scale, const float* bias,
                        const void* noise, const float* noise_scale, float slope, int N, int H, int W, int Cin, int up2,
                        int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad, const ConvFuse& fu,
                        hipStream_t st);

int64_t gan_conv2d_fwd_wide_workspace(int N, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int Kpad);

// conv_wide_halo.hip: the 3x3 case of the above with the input halo resident in LDS across the nine taps.
int gan_conv2d_fwd_wide_halo(const void* x, const void* wt, void* y, const float* out_scale, const float* bias,
                             const void* noise, const float* noise_scale, float slope, int N, int H, int W, int Cin,
                             int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad,
                             const ConvFuse& fu, hipStream_t st);

// Fused stages of the weight gradient (include/gan_amd.h): leaky-ReLU of x as it is staged, gy given at half
// resolution (nearest-neighbour upsampled in the gather: the adjoint of 2x2 pooling, with gain 0.25), scalar gain.
struct WgradFuse {
    float x_slope;  // 1 = none
    int gy_up2;
    float gain;
};

// Where a weight gradient ends up: fp32 (cout, cin, kh, kw), the parameter's own layout, without the channel
// padding of the kernels' operands.
struct DwOut {
    float* out;
    int cout, cin;
    int accumulate;  // 1: dw += result (the caller's gradient buffer), 0: dw = result
    // the bias gradient of the same layer, sum over (n, oy, ox) of gy (cout floats), or nullptr: a by-product of the
    // thin kernel (one more MFMA per tile row against a fragment of ones); other paths run the channel-sum kernel
    float* bias;
    int bias_accumulate;
};

// conv_thin_wgrad.hip: weight gradient for thin layers (bf16, needs the workspace it reports through `need`).
int gan_conv2d_wgrad_thin(const void* x, const void* gy, const DwOut& dwo, const float* in_scale, int N, int H, int W,
                          int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, float* ws,
                          int64_t ws_floats, int64_t* need, const WgradFuse& fu, hipStream_t st);
// conv_wgrad_tiled.hip: weight gradient for wide layers / small maps (bf16, needs a workspace) and the
// slice reducer: dwo.out[co][ci][tap] = gain * sum_s ws[s][tap][co][ci] (ws: S slices of taps x Cout x Cin).
int64_t gan_wgrad_tiled_workspace(int N, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int py, int px);
int gan_conv2d_wgrad_tiled(const void* x, const void* gy, const DwOut& dwo, const float* in_scale, int N, int H, int W,
                           int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, float* ws,
                           int64_t ws_floats, const WgradFuse& fu, hipStream_t st);
// bias_plane != 0: every slice is followed by Cout channel sums (slice stride taps*Cout*Cin + Cout), folded into dwo.bias
int gan_finalize_dw(const float* ws, int S, int taps, int Cout, int Cin, const DwOut& dwo, float gain, hipStream_t st, int bias_plane = 0,
                    float bias_gain = 1.f);

// upfirdn_mfma.hip: bf16 upfirdn2d with the FIR as banded matrix products on MFMA.  1 / 0 / < 0 as above.
int gan_upfirdn2d_direct(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx, int up, int down,
                         int px1, int py1, int Ho, int Wo, hipStream_t st);
int gan_upfirdn2d_mfma(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx, int up,
                       int down, int px1, int py1, int Ho, int Wo, hipStream_t st);

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
