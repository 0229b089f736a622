// Normalisation layers of the StyleGAN-v1 and Progressive-GAN generators (gfx950), channels-last, HBM-streaming:
//   AdaIN        modelling/common.py:7-16    F.instance_norm(x) * (style_w(w) + 1) + style_b(w)   per (sample, channel) over H x W
//   LayerNorm2d  modelling/common.py:19-24   nn.LayerNorm over the channels of every pixel
// Both have eps = 1e-5 inside the square root and the biased variance (torch defaults).  The LeakyReLU(0.2) that follows
// either of them in the generator blocks (stylegan.py:33-34, progressive_gan.py:89-90,124-125) rides in the same pass.
#include "common.h"

namespace {

constexpr int NTPB = 256;

// ---- per-(n, c) statistics over the pixels: one workgroup per (sample, chunk of CVT channel vectors) -------------
// Two sweeps over the sample's slab (the second one re-reads it from L2 for the slabs that matter): mean first, then the
// centred second moment -- no E[x^2] - E[x]^2 cancellation.
template <typename T, int V>
__global__ void __launch_bounds__(NTPB) chan_stats_k(const T* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t HW, int C, int CVT, int R, float eps) {
    __shared__ float red[NTPB * V];
    __shared__ float mu[NTPB * V];
    const int tid = threadIdx.x;
    const int tv = tid % CVT, tr = tid / CVT;
    const int64_t n = blockIdx.y;
    const int cv = C / V;
    const int vec = blockIdx.x * CVT + tv;
    const bool active = tr < R && vec < cv;
    const int c = vec * V;
    const T* __restrict__ xs = x + n * HW * C + c;
    float acc[V];
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        if (active) {
            for (int64_t row = tr; row < HW; row += R) {
                float v[V];
                load_f<T, V>(xs + row * C, v);
                if (pass == 0) {
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float d = v[e] - mu[tv * V + e];
                        acc[e] = fmaf(d, d, acc[e]);
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) red[tid * V + e] = acc[e];
        __syncthreads();
        if (tr == 0 && vec < cv) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += red[(r * CVT + tv) * V + e];
                if (pass == 0) {
                    mu[tv * V + e] = s / (float)HW;
                } else {
                    mean[n * C + c + e] = mu[tv * V + e];
                    rstd[n * C + c + e] = rsqrtf(s / (float)HW + eps);
                }
            }
        }
        __syncthreads();
    }
}

// y = lrelu_slope(x * a[n, c] + b[n, c])
template <typename T, int V>
__global__ void __launch_bounds__(NTPB) chan_affine_k(const T* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                                      T* __restrict__ y, int64_t HW, int C, int64_t total, float slope) {
    const int64_t idx = (int64_t)blockIdx.x * NTPB + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / V;
    const int c = (int)(idx % cv) * V;
    const int64_t n = (idx / cv) / HW;
    float v[V];
    load_f<T, V>(x + idx * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = lrelu_f(fmaf(v[e], a[n * C + c + e], b[n * C + c + e]), slope);
    store_f<T, V>(y + idx * V, v);
}

// out = g * m(ymask) * a1[n, c] + x * a2[n, c] + a3[n, c]     (the instance-norm backward in one pass)
template <typename T, int V>
__global__ void __launch_bounds__(NTPB) chan_lin2_k(const T* __restrict__ g, const T* __restrict__ ymask, const T* __restrict__ x,
                                                    const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ a3,
                                                    T* __restrict__ out, int64_t HW, int C, int64_t total, float slope) {
    const int64_t idx = (int64_t)blockIdx.x * NTPB + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / V;
    const int c = (int)(idx % cv) * V;
    const int64_t n = (idx / cv) / HW;
    float gv[V], xv[V];
    load_f<T, V>(g + idx * V, gv);
    load_f<T, V>(x + idx * V, xv);
    if (ymask) {
        float m[V];
        load_f<T, V>(ymask + idx * V, m);
#pragma unroll
        for (int e = 0; e < V; ++e) gv[e] *= lrelu_mask(m[e], slope);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int64_t k = n * C + c + e;
        gv[e] = fmaf(gv[e], a1[k], fmaf(xv[e], a2[k], a3[k]));
    }
    store_f<T, V>(out + idx * V, gv);
}

// ---- LayerNorm over the channels of each pixel: 16 lanes per pixel (4 pixels per wave, 16 per workgroup) ----------
// y = lrelu_slope((x - mean_p) * rstd_p * w[c] + b[c]);  stats[p] = (mean_p, rstd_p) kept for the backward pass.
template <typename T>
__global__ void __launch_bounds__(NTPB) ln2d_fwd_k(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                   T* __restrict__ y, float2* __restrict__ stats, int64_t NP, int C, float eps, float slope) {
    const int sub = threadIdx.x & 15;
    const int64_t p = (int64_t)blockIdx.x * (NTPB / 16) + (threadIdx.x >> 4);
    const bool live = p < NP;
    const T* __restrict__ xp = x + (live ? p : 0) * C;
    float s = 0.f;
    for (int c = sub; c < C; c += 16) s += (float)xp[c];
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = sub; c < C; c += 16) {
        const float d = (float)xp[c] - mean;
        q = fmaf(d, d, q);
    }
    for (int off = 8; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    if (sub == 0) stats[p] = make_float2(mean, rstd);
    T* __restrict__ yp = y + p * C;
    for (int c = sub; c < C; c += 16) yp[c] = (T)lrelu_f(fmaf(((float)xp[c] - mean) * rstd, w[c], b[c]), slope);
}

// gz = g * m(y);  h = gz * w;  gx = rstd * (h - mean_c(h) - xhat * mean_c(h * xhat));  gw[c] += gz * xhat, gb[c] += gz
// (fp32 atomics on a zeroed [2][C] buffer, after a per-workgroup fold in LDS).
template <typename T>
__global__ void __launch_bounds__(NTPB) ln2d_bwd_k(const T* __restrict__ g, const T* __restrict__ x, const T* __restrict__ ymask,
                                                   const float* __restrict__ w, const float2* __restrict__ stats, T* __restrict__ gx,
                                                   float* __restrict__ gwb, int64_t NP, int C, float slope, int pix_per_block) {
    extern __shared__ float acc_lds[];  // [2][C]
    for (int i = threadIdx.x; i < 2 * C; i += NTPB) acc_lds[i] = 0.f;
    __syncthreads();
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    for (int i = grp; i < pix_per_block; i += NTPB / 16) {
        const int64_t p = p0 + i;
        const bool live = p < NP;  // uniform over the 16 lanes of a pixel
        const int64_t base = (live ? p : 0) * C;
        const float2 st = live ? stats[p] : make_float2(0.f, 0.f);
        float s1 = 0.f, s2 = 0.f;
        for (int c = sub; c < C; c += 16) {
            float gz = live ? (float)g[base + c] : 0.f;
            if (ymask) gz *= lrelu_mask((float)ymask[base + c], slope);
            const float xh = ((float)x[base + c] - st.x) * st.y;
            const float h = gz * w[c];
            s1 += h;
            s2 = fmaf(h, xh, s2);
            if (live) {
                atomicAdd(&acc_lds[c], gz * xh);
                atomicAdd(&acc_lds[C + c], gz);
            }
        }
        for (int off = 8; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        s1 /= (float)C;
        s2 /= (float)C;
        if (live && gx) {
            for (int c = sub; c < C; c += 16) {
                float gz = (float)g[base + c];
                if (ymask) gz *= lrelu_mask((float)ymask[base + c], slope);
                const float xh = ((float)x[base + c] - st.x) * st.y;
                gx[base + c] = (T)(st.y * (gz * w[c] - s1 - xh * s2));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += NTPB) atomicAdd(&gwb[i], acc_lds[i]);
}

template <typename T>
int chan_stats_launch(const void* x, float* mean, float* rstd, int N, int64_t HW, int C, float eps, hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const bool vec = (C % VV) == 0;
    const int V = vec ? VV : 1;
    const int cv = C / V;
    int CVT = cv < 8 ? cv : 8;  // few channel vectors per workgroup: N * C / (8 V) workgroups share the sweep
    const int R = NTPB / CVT;
    const dim3 grid((unsigned)((cv + CVT - 1) / CVT), (unsigned)N);
    if (vec)
        hipLaunchKernelGGL((chan_stats_k<T, VV>), grid, dim3(NTPB), 0, st, (const T*)x, mean, rstd, HW, C, CVT, R, eps);
    else
        hipLaunchKernelGGL((chan_stats_k<T, 1>), grid, dim3(NTPB), 0, st, (const T*)x, mean, rstd, HW, C, CVT, R, eps);
    GAN_CHECK_LAUNCH("chan_stats");
    return GAN_OK;
}

}  // namespace

extern "C" int gan_chan_stats(const void* x, float* mean, float* rstd, int N, int64_t HW, int C, float eps, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(x && mean && rstd, "chan_stats: null pointer");
    GAN_CHECK_ARG(N > 0 && N <= 65535 && HW > 0 && C > 0, "chan_stats: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "chan_stats: bad dtype %d", dtype);
    if (dtype == GAN_F32) return chan_stats_launch<float>(x, mean, rstd, N, HW, C, eps, (hipStream_t)stream);
    return chan_stats_launch<bf16>(x, mean, rstd, N, HW, C, eps, (hipStream_t)stream);
}

template <typename T>
static int chan_affine_launch(const void* x, const float* a, const float* b, void* y, int N, int64_t HW, int C, float slope, hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const bool vec = (C % VV) == 0;
    const int64_t total = (int64_t)N * HW * (vec ? C / VV : C);
    const int64_t blocks = ceil_div64(total, NTPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "chan_affine: tensor too large");
    if (vec)
        hipLaunchKernelGGL((chan_affine_k<T, VV>), dim3((unsigned)blocks), dim3(NTPB), 0, st, (const T*)x, a, b, (T*)y, HW, C, total, slope);
    else
        hipLaunchKernelGGL((chan_affine_k<T, 1>), dim3((unsigned)blocks), dim3(NTPB), 0, st, (const T*)x, a, b, (T*)y, HW, C, total, slope);
    GAN_CHECK_LAUNCH("chan_affine");
    return GAN_OK;
}

extern "C" int gan_chan_affine(const void* x, const float* a, const float* b, void* y, int N, int64_t HW, int C, float slope, int dtype,
                               gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "chan_affine: bad shape");
    if (N == 0 || HW == 0) return GAN_OK;
    GAN_CHECK_ARG(x && a && b && y, "chan_affine: null pointer");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "chan_affine: bad dtype %d", dtype);
    if (dtype == GAN_F32) return chan_affine_launch<float>(x, a, b, y, N, HW, C, slope, (hipStream_t)stream);
    return chan_affine_launch<bf16>(x, a, b, y, N, HW, C, slope, (hipStream_t)stream);
}

template <typename T>
static int chan_lin2_launch(const void* g, const void* ymask, const void* x, const float* a1, const float* a2, const float* a3, void* out,
                            int N, int64_t HW, int C, float slope, hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const bool vec = (C % VV) == 0;
    const int64_t total = (int64_t)N * HW * (vec ? C / VV : C);
    const int64_t blocks = ceil_div64(total, NTPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "chan_lin2: tensor too large");
    if (vec)
        hipLaunchKernelGGL((chan_lin2_k<T, VV>), dim3((unsigned)blocks), dim3(NTPB), 0, st, (const T*)g, (const T*)ymask, (const T*)x, a1, a2, a3,
                           (T*)out, HW, C, total, slope);
    else
        hipLaunchKernelGGL((chan_lin2_k<T, 1>), dim3((unsigned)blocks), dim3(NTPB), 0, st, (const T*)g, (const T*)ymask, (const T*)x, a1, a2, a3,
                           (T*)out, HW, C, total, slope);
    GAN_CHECK_LAUNCH("chan_lin2");
    return GAN_OK;
}

extern "C" int gan_chan_lin2(const void* g, const void* ymask, const void* x, const float* a1, const float* a2, const float* a3, void* out,
                             int N, int64_t HW, int C, float slope, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "chan_lin2: bad shape");
    if (N == 0 || HW == 0) return GAN_OK;
    GAN_CHECK_ARG(g && x && a1 && a2 && a3 && out, "chan_lin2: null pointer");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "chan_lin2: bad dtype %d", dtype);
    if (dtype == GAN_F32) return chan_lin2_launch<float>(g, ymask, x, a1, a2, a3, out, N, HW, C, slope, (hipStream_t)stream);
    return chan_lin2_launch<bf16>(g, ymask, x, a1, a2, a3, out, N, HW, C, slope, (hipStream_t)stream);
}

extern "C" int gan_layernorm2d_fwd(const void* x, const float* w, const float* b, void* y, float* stats, int64_t NP, int C, float eps,
                                   float slope, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(NP >= 0 && C > 0, "layernorm2d_fwd: bad shape");
    if (NP == 0) return GAN_OK;
    GAN_CHECK_ARG(x && w && b && y && stats, "layernorm2d_fwd: null pointer");
    GAN_CHECK_ARG(((uintptr_t)stats % 8) == 0, "layernorm2d_fwd: stats must be 8-byte aligned");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "layernorm2d_fwd: bad dtype %d", dtype);
    const int64_t blocks = ceil_div64(NP, NTPB / 16);
    GAN_CHECK_ARG(blocks < (1LL << 31), "layernorm2d_fwd: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32)
        hipLaunchKernelGGL((ln2d_fwd_k<float>), dim3((unsigned)blocks), dim3(NTPB), 0, st, (const float*)x, w, b, (float*)y, (float2*)stats, NP, C, eps, slope);
    else
        hipLaunchKernelGGL((ln2d_fwd_k<bf16>), dim3((unsigned)blocks), dim3(NTPB), 0, st, (const bf16*)x, w, b, (bf16*)y, (float2*)stats, NP, C, eps, slope);
    GAN_CHECK_LAUNCH("layernorm2d_fwd");
    return GAN_OK;
}

extern "C" int gan_layernorm2d_bwd(const void* g, const void* x, const void* ymask, const float* w, const float* stats, void* gx, float* gwb,
                                   int64_t NP, int C, float slope, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(NP >= 0 && C > 0 && C <= 8192, "layernorm2d_bwd: bad shape (C <= 8192)");
    GAN_CHECK_ARG(gwb, "layernorm2d_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(gwb, 0, sizeof(float) * 2 * (size_t)C, st) != hipSuccess) {
        gan_set_error("layernorm2d_bwd: memset failed");
        return GAN_ERR_LAUNCH;
    }
    if (NP == 0) return GAN_OK;
    GAN_CHECK_ARG(g && x && w && stats, "layernorm2d_bwd: null pointer");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "layernorm2d_bwd: bad dtype %d", dtype);
    int ppb = (int)(NP / 2048 > 16 ? (NP / 2048 > 4096 ? 4096 : NP / 2048) : 16);  // ~2048 workgroups, >= one pixel per lane group
    ppb = (ppb + 15) / 16 * 16;
    const int64_t blocks = ceil_div64(NP, ppb);
    GAN_CHECK_ARG(blocks < (1LL << 31), "layernorm2d_bwd: tensor too large");
    const size_t lds = sizeof(float) * 2 * (size_t)C;
    if (dtype == GAN_F32)
        hipLaunchKernelGGL((ln2d_bwd_k<float>), dim3((unsigned)blocks), dim3(NTPB), lds, st, (const float*)g, (const float*)x, (const float*)ymask, w,
                           (const float2*)stats, (float*)gx, gwb, NP, C, slope, ppb);
    else
        hipLaunchKernelGGL((ln2d_bwd_k<bf16>), dim3((unsigned)blocks), dim3(NTPB), lds, st, (const bf16*)g, (const bf16*)x, (const bf16*)ymask, w,
                           (const float2*)stats, (bf16*)gx, gwb, NP, C, slope, ppb);
    GAN_CHECK_LAUNCH("layernorm2d_bwd");
    return GAN_OK;
}
