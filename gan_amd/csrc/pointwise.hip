// HBM-streaming channels-last kernels: upfirdn2d, fused affine/noise/residual/leaky-ReLU,
// its backward building block, channel reductions, MinibatchStdDev statistic, EMA lerp.
// Every thread moves 16 bytes per access (8 bf16 / 4 fp32) when C allows, scalar otherwise.
#include "common.h"

static constexpr int TPB = 256;

// =========================================================================================
// upfirdn2d  (reference: modelling/nvidia_ops.py:18-34)
// Polyphase form: for output (oy, ox) only taps i with (oy*down + i - py1) % up == 0 touch a
// real sample, so the tap loops start at the first such tap and step by `up`; no zero-stuffed
// tensor is ever materialised.
// =========================================================================================
template <typename T, int V>
__global__ void __launch_bounds__(TPB) upfirdn2d_k(const T* __restrict__ x, const float* __restrict__ taps,
                                                   T* __restrict__ y, int H, int W, int C, int ky, int kx, int up,
                                                   int down, int px1, int py1, int Ho, int Wo, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / V;
    const int c = (int)(idx % cv) * V;
    int64_t p = idx / cv;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int64_t n = p / Ho;

    const int by = oy * down - py1;
    const int bx = ox * down - px1;
    int i0 = (-by) % up;
    if (i0 < 0) i0 += up;
    int j0 = (-bx) % up;
    if (j0 < 0) j0 += up;

    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;

    for (int i = i0; i < ky; i += up) {
        const int iy = (by + i) / up;  // exact: by + i is a multiple of up
        if (iy < 0 || iy >= H) continue;
        for (int j = j0; j < kx; j += up) {
            const int ix = (bx + j) / up;
            if (ix < 0 || ix >= W) continue;
            const float t = taps[i * kx + j];
            float v[V];
            load_f<T, V>(x + ((n * H + iy) * W + ix) * C + c, v);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = fmaf(t, v[e], acc[e]);
        }
    }
    store_f<T, V>(y + idx * V, acc);
}

// LDS-tiled form (C a multiple of the 16-byte vector): a workgroup owns TH x 32 output pixels x NVB
// channel vectors.  The input region those outputs touch is read from HBM once with 16-byte loads into
// LDS ([row][col][vector]); every tap is then an LDS read, so HBM sees ~(1 + halo) x the input instead
// of each tap missing L1 separately.
template <typename T, int V>
__global__ void __launch_bounds__(TPB) upfirdn2d_tile_k(const T* __restrict__ x, const float* __restrict__ taps,
                                                        T* __restrict__ y, int H, int W, int C, int ky, int kx,
                                                        int up, int down, int px1, int py1, int Ho, int Wo, int TH,
                                                        int nvb_log2, int tiles_x, int RWmax) {
    extern __shared__ __attribute__((aligned(16))) char tile_lds[];
    __shared__ float tap_lds[256];
    const int tid = threadIdx.x;
    const int NVB = 1 << nvb_log2;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int vb = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int th = min(TH, Ho - oy0), tw = min(32, Wo - ox0);

    for (int i = tid; i < ky * kx; i += TPB) tap_lds[i] = taps[i];

    // input region [iy_lo, iy_hi] x [ix_lo, ix_hi] touched by this tile (clamped to the image)
    const int by0 = oy0 * down - py1, bx0 = ox0 * down - px1;
    int iy_lo = by0 >= 0 ? (by0 + up - 1) / up : -((-by0) / up);
    int ix_lo = bx0 >= 0 ? (bx0 + up - 1) / up : -((-bx0) / up);
    const int ey = (oy0 + th - 1) * down - py1 + ky - 1, ex = (ox0 + tw - 1) * down - px1 + kx - 1;
    int iy_hi = ey >= 0 ? ey / up : -1, ix_hi = ex >= 0 ? ex / up : -1;
    iy_lo = max(iy_lo, 0);
    ix_lo = max(ix_lo, 0);
    iy_hi = min(iy_hi, H - 1);
    ix_hi = min(ix_hi, W - 1);
    const int RH = iy_hi - iy_lo + 1, RW = ix_hi - ix_lo + 1;
    if (RH > 0 && RW > 0) {
        const int rowv = RW << nvb_log2;
        const int total = RH * rowv;
        const T* __restrict__ xb = x + ((n * H + iy_lo) * W + ix_lo) * C + (int64_t)vb * NVB * V;
        for (int idx = tid; idx < total; idx += TPB) {
            const int ry = idx / rowv, rem = idx - ry * rowv;
            const int rx = rem >> nvb_log2, v = rem & (NVB - 1);
            const uint4 q = *reinterpret_cast<const uint4*>(xb + ((int64_t)ry * W + rx) * C + v * V);
            *reinterpret_cast<uint4*>(tile_lds + ((ry * RWmax + rx) * NVB + v) * 16) = q;
        }
    }
    __syncthreads();

    const int outs = (th * 32) << nvb_log2;
    for (int idx = tid; idx < outs; idx += TPB) {
        const int v = idx & (NVB - 1);
        const int pix = idx >> nvb_log2;
        const int ox_l = pix & 31, oy_l = pix >> 5;
        if (ox_l >= tw) continue;
        const int oy = oy0 + oy_l, ox = ox0 + ox_l;
        const int by = oy * down - py1, bx = ox * down - px1;
        int i0 = (-by) % up;
        if (i0 < 0) i0 += up;
        int j0 = (-bx) % up;
        if (j0 < 0) j0 += up;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int i = i0; i < ky; i += up) {
            const int iy = (by + i) / up;
            if (iy < iy_lo || iy > iy_hi) continue;
            const char* rowp = tile_lds + (((iy - iy_lo) * RWmax - ix_lo) * NVB + v) * 16;
            for (int j = j0; j < kx; j += up) {
                const int ix = (bx + j) / up;
                if (ix < ix_lo || ix > ix_hi) continue;
                const float t = tap_lds[i * kx + j];
                float vv[V];
                load_f<T, V>(reinterpret_cast<const T*>(rowp + ix * NVB * 16), vv);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(t, vv[e], acc[e]);
            }
        }
        store_f<T, V>(y + ((n * Ho + oy) * Wo + ox) * C + ((int64_t)vb * NVB + v) * V, acc);
    }
}

template <typename T>
static int upfirdn2d_launch(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx,
                            int up, int down, int px1, int py1, int Ho, int Wo, hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const bool vec = (C % VV) == 0;
    const int64_t total = (int64_t)N * Ho * Wo * (vec ? C / VV : C);
    if (total == 0) return GAN_OK;
    if constexpr (sizeof(T) == 2) {
        // a factor of two: direct polyphase kernel (upfirdn_direct.hip); stride 1: the matrix-core kernel
        const int rd = gan_upfirdn2d_direct(x, taps, y, N, H, W, C, ky, kx, up, down, px1, py1, Ho, Wo, st);
        if (rd != 0) return rd < 0 ? rd : GAN_OK;
        const int rc = gan_upfirdn2d_mfma(x, taps, y, N, H, W, C, ky, kx, up, down, px1, py1, Ho, Wo, st);
        if (rc != 0) return rc < 0 ? rc : GAN_OK;
    }
    if (vec && ky * kx <= 256 && N <= 65535 && Wo >= 8) {
        const int cv = C / VV;
        int nvb_log2 = 0;
        while (nvb_log2 < 3 && (cv % (2 << nvb_log2)) == 0) ++nvb_log2;
        const int NVB = 1 << nvb_log2;
        // rows per tile: as many as keep the LDS region under 48 KiB (at least 1), at most 8
        const int RWmax = (31 * down + kx - 1) / up + 2;
        int TH = 8;
        auto region_bytes = [&](int th) { return (((th - 1) * down + ky - 1) / up + 2) * RWmax * NVB * 16; };
        while (TH > 1 && region_bytes(TH) > 48 * 1024) TH >>= 1;
        if (TH > Ho) TH = Ho;
        const int lds = region_bytes(TH);
        const int tiles_x = (Wo + 31) / 32, tiles_y = (Ho + TH - 1) / TH;
        if (lds <= 64 * 1024 && (int64_t)tiles_x * tiles_y < (1LL << 31) && cv / NVB <= 65535) {
            hipLaunchKernelGGL((upfirdn2d_tile_k<T, VV>), dim3((unsigned)(tiles_x * tiles_y), (unsigned)(cv / NVB), (unsigned)N),
                               dim3(TPB), (size_t)lds, st, (const T*)x, taps, (T*)y, H, W, C, ky, kx, up, down, px1, py1,
                               Ho, Wo, TH, nvb_log2, tiles_x, RWmax);
            GAN_CHECK_LAUNCH("upfirdn2d(tile)");
            return GAN_OK;
        }
    }
    const int64_t blocks = ceil_div64(total, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "upfirdn2d: tensor too large");
    if (vec)
        hipLaunchKernelGGL((upfirdn2d_k<T, VV>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const T*)x, taps, (T*)y, H,
                           W, C, ky, kx, up, down, px1, py1, Ho, Wo, total);
    else
        hipLaunchKernelGGL((upfirdn2d_k<T, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const T*)x, taps, (T*)y, H,
                           W, C, ky, kx, up, down, px1, py1, Ho, Wo, total);
    GAN_CHECK_LAUNCH("upfirdn2d");
    return GAN_OK;
}

extern "C" int gan_upfirdn2d(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx,
                             int up, int down, int px1, int py1, int Ho, int Wo, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0 && C > 0 && Ho >= 0 && Wo >= 0, "upfirdn2d: bad shape");
    if (N == 0 || Ho == 0 || Wo == 0) return GAN_OK;  // empty batch: nothing to launch, pointers may be null
    GAN_CHECK_ARG(x && taps && y, "upfirdn2d: null pointer");
    GAN_CHECK_ARG(ky >= 1 && kx >= 1 && ky <= 64 && kx <= 64, "upfirdn2d: bad FIR size %dx%d", ky, kx);
    GAN_CHECK_ARG(up >= 1 && down >= 1, "upfirdn2d: up/down must be >= 1");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "upfirdn2d: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32) return upfirdn2d_launch<float>(x, taps, y, N, H, W, C, ky, kx, up, down, px1, py1, Ho, Wo, st);
    return upfirdn2d_launch<bf16>(x, taps, y, N, H, W, C, ky, kx, up, down, px1, py1, Ho, Wo, st);
}

// =========================================================================================
// fused affine + noise + residual + leaky-ReLU forward
// =========================================================================================
template <typename T, int V>
__global__ void __launch_bounds__(TPB) affine_act_fwd_k(const T* __restrict__ x, const float* __restrict__ a,
                                                        int64_t a_sn, const float* __restrict__ b,
                                                        const T* __restrict__ noise, const float* __restrict__ ns,
                                                        const T* __restrict__ r, T* __restrict__ y, int64_t HW, int C,
                                                        int64_t total, float slope) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / V;
    const int c = (int)(idx % cv) * V;
    const int64_t pix = idx / cv;
    float v[V];
    load_f<T, V>(x + idx * V, v);
    if (a) {
        const int64_t n = pix / HW;
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= a[n * a_sn + c + e];
    }
    if (b) {
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] += b[c + e];
    }
    if (noise) {
        const float nz = (float)noise[pix];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = fmaf(nz, ns[c + e], v[e]);
    }
    if (r) {
        float rv[V];
        load_f<T, V>(r + idx * V, rv);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] += rv[e];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = lrelu_f(v[e], slope);
    store_f<T, V>(y + idx * V, v);
}

template <typename T>
static int affine_act_fwd_launch(const void* x, const float* a, int64_t a_sn, const float* b, const void* noise,
                                 const float* ns, const void* r, void* y, int N, int64_t HW, int C, float slope,
                                 hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const bool vec = (C % VV) == 0;
    const int64_t total = (int64_t)N * HW * (vec ? C / VV : C);
    if (total == 0) return GAN_OK;
    const int64_t blocks = ceil_div64(total, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "affine_act_fwd: tensor too large");
    if (vec)
        hipLaunchKernelGGL((affine_act_fwd_k<T, VV>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const T*)x, a, a_sn, b,
                           (const T*)noise, ns, (const T*)r, (T*)y, HW, C, total, slope);
    else
        hipLaunchKernelGGL((affine_act_fwd_k<T, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const T*)x, a, a_sn, b,
                           (const T*)noise, ns, (const T*)r, (T*)y, HW, C, total, slope);
    GAN_CHECK_LAUNCH("affine_act_fwd");
    return GAN_OK;
}

extern "C" int gan_affine_act_fwd(const void* x, const float* a, int64_t a_stride_n, const float* b, const void* noise,
                                  const float* ns, const void* r, void* y, int N, int64_t HW, int C, float slope,
                                  int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "affine_act_fwd: bad shape");
    if (N == 0 || HW == 0) return GAN_OK;
    GAN_CHECK_ARG(x && y, "affine_act_fwd: null pointer");
    GAN_CHECK_ARG((noise == nullptr) == (ns == nullptr), "affine_act_fwd: noise and noise scale go together");
    GAN_CHECK_ARG(a_stride_n == 0 || a_stride_n == C, "affine_act_fwd: a_stride_n must be 0 or C");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "affine_act_fwd: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32) return affine_act_fwd_launch<float>(x, a, a_stride_n, b, noise, ns, r, y, N, HW, C, slope, st);
    return affine_act_fwd_launch<bf16>(x, a, a_stride_n, b, noise, ns, r, y, N, HW, C, slope, st);
}

// =========================================================================================
// out = g * m(ymask) * a      (backward / double-backward building block)
// =========================================================================================
template <typename T, int V>
__global__ void __launch_bounds__(TPB) mask_scale_k(const T* __restrict__ g, const T* __restrict__ ymask,
                                                    const float* __restrict__ a, int64_t a_sn, T* __restrict__ out,
                                                    int64_t HW, int C, int64_t total, float slope) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / V;
    const int c = (int)(idx % cv) * V;
    float v[V];
    load_f<T, V>(g + idx * V, v);
    if (ymask) {
        float m[V];
        load_f<T, V>(ymask + idx * V, m);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= lrelu_mask(m[e], slope);
    }
    if (a) {
        const int64_t n = (idx / cv) / HW;
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= a[n * a_sn + c + e];
    }
    store_f<T, V>(out + idx * V, v);
}

template <typename T>
static int mask_scale_launch(const void* g, const void* ymask, const float* a, int64_t a_sn, void* out, int N,
                             int64_t HW, int C, float slope, hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const bool vec = (C % VV) == 0;
    const int64_t total = (int64_t)N * HW * (vec ? C / VV : C);
    if (total == 0) return GAN_OK;
    const int64_t blocks = ceil_div64(total, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "mask_scale: tensor too large");
    if (vec)
        hipLaunchKernelGGL((mask_scale_k<T, VV>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const T*)g,
                           (const T*)ymask, a, a_sn, (T*)out, HW, C, total, slope);
    else
        hipLaunchKernelGGL((mask_scale_k<T, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const T*)g, (const T*)ymask,
                           a, a_sn, (T*)out, HW, C, total, slope);
    GAN_CHECK_LAUNCH("mask_scale");
    return GAN_OK;
}

extern "C" int gan_mask_scale(const void* g, const void* ymask, const float* a, int64_t a_stride_n, void* out, int N,
                              int64_t HW, int C, float slope, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "mask_scale: bad shape");
    if (N == 0 || HW == 0) return GAN_OK;
    GAN_CHECK_ARG(g && out, "mask_scale: null pointer");
    GAN_CHECK_ARG(a_stride_n == 0 || a_stride_n == C, "mask_scale: a_stride_n must be 0 or C");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "mask_scale: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32) return mask_scale_launch<float>(g, ymask, a, a_stride_n, out, N, HW, C, slope, st);
    return mask_scale_launch<bf16>(g, ymask, a, a_stride_n, out, N, HW, C, slope, st);
}

// =========================================================================================
// channel reductions:  out[n*out_sn + c] += sum_hw A * m(ymask) * Bterm
// Block = [R rows][CVT channel vectors]; each thread walks rows r, r+R, ...; partial sums are
// folded across the lanes of a wave that hold the same channel vector with __shfl_xor (when
// CVT is a power of two below 64), then across waves through LDS; every workgroup then STORES its
// partial row to part[n][block][C] and chan_fold_k sums the rows in a fixed order: no atomics (deterministic), nothing to
// zero, and the batch sum needs no separate reduction over the (N, C) partials.
// =========================================================================================
template <typename T, int V>
__global__ void __launch_bounds__(TPB) chan_dot_k(const T* __restrict__ A, const T* __restrict__ B, int b_mode,
                                                  const T* __restrict__ ymask, float slope, float* __restrict__ part,
                                                  int64_t HW, int C, int CVT, int R, int64_t rows_per_block) {
    __shared__ float red[TPB * V];
    const int tid = threadIdx.x;
    const int tv = tid % CVT;
    const int tr = tid / CVT;
    const int64_t n = blockIdx.y;
    const int cv = C / V;
    const int vec = blockIdx.z * CVT + tv;  // channel vector index
    const bool active = (tr < R) && (vec < cv);
    const int c = vec * V;

    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;

    if (active) {
        const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t row1 = row0 + rows_per_block;
        if (row1 > HW) row1 = HW;
        for (int64_t row = row0 + tr; row < row1; row += R) {
            const int64_t pix = n * HW + row;
            const int64_t off = pix * C + c;
            float v[V];
            load_f<T, V>(A + off, v);
            if (ymask) {
                float m[V];
                load_f<T, V>(ymask + off, m);
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] *= lrelu_mask(m[e], slope);
            }
            if (b_mode == 1) {
                float w[V];
                load_f<T, V>(B + off, w);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(v[e], w[e], acc[e]);
            } else if (b_mode == 2) {
                const float w = (float)B[pix];
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(v[e], w, acc[e]);
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += v[e];
            }
        }
    }

    const bool pow2 = (CVT & (CVT - 1)) == 0 && CVT < 64;
    if (pow2) {
        // lanes l, l+CVT, l+2CVT, ... of a wave hold the same channel vector (TPB % CVT == 0,
        // so idle threads do not exist and every lane carries a valid partial or zeros).
        for (int off = CVT; off < 64; off <<= 1) {
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += __shfl_xor(acc[e], off);
        }
        const int lane = tid & 63;
        const int wave = tid >> 6;
        if (lane < CVT) {
#pragma unroll
            for (int e = 0; e < V; ++e) red[(wave * CVT + lane) * V + e] = acc[e];
        }
        __syncthreads();
        if (tid < CVT && (blockIdx.z * CVT + tid) < cv) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float s = 0.f;
                for (int w = 0; w < TPB / 64; ++w) s += red[(w * CVT + tid) * V + e];
                part[(n * gridDim.x + blockIdx.x) * C + (int64_t)(blockIdx.z * CVT + tid) * V + e] = s;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) red[tid * V + e] = acc[e];
        __syncthreads();
        if (tr == 0 && vec < cv) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += red[(r * CVT + tv) * V + e];
                part[(n * gridDim.x + blockIdx.x) * C + c + e] = s;
            }
        }
    }
}

// out[g][c] = sum_r part[g * rows + r][c]: G groups of `rows` partial rows (per-sample sums: G = N, rows = blocks per
// sample; batch sums: G = 1, rows = N * blocks).  A workgroup = RG row lanes x CW channel vectors of V floats (RG a power
// of two sized to the row count by the launcher); row lanes are folded by a tree through LDS, in a fixed order.
template <int V>
__global__ void __launch_bounds__(TPB) chan_fold_k(const float* __restrict__ part, float* __restrict__ out, int rows, int C, int CW, int RG,
                                                   int accum) {
    __shared__ float red[TPB * V];
    const int tc = threadIdx.x % CW, tr = threadIdx.x / CW;
    const int c = (blockIdx.x * CW + tc) * V;
    const int64_t g = blockIdx.y;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (c < C && tr < RG) {  // (CW is capped by the channel count: lanes past RG * CW idle)
        const float* __restrict__ src = part + g * rows * (int64_t)C + c;
#pragma unroll 4
        for (int r = tr; r < rows; r += RG) {
            float t[V];
            load_f<float, V>(src + (int64_t)r * C, t);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) red[threadIdx.x * V + e] = acc[e];
    __syncthreads();
    for (int s = RG >> 1; s > 0; s >>= 1) {
        if (tr < s) {
#pragma unroll
            for (int e = 0; e < V; ++e) red[threadIdx.x * V + e] += red[(threadIdx.x + s * CW) * V + e];
        }
        __syncthreads();
    }
    if (tr == 0 && c < C) {
#pragma unroll
        for (int e = 0; e < V; ++e) out[g * C + c + e] = (accum ? out[g * C + c + e] : 0.f) + red[threadIdx.x * V + e];
    }
}

struct ChanDotPlan {
    int V, cv, CVT, R, zchunks;
    int64_t bps, rpb;
};

template <typename T>
static ChanDotPlan chan_dot_plan(int N, int64_t HW, int C) {
    ChanDotPlan pl;
    constexpr int VV = VecOf<T>::V;
    pl.V = (C % VV) == 0 ? VV : 1;
    pl.cv = C / pl.V;
    pl.CVT = pl.cv < TPB ? pl.cv : TPB;
    pl.R = TPB / pl.CVT;
    pl.zchunks = (pl.cv + pl.CVT - 1) / pl.CVT;
#ifndef GAN_CHANDOT_ROWS
#define GAN_CHANDOT_ROWS 4
#endif
    int64_t bps = ceil_div64(HW, (int64_t)pl.R * GAN_CHANDOT_ROWS);  // >= this many rows per thread when possible
    const int64_t cap = 4096 / ((int64_t)N * pl.zchunks) > 0 ? 4096 / ((int64_t)N * pl.zchunks) : 1;
    if (bps > cap) bps = cap;
    if (bps < 1) bps = 1;
    int64_t rpb = ceil_div64(HW, bps);
    rpb = ceil_div64(rpb, pl.R) * pl.R;
    pl.rpb = rpb;
    pl.bps = ceil_div64(HW, rpb);
    return pl;
}

template <typename T>
static int chan_dot_launch(const void* A, const void* B, int b_mode, const void* ymask, float slope, float* out,
                           int64_t out_sn, int N, int64_t HW, int C, float* ws, int accum, hipStream_t st) {
    constexpr int VV = VecOf<T>::V;
    const ChanDotPlan pl = chan_dot_plan<T>(N, HW, C);
    GAN_CHECK_ARG(N <= 65535 && pl.zchunks <= 65535, "chan_dot: N or C too large for the grid");
    const bool direct = out_sn != 0 && pl.bps == 1 && !accum;  // one workgroup per (sample, channel chunk): its row IS the result
    float* part = direct ? out : ws;
    dim3 grid((unsigned)pl.bps, (unsigned)N, (unsigned)pl.zchunks);
    if (pl.V == VV)
        hipLaunchKernelGGL((chan_dot_k<T, VV>), grid, dim3(TPB), 0, st, (const T*)A, (const T*)B, b_mode,
                           (const T*)ymask, slope, part, HW, C, pl.CVT, pl.R, pl.rpb);
    else
        hipLaunchKernelGGL((chan_dot_k<T, 1>), grid, dim3(TPB), 0, st, (const T*)A, (const T*)B, b_mode,
                           (const T*)ymask, slope, part, HW, C, pl.CVT, pl.R, pl.rpb);
    GAN_CHECK_LAUNCH("chan_dot");
    if (direct) return GAN_OK;
    const int rows = (int)(out_sn ? pl.bps : (int64_t)N * pl.bps);
    const int FV = (C % 4) == 0 ? 4 : 1, cvf = C / FV;
    int RG = 1;  // row lanes: ~4 rows per lane, at most the whole workgroup
    while (RG < TPB && RG * 4 < rows) RG <<= 1;
    int CW = TPB / RG;
    if (CW > cvf) CW = cvf;
    const dim3 fgrid((unsigned)((cvf + CW - 1) / CW), (unsigned)(out_sn ? N : 1));
    if (FV == 4)
        hipLaunchKernelGGL(chan_fold_k<4>, fgrid, dim3(TPB), 0, st, ws, out, rows, C, CW, RG, accum);
    else
        hipLaunchKernelGGL(chan_fold_k<1>, fgrid, dim3(TPB), 0, st, ws, out, rows, C, CW, RG, accum);
    GAN_CHECK_LAUNCH("chan_dot(fold)");
    return GAN_OK;
}

extern "C" int64_t gan_chan_dot_workspace(int N, int64_t HW, int C, int dtype) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const ChanDotPlan pl = dtype == GAN_F32 ? chan_dot_plan<float>(N, HW, C) : chan_dot_plan<bf16>(N, HW, C);
    return (int64_t)N * pl.bps * C;
}

extern "C" int gan_chan_dot(const void* A, const void* B, int b_mode, const void* ymask, float slope, float* out,
                            int64_t out_stride_n, int N, int64_t HW, int C, float* workspace, int64_t workspace_floats, int accumulate,
                            int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(accumulate == 0 || accumulate == 1, "chan_dot: accumulate must be 0 or 1");
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "chan_dot: bad shape");
    GAN_CHECK_ARG((A || N == 0 || HW == 0) && (out || (out_stride_n && N == 0)), "chan_dot: null pointer");
    GAN_CHECK_ARG(b_mode >= 0 && b_mode <= 2 && ((b_mode == 0) == (B == nullptr)), "chan_dot: b_mode %d vs B", b_mode);
    GAN_CHECK_ARG(out_stride_n == 0 || out_stride_n == C, "chan_dot: out_stride_n must be 0 or C");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "chan_dot: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(float) * (size_t)(out_stride_n ? (int64_t)N * C : C);
    if (bytes == 0) return GAN_OK;
    if (N == 0 || HW == 0) {
        if (accumulate) return GAN_OK;
        if (hipMemsetAsync(out, 0, bytes, st) != hipSuccess) {
            gan_set_error("chan_dot: memset failed");
            return GAN_ERR_LAUNCH;
        }
        return GAN_OK;
    }
    GAN_CHECK_ARG(workspace != nullptr && workspace_floats >= gan_chan_dot_workspace(N, HW, C, dtype),
                  "chan_dot: needs a workspace of gan_chan_dot_workspace(...) = %lld floats", (long long)gan_chan_dot_workspace(N, HW, C, dtype));
    GAN_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "chan_dot: the workspace must be 16-byte aligned (the fold reads it with 16-byte loads)");
    if (dtype == GAN_F32) return chan_dot_launch<float>(A, B, b_mode, ymask, slope, out, out_stride_n, N, HW, C, workspace, accumulate, st);
    return chan_dot_launch<bf16>(A, B, b_mode, ymask, slope, out, out_stride_n, N, HW, C, workspace, accumulate, st);
}

// =========================================================================================
// Fused passes of the modulated convolution's backward (autograd of ModulatedConv2d, stylegan2.py:29-36):
// what used to be three channel reductions and a mask/scale pass over the same two tensors is ONE pass.
//   modconv_bwd_pre:  gz = gy * m(y) * dinv[n,c];  sums[n][0][c] = sum_hw gy*m(y)        (bias gradient)
//                                                  sums[n][1][c] = sum_hw gy*m(y)*noise  (noise-scale gradient)
//                                                  sums[n][2][c] = sum_hw gy*y           (demodulation gradient)
//   scale_dot:        out = A * a[n,c];            dot[n][c]     = sum_hw A*B            (style gradient + dx)
// Thread mapping and the block reduction are chan_dot_k's.
// =========================================================================================
template <int V>
__device__ __forceinline__ void chan_block_reduce(float (&acc)[V], float* red, float* __restrict__ dst, int CVT, int R, int cv,
                                                  int tv, int tr, int vec, int c) {
    const int tid = threadIdx.x;
    const bool pow2 = (CVT & (CVT - 1)) == 0 && CVT < 64;
    __syncthreads();  // `red` may still be read by the previous reduction
    if (pow2) {
        for (int off = CVT; off < 64; off <<= 1) {
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += __shfl_xor(acc[e], off);
        }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < CVT) {
#pragma unroll
            for (int e = 0; e < V; ++e) red[(wave * CVT + lane) * V + e] = acc[e];
        }
        __syncthreads();
        if (tid < CVT && (blockIdx.z * CVT + tid) < cv) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float sum = 0.f;
                for (int w = 0; w < TPB / 64; ++w) sum += red[(w * CVT + tid) * V + e];
                atomicAdd(dst + (int64_t)(blockIdx.z * CVT + tid) * V + e, sum);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) red[tid * V + e] = acc[e];
        __syncthreads();
        if (tr == 0 && vec < cv) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float sum = 0.f;
                for (int r = 0; r < R; ++r) sum += red[(r * CVT + tv) * V + e];
                atomicAdd(dst + c + e, sum);
            }
        }
    }
}

template <typename T, int V>
__global__ void __launch_bounds__(TPB) modconv_bwd_pre_k(const T* __restrict__ gy, const T* __restrict__ y, const T* __restrict__ noise,
                                                         const float* __restrict__ dinv, float slope, T* __restrict__ gz,
                                                         float* __restrict__ sums, int64_t HW, int C, int CVT, int R,
                                                         int64_t rows_per_block) {
    __shared__ float red[TPB * V];
    const int tid = threadIdx.x;
    const int tv = tid % CVT, tr = tid / CVT;
    const int64_t n = blockIdx.y;
    const int cv = C / V;
    const int vec = blockIdx.z * CVT + tv;
    const bool active = (tr < R) && (vec < cv);
    const int c = vec * V;
    float a0[V], a1[V], a2[V], dv[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        a0[e] = a1[e] = a2[e] = 0.f;
        dv[e] = (dinv && active) ? dinv[n * C + c + e] : 1.f;
    }
    if (active) {
        const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t row1 = row0 + rows_per_block;
        if (row1 > HW) row1 = HW;
        for (int64_t row = row0 + tr; row < row1; row += R) {
            const int64_t pix = n * HW + row;
            const int64_t off = pix * C + c;
            float g[V], yv[V], o[V];
            load_f<T, V>(gy + off, g);
            if (y) load_f<T, V>(y + off, yv);
            const float nz = noise ? (float)noise[pix] : 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float gm = y ? g[e] * lrelu_mask(yv[e], slope) : g[e];
                a0[e] += gm;
                a1[e] = fmaf(gm, nz, a1[e]);
                if (y) a2[e] = fmaf(g[e], yv[e], a2[e]);
                o[e] = gm * dv[e];
            }
            store_f<T, V>(gz + off, o);
        }
    }
    float* const dst = sums + n * 3 * C;
    chan_block_reduce<V>(a0, red, dst, CVT, R, cv, tv, tr, vec, c);
    if (noise) chan_block_reduce<V>(a1, red, dst + C, CVT, R, cv, tv, tr, vec, c);
    if (y) chan_block_reduce<V>(a2, red, dst + 2 * C, CVT, R, cv, tv, tr, vec, c);
}

template <typename T, int V>
__global__ void __launch_bounds__(TPB) scale_dot_k(const T* __restrict__ A, const T* __restrict__ B, const float* __restrict__ a,
                                                   T* __restrict__ out, float* __restrict__ dot, int64_t HW, int C, int CVT, int R,
                                                   int64_t rows_per_block) {
    __shared__ float red[TPB * V];
    const int tid = threadIdx.x;
    const int tv = tid % CVT, tr = tid / CVT;
    const int64_t n = blockIdx.y;
    const int cv = C / V;
    const int vec = blockIdx.z * CVT + tv;
    const bool active = (tr < R) && (vec < cv);
    const int c = vec * V;
    float acc[V], sv[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        acc[e] = 0.f;
        sv[e] = active ? a[n * C + c + e] : 0.f;
    }
    if (active) {
        const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t row1 = row0 + rows_per_block;
        if (row1 > HW) row1 = HW;
        for (int64_t row = row0 + tr; row < row1; row += R) {
            const int64_t off = (n * HW + row) * C + c;
            float g[V], xv[V], o[V];
            load_f<T, V>(A + off, g);
            load_f<T, V>(B + off, xv);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                acc[e] = fmaf(g[e], xv[e], acc[e]);
                o[e] = g[e] * sv[e];
            }
            store_f<T, V>(out + off, o);
        }
    }
    chan_block_reduce<V>(acc, red, dot + n * C, CVT, R, cv, tv, tr, vec, c);
}

#ifndef GAN_CHAN_ROWS
#define GAN_CHAN_ROWS 8  // rows per thread before a workgroup pays its block reduction and atomics (4: -0.25 ms per iteration at batch 32)
#endif
#ifndef GAN_CHAN_CAP
#define GAN_CHAN_CAP 8192
#endif
struct ChanGrid {
    int V, CVT, R, zchunks;
    int64_t bps, rpb;
};
template <typename T>
static ChanGrid chan_grid(int N, int64_t HW, int C) {
    ChanGrid g;
    constexpr int VV = VecOf<T>::V;
    g.V = (C % VV) == 0 ? VV : 1;
    const int cv = C / g.V;
    g.CVT = cv < TPB ? cv : TPB;
    g.R = TPB / g.CVT;
    g.zchunks = (cv + g.CVT - 1) / g.CVT;
    int64_t bps = ceil_div64(HW, (int64_t)g.R * GAN_CHAN_ROWS);
    const int64_t cap = GAN_CHAN_CAP / ((int64_t)N * g.zchunks) > 0 ? GAN_CHAN_CAP / ((int64_t)N * g.zchunks) : 1;
    if (bps > cap) bps = cap;
    if (bps < 1) bps = 1;
    int64_t rpb = ceil_div64(HW, bps);
    rpb = ceil_div64(rpb, g.R) * g.R;
    g.bps = ceil_div64(HW, rpb);
    g.rpb = rpb;
    return g;
}

template <typename T>
static int modconv_bwd_pre_launch(const void* gy, const void* y, const void* noise, const float* dinv, float slope, void* gz,
                                  float* sums, int N, int64_t HW, int C, hipStream_t st) {
    const ChanGrid g = chan_grid<T>(N, HW, C);
    GAN_CHECK_ARG(N <= 65535 && g.zchunks <= 65535, "modconv_bwd_pre: N or C too large for the grid");
    dim3 grid((unsigned)g.bps, (unsigned)N, (unsigned)g.zchunks);
    if (g.V > 1)
        hipLaunchKernelGGL((modconv_bwd_pre_k<T, VecOf<T>::V>), grid, dim3(TPB), 0, st, (const T*)gy, (const T*)y, (const T*)noise, dinv,
                           slope, (T*)gz, sums, HW, C, g.CVT, g.R, g.rpb);
    else
        hipLaunchKernelGGL((modconv_bwd_pre_k<T, 1>), grid, dim3(TPB), 0, st, (const T*)gy, (const T*)y, (const T*)noise, dinv, slope,
                           (T*)gz, sums, HW, C, g.CVT, g.R, g.rpb);
    GAN_CHECK_LAUNCH("modconv_bwd_pre");
    return GAN_OK;
}

extern "C" int gan_modconv_bwd_pre(const void* gy, const void* y, const void* noise, const float* dinv, float slope, void* gz,
                                   float* sums, int sums_zeroed, int N, int64_t HW, int C, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "modconv_bwd_pre: bad shape");
    GAN_CHECK_ARG(sums && (N == 0 || HW == 0 || (gy && gz)), "modconv_bwd_pre: null pointer");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "modconv_bwd_pre: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(float) * (size_t)N * 3 * C;
    if (bytes == 0) return GAN_OK;
    if (!sums_zeroed && hipMemsetAsync(sums, 0, bytes, st) != hipSuccess) {
        gan_set_error("modconv_bwd_pre: memset failed");
        return GAN_ERR_LAUNCH;
    }
    if (HW == 0) return GAN_OK;
    if (dtype == GAN_F32) return modconv_bwd_pre_launch<float>(gy, y, noise, dinv, slope, gz, sums, N, HW, C, st);
    return modconv_bwd_pre_launch<bf16>(gy, y, noise, dinv, slope, gz, sums, N, HW, C, st);
}

// Folds modconv_bwd_pre's per-sample sums into the parameter gradients (one launch for what were two
// reductions over N and five (N, C) elementwise ops per layer):
//   gb[c] = sum_n t0,  gns[c] = sum_n t1,  gdinv[n][c] = (t2 - t0*bias[c] - t1*ns[c]) / dinv[n][c]
// with z*dinv = lrelu^-1(y) - b - noise*ns and gy*m(y)*lrelu^-1(y) = gy*y  (t_k = sums[n][k][c]).
__global__ void __launch_bounds__(1024) modconv_bwd_post_k(const float* __restrict__ sums, const float* __restrict__ bias,
                                                           const float* __restrict__ ns, const float* __restrict__ dinv,
                                                           float* __restrict__ gb, float* __restrict__ gns, float* __restrict__ gdinv,
                                                           int N, int C) {
    // block = 64 channels x 16 sample groups (a serial walk over N by one thread per channel took 91 us at N = 256)
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 63, ng = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        const float b = bias ? bias[c] : 0.f, s = ns ? ns[c] : 0.f;
        for (int n = ng; n < N; n += 16) {
            const float* t = sums + (int64_t)n * 3 * C + c;
            const float t0 = t[0], t1 = t[C], t2 = t[2 * C];
            a0 += t0;
            a1 += t1;
            if (gdinv) gdinv[(int64_t)n * C + c] = (t2 - t0 * b - t1 * s) / dinv[(int64_t)n * C + c];
        }
    }
    red[0][ng][cl] = a0;
    red[1][ng][cl] = a1;
    __syncthreads();
    if (ng == 0 && c < C) {
#pragma unroll
        for (int g = 1; g < 16; ++g) {
            a0 += red[0][g][cl];
            a1 += red[1][g][cl];
        }
        if (gb) gb[c] = a0;
        if (gns) gns[c] = a1;
    }
}

extern "C" int gan_modconv_bwd_post(const float* sums, const float* bias, const float* ns, const float* dinv, float* gb, float* gns,
                                    float* gdinv, int N, int C, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && C > 0, "modconv_bwd_post: bad shape");
    GAN_CHECK_ARG(sums || N == 0, "modconv_bwd_post: null pointer");
    GAN_CHECK_ARG(gdinv == nullptr || dinv != nullptr, "modconv_bwd_post: gdinv needs dinv");
    hipLaunchKernelGGL(modconv_bwd_post_k, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, sums, bias, ns, dinv,
                       gb, gns, gdinv, N, C);
    GAN_CHECK_LAUNCH("modconv_bwd_post");
    return GAN_OK;
}

template <typename T>
static int scale_dot_launch(const void* A, const void* B, const float* a, void* out, float* dot, int N, int64_t HW, int C,
                            hipStream_t st) {
    const ChanGrid g = chan_grid<T>(N, HW, C);
    GAN_CHECK_ARG(N <= 65535 && g.zchunks <= 65535, "scale_dot: N or C too large for the grid");
    dim3 grid((unsigned)g.bps, (unsigned)N, (unsigned)g.zchunks);
    if (g.V > 1)
        hipLaunchKernelGGL((scale_dot_k<T, VecOf<T>::V>), grid, dim3(TPB), 0, st, (const T*)A, (const T*)B, a, (T*)out, dot, HW, C, g.CVT,
                           g.R, g.rpb);
    else
        hipLaunchKernelGGL((scale_dot_k<T, 1>), grid, dim3(TPB), 0, st, (const T*)A, (const T*)B, a, (T*)out, dot, HW, C, g.CVT, g.R, g.rpb);
    GAN_CHECK_LAUNCH("scale_dot");
    return GAN_OK;
}

extern "C" int gan_scale_dot(const void* A, const void* B, const float* a, void* out, float* dot, int dot_zeroed, int N, int64_t HW, int C,
                             int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && HW >= 0 && C > 0, "scale_dot: bad shape");
    GAN_CHECK_ARG(dot && a && (N == 0 || HW == 0 || (A && B && out)), "scale_dot: null pointer");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "scale_dot: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(float) * (size_t)N * C;
    if (bytes == 0) return GAN_OK;
    if (!dot_zeroed && hipMemsetAsync(dot, 0, bytes, st) != hipSuccess) {
        gan_set_error("scale_dot: memset failed");
        return GAN_ERR_LAUNCH;
    }
    if (HW == 0) return GAN_OK;
    if (dtype == GAN_F32) return scale_dot_launch<float>(A, B, a, out, dot, N, HW, C, st);
    return scale_dot_launch<bf16>(A, B, a, out, dot, N, HW, C, st);
}

// =========================================================================================
// MinibatchStdDev statistic (reference: modelling/progressive_gan.py:25-36)
// one workgroup per group column j; two-pass variance over the G members in registers.
// =========================================================================================
template <typename T>
__global__ void __launch_bounds__(TPB) mbstd_fwd_k(const T* __restrict__ x, float* __restrict__ stat, int M, int G,
                                                   int64_t E /* HW*C */) {
    __shared__ float red[TPB / 64];
    const int j = blockIdx.x;
    float acc = 0.f;
    for (int64_t e = threadIdx.x; e < E; e += TPB) {
        float mean = 0.f;
        for (int k = 0; k < G; ++k) mean += (float)x[((int64_t)(j + k * M)) * E + e];
        mean /= (float)G;
        float var = 0.f;
        for (int k = 0; k < G; ++k) {
            const float d = (float)x[((int64_t)(j + k * M)) * E + e] - mean;
            var = fmaf(d, d, var);
        }
        acc += sqrtf(var / (float)G + 1e-6f);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < TPB / 64; ++w) s += red[w];
        stat[j] = s / (float)E;
    }
}

// The shape the discriminator has (group of 4, E a multiple of the vector width): every element of the group column is
// read ONCE, as 16-byte vectors, the four members' loads in flight together, 1024 threads per column.  (The loop above
// is 32 trips of eight dependent 2-byte loads per thread at E = 8192: 46 us whatever the batch, three times per step.)
constexpr int MBSTD_TPB = 1024;
template <typename T>
__global__ void __launch_bounds__(MBSTD_TPB) mbstd_fwd4_k(const T* __restrict__ x, float* __restrict__ stat, int M, int64_t E) {
    constexpr int V = VecOf<T>::V;
    __shared__ float red[MBSTD_TPB / 64];
    const int j = blockIdx.x;
    float acc = 0.f;
    for (int64_t e = (int64_t)threadIdx.x * V; e < E; e += (int64_t)MBSTD_TPB * V) {
        float v[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) load_f<T, V>(x + ((int64_t)(j + k * M)) * E + e, v[k]);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float mean = (v[0][i] + v[1][i] + v[2][i] + v[3][i]) * 0.25f;
            float var = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[k][i] - mean;
                var = fmaf(d, d, var);
            }
            acc += sqrtf(var * 0.25f + 1e-6f);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < MBSTD_TPB / 64; ++w) t += red[w];
        stat[j] = t / (float)E;
    }
}

extern "C" int gan_mbstd_fwd(const void* x, float* stat, int N, int G, int64_t HW, int C, int dtype,
                             gan_stream_t stream) {
    GAN_CHECK_ARG(x && stat, "mbstd_fwd: null pointer");
    GAN_CHECK_ARG(G >= 1 && N > 0 && N % G == 0, "mbstd_fwd: batch %d is not a multiple of group size %d", N, G);
    GAN_CHECK_ARG(HW > 0 && C > 0, "mbstd_fwd: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "mbstd_fwd: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int M = N / G;
    const int64_t E = HW * C;
    const bool vec4 = G == 4 && ((uintptr_t)x & 15) == 0 && E % (dtype == GAN_F32 ? 4 : 8) == 0;
    if (vec4 && dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_fwd4_k<float>), dim3(M), dim3(MBSTD_TPB), 0, st, (const float*)x, stat, M, E);
    else if (vec4)
        hipLaunchKernelGGL((mbstd_fwd4_k<bf16>), dim3(M), dim3(MBSTD_TPB), 0, st, (const bf16*)x, stat, M, E);
    else if (dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_fwd_k<float>), dim3(M), dim3(TPB), 0, st, (const float*)x, stat, M, G, E);
    else
        hipLaunchKernelGGL((mbstd_fwd_k<bf16>), dim3(M), dim3(TPB), 0, st, (const bf16*)x, stat, M, G, E);
    GAN_CHECK_LAUNCH("mbstd_fwd");
    return GAN_OK;
}

// Backward of the statistic: gx[j + k*M][e] = gstat[j] * (x_k - mean_e) / (G * std_e * E), with the mean and the
// standard deviation over the G group members recomputed per element (one thread per (group, element)).
template <typename T>
__global__ void __launch_bounds__(TPB) mbstd_bwd_k(const T* __restrict__ x, const float* __restrict__ gstat, T* __restrict__ gx, int M,
                                                   int G, int64_t E) {
    const int j = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    float mean = 0.f;
    for (int k = 0; k < G; ++k) mean += (float)x[((int64_t)(j + k * M)) * E + e];
    mean /= (float)G;
    float var = 0.f;
    for (int k = 0; k < G; ++k) {
        const float d = (float)x[((int64_t)(j + k * M)) * E + e] - mean;
        var = fmaf(d, d, var);
    }
    const float f = gstat[j] / ((float)G * sqrtf(var / (float)G + 1e-6f) * (float)E);
    for (int k = 0; k < G; ++k) {
        const int64_t o = ((int64_t)(j + k * M)) * E + e;
        gx[o] = (T)(((float)x[o] - mean) * f);
    }
}

extern "C" int gan_mbstd_bwd(const void* x, const float* gstat, void* gx, int N, int G, int64_t HW, int C, int dtype,
                             gan_stream_t stream) {
    GAN_CHECK_ARG(x && gstat && gx, "mbstd_bwd: null pointer");
    GAN_CHECK_ARG(G >= 1 && N > 0 && N % G == 0, "mbstd_bwd: batch %d is not a multiple of group size %d", N, G);
    GAN_CHECK_ARG(HW > 0 && C > 0, "mbstd_bwd: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "mbstd_bwd: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int M = N / G;
    const int64_t E = HW * C;
    GAN_CHECK_ARG(M <= 65535 && ceil_div64(E, TPB) < (1LL << 31), "mbstd_bwd: too large for the grid");
    const dim3 grid((unsigned)ceil_div64(E, TPB), (unsigned)M);
    if (dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_bwd_k<float>), grid, dim3(TPB), 0, st, (const float*)x, gstat, (float*)gx, M, G, E);
    else
        hipLaunchKernelGGL((mbstd_bwd_k<bf16>), grid, dim3(TPB), 0, st, (const bf16*)x, gstat, (bf16*)gx, M, G, E);
    GAN_CHECK_LAUNCH("mbstd_bwd");
    return GAN_OK;
}

// Second order (R1 differentiates the statistic twice, losses.py:100-103): with B(x, gstat) the backward above and u
// the cotangent of its output gx, per group column j and element e, over the G members k
//   c = gstat[j] / (G E),  S = sum_k u_k d_k,  ubar = mean_k u_k,  d_k = x_k - mean,  sigma = sqrt(var + 1e-6)
//   d<u, B> / d x_m      = c * ( (u_m - ubar) / sigma  -  S d_m / (G sigma^3) )
//   d<u, B> / d gstat[j] = sum_e S / (G E sigma)
// One workgroup per group column (deterministic block reduction: no atomics, nothing to zero).
template <typename T>
__global__ void __launch_bounds__(TPB) mbstd_bwd2_k(const T* __restrict__ x, const float* __restrict__ gstat, const T* __restrict__ u,
                                                    T* __restrict__ gx2, float* __restrict__ ggstat, int M, int G, int64_t E) {
    __shared__ float red[TPB / 64];
    const int j = blockIdx.x;
    const float invG = 1.f / (float)G;
    const float c = gstat[j] * invG / (float)E;
    float acc = 0.f;
    for (int64_t e = threadIdx.x; e < E; e += TPB) {
        float mean = 0.f, ubar = 0.f;
        for (int k = 0; k < G; ++k) {
            const int64_t o = ((int64_t)(j + k * M)) * E + e;
            mean += (float)x[o];
            ubar += (float)u[o];
        }
        mean *= invG;
        ubar *= invG;
        float var = 0.f, S = 0.f;
        for (int k = 0; k < G; ++k) {
            const int64_t o = ((int64_t)(j + k * M)) * E + e;
            const float d = (float)x[o] - mean;
            var = fmaf(d, d, var);
            S = fmaf((float)u[o], d, S);
        }
        const float sigma = sqrtf(var * invG + 1e-6f);
        const float is = 1.f / sigma;
        acc = fmaf(S, is, acc);
        if (gx2) {
            const float a = c * is, b = c * S * invG * is * is * is;
            for (int k = 0; k < G; ++k) {
                const int64_t o = ((int64_t)(j + k * M)) * E + e;
                gx2[o] = (T)(a * ((float)u[o] - ubar) - b * ((float)x[o] - mean));
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && ggstat) {
        float t = 0.f;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        ggstat[j] = t * invG / (float)E;
    }
}

// group of 4, vector loads: as mbstd_fwd4_k (the scalar loop: 79 us whatever the batch)
template <typename T>
__global__ void __launch_bounds__(MBSTD_TPB) mbstd_bwd2_4_k(const T* __restrict__ x, const float* __restrict__ gstat, const T* __restrict__ u,
                                                            T* __restrict__ gx2, float* __restrict__ ggstat, int M, int64_t E) {
    constexpr int V = VecOf<T>::V;
    __shared__ float red[MBSTD_TPB / 64];
    const int j = blockIdx.x;
    const float c = gstat[j] * 0.25f / (float)E;
    float acc = 0.f;
    for (int64_t e = (int64_t)threadIdx.x * V; e < E; e += (int64_t)MBSTD_TPB * V) {
        float xv[4][V], uv[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            load_f<T, V>(x + ((int64_t)(j + k * M)) * E + e, xv[k]);
            load_f<T, V>(u + ((int64_t)(j + k * M)) * E + e, uv[k]);
        }
        float o[4][V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float mean = (xv[0][i] + xv[1][i] + xv[2][i] + xv[3][i]) * 0.25f;
            const float ubar = (uv[0][i] + uv[1][i] + uv[2][i] + uv[3][i]) * 0.25f;
            float var = 0.f, S = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = xv[k][i] - mean;
                var = fmaf(d, d, var);
                S = fmaf(uv[k][i], d, S);
            }
            const float is = 1.f / sqrtf(var * 0.25f + 1e-6f);
            acc = fmaf(S, is, acc);
            const float a = c * is, b = c * S * 0.25f * is * is * is;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][i] = a * (uv[k][i] - ubar) - b * (xv[k][i] - mean);
        }
        if (gx2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) store_f<T, V>(gx2 + ((int64_t)(j + k * M)) * E + e, o[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && ggstat) {
        float t = 0.f;
        for (int w = 0; w < MBSTD_TPB / 64; ++w) t += red[w];
        ggstat[j] = t * 0.25f / (float)E;
    }
}

extern "C" int gan_mbstd_bwd2(const void* x, const float* gstat, const void* u, void* gx2, float* ggstat, int N, int G, int64_t HW,
                              int C, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(x && gstat && u && (gx2 || ggstat), "mbstd_bwd2: null pointer");
    GAN_CHECK_ARG(G >= 1 && N > 0 && N % G == 0, "mbstd_bwd2: batch %d is not a multiple of group size %d", N, G);
    GAN_CHECK_ARG(HW > 0 && C > 0, "mbstd_bwd2: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "mbstd_bwd2: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int M = N / G;
    const int64_t E = HW * C;
    const bool vec4 = G == 4 && (((uintptr_t)x | (uintptr_t)u | (uintptr_t)gx2) & 15) == 0 && E % (dtype == GAN_F32 ? 4 : 8) == 0;
    if (vec4 && dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_bwd2_4_k<float>), dim3(M), dim3(MBSTD_TPB), 0, st, (const float*)x, gstat, (const float*)u, (float*)gx2, ggstat, M, E);
    else if (vec4)
        hipLaunchKernelGGL((mbstd_bwd2_4_k<bf16>), dim3(M), dim3(MBSTD_TPB), 0, st, (const bf16*)x, gstat, (const bf16*)u, (bf16*)gx2, ggstat, M, E);
    else if (dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_bwd2_k<float>), dim3(M), dim3(TPB), 0, st, (const float*)x, gstat, (const float*)u, (float*)gx2, ggstat, M, G, E);
    else
        hipLaunchKernelGGL((mbstd_bwd2_k<bf16>), dim3(M), dim3(TPB), 0, st, (const bf16*)x, gstat, (const bf16*)u, (bf16*)gx2, ggstat, M, G, E);
    GAN_CHECK_LAUNCH("mbstd_bwd2");
    return GAN_OK;
}

// MinibatchStdDev's output tensor (progressive_gan.py:31-36: the statistic repeated over the group, expanded to a
// plane and concatenated as channel C): out[n][hw][c] = x[n][hw][c] for c < C, stat[n % M] for c == C, in one pass
// (the reference's repeat / view / expand / cat are five launches).  And its adjoint: gx = g[..., :C],
// gstat[j] = sum_k sum_hw g[j + k*M][hw][C].  Both maps are linear, each is the other's adjoint.
template <typename T>
__global__ void __launch_bounds__(TPB) mbstd_concat_k(const T* __restrict__ x, const float* __restrict__ stat, T* __restrict__ out,
                                                      int64_t NP /* N*HW */, int64_t HW, int C, int M) {
    const int Co = C + 1;
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= NP * Co) return;
    const int64_t pix = i / Co;
    const int c = (int)(i - pix * Co);
    out[i] = c < C ? x[pix * C + c] : (T)stat[(pix / HW) % M];
}

template <typename T>
__global__ void __launch_bounds__(TPB) mbstd_concat_adj_k(const T* __restrict__ g, T* __restrict__ gx, float* __restrict__ gstat,
                                                          int64_t NP, int64_t HW, int C, int M, int G, unsigned copy_blocks) {
    const int Co = C + 1;
    if (blockIdx.x < copy_blocks) {
        const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
        if (gx == nullptr || i >= NP * C) return;
        const int64_t pix = i / C;
        gx[i] = g[pix * Co + (i - pix * C)];
        return;
    }
    // one block per group column j: sum of the plane channel over the G members' pixels
    __shared__ float red[TPB / 64];
    const int j = blockIdx.x - copy_blocks;
    float acc = 0.f;
    for (int64_t e = threadIdx.x; e < (int64_t)G * HW; e += TPB) {
        const int k = (int)(e / HW);
        const int64_t hw = e - (int64_t)k * HW;
        acc += (float)g[(((int64_t)(j + k * M)) * HW + hw) * Co + C];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        gstat[j] = t;
    }
}

extern "C" int gan_mbstd_concat(const void* x, const float* stat, void* out, int N, int G, int64_t HW, int C, int dtype,
                                gan_stream_t stream) {
    GAN_CHECK_ARG(x && stat && out, "mbstd_concat: null pointer");
    GAN_CHECK_ARG(G >= 1 && N > 0 && N % G == 0, "mbstd_concat: batch %d is not a multiple of group size %d", N, G);
    GAN_CHECK_ARG(HW > 0 && C > 0, "mbstd_concat: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "mbstd_concat: bad dtype %d", dtype);
    const int64_t NP = (int64_t)N * HW, total = NP * (C + 1);
    const int64_t blocks = ceil_div64(total, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "mbstd_concat: too large for the grid");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_concat_k<float>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const float*)x, stat, (float*)out, NP, HW, C, N / G);
    else
        hipLaunchKernelGGL((mbstd_concat_k<bf16>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const bf16*)x, stat, (bf16*)out, NP, HW, C, N / G);
    GAN_CHECK_LAUNCH("mbstd_concat");
    return GAN_OK;
}

extern "C" int gan_mbstd_concat_adj(const void* g, void* gx, float* gstat, int N, int G, int64_t HW, int C, int dtype,
                                    gan_stream_t stream) {
    GAN_CHECK_ARG(g && gstat, "mbstd_concat_adj: null pointer");
    GAN_CHECK_ARG(G >= 1 && N > 0 && N % G == 0, "mbstd_concat_adj: batch %d is not a multiple of group size %d", N, G);
    GAN_CHECK_ARG(HW > 0 && C > 0, "mbstd_concat_adj: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "mbstd_concat_adj: bad dtype %d", dtype);
    const int64_t NP = (int64_t)N * HW;
    const int M = N / G;
    const int64_t copy_blocks = gx ? ceil_div64(NP * C, TPB) : 0;
    GAN_CHECK_ARG(copy_blocks + M < (1LL << 31), "mbstd_concat_adj: too large for the grid");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32)
        hipLaunchKernelGGL((mbstd_concat_adj_k<float>), dim3((unsigned)(copy_blocks + M)), dim3(TPB), 0, st, (const float*)g, (float*)gx, gstat,
                           NP, HW, C, M, G, (unsigned)copy_blocks);
    else
        hipLaunchKernelGGL((mbstd_concat_adj_k<bf16>), dim3((unsigned)(copy_blocks + M)), dim3(TPB), 0, st, (const bf16*)g, (bf16*)gx, gstat, NP,
                           HW, C, M, G, (unsigned)copy_blocks);
    GAN_CHECK_LAUNCH("mbstd_concat_adj");
    return GAN_OK;
}

// =========================================================================================
// Space-to-depth with zero padding and its adjoint: what turns a stride-2 convolution (DCGAN's 4x4 / stride 2 / pad 1
// layers, modelling/dcgan.py:21-29,50-60) into a stride-1 convolution with no wasted products.
//   s2d: out[n][by][bx][(dy*2+dx)*C + c] = x[n][2by+dy-pad][2bx+dx-pad][c]  (zero outside), out: [N][Hb][Wb][4C]
//   d2s: out[n][iy][ix][c] = in[n][(iy+pad)/2][(ix+pad)/2][((iy+pad)%2*2 + (ix+pad)%2)*C + c],  in: [N][Hb][Wb][4C]
// Both are gathers; each is the other's adjoint.
// =========================================================================================
template <typename T, int V>
__global__ void __launch_bounds__(TPB) s2d_k(const T* __restrict__ x, T* __restrict__ out, int H, int W, int C, int Hb, int Wb, int pad,
                                             int64_t total, int to_depth) {
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= total) return;
    const int cv = C / V;
    if (to_depth) {  // idx over out: [n][by][bx][d][c/V]
        const int c = (int)(idx % cv) * V;
        int64_t r = idx / cv;
        const int d = (int)(r % 4);
        r /= 4;
        const int bx = (int)(r % Wb);
        r /= Wb;
        const int by = (int)(r % Hb);
        const int64_t n = r / Hb;
        const int iy = 2 * by + (d >> 1) - pad, ix = 2 * bx + (d & 1) - pad;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) load_f<T, V>(x + ((n * H + iy) * W + ix) * C + c, v);
        store_f<T, V>(out + idx * V, v);
    } else {  // idx over out: [n][iy][ix][c/V]
        const int c = (int)(idx % cv) * V;
        int64_t r = idx / cv;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H);
        const int64_t n = r / H;
        const int py = iy + pad, px = ix + pad;
        const int d = (py & 1) * 2 + (px & 1);
        float v[V];
        load_f<T, V>(x + (((n * Hb + (py >> 1)) * Wb + (px >> 1)) * 4 + d) * C + c, v);
        store_f<T, V>(out + idx * V, v);
    }
}

extern "C" int gan_space_to_depth(const void* x, void* out, int N, int H, int W, int C, int pad, int to_depth, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0 && C > 0 && pad >= 0, "space_to_depth: bad shape");
    GAN_CHECK_ARG(((H + 2 * pad) & 1) == 0 && ((W + 2 * pad) & 1) == 0, "space_to_depth: padded size %dx%d must be even", H + 2 * pad, W + 2 * pad);
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "space_to_depth: bad dtype %d", dtype);
    if (N == 0) return GAN_OK;
    GAN_CHECK_ARG(x && out, "space_to_depth: null pointer");
    const int Hb = (H + 2 * pad) / 2, Wb = (W + 2 * pad) / 2;
    const int VV = dtype == GAN_F32 ? 4 : 8;
    const bool vec = (C % VV) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const int64_t elems = to_depth ? (int64_t)N * Hb * Wb * 4 * C : (int64_t)N * H * W * C;
    const int64_t total = vec ? elems / VV : elems;
    const int64_t blocks = ceil_div64(total, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "space_to_depth: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32) {
        if (vec) hipLaunchKernelGGL((s2d_k<float, 4>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const float*)x, (float*)out, H, W, C, Hb, Wb, pad, total, to_depth);
        else hipLaunchKernelGGL((s2d_k<float, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const float*)x, (float*)out, H, W, C, Hb, Wb, pad, total, to_depth);
    } else {
        if (vec) hipLaunchKernelGGL((s2d_k<bf16, 8>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const bf16*)x, (bf16*)out, H, W, C, Hb, Wb, pad, total, to_depth);
        else hipLaunchKernelGGL((s2d_k<bf16, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const bf16*)x, (bf16*)out, H, W, C, Hb, Wb, pad, total, to_depth);
    }
    GAN_CHECK_LAUNCH("space_to_depth");
    return GAN_OK;
}

// =========================================================================================
// Convolution parameters scaled per output channel (reference: progressive_gan.py:54,
// `shortcut(x) + layers(x) * scale`): conv(x, W) * scale + b * scale = conv(x, W * scale) + b * scale, so the
// residual merge of a discriminator block needs no pass over the activations -- only these weight-sized maps.
//   forward   ws[r][:] = w[r][:] * s[r],  bs[r] = b[r] * s[r]
//   adjoint   gw[r][:] (+)= gws[r][:] * s[r],  gb[r] (+)= gbs[r] * s[r],  gs[r] (+)= sum_c gws[r][c] * w[r][c] + gbs[r] * b[r]
// One workgroup per row, fixed summation order (deterministic).
// =========================================================================================
__global__ void __launch_bounds__(TPB) scale_rows_k(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ s,
                                                    float* __restrict__ ws, float* __restrict__ bs, int64_t cols) {
    const int r = blockIdx.x;
    const float sr = s[r];
    const float* __restrict__ wr = w + (int64_t)r * cols;
    float* __restrict__ wo = ws + (int64_t)r * cols;
    for (int64_t c = threadIdx.x; c < cols; c += TPB) wo[c] = wr[c] * sr;
    if (threadIdx.x == 0 && bs) bs[r] = b[r] * sr;
}

__global__ void __launch_bounds__(TPB) scale_rows_bwd_k(const float* __restrict__ gws, const float* __restrict__ gbs, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ s, float* __restrict__ gw, int acc_w,
                                                        float* __restrict__ gb, int acc_b, float* __restrict__ gs, int acc_s, int64_t cols) {
    __shared__ float red[TPB / 64];
    const int r = blockIdx.x;
    const float sr = s[r];
    float dot = 0.f;
    if (gws) {
        const float* __restrict__ gr = gws + (int64_t)r * cols;
        const float* __restrict__ wr = w + (int64_t)r * cols;
        float* __restrict__ go = gw ? gw + (int64_t)r * cols : nullptr;
        for (int64_t c = threadIdx.x; c < cols; c += TPB) {
            const float g = gr[c];
            dot = fmaf(g, wr[c], dot);
            if (go) go[c] = acc_w ? fmaf(g, sr, go[c]) : g * sr;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < TPB / 64; ++k) t += red[k];
        const float gbr = gbs ? gbs[r] : 0.f;
        if (gbs && b) t = fmaf(gbr, b[r], t);
        if (gs) gs[r] = acc_s ? gs[r] + t : t;
        if (gb && gbs) gb[r] = acc_b ? fmaf(gbr, sr, gb[r]) : gbr * sr;
    }
}

extern "C" int gan_scale_rows(const float* w, const float* b, const float* s, float* ws, float* bs, int rows, int64_t cols, gan_stream_t stream) {
    GAN_CHECK_ARG(rows >= 0 && cols >= 0, "scale_rows: bad shape");
    if (rows == 0) return GAN_OK;
    GAN_CHECK_ARG(w && s && ws, "scale_rows: null pointer");
    GAN_CHECK_ARG((b == nullptr) == (bs == nullptr), "scale_rows: b and bs go together");
    hipLaunchKernelGGL(scale_rows_k, dim3((unsigned)rows), dim3(TPB), 0, (hipStream_t)stream, w, b, s, ws, bs, cols);
    GAN_CHECK_LAUNCH("scale_rows");
    return GAN_OK;
}

extern "C" int gan_scale_rows_bwd(const float* gws, const float* gbs, const float* w, const float* b, const float* s, float* gw, int acc_w,
                                  float* gb, int acc_b, float* gs, int acc_s, int rows, int64_t cols, gan_stream_t stream) {
    GAN_CHECK_ARG(rows >= 0 && cols >= 0, "scale_rows_bwd: bad shape");
    if (rows == 0) return GAN_OK;
    GAN_CHECK_ARG(s && (gws == nullptr || w != nullptr), "scale_rows_bwd: null pointer");
    GAN_CHECK_ARG(gbs == nullptr || b != nullptr, "scale_rows_bwd: gbs needs b");
    hipLaunchKernelGGL(scale_rows_bwd_k, dim3((unsigned)rows), dim3(TPB), 0, (hipStream_t)stream, gws, gbs, w, b, s, gw, acc_w, gb, acc_b, gs,
                       acc_s, cols);
    GAN_CHECK_LAUNCH("scale_rows_bwd");
    return GAN_OK;
}

// =========================================================================================
// EMA lerp over one flat fp32 arena (reference: utils.py:95-96)
// =========================================================================================
__global__ void __launch_bounds__(TPB) ema_lerp_k(float* __restrict__ ema, const float* __restrict__ p, int64_t n4,
                                                  int64_t n, float w) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n4) {
        float4 e = reinterpret_cast<float4*>(ema)[i];
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        e.x = fmaf(q.x - e.x, w, e.x);
        e.y = fmaf(q.y - e.y, w, e.y);
        e.z = fmaf(q.z - e.z, w, e.z);
        e.w = fmaf(q.w - e.w, w, e.w);
        reinterpret_cast<float4*>(ema)[i] = e;
    } else {
        const int64_t t = n4 * 4 + (i - n4);
        if (t < n) ema[t] = fmaf(p[t] - ema[t], w, ema[t]);
    }
}

extern "C" int gan_ema_lerp(float* ema, const float* p, int64_t n, float weight, gan_stream_t stream) {
    GAN_CHECK_ARG(ema && p, "ema_lerp: null pointer");
    GAN_CHECK_ARG(n >= 0, "ema_lerp: negative size");
    GAN_CHECK_ARG(((uintptr_t)ema % 16) == 0 && ((uintptr_t)p % 16) == 0, "ema_lerp: arenas must be 16-byte aligned");
    if (n == 0) return GAN_OK;
    const int64_t n4 = n / 4;
    const int64_t threads = n4 + (n - n4 * 4);
    const int64_t blocks = ceil_div64(threads, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "ema_lerp: arena too large");
    hipLaunchKernelGGL(ema_lerp_k, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, ema, p, n4, n, weight);
    GAN_CHECK_LAUNCH("ema_lerp");
    return GAN_OK;
}


// ---- weight packing -------------------------------------------------------------------------
// (Cout, Cin, kh, kw) parameter -> GEMM-B image [rows][Kpad], k = tap*Cin_p + ci, zero padded, in the compute
// dtype; `transposed` packs the flipped, channel-swapped weights of the input-gradient correlation
// (rows = Cin, ci <- co, tap <- kh*kw-1-tap).  One launch instead of flip + permute + pad + cast.
template <typename TI, typename TO>
__device__ __forceinline__ void pack_weight_elem(const TI* __restrict__ w, TO* __restrict__ out, int64_t e, int Cin, int taps, int cols,
                                                 int cols_p, int Kpad, int transposed) {
    const int r = (int)(e / Kpad), k = (int)(e - (int64_t)r * Kpad);
    const int tap = k / cols_p, c = k - tap * cols_p;
    float v = 0.f;
    if (tap < taps && c < cols) {
        const int co = transposed ? c : r, ci = transposed ? r : c;
        const int t = transposed ? taps - 1 - tap : tap;
        v = (float)w[((int64_t)co * Cin + ci) * taps + t];
    }
    out[e] = (TO)v;
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) pack_weight_k(const TI* __restrict__ w, TO* __restrict__ out, int Cout, int Cin, int taps,
                                                     int rows, int cols, int cols_p, int Kpad, int transposed) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)rows * Kpad) return;
    pack_weight_elem<TI, TO>(w, out, e, Cin, taps, cols, cols_p, Kpad, transposed);
}

// both orientations of one weight in one launch: [0, n0) -> the forward image, [n0, n0 + n1) -> the transposed one
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) pack_weight_pair_k(const TI* __restrict__ w, TO* __restrict__ out, TO* __restrict__ out_t, int Cout,
                                                          int Cin, int taps, int cols_p, int Kpad, int cols_p_t, int Kpad_t) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n0 = (int64_t)Cout * Kpad, n1 = (int64_t)Cin * Kpad_t;
    if (e < n0)
        pack_weight_elem<TI, TO>(w, out, e, Cin, taps, Cin, cols_p, Kpad, 0);
    else if (e < n0 + n1)
        pack_weight_elem<TI, TO>(w, out_t, e - n0, Cin, taps, Cout, cols_p_t, Kpad_t, 1);
}

extern "C" int gan_pack_weight(const void* w, int w_dtype, void* out, int out_dtype, int Cout, int Cin, int kh, int kw,
                               int cols_p, int Kpad, int transposed, gan_stream_t stream) {
    GAN_CHECK_ARG(w && out, "pack_weight: null pointer");
    GAN_CHECK_ARG((w_dtype == GAN_F32 || w_dtype == GAN_BF16) && (out_dtype == GAN_F32 || out_dtype == GAN_BF16), "pack_weight: bad dtype");
    GAN_CHECK_ARG(Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "pack_weight: bad shape");
    const int rows = transposed ? Cin : Cout, cols = transposed ? Cout : Cin, taps = kh * kw;
    GAN_CHECK_ARG(cols_p >= cols && Kpad >= taps * cols_p, "pack_weight: padded sizes too small");
    const int64_t total = (int64_t)rows * Kpad;
    const int64_t blocks = ceil_div64(total, 256);
    GAN_CHECK_ARG(blocks < (1LL << 31), "pack_weight: too large");
    hipStream_t st = (hipStream_t)stream;
#define PACK_LAUNCH(TI, TO) \
    hipLaunchKernelGGL((pack_weight_k<TI, TO>), dim3((unsigned)blocks), dim3(256), 0, st, (const TI*)w, (TO*)out, Cout, Cin, taps, rows, cols, cols_p, Kpad, transposed)
    if (w_dtype == GAN_F32 && out_dtype == GAN_BF16) PACK_LAUNCH(float, bf16);
    else if (w_dtype == GAN_F32) PACK_LAUNCH(float, float);
    else if (out_dtype == GAN_BF16) PACK_LAUNCH(bf16, bf16);
    else PACK_LAUNCH(bf16, float);
#undef PACK_LAUNCH
    GAN_CHECK_LAUNCH("pack_weight");
    return GAN_OK;
}


extern "C" int gan_pack_weight_pair(const void* w, int w_dtype, void* out, int cols_p, int Kpad, void* out_t, int cols_p_t, int Kpad_t,
                                    int out_dtype, int Cout, int Cin, int kh, int kw, gan_stream_t stream) {
    GAN_CHECK_ARG(w && out && out_t, "pack_weight_pair: null pointer");
    GAN_CHECK_ARG((w_dtype == GAN_F32 || w_dtype == GAN_BF16) && (out_dtype == GAN_F32 || out_dtype == GAN_BF16), "pack_weight_pair: bad dtype");
    GAN_CHECK_ARG(Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "pack_weight_pair: bad shape");
    const int taps = kh * kw;
    GAN_CHECK_ARG(cols_p >= Cin && Kpad >= taps * cols_p && cols_p_t >= Cout && Kpad_t >= taps * cols_p_t, "pack_weight_pair: padded sizes too small");
    const int64_t total = (int64_t)Cout * Kpad + (int64_t)Cin * Kpad_t;
    const int64_t blocks = ceil_div64(total, 256);
    GAN_CHECK_ARG(blocks < (1LL << 31), "pack_weight_pair: too large");
    hipStream_t st = (hipStream_t)stream;
#define PACK2_LAUNCH(TI, TO) \
    hipLaunchKernelGGL((pack_weight_pair_k<TI, TO>), dim3((unsigned)blocks), dim3(256), 0, st, (const TI*)w, (TO*)out, (TO*)out_t, Cout, Cin, taps, cols_p, Kpad, cols_p_t, Kpad_t)
    if (w_dtype == GAN_F32 && out_dtype == GAN_BF16) PACK2_LAUNCH(float, bf16);
    else if (w_dtype == GAN_F32) PACK2_LAUNCH(float, float);
    else if (out_dtype == GAN_BF16) PACK2_LAUNCH(bf16, bf16);
    else PACK2_LAUNCH(bf16, float);
#undef PACK2_LAUNCH
    GAN_CHECK_LAUNCH("pack_weight_pair");
    return GAN_OK;
}

// Every GEMM-B image of a list of weights in ONE launch (an iteration packed ~60 weights with ~60 launches of 5-8 us
// for ~25 us of traffic).  table: `count` rows of 12 int64 in device memory:
//   [0] w  [1] out (forward image or 0)  [2] out_t (transposed image or 0)  [3] wsq (fp32 [Cout][Cin] tap sums of W^2, or 0)
//   [4] first workgroup of this row  [5] Cout  [6] Cin  [7] taps  [8] cols_p  [9] Kpad  [10] cols_p_t  [11] Kpad_t
// A workgroup owns a 64 (co) x 64 (ci) tile of one weight with all its taps: the tile is read ONCE, coalesced (a
// (co, ci) pair is `taps` consecutive elements), the tap sums are taken from the fp32 values, the tile is parked in LDS
// in the output type, and both images are written from there in 16-byte pieces that form 128-byte runs (64 ci of one
// (co, tap) / 64 co of one (ci, tap)).  A row owns ceil(max(Cout, cols_p_t) / 64) * ceil(max(Cin, cols_p) / 64)
// workgroups; dynamic LDS = 4096 * max taps * sizeof(TO).  (First version: a thread per output element gathered its
// source through the strides of the permutation -- 80 us per launch; the transposed image read a new sector per element.)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) pack_weights_multi_k(const long long* __restrict__ table, int count) {
    extern __shared__ __attribute__((aligned(16))) char pw_lds[];
    TO* const tile = reinterpret_cast<TO*>(pw_lds);
    __shared__ int which;
    for (int i = threadIdx.x; i < count; i += 256) {
        const long long b0 = table[i * 12 + 4];
        const long long b1 = i + 1 < count ? table[(i + 1) * 12 + 4] : (long long)gridDim.x;
        if ((long long)blockIdx.x >= b0 && (long long)blockIdx.x < b1) which = i;
    }
    __syncthreads();
    const long long* __restrict__ d = table + which * 12;
    const TI* __restrict__ w = reinterpret_cast<const TI*>(d[0]);
    TO* __restrict__ out = reinterpret_cast<TO*>(d[1]);
    TO* __restrict__ out_t = reinterpret_cast<TO*>(d[2]);
    float* __restrict__ wsq = reinterpret_cast<float*>(d[3]);
    const int Cout = (int)d[5], Cin = (int)d[6], taps = (int)d[7];
    const int cols_p = (int)d[8], Kpad = (int)d[9], cols_p_t = (int)d[10], Kpad_t = (int)d[11];
    const int ci_span = cols_p > Cin ? cols_p : Cin, ci_tiles = (ci_span + 63) >> 6;
    const int local = (int)((long long)blockIdx.x - d[4]);
    const int co0 = (local / ci_tiles) * 64, ci0 = (local % ci_tiles) * 64;
    const int tid = threadIdx.x;
    // ---- stage: pair p = (co, ci) -> `taps` consecutive source elements; zeros outside the weight
    for (int p = tid; p < 4096; p += 256) {
        const int co = co0 + (p >> 6), ci = ci0 + (p & 63);
        TO* __restrict__ dst = tile + p * taps;
        if (co < Cout && ci < Cin) {
            const TI* __restrict__ src = w + ((int64_t)co * Cin + ci) * taps;
            float a = 0.f;
            for (int t = 0; t < taps; ++t) {
                const float v = (float)src[t];
                a = fmaf(v, v, a);
                dst[t] = (TO)v;
            }
            if (wsq) wsq[(int64_t)co * Cin + ci] = a;
        } else {
            for (int t = 0; t < taps; ++t) dst[t] = (TO)0.f;
        }
    }
    __syncthreads();
    constexpr int PW = 16 / (int)sizeof(TO);  // elements per 16-byte piece (8 bf16 / 4 fp32)
    constexpr int PPR = 64 / PW;              // pieces per 64-channel run
    const int npieces = 64 * taps * PPR;
    // ---- forward image: out[co][tap * cols_p + ci], rows co < Cout, columns < cols_p
    if (out) {
        for (int e = tid; e < npieces; e += 256) {
            const int pc = e % PPR, rest = e / PPR, tap = rest % taps, r = rest / taps;
            const int co = co0 + r, c = ci0 + pc * PW;
            if (co >= Cout || c >= cols_p) continue;
            TO v[PW];
#pragma unroll
            for (int j = 0; j < PW; ++j) v[j] = tile[(r * 64 + pc * PW + j) * taps + tap];
            TO* __restrict__ dst = out + (int64_t)co * Kpad + tap * cols_p + c;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
        }
        if (ci0 == 0) {  // the zero tail of each row: k in [taps * cols_p, Kpad)
            const int tail = (Kpad - taps * cols_p) / PW;
            for (int e = tid; e < 64 * tail; e += 256) {
                const int co = co0 + e / tail;
                if (co < Cout) *reinterpret_cast<uint4*>(out + (int64_t)co * Kpad + taps * cols_p + (e % tail) * PW) = make_uint4(0, 0, 0, 0);
            }
        }
    }
    // ---- transposed image: out_t[ci][(taps - 1 - tap) * cols_p_t + co], rows ci < Cin, columns < cols_p_t
    if (out_t) {
        for (int e = tid; e < npieces; e += 256) {
            const int pc = e % PPR, rest = e / PPR, tap = rest % taps, r = rest / taps;
            const int ci = ci0 + r, c = co0 + pc * PW;
            if (ci >= Cin || c >= cols_p_t) continue;
            TO v[PW];
#pragma unroll
            for (int j = 0; j < PW; ++j) v[j] = tile[((pc * PW + j) * 64 + r) * taps + tap];
            TO* __restrict__ dst = out_t + (int64_t)ci * Kpad_t + (taps - 1 - tap) * cols_p_t + c;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
        }
        if (co0 == 0) {
            const int tail = (Kpad_t - taps * cols_p_t) / PW;
            for (int e = tid; e < 64 * tail; e += 256) {
                const int ci = ci0 + e / tail;
                if (ci < Cin) *reinterpret_cast<uint4*>(out_t + (int64_t)ci * Kpad_t + taps * cols_p_t + (e % tail) * PW) = make_uint4(0, 0, 0, 0);
            }
        }
    }
}

extern "C" int gan_pack_weights_multi(const int64_t* table, int count, int64_t blocks, int max_taps, int w_dtype, int out_dtype,
                                      gan_stream_t stream) {
    GAN_CHECK_ARG(count >= 0 && blocks >= 0 && blocks < (1LL << 31), "pack_weights_multi: bad sizes");
    GAN_CHECK_ARG((w_dtype == GAN_F32 || w_dtype == GAN_BF16) && (out_dtype == GAN_F32 || out_dtype == GAN_BF16), "pack_weights_multi: bad dtype");
    if (count == 0 || blocks == 0) return GAN_OK;
    GAN_CHECK_ARG(table, "pack_weights_multi: null table");
    const size_t lds = (size_t)4096 * (size_t)(max_taps > 0 ? max_taps : 0) * (out_dtype == GAN_BF16 ? 2 : 4);
    GAN_CHECK_ARG(max_taps >= 1 && lds <= 150 * 1024, "pack_weights_multi: a 64 x 64 tile with %d taps does not fit LDS (pack that weight alone)", max_taps);
    hipStream_t st = (hipStream_t)stream;
    // more than 64 KB of dynamic LDS needs the attribute (set on every call: it is per device, and cheap)
#define PACKM_LAUNCH(TI, TO)                                                                                                          \
    do {                                                                                                                              \
        if (lds > 64 * 1024 &&                                                                                                        \
            hipFuncSetAttribute((const void*)pack_weights_multi_k<TI, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            gan_set_error("pack_weights_multi: cannot reserve %zu bytes of LDS", lds);                                                \
            return GAN_ERR_LAUNCH;                                                                                                    \
        }                                                                                                                             \
        hipLaunchKernelGGL((pack_weights_multi_k<TI, TO>), dim3((unsigned)blocks), dim3(256), lds, st, (const long long*)table, count); \
    } while (0)
    if (w_dtype == GAN_F32 && out_dtype == GAN_BF16) PACKM_LAUNCH(float, bf16);
    else if (w_dtype == GAN_F32) PACKM_LAUNCH(float, float);
    else if (out_dtype == GAN_BF16) PACKM_LAUNCH(bf16, bf16);
    else PACKM_LAUNCH(bf16, float);
#undef PACKM_LAUNCH
    GAN_CHECK_LAUNCH("pack_weights_multi");
    return GAN_OK;
}

// ---- RGB skip of the generator: out = nearest_up2(lo) + hi  (stylegan2.py:99) ---------------------
// One pass instead of an upsampling pass that materialises the 4x tensor and an add pass.  Rows are treated as
// flat runs of 2W*C elements, V per thread (16 bytes of hi / out); the matching elements of the half-resolution
// row are a gather of V scalars from a row that stays in cache.
template <typename T, int V>
__global__ void __launch_bounds__(TPB) up2_add_k(const T* __restrict__ lo, const T* __restrict__ hi, T* __restrict__ out, int W, int C,
                                                 int64_t rows /* N * 2H */, int H2) {
    const int rowlen = 2 * W * C;
    const int per_row = rowlen / V;
    const int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (t >= rows * per_row) return;
    const int64_t row = t / per_row;
    const int e0 = (int)(t - row * per_row) * V;
    const int64_t n = row / H2;
    const int y = (int)(row - n * H2);
    const T* __restrict__ lrow = lo + ((n * (H2 >> 1) + (y >> 1)) * W) * (int64_t)C;
    float h[V];
    load_f<T, V>(hi + row * rowlen + e0, h);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int e = e0 + j;
        const int x = e / C, c = e - x * C;
        h[j] += (float)lrow[(x >> 1) * C + c];
    }
    store_f<T, V>(out + row * rowlen + e0, h);
}

extern "C" int gan_up2_add(const void* lo, const void* hi, void* out, int N, int H, int W, int C, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(lo && hi && out, "up2_add: null pointer");
    GAN_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0, "up2_add: bad shape");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "up2_add: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)N * 2 * H;
    const int rowlen = 2 * W * C;
    const int VV = dtype == GAN_F32 ? VecOf<float>::V : VecOf<bf16>::V;
    const bool vec = (rowlen % VV) == 0 && ((uintptr_t)hi % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const int64_t threads = rows * (vec ? rowlen / VV : rowlen);
    const int64_t blocks = ceil_div64(threads, TPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "up2_add: tensor too large");
    if (dtype == GAN_F32) {
        if (vec) hipLaunchKernelGGL((up2_add_k<float, VecOf<float>::V>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const float*)lo, (const float*)hi, (float*)out, W, C, rows, 2 * H);
        else hipLaunchKernelGGL((up2_add_k<float, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const float*)lo, (const float*)hi, (float*)out, W, C, rows, 2 * H);
    } else {
        if (vec) hipLaunchKernelGGL((up2_add_k<bf16, VecOf<bf16>::V>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const bf16*)lo, (const bf16*)hi, (bf16*)out, W, C, rows, 2 * H);
        else hipLaunchKernelGGL((up2_add_k<bf16, 1>), dim3((unsigned)blocks), dim3(TPB), 0, st, (const bf16*)lo, (const bf16*)hi, (bf16*)out, W, C, rows, 2 * H);
    }
    GAN_CHECK_LAUNCH("up2_add");
    return GAN_OK;
}

// ---- image-like tensors (C < 8: RGB) -> channels-last, channel-padded, compute dtype ------------
// out[n][h][w][0..Cp) = x[n][c][h][w] for c < C, zeros above; x has arbitrary element strides (NCHW or
// channels-last, fp32 or bf16).  One pass instead of cast + layout copy + zero fill + strided copy (the
// from-RGB / to-RGB tensors are the largest in the model: 0.5 GB padded at 256x256, batch 256).
// Block = one image row; a thread writes one 8-channel chunk of a pixel (16 B in bf16).
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) pad_channels_k(const TI* __restrict__ x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                                      TO* __restrict__ out, int H, int W, int C, int Cp) {
    const int row = blockIdx.x;  // n * H + h
    const int n = row / H, h = row - n * H;
    const TI* __restrict__ src = x + (int64_t)n * sn + (int64_t)h * sh;
    TO* __restrict__ dst = out + (int64_t)row * W * Cp;
    const int chunks = Cp >> 3;
    for (int e = threadIdx.x; e < W * chunks; e += 256) {
        const int w = e / chunks, c0 = (e - w * chunks) * 8;
        TO v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (TO)(c0 + j < C ? (float)src[(int64_t)w * sw + (int64_t)(c0 + j) * sc] : 0.f);
        TO* o = dst + (int64_t)e * 8;
        if (sizeof(TO) == 2) {
            *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(v);
        } else {
            *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(v);
            *reinterpret_cast<uint4*>(o + 4) = *reinterpret_cast<const uint4*>(v + 4);
        }
    }
}

extern "C" int gan_pad_channels(const void* x, int x_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, void* out, int out_dtype,
                                int N, int C, int H, int W, int Cp, gan_stream_t stream) {
    GAN_CHECK_ARG(x && out, "pad_channels: null pointer");
    GAN_CHECK_ARG((x_dtype == GAN_F32 || x_dtype == GAN_BF16) && (out_dtype == GAN_F32 || out_dtype == GAN_BF16), "pad_channels: bad dtype");
    GAN_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && (Cp & 7) == 0, "pad_channels: bad shape (Cp must be a multiple of 8, >= C)");
    GAN_CHECK_ARG((int64_t)N * H < (1LL << 31) && (int64_t)W * (Cp >> 3) < (1LL << 31), "pad_channels: too large");
    GAN_CHECK_ARG(((uintptr_t)out % 16) == 0, "pad_channels: output must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
#define PADC_LAUNCH(TI, TO) \
    hipLaunchKernelGGL((pad_channels_k<TI, TO>), dim3((unsigned)(N * H)), dim3(256), 0, st, (const TI*)x, sn, sc, sh, sw, (TO*)out, H, W, C, Cp)
    if (x_dtype == GAN_F32 && out_dtype == GAN_BF16) PADC_LAUNCH(float, bf16);
    else if (x_dtype == GAN_F32) PADC_LAUNCH(float, float);
    else if (out_dtype == GAN_BF16) PADC_LAUNCH(bf16, bf16);
    else PADC_LAUNCH(bf16, float);
#undef PADC_LAUNCH
    GAN_CHECK_LAUNCH("pad_channels");
    return GAN_OK;
}


// =========================================================================================
// Demodulation factor of ModulatedConv2d (reference: F.normalize(weight * s, dim=(2,3,4), eps=1e-8),
// modelling/stylegan2.py:33) without the (N, Cout, Cin, k, k) tensor:
//   wsq[co][ci] = sum_taps W^2,   dinv[n][co] = 1 / max(sqrt(sum_ci s[n,ci]^2 * wsq[co,ci]), 1e-8)
// and its backward; all fp32, a few hundred KB: one launch each instead of ~8 / ~14 tensor ops per layer.
// =========================================================================================
__global__ void __launch_bounds__(TPB) wsq_k(const float* __restrict__ w, float* __restrict__ wsq, int64_t n, int taps) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= n) return;
    float a = 0.f;
    for (int t = 0; t < taps; ++t) {
        const float v = w[e * taps + t];
        a = fmaf(v, v, a);
    }
    wsq[e] = a;
}

__device__ __forceinline__ float demod_dq(float g, float d) { return d < 1e8f ? -0.5f * d * d * d * g : 0.f; }

// The three products of the demodulation factor are small fp32 GEMMs (M, Nn, K out of {N, Cin, Cout}) with an
// elementwise map on the way in and out:
//   MODE 0  dinv[n][co] = 1 / max(sqrt(sum_ci s[n,ci]^2 wsq[co,ci]), 1e-8)            M = N,    Nn = Cout, K = Cin
//   MODE 1  gs[n][ci]   = 2 s[n,ci] sum_co dq[n,co] wsq[co,ci]                         M = N,    Nn = Cin,  K = Cout
//   MODE 2  gw[co][ci][t] (+)= 2 W[co,ci,t] sum_n dq[n,co] s[n,ci]^2                   M = Cout, Nn = Cin,  K = N
// One kernel on the exact fp32 MFMA (v_mfma_f32_16x16x4_f32): a workgroup owns a 16 x 16 tile of the result, its four
// waves split K; a lane loads the operand values of its (row, k) / (k, column) slots straight from global memory --
// up to 128 k per wave in flight at once, no LDS staging, no barrier in the loop -- and the four partial tiles are
// summed through LDS in a fixed order.  History: one wave per output re-read the second operand from L2 once per
// output row (268 MB per call at batch 256: 20-24 us); an LDS-tiled VALU version had 8 workgroups of work at batch 32
// and a load -> barrier -> compute chain per K chunk (24 us).  This one: M/16 x Nn/16 workgroups, one load round trip.
struct DemodArgs {
    const float* g;
    const float* dinv;
    const float* s;
    const float* w;
    const float* wsq;
    float* out;
    int N, Cin, Cout, taps, accum;
    int lds, ldd;  // row pitch of s / gs, and of g / dinv (the layer's columns of a matrix that holds every layer's: demod_multi_k)
};

template <int MODE>
__device__ __forceinline__ void demod_tile(const DemodArgs& p, const int bx, const int by) {
    __shared__ float red[4][256];
    __shared__ float tile[256];
    const int M = MODE == 2 ? p.Cout : p.N, Nn = MODE == 0 ? p.Cout : p.Cin, K = MODE == 0 ? p.Cin : (MODE == 1 ? p.Cout : p.N);
    const int m0 = by * 16, n0 = bx * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int kq = ((((K + 3) >> 2) + 15) >> 4) << 4;  // K range of a wave, a multiple of 16
    const int kb = wave * kq, ke = min(K, kb + kq);
    const int m = m0 + i, n = n0 + i;
    const bool mok = m < M, nok = n < Nn;
    // rows that are contiguous in k can be read 16 bytes at a time
    const bool vecA = MODE == 0 ? (p.lds & 3) == 0 && ((uintptr_t)p.s & 15) == 0
                                : (MODE == 1 ? (p.ldd & 3) == 0 && (((uintptr_t)p.g | (uintptr_t)p.dinv) & 15) == 0 : false);
    const bool vecB = MODE == 0 ? (p.Cin & 3) == 0 && ((uintptr_t)p.wsq & 15) == 0 : false;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kb; k0 < ke; k0 += 128) {
        float a[32], b[32];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = k0 + 16 * t + 4 * g;  // this lane's four k of group t
            // ---- A(m, k..k+3)
            if (MODE == 0 || MODE == 1) {
                const int ld = MODE == 0 ? p.lds : p.ldd;
                const int64_t off = (int64_t)m * ld + k;
                float x0[4] = {0.f, 0.f, 0.f, 0.f}, x1[4] = {1.f, 1.f, 1.f, 1.f};
                if (mok && vecA && k + 3 < ke) {
                    const float4 v = *reinterpret_cast<const float4*>((MODE == 0 ? p.s : p.g) + off);
                    x0[0] = v.x; x0[1] = v.y; x0[2] = v.z; x0[3] = v.w;
                    if (MODE == 1) {
                        const float4 d = *reinterpret_cast<const float4*>(p.dinv + off);
                        x1[0] = d.x; x1[1] = d.y; x1[2] = d.z; x1[3] = d.w;
                    }
                } else if (mok) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < ke) {
                            x0[j] = (MODE == 0 ? p.s : p.g)[off + j];
                            if (MODE == 1) x1[j] = p.dinv[off + j];
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) a[4 * t + j] = MODE == 0 ? x0[j] * x0[j] : demod_dq(x0[j], x1[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
                    if (mok && k + j < ke) v = demod_dq(p.g[(int64_t)(k + j) * p.ldd + m], p.dinv[(int64_t)(k + j) * p.ldd + m]);
                    a[4 * t + j] = v;
                }
            }
            // ---- B(k..k+3, n)
            if (MODE == 0) {
                const int64_t off = (int64_t)n * p.Cin + k;
                if (nok && vecB && k + 3 < ke) {
                    const float4 v = *reinterpret_cast<const float4*>(p.wsq + off);
                    b[4 * t] = v.x; b[4 * t + 1] = v.y; b[4 * t + 2] = v.z; b[4 * t + 3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[4 * t + j] = nok && k + j < ke ? p.wsq[off + j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
                    if (nok && k + j < ke) {
                        v = MODE == 1 ? p.wsq[(int64_t)(k + j) * p.Cin + n] : p.s[(int64_t)(k + j) * p.lds + n];
                        if (MODE == 2) v *= v;
                    }
                    b[4 * t + j] = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * g + r) * 16 + i] = c[r];
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const float v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    const int mo = m0 + row, no = n0 + col;
    if (MODE == 0) {
        if (mo < M && no < Nn) p.out[(int64_t)mo * p.ldd + no] = 1.f / fmaxf(sqrtf(v), 1e-8f);
    } else if (MODE == 1) {
        if (mo < M && no < Nn) p.out[(int64_t)mo * p.lds + no] = 2.f * p.s[(int64_t)mo * p.lds + no] * v;
    } else {
        tile[tid] = v;
        __syncthreads();
        if (mo >= M) return;
        // a row of the tile is min(16, Cin - n0) x taps consecutive floats of W / gw
        const int run = min(16, Nn - n0) * p.taps;
        const int64_t base = ((int64_t)mo * p.Cin + n0) * p.taps;
        for (int e = col; e < run; e += 16) {
            const float r = 2.f * p.w[base + e] * tile[row * 16 + e / p.taps];
            p.out[base + e] = p.accum ? p.out[base + e] + r : r;
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) demod_mfma_k(const DemodArgs p) {
    demod_tile<MODE>(p, blockIdx.x, blockIdx.y);
}

// Every demodulating layer of a generator in ONE launch per product: the layers' styles are the column blocks of one
// (N, sum Cin) matrix (they come out of one GEMM, stylegan2.py:30 for every layer at once) and their factors / factor
// gradients the column blocks of one (N, sum Cout) matrix.  `table`: device memory, `count` rows of 10 int64:
// [0] wsq, [1] w, [2] gw (MODE 2), [3] Cin, [4] Cout, [5] taps, [6] first style column, [7] first factor column,
// [8] first workgroup of the row, [9] accumulate into gw.  (13 layers: 26 + 26 launches of 5-14 us per iteration, the same
// at any batch, become 2 + 2.)
constexpr int DEMOD_ROW = 10;
template <int MODE>
__global__ void __launch_bounds__(256) demod_multi_k(const long long* __restrict__ table, int count, const float* __restrict__ s_all, int lds,
                                                     const float* __restrict__ g_all, const float* __restrict__ dinv_all, int ldd,
                                                     float* __restrict__ out_all, int N) {
    __shared__ int which;
    for (int i = threadIdx.x; i < count; i += 256) {
        const long long b0 = table[i * DEMOD_ROW + 8];
        const long long b1 = i + 1 < count ? table[(i + 1) * DEMOD_ROW + 8] : (long long)gridDim.x;
        if ((long long)blockIdx.x >= b0 && (long long)blockIdx.x < b1) which = i;
    }
    __syncthreads();
    const long long* __restrict__ d = table + which * DEMOD_ROW;
    DemodArgs p;
    p.Cin = (int)d[3]; p.Cout = (int)d[4]; p.taps = (int)d[5]; p.N = N; p.accum = (int)d[9];
    p.lds = lds; p.ldd = ldd;
    const int soff = (int)d[6], doff = (int)d[7];
    p.wsq = reinterpret_cast<const float*>(d[0]);
    p.w = reinterpret_cast<const float*>(d[1]);
    p.s = s_all + soff;
    p.g = g_all ? g_all + doff : nullptr;
    p.dinv = dinv_all ? dinv_all + doff : nullptr;
    p.out = MODE == 0 ? out_all + doff : (MODE == 1 ? out_all + soff : reinterpret_cast<float*>(d[2]));
    const int Nn = MODE == 0 ? p.Cout : p.Cin;
    const int tx = (Nn + 15) / 16;
    const int local = (int)((long long)blockIdx.x - d[8]);
    demod_tile<MODE>(p, local % tx, local / tx);
}

template <int MODE>
static void demod_launch(const DemodArgs& a, hipStream_t st) {
    const int M = MODE == 2 ? a.Cout : a.N, Nn = MODE == 0 ? a.Cout : a.Cin;
    hipLaunchKernelGGL(demod_mfma_k<MODE>, dim3((unsigned)((Nn + 15) / 16), (unsigned)((M + 15) / 16)), dim3(256), 0, st, a);
}

extern "C" int gan_demod_fwd(const float* s, const float* w, float* wsq, float* dinv, int N, int Cin, int Cout, int taps,
                             gan_stream_t stream) {
    GAN_CHECK_ARG(s && wsq && dinv, "demod_fwd: null pointer");
    GAN_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && taps > 0 && (N + 15) / 16 <= 65535, "demod_fwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (w) {  // w == NULL: wsq already holds the tap sums of this weight (gan_pack_weights_multi writes them)
        const int64_t ne = (int64_t)Cout * Cin;
        hipLaunchKernelGGL(wsq_k, dim3((unsigned)ceil_div64(ne, TPB)), dim3(TPB), 0, st, w, wsq, ne, taps);
    }
    DemodArgs a{nullptr, nullptr, s, nullptr, wsq, dinv, N, Cin, Cout, taps, 0, Cin, Cout};
    demod_launch<0>(a, st);
    GAN_CHECK_LAUNCH("demod_fwd");
    return GAN_OK;
}

extern "C" int gan_demod_bwd(const float* g, const float* dinv, const float* s, const float* w, const float* wsq, float* gs,
                             float* gw, int gw_accumulate, int N, int Cin, int Cout, int taps, gan_stream_t stream) {
    GAN_CHECK_ARG(g && dinv && s && w && wsq, "demod_bwd: null pointer");
    GAN_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && taps > 0 && (N + 15) / 16 <= 65535 && (Cout + 15) / 16 <= 65535, "demod_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (gs) {
        DemodArgs a{g, dinv, s, w, wsq, gs, N, Cin, Cout, taps, 0, Cin, Cout};
        demod_launch<1>(a, st);
    }
    if (gw) {
        DemodArgs a{g, dinv, s, w, wsq, gw, N, Cin, Cout, taps, gw_accumulate, Cin, Cout};
        demod_launch<2>(a, st);
    }
    GAN_CHECK_LAUNCH("demod_bwd");
    return GAN_OK;
}

// mode 0: dinv_all (N, ldd) from s_all (N, lds); 1: gs_all (N, lds) from (g_all, dinv_all, s_all); 2: every row's gw.
// `blocks`: the table's total workgroup count for this mode (the caller laid out column [8] for it).
extern "C" int gan_demod_multi(int mode, const void* table, int count, int64_t blocks, const float* s_all, int lds, const float* g_all,
                               const float* dinv_all, int ldd, float* out_all, int N, gan_stream_t stream) {
    GAN_CHECK_ARG(mode >= 0 && mode <= 2 && table && count > 0 && blocks > 0 && blocks < (1LL << 31), "demod_multi: bad arguments");
    GAN_CHECK_ARG(s_all && N > 0 && lds > 0 && ldd > 0, "demod_multi: bad shape");
    GAN_CHECK_ARG(mode == 0 ? out_all != nullptr : (g_all && dinv_all && (mode == 2 || out_all)), "demod_multi: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long long* t = (const long long*)table;
    if (mode == 0) hipLaunchKernelGGL(demod_multi_k<0>, dim3((unsigned)blocks), dim3(256), 0, st, t, count, s_all, lds, g_all, dinv_all, ldd, out_all, N);
    else if (mode == 1) hipLaunchKernelGGL(demod_multi_k<1>, dim3((unsigned)blocks), dim3(256), 0, st, t, count, s_all, lds, g_all, dinv_all, ldd, out_all, N);
    else hipLaunchKernelGGL(demod_multi_k<2>, dim3((unsigned)blocks), dim3(256), 0, st, t, count, s_all, lds, g_all, dinv_all, ldd, out_all, N);
    GAN_CHECK_LAUNCH("demod_multi");
    return GAN_OK;
}
