// Implicit-GEMM convolution on MFMA (gfx950), channels-last, stride 1.
//
//   GEMM view:  M = N*Ho*Wo output pixels,  N = Cout,  K = kh*kw*Cin  (k = tap*Cin + ci)
//   A[m][k]  = X[n, oy+dy-py, ox+dx-px, ci] * in_scale[n][ci]      gathered on the fly (zero outside)
//   B[co][k] = Wt[co][k]                                           K-contiguous weights
//   D[m][co] = act(out_scale[n][co] * sum_k A*B + bias[co] + noise[m]*noise_scale[co])
//
// One workgroup = 4 waves = a BM x BN tile of D.  Per 32-deep K step both operand tiles are
// staged global -> registers -> LDS (double-buffered LDS, one barrier per step, next step's
// global loads issued before the current step's MFMAs) and consumed as 16x16x32 bf16 MFMA
// fragments (fp32: eight 16x16x4 fp32 MFMAs per fragment, exact fp32 products).  Each lane's
// fragment is 8 consecutive k of one row, so an LDS row (32 k) is read with one 16-byte
// ds_read per lane; rows are XOR-swizzled at 16-byte granularity to spread the four k-chunks
// of a 16-row fragment over all LDS banks.
#include "common.h"

namespace {

struct ConvArgs {
    const void* x;
    const void* wt;
    void* y;
    const float* in_scale;
    const float* out_scale;
    const float* bias;
    const void* noise;
    const float* noise_scale;
    float slope;
    int N, H, W, Cin;  // physical input
    int Hl, Wl;        // logical input (2H, 2W when up2)
    int up2;
    int Ho, Wo, Cout;
    int kh, kw, py, px;
    int Kpad, ksteps;
    int64_t M;
    ConvFuse fu;
};

template <typename T>
struct Mma;

template <>
struct Mma<bf16> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ Frag load(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <>
struct Mma<float> {
    struct Frag {
        float4 lo, hi;
    };
    static __device__ __forceinline__ Frag load(const char* p) {
        Frag f;
        f.lo = *reinterpret_cast<const float4*>(p);
        f.hi = *reinterpret_cast<const float4*>(p + 16);
        return f;
    }
    // Lane group q = lane>>4 holds k = 8q..8q+7 of the 32-deep step; MFMA j sums element j of
    // the four groups.  A and B use the same assignment, so the k order is immaterial.
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, c, 0, 0, 0);
    }
};

// byte offset of 16-byte chunk `ch` of LDS row `row`
template <typename T>
__device__ __forceinline__ int lds_off(int row, int ch);
template <>
__device__ __forceinline__ int lds_off<bf16>(int row, int ch) {
    return row * 64 + ((ch ^ ((-(row >> 2)) & 3)) << 4);
}
template <>
__device__ __forceinline__ int lds_off<float>(int row, int ch) {
    // keep the two chunks of a lane's 32-byte fragment adjacent; rotate chunk pairs by row
    return row * 128 + ((((ch >> 1) ^ ((row >> 1) & 3)) << 5) | ((ch & 1) << 4));
}

__device__ __forceinline__ uint4 scale_chunk(uint4 raw, const float* s, bf16) {
    bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
    const float4 s0 = *reinterpret_cast<const float4*>(s);
    const float4 s1 = *reinterpret_cast<const float4*>(s + 4);
    v[0] = (bf16)((float)v[0] * s0.x);
    v[1] = (bf16)((float)v[1] * s0.y);
    v[2] = (bf16)((float)v[2] * s0.z);
    v[3] = (bf16)((float)v[3] * s0.w);
    v[4] = (bf16)((float)v[4] * s1.x);
    v[5] = (bf16)((float)v[5] * s1.y);
    v[6] = (bf16)((float)v[6] * s1.z);
    v[7] = (bf16)((float)v[7] * s1.w);
    return *reinterpret_cast<uint4*>(&v);
}
__device__ __forceinline__ uint4 scale_chunk(uint4 raw, const float* s, float) {
    float4 v = *reinterpret_cast<float4*>(&raw);
    const float4 s0 = *reinterpret_cast<const float4*>(s);
    v.x *= s0.x;
    v.y *= s0.y;
    v.z *= s0.z;
    v.w *= s0.w;
    return *reinterpret_cast<uint4*>(&v);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) conv_fwd_k(const ConvArgs p) {
    constexpr int ROWB = 32 * (int)sizeof(T);  // bytes per LDS row (32 k)
    constexpr int CH = ROWB / 16;               // 16-byte chunks per row
    constexpr int EPC = 16 / (int)sizeof(T);    // elements per chunk
    constexpr int NA = (BM * CH + 255) / 256;   // A chunks staged per thread
    constexpr int NB = (BN * CH + 255) / 256;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    typedef typename Mma<T>::Frag Frag;

    __shared__ __attribute__((aligned(16))) char lds[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int64_t bm0 = (int64_t)blockIdx.x * BM;
    const int bn0 = blockIdx.y * BN;

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ wt = (const T*)p.wt;
    const int HoWo = p.Ho * p.Wo;
    const int ntaps = p.kh * p.kw;

    // ---- per-thread staging bookkeeping -------------------------------------------------
    const int ch = tid % CH;  // same chunk column for every row this thread stages
    int a_n[NA], a_oy[NA], a_ox[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (tid + i * 256) / CH;
        const int64_t m = bm0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        const int64_t mm = a_ok[i] ? m : 0;
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (int64_t)n * HoWo);
        a_n[i] = n;
        if (p.fu.pool) {
            // pooled launches order the GEMM rows quad-major: rows 4q..4q+3 are the 2x2 window of pooled
            // pixel q, so the window ends up in ONE lane's four accumulator registers
            const int q = rem >> 2, wo2 = p.Wo >> 1;
            const int qy = q / wo2;
            a_oy[i] = 2 * qy + ((rem >> 1) & 1);
            a_ox[i] = 2 * (q - qy * wo2) + (rem & 1);
        } else {
            a_oy[i] = rem / p.Wo;
            a_ox[i] = rem - a_oy[i] * p.Wo;
        }
    }
    // running (tap, ci) of this thread's chunk; advances by 32 elements per K step
    int kelem = ch * EPC;
    int tap = kelem / p.Cin;
    int ci = kelem - tap * p.Cin;
    int dy = tap / p.kw;
    int dx = tap - dy * p.kw;

    // Staging loads are unconditional (ld16_or_zero) and the modulation vectors are fetched beside them;
    // the multiply happens when the tile is written to LDS one step later.  Loads under `if (ok)` were
    // issued one branch + wait at a time.
    uint4 a_reg[NA], b_reg[NB];
    constexpr int SPC = EPC / 4;  // float4 scale vectors per 16-byte chunk (2 for bf16, 1 for fp32)
    float4 a_sc[NA][SPC];

    auto load_step = [&](int s) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int iy = a_oy[i] + dy - p.py;
            const int ix = a_ox[i] + dx - p.px;
            const bool ok = a_ok[i] && tap < ntaps && (unsigned)iy < (unsigned)p.Hl && (unsigned)ix < (unsigned)p.Wl;
            const int64_t off = (((int64_t)a_n[i] * p.H + (iy >> p.up2)) * p.W + (ix >> p.up2)) * p.Cin + ci;
            a_reg[i] = ld16_or_zero(x, off * (int64_t)sizeof(T), ok);
            if (p.in_scale) {
                const float* sp = p.in_scale + (ok ? (int64_t)a_n[i] * p.Cin + ci : 0);
#pragma unroll
                for (int j = 0; j < SPC; ++j) a_sc[i][j] = *reinterpret_cast<const float4*>(sp + 4 * j);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (tid + i * 256) / CH;
            const int co = bn0 + row;
            const bool ok = row < BN && co < p.Cout;
            b_reg[i] = ld16_or_zero(wt, ((int64_t)co * p.Kpad + (int64_t)s * 32 + ch * EPC) * (int64_t)sizeof(T), ok);
        }
        // advance the k tracker to the next step
        ci += 32;
        while (ci >= p.Cin) {
            ci -= p.Cin;
            ++tap;
            if (++dx == p.kw) {
                dx = 0;
                ++dy;
            }
        }
    };

    auto store_step = [&](int buf) {
        char* A = lds + buf * (A_BYTES + B_BYTES);
        char* B = A + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (tid + i * 256) / CH;
            uint4 v = a_reg[i];
            if (p.fu.in_slope != 1.f) v = lrelu_chunk(v, p.fu.in_slope, T());
            if (p.in_scale) {
                if constexpr (sizeof(T) == 2) {
                    bf16x8 h = *reinterpret_cast<bf16x8*>(&v);
                    h[0] = (bf16)((float)h[0] * a_sc[i][0].x); h[1] = (bf16)((float)h[1] * a_sc[i][0].y);
                    h[2] = (bf16)((float)h[2] * a_sc[i][0].z); h[3] = (bf16)((float)h[3] * a_sc[i][0].w);
                    h[4] = (bf16)((float)h[4] * a_sc[i][SPC - 1].x); h[5] = (bf16)((float)h[5] * a_sc[i][SPC - 1].y);
                    h[6] = (bf16)((float)h[6] * a_sc[i][SPC - 1].z); h[7] = (bf16)((float)h[7] * a_sc[i][SPC - 1].w);
                    v = *reinterpret_cast<uint4*>(&h);
                } else {
                    float4 f = *reinterpret_cast<float4*>(&v);
                    f.x *= a_sc[i][0].x; f.y *= a_sc[i][0].y; f.z *= a_sc[i][0].z; f.w *= a_sc[i][0].w;
                    v = *reinterpret_cast<uint4*>(&f);
                }
            }
            if (row < BM) *reinterpret_cast<uint4*>(A + lds_off<T>(row, ch)) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (tid + i * 256) / CH;
            if (row < BN) *reinterpret_cast<uint4*>(B + lds_off<T>(row, ch)) = b_reg[i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;
    constexpr int CPF = (int)sizeof(Frag) / 16;  // chunks per fragment (1 bf16, 2 fp32)

    load_step(0);
    store_step(0);
    __syncthreads();

    for (int s = 0; s < p.ksteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < p.ksteps) load_step(s + 1);

        const char* A = lds + buf * (A_BYTES + B_BYTES);
        const char* B = A + A_BYTES;
        Frag af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = wm * TM + i * 16 + frow;
            af[i] = Mma<T>::load(A + lds_off<T>(row, fq * CPF));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int row = wn * TN + j * 16 + frow;
            bf[j] = Mma<T>::load(B + lds_off<T>(row, fq * CPF));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) Mma<T>::mma(af[i], bf[j], acc[i][j]);

        if (s + 1 < p.ksteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D layout col = lane&15 (co), row = (lane>>4)*4 + r (pixel) -------------
    T* __restrict__ y = (T*)p.y;
    const T* __restrict__ noise = (const T*)p.noise;
    const T* __restrict__ aux = (const T*)p.fu.aux;
    const T* __restrict__ resid = (const T*)p.fu.resid;
    const int nr = p.fu.pool ? 1 : 4;               // pooled: the lane's four rows are one 2x2 window
    const int out_per_img = p.fu.pool ? HoWo >> 2 : HoWo;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int64_t mrow = bm0 + wm * TM + i * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nr || mrow + r >= p.M) continue;
            const int64_t m = p.fu.pool ? (mrow >> 2) : mrow + r;  // output pixel
            const int n = (int)(m / out_per_img);
            const float nz = noise ? (float)noise[m] : 0.f;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int co = bn0 + wn * TN + j * 16 + frow;
                if (co >= p.Cout) continue;
                const f32x4 a4 = acc[i][j];
                float v = p.fu.pool ? 0.25f * ((a4[0] + a4[1]) + (a4[2] + a4[3])) : a4[r];
                v *= p.fu.gain;
                if (p.out_scale) v *= p.out_scale[(int64_t)n * p.Cout + co];
                if (p.bias) v += p.bias[co];
                if (noise) v = fmaf(nz, p.noise_scale[co], v);
                v = lrelu_f(v, p.slope);
                if (aux) v *= lrelu_mask((float)aux[m * p.Cout + co], p.fu.aux_slope);
                if (resid) v += (float)resid[m * p.Cout + co];
                y[m * p.Cout + co] = (T)v;
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const ConvArgs& a, hipStream_t st) {
    const int64_t gm = ceil_div64(a.M, BM);
    const int gn = (a.Cout + BN - 1) / BN;
    if (gm >= (1LL << 31) || gn > 65535) {
        gan_set_error("conv2d_fwd: grid too large");
        return GAN_ERR_ARG;
    }
    hipLaunchKernelGGL((conv_fwd_k<T, BM, BN, WM, WN>), dim3((unsigned)gm, (unsigned)gn), dim3(256), 0, st, a);
    GAN_CHECK_LAUNCH("conv2d_fwd");
    return GAN_OK;
}

template <typename T>
int launch(const ConvArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        // fp32 tiles are twice the bytes: keep 2 x (BM + BN) x 128 B within 64 KiB of LDS
        if (a.Cout <= 16) return launch_cfg<T, 128, 16, 4, 1>(a, st);
        if (a.Cout <= 32) return launch_cfg<T, 128, 32, 4, 1>(a, st);
        if (a.M >= 64 * 1024) return launch_cfg<T, 128, 64, 2, 2>(a, st);
        return launch_cfg<T, 64, 64, 2, 2>(a, st);
    } else {
        if (a.Cout <= 16) return launch_cfg<T, 256, 16, 4, 1>(a, st);
        if (a.Cout <= 32) return launch_cfg<T, 256, 32, 4, 1>(a, st);
        if (a.Cout <= 64) {
            if (a.M >= 64 * 1024) return launch_cfg<T, 128, 64, 2, 2>(a, st);
            return launch_cfg<T, 64, 64, 2, 2>(a, st);
        }
        const int64_t big_blocks = ceil_div64(a.M, 128) * ((a.Cout + 127) / 128);
        if (big_blocks >= 512) return launch_cfg<T, 128, 128, 2, 2>(a, st);
        return launch_cfg<T, 64, 64, 2, 2>(a, st);
    }
}

}  // namespace

// ---- full-window convolution with a handful of output channels -----------------------------------
// kh == H, kw == W, no padding: one output pixel per image (the discriminator's last layer, 4x4x512 -> 1,
// progressive_gan.py:76).  As an implicit GEMM it is 256 x 1 x 8192: a tile kernel runs it in one or two
// workgroups that walk K serially (243 us at batch 256).  Here a workgroup owns one image and its lanes split K:
// y[n][co] = act(gain * sum_k lrelu_in(x[n][k]) * wt[co][k] + bias[co]); x's NHWC order IS the k order.
template <typename T>
__global__ void __launch_bounds__(256) conv_full_k(const T* __restrict__ x, const T* __restrict__ wt, T* __restrict__ y,
                                                   const float* __restrict__ bias, int K, int Kpad, int Cout, float in_slope,
                                                   float gain, float slope) {
    constexpr int V = 16 / sizeof(T);
    __shared__ float red[4][8];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* __restrict__ xr = x + (int64_t)n * K;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int k = tid * V; k < K; k += 256 * V) {  // K % V == 0 (Cin % 8 == 0)
        const uint4 xv = *reinterpret_cast<const uint4*>(xr + k);
        const T* xe = reinterpret_cast<const T*>(&xv);
        float xf[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float v = (float)xe[j];
            xf[j] = v > 0.f ? v : v * in_slope;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < Cout) {
                const uint4 wv = *reinterpret_cast<const uint4*>(wt + (int64_t)c * Kpad + k);
                const T* we = reinterpret_cast<const T*>(&wv);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[c] += xf[j] * (float)we[j];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
        if (lane == 0) red[wave][c] = acc[c];
    }
    __syncthreads();
    if (tid < Cout) {
        float v = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) * gain;
        if (bias) v += bias[tid];
        v = v > 0.f ? v : v * slope;
        y[(int64_t)n * Cout + tid] = (T)v;
    }
}

// ---- 1x1 convolution into <= 4 channels on maps too small for the thin kernel: the generator's to-RGB layers at 4x4 .. 32x32
// (stylegan2.py:64, 512 .. 128 channels -> 3).  The implicit-GEMM kernel gives such a layer M / 256 workgroups (two at 4x4,
// batch 32) that each walk all of K: 15-32 us whatever the batch, four layers, two generator passes per iteration.  Here a
// wave owns four pixels of one image: a lane holds 8 channels of the (style-scaled) weights in registers, reads 16 bytes
// of each pixel, and the wave reduces the 4 x <= 4 sums.
__global__ void __launch_bounds__(256) conv1x1_narrow_k(const bf16* __restrict__ x, const bf16* __restrict__ wt, bf16* __restrict__ y,
                                                        const float* __restrict__ in_scale, const float* __restrict__ bias, int64_t M, int HW,
                                                        int Cin, int Cout, int Kpad, float gain, float slope) {
    // four pixels of one image per wave (HW % 4 == 0), their loads, the weights' and the style's all issued before the first use
    constexpr int PPW = 4;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW;
    if (p0 >= M) return;
    const int c0 = lane * 8;
    const bool act = c0 < Cin;
    const int n = (int)(p0 / HW);
    float xv[PPW][8], w[4][8], sc[8];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[q][j] = 0.f;
        if (act && p0 + q < M) load_f<bf16, 8>(x + (p0 + q) * Cin + c0, xv[q]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[r][j] = 0.f;
        if (act && r < Cout) load_f<bf16, 8>(wt + (int64_t)r * Kpad + c0, w[r]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = 1.f;
    if (in_scale && act) {
        const float4 a = *reinterpret_cast<const float4*>(in_scale + (int64_t)n * Cin + c0);
        const float4 b = *reinterpret_cast<const float4*>(in_scale + (int64_t)n * Cin + c0 + 4);
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[r][j] *= sc[j];
    float acc[PPW][4];
#pragma unroll
    for (int q = 0; q < PPW; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf(xv[q][j], w[r][j], a);
            acc[q][r] = a;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int q = 0; q < PPW; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] += __shfl_xor(acc[q][r], off);
    // lane 4q + r stores channel r of pixel q
    const int q = lane >> 2, r = lane & 3;
    if (lane < 4 * PPW && r < Cout && p0 + q < M) {
        float v = 0.f;
#pragma unroll
        for (int qq = 0; qq < PPW; ++qq)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (qq == q && rr == r) v = acc[qq][rr];
        v *= gain;
        if (bias) v += bias[r];
        v = v > 0.f ? v : v * slope;
        y[(p0 + q) * Cout + r] = (bf16)v;
    }
}

static int conv2d_fwd_impl(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                           const float* bias, const void* noise, const float* noise_scale, float slope, int N,
                           int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px,
                           int Kpad, float in_slope, int pool, const void* aux, float aux_slope, float gain, const void* resid,
                           float* workspace, int64_t workspace_floats, int dtype, gan_stream_t stream, const void* dot_x = nullptr,
                           float* dot_out = nullptr, const void* sc_x = nullptr, const void* sc_w = nullptr, int sc_c = 0, float sc_gain = 0.f) {
    GAN_CHECK_ARG(N == 0 || (x && wt && y), "conv2d_fwd: null pointer");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "conv2d_fwd: bad dtype %d", dtype);
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "conv2d_fwd: bad shape");
    GAN_CHECK_ARG(Cin % 8 == 0, "conv2d_fwd: Cin=%d must be a multiple of 8 (pad the input)", Cin);
    GAN_CHECK_ARG(kh >= 1 && kw >= 1 && kh * kw <= 64, "conv2d_fwd: bad kernel %dx%d", kh, kw);
    GAN_CHECK_ARG(up2 == 0 || up2 == 1, "conv2d_fwd: up2 must be 0 or 1");
    GAN_CHECK_ARG(Kpad % 32 == 0 && Kpad >= kh * kw * Cin, "conv2d_fwd: Kpad=%d must be a multiple of 32 and >= %d",
                  Kpad, kh * kw * Cin);
    GAN_CHECK_ARG((noise == nullptr) == (noise_scale == nullptr), "conv2d_fwd: noise and noise_scale go together");
    const int Hl = up2 ? 2 * H : H, Wl = up2 ? 2 * W : W;
    GAN_CHECK_ARG(Ho == Hl + 2 * py - kh + 1 && Wo == Wl + 2 * px - kw + 1,
                  "conv2d_fwd: output %dx%d does not match input %dx%d, kernel %dx%d, pad %d,%d", Ho, Wo, Hl, Wl, kh,
                  kw, py, px);
    GAN_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)wt % 16) == 0, "conv2d_fwd: x and wt must be 16-byte aligned");
    GAN_CHECK_ARG(pool == 0 || (pool == 1 && Ho % 2 == 0 && Wo % 2 == 0), "conv2d_fwd: pooling needs an even %dx%d output", Ho, Wo);
    if (N == 0) return GAN_OK;
    ConvArgs a;
    a.fu.in_slope = in_slope; a.fu.pool = pool; a.fu.aux = aux; a.fu.aux_slope = aux_slope; a.fu.gain = gain; a.fu.resid = resid;
    a.fu.ws = workspace; a.fu.ws_floats = workspace ? workspace_floats : 0;
    a.fu.dot_x = dot_x; a.fu.dot_out = dot_out;
    a.fu.sc_x = sc_x; a.fu.sc_w = sc_w; a.fu.sc_c = sc_c; a.fu.sc_gain = sc_gain;
    a.x = x; a.wt = wt; a.y = y;
    a.in_scale = in_scale; a.out_scale = out_scale; a.bias = bias;
    a.noise = noise; a.noise_scale = noise_scale; a.slope = slope;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Hl = Hl; a.Wl = Wl; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.kh = kh; a.kw = kw; a.py = py; a.px = px;
    a.Kpad = Kpad;
    a.ksteps = (kh * kw * Cin + 31) / 32;
    a.M = (int64_t)N * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
    if (Ho == 1 && Wo == 1 && py == 0 && px == 0 && !up2 && Cout <= 8 && !in_scale && !out_scale && !noise && !pool && !aux && !resid) {
        const int K = H * W * Cin;
        if (dtype == GAN_F32)
            hipLaunchKernelGGL(conv_full_k<float>, dim3((unsigned)N), dim3(256), 0, st, (const float*)x, (const float*)wt, (float*)y, bias,
                               K, Kpad, Cout, in_slope, gain, slope);
        else
            hipLaunchKernelGGL(conv_full_k<bf16>, dim3((unsigned)N), dim3(256), 0, st, (const bf16*)x, (const bf16*)wt, (bf16*)y, bias, K,
                               Kpad, Cout, in_slope, gain, slope);
        GAN_CHECK_LAUNCH("conv2d_fwd(full window)");
        return GAN_OK;
    }
    if (dtype == GAN_F32) return launch<float>(a, st);
    // thin layers (Cin, Cout <= 64 at >= 32 pixels wide) are HBM-bound: halo-tiled direct kernel
    const int thin = gan_conv2d_fwd_thin(x, wt, y, in_scale, out_scale, bias, noise, noise_scale, slope, N, H, W, Cin,
                                         up2, Ho, Wo, Cout, kh, kw, py, px, Kpad, a.fu, st);
    if (thin != 0) return thin < 0 ? thin : GAN_OK;
    if (dot_out || sc_x) {
        gan_set_error("conv2d_fwd_dot / conv2d_fwd_short: shape not served (ask gan_conv2d_fwd_dot_supported / _short_supported first)");
        return GAN_ERR_ARG;
    }
    if (kh == 1 && kw == 1 && py == 0 && px == 0 && !up2 && Cout <= 4 && Cin <= 512 && !out_scale && !noise && !pool && !aux && !resid &&
        in_slope == 1.f && ((uintptr_t)in_scale % 16) == 0 && (Ho * Wo) % 4 == 0) {
        static const bool narrow = !(getenv("GAN_AMD_NARROW1X1") && atoi(getenv("GAN_AMD_NARROW1X1")) == 0);  // 0: the implicit-GEMM kernel (A/B runs)
        if (narrow) {
            const int HW = Ho * Wo;
            const int64_t waves = (a.M + 3) / 4;
            hipLaunchKernelGGL(conv1x1_narrow_k, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const bf16*)x, (const bf16*)wt, (bf16*)y,
                               in_scale, bias, a.M, HW, Cin, Cout, Kpad, gain, slope);
            GAN_CHECK_LAUNCH("conv2d_fwd(narrow 1x1)");
            return GAN_OK;
        }
    }
    if (!in_scale) {
        const int wide = gan_conv2d_fwd_wide(x, wt, y, out_scale, bias, noise, noise_scale, slope, N, H, W, Cin, up2, Ho, Wo,
                                             Cout, kh, kw, py, px, Kpad, a.fu, st);
        if (wide != 0) return wide < 0 ? wide : GAN_OK;
    }
    return launch<bf16>(a, st);
}

extern "C" int gan_conv2d_fwd(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                              const float* bias, const void* noise, const float* noise_scale, float slope, int N,
                              int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px,
                              int Kpad, float in_slope, int pool, const void* aux, float aux_slope, float gain, const void* resid,
                              int dtype, gan_stream_t stream) {
    return conv2d_fwd_impl(x, wt, y, in_scale, out_scale, bias, noise, noise_scale, slope, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, Kpad,
                           in_slope, pool, aux, aux_slope, gain, resid, nullptr, 0, dtype, stream);
}

// ---- the same call with an optional caller workspace (include/gan_amd.h: gan_conv2d_fwd_ws) -----------------------------
extern "C" int64_t gan_conv2d_fwd_workspace(int N, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int Kpad, int has_in_scale, int dtype) {
    if (dtype != GAN_BF16 || has_in_scale || N <= 0) return 0;
    return gan_conv2d_fwd_wide_workspace(N, Cin, Ho, Wo, Cout, kh, kw, Kpad);
}

extern "C" int gan_conv2d_fwd_ws(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                                 const float* bias, const void* noise, const float* noise_scale, float slope, int N,
                                 int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px,
                                 int Kpad, float in_slope, int pool, const void* aux, float aux_slope, float gain, const void* resid,
                                 float* workspace, int64_t workspace_floats, int dtype, gan_stream_t stream) {
    return conv2d_fwd_impl(x, wt, y, in_scale, out_scale, bias, noise, noise_scale, slope, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, Kpad,
                           in_slope, pool, aux, aux_slope, gain, resid, workspace, workspace_floats, dtype, stream);
}

// ---- the input-gradient convolution of a modulated layer with its scale + dot pass fused (include/gan_amd.h) ------------
extern "C" int gan_conv2d_fwd_dot_supported(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad,
                                            int pool, int dtype) {
    // mirror of gan_conv2d_fwd_thin's shape rules for the THIN_DS / THIN_DPS modes
    if (dtype != GAN_BF16 || N <= 0) return 0;
    if (kh != kw || (kh != 1 && kh != 3) || py != (kh - 1) / 2 || px != py) return 0;
    if (Cin != 8 && Cin != 16 && Cin != 32 && Cin != 64) return 0;
    if (Cout > 64 || (Cout & 3) || Wo < 32 || Ho < 8) return 0;
    if (Kpad != (kh * kw * Cin + 31) / 32 * 32) return 0;
    if (Ho != H + 2 * py - kh + 1 || Wo != W + 2 * px - kw + 1) return 0;
    if (pool && ((Ho | Wo) & 1)) return 0;
    const int64_t xb = (int64_t)N * H * W * Cin * 2;
    const int64_t yb = (int64_t)N * (pool ? Ho / 2 : Ho) * (pool ? Wo / 2 : Wo) * Cout * 2;
    if (xb >= (1LL << 31) || yb >= (1LL << 31) || (int64_t)N * Ho * Wo >= (1LL << 31)) return 0;
    return 1;
}

extern "C" int gan_conv2d_fwd_dot(const void* x, const void* wt, void* y, const float* out_scale, const void* dot_x, float* dot_out, int N, int H,
                                  int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad, int pool, float gain,
                                  int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(out_scale && dot_x && dot_out, "conv2d_fwd_dot: null pointer");
    GAN_CHECK_ARG(gan_conv2d_fwd_dot_supported(N, H, W, Cin, Ho, Wo, Cout, kh, kw, py, px, Kpad, pool, dtype) == 1,
                  "conv2d_fwd_dot: shape not served (thin layers only: bf16, Cin in {8, 16, 32, 64}, Cout <= 64 and a multiple of 4, >= 32 pixels wide)");
    GAN_CHECK_ARG(((uintptr_t)y % 8) == 0 && ((uintptr_t)dot_x % 8) == 0, "conv2d_fwd_dot: y and dot_x must be 8-byte aligned");
    return conv2d_fwd_impl(x, wt, y, nullptr, out_scale, nullptr, nullptr, nullptr, 1.f, N, H, W, Cin, 0, Ho, Wo, Cout, kh, kw, py, px, Kpad, 1.f, pool,
                           nullptr, 1.f, gain, nullptr, nullptr, 0, dtype, stream, dot_x, dot_out);
}

// ---- conv1's input gradient of a D block with the shortcut's input gradient formed in the kernel (include/gan_amd.h) ------
extern "C" int gan_conv2d_fwd_short_supported(int N, int H, int W, int Cin, int Cout, int kh, int kw, int py, int px, int Kpad, int sc_c, int dtype) {
    if (dtype != GAN_BF16 || N <= 0) return 0;
    if (kh != 3 || kw != 3 || py != 1 || px != 1) return 0;
    if (Cin != 8 && Cin != 16 && Cin != 32 && Cin != 64) return 0;
    if ((Cout != 16 && Cout != 32 && !(Cout == 64 && Cin == 64)) || sc_c != 2 * Cout || W < 32 || H < 8 || (H & 1) || (W & 1)) return 0;
    if (Kpad != (9 * Cin + 31) / 32 * 32) return 0;
    const int64_t big = (int64_t)N * H * W * (Cin > Cout ? Cin : Cout) * 2;
    if (big >= (1LL << 31) || (int64_t)N * H * W >= (1LL << 31)) return 0;
    return 1;
}

extern "C" int gan_conv2d_fwd_short(const void* x, const void* wt, void* y, const void* aux, float aux_slope, float gain, const void* sc_x,
                                    const void* sc_w, int sc_c, float sc_gain, int N, int H, int W, int Cin, int Cout, int kh, int kw, int py,
                                    int px, int Kpad, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(aux && sc_x && sc_w, "conv2d_fwd_short: null pointer");
    GAN_CHECK_ARG(gan_conv2d_fwd_short_supported(N, H, W, Cin, Cout, kh, kw, py, px, Kpad, sc_c, dtype) == 1,
                  "conv2d_fwd_short: shape not served (bf16, 3 x 3, Cout 16 or 32 -- 64 with Cin 64 --, sc_c = 2 Cout, even maps >= 32 pixels wide)");
    GAN_CHECK_ARG(((uintptr_t)y % 8) == 0 && ((uintptr_t)aux % 8) == 0 && ((uintptr_t)sc_x % 16) == 0 && ((uintptr_t)sc_w % 16) == 0,
                  "conv2d_fwd_short: y / aux must be 8-byte, sc_x / sc_w 16-byte aligned");
    return conv2d_fwd_impl(x, wt, y, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, N, H, W, Cin, 0, H, W, Cout, kh, kw, py, px, Kpad, 1.f, 0, aux,
                           aux_slope, gain, nullptr, nullptr, 0, dtype, stream, nullptr, nullptr, sc_x, sc_w, sc_c, sc_gain);
}
