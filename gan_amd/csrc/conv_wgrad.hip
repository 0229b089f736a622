// Convolution weight gradient on MFMA (gfx950), channels-last, stride 1.
//
//   dw[tap][co][ci] = sum_m gy[m][co] * X[pix(m)+tap][ci] * in_scale[n(m)][ci]
//
// GEMM view per tap: rows = co, cols = ci, reduction over output pixels m.  Both operands are
// stored pixel-major (channels contiguous), i.e. the reduction index is the SLOW one, which is
// the transposed orientation for MFMA fragments (a lane needs 8 consecutive k of one row).
// bf16 uses the gfx950 transposing LDS read (ds_read_b64_tr_b16: 4 pixels x 16 channels per
// 16-lane group delivered channel-major); fp32 reads its fragment element-wise.
//
// One workgroup = 4 waves.  Per step it stages 128 pixels of gy (TCO channels) and of the
// shifted, optionally modulated input (TCI channels) into LDS with coalesced 16-byte loads; wave
// w reduces pixels [32w, 32w+32) of the step into its own TCO x TCI accumulator tile.  The
// pixel range is split over blockIdx.z; partial tiles are added to dw with fp32 atomics.
#include "common.h"

namespace {

struct WgradArgs {
    const void* x;
    const void* gy;
    float* dw;
    const float* in_scale;
    int N, H, W, Cin, Hl, Wl, up2;
    int Ho, Wo, Cout, kh, kw, py, px;
    int ci_tiles;
    int64_t M;
    int64_t pix_per_split;
    WgradFuse fu;
};

template <typename T>
struct WFrag;

template <>
struct WFrag<bf16> {
    typedef bf16x8 Frag;
    // `base` points at element [k0][c0] of a row-major [pixels][TC] LDS tile
    static __device__ __forceinline__ Frag load(const bf16* tile, int TC, int k0, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
        const bf16* p0 = tile + (k0 + 8 * g + q) * TC + c0 + 4 * pp;
        const bf16* p1 = p0 + 4 * TC;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p1);
        Frag f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <>
struct WFrag<float> {
    struct Frag {
        float v[8];
    };
    static __device__ __forceinline__ Frag load(const float* tile, int TC, int k0, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = tile[(k0 + 8 * g + j) * TC + c0 + i];
        return f;
    }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    }
};

__device__ __forceinline__ uint4 wscale_chunk(uint4 raw, const float* s, bf16) {
    bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] * s[e]);
    return *reinterpret_cast<uint4*>(&v);
}
__device__ __forceinline__ uint4 wscale_chunk(uint4 raw, const float* s, float) {
    float4 v = *reinterpret_cast<float4*>(&raw);
    v.x *= s[0];
    v.y *= s[1];
    v.z *= s[2];
    v.w *= s[3];
    return *reinterpret_cast<uint4*>(&v);
}

template <typename T, int TCO, int TCI>
__global__ void __launch_bounds__(256) conv_wgrad_k(const WgradArgs p) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CHA = TCO / EPC;  // 16-byte chunks per LDS row
    constexpr int CHB = TCI / EPC;
    constexpr int NA = CHA / 2, NB = CHB / 2;  // chunks per thread (two threads share a row)
    constexpr int FM = TCO / 16, FN = TCI / 16;
    constexpr int PIX = 128;
    typedef typename WFrag<T>::Frag Frag;

    __shared__ __attribute__((aligned(16))) T A_lds[PIX * TCO];
    __shared__ __attribute__((aligned(16))) T B_lds[PIX * TCI];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int co0 = (blockIdx.x / p.ci_tiles) * TCO;
    const int ci0 = (blockIdx.x % p.ci_tiles) * TCI;
    const int tap = blockIdx.y;
    const int dy = tap / p.kw - p.py;
    const int dx = tap % p.kw - p.px;
    const int HoWo = p.Ho * p.Wo;

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ gy = (const T*)p.gy;

    const int64_t m_begin = (int64_t)blockIdx.z * p.pix_per_split;
    int64_t m_end = m_begin + p.pix_per_split;
    if (m_end > p.M) m_end = p.M;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row = tid >> 1;   // pixel row of the step this thread stages
    const int half = tid & 1;   // which half of the row's chunks

    for (int64_t m0 = m_begin; m0 < m_end; m0 += PIX) {
        const int64_t m = m0 + row;
        const bool in_range = m < m_end;
        const int64_t mm = in_range ? m : 0;
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (int64_t)n * HoWo);
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        // ---- gy tile (gy_up2: gy is stored at half resolution) ---------------------------
        uint4 a_reg[NA];
        {
            const int gsh = p.fu.gy_up2;
            const int64_t gbase = (((int64_t)n * (p.Ho >> gsh) + (oy >> gsh)) * (p.Wo >> gsh) + (ox >> gsh)) * p.Cout + co0;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a_reg[i] = ld16_or_zero(gy, (gbase + (half + 2 * i) * EPC) * (int64_t)sizeof(T), in_range);
        }
        // ---- shifted input tile -------------------------------------------------------
        uint4 b_reg[NB];
        {
            const int iy = oy + dy, ix = ox + dx;
            const bool ok = in_range && (unsigned)iy < (unsigned)p.Hl && (unsigned)ix < (unsigned)p.Wl;
            const int64_t base = (((int64_t)n * p.H + (iy >> p.up2)) * p.W + (ix >> p.up2)) * p.Cin + ci0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = half + 2 * i;
                uint4 v = ld16_or_zero(x, (base + c * EPC) * (int64_t)sizeof(T), ok);
                if (p.fu.x_slope != 1.f) v = lrelu_chunk(v, p.fu.x_slope, T());
                if (p.in_scale) v = wscale_chunk(v, p.in_scale + (int64_t)n * p.Cin + ci0 + c * EPC, T());
                b_reg[i] = v;
            }
        }
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(A_lds + row * TCO + (half + 2 * i) * EPC) = a_reg[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<uint4*>(B_lds + row * TCI + (half + 2 * i) * EPC) = b_reg[i];
        __syncthreads();

        const int k0 = wid * 32;
        Frag af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = WFrag<T>::load(A_lds, TCO, k0, i * 16, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = WFrag<T>::load(B_lds, TCI, k0, j * 16, lane);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) WFrag<T>::mma(af[i], bf[j], acc[i][j]);
    }

    // D layout: col = lane&15 -> ci, row = (lane>>4)*4 + r -> co
    float* __restrict__ dw = p.dw + (int64_t)tap * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + i * 16 + (lane >> 4) * 4 + r;
                const int ci = ci0 + j * 16 + (lane & 15);
                atomicAdd(dw + (int64_t)co * p.Cin + ci, acc[i][j][r] * p.fu.gain);
            }
}

template <typename T, int TCO, int TCI>
int launch_cfg(WgradArgs a, hipStream_t st) {
    const int co_tiles = a.Cout / TCO;
    a.ci_tiles = a.Cin / TCI;
    const int taps = a.kh * a.kw;
    const int64_t base_blocks = (int64_t)co_tiles * a.ci_tiles * taps;
    int64_t splits = ceil_div64(2048, base_blocks);
    const int64_t max_splits = ceil_div64(a.M, 128);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    a.pix_per_split = ceil_div64(ceil_div64(a.M, splits), 128) * 128;
    splits = ceil_div64(a.M, a.pix_per_split);
    hipLaunchKernelGGL((conv_wgrad_k<T, TCO, TCI>), dim3((unsigned)(co_tiles * a.ci_tiles), (unsigned)taps, (unsigned)splits),
                       dim3(256), 0, st, a);
    GAN_CHECK_LAUNCH("conv2d_wgrad");
    return GAN_OK;
}

template <typename T, int TCO>
int launch_ci(const WgradArgs& a, hipStream_t st) {
    if (a.Cin % 64 == 0) return launch_cfg<T, TCO, 64>(a, st);
    if (a.Cin % 32 == 0) return launch_cfg<T, TCO, 32>(a, st);
    return launch_cfg<T, TCO, 16>(a, st);
}

template <typename T>
int launch(const WgradArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        // fp32: 128 x (TCO + TCI) x 4 B of LDS, keep the tile at 32 + 32 channels
        if (a.Cout % 32 == 0) {
            if (a.Cin % 32 == 0) return launch_cfg<T, 32, 32>(a, st);
            return launch_cfg<T, 32, 16>(a, st);
        }
        if (a.Cin % 32 == 0) return launch_cfg<T, 16, 32>(a, st);
        return launch_cfg<T, 16, 16>(a, st);
    } else {
        if (a.Cout % 64 == 0) return launch_ci<T, 64>(a, st);
        if (a.Cout % 32 == 0) return launch_ci<T, 32>(a, st);
        return launch_ci<T, 16>(a, st);
    }
}

}  // namespace

extern "C" int64_t gan_conv2d_wgrad_workspace(int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh,
                                              int kw, int py, int px, int dtype) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0) return 0;
    const int64_t L = (int64_t)kh * kw * Cout * Cin;  // every path folds at least one padded slice into dw
    if (dtype != GAN_BF16) return L;
    int64_t need = 0;
    gan_conv2d_wgrad_thin(nullptr, nullptr, DwOut{nullptr, 0, 0, 0, nullptr, 0}, nullptr, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, nullptr, 0,
                          &need, WgradFuse{1.f, 0, 1.f}, nullptr);
    if (need == 0) need = gan_wgrad_tiled_workspace(N, Ho, Wo, Cin, Cout, kh, kw, py, px);
    return need > L ? need : L;
}

extern "C" int gan_chan_dot(const void* A, const void* B, int b_mode, const void* ymask, float slope, float* out, int64_t out_stride_n, int N,
                            int64_t HW, int C, float* workspace, int64_t workspace_floats, int accumulate, int dtype, gan_stream_t stream);
extern "C" int64_t gan_chan_dot_workspace(int N, int64_t HW, int C, int dtype);

static int wgrad_impl(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, const float* in_scale, int N,
                      int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px,
                      float* workspace, int64_t workspace_floats, float x_slope, int gy_up2, float gain, int accumulate,
                      int dtype, gan_stream_t stream, float* dbias = nullptr, int dbias_accumulate = 0) {
    GAN_CHECK_ARG(dw && (N == 0 || (x && gy)), "conv2d_wgrad: null pointer");
    GAN_CHECK_ARG(accumulate == 0 || accumulate == 1, "conv2d_wgrad: accumulate must be 0 or 1");
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "conv2d_wgrad: bad dtype %d", dtype);
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "conv2d_wgrad: bad shape");
    GAN_CHECK_ARG(Cin % 16 == 0 && Cout % 16 == 0, "conv2d_wgrad: Cin=%d and Cout=%d must be multiples of 16 (pad)",
                  Cin, Cout);
    GAN_CHECK_ARG(Cout_out >= 1 && Cout_out <= Cout && Cin_out >= 1 && Cin_out <= Cin,
                  "conv2d_wgrad: dw is %d x %d channels, the padded operands have %d x %d", Cout_out, Cin_out, Cout, Cin);
    GAN_CHECK_ARG(kh >= 1 && kw >= 1 && kh * kw <= 64, "conv2d_wgrad: bad kernel %dx%d", kh, kw);
    GAN_CHECK_ARG(up2 == 0 || up2 == 1, "conv2d_wgrad: up2 must be 0 or 1");
    const int Hl = up2 ? 2 * H : H, Wl = up2 ? 2 * W : W;
    GAN_CHECK_ARG(Ho == Hl + 2 * py - kh + 1 && Wo == Wl + 2 * px - kw + 1,
                  "conv2d_wgrad: output %dx%d does not match input %dx%d, kernel %dx%d, pad %d,%d", Ho, Wo, Hl, Wl,
                  kh, kw, py, px);
    GAN_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)gy % 16) == 0, "conv2d_wgrad: x and gy must be 16-byte aligned");
    GAN_CHECK_ARG(gy_up2 == 0 || (gy_up2 == 1 && Ho % 2 == 0 && Wo % 2 == 0), "conv2d_wgrad: gy_up2 needs an even %dx%d output", Ho, Wo);
    const WgradFuse fu{x_slope, gy_up2, gain};
    GAN_CHECK_ARG(dbias == nullptr || Cout_out == Cout, "conv2d_wgrad_bias: the bias gradient needs an unpadded Cout (%d of %d)", Cout_out, Cout);
    GAN_CHECK_ARG(dbias_accumulate == 0 || dbias_accumulate == 1, "conv2d_wgrad_bias: dbias_accumulate must be 0 or 1");
    DwOut dwo{dw, Cout_out, Cin_out, accumulate, dbias, dbias_accumulate};
    hipStream_t st = (hipStream_t)stream;
    const int64_t L = (int64_t)kh * kw * Cout * Cin;
    // the bias gradient on the paths whose kernels do not produce it: the channel-sum kernel over gy (after the weight
    // gradient: it reuses the workspace)
    const int64_t HWg = (int64_t)(Ho >> gy_up2) * (Wo >> gy_up2);
    auto bias_by_sum = [&]() -> int {
        if (!dbias) return GAN_OK;
        return gan_chan_dot(gy, nullptr, 0, nullptr, 1.f, dbias, 0, N, HWg, Cout, workspace, workspace_floats, dbias_accumulate, dtype, stream);
    };
    if (N == 0) {
        if (dbias && !dbias_accumulate && hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st) != hipSuccess) {
            gan_set_error("conv2d_wgrad: memset failed");
            return GAN_ERR_LAUNCH;
        }
        if (accumulate) return GAN_OK;
        if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)kh * kw * Cout_out * Cin_out, st) != hipSuccess) {
            gan_set_error("conv2d_wgrad: memset failed");
            return GAN_ERR_LAUNCH;
        }
        return GAN_OK;
    }
    int64_t need = gan_conv2d_wgrad_workspace(N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, dtype);
    if (dbias) {
        const int64_t nb = gan_chan_dot_workspace(N, HWg, Cout, dtype);
        if (nb > need) need = nb;
    }
    GAN_CHECK_ARG(workspace != nullptr && workspace_floats >= need && ((uintptr_t)workspace % 16) == 0,
                  "conv2d_wgrad: needs a 16-byte aligned workspace of %lld floats (gan_conv2d_wgrad_workspace), got %lld",
                  (long long)need, (long long)workspace_floats);
    if (dtype == GAN_BF16) {
        int rc = gan_conv2d_wgrad_thin(x, gy, dwo, in_scale, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, workspace,
                                       workspace_floats, nullptr, fu, st);
        if (rc != 0) return rc < 0 ? rc : GAN_OK;  // (the thin and the tiled kernel sum gy on the way: dwo.bias)
        rc = gan_conv2d_wgrad_tiled(x, gy, dwo, in_scale, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, workspace,
                                    workspace_floats, fu, st);
        if (rc != 0) return rc < 0 ? rc : (rc == 2 ? bias_by_sum() : GAN_OK);
    }
    dwo.bias = nullptr;
    // register-staged kernel: fp32 atomics into one zeroed, padded slice of the workspace, then the same reducer
    if (hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)L, st) != hipSuccess) {
        gan_set_error("conv2d_wgrad: memset failed");
        return GAN_ERR_LAUNCH;
    }
    WgradArgs a;
    a.fu = fu;
    a.x = x; a.gy = gy; a.dw = workspace; a.in_scale = in_scale;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Hl = Hl; a.Wl = Wl; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.kh = kh; a.kw = kw; a.py = py; a.px = px;
    a.ci_tiles = 1;
    a.M = (int64_t)N * Ho * Wo;
    a.pix_per_split = 128;
    const int rc = dtype == GAN_F32 ? launch<float>(a, st) : launch<bf16>(a, st);
    if (rc != GAN_OK) return rc;
    const int rf = gan_finalize_dw(workspace, 1, kh * kw, Cout, Cin, dwo, 1.f, st);  // the kernel applied the gain
    return rf != GAN_OK ? rf : bias_by_sum();
}

extern "C" int gan_conv2d_wgrad(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, const float* in_scale, int N,
                                int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px,
                                float* workspace, int64_t workspace_floats, float x_slope, int gy_up2, float gain, int accumulate,
                                int dtype, gan_stream_t stream) {
    return wgrad_impl(x, gy, dw, Cout_out, Cin_out, in_scale, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, workspace,
                      workspace_floats, x_slope, gy_up2, gain, accumulate, dtype, stream);
}

extern "C" int64_t gan_conv2d_wgrad_bias_workspace(int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py,
                                                   int px, int gy_up2, int dtype) {
    const int64_t a = gan_conv2d_wgrad_workspace(N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, dtype);
    const int64_t b = gan_chan_dot_workspace(N, (int64_t)(Ho >> (gy_up2 ? 1 : 0)) * (Wo >> (gy_up2 ? 1 : 0)), Cout, dtype);
    return a > b ? a : b;
}

extern "C" int gan_conv2d_wgrad_bias(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, float* dbias, int dbias_accumulate,
                                     const float* in_scale, int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw,
                                     int py, int px, float* workspace, int64_t workspace_floats, float x_slope, int gy_up2, float gain,
                                     int accumulate, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(dbias != nullptr, "conv2d_wgrad_bias: null dbias (gan_conv2d_wgrad computes the weight gradient alone)");
    return wgrad_impl(x, gy, dw, Cout_out, Cin_out, in_scale, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, workspace,
                      workspace_floats, x_slope, gy_up2, gain, accumulate, dtype, stream, dbias, dbias_accumulate);
}

// ---- the entry point that allocates nothing (include/gan_amd.h: gan_conv2d_wgrad_static) ------------------------
// The split-K slices go to a file-scope device array instead of a caller-owned workspace.  In host code the
// array's NAME is only a stand-in for the symbol: its device address is resolved once per device with
// hipGetSymbolAddress (that device current) and kept; every use below goes through that pointer.
static __device__ __align__(256) float scratch[GAN_WGRAD_STATIC_MAX_FLOATS];

static constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices];  // per device: each GPU holds its own copy of the array at its own address
static int g_scratch_state[kMaxDevices];   // 0 = not resolved, 1 = resolved

static int resolve_scratch(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        gan_set_error("conv2d_wgrad_static: no current HIP device (or device index >= %d)", kMaxDevices);
        return GAN_ERR_LAUNCH;
    }
    if (__atomic_load_n(&g_scratch_state[dev], __ATOMIC_ACQUIRE) != 1) {
        float* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            gan_set_error("conv2d_wgrad_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return GAN_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = p;  // racing first calls resolve the same address
        __atomic_store_n(&g_scratch_state[dev], 1, __ATOMIC_RELEASE);
    }
    *out = g_scratch_ptr[dev];
    return GAN_OK;
}

extern "C" int gan_amd_init(void) {
    float* p = nullptr;
    return resolve_scratch(&p);
}

extern "C" int64_t gan_conv2d_wgrad_static_max_floats(void) { return (int64_t)GAN_WGRAD_STATIC_MAX_FLOATS; }

extern "C" int gan_conv2d_wgrad_static(const void* x, const void* gy, float* dw, int Cout_out, int Cin_out, const float* in_scale,
                                       int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py,
                                       int px, float x_slope, int gy_up2, float gain, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "conv2d_wgrad_static: bad dtype %d", dtype);
    const int64_t need = gan_conv2d_wgrad_workspace(N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, dtype);
    GAN_CHECK_ARG(need <= (int64_t)GAN_WGRAD_STATIC_MAX_FLOATS,
                  "conv2d_wgrad_static: this shape needs %lld floats of split-K scratch, the built-in buffer holds %lld "
                  "(use gan_conv2d_wgrad with a caller-owned workspace)",
                  (long long)need, (long long)GAN_WGRAD_STATIC_MAX_FLOATS);
    float* ws = nullptr;
    const int rc = resolve_scratch(&ws);
    if (rc != GAN_OK) return rc;
    // Every slice the kernels read they have written in this call (plain stores), and the atomics path clears its one
    // slice with hipMemsetAsync on `stream` first: nothing of the previous call's contents is read.
    return wgrad_impl(x, gy, dw, Cout_out, Cin_out, in_scale, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py, px, ws,
                      (int64_t)GAN_WGRAD_STATIC_MAX_FLOATS, x_slope, gy_up2, gain, 0, dtype, stream);
}
