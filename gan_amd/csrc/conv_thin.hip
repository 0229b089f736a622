// Direct convolution for THIN layers on MFMA (gfx950): Cin, Cout <= 64 at large spatial size.
//
// These are the 128x128 / 256x256 layers of the StyleGAN2 generator and discriminator
// (modelling/stylegan2.py:29-36, modelling/progressive_gan.py:45-50,61): 16-64 channels, 65 K - 16 M
// pixels.  Their arithmetic intensity (K = 9*Cin = 144..576 MACs per output element) puts them
// on the HBM roofline, not the MFMA one, so the kernel is organised around bytes:
//
//   * A workgroup (4 waves) owns a TH x 32 tile of output pixels.  The (TH+2) x 34 input halo is
//     read from HBM ONCE with 16-byte loads (nearest-2x upsampling and the per-sample style
//     modulation are applied on the way), and parked in LDS as [8-channel chunk][pixel][8] so
//     that an MFMA operand (8 consecutive channels of 16 consecutive pixels) is one conflict-free
//     ds_read_b128 for every tap -- the 9 taps re-read LDS, never HBM/L2.
//   * The whole weight matrix [Cout][9*Cin] lives in LDS for the lifetime of the workgroup; a
//     workgroup walks a contiguous range of tiles (persistent), and the halo of tile t+1 is in
//     flight in registers while tile t is on the matrix cores.
//   * Operand roles are swapped relative to the big-GEMM kernel: A = weights (rows = co),
//     B = pixels (cols), so each lane ends up with 4 CONSECUTIVE output channels of one pixel
//     and the epilogue (demodulation, bias, noise, leaky-ReLU) stores 8 bytes per lane, 512
//     contiguous bytes per wave instruction for Cout = 16.
//
// Same semantics as conv_fwd.hip (see include/gan_amd.h: gan_conv2d_fwd); bf16 only.
#include "common.h"

namespace {

struct ThinArgs {
    const void* x;
    const void* wt;
    void* y;
    const float* in_scale;
    const float* out_scale;
    const float* bias;
    const void* noise;
    const float* noise_scale;
    float slope;
    int N, H, W, Hl, Wl, up2, Ho, Wo, Cout, py, px, Kpad;
    int tiles_x, tiles_y;
    ConvFuse fu;
};

__device__ __forceinline__ uint4 mod_chunk(uint4 raw, const float* s) {
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

template <int CIN, int COUT, int KS, int RPW>
struct ThinCfg {
    static constexpr int CP = CIN / 8;          // 16-byte channel chunks per pixel
    static constexpr int TH = 4 * RPW, TW = 32; // output tile
    static constexpr int HH = TH + KS - 1, HWD = TW + KS - 1, NPIX = HH * HWD;
    static constexpr int PLANE = (NPIX * 16 + 255) / 256 * 256;  // bytes per chunk plane, 256-B multiple
    static constexpr int KTOT = KS * KS * CIN, NSTEP = (KTOT + 31) / 32, KPAD = NSTEP * 32;
    static constexpr int FCO = COUT / 16, FPX = 2 * RPW;
    static constexpr int NCHUNK = ((NPIX + 7) / 8) * 8 * CP;
    static constexpr int NLD = (NCHUNK + 255) / 256;
    static constexpr int X_BYTES = CP * PLANE;
    static constexpr int W_BYTES = KPAD * COUT * 2;
    static constexpr int P_BYTES = 3 * COUT * 4;  // bias, noise scale, this tile's demodulation row (fp32)
    static constexpr int LDS_BYTES = X_BYTES + W_BYTES + P_BYTES;
};

template <int CIN, int COUT, int KS, int RPW>
__global__ void __launch_bounds__(256) conv_thin_k(const ThinArgs p) {
    typedef ThinCfg<CIN, COUT, KS, RPW> C;
    constexpr int CP = C::CP, TH = C::TH, HWD = C::HWD, NPIX = C::NPIX, PLANE = C::PLANE;
    constexpr int NSTEP = C::NSTEP, KPAD = C::KPAD, FCO = C::FCO, FPX = C::FPX, NLD = C::NLD;

    __shared__ __attribute__((aligned(256))) char lds[C::LDS_BYTES];
    char* const xl = lds;
    char* const wl = lds + C::X_BYTES;
    // Per-channel epilogue vectors live in LDS: read back with ds_read (lgkmcnt).  As global loads inside
    // the store loop each one made the compiler wait for vmcnt(0), i.e. for the PREVIOUS 8-byte store to
    // complete and for the prefetched halo: eight store round trips per tile, 10 us a tile.
    float* const pl = reinterpret_cast<float*>(lds + C::X_BYTES + C::W_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bf16* __restrict__ x = (const bf16*)p.x;
    const bf16* __restrict__ wt = (const bf16*)p.wt;

    // ---- weights -> LDS as [k/8][co][8]; rows co >= Cout are zero ------------------------------
    for (int idx = tid; idx < (KPAD / 8) * COUT; idx += 256) {
        const int co = idx % COUT;
        const int kc = idx / COUT;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (co < p.Cout) v = *reinterpret_cast<const uint4*>(wt + (int64_t)co * p.Kpad + kc * 8);
        *reinterpret_cast<uint4*>(wl + idx * 16) = v;
    }
    if (tid < COUT) {
        pl[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
        pl[COUT + tid] = (p.noise_scale && tid < p.Cout) ? p.noise_scale[tid] : 0.f;
    }

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int64_t ntiles = (int64_t)tiles_per_img * p.N;
    const int64_t t0 = ntiles * blockIdx.x / gridDim.x;
    const int64_t t1 = ntiles * (blockIdx.x + 1) / gridDim.x;

    // ---- halo loader: chunk idx -> (pixel, channel chunk); 8 consecutive lanes = 8 consecutive
    // pixels of one chunk plane (conflict-free ds_write_b128), 8*CP lanes = one contiguous run ----
    uint4 reg[NLD];
    float osc_reg = 1.f;
    unsigned okmask = 0;
    int n_ld = 0;
    auto issue = [&](int64_t t) {
        const int n = (int)(t / tiles_per_img);
        const int rem = (int)(t - (int64_t)n * tiles_per_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int ly0 = ty * TH - p.py, lx0 = tx * 32 - p.px;
        n_ld = n;
        okmask = 0;
        if (p.out_scale) osc_reg = p.out_scale[(int64_t)n * p.Cout + (tid < p.Cout ? tid : 0)];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = i * 256 + tid;
            const int p8 = idx / (8 * CP), r8 = idx % (8 * CP);
            const int c = r8 >> 3;
            const int pix = p8 * 8 + (r8 & 7);
            const int hy = pix / HWD, hx = pix - hy * HWD;
            const int ly = ly0 + hy, lx = lx0 + hx;
            const bool ok = pix < NPIX && (unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const int64_t off = (((int64_t)n * p.H + (ly >> p.up2)) * p.W + (lx >> p.up2)) * CIN + c * 8;
                v = *reinterpret_cast<const uint4*>(x + off);
                okmask |= 1u << i;
            }
            reg[i] = v;
        }
    };
    auto commit = [&]() {
        if (tid < COUT) pl[2 * COUT + tid] = osc_reg;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = i * 256 + tid;
            const int p8 = idx / (8 * CP), r8 = idx % (8 * CP);
            const int c = r8 >> 3;
            const int pix = p8 * 8 + (r8 & 7);
            if (pix < NPIX) {
                uint4 v = reg[i];
                if (p.fu.in_slope != 1.f) v = lrelu_chunk(v, p.fu.in_slope, bf16());
                if (p.in_scale && ((okmask >> i) & 1)) v = mod_chunk(v, p.in_scale + (int64_t)n_ld * CIN + c * 8);
                *reinterpret_cast<uint4*>(xl + c * PLANE + pix * 16) = v;
            }
        }
    };

    const int lpx = lane & 15;
    const int kg = lane >> 4;
    const char* const wlane = wl + (kg * COUT + lpx) * 16;
    const char* const xlane = xl + (wave * RPW * HWD + lpx) * 16;
    bf16* __restrict__ y = (bf16*)p.y;
    const bf16* __restrict__ noise = (const bf16*)p.noise;
    const bool vec_store = (p.Cout & 3) == 0;

    if (t0 < t1) issue(t0);
    for (int64_t t = t0; t < t1; ++t) {
        __syncthreads();  // the previous tile's operand reads (and the weight stores) are done
        const int n = n_ld;
        commit();
        __syncthreads();
        if (t + 1 < t1) issue(t + 1);  // next halo in flight while this tile computes

        f32x4 acc[FCO][FPX];
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int f = 0; f < FPX; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int k0 = s * 32 + kg * 8;
            int tap = k0 / CIN;
            const int c = (k0 % CIN) >> 3;
            if (tap > KS * KS - 1) tap = KS * KS - 1;  // zero-weight padding taps: any resident pixel
            const int dy = tap / KS, dx = tap - dy * KS;
            const char* const xs = xlane + c * PLANE + (dy * HWD + dx) * 16;
            bf16x8 a[FCO];
#pragma unroll
            for (int i = 0; i < FCO; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wlane + s * 4 * COUT * 16 + i * 256);
#pragma unroll
            for (int f = 0; f < FPX; ++f) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(xs + ((f >> 1) * HWD + 16 * (f & 1)) * 16);
#pragma unroll
                for (int i = 0; i < FCO; ++i) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][f], 0, 0, 0);
            }
        }

        // ---- epilogue: lane holds channels co0..co0+3 (co0 = 16*i + 4*kg) of pixel (oy, ox) ------
        const int rem = (int)(t - (int64_t)n * tiles_per_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        // Output slots of this lane: NE pixels x FCO channel quads.  Unpooled: slot e = fragment f.  Pooled: the
        // 2x2 average has its row pair in this lane (fragments f, f+2) and its column pair in lane^1.
        const bool pool = p.fu.pool != 0;
        const int OH = pool ? p.Ho >> 1 : p.Ho, OW = pool ? p.Wo >> 1 : p.Wo;
        const float gain = pool ? 0.25f * p.fu.gain : p.fu.gain;
        const bf16* __restrict__ aux = (const bf16*)p.fu.aux;
        // Batches of EB slots: FIRST every per-pixel global load of the batch (noise, mask), THEN the stores, so
        // the wait for those loads is paid once per batch and never sits between two stores.
        constexpr int EB = FCO >= 4 ? 1 : 4 / FCO;
#pragma unroll
        for (int e0 = 0; e0 < FPX; e0 += EB) {
            float nz[EB];
            uint2 ax[EB][FCO];
            int mm[EB];
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const int e = e0 + k;
                int oy, ox;
                bool live;
                if (pool) {
                    oy = (ty * TH + wave * RPW) / 2 + (e >> 1);
                    ox = tx * 16 + 8 * (e & 1) + (lpx >> 1);
                    live = e < RPW && (lpx & 1) == 0;
                } else {
                    oy = ty * TH + wave * RPW + (e >> 1);
                    ox = tx * 32 + 16 * (e & 1) + lpx;
                    live = true;
                }
                live = live && oy < OH && ox < OW;
                mm[k] = live ? (n * OH + oy) * OW + ox : -1;
                const int64_t ms = live ? mm[k] : 0;
                nz[k] = noise ? (float)noise[ms] : 0.f;
                if (aux && vec_store) {
#pragma unroll
                    for (int i = 0; i < FCO; ++i) {
                        const int co0 = i * 16 + kg * 4;
                        ax[k][i] = *reinterpret_cast<const uint2*>(aux + (co0 < p.Cout ? ms * p.Cout + co0 : 0));
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const int e = e0 + k;
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    const int co0 = i * 16 + kg * 4;
                    float v[4];
                    if (pool) {
                        // executed by every lane (the shuffle needs its partner); fragments past RPW are not slots
                        const int fa = e < RPW ? 4 * (e >> 1) + (e & 1) : 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = acc[i][fa][j] + acc[i][fa + 2][j];
                            v[j] = (a + __shfl_xor(a, 1)) * gain;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[i][e][j] * gain;
                    }
                    const int64_t m = mm[k];
                    if (mm[k] < 0 || co0 >= p.Cout) continue;
                    const float4 b4 = *reinterpret_cast<const float4*>(pl + co0);
                    const float4 s4 = *reinterpret_cast<const float4*>(pl + COUT + co0);
                    const float4 d4 = *reinterpret_cast<const float4*>(pl + 2 * COUT + co0);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
                    const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = lrelu_f(fmaf(nz[k], ss[j], fmaf(v[j], dd[j], bb[j])), p.slope);
                    if (vec_store) {
                        if (aux) {
                            const bf16x4 a4 = *reinterpret_cast<const bf16x4*>(&ax[k][i]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= lrelu_mask((float)a4[j], p.fu.aux_slope);
                        }
                        bf16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
                        *reinterpret_cast<bf16x4*>(y + m * p.Cout + co0) = o;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = co0 + j;
                            if (co >= p.Cout) continue;
                            float u = v[j];
                            if (aux) u *= lrelu_mask((float)aux[m * p.Cout + co], p.fu.aux_slope);
                            y[m * p.Cout + co] = (bf16)u;
                        }
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int KS, int RPW>
int launch_thin(ThinArgs a, hipStream_t st) {
    typedef ThinCfg<CIN, COUT, KS, RPW> C;
    static_assert(C::LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + C::TH - 1) / C::TH;
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N;
    int per_cu = (160 * 1024) / C::LDS_BYTES;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int64_t grid = 256LL * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((conv_thin_k<CIN, COUT, KS, RPW>), dim3((unsigned)grid), dim3(256), 0, st, a);
    GAN_CHECK_LAUNCH("conv2d_fwd(thin)");
    return GAN_OK;
}

template <int CIN, int KS, int RPW>
int pick_cout(const ThinArgs& a, hipStream_t st) {
    if (a.Cout <= 16) return launch_thin<CIN, 16, KS, RPW>(a, st);
    if (a.Cout <= 32) return launch_thin<CIN, 32, KS, RPW>(a, st);
    return launch_thin<CIN, 64, KS, RPW>(a, st);
}

template <int KS>
int pick_cin(const ThinArgs& a, int Cin, hipStream_t st) {
    switch (Cin) {
        case 8: return pick_cout<8, KS, 4>(a, st);
        case 16: return pick_cout<16, KS, 4>(a, st);
        case 32: return pick_cout<32, KS, 2>(a, st);
        default: return pick_cout<64, KS, 2>(a, st);
    }
}

}  // namespace

// Returns 1 if the shape is served by the thin kernel (and enqueues it), 0 if not, < 0 on error.
int gan_conv2d_fwd_thin(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                        const float* bias, const void* noise, const float* noise_scale, float slope, int N, int H,
                        int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad,
                        const ConvFuse& fu, hipStream_t st) {
    if (kh != kw || (kh != 1 && kh != 3)) return 0;
    if (py != (kh - 1) / 2 || px != py) return 0;
    if (Cin != 8 && Cin != 16 && Cin != 32 && Cin != 64) return 0;
    if (Cout > 64 || Wo < 32 || Ho < 8) return 0;
    if (Kpad != (kh * kw * Cin + 31) / 32 * 32) return 0;
    if (((uintptr_t)y % 8) != 0 || ((uintptr_t)fu.aux % 8) != 0) return 0;
    ThinArgs a;
    a.fu = fu;
    a.x = x; a.wt = wt; a.y = y;
    a.in_scale = in_scale; a.out_scale = out_scale; a.bias = bias;
    a.noise = noise; a.noise_scale = noise_scale; a.slope = slope;
    a.N = N; a.H = H; a.W = W; a.Hl = up2 ? 2 * H : H; a.Wl = up2 ? 2 * W : W; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.py = py; a.px = px; a.Kpad = Kpad;
    a.tiles_x = a.tiles_y = 0;
    const int rc = kh == 3 ? pick_cin<3>(a, Cin, st) : pick_cin<1>(a, Cin, st);
    return rc == GAN_OK ? 1 : rc;
}
