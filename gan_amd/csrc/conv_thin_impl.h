// Direct convolution for THIN layers on MFMA (gfx950): Cin, Cout <= 64 at large spatial size, bf16.
// Included by conv_thin_{g,d,da,dp}.hip, one translation unit per MODE (see ThinMode below).
//
// These are the 128x128 / 256x256 layers of the StyleGAN2 generator and discriminator
// (modelling/stylegan2.py:29-36, modelling/progressive_gan.py:45-50,61): 16-64 channels, 65 K - 16 M
// pixels.  Their arithmetic intensity (K = 9*Cin = 144..576 MACs per output element) puts them
// on the HBM roofline, not the MFMA one, so the kernel is organised around bytes and latency:
//
//   * A workgroup (4 waves) owns a TH x 32 tile of output pixels.  The (TH+2) x 34 input halo is
//     read from HBM ONCE with 16-byte loads (nearest-2x upsampling, the activation that precedes the
//     convolution and the per-sample style modulation are applied on the way), and parked in LDS as
//     [8-channel chunk][pixel][8] so that an MFMA operand (8 consecutive channels of 16 consecutive
//     pixels) is one conflict-free ds_read_b128 for every tap -- the 9 taps re-read LDS, never HBM/L2.
//   * The whole weight matrix [Cout][9*Cin] lives in LDS for the lifetime of the workgroup; a
//     workgroup walks a contiguous range of tiles (persistent), and the halo of tile t+1 is in
//     flight in registers while tile t is on the matrix cores.
//   * Operand roles are swapped relative to the big-GEMM kernel: A = weights (rows = co),
//     B = pixels (cols), so each lane ends up with 4 CONSECUTIVE output channels of one pixel
//     and the epilogue stores 8 bytes per lane, 512 contiguous bytes per wave instruction for Cout = 16.
//   * Every global access is a raw BUFFER load/store whose offset is pushed out of range for lanes past
//     the image (the hardware returns zeros / drops the store): no branch sits between memory
//     instructions, so the compiler counts them exactly and the wait for tile t+1's halo at the top of
//     the loop is vmcnt(<stores issued after it>) instead of vmcnt(0).  With `if (ok)` around them the
//     wave waited for its own 8 output stores to reach memory before every tile (PMC: 54 % of the wave
//     cycles parked in s_waitcnt at 2 waves/SIMD).
//   * One kernel per MODE: the generator's modulated convolution and the discriminator's convolutions
//     use disjoint sets of fused stages; compiling them apart keeps the registers (occupancy) of each.
#pragma once
#include <stdlib.h>

#include "common.h"

enum ThinMode {
    THIN_G = 0,   // in_scale, out_scale, bias, noise, slope            (ModulatedConv2d, stylegan2.py:29-36)
    THIN_D = 1,   // in_slope, bias, slope, gain                        (LeakyReLU -> Conv2d, progressive_gan.py:45-48)
    THIN_DA = 2,  // THIN_D + leaky-ReLU mask of `aux` on the result, + resid  (input gradients of the above)
    THIN_DP = 3,  // THIN_D + 2x2 average pooling of the result         (LeakyReLU -> Conv2d -> AvgPool2d, :48-50)
    THIN_DPR = 4, // THIN_DP + resid (pooled resolution) added last: the block's `shortcut + layers * scale` (:54) with the
                  // scale folded into the weights, so the residual merge costs no pass of its own
    THIN_DS = 5,  // the modulated convolution's input gradient (stylegan2.py:29-36 backward): z = conv * gain,
                  // dot_out[n][co] += sum_pixels z * dot_x, y = z * out_scale[n][co] -- the `scale_dot` pass on the accumulators
    THIN_DPS = 6, // THIN_DS behind the 2x2 pooling of the result (the layer had nearest-2x upsampling in front)
    THIN_DAS = 7, // THIN_DA whose residual is computed here: + sc_gain * NearestUp2( conv1x1(sc_x, sc_w) ), sc_x at half
                  // resolution with 2 * COUT channels -- the input gradient of the D block's shortcut (AvgPool2d -> Conv2d 1x1,
                  // progressive_gan.py:52) as extra K steps of the input-gradient convolution of conv1, instead of a kernel of
                  // its own writing a full-resolution tensor that this one reads back
};

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
    unsigned x_bytes, y_bytes;  // < 2^31: 32-bit buffer offsets
    unsigned sc_bytes;          // bytes of fu.sc_x
    ConvFuse fu;
};

int thin_launch_g(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_d(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_da(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_dp(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_dpr(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_ds(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_dps(const ThinArgs& a, int Cin, int ks, hipStream_t st);
int thin_launch_das(const ThinArgs& a, int Cin, int ks, hipStream_t st);

#ifdef THIN_MODE
namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr unsigned THIN_INV = 0x80000000u;  // buffer offset of a lane that must not touch memory

__device__ __forceinline__ __amdgpu_buffer_rsrc_t thin_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? bytes : 0u, 0x00020000);
}

template <int CIN, int COUT, int KS, int RPW, int NW = 4>
struct ThinCfg {
    static constexpr int NT = 64 * NW;          // threads (NW waves, RPW output rows each)
    static constexpr int CP = CIN / 8;          // 16-byte channel chunks per pixel
    static constexpr int TH = NW * RPW, TW = 32; // output tile
    static constexpr int HH = TH + KS - 1, HWD = TW + KS - 1, NPIX = HH * HWD;
    static constexpr int PLANE = (NPIX * 16 + 255) / 256 * 256;  // bytes per chunk plane, 256-B multiple
    static constexpr int KTOT = KS * KS * CIN, NSTEP = (KTOT + 31) / 32, KPAD = NSTEP * 32;
    static constexpr int FCO = COUT / 16, FPX = 2 * RPW;
    static constexpr int NCHUNK = ((NPIX + 7) / 8) * 8 * CP;
    static constexpr int NLD = (NCHUNK + NT - 1) / NT;
    static constexpr int X_BYTES = CP * PLANE;
    static constexpr int W_BYTES = KPAD * COUT * 2;
    static constexpr int P_BYTES = 3 * COUT * 4;  // bias, noise scale, this tile's demodulation row (fp32)
    static constexpr int D_BYTES = NT * 16;       // per-thread dump slot for staging chunks past the halo
    static constexpr int LDS_BYTES = X_BYTES + W_BYTES + P_BYTES + D_BYTES;
};

// VEC: Cout % 4 == 0, 8-byte stores; !VEC (to-/from-RGB, Cout = 3): 2-byte stores.  A template parameter, not
// a branch: an if/else with different stores on its two sides makes hipcc forget how many memory
// instructions are in flight (it then waits for all of them).
template <int CIN, int COUT, int KS, int RPW, int MODE, bool VEC, int NW = 4>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) conv_thin_k(const ThinArgs p) {
    typedef ThinCfg<CIN, COUT, KS, RPW, NW> C;
    constexpr int NT = C::NT;
    constexpr int CP = C::CP, TH = C::TH, HWD = C::HWD, NPIX = C::NPIX, PLANE = C::PLANE;
    constexpr int NSTEP = C::NSTEP, KPAD = C::KPAD, FCO = C::FCO, FPX = C::FPX, NLD = C::NLD;
    constexpr bool DAS = MODE == THIN_DAS;
    constexpr bool G = MODE == THIN_G, AUX = MODE == THIN_DA || DAS, POOL = MODE == THIN_DP || MODE == THIN_DPR || MODE == THIN_DPS, RES = MODE == THIN_DPR;
    constexpr bool DOT = MODE == THIN_DS || MODE == THIN_DPS;

    // THIN_DAS: the shortcut operand's tile (TH/2 x 16 source pixels x 2 COUT channels, [8-channel chunk][pixel][8] like the
    // halo) and its weights ([k/8][co][8])
    constexpr int SCC = 2 * COUT, CPS = SCC / 8, SPIX = (TH / 2) * 16, SPLANE = SPIX * 16;
    constexpr int S_BYTES = DAS ? CPS * SPLANE : 0, SW_BYTES = DAS ? SCC * COUT * 2 : 0;
    constexpr int NSL = DAS ? (SPIX * CPS + NT - 1) / NT : 1;
    static_assert(!DAS || (SPIX % 8 == 0 && NT % (8 * CPS) == 0), "shortcut tile staging");
    __shared__ __attribute__((aligned(256))) char lds[C::LDS_BYTES + S_BYTES + SW_BYTES];
    char* const sl = lds + C::LDS_BYTES;
    char* const swl = lds + C::LDS_BYTES + S_BYTES;
    char* const xl = lds;
    char* const wl = lds + C::X_BYTES;
    // per-channel epilogue vectors in LDS (ds_read: they never sit on the vector-memory counter)
    float* const pl = reinterpret_cast<float*>(lds + C::X_BYTES + C::W_BYTES);
    char* const dump = lds + C::X_BYTES + C::W_BYTES + C::P_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bf16* __restrict__ wt = (const bf16*)p.wt;

    // ---- weights -> LDS as [k/8][co][8]; rows co >= Cout are zero ------------------------------
    for (int idx = tid; idx < (KPAD / 8) * COUT; idx += NT) {
        const int co = idx % COUT;
        const int kc = idx / COUT;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (co < p.Cout) v = *reinterpret_cast<const uint4*>(wt + (int64_t)co * p.Kpad + kc * 8);
        *reinterpret_cast<uint4*>(wl + idx * 16) = v;
    }
    if (DAS) {
        const bf16* __restrict__ scw = (const bf16*)p.fu.sc_w;
        for (int idx = tid; idx < CPS * COUT; idx += NT) {
            const int co = idx % COUT, kc = idx / COUT;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (co < p.Cout) v = *reinterpret_cast<const uint4*>(scw + (int64_t)co * SCC + kc * 8);
            *reinterpret_cast<uint4*>(swl + idx * 16) = v;
        }
    }
    if (tid < COUT) {
        pl[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
        pl[COUT + tid] = (G && p.noise_scale && tid < p.Cout) ? p.noise_scale[tid] : 0.f;
        pl[2 * COUT + tid] = 1.f;
    }

    const __amdgpu_buffer_rsrc_t rx = thin_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ry = thin_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t raux = thin_rsrc(AUX ? p.fu.aux : (DOT ? p.fu.dot_x : nullptr), p.y_bytes);
    const __amdgpu_buffer_rsrc_t rres = thin_rsrc(((AUX && !DAS) || RES) ? p.fu.resid : nullptr, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rsc = thin_rsrc(DAS ? p.fu.sc_x : nullptr, p.sc_bytes);
    const __amdgpu_buffer_rsrc_t rnoise = thin_rsrc(G ? p.noise : nullptr, p.Cout ? p.y_bytes / (unsigned)p.Cout : 0u);
    const __amdgpu_buffer_rsrc_t rosc = thin_rsrc((G || DOT) ? p.out_scale : nullptr, (unsigned)(p.N * p.Cout) * 4u);
    const __amdgpu_buffer_rsrc_t rsty = thin_rsrc(G ? p.in_scale : nullptr, (unsigned)(p.N * CIN) * 4u);

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.N;
    const int t0 = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t1 = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);

    // ---- halo loader: chunk idx -> (pixel, channel chunk); 8 consecutive lanes = 8 consecutive
    // pixels of one chunk plane (conflict-free ds_write_b128), 8*CP lanes = one contiguous run.
    // NT % (8*CP) == 0, so a thread's channel chunk c is the same for all of its slots. ----------
    const int cch = (tid % (8 * CP)) >> 3;
    int hyx[NLD];  // (hy << 16) | hx of slot i, or -1 past the halo
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = i * NT + tid;
        const int pix = (idx / (8 * CP)) * 8 + (idx & 7);
        const int hy = pix / HWD, hx = pix - hy * HWD;
        hyx[i] = pix < NPIX ? (hy << 16) | hx : -1;
    }
    v4u reg[NLD];
    v4u sreg[NSL];
    int spc[NSL];  // THIN_DAS: (pixel << 8) | chunk of shortcut-tile slot i, -1 past the tile (zeros into the dump slot)
    if (DAS) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int idx = i * NT + tid;
            const int pix = (idx / (8 * CPS)) * 8 + (idx & 7);
            spc[i] = pix < SPIX ? (pix << 8) | ((idx % (8 * CPS)) >> 3) : -1;
        }
    }
    float4 sty0 = make_float4(1.f, 1.f, 1.f, 1.f), sty1 = sty0;
    float osc_reg = 1.f;
    auto issue = [&](int t) {
        const int n = t / tiles_per_img;
        const int rem = t - n * tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int ly0 = ty * TH - p.py, lx0 = tx * 32 - p.px;
        if (G || DOT)
            osc_reg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rosc, tid < p.Cout ? (n * p.Cout + tid) * 4 : THIN_INV, 0, 0));
        if (G) {
            const v4u s0 = __builtin_amdgcn_raw_buffer_load_b128(rsty, (n * CIN + cch * 8) * 4, 0, 0);
            const v4u s1 = __builtin_amdgcn_raw_buffer_load_b128(rsty, (n * CIN + cch * 8 + 4) * 4, 0, 0);
            sty0 = __builtin_bit_cast(float4, s0);
            sty1 = __builtin_bit_cast(float4, s1);
        }
        if (DAS) {
            const int Hs = p.Ho >> 1, Ws = p.Wo >> 1;
            const int sy0 = ty * (TH / 2), sx0 = tx * 16;
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int pix = spc[i] >> 8, c = spc[i] & 0xff;
                const int sy = sy0 + (pix >> 4), sx = sx0 + (pix & 15);
                const bool ok = spc[i] >= 0 && sy < Hs && sx < Ws;
                sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsc, ok ? (unsigned)((((n * Hs + sy) * Ws + sx) * SCC + c * 8) * 2) : THIN_INV, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ly = ly0 + (hyx[i] >> 16), lx = lx0 + (hyx[i] & 0xffff);
            const bool ok = hyx[i] >= 0 && (unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl;
            const unsigned off = (unsigned)(((n * p.H + (ly >> p.up2)) * p.W + (lx >> p.up2)) * (CIN * 2) + cch * 16);
#ifdef THIN_ABL_NOLOAD  // diagnostic build: no input traffic
            reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (ok && off == 0x7ffffff0u) ? off : THIN_INV, 0, 0);
#else
            reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : THIN_INV, 0, 0);
#endif
        }
    };
    auto commit = [&]() {
        if ((G || DOT) && tid < COUT) pl[2 * COUT + tid] = p.out_scale ? osc_reg : 1.f;
        if (DAS) {
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                char* const dst = spc[i] >= 0 ? sl + (spc[i] & 0xff) * SPLANE + (spc[i] >> 8) * 16 : dump + tid * 16;
                *reinterpret_cast<v4u*>(dst) = sreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = i * NT + tid;
            const int pix = (idx / (8 * CP)) * 8 + (idx & 7);
            uint4 v = __builtin_bit_cast(uint4, reg[i]);
            if (!G && p.fu.in_slope != 1.f) v = lrelu_chunk(v, p.fu.in_slope, bf16());
            if (G && p.in_scale) {
                bf16x8 h = *reinterpret_cast<bf16x8*>(&v);
                h[0] = (bf16)((float)h[0] * sty0.x); h[1] = (bf16)((float)h[1] * sty0.y);
                h[2] = (bf16)((float)h[2] * sty0.z); h[3] = (bf16)((float)h[3] * sty0.w);
                h[4] = (bf16)((float)h[4] * sty1.x); h[5] = (bf16)((float)h[5] * sty1.y);
                h[6] = (bf16)((float)h[6] * sty1.z); h[7] = (bf16)((float)h[7] * sty1.w);
                v = *reinterpret_cast<uint4*>(&h);
            }
            char* const dst = pix < NPIX ? xl + cch * PLANE + pix * 16 : dump + tid * 16;
            *reinterpret_cast<uint4*>(dst) = v;
        }
    };

    const int lpx = lane & 15;
    const int kg = lane >> 4;
    const char* const wlane = wl + (kg * COUT + lpx) * 16;
    const char* const xlane = xl + (wave * RPW * HWD + lpx) * 16;
    const int OH = POOL ? p.Ho >> 1 : p.Ho, OW = POOL ? p.Wo >> 1 : p.Wo;
    const float gain = POOL ? 0.25f * p.fu.gain : p.fu.gain;

    // THIN_DS / THIN_DPS: per-lane partial sums of z * dot_x for the lane's 4 x FCO channels, carried across the tiles of one
    // image (a workgroup's tiles are consecutive: it meets at most a few images) and flushed -- folded over the 16 lanes that
    // share the channels, then one fp32 atomic per channel and wave -- when the image changes and at the end
    float dacc[DOT ? FCO : 1][4];
#pragma unroll
    for (int i = 0; i < (DOT ? FCO : 1); ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dacc[i][j] = 0.f;
    int n_acc = -1;
    auto flush_dot = [&](int nn) {
#pragma unroll
        for (int i = 0; i < (DOT ? FCO : 1); ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = dacc[i][j];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                const int co = i * 16 + kg * 4 + j;
                if (lpx == 0 && co < p.Cout) atomicAdd(p.fu.dot_out + (int64_t)nn * p.Cout + co, v);
                dacc[i][j] = 0.f;
            }
    };

    // The wait for a halo and every memory instruction issued after its loads sit in the SAME iteration
    // (issue -> compute -> stores -> commit), so the compiler's count is exact; with the commit at the top
    // of the loop the count was merged with the first iteration's (no stores yet) and degraded to "all".
    if (t0 < t1) {
        issue(t0);
        commit();
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        issue(t + 1 < t1 ? t + 1 : t);  // next halo in flight while this tile computes (last tile: a harmless re-read)

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
#ifdef THIN_ABL_ONESTEP  // diagnostic build: one K step per tile (no MFMA / fragment-read phase to speak of)
            break;
#endif
        }

        // ---- THIN_DAS: the shortcut's 1x1 transposed convolution as SCC / 32 more K steps into a second set of accumulators
        // (the mask applies to the main path only).  Fragment pixel lpx of row r reads source pixel (r / 2, x / 2).
        f32x4 acc2[DAS ? FCO : 1][DAS ? FPX : 1];
        if (DAS) {
#pragma unroll
            for (int i = 0; i < FCO; ++i)
#pragma unroll
                for (int f = 0; f < FPX; ++f) acc2[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ss = 0; ss < SCC / 32; ++ss) {
                bf16x8 a2[FCO];
#pragma unroll
                for (int i = 0; i < FCO; ++i) a2[i] = *reinterpret_cast<const bf16x8*>(swl + ((ss * 4 + kg) * COUT + i * 16 + lpx) * 16);
#pragma unroll
                for (int f = 0; f < FPX; ++f) {
                    const int r = wave * RPW + (f >> 1), xl = 16 * (f & 1) + lpx;
                    const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(sl + (ss * 4 + kg) * SPLANE + ((r >> 1) * 16 + (xl >> 1)) * 16);
#pragma unroll
                    for (int i = 0; i < FCO; ++i) acc2[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[i], b2, acc2[i][f], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: lane holds channels co0..co0+3 (co0 = 16*i + 4*kg) of one pixel per slot --------
        // Unpooled: slot e = fragment e.  Pooled: the 2x2 window has its row pair in this lane (fragments
        // f, f+2) and its column pair in lane^1; even lanes store.
        const int n = t / tiles_per_img;
        const int rem = t - n * tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        constexpr int NE = POOL ? RPW : FPX;
        if (DOT) {
            if (n != n_acc && n_acc >= 0) flush_dot(n_acc);  // (wave-uniform)
            n_acc = n;
        }
        // batches of EB slots: first the batch's per-pixel loads (noise / mask), then its stores
        constexpr int EB = (G || AUX || RES || DOT) ? (FCO >= 4 ? 1 : (4 / FCO < NE ? 4 / FCO : NE)) : NE;
#pragma unroll
        for (int e0 = 0; e0 < NE; e0 += EB) {
            float nz[EB];
            v2u ax[EB][FCO], rs[EB][FCO];
            unsigned mo[EB];  // byte offset of the pixel's channel 0 in y (and aux), THIN_INV for a dead slot
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const int e = e0 + k;
                int oy, ox;
                bool live;
                if (POOL) {
                    oy = (ty * TH + wave * RPW) / 2 + (e >> 1);
                    ox = tx * 16 + 8 * (e & 1) + (lpx >> 1);
                    live = (lpx & 1) == 0;
                } else {
                    oy = ty * TH + wave * RPW + (e >> 1);
                    ox = tx * 32 + 16 * (e & 1) + lpx;
                    live = true;
                }
                live = live && oy < OH && ox < OW;
                const unsigned m = (unsigned)((n * OH + oy) * OW + ox);
                mo[k] = live ? m * (unsigned)(p.Cout * 2) : THIN_INV;
                nz[k] = 0.f;
                if (G) {
                    const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rnoise, live ? m * 2u : THIN_INV, 0, 0);
                    nz[k] = (float)__builtin_bit_cast(bf16, h);
                }
                if (AUX) {
#pragma unroll
                    for (int i = 0; i < FCO; ++i) {
                        const int co0 = i * 16 + kg * 4;
                        ax[k][i] = __builtin_amdgcn_raw_buffer_load_b64(raux, (live && co0 < p.Cout) ? mo[k] + co0 * 2 : THIN_INV, 0, 0);
                        if (!DAS) rs[k][i] = __builtin_amdgcn_raw_buffer_load_b64(rres, (live && co0 < p.Cout) ? mo[k] + co0 * 2 : THIN_INV, 0, 0);
                    }
                }
                if (RES) {
#pragma unroll
                    for (int i = 0; i < FCO; ++i) {
                        const int co0 = i * 16 + kg * 4;
                        rs[k][i] = __builtin_amdgcn_raw_buffer_load_b64(rres, (live && co0 < p.Cout) ? mo[k] + co0 * 2 : THIN_INV, 0, 0);
                    }
                }
                if (DOT) {  // (dead slots and channels read zeros: they add nothing to the sums)
#pragma unroll
                    for (int i = 0; i < FCO; ++i) {
                        const int co0 = i * 16 + kg * 4;
                        ax[k][i] = __builtin_amdgcn_raw_buffer_load_b64(raux, (live && co0 < p.Cout) ? mo[k] + co0 * 2 : THIN_INV, 0, 0);
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
                    if (POOL) {
                        const int fa = 4 * (e >> 1) + (e & 1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = acc[i][fa][j] + acc[i][fa + 2][j];
                            v[j] = (a + quad_xor1(a)) * gain;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = G ? acc[i][e][j] : acc[i][e][j] * gain;
                    }
                    const float4 b4 = *reinterpret_cast<const float4*>(pl + co0);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                    if (G) {
                        const float4 s4 = *reinterpret_cast<const float4*>(pl + COUT + co0);
                        const float4 d4 = *reinterpret_cast<const float4*>(pl + 2 * COUT + co0);
                        const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaf(nz[k], ss[j], fmaf(v[j], dd[j], bb[j]));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += bb[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = lrelu_f(v[j], p.slope);
                    if (AUX) {
                        const bf16x4 a4 = __builtin_bit_cast(bf16x4, ax[k][i]);
                        if (DAS) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], lrelu_mask((float)a4[j], p.fu.aux_slope), acc2[i][e][j] * p.fu.sc_gain);
                        } else {
                            const bf16x4 r4 = __builtin_bit_cast(bf16x4, rs[k][i]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], lrelu_mask((float)a4[j], p.fu.aux_slope), (float)r4[j]);  // resid: zeros if absent
                        }
                    }
                    if (RES) {
                        const bf16x4 r4 = __builtin_bit_cast(bf16x4, rs[k][i]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)r4[j];
                    }
                    if (DOT) {
                        const bf16x4 a4 = __builtin_bit_cast(bf16x4, ax[k][i]);
                        const float4 d4 = *reinterpret_cast<const float4*>(pl + 2 * COUT + co0);
                        const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            dacc[i][j] = fmaf(v[j], (float)a4[j], dacc[i][j]);
                            v[j] *= dd[j];
                        }
                    }
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
                    if (VEC) {
#ifdef THIN_ABL_NOSTORE  // diagnostic build: no output traffic
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), ry, (co0 < p.Cout && o[0] == (bf16)12345.f && o[1] == (bf16)777.f) ? mo[k] + co0 * 2 : THIN_INV, 0, 0);
#else
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), ry, co0 < p.Cout ? mo[k] + co0 * 2 : THIN_INV, 0, 0);
#endif
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bf16 oj = (bf16)v[j];  // (bit_cast of a vector ELEMENT stored lane 0's value four times)
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, oj), ry,
                                                                  co0 + j < p.Cout ? mo[k] + (co0 + j) * 2 : THIN_INV, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();  // every wave has read its operands of tile t
        commit();         // waits for this iteration's halo loads only: the stores above stay in flight
        __syncthreads();
    }
    if (DOT && n_acc >= 0) flush_dot(n_acc);
}

template <int CIN, int COUT, int KS, int RPW, bool VEC, int NW = 4>
int launch_thin_v(ThinArgs a, hipStream_t st) {
    typedef ThinCfg<CIN, COUT, KS, RPW, NW> C;
    static_assert(C::LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + C::TH - 1) / C::TH;
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N;
    // resident workgroups per CU of THIS instantiation (registers and LDS), asked once
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_thin_k<CIN, COUT, KS, RPW, THIN_MODE, VEC, NW>, C::NT, 0) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb > 8 ? 8 : nb;
    }
    int64_t grid = 256LL * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((conv_thin_k<CIN, COUT, KS, RPW, THIN_MODE, VEC, NW>), dim3((unsigned)grid), dim3(C::NT), 0, st, a);
    GAN_CHECK_LAUNCH("conv2d_fwd(thin)");
    return GAN_OK;
}

template <int CIN, int COUT, int KS, int RPW>
int launch_thin(const ThinArgs& a, hipStream_t st) {
    if ((a.Cout & 3) == 0) return launch_thin_v<CIN, COUT, KS, RPW, true>(a, st);
    if constexpr (COUT == 16 && THIN_MODE != THIN_DP && THIN_MODE != THIN_DPR && THIN_MODE != THIN_DS && THIN_MODE != THIN_DPS && THIN_MODE != THIN_DAS)
        return launch_thin_v<CIN, COUT, KS, RPW, false>(a, st);
    gan_set_error("conv2d_fwd(thin): Cout=%d not served", a.Cout);  // the dispatcher filters these shapes
    return GAN_ERR_ARG;
}

template <int CIN, int KS, int RPW>
int pick_cout(const ThinArgs& a, hipStream_t st) {
    if (a.Cout <= 16) return launch_thin<CIN, 16, KS, RPW>(a, st);
    if (a.Cout <= 32) return launch_thin<CIN, 32, KS, RPW>(a, st);
    if constexpr (THIN_MODE == THIN_DAS && (CIN != 64 || KS != 3)) {
        // (64 output channels: the shortcut tile has 128 channels -- 32 KB more LDS, instantiated for the one layer that has it)
        gan_set_error("conv2d_fwd(thin, shortcut): Cout=%d not served", a.Cout);
        return GAN_ERR_ARG;
    } else {
        return launch_thin<CIN, 64, KS, RPW>(a, st);
    }
}

template <int KS>
int pick_cin(const ThinArgs& a, int Cin, hipStream_t st) {
    switch (Cin) {
        case 8: return pick_cout<8, KS, 4>(a, st);
        case 16: return pick_cout<16, KS, 4>(a, st);
        case 32: return pick_cout<32, KS, 2>(a, st);
        default:
            // 64 input channels: with 8-row tiles the halo + weights are 85 KB (Cout <= 32: one workgroup per CU); 4-row
            // tiles bring that to 67 KB = two per CU (their load and MFMA phases overlap).  Cout = 64 stays at 8 rows: its
            // 74 KB weight image alone pins it to one workgroup per CU either way.
            // (not the pooled mode: a wave pools the row pair it owns)
            if constexpr (KS == 3 && THIN_MODE != THIN_DP && THIN_MODE != THIN_DPR && THIN_MODE != THIN_DPS) {
                if (a.Cout <= 32 && getenv("GAN_AMD_THIN64_RPW2") == nullptr) return pick_cout<64, KS, 1>(a, st);
            }
            // Cout = 64, 3 x 3: the 74 KB weight image leaves one workgroup per CU -- as 4 waves that is ONE wave per SIMD (PMC:
            // MFMA pipe 33 % busy).  Measured in round 3 and NOT the default (GAN_AMD_THIN64_W8=1 selects it): 8 waves x 2 rows =
            // 16 x 32 tiles, halo 78 KB + weights 74 KB, two waves per SIMD: 102.9 -> 99.0 us on 64 -> 64 @ 64^2, batch 256
            // (pooled 85.4 -> 79.3), the train step unchanged (48.85 / 48.95 ms, same box) -- the kernel's LDS reads and its
            // MFMAs each take ~31 us at 2.4 GHz and the second wave does not make them overlap.  16-row tiles on 4 waves
            // (RPW = 4): 172 us.
            if constexpr (KS == 3 && THIN_MODE != THIN_DAS) {
                static const bool w8 = getenv("GAN_AMD_THIN64_W8") && atoi(getenv("GAN_AMD_THIN64_W8")) == 1;
                if (w8 && a.Cout > 32 && (a.Cout & 3) == 0) return launch_thin_v<64, 64, KS, 2, true, 8>(a, st);
            }
            return pick_cout<64, KS, 2>(a, st);
    }
}

}  // namespace

#define THIN_ENTRY(name)                                                     \
    int name(const ThinArgs& a, int Cin, int ks, hipStream_t st) {           \
        return ks == 3 ? pick_cin<3>(a, Cin, st) : pick_cin<1>(a, Cin, st);  \
    }
#endif  // THIN_MODE
