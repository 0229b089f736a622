// The image boundary of the discriminator (gfx950): nn.Conv2d(img_channels = 3, base_dim, 1) (progressive_gan.py:61)
// and its two adjoints, as HBM-streaming kernels that read / write the 3-channel tensor as it is.
//
// Round 1 padded the image to 16 channels (gan_pad_channels: 5.3x its bytes written) and ran the generic thin kernels on
// the padded copy: per discriminator pass at 256 x 256, batch 256 that moved 1.7 GB where 0.7 GB is compulsory, and the
// fp32 reals took a cast + layout pass in front.  Here
//   fwd  : y[n][p][co]  = act( sum_c x[n, c, p] * W[co][c] + b[co] )        x: any strides, fp32 or bf16; y: [N][HW][Cout]
//   dgrad: gx[n][p][c]  = sum_co gy[n][p][co] * W[co][c]                      gx: [N][HW][Cin] (channels-last, Cin <= 4)
//   wgrad: dw[co][c]    = sum_{n,p} gy[n][p][co] * x[n, c, p]                 fp32, two stages through a caller workspace
// All three are bilinear in their two tensor arguments and closed under differentiation (fwd <-> dgrad <-> wgrad), so
// R1's double backward through the first layer runs on them too.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int RTPB = 256;
constexpr int RGB_MAXC = 4;       // image channels
constexpr int RGB_WG_BLOCKS = 1024;
constexpr unsigned RGB_FWD_PX = 4;  // pixels per thread of the forward kernel
constexpr int RGB_DG_IT = 1;        // pixel groups per workgroup of the input-gradient kernel (4: 171 -> 180 us at 256 x 256, batch 256)
constexpr int RGB_DG16_IT = 2;      // pixels per thread of its 16-channel variant

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

// Element offset of pixel p (0 .. HW) of sample n in the strided image.  `lin`: sh == W * sw (dense NCHW and
// channels-last both are), so the pixel index itself is the spatial offset and no division is needed.
__device__ __forceinline__ int64_t rgb_xoff(unsigned n, unsigned p, int64_t sn, int64_t sh, int64_t sw, unsigned W, bool lin) {
    if (lin) return (int64_t)n * sn + (int64_t)p * sw;
    const unsigned h = p / W, w = p - h * W;
    return (int64_t)n * sn + (int64_t)h * sh + (int64_t)w * sw;
}

// fwd: one thread = one pixel x 8 output channels; CV = Cout / 8 consecutive lanes share a pixel, so a wave's 16-byte
// stores are one contiguous run.  grid.y = sample.  Weights sit in LDS padded to 4 input channels (one ds_read_b128
// per output channel, the same address for every lane: a broadcast).
// DENSE: a dense channels-last 2-byte image (the generator's output): a workgroup's pixels are one contiguous byte run, read
// with 16-byte loads into LDS (a pixel is 6 bytes: per-lane 2-byte loads at a 6-byte stride ran at 3.3 TB/s of in + out bytes)
template <typename TI, typename TO, int CV, bool DENSE = false>
__global__ void __launch_bounds__(RTPB) rgb_fwd_k(const TI* __restrict__ x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, unsigned W, int lin,
                                                  const float* __restrict__ wgt, const float* __restrict__ bias, TO* __restrict__ y, unsigned HW,
                                                  int Cin, float slope) {
    constexpr int COUT = CV * 8;
    __shared__ __attribute__((aligned(16))) TI xl[DENSE ? RGB_FWD_PX * (RTPB / CV) * RGB_MAXC + 8 : 8];
    __shared__ float4 wl[COUT];
    __shared__ float bl[COUT];
    for (int i = threadIdx.x; i < COUT; i += RTPB) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; ++c) t[c] = wgt[i * Cin + c];
        wl[i] = make_float4(t[0], t[1], t[2], t[3]);
        bl[i] = bias ? bias[i] : 0.f;
    }
    __syncthreads();
    // a thread keeps the weights of its 8 output channels in registers and walks RGB_FWD_PX pixels
    const int c8 = (int)(threadIdx.x % CV) * 8;
    const unsigned lp = threadIdx.x / CV;
    constexpr unsigned PPB = RTPB / CV;
    float4 wr[8];
    float br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wr[j] = wl[c8 + j];
        br[j] = bl[c8 + j];
    }
    const unsigned n = blockIdx.y;
    if (DENSE) {  // this workgroup's pixels: elements [p0 * Cin, p1 * Cin) of image n, 16-byte aligned (host-checked)
        constexpr int EPV = 16 / sizeof(TI);
        const unsigned p0 = blockIdx.x * RGB_FWD_PX * PPB;
        const unsigned p1 = p0 + RGB_FWD_PX * PPB < HW ? p0 + RGB_FWD_PX * PPB : HW;
        const int nel = p1 > p0 ? (int)(p1 - p0) * Cin : 0;
        const TI* __restrict__ src = x + ((int64_t)n * HW + p0) * Cin;
        for (int i = threadIdx.x * EPV; i < nel; i += RTPB * EPV) {
            if (i + EPV <= nel) {
                *reinterpret_cast<uint4*>(xl + i) = *reinterpret_cast<const uint4*>(src + i);
            } else {
                for (int e = i; e < nel; ++e) xl[e] = src[e];
            }
        }
        __syncthreads();
    }
    // fp32 images: the values of all RGB_FWD_PX pixels first (their loads in flight together: 154 -> 140 us at 256 x 3 x 256^2),
    // then the products and the stores.  2-byte channels-last pixels are loaded pixel by pixel (batched: 192 -> 230 us).
    constexpr bool BATCH = sizeof(TI) == 4;
    float xv[RGB_FWD_PX][RGB_MAXC];
    unsigned pp[RGB_FWD_PX];
#pragma unroll
    for (unsigned it = 0; it < RGB_FWD_PX; ++it) {
        const unsigned p = (blockIdx.x * RGB_FWD_PX + it) * PPB + lp;
        pp[it] = p;
        if (BATCH && !DENSE) {
            const TI* __restrict__ xp = x + rgb_xoff(n, p < HW ? p : 0u, sn, sh, sw, W, lin);
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c) xv[it][c] = c < Cin ? ldf(xp + c * sc) : 0.f;
        }
    }
#pragma unroll
    for (unsigned it = 0; it < RGB_FWD_PX; ++it) {
        const unsigned p = pp[it];
        if (p >= HW) break;
        if (DENSE) {
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c) xv[it][c] = c < Cin ? (float)xl[(it * PPB + lp) * Cin + c] : 0.f;
        } else if (!BATCH) {
            const TI* __restrict__ xp = x + rgb_xoff(n, p, sn, sh, sw, W, lin);
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c) xv[it][c] = c < Cin ? ldf(xp + c * sc) : 0.f;
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = lrelu_f(fmaf(xv[it][0], wr[j].x, fmaf(xv[it][1], wr[j].y, fmaf(xv[it][2], wr[j].z, fmaf(xv[it][3], wr[j].w, br[j])))), slope);
        TO* __restrict__ yp = y + ((int64_t)n * HW + p) * COUT + c8;
        if constexpr (sizeof(TO) == 2) {
            store_f<TO, 8>(yp, o);
        } else {
            const float lo[4] = {o[0], o[1], o[2], o[3]}, hi[4] = {o[4], o[5], o[6], o[7]};
            store_f<TO, 4>(yp, lo);
            store_f<TO, 4>(yp + 4, hi);
        }
    }
}

// dgrad: CV consecutive lanes share a pixel, each takes one 16-byte (8-channel) chunk of its gradient -- a wave's loads
// are one contiguous run -- and the partial dot products are folded with xor-shuffles.  Results go through LDS so that
// the Cin-channel output (6 bytes per pixel for RGB in bf16) leaves as contiguous 4-byte stores.
template <typename T, int CV>
__global__ void __launch_bounds__(RTPB) rgb_dgrad_k(const T* __restrict__ gy, const float* __restrict__ wgt, T* __restrict__ gx, int64_t NP, int Cin) {
    // RGB_DG_IT pixel groups per workgroup, their loads issued before the first product (round 3: one group per workgroup
    // was 131 k workgroups of two barriers and a weight load each at 256 x 256, batch 256: 3.6 TB/s)
    constexpr int COUT = CV * 8, PPB = RTPB / CV, IT = RGB_DG_IT;  // PPB pixels per group
    __shared__ float4 wl[COUT];
    __shared__ __attribute__((aligned(16))) T ol[IT * PPB * RGB_MAXC];
    for (int i = threadIdx.x; i < COUT; i += RTPB) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; ++c) t[c] = wgt[i * Cin + c];
        wl[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
    const int lp = threadIdx.x / CV, ch = threadIdx.x % CV;
    const int64_t pix0 = (int64_t)blockIdx.x * (PPB * IT);
    float g[IT][8];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t pix = pix0 + it * PPB + lp;
#pragma unroll
        for (int j = 0; j < 8; ++j) g[it][j] = 0.f;
        if (pix < NP) {
            if constexpr (sizeof(T) == 2) {
                load_f<T, 8>(gy + pix * COUT + ch * 8, g[it]);
            } else {
                float lo[4], hi[4];
                load_f<T, 4>(gy + pix * COUT + ch * 8, lo);
                load_f<T, 4>(gy + pix * COUT + ch * 8 + 4, hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) { g[it][j] = lo[j]; g[it][4 + j] = hi[j]; }
            }
        }
    }
    __syncthreads();
    float4 wr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[j] = wl[ch * 8 + j];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        float acc[RGB_MAXC] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0] = fmaf(g[it][j], wr[j].x, acc[0]); acc[1] = fmaf(g[it][j], wr[j].y, acc[1]);
            acc[2] = fmaf(g[it][j], wr[j].z, acc[2]); acc[3] = fmaf(g[it][j], wr[j].w, acc[3]);
        }
#pragma unroll
        for (int off = 1; off < CV; off <<= 1)
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c) acc[c] += __shfl_xor(acc[c], off);
        if (ch == 0)
            for (int c = 0; c < Cin; ++c) ol[(it * PPB + lp) * Cin + c] = (T)acc[c];
    }
    __syncthreads();
    // the workgroup's IT * PPB * Cin outputs are contiguous in gx (channels-last, Cin channels)
    int64_t live = NP - pix0;
    if (live > PPB * IT) live = PPB * IT;
    const int nel = (int)live * Cin;
    T* __restrict__ dst = gx + pix0 * Cin;
    constexpr int EPW = 4 / sizeof(T);  // elements per 4-byte word
    if ((pix0 * Cin * (int64_t)sizeof(T)) % 4 == 0) {
        for (int i = threadIdx.x; i < nel / EPW; i += RTPB) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(ol)[i];
        for (int i = (nel / EPW) * EPW + threadIdx.x; i < nel; i += RTPB) dst[i] = ol[i];
    } else {
        for (int i = threadIdx.x; i < nel; i += RTPB) dst[i] = ol[i];
    }
}

// dgrad, Cout = 16 (round 3): a lane owns a whole pixel -- 32 contiguous bytes, two 16-byte loads, a wave's loads one 2 KB
// run -- so there is no cross-lane fold; RGB_DG16_IT pixels per thread with their loads issued up front.
template <typename T>
__global__ void __launch_bounds__(RTPB) rgb_dgrad16_k(const T* __restrict__ gy, const float* __restrict__ wgt, T* __restrict__ gx, int64_t NP, int Cin) {
    constexpr int IT = RGB_DG16_IT;
    __shared__ float4 wl[16];
    __shared__ __attribute__((aligned(16))) T ol[IT * RTPB * RGB_MAXC];
    if (threadIdx.x < 16) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; ++c) t[c] = wgt[threadIdx.x * Cin + c];
        wl[threadIdx.x] = make_float4(t[0], t[1], t[2], t[3]);
    }
    const int64_t pix0 = (int64_t)blockIdx.x * (RTPB * IT);
    float g[IT][16];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t pix = pix0 + it * RTPB + threadIdx.x;
#pragma unroll
        for (int j = 0; j < 16; ++j) g[it][j] = 0.f;
        if (pix < NP) {
            if constexpr (sizeof(T) == 2) {
                float lo[8], hi[8];
                load_f<T, 8>(gy + pix * 16, lo);
                load_f<T, 8>(gy + pix * 16 + 8, hi);
#pragma unroll
                for (int j = 0; j < 8; ++j) { g[it][j] = lo[j]; g[it][8 + j] = hi[j]; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v4[4];
                    load_f<T, 4>(gy + pix * 16 + 4 * q, v4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[it][4 * q + j] = v4[j];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        float acc[RGB_MAXC] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 w4 = wl[j];  // (the same address for every lane: a broadcast)
            acc[0] = fmaf(g[it][j], w4.x, acc[0]); acc[1] = fmaf(g[it][j], w4.y, acc[1]);
            acc[2] = fmaf(g[it][j], w4.z, acc[2]); acc[3] = fmaf(g[it][j], w4.w, acc[3]);
        }
        for (int c = 0; c < Cin; ++c) ol[(it * RTPB + threadIdx.x) * Cin + c] = (T)acc[c];
    }
    __syncthreads();
    int64_t live = NP - pix0;
    if (live > RTPB * IT) live = RTPB * IT;
    const int nel = (int)live * Cin;
    T* __restrict__ dst = gx + pix0 * Cin;
    constexpr int EPW = 4 / sizeof(T);
    if ((pix0 * Cin * (int64_t)sizeof(T)) % 4 == 0) {
        for (int i = threadIdx.x; i < nel / EPW; i += RTPB) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(ol)[i];
        for (int i = (nel / EPW) * EPW + threadIdx.x; i < nel; i += RTPB) dst[i] = ol[i];
    } else {
        for (int i = threadIdx.x; i < nel; i += RTPB) dst[i] = ol[i];
    }
}

// wgrad stage 1: a workgroup walks pixels p = block, block + grid, ...; a thread owns one pixel at a time and keeps Cout x Cin
// partial sums for ONE 8-channel chunk of Cout (blockIdx.y), folded over the workgroup with shuffles + LDS.
template <typename TX, typename TG, int CH>
__global__ void __launch_bounds__(RTPB) rgb_wgrad_k(const TX* __restrict__ x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, unsigned W, int lin,
                                                    const TG* __restrict__ gy, float* __restrict__ ws, unsigned NP, unsigned HW, int Cin, int Cout,
                                                    int want_bias) {
    // want_bias: the free channel slot Cin (< RGB_MAXC) carries a constant 1, so column Cin of the result is sum(gy): the
    // bias gradient out of the products the kernel forms anyway (rows of ws are then Cin + 1 wide).
    // CH = 16 where Cout allows (round 3): a lane reads 32 contiguous bytes of its pixel's gradient -- with Cout = 16 a wave's
    // loads are one 2 KB run where the 8-channel chunks of two workgroups each took every other 16 bytes -- and the image
    // is read once instead of once per chunk; two pixels per thread are in flight.
    const int CW = Cin + want_bias;
    __shared__ float red[RTPB / 64][CH * RGB_MAXC];
    const int c8 = blockIdx.y * CH;
    float acc[CH][RGB_MAXC];
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c) acc[j][c] = 0.f;
    auto load_px = [&](unsigned pix, float (&xv)[RGB_MAXC], float (&g)[CH]) {
        const bool ok = pix < NP;
        const unsigned pq = ok ? pix : 0u;
        const unsigned n = pq / HW, p = pq - n * HW;
        const TX* __restrict__ xp = x + rgb_xoff(n, p, sn, sh, sw, W, lin);
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c) xv[c] = c < Cin ? (ok ? ldf(xp + c * sc) : 0.f) : (c == Cin && want_bias && ok ? 1.f : 0.f);
#pragma unroll
        for (int h = 0; h < CH / 8; ++h) {
            float g8[8];
            if constexpr (sizeof(TG) == 2) {
                load_f<TG, 8>(gy + (int64_t)pq * Cout + c8 + 8 * h, g8);
            } else {
                float lo[4], hi[4];
                load_f<TG, 4>(gy + (int64_t)pq * Cout + c8 + 8 * h, lo);
                load_f<TG, 4>(gy + (int64_t)pq * Cout + c8 + 8 * h + 4, hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) { g8[j] = lo[j]; g8[4 + j] = hi[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) g[8 * h + j] = ok ? g8[j] : 0.f;
        }
    };
    const unsigned stride = gridDim.x * RTPB;
    for (unsigned pix = blockIdx.x * RTPB + threadIdx.x; pix < NP; pix += 2 * stride) {
        float xa[RGB_MAXC], xb[RGB_MAXC], ga[CH], gb[CH];
        load_px(pix, xa, ga);
        load_px(pix + stride, xb, gb);
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c) acc[j][c] = fmaf(gb[j], xb[c], fmaf(ga[j], xa[c], acc[j][c]));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c) {
            float v = acc[j][c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) red[wave][j * RGB_MAXC + c] = v;
        }
    __syncthreads();
    if (threadIdx.x < CH * RGB_MAXC) {
        float t = 0.f;
        for (int wv = 0; wv < RTPB / 64; ++wv) t += red[wv][threadIdx.x];
        const int j = threadIdx.x / RGB_MAXC, c = threadIdx.x % RGB_MAXC;
        if (c < CW) ws[((int64_t)blockIdx.x * Cout + c8 + j) * CW + c] = t;
    }
}

// stage 2: dw[co][c] = sum over the partials of the nb workgroups (deterministic order)
__global__ void __launch_bounds__(64) rgb_wgrad_fold_k(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ dbias, int nb, int L,
                                                       int Cin, int CW) {
    const int e = blockIdx.x;  // one wave per element of the (Cout, CW) partial rows
    float t = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) t += ws[(int64_t)b * L + e];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (threadIdx.x == 0) {
        const int co = e / CW, c = e - co * CW;
        if (c < Cin) dw[co * Cin + c] = t;
        else dbias[co] = t;
    }
}

bool rgb_shape_ok(int Cin, int Cout) { return Cin >= 1 && Cin <= RGB_MAXC && Cout >= 8 && Cout <= 64 && (Cout & 7) == 0; }

}  // namespace

extern "C" int gan_rgb_conv1x1_fwd(const void* x, int x_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const float* w, const float* bias,
                                   void* y, int y_dtype, int N, int Cin, int H, int W, int Cout, float slope, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0, "rgb_conv1x1_fwd: bad shape");
    GAN_CHECK_ARG(rgb_shape_ok(Cin, Cout), "rgb_conv1x1_fwd: serves Cin <= 4 and Cout a multiple of 8 up to 64 (got %d -> %d)", Cin, Cout);
    GAN_CHECK_ARG((x_dtype == GAN_F32 || x_dtype == GAN_BF16) && (y_dtype == GAN_F32 || y_dtype == GAN_BF16), "rgb_conv1x1_fwd: bad dtype");
    if (N == 0) return GAN_OK;
    GAN_CHECK_ARG(x && w && y && ((uintptr_t)y % 16) == 0, "rgb_conv1x1_fwd: null or unaligned pointer");
    const int64_t HW = (int64_t)H * W;
    GAN_CHECK_ARG(HW * (Cout / 8) < (1LL << 31) && N <= 65535, "rgb_conv1x1_fwd: tensor too large for the grid");
    const int CV = Cout / 8;
    GAN_CHECK_ARG(CV == 1 || CV == 2 || CV == 4 || CV == 8, "rgb_conv1x1_fwd: Cout must be 8, 16, 32 or 64");
    const dim3 grid((unsigned)ceil_div64(HW * CV, (int64_t)RTPB * RGB_FWD_PX), (unsigned)N);
    const int lin = sh == (int64_t)W * sw;
    hipStream_t st = (hipStream_t)stream;
    // a dense channels-last 2-byte image whose per-workgroup runs start on 16-byte boundaries
    const bool dense = x_dtype == GAN_BF16 && sc == 1 && sw == Cin && sh == (int64_t)W * Cin && sn == HW * Cin && ((uintptr_t)x % 16) == 0 &&
                       (HW * Cin * 2) % 16 == 0 && ((int64_t)RGB_FWD_PX * (RTPB / CV) * Cin * 2) % 16 == 0;
#define RGB_FWD3(TI, TO, CV_)                                                                                                                      \
    do {                                                                                                                                           \
        if (dense && sizeof(TI) == 2)                                                                                                              \
            hipLaunchKernelGGL((rgb_fwd_k<TI, TO, CV_, sizeof(TI) == 2>), grid, dim3(RTPB), 0, st, (const TI*)x, sn, sc, sh, sw, (unsigned)W, lin, w, \
                               bias, (TO*)y, (unsigned)HW, Cin, slope);                                                                            \
        else                                                                                                                                       \
            hipLaunchKernelGGL((rgb_fwd_k<TI, TO, CV_>), grid, dim3(RTPB), 0, st, (const TI*)x, sn, sc, sh, sw, (unsigned)W, lin, w, bias, (TO*)y,  \
                               (unsigned)HW, Cin, slope);                                                                                          \
    } while (0)
#define RGB_FWD(TI, TO)                     \
    do {                                    \
        if (CV == 1) RGB_FWD3(TI, TO, 1);   \
        else if (CV == 2) RGB_FWD3(TI, TO, 2); \
        else if (CV == 4) RGB_FWD3(TI, TO, 4); \
        else RGB_FWD3(TI, TO, 8);           \
    } while (0)
    if (x_dtype == GAN_F32 && y_dtype == GAN_F32) RGB_FWD(float, float);
    else if (x_dtype == GAN_F32) RGB_FWD(float, bf16);
    else if (y_dtype == GAN_F32) RGB_FWD(bf16, float);
    else RGB_FWD(bf16, bf16);
#undef RGB_FWD
#undef RGB_FWD3
    GAN_CHECK_LAUNCH("rgb_conv1x1_fwd");
    return GAN_OK;
}

extern "C" int gan_rgb_conv1x1_dgrad(const void* gy, const float* w, void* gx, int64_t NP, int Cin, int Cout, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(NP >= 0, "rgb_conv1x1_dgrad: bad shape");
    GAN_CHECK_ARG(rgb_shape_ok(Cin, Cout), "rgb_conv1x1_dgrad: serves Cin <= 4 and Cout a multiple of 8 up to 64 (got %d <- %d)", Cin, Cout);
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "rgb_conv1x1_dgrad: bad dtype %d", dtype);
    if (NP == 0) return GAN_OK;
    GAN_CHECK_ARG(gy && w && gx && ((uintptr_t)gy % 16) == 0, "rgb_conv1x1_dgrad: null or unaligned pointer");
    const int CV = Cout / 8;
    GAN_CHECK_ARG(CV == 1 || CV == 2 || CV == 4 || CV == 8, "rgb_conv1x1_dgrad: Cout must be 8, 16, 32 or 64");
    const int64_t blocks = ceil_div64(NP, (int64_t)(RTPB / CV) * RGB_DG_IT);
    GAN_CHECK_ARG(blocks < (1LL << 31), "rgb_conv1x1_dgrad: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    static const bool dg16 = !(getenv("GAN_AMD_RGB_DG16") && atoi(getenv("GAN_AMD_RGB_DG16")) == 0);  // A/B runs
    if (Cout == 16 && dg16) {
        const int64_t b16 = ceil_div64(NP, (int64_t)RTPB * RGB_DG16_IT);
        GAN_CHECK_ARG(b16 < (1LL << 31), "rgb_conv1x1_dgrad: tensor too large");
        if (dtype == GAN_F32) hipLaunchKernelGGL(rgb_dgrad16_k<float>, dim3((unsigned)b16), dim3(RTPB), 0, st, (const float*)gy, w, (float*)gx, NP, Cin);
        else hipLaunchKernelGGL(rgb_dgrad16_k<bf16>, dim3((unsigned)b16), dim3(RTPB), 0, st, (const bf16*)gy, w, (bf16*)gx, NP, Cin);
        GAN_CHECK_LAUNCH("rgb_conv1x1_dgrad");
        return GAN_OK;
    }
#define RGB_DG3(T_, CV_) hipLaunchKernelGGL((rgb_dgrad_k<T_, CV_>), dim3((unsigned)blocks), dim3(RTPB), 0, st, (const T_*)gy, w, (T_*)gx, NP, Cin)
#define RGB_DG(T_)                   \
    do {                             \
        if (CV == 1) RGB_DG3(T_, 1); \
        else if (CV == 2) RGB_DG3(T_, 2); \
        else if (CV == 4) RGB_DG3(T_, 4); \
        else RGB_DG3(T_, 8);         \
    } while (0)
    if (dtype == GAN_F32) RGB_DG(float);
    else RGB_DG(bf16);
#undef RGB_DG
#undef RGB_DG3
    GAN_CHECK_LAUNCH("rgb_conv1x1_dgrad");
    return GAN_OK;
}

extern "C" int64_t gan_rgb_conv1x1_wgrad_workspace(int Cin, int Cout) { return (int64_t)RGB_WG_BLOCKS * Cout * (Cin + 1); }

extern "C" int gan_rgb_conv1x1_wgrad(const void* x, int x_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw, const void* gy, int gy_dtype,
                                     float* dw, float* dbias, float* workspace, int64_t workspace_floats, int N, int Cin, int H, int W, int Cout,
                                     gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0, "rgb_conv1x1_wgrad: bad shape");
    GAN_CHECK_ARG(rgb_shape_ok(Cin, Cout), "rgb_conv1x1_wgrad: serves Cin <= 4 and Cout a multiple of 8 up to 64 (got %d, %d)", Cin, Cout);
    GAN_CHECK_ARG((x_dtype == GAN_F32 || x_dtype == GAN_BF16) && (gy_dtype == GAN_F32 || gy_dtype == GAN_BF16), "rgb_conv1x1_wgrad: bad dtype");
    GAN_CHECK_ARG(dw, "rgb_conv1x1_wgrad: null pointer");
    GAN_CHECK_ARG(dbias == nullptr || Cin < RGB_MAXC, "rgb_conv1x1_wgrad: the bias gradient rides in a free channel slot (Cin <= %d)", RGB_MAXC - 1);
    hipStream_t st = (hipStream_t)stream;
    const int want_bias = dbias != nullptr, CW = Cin + want_bias;
    const int L = Cout * CW;
    if (N == 0) {
        if ((dbias && hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, st) != hipSuccess) ||
            hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * Cin, st) != hipSuccess) {
            gan_set_error("rgb_conv1x1_wgrad: memset failed");
            return GAN_ERR_LAUNCH;
        }
        return GAN_OK;
    }
    GAN_CHECK_ARG(x && gy && ((uintptr_t)gy % 16) == 0, "rgb_conv1x1_wgrad: null or unaligned pointer");
    const int64_t NP = (int64_t)N * H * W;
    int64_t nb = ceil_div64(NP, (int64_t)RTPB * 8);  // >= 8 pixels per thread where the tensor allows
    if (nb > RGB_WG_BLOCKS) nb = RGB_WG_BLOCKS;
    if (nb < 1) nb = 1;
    GAN_CHECK_ARG(workspace && workspace_floats >= nb * L, "rgb_conv1x1_wgrad: needs a workspace of gan_rgb_conv1x1_wgrad_workspace(...) floats");
    const int CH = (Cout % 16 == 0) ? 16 : 8;  // output channels per thread
    const dim3 grid((unsigned)nb, (unsigned)(Cout / CH));
    GAN_CHECK_ARG(NP < (1LL << 31), "rgb_conv1x1_wgrad: tensor too large");  // (pix + 2 * stride stays below 2^32)
    const int lin = sh == (int64_t)W * sw;
#define RGB_WG2(TX, TG, CH_) \
    hipLaunchKernelGGL((rgb_wgrad_k<TX, TG, CH_>), grid, dim3(RTPB), 0, st, (const TX*)x, sn, sc, sh, sw, (unsigned)W, lin, (const TG*)gy, workspace, (unsigned)NP, (unsigned)(H * W), Cin, Cout, want_bias)
#define RGB_WG(TX, TG)                      \
    do {                                    \
        if (CH == 16) RGB_WG2(TX, TG, 16);  \
        else RGB_WG2(TX, TG, 8);            \
    } while (0)
    if (x_dtype == GAN_F32 && gy_dtype == GAN_F32) RGB_WG(float, float);
    else if (x_dtype == GAN_F32) RGB_WG(float, bf16);
    else if (gy_dtype == GAN_F32) RGB_WG(bf16, float);
    else RGB_WG(bf16, bf16);
#undef RGB_WG
#undef RGB_WG2
    GAN_CHECK_LAUNCH("rgb_conv1x1_wgrad");
    hipLaunchKernelGGL(rgb_wgrad_fold_k, dim3((unsigned)L), dim3(64), 0, st, workspace, dw, dbias, (int)nb, L, Cin, CW);
    GAN_CHECK_LAUNCH("rgb_conv1x1_wgrad(fold)");
    return GAN_OK;
}
