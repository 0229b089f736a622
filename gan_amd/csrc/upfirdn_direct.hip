// upfirdn2d with a resampling factor (2x down or 2x up), bf16 channels-last: direct polyphase evaluation on the VALU.
//
// The op (modelling/nvidia_ops.py:18-34) is HBM-bound: (numel_in + numel_out) * 2 bytes.  The stride-1 FIR is served by
// the matrix-core kernel (upfirdn_mfma.hip: 9-16 taps per input element would saturate the VALU); with a factor of two
// the polyphase form has only a QUARTER of that arithmetic per byte moved --
//   down = 2: 16 taps per output = 4 fmas per input element;   up = 2: 2 x 2 live taps per output = 4 fmas per output
// element -- while the banded-matrix form pays for the factor twice (K = 64 input pixels per 16 outputs, or a region a
// quarter of the output with every second tap row skipped), stages through LDS and synchronises per tile.  Here a thread
// owns ONE 16-byte output vector (8 channels of one pixel): its <= 16 input vectors are independent 16-byte buffer loads,
// all in flight at once (pixels outside the image: out-of-range offset, the hardware returns zeros), the taps are scalars,
// and consecutive lanes write consecutive 16-byte pieces of the output row.  No LDS, no barrier: the memory system sees
// ~20 waves per CU, each with its loads outstanding.  Neighbouring outputs share inputs through the vector L1; rows are
// handed to the XCDs in contiguous bands (the hardware deals consecutive workgroups round-robin over the 8 XCDs, each with
// its own L2), so the input rows two output rows share are fetched from HBM once.
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned DIRECT_INV = 0x80000000u;

struct DirectArgs {
    const bf16* x;
    const float* taps;
    bf16* y;
    int H, W, C, CV, Ho, Wo, px1, py1;
    int gx;             // workgroups per output row
    int blocks_per_img; // gx * rows (row pairs)
    int band;           // blocks_per_img / 8 when that is exact (XCD bands), else 0
    unsigned x_img_bytes, y_img_bytes;
};

__device__ __forceinline__ void fma8(float (&acc)[8], const v4u v, const float t) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        acc[2 * d] = fmaf(t, __builtin_bit_cast(float, v[d] << 16), acc[2 * d]);
        acc[2 * d + 1] = fmaf(t, __builtin_bit_cast(float, v[d] & 0xffff0000u), acc[2 * d + 1]);
    }
}

// K x K taps (zero padded to 4 x 4 in scalar registers), UP / DOWN in {1, 2} with exactly one of them 2
template <int UP, int DOWN, int K, int ROWS>
__global__ void __launch_bounds__(256) upfirdn2d_direct_k(const DirectArgs p) {
    // ---- workgroup -> (image, output row, x block): rows in bands per XCD -------------------------------------------
    const int n = blockIdx.x / p.blocks_per_img;
    int b = blockIdx.x - n * p.blocks_per_img;
    if (p.band) b = (b & 7) * p.band + (b >> 3);
    const int orow = b / p.gx;  // row (ROWS = 1) or row pair
    const int oy = orow * ROWS;
    const int idx = (b - orow * p.gx) * 256 + threadIdx.x;
    const int ox = idx / p.CV, cv = idx - ox * p.CV;
    const bool live = ox < p.Wo;

    float t[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t[i][j] = (i < K && j < K) ? p.taps[i * K + j] : 0.f;

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x) + (int64_t)n * p.H * p.W * p.C, 0, p.x_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)n * p.Ho * p.Wo * p.C, 0, p.y_img_bytes, 0x00020000);

    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;

    if constexpr (DOWN == 2 && ROWS == 2) {
        // two output rows per thread: they share K - 2 of their K input rows (24 loads for two outputs instead of 32)
        const int iy0 = oy * 2 - p.py1, ix0 = ox * 2 - p.px1;
        v4u v[K + 2][K];
#pragma unroll
        for (int i = 0; i < K + 2; ++i) {
            const int iy = iy0 + i;
            const bool rok = live && (unsigned)iy < (unsigned)p.H;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int ix = ix0 + j;
                const bool ok = rok && (unsigned)ix < (unsigned)p.W;
                v[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((iy * p.W + ix) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
            }
        }
        float acc1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc1[e] = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                fma8(acc, v[i][j], t[i][j]);
                fma8(acc1, v[i + 2][j], t[i][j]);
            }
        bf16x8 o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) o1[e] = (bf16)acc1[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o1), ry,
                                               live && oy + 1 < p.Ho ? (unsigned)((((oy + 1) * p.Wo + ox) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
    } else if constexpr (DOWN == 2) {
        const int iy0 = oy * 2 - p.py1, ix0 = ox * 2 - p.px1;
        v4u v[K][K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int iy = iy0 + i;
            const bool rok = live && (unsigned)iy < (unsigned)p.H;  // (wave-uniform but for `live`)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int ix = ix0 + j;
                const bool ok = rok && (unsigned)ix < (unsigned)p.W;
                v[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((iy * p.W + ix) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) fma8(acc, v[i][j], t[i][j]);
    } else if constexpr (ROWS >= 2) {
        // zero insertion, ROWS = 2 RP output rows per thread.  Row oy meets real samples at tap rows i0, i0 + 2 (i0 = parity
        // of oy - py1) and input rows iyb, iyb + 1; row oy + 1 at the other tap rows and -- i0 = 0: input rows iyb + 1,
        // iyb + 2; i0 = 1: the same two input rows; the next pair of rows one input row further.  i0 is wave-uniform: one
        // branch around two bodies with compile-time indices.
        constexpr int RP = ROWS / 2;
        const int by = oy - p.py1, bx = ox - p.px1;
        const int i0 = by & 1, j0 = bx & 1;
        const int iyb = (by + i0) >> 1, ixb = (bx + j0) >> 1;
        constexpr int NA = (K + 1) / 2;
        v4u v[NA + RP][NA];
#pragma unroll
        for (int a = 0; a < NA + RP; ++a) {
            const int iy = iyb + a;
            const bool rok = live && (unsigned)iy < (unsigned)p.H && (a < NA + RP - 1 || i0 == 0);
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int ix = ixb + c;
                const bool ok = rok && (unsigned)ix < (unsigned)p.W;
                v[a][c] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((iy * p.W + ix) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            float acc0[8], acc1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc0[e] = acc1[e] = 0.f;
            if (i0 == 0) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int c = 0; c < NA; ++c) {
                        fma8(acc0, v[q + a][c], j0 ? t[2 * a][2 * c + 1] : t[2 * a][2 * c]);
                        fma8(acc1, v[q + a + 1][c], j0 ? t[2 * a + 1][2 * c + 1] : t[2 * a + 1][2 * c]);
                    }
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int c = 0; c < NA; ++c) {
                        fma8(acc0, v[q + a][c], j0 ? t[2 * a + 1][2 * c + 1] : t[2 * a + 1][2 * c]);
                        fma8(acc1, v[q + a][c], j0 ? t[2 * a][2 * c + 1] : t[2 * a][2 * c]);
                    }
            }
            bf16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0[e] = (bf16)acc0[e];
                o1[e] = (bf16)acc1[e];
            }
            const int r0 = oy + 2 * q;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o0), ry,
                                                   live && r0 < p.Ho ? (unsigned)(((r0 * p.Wo + ox) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o1), ry,
                                                   live && r0 + 1 < p.Ho ? (unsigned)((((r0 + 1) * p.Wo + ox) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
        }
        return;
    } else {
        // zero insertion: output (oy, ox) meets real samples at taps i = i0 + 2a, j = j0 + 2b, with (oy - py1 + i0) and
        // (ox - px1 + j0) even.  i0 is the row's (scalar), j0 the lane's: its two tap columns are selected per lane.
        const int by = oy - p.py1, bx = ox - p.px1;
        const int i0 = by & 1, j0 = bx & 1;
        const int iyb = (by + i0) >> 1, ixb = (bx + j0) >> 1;  // arithmetic shifts: exact, the sums are even
        constexpr int NA = (K + 1) / 2;
        v4u v[NA][NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int iy = iyb + a;
            const bool rok = live && (unsigned)iy < (unsigned)p.H;
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int ix = ixb + c;
                const bool ok = rok && (unsigned)ix < (unsigned)p.W;
                v[a][c] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((iy * p.W + ix) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                // t[i0 + 2a][j0 + 2c] (indices <= 3: rows / columns past K hold zeros -- K = 3: the odd phase has one tap)
                const float te = i0 ? t[2 * a + 1][2 * c] : t[2 * a][2 * c];
                const float to = i0 ? t[2 * a + 1][2 * c + 1] : t[2 * a][2 * c + 1];
                const float tap = j0 ? to : te;
                fma8(acc, v[a][c], tap);
            }
        }
    }

    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), ry, live ? (unsigned)(((oy * p.Wo + ox) * p.C + cv * 8) * 2) : DIRECT_INV, 0, 0);
}

}  // namespace

// 1 = served, 0 = shape not eligible (the caller goes on to the matrix-core / tiled kernels), < 0 = error.
int gan_upfirdn2d_direct(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx, int up, int down,
                         int px1, int py1, int Ho, int Wo, hipStream_t st) {
    if (!((up == 1 && down == 2) || (up == 2 && down == 1))) return 0;
    if (ky != kx || ky < 2 || ky > 4 || (C & 7) || N <= 0) return 0;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 15)) return 0;
    const int64_t xb = (int64_t)H * W * C * 2, yb = (int64_t)Ho * Wo * C * 2;
    if (xb >= (1LL << 31) || yb >= (1LL << 31)) return 0;  // 32-bit buffer offsets inside an image
    static const bool off = getenv("GAN_AMD_FIR_DIRECT") && atoi(getenv("GAN_AMD_FIR_DIRECT")) == 0;  // A/B runs
    if (off) return 0;
    DirectArgs a;
    a.x = (const bf16*)x; a.taps = taps; a.y = (bf16*)y;
    a.H = H; a.W = W; a.C = C; a.CV = C / 8; a.Ho = Ho; a.Wo = Wo; a.px1 = px1; a.py1 = py1;
    const int64_t row_vecs = (int64_t)Wo * a.CV;
    a.gx = (int)((row_vecs + 255) / 256);
    // rows per thread.  Measured on 32 x C x S x S (TB/s of in + out bytes; 1 / 2 / 4 / 8 rows): 2x down, 16 x 1024^2: 3.8 / 5.6;
    // 64 x 256^2: 4.4 / 5.8.  2x up, 64 x 256^2: 3.6 / 4.3 / 5.4 / 5.5; 256 x 64^2: 3.3 / 4.0 / 4.6 / 5.1 -- more bytes in
    // flight per wave, a quarter / an eighth of the waves to launch; small maps keep enough workgroups to fill the chip.
    static const int rows_env = getenv("GAN_AMD_FIR_ROWS") ? atoi(getenv("GAN_AMD_FIR_ROWS")) : 0;  // A/B runs
    int rows = down == 2 ? 2 : (Ho >= 128 ? 8 : Ho >= 16 ? 4 : 2);
    if (rows_env >= 1) rows = down == 2 ? (rows_env >= 2 ? 2 : 1) : (rows_env >= 8 ? 8 : rows_env >= 4 ? 4 : rows_env >= 2 ? 2 : 1);
    const int64_t bpi = (int64_t)a.gx * ((Ho + rows - 1) / rows);
    if (bpi * N >= (1LL << 31)) return 0;
    a.blocks_per_img = (int)bpi;
    a.band = (bpi % 8 == 0) ? (int)(bpi / 8) : 0;
    a.x_img_bytes = (unsigned)xb; a.y_img_bytes = (unsigned)yb;
    const dim3 grid((unsigned)(bpi * N));
#define DIRECT_LAUNCH(UP_, DOWN_, ROWS_)                                                                            \
    do {                                                                                                            \
        if (ky == 2) hipLaunchKernelGGL((upfirdn2d_direct_k<UP_, DOWN_, 2, ROWS_>), grid, dim3(256), 0, st, a);      \
        else if (ky == 3) hipLaunchKernelGGL((upfirdn2d_direct_k<UP_, DOWN_, 3, ROWS_>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((upfirdn2d_direct_k<UP_, DOWN_, 4, ROWS_>), grid, dim3(256), 0, st, a);              \
    } while (0)
    if (down == 2 && rows == 2) DIRECT_LAUNCH(1, 2, 2);
    else if (down == 2) DIRECT_LAUNCH(1, 2, 1);
    else if (rows == 8) DIRECT_LAUNCH(2, 1, 8);
    else if (rows == 4) DIRECT_LAUNCH(2, 1, 4);
    else if (rows == 2) DIRECT_LAUNCH(2, 1, 2);
    else DIRECT_LAUNCH(2, 1, 1);
#undef DIRECT_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gan_set_error("upfirdn2d(direct): launch failed: %s", hipGetErrorString(e));
        return GAN_ERR_LAUNCH;
    }
    return 1;
}
