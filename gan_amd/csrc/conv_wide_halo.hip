// 3x3 convolution for WIDE layers with a resident input halo (gfx950, bf16): Cin % 64 == 0 on 8x8 .. 64x64 maps,
// large grids (the inner blocks of the StyleGAN2 generator and discriminator at training batch sizes,
// modelling/stylegan2.py:29-36, modelling/progressive_gan.py:45-50, and their input gradients).
//
// conv_wide.hip runs the same layers as an implicit GEMM that re-stages the 128-pixel operand tile for every
// tap: 8 LDS-DMA pieces per wave and K step, and the issue of those pieces -- not MFMA, not L2 bandwidth -- set
// its step time (rocprof: DMA-only 90 us, MFMA-only 86 us, together 110 us on 128 -> 128 @ 32x32, batch 256).
// Here a workgroup owns 256 output pixels (16 x 16 of one image, or 8 x 8 of four images) x BN channels, and
//   * the input HALO of the tile (18 x 18 pixels, or 4 x 10 x 10; 10 x 10 source pixels under nearest-2x
//     upsampling) x 64 channels is staged ONCE per channel block and read by all nine taps: 7 pieces per wave
//     per 9 K steps instead of 36, double buffered, one piece per step so no step carries a burst;
//   * only the weight tile (BN x 64 of one tap) moves every step: BN / 64 pieces per wave, a 3-stage ring;
//   * 8 waves, 64 x (BN / 2) ... wave tiles as in conv_wide.hip; the nine taps are unrolled, so ring stages,
//     `s_waitcnt vmcnt(N)` immediates and tap offsets are compile-time.
// LDS image: one 128-byte row per halo pixel / weight row, 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the
// DMA's source side.  A tap reads 16 consecutive halo pixels from an ARBITRARY start, so the aligned-tile
// argument of conv_wide.hip does not hold; fragments are read as two ds_read_b64 in which lanes of odd k-group
// take the chunk's halves in the opposite order: any 8 consecutive even (odd) rows then cover all 32 8-byte bank
// pairs, whatever the start.  Weight fragments are read the same way so both MFMA operands see the same k order.
// Output stages (demodulation, bias, noise, leaky-ReLU, mask, residual, 2x2 average pooling) as in conv_wide.hip;
// pooling adds two fragments held by one lane (vertical) and one DPP neighbour (horizontal).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned HALO_INV = 0x80000000u;
constexpr int HALO_MAXP = 7;  // halo pieces (1 KiB each) per wave and channel block

struct HaloArgs {
    const void* x;
    const void* wt;
    void* y;
    const float* out_scale;
    const float* bias;
    const void* noise;
    const float* noise_scale;
    const void* aux;
    const void* resid;
    float slope, aux_slope, gain;
    int N, H, W, Cin, up2, Ho, Wo, Cout, Kpad, pool;
    int tiles_x, tiles_y, ntiles, t8, P, IS, HH, perh, ncb, wfrag;
    unsigned x_bytes, w_bytes, y_bytes;
};

// Diagnostic build (-DGAN_HALO_STAMPS, never the shipped library): s_memtime stamps of one K step and of the kernel's
// phases, per wave, into a buffer of their own that no output depends on; tools/halo_stamps.py reads them.
#ifdef GAN_HALO_STAMPS
__device__ unsigned long long halo_stamps[8192 * 16];
#define STAMP(k) st[k] = __builtin_readcyclecounter()
#else
#define STAMP(k)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one LDS-DMA piece: 64 lanes x 16 bytes from buffer offsets voff (+ the wave-uniform soff) to the 1 KiB at dst.
// (A __device__ function on purpose: the host pass cannot see this builtin, and a kernel body that names it directly
// loses its host-side stub without a diagnostic.)
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rsrc, char* dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

// a fragment (8 bf16 of one row) as two 8-byte LDS reads; `a` addresses the half this lane takes first
__device__ __forceinline__ bf16x8 frag_read(const char* lds, unsigned a) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(lds + a);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(lds + (a ^ 8u));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// This lane's pixel (image of the tile, row, column) in fragment g = wm*FP + f of the tile:
//   16-wide tile (16 or 8 rows): row g, x = frow;   4 x (8 x 8): row 2*(g >> 2) + (g & 1) of images 2*((g >> 1) & 1) + {0, 1}.
// Fragments (g, g + 1), g even, are vertical neighbours starting at an even row either way: what pooling needs.
__device__ __forceinline__ void frag_pixel(int g, int frow, int t8, int& img, int& y, int& x) {
    img = t8 ? 2 * ((g >> 1) & 1) + (frow >> 3) : 0;
    y = t8 ? 2 * (g >> 2) + (g & 1) : g;
    x = t8 ? (frow & 7) : frow;
}

// ---- output stages: lane holds channels co0 .. co0+3 (co0 = bn0 + wn*64 + 16*i + 4*fq) of its pixel of fragment f.
// Absent stages cost nothing (wave-uniform branches, no out-of-range loads).  Shared by the two kernels below.
template <int FC, int FP>
__device__ __forceinline__ void halo_epilogue(const HaloArgs& p, f32x4 (&acc)[FC][FP], const int wm, const int wn, const int frow, const int fq,
                                              const int n0, const int ty0, const int tx0, const int bn0) {
    // ---- epilogue: lane holds channels co0 .. co0+3 (co0 = bn0 + wn*64 + 16*i + 4*fq) of its pixel of fragment f.
    // Absent stages cost nothing (wave-uniform branches, no out-of-range loads).
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.aux), 0, p.aux ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0, p.resid ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rnz =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.noise), 0, p.noise ? p.y_bytes / (unsigned)p.Cout : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rosc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.out_scale), 0, p.out_scale ? (unsigned)(p.N * p.Cout) * 4u : 0u, 0x00020000);
    const float gain = p.pool ? 0.25f * p.gain : p.gain;
    const int Hy = p.pool ? p.Ho >> 1 : p.Ho, Wy = p.pool ? p.Wo >> 1 : p.Wo;
    float4 bia[FC], nsc[FC];
    int co0[FC];
#pragma unroll
    for (int i = 0; i < FC; ++i) {
        co0[i] = bn0 + wn * 64 + i * 16 + fq * 4;
        const bool cok = co0[i] < p.Cout;  // Cout % 8 == 0
        bia[i] = (p.bias && cok) ? *reinterpret_cast<const float4*>(p.bias + co0[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        nsc[i] = (p.noise_scale && cok) ? *reinterpret_cast<const float4*>(p.noise_scale + co0[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    struct OutPixel {
        int n;
        unsigned obase;  // byte offset of the pixel's channel 0 in y / aux / resid, HALO_INV when this lane has none
        float nz;
    };
    // The output stages are compiled three times -- plain (scale, bias, activation), with the input-gradient stages
    // (mask / residual), with everything -- and picked once per workgroup: tested per fragment, the wave-uniform
    // branches split the stage code into ~100 basic blocks with a wait in front of most.
    auto epilogue = [&](auto modc, auto arc) {
    constexpr bool MOD = decltype(modc)::value, AR = decltype(arc)::value;
    // this lane's output pixel of fragment f (pooling: of the 2x2 window that fragments f, f+1 and lanes frow, frow^1 hold)
    auto out_pixel = [&](const int f) {
        int img, ly, lx;
        frag_pixel(wm * FP + f, frow, p.t8, img, ly, lx);
        OutPixel q;
        q.n = n0 + img;
        const int oy = ty0 + ly, ox = tx0 + lx;
        const bool live = q.n < p.N && (!p.pool || (frow & 1) == 0);
        const int opix = p.pool ? (q.n * Hy + (oy >> 1)) * Wy + (ox >> 1) : (q.n * Hy + oy) * Wy + ox;
        q.obase = live ? (unsigned)opix * (unsigned)(p.Cout * 2) : HALO_INV;
        q.nz = 0.f;
        if (MOD && p.noise) q.nz = (float)__builtin_bit_cast(bf16, __builtin_amdgcn_raw_buffer_load_b16(rnz, live ? (unsigned)opix * 2u : HALO_INV, 0, 0));
        return q;
    };
    // every output stage on the lane's 4 channels of fragment column i
    // The mask / residual operands of the WHOLE tile are requested before the first output stage runs (a tensor that is
    // absent has a zero-sized resource: its loads return zeros and cost no memory traffic).  Loaded where they are used,
    // behind the wave-uniform `if (p.aux)`, every (fragment, column) pair paid its own round trip to memory: 32 in a row
    // per tile, +13 us on a 20 us tile (128 -> 128 @ 32x32: 81 us plain, 134 us as the D block's input gradient).
    OutPixel qs[FP];
    v2u axv[AR ? FC : 1][AR ? FP : 1], rsv[AR ? FC : 1][AR ? FP : 1];
#pragma unroll
    for (int f = 0; f < FP; f += (1)) {
        if (p.pool && (f & 1)) continue;
        qs[f] = out_pixel(f);
    }
    if constexpr (AR) {
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            if (p.pool && (f & 1)) continue;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const unsigned off = (qs[f].obase != HALO_INV && co0[i] < p.Cout) ? qs[f].obase + co0[i] * 2 : HALO_INV;
                axv[i][f] = __builtin_amdgcn_raw_buffer_load_b64(raux, off, 0, 0);
                rsv[i][f] = __builtin_amdgcn_raw_buffer_load_b64(rres, off, 0, 0);
            }
        }
    }
    auto finish = [&](const int i, const int f, const f32x4 a) {
        const OutPixel& q = qs[f];
        const unsigned off = (q.obase != HALO_INV && co0[i] < p.Cout) ? q.obase + co0[i] * 2 : HALO_INV;
        float4 d4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (MOD && p.out_scale)
            d4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rosc, off != HALO_INV ? (unsigned)(q.n * p.Cout + co0[i]) * 4u : HALO_INV, 0, 0));
        // (absent stages are wave-uniform branches: the plain layer is one fma per element, not a mul and two fmas)
        float v[4];
        if (MOD && p.out_scale) {
            v[0] = fmaf(a[0] * gain, d4.x, bia[i].x);
            v[1] = fmaf(a[1] * gain, d4.y, bia[i].y);
            v[2] = fmaf(a[2] * gain, d4.z, bia[i].z);
            v[3] = fmaf(a[3] * gain, d4.w, bia[i].w);
        } else {
            v[0] = fmaf(a[0], gain, bia[i].x);
            v[1] = fmaf(a[1], gain, bia[i].y);
            v[2] = fmaf(a[2], gain, bia[i].z);
            v[3] = fmaf(a[3], gain, bia[i].w);
        }
        if (MOD && p.noise) {
            v[0] = fmaf(q.nz, nsc[i].x, v[0]);
            v[1] = fmaf(q.nz, nsc[i].y, v[1]);
            v[2] = fmaf(q.nz, nsc[i].z, v[2]);
            v[3] = fmaf(q.nz, nsc[i].w, v[3]);
        }
        if (p.slope != 1.f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = lrelu_f(v[r], p.slope);
        }
        if constexpr (AR) {
            if (p.aux) {
                const bf16x4 a4 = __builtin_bit_cast(bf16x4, axv[i][f]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= lrelu_mask((float)a4[r], p.aux_slope);
            }
            const bf16x4 r4 = __builtin_bit_cast(bf16x4, rsv[i][f]);  // zeros when there is no residual
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
        return __builtin_bit_cast(v2u, o);
    };
    if (p.pool) {
#pragma unroll
        for (int f = 0; f < FP; f += 2) {
            const OutPixel& q = qs[f];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                f32x4 a;
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // vertical neighbour in fragment f+1, horizontal one lane over (every lane executes the DPP add)
                    a[r] = acc[i][f][r] + acc[i][f + 1][r];
                    a[r] += quad_xor1(a[r]);
                }
                const v2u o = finish(i, f, a);
                __builtin_amdgcn_raw_buffer_store_b64(o, ry, (q.obase != HALO_INV && co0[i] < p.Cout) ? q.obase + co0[i] * 2 : HALO_INV, 0, 0);
            }
        }
    } else {
        // Two fragments at a time: v_permlane16_swap trades the k-group-odd lanes' results of fragment f for the k-group-even
        // lanes' results of fragment f+1, after which a lane holds 8 consecutive channels of ONE pixel: 16-byte stores, half as
        // many of them (the store tail of a tile is issue-bound).
#pragma unroll
        for (int f = 0; f < FP; f += 2) {
            const unsigned obase = (fq & 1) ? qs[f + 1].obase : qs[f].obase;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                v2u xa = finish(i, f, acc[i][f]), xb = finish(i, f + 1, acc[i][f + 1]);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(xa[d], xb[d], false, false);
                    xa[d] = sw[0];
                    xb[d] = sw[1];
                }
                const int cb8 = bn0 + wn * 64 + i * 16 + (fq >> 1) * 8;
                const v4u o = {xa[0], xa[1], xb[0], xb[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, ry, (obase != HALO_INV && cb8 < p.Cout) ? obase + cb8 * 2 : HALO_INV, 0, 0);
            }
        }
    }
    };
    if (p.out_scale || p.noise) epilogue(std::true_type{}, std::true_type{});
    else if (p.aux || p.resid) epilogue(std::false_type{}, std::true_type{});
    else epilogue(std::false_type{}, std::false_type{});
}

template <int BN, int WAVES, bool UP2>
__global__ void __launch_bounds__(WAVES * 64, 2) conv_wide_halo_k(const HaloArgs p) {
    constexpr int FC = 4, WN = BN / 64, WM = WAVES / WN;  // wave tile: FP*16 pixels x 64 channels
    constexpr int NFRAG = 2 * WAVES;                       // 16-pixel fragments per tile: 256 pixels per 8 waves, 128 per 4
    constexpr int FP = NFRAG / WM;
    constexpr int PERW = BN / (8 * WAVES);                 // weight pieces per wave and step
    constexpr int WSTG = WAVES == 8 ? 3 : 2;               // weight ring: what fits beside the halo at 1 / 2 workgroups per CU
    constexpr int W_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    char* const lds = lds_raw;
    const int HB = p.perh * WAVES * 1024;  // one halo buffer; the weight ring follows the two of them
    const unsigned wring = 2u * (unsigned)HB;

#ifdef GAN_HALO_STAMPS
    unsigned long long st[16] = {};
#endif
    STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int frow = lane & 15, fq = lane >> 4, rsub = lane >> 3;
    const int TW = p.t8 ? 8 : 16, TH = p.t8 ? 8 : NFRAG, imgs = p.t8 ? NFRAG / 4 : 1;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, p.w_bytes, 0x00020000);

    // ---- DMA bookkeeping: piece q = j*WAVES + wave covers LDS rows 8q .. 8q+7, 8 lanes (chunks) per row.  The slot
    // `lane & 7` of row r holds source chunk (lane & 7) ^ ((r >> 1) & 7); (r >> 1) & 7 = (wave & 1) * 4 + rsub / 2.
    const int csrc = (lane & 7) ^ (((wave & 1) << 2) | (rsub >> 1));
    // ---- a workgroup's work item: a pixel tile x a BN-channel block of Cout (tiles of one channel block are consecutive
    // workgroups: they share the weights in L2)
    struct Item {
        int n0, ty0, tx0, bn0;
    };
    auto decode = [&](const int item) {
        const int by = item / p.ntiles;
        int bx = item - by * p.ntiles;
        const int tx = bx % p.tiles_x;
        bx /= p.tiles_x;
        const int ty = bx % p.tiles_y, ng = bx / p.tiles_y;
        return Item{ng * imgs, ty * TH, tx * TW, by * BN};
    };
    unsigned hv[HALO_MAXP], b_off[PERW];
    const float inv_is = 1.0f / (float)p.IS, inv_p = 1.0f / (float)p.P;
    // source offsets of this lane's halo / weight pieces for an item, and the item's first pieces: the halo of channel
    // block 0 and the weights of the first WSTG - 1 steps
    auto prologue = [&](const Item& it) {
        const int sy0 = (UP2 ? it.ty0 >> 1 : it.ty0) - 1, sx0 = (UP2 ? it.tx0 >> 1 : it.tx0) - 1;
#pragma unroll
        for (int j = 0; j < HALO_MAXP; ++j) {
            // (hidx < 512 and the divisors are <= 328: a float multiply by the reciprocal is exact after the +0.5, and costs
            // two instructions where an integer division costs forty)
            const int hidx = (j * WAVES + wave) * 8 + rsub;
            const int img = (int)(((float)hidx + 0.5f) * inv_is), rem = hidx - img * p.IS;
            const int hy = (int)(((float)rem + 0.5f) * inv_p), hx = rem - hy * p.P;
            const int n = it.n0 + img, sy = sy0 + hy, sx = sx0 + hx;
            const bool ok = j < p.perh && img < imgs && hy < p.HH && n < p.N && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
            hv[j] = ok ? (unsigned)(((n * p.H + sy) * p.W + sx) * p.Cin + csrc * 8) * 2u : HALO_INV;
        }
#pragma unroll
        for (int j = 0; j < PERW; ++j) {
            const int co = it.bn0 + (j * WAVES + wave) * 8 + rsub;
            b_off[j] = co < p.Cout ? (unsigned)(co * p.Kpad + csrc * 8) * 2u : HALO_INV;
        }
#pragma unroll
        for (int j = 0; j < HALO_MAXP; ++j)
            if (j < p.perh) dma_piece(rx, lds + (j * WAVES + wave) * 1024, hv[j], 0);
#pragma unroll
        for (int s = 0; s < WSTG - 1; ++s)
#pragma unroll
            for (int j = 0; j < PERW; ++j)
                dma_piece(rw, lds + wring + s * W_BYTES + (j * WAVES + wave) * 1024, b_off[j], s * p.Cin * 2);
    };

    // halo index of this lane's pixel at tap (0, 0), per fragment; under upsampling the index depends on the tap's parity
    // arithmetic, so the register holds (x << 24 | y << 16 | halo index of the image) instead
    int hb[FP];
#pragma unroll
    for (int f = 0; f < FP; ++f) {
        int img, y, x;
        frag_pixel(wm * FP + f, frow, p.t8, img, y, x);
        hb[f] = UP2 ? (x << 24 | y << 16 | img * p.IS) : img * p.IS + y * p.P + x;
    }

    f32x4 acc[FC][FP];

    // fragment addressing: row r, k-group fq of K half h -> chunk (4h + fq) ^ ((r >> 1) & 7), first half fq & 1
    const unsigned c8 = (unsigned)(fq & 1) << 3;
    // weight rows: (r >> 1) & 7 = frow >> 1.  One register per fragment, their distance (16 rows = 2048 bytes) a kernel
    // argument: with a distance it can see, the compiler fuses the reads of fragments i and i+1 into
    // ds_read2st64_b64, which runs at half the LDS rate (32-bank mode)
    unsigned wa[FC];
#pragma unroll
    for (int i = 0; i < FC; ++i)
        wa[i] = wring + (unsigned)((wn * 64 + frow) * 128 + i * p.wfrag) + (unsigned)((fq ^ (frow >> 1)) << 4) + c8;

    // One K step = one tap of one channel block.  TAP is a constant expression (ring stages, vmcnt immediates, tap offsets):
    //  - W(t) has landed once only the pieces issued after it are outstanding: with three stages W(t+1) and the halo
    //    piece of step t-1, with two stages nothing;
    //  - the barrier says so for every wave, and that every wave is done with step t-1's stage;
    //  - then one piece of the next channel block's halo (taps 1..7), the weights of step t+WSTG-1, the MFMAs of step t.
    auto step = [&](auto tapc, const int cbi, const bool last_cb) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int DY = TAP / 3, DX = TAP % 3;
        const bool stamped = TAP == 4 && cbi == 1;
        (void)stamped;
        if (stamped) STAMP(3);
        if (WSTG == 2 || (last_cb && TAP == 8)) {
            wait_vm<0>();
        } else if (TAP >= 2 && (TAP - 2) < p.perh && !last_cb) {
            wait_vm<PERW + 1>();
        } else {
            wait_vm<PERW>();
        }
        if (stamped) STAMP(4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (stamped) STAMP(5);
        if constexpr (TAP >= 1 && TAP <= HALO_MAXP) {
            if ((TAP - 1) < p.perh && !last_cb) dma_piece(rx, lds + ((cbi + 1) & 1) * HB + ((TAP - 1) * WAVES + wave) * 1024, hv[TAP - 1], (cbi + 1) * 128);
        }
        {
            constexpr int T2 = (TAP + WSTG - 1) % 9;  // 9 taps are a multiple of WSTG = 3; with 2 stages the parity of the step decides
            const int cb2 = cbi + (TAP + WSTG - 1 >= 9 ? 1 : 0);
            const int stg = WSTG == 3 ? T2 % 3 : (cb2 + T2) & 1;
            if (cb2 < p.ncb) {
#pragma unroll
                for (int j = 0; j < PERW; ++j)
                    dma_piece(rw, lds + wring + stg * W_BYTES + (j * WAVES + wave) * 1024, b_off[j], (T2 * p.Cin + cb2 * 64) * 2);
            }
        }
        if (stamped) STAMP(6);
        unsigned ha[FP];
        const unsigned hbuf = (unsigned)((cbi & 1) * HB);
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            int hidx;
            if constexpr (UP2) {
                const int y = hb[f] >> 16 & 0xff, x = hb[f] >> 24;
                hidx = (hb[f] & 0xffff) + (((y + DY - 1) >> 1) + 1) * p.P + (((x + DX - 1) >> 1) + 1);
            } else {
                hidx = hb[f] + DY * p.P + DX;
            }
            ha[f] = hbuf + (unsigned)(hidx << 7) + (unsigned)(((fq ^ (hidx >> 1)) & 7) << 4) + c8;
        }
        const unsigned WST = (WSTG == 3 ? TAP % 3 : (cbi + TAP) & 1) * W_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 wf[FC], xf[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) wf[i] = frag_read(lds + WST, wa[i] ^ (h ? 64u : 0u));
#pragma unroll
            for (int f = 0; f < FP; ++f) xf[f] = frag_read(lds, ha[f] ^ (h ? 64u : 0u));
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int f = 0; f < FP; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[f], acc[i][f], 0, 0, 0);
            if (stamped) STAMP(7 + h);
        }
    };

    const Item cur = decode(blockIdx.x);
    prologue(cur);
    STAMP(1);
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cbi = 0; cbi < p.ncb; ++cbi) {
        const bool last_cb = cbi + 1 == p.ncb;
        step(std::integral_constant<int, 0>{}, cbi, last_cb);
        step(std::integral_constant<int, 1>{}, cbi, last_cb);
        step(std::integral_constant<int, 2>{}, cbi, last_cb);
        step(std::integral_constant<int, 3>{}, cbi, last_cb);
        step(std::integral_constant<int, 4>{}, cbi, last_cb);
        step(std::integral_constant<int, 5>{}, cbi, last_cb);
        step(std::integral_constant<int, 6>{}, cbi, last_cb);
        step(std::integral_constant<int, 7>{}, cbi, last_cb);
        step(std::integral_constant<int, 8>{}, cbi, last_cb);
    }

    STAMP(9);
    halo_epilogue<FC, FP>(p, acc, wm, wn, frow, fq, cur.n0, cur.ty0, cur.tx0, cur.bn0);
#ifdef GAN_HALO_STAMPS
    STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(13);
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 1024)
        for (int k = 0; k < 16; ++k) halo_stamps[(blockIdx.x * WAVES + wave) * 16 + k] = st[k];
#endif
}


// ---- the 16-wide tiling (8 x 16 output pixels, 4 waves, two workgroups per CU) with an address-free K loop (round 3) ----
// PMC on the kernel above (profiles/r03_pmc_summary.txt): 3.9 VALU instructions per MFMA, ~70 per K step -- fragment
// addresses recomputed for every tap (the swizzle key depended on the halo row index, so on the tap), XORs for the second
// K half and the second 8 bytes of every fragment, ring-stage arithmetic -- on a SIMD whose vector issue an MFMA holds
// for 8 of its 16 cycles; and ~300 cycles per step of LDS-DMA issue in a phase of its own.  Here
//   * the swizzle key of a halo row is a function of its COLUMN only ((hx >> 1) & 7: the pitch is even, so the 16 rows of
//     a fragment still split into 8 even and 8 odd rows with distinct keys); a fragment's address is then
//     base(lane, tap column DX) + a compile-time offset (tap row, fragment row, halo buffer): 3 x 4 base registers per
//     lane for the halo and 4 for the weights, computed once per tile, every other term an instruction immediate;
//   * the channel-block loop is unrolled by two so halo buffer and ring stage are compile-time;
//   * LDS reads are volatile: with offsets it can see, the compiler fuses pairs of them into ds_read2st64_b64, which runs
//     at half the LDS rate (32-bank mode);
//   * the step's DMA pieces are issued between its MFMAs instead of in front of them.
template <int OFF>
__device__ __forceinline__ bf16x8 frag_read_imm(const __attribute__((address_space(3))) char* lds3, const unsigned a_lo, const unsigned a_hi) {
    typedef const volatile __attribute__((address_space(3))) v2u* vp;
    const v2u lo = *reinterpret_cast<vp>(lds3 + a_lo + OFF);
    const v2u hi = *reinterpret_cast<vp>(lds3 + a_hi + OFF);
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int BN, bool UP2>
__global__ void __launch_bounds__(256, 2) conv_wide_halo4_k(const HaloArgs p) {
    constexpr int WAVES = 4, FC = 4, WN = BN / 64, WM = WAVES / WN, NFRAG = 8, FP = NFRAG / WM;
    constexpr int PERW = BN / (8 * WAVES);            // weight pieces per wave and step
    constexpr int W_BYTES = BN * 128;
    constexpr int TW = 16, TH = 8;
    constexpr int P = (UP2 ? TW / 2 : TW) + 2, HH = (UP2 ? TH / 2 : TH) + 2;  // halo pitch / rows in source pixels
    constexpr int PERH = (P * HH + 8 * WAVES - 1) / (8 * WAVES);               // halo pieces per wave and channel block: 6 (2 upsampled)
    constexpr int HB = PERH * WAVES * 1024;           // one halo buffer; the two-stage weight ring follows the two of them
    constexpr int WRING = 2 * HB;
    constexpr int NP = PERW + 1;  // DMA pieces per wave and step: one of the next halo, PERW of the next weights
    static_assert(FP % 2 == 0, "tile");
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    char* const lds = lds_raw;
    const __attribute__((address_space(3))) char* const lds3 = (const __attribute__((address_space(3))) char*)lds_raw;

#ifdef GAN_HALO_STAMPS
    unsigned long long st[16] = {};
#endif
    STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: every DMA destination (M0) is then scalar arithmetic
    const int wn = wave % WN, wm = wave / WN;
    const int frow = lane & 15, fq = lane >> 4, rsub = lane >> 3;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, p.w_bytes, 0x00020000);

    // work item: a pixel tile x a BN-channel block of Cout (tiles of one channel block are consecutive workgroups)
    const int item = blockIdx.x;
    const int by = item / p.ntiles;
    int bx = item - by * p.ntiles;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y, n0 = bx / p.tiles_y;
    const int ty0 = ty * TH, tx0 = tx * TW, bn0 = by * BN;

    // ---- DMA sources.  Piece q = j*WAVES + wave covers LDS rows 8q .. 8q+7, 8 lanes (16-byte chunks) per row; slot
    // `lane & 7` of a row holds source chunk (lane & 7) ^ key(row): halo rows key = (column >> 1) & 7, weight rows
    // key = (row >> 1) & 7 = (wave & 1) * 4 + rsub / 2.
    unsigned hv[PERH], b_off[PERW];
    {
        const int sy0 = (UP2 ? ty0 >> 1 : ty0) - 1, sx0 = (UP2 ? tx0 >> 1 : tx0) - 1;
        constexpr float inv_p = 1.0f / (float)P;
#pragma unroll
        for (int j = 0; j < PERH; ++j) {
            const int hidx = (j * WAVES + wave) * 8 + rsub;            // < 256: the reciprocal multiply is exact after the +0.5
            const int hy = (int)(((float)hidx + 0.5f) * inv_p), hx = hidx - hy * P;
            const int sy = sy0 + hy, sx = sx0 + hx;
            const bool ok = hy < HH && n0 < p.N && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
            const int csrc = (lane & 7) ^ ((hx >> 1) & 7);
            hv[j] = ok ? (unsigned)(((n0 * p.H + sy) * p.W + sx) * p.Cin + csrc * 8) * 2u : HALO_INV;
        }
        const int csw = (lane & 7) ^ (((wave & 1) << 2) | (rsub >> 1));
#pragma unroll
        for (int j = 0; j < PERW; ++j) {
            const int co = bn0 + (j * WAVES + wave) * 8 + rsub;
            b_off[j] = co < p.Cout ? (unsigned)(co * p.Kpad + csw * 8) * 2u : HALO_INV;
        }
#pragma unroll
        for (int j = 0; j < PERH; ++j) dma_piece(rx, lds + (j * WAVES + wave) * 1024, hv[j], 0);
#pragma unroll
        for (int j = 0; j < PERW; ++j) dma_piece(rw, lds + WRING + (j * WAVES + wave) * 1024, b_off[j], 0);
    }

    // ---- fragment base addresses: [0] first half, [1] second half of the K-half-0 chunk, [2], [3] the same for K half 1
    const unsigned c8 = (unsigned)(fq & 1) << 3;
    unsigned hx4[3][4], wa4[4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int sx = UP2 ? ((frow + dx - 1) >> 1) + 1 : frow + dx;          // halo column of this lane's pixel at tap column dx
        const int row0 = (UP2 ? wm * (FP / 2) : wm * FP) * P + sx;             // + the wave's first tile row (source rows when upsampled)
        const unsigned b = (unsigned)(row0 << 7) + (unsigned)(((fq ^ (sx >> 1)) & 7) << 4) + c8;
        hx4[dx][0] = b; hx4[dx][1] = b ^ 8u; hx4[dx][2] = b ^ 64u; hx4[dx][3] = b ^ 72u;
    }
    {
        const unsigned b = (unsigned)((wn * 64 + frow) << 7) + (unsigned)((fq ^ (frow >> 1)) << 4) + c8;
        wa4[0] = b; wa4[1] = b ^ 8u; wa4[2] = b ^ 64u; wa4[3] = b ^ 72u;
    }

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    STAMP(1);

    // One K step = one tap of one channel block; TAP and the parity CB of the channel block are constant expressions.
    //  - W(t) and (tap 0) the block's halo have landed once nothing this wave issued is outstanding; the barrier says so
    //    for every wave, and that every wave is done with step t-1's ring stage;
    //  - all 2 * (FC + FP) fragment reads of the step, then the MFMAs of the first K half with, between them, one piece of
    //    the next channel block's halo (taps 1 .. PERH) and the weights of step t+1; then the second K half.
    auto step = [&](auto tapc, auto cbc, const int cbi, const bool last_cb) {
        constexpr int TAP = decltype(tapc)::value, CB = decltype(cbc)::value;
        constexpr int DY = TAP / 3, DX = TAP % 3;
        constexpr int HOFF = CB * HB;
        constexpr int WOFF = WRING + ((CB + TAP) & 1) * W_BYTES;
        constexpr int T2 = (TAP + 1) % 9, CB2 = CB + (TAP == 8 ? 1 : 0);  // the step whose weights are issued here
        constexpr int WNEXT = WRING + ((CB2 + T2) & 1) * W_BYTES;
        const bool stamped = TAP == 4 && cbi == 1;
        (void)stamped;
        if (stamped) STAMP(3);
        wait_vm<0>();
        if (stamped) STAMP(4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (stamped) STAMP(5);
        const int cb2 = cbi + (TAP == 8 ? 1 : 0);
        const bool w_next = cb2 < p.ncb, h_next = TAP >= 1 && TAP <= PERH && !last_cb;
        auto issue = [&](const int k) {  // piece k of this step's DMA: 0 = halo, 1 .. PERW = weights
            if (k == 0) {
                if constexpr (TAP >= 1 && TAP <= PERH) {
                    if (h_next) dma_piece(rx, lds + (CB ^ 1) * HB + ((TAP - 1) * WAVES + wave) * 1024, hv[TAP - 1], (cbi + 1) * 128);
                }
            } else if (w_next) {
                dma_piece(rw, lds + WNEXT + ((k - 1) * WAVES + wave) * 1024, b_off[k - 1], (T2 * p.Cin + cb2 * 64) * 2);
            }
        };
        // first K half's fragments, then the step's DMA pieces (issued while those reads are in flight), then per half
        // 16 MFMAs; the second half's reads go out under the first half's MFMAs
        auto read_half = [&](auto hc, bf16x8 (&wf)[FC], bf16x8 (&xf)[FP]) {
            constexpr int H = decltype(hc)::value;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                wf[i] = i == 0 ? frag_read_imm<WOFF>(lds3, wa4[2 * H], wa4[2 * H + 1])
                      : i == 1 ? frag_read_imm<WOFF + 2048>(lds3, wa4[2 * H], wa4[2 * H + 1])
                      : i == 2 ? frag_read_imm<WOFF + 4096>(lds3, wa4[2 * H], wa4[2 * H + 1])
                               : frag_read_imm<WOFF + 6144>(lds3, wa4[2 * H], wa4[2 * H + 1]);
            }
#pragma unroll
            for (int f = 0; f < FP; ++f) {
                // row of fragment f at this tap: (f + DY) rows below the wave's first; source rows when upsampled
                auto rd = [&](auto fc) {
                    constexpr int F = decltype(fc)::value;
                    constexpr int ROW = UP2 ? ((F + DY - 1) >> 1) + 1 : F + DY;
                    return frag_read_imm<HOFF + ROW * P * 128>(lds3, hx4[DX][2 * H], hx4[DX][2 * H + 1]);
                };
                xf[f] = f == 0 ? rd(std::integral_constant<int, 0>{}) : f == 1 ? rd(std::integral_constant<int, 1>{})
                      : f == 2 ? rd(std::integral_constant<int, 2>{}) : rd(std::integral_constant<int, 3>{});
            }
        };
        bf16x8 wf0[FC], xf0[FP], wf1[FC], xf1[FP];
        read_half(std::integral_constant<int, 0>{}, wf0, xf0);
#pragma unroll
        for (int k = 0; k < NP; ++k) issue(k);
        if (stamped) STAMP(6);
        read_half(std::integral_constant<int, 1>{}, wf1, xf1);
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], xf0[f], acc[i][f], 0, 0, 0);
        if (stamped) STAMP(7);
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int f = 0; f < FP; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], xf1[f], acc[i][f], 0, 0, 0);
        if (stamped) STAMP(8);
    };
    auto block = [&](auto cbc, const int cbi, const bool last_cb) {
        step(std::integral_constant<int, 0>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 1>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 2>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 3>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 4>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 5>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 6>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 7>{}, cbc, cbi, last_cb);
        step(std::integral_constant<int, 8>{}, cbc, cbi, last_cb);
    };
    for (int cbi = 0; cbi < p.ncb; cbi += 2) {
        block(std::integral_constant<int, 0>{}, cbi, cbi + 1 == p.ncb);
        if (cbi + 1 == p.ncb) break;
        block(std::integral_constant<int, 1>{}, cbi + 1, cbi + 2 == p.ncb);
    }

    STAMP(9);
    halo_epilogue<FC, FP>(p, acc, wm, wn, frow, fq, n0, ty0, tx0, bn0);
#ifdef GAN_HALO_STAMPS
    STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(13);
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 1024)
        for (int k = 0; k < 16; ++k) halo_stamps[(blockIdx.x * WAVES + wave) * 16 + k] = st[k];
#endif
}

}  // namespace

#ifdef GAN_HALO_STAMPS
extern "C" int gan_debug_halo_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(halo_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif

// 1 = served, 0 = shape not eligible (conv_wide.hip's tap-by-tap kernel takes it), < 0 = error.
// GAN_AMD_HALO: 0 = never, 2 = whenever the shape allows (tests); default: when the grid fills the chip.
int gan_conv2d_fwd_wide_halo(const void* x, const void* wt, void* y, const float* out_scale, const float* bias, const void* noise,
                             const float* noise_scale, float slope, int N, int H, int W, int Cin, int up2, int Ho, int Wo, int Cout,
                             int kh, int kw, int py, int px, int Kpad, const ConvFuse& fu, hipStream_t st) {
    const char* env = getenv("GAN_AMD_HALO");
    const int mode = env ? atoi(env) : 1;
    if (mode == 0) return 0;
    if (kh != 3 || kw != 3 || py != 1 || px != 1 || Cin % 64 != 0 || Kpad != 9 * Cin || (Cout & 7) || fu.in_slope != 1.f) return 0;
    if (Ho != (up2 ? 2 * H : H) || Wo != (up2 ? 2 * W : W)) return 0;
    int t8;
    if (Ho % 16 == 0 && Wo % 16 == 0) t8 = 0;
    else if (Ho % 8 == 0 && Wo % 8 == 0) t8 = 1;
    else return 0;
    const int64_t M = (int64_t)N * Ho * Wo;
    const int64_t xb = (int64_t)N * H * W * Cin * 2, wb = (int64_t)Cout * Kpad * 2;
    const int64_t yb = (fu.pool ? M / 4 : M) * Cout * 2;
    if (M >= (1LL << 31) - 256 || xb >= (1LL << 31) || wb >= (1LL << 31) || yb >= (1LL << 31)) return 0;
    if (((uintptr_t)y & 15) || ((uintptr_t)fu.aux & 7) || ((uintptr_t)fu.resid & 7) || ((uintptr_t)out_scale & 15) || ((uintptr_t)bias & 15) ||
        ((uintptr_t)noise_scale & 15))
        return 0;
    HaloArgs a;
    a.x = x; a.wt = wt; a.y = y; a.out_scale = out_scale; a.bias = bias; a.noise = noise; a.noise_scale = noise_scale;
    a.aux = fu.aux; a.resid = fu.resid; a.slope = slope; a.aux_slope = fu.aux_slope; a.gain = fu.gain;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.up2 = up2; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.Kpad = Kpad; a.pool = fu.pool;
    a.t8 = t8;
    // 16-wide tilings run as 4-wave workgroups of 8 x 16 pixels, two per CU: one's prologue and epilogue (a quarter of a
    // wave's lifetime at Cin = 128) overlap the other's K loop (128 -> 128 @ 32x32: 120 -> 106 us; 128 -> 64 upsampled to
    // 64x64: 232 -> 195 us).  The 8 x 8 tiling needs four images' halos at once: 8 waves, 256 pixels, one workgroup per CU.
    const int BN = Cout > 64 ? 128 : 64;
    const int waves = !t8 ? 4 : 8;
    const int TW = t8 ? 8 : 16, TH = t8 ? 8 : 2 * waves, imgs = t8 ? 4 : 1;
    a.tiles_x = Wo / TW; a.tiles_y = Ho / TH;
    a.P = (up2 ? TW / 2 : TW) + 2;   // halo row pitch and rows per image, in source pixels
    a.HH = (up2 ? TH / 2 : TH) + 2;
    // image stride: 8 mod 16 for the two-image fragments of the 8 x 8 tiling (the two runs of 8 rows then fall on
    // complementary bank halves)
    a.IS = a.P * a.HH;
    if (t8)
        while (a.IS % 16 != 8) ++a.IS;
    const int hpx = imgs * a.IS;
    a.perh = (hpx + 8 * waves - 1) / (8 * waves);
    if (a.perh > HALO_MAXP) return 0;
    a.ncb = Cin / 64;
    a.wfrag = 16 * 128;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
    const int64_t tiles = (int64_t)((N + imgs - 1) / imgs) * a.tiles_x * a.tiles_y;
    const int64_t gn = (Cout + BN - 1) / BN;
    if (tiles * gn >= (1LL << 31)) return 0;
    static int cus_of[64];  // CUs per device, asked once (benign race: every thread writes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus_of[dev] == 0 && hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus_of[dev] = 0;
    const int cus = cus_of[dev];
    if (cus <= 0) return 0;
    const int64_t slots = (int64_t)cus * (waves == 4 ? 2 : 1);  // workgroups resident at once (LDS-bound: 80 / 144..160 KiB each)
    // fewer work items than slots: the 128-pixel tiles and deeper ring of conv_wide.hip fill more of the chip
    if (mode != 2 && tiles * gn < slots - slots / 8) return 0;
    a.ntiles = (int)tiles;
    const int64_t items = tiles * gn;
    const size_t lds = (size_t)2 * a.perh * waves * 1024 + (size_t)(waves == 8 ? 3 : 2) * BN * 128;
    const dim3 grid((unsigned)items);
    static const bool halo4 = !(getenv("GAN_AMD_HALO4") && atoi(getenv("GAN_AMD_HALO4")) == 0);  // 0: the round-2 kernel (A/B runs)
    if (waves == 4 && halo4 && BN == 128 && up2)
        hipLaunchKernelGGL((conv_wide_halo4_k<128, true>), grid, dim3(256), lds, st, a);
    else if (waves == 4 && halo4 && BN == 128)
        hipLaunchKernelGGL((conv_wide_halo4_k<128, false>), grid, dim3(256), lds, st, a);
    else if (waves == 4 && halo4 && up2)
        hipLaunchKernelGGL((conv_wide_halo4_k<64, true>), grid, dim3(256), lds, st, a);
    else if (waves == 4 && halo4)
        hipLaunchKernelGGL((conv_wide_halo4_k<64, false>), grid, dim3(256), lds, st, a);
    else if (waves == 4 && BN == 128 && up2)
        hipLaunchKernelGGL((conv_wide_halo_k<128, 4, true>), grid, dim3(256), lds, st, a);
    else if (waves == 4 && BN == 128)
        hipLaunchKernelGGL((conv_wide_halo_k<128, 4, false>), grid, dim3(256), lds, st, a);
    else if (waves == 4 && up2)
        hipLaunchKernelGGL((conv_wide_halo_k<64, 4, true>), grid, dim3(256), lds, st, a);
    else if (waves == 4)
        hipLaunchKernelGGL((conv_wide_halo_k<64, 4, false>), grid, dim3(256), lds, st, a);
    else if (BN == 128 && up2)
        hipLaunchKernelGGL((conv_wide_halo_k<128, 8, true>), grid, dim3(512), lds, st, a);
    else if (BN == 128)
        hipLaunchKernelGGL((conv_wide_halo_k<128, 8, false>), grid, dim3(512), lds, st, a);
    else if (up2)
        hipLaunchKernelGGL((conv_wide_halo_k<64, 8, true>), grid, dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((conv_wide_halo_k<64, 8, false>), grid, dim3(512), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gan_set_error("conv2d_fwd(wide, halo): launch failed: %s", hipGetErrorString(e));
        return GAN_ERR_LAUNCH;
    }
    return 1;
}
