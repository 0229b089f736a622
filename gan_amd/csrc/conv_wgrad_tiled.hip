// Weight gradient for WIDE layers / small feature maps on MFMA (gfx950), bf16, plus the slice reducer
// shared with conv_thin_wgrad.hip.
//
//   dw[tap][co][ci] = sum_{n,oy,ox} gy[n,oy,ox,co] * X[n, oy+dy-py, ox+dx-px, ci] * in_scale[n][ci]
//
// Same idea as the thin kernel -- stage a block of pixels ONCE in LDS and form every tap from that
// image -- organised for 128..512 channels on 4x4..64x64 maps:
//   * a workgroup owns a 64 x 64 (co x ci) channel tile and a contiguous range of 128-pixel
//     spatial tiles: TH x 32 rows of one image, half / whole 16 x 16 images, 2 images of 8 x 8, 8 of 4 x 4;
//   * consumer wave w (four, one per SIMD) owns input channels 16w .. 16w+15 of the tile: taps x 4 fragment
//     accumulators (144 registers for 3x3) that live across all tiles of the workgroup;
//   * k = 32 consecutive tile-linear pixels per MFMA; both operands are pixel-major, stored as
//     [16-channel group][pixel][16 ch] and read with ds_read_b64_tr_b16 (per-lane halo offsets);
//   * split-K over pixel tiles: every workgroup writes its partial tile with plain stores into slice
//     s of a caller-provided workspace [S][taps][Cout][Cin]; finalize_dw_k folds the slices.
//     (fp32 atomics from ~500 workgroups into one 147 KB tile ran at 0.3 TB/s and dominated.)
#include "common.h"

namespace {

struct TWArgs {
    const void* x;
    const void* gy;
    float* ws;
    const float* in_scale;
    int N, H, W, Hl, Wl, up2, Ho, Wo, Cin, Cout, py, px;
    int tiles_x, tiles_y;           // per image (NB == 1)
    int ntiles, S, ci_tiles;
    unsigned x_bytes, gy_bytes;
    WgradFuse fu;
    int want_bias;  // the workgroups of ci tile 0 also sum gy over the pixels (wave 7): Cout floats behind every slice
    int64_t Lp;     // slice stride: taps * Cout * Cin (+ Cout)
};

__device__ __forceinline__ uint4 tmod_chunk(uint4 raw, const float* s) {
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

__device__ __forceinline__ bf16x8 tr_frag2(const char* plo, const char* phi) {
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)plo);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)phi);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

#ifndef GAN_TW_KS_UNROLL
#define GAN_TW_KS_UNROLL 4
#endif
#ifndef GAN_TW_LOADERS
#define GAN_TW_LOADERS 4
#endif
constexpr int TW_CONSUMERS = 4;               // MFMA consumer waves 0..3, one per SIMD: wave w owns input channels 16w .. 16w+15 of the tile
constexpr int TW_LOADERS = GAN_TW_LOADERS;    // producer waves (waves 4..7)
constexpr int TW_BIAS_WAVE = TW_CONSUMERS + TW_LOADERS;  // one more wave: sum_pixels gy = gy^T x ones on the MFMA (the bias gradient)
constexpr int TW_THREADS = 64 * TW_CONSUMERS + 64 * TW_LOADERS + 64;


// Tile geometry is a template parameter: the producers turn a chunk index into (image, row, column) with
// divisions by HALO and HWD for every 16-byte load, which as run-time divisions made them VALU-bound
// (PMC: consumers waited 67 % of their cycles on the stage barrier).
template <int KS, int TWL, int THL, int NBL, bool MOD, bool BIAS>
__global__ void __launch_bounds__(BIAS ? TW_THREADS : TW_THREADS - 64) conv_wgrad_tiled_k(const TWArgs p) {
    extern __shared__ __attribute__((aligned(256))) char tw_lds[];
    constexpr int NT = KS;  // taps (dx) per consumer wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NB = 1 << NBL;
    constexpr int HWD = TW + KS - 1, HALO = HWD * (TH + KS - 1), NPX = NB * HALO;
    constexpr int PLX = NPX * 32, PLG = 128 * 32;
    constexpr int BUF = 4 * PLX + 4 * PLG;  // one stage: x planes, then gy planes
    const int co0 = (blockIdx.x / p.ci_tiles) * 64, ci0 = (blockIdx.x % p.ci_tiles) * 64;
    const int s = blockIdx.y;
    const int t0 = (int)((int64_t)p.ntiles * s / p.S), t1 = (int)((int64_t)p.ntiles * (s + 1) / p.S);
    constexpr int thw_log2 = THL + TWL;
    const int tiles_per_img = p.tiles_x * p.tiles_y;

    if (BIAS && wave == TW_BIAS_WAVE) {  // (a template flag: with the branch compiled in, the consumers spill 12-36 bytes more and lose 12 %)
        // ================= the bias gradient: the gy planes of every stage against a fragment of ones ==============
        // (the consumers' 192 accumulator registers leave no room for it; a wave of its own has its own registers and
        // adds 16 MFMAs per tile to a SIMD that runs 192.  Every column of the result holds the sum.)
        const bool act = p.want_bias && ci0 == 0;
        f32x4 accb[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
        for (int t = t0; t < t1; ++t) {
            __syncthreads();  // tile t is in its stage (the consumers' barrier)
            if (!act) continue;
            const char* const gl = tw_lds + ((t - t0) & 1) * BUF + 4 * PLX;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int Plo = ks * 32 + 4 * g + q, Phi = Plo + 16;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const bf16x8 a = tr_frag2(gl + o * PLG + Plo * 32 + pp * 8, gl + o * PLG + Phi * 32 + pp * 8);
                    accb[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, accb[o], 0, 0, 0);
                }
            }
        }
        if (act && li == 0) {
            float* __restrict__ dst = p.ws + (int64_t)s * p.Lp + (int64_t)KS * KS * p.Cout * p.Cin;
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + o * 16 + g * 4 + r;
                    if (co < p.Cout) dst[co] = accb[o][r];
                }
        }
    } else if (wave >= TW_CONSUMERS) {
        // ================= producers: tile t -> LDS stage (t - t0) & 1, TW_U loads in flight per lane ====
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        constexpr unsigned INV = 0x80000000u;  // past num_records (tensors < 2^31 bytes): the load returns zeros
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gy), 0, p.gy_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_scale), 0, MOD ? (unsigned)(p.N * p.Cin) * 4u : 0u, 0x00020000);
        const int ltid = tid - 64 * TW_CONSUMERS;
        constexpr int PL = 64 * TW_LOADERS;
        constexpr int nxc = ((NPX + 7) >> 3) * 64;  // x chunks (8 pixels x 8 channel chunks per 64)
        // Branch-free, division-free staging.  Slot `sl` of this lane always maps to the same (image b, row,
        // col, channel chunk) of a tile; slots [0, NSX) are x chunks and [NSX, NSLOT) gy chunks, so which
        // tensor a slot reads is known at compile time.  Tile origins are even, which makes the source
        // offset LINEAR in the tile origin even through the 2x gathers ((o + r) >> 1 = (o >> 1) + (r >> 1)
        // for even o): decoded ONCE here into rel[] (byte offset from the tile's base) and packed[] (halo
        // coordinates for the bounds test), a slot then costs ~10 VALU ops per tile, one buffer load (lanes
        // past the image read zeros at offset INV) and one LDS store (chunks past the tile go to a per-lane
        // dump slot behind the two stages).  History: `if`s around the loads cost a branch and a vmcnt wait per
        // load, run-time divisions ~115 VALU ops per load, the full offset arithmetic per tile ~40 (the
        // producers, not the MFMAs, set the pace: 2 -> 4 producer waves gave 1.3-1.5x).
        constexpr int NSX = (nxc + PL - 1) / PL, NSG = (128 * 8 + PL - 1) / PL, NSLOT = NSX + NSG;
        constexpr int U = MOD ? 7 : NSLOT;  // loads in flight per lane and batch
        int packed[NSLOT], dsto[NSLOT], rel[NSLOT], cls[NSLOT];
        const int c8l = (ltid >> 3) & 7;  // channel chunk: the same for every slot of a lane
        const int cix = ci0 + c8l * 8, cog = co0 + c8l * 8;
        const int shx = p.up2, shg = p.fu.gy_up2;
        const int Hg = p.Ho >> shg, Wg = p.Wo >> shg;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const bool isx = sl < NSX;
            const int idx = (isx ? sl : sl - NSX) * PL + ltid;
            const int pix = (idx >> 6) * 8 + (idx & 7);
            int bb, yy, xx, okc, d;
            if (isx) {
                bb = pix / HALO;
                const int r = pix - bb * HALO;
                yy = r / HWD;
                xx = r - yy * HWD;
                okc = pix < NPX && cix < p.Cin;
                d = pix < NPX ? (c8l >> 1) * PLX + pix * 32 + (c8l & 1) * 16 : -1;
                rel[sl] = (((bb * p.H + ((yy - p.py) >> shx)) * p.W + ((xx - p.px) >> shx)) * p.Cin + cix) * 2;
            } else {
                bb = pix >> thw_log2;
                yy = (pix >> TWL) & (TH - 1);
                xx = pix & (TW - 1);
                okc = idx < 128 * 8 && cog < p.Cout;
                d = idx < 128 * 8 ? 4 * PLX + (c8l >> 1) * PLG + pix * 32 + (c8l & 1) * 16 : -1;
                rel[sl] = (((bb * Hg + (yy >> shg)) * Wg + (xx >> shg)) * p.Cout + cog) * 2;
            }
            packed[sl] = yy | (xx << 8) | (bb << 16) | (okc << 25);
            dsto[sl] = d;
            // when the tiles divide the map (`regular` below), a slot's validity is a STATIC class tested against four
            // per-tile flags: bit 0 / 1 = outside the image when the tile touches the top / bottom edge, bit 2 / 3 = left /
            // right edge, bit 4 = never valid (channel or chunk past the tile)
            int c = okc ? 0 : 16;
            if (isx) {
                if (yy < p.py) c |= 1;
                if (yy >= (p.Hl - p.Ho) + TH + p.py) c |= 2;
                if (xx < p.px) c |= 4;
                if (xx >= (p.Wl - p.Wo) + TW + p.px) c |= 8;
            }
            cls[sl] = c;
        }
        auto origin = [&](const int t, int& n0, int& oy0, int& ox0) {
            if (NB == 1) {
                n0 = t / tiles_per_img;
                const int rem = t - n0 * tiles_per_img;
                const int ty = rem / p.tiles_x;
                oy0 = ty * TH;
                ox0 = (rem - ty * p.tiles_x) * TW;
            } else {
                n0 = t * NB;
                oy0 = ox0 = 0;
            }
        };
        if constexpr (!MOD) {
            // Round 3: the loads of tile t+1 are issued right after tile t's chunks have gone to LDS, BEFORE the stage
            // barrier, and stay in flight in registers while the producers wait for the consumers: the memory latency
            // of a tile no longer sits between two barriers (a tile took ~5700 cycles for 2300 of MFMA work).
            v4u v[NSLOT];
            // tiles that divide the map and the batch: ~4 VALU instructions per chunk instead of ~27 (PMC, round 3: the
            // producers' 350 VALU instructions per tile, issued on SIMDs whose vector issue the consumers' MFMAs hold
            // half of the time, paced the kernel: 44 % MFMA utilisation with no wait on memory)
            const bool regular = p.Ho % TH == 0 && p.Wo % TW == 0 && (NB == 1 || p.N % NB == 0);
            auto issue = [&](const int t) {
                int n0, oy0, ox0;
                origin(t, n0, oy0, ox0);
                const int bx = ((n0 * p.H + (oy0 >> shx)) * p.W + (ox0 >> shx)) * p.Cin * 2;
                const int bg = ((n0 * Hg + (oy0 >> shg)) * Wg + (ox0 >> shg)) * p.Cout * 2;
                if (regular) {
                    const int flags = 16 | (oy0 == 0 ? 1 : 0) | (oy0 + TH == p.Ho ? 2 : 0) | (ox0 == 0 ? 4 : 0) | (ox0 + TW == p.Wo ? 8 : 0);
#pragma unroll
                    for (int k = 0; k < NSLOT; ++k) {
                        const bool isx = k < NSX;
                        const unsigned off = (cls[k] & flags) ? INV : (unsigned)((isx ? bx : bg) + rel[k]);
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(isx ? rx : rg, off, 0, 0);
                    }
                    return;
                }
#pragma unroll
                for (int k = 0; k < NSLOT; ++k) {
                    const bool isx = k < NSX;
                    int pk = packed[k];
                    asm volatile("" : "+v"(pk));  // keeps the decode below out of the tile loop's preheader (registers)
                    const int nn = n0 + ((pk >> 16) & 0xff);
                    const int y = (isx ? oy0 - p.py : oy0) + (pk & 0xff);
                    const int xq = (isx ? ox0 - p.px : ox0) + ((pk >> 8) & 0xff);
                    const bool ok = ((pk >> 25) & 1) && nn < p.N && (unsigned)y < (unsigned)(isx ? p.Hl : p.Ho) &&
                                    (unsigned)xq < (unsigned)(isx ? p.Wl : p.Wo);
                    const unsigned off = ok ? (unsigned)((isx ? bx : bg) + rel[k]) : INV;
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(isx ? rx : rg, off, 0, 0);
                }
            };
            if (t0 < t1) issue(t0);
            for (int t = t0; t < t1; ++t) {
                char* const xl = tw_lds + ((t - t0) & 1) * BUF;
                const int dump = 2 * BUF + ltid * 16 - ((t - t0) & 1) * BUF;
#pragma unroll
                for (int k = 0; k < NSLOT; ++k) {
                    uint4 u = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
                    if (k < NSX && p.fu.x_slope != 1.f) u = lrelu_chunk(u, p.fu.x_slope, bf16());
                    const int d = dsto[k];
                    *reinterpret_cast<uint4*>(xl + (d >= 0 ? d : dump)) = u;
                }
                if (t + 1 < t1) issue(t + 1);
                __syncthreads();  // stage ready; consumers have left the stage this overwrites two tiles later
            }
        } else {
        for (int t = t0; t < t1; ++t) {
            char* const xl = tw_lds + ((t - t0) & 1) * BUF;
            int n0, oy0, ox0;
            origin(t, n0, oy0, ox0);
            // wave-uniform bases of the tile (bytes)
            const int bx = ((n0 * p.H + (oy0 >> shx)) * p.W + (ox0 >> shx)) * p.Cin * 2;
            const int bg = ((n0 * Hg + (oy0 >> shg)) * Wg + (ox0 >> shg)) * p.Cout * 2;
            const int dump = 2 * BUF + ltid * 16 - ((t - t0) & 1) * BUF;
#pragma unroll
            for (int b0 = 0; b0 < NSLOT; b0 += U) {
                v4u v[U];
                v4u s0[MOD ? U : 1], s1[MOD ? U : 1];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (b0 + k < NSLOT) {
                        const bool isx = b0 + k < NSX;
                        int pk = packed[b0 + k];
                        asm volatile("" : "+v"(pk));  // keeps the decode below out of the tile loop's preheader (registers)
                        const int nn = n0 + ((pk >> 16) & 0xff);
                        const int y = (isx ? oy0 - p.py : oy0) + (pk & 0xff);
                        const int xq = (isx ? ox0 - p.px : ox0) + ((pk >> 8) & 0xff);
                        const bool ok = ((pk >> 25) & 1) && nn < p.N && (unsigned)y < (unsigned)(isx ? p.Hl : p.Ho) &&
                                        (unsigned)xq < (unsigned)(isx ? p.Wl : p.Wo);
                        const unsigned off = ok ? (unsigned)((isx ? bx : bg) + rel[b0 + k]) : INV;
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(isx ? rx : rg, off, 0, 0);
                        if (MOD && isx) {
                            const unsigned so = ok ? (unsigned)(nn * p.Cin + cix) * 4u : INV;
                            s0[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, so, 0, 0);
                            s1[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, so + 16u, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (b0 + k < NSLOT) {
                        const bool isx = b0 + k < NSX;
                        uint4 u = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
                        if (isx && p.fu.x_slope != 1.f) u = lrelu_chunk(u, p.fu.x_slope, bf16());
                        if (MOD && isx) {
                            bf16x8 h = *reinterpret_cast<bf16x8*>(&u);
                            const float4 a0 = *reinterpret_cast<const float4*>(&s0[k]);
                            const float4 a1 = *reinterpret_cast<const float4*>(&s1[k]);
                            h[0] = (bf16)((float)h[0] * a0.x); h[1] = (bf16)((float)h[1] * a0.y);
                            h[2] = (bf16)((float)h[2] * a0.z); h[3] = (bf16)((float)h[3] * a0.w);
                            h[4] = (bf16)((float)h[4] * a1.x); h[5] = (bf16)((float)h[5] * a1.y);
                            h[6] = (bf16)((float)h[6] * a1.z); h[7] = (bf16)((float)h[7] * a1.w);
                            u = *reinterpret_cast<uint4*>(&h);
                        }
                        const int d = dsto[b0 + k];
                        *reinterpret_cast<uint4*>(xl + (d >= 0 ? d : dump)) = u;
                    }
                }
            }
            __syncthreads();  // stage ready; consumers have left the stage this overwrites two tiles later
        }
        }
    } else {
        // ================= consumers: wave = 16-channel group of the tile's input channels, every tap ============
        // (round 3.  Rounds 1-2 gave each of THREE waves a kernel row: a CU's fourth SIMD issued no MFMA, and 1x1 layers,
        // whose waves split the K steps instead, folded three partial tiles through LDS at the end.  With the ci split a
        // tile is 144 MFMAs on each of four SIMDs instead of 192 on three, the accumulators are 144 registers (were
        // 192), and every wave stores its own columns of every tap.)
        f32x4 acc[NT * NT][4];
        // K steps unrolled: the next step's fragment reads go out under this step's MFMAs (rolled, a tile exposed four LDS
        // latencies: 84 -> 79 us on 128 ch @ 32x32, 103 -> 83 us on 256 ch @ 16x16).  Not with the bias wave: nine waves
        // cap the registers at 168.  (Measured, not kept: s_setprio 2 for the consumers -- no change on nine shapes.)
        constexpr int KSU = BIAS ? 1 : GAN_TW_KS_UNROLL;
#pragma unroll
        for (int j = 0; j < NT * NT; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[j][o] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = t0; t < t1; ++t) {
            __syncthreads();  // tile t is in its stage
            const char* const xl = tw_lds + ((t - t0) & 1) * BUF + wave * PLX;
            const char* const gl = tw_lds + ((t - t0) & 1) * BUF + 4 * PLX;
#pragma unroll KSU
            for (int ks = 0; ks < 4; ++ks) {
                const int Plo = ks * 32 + 4 * g + q, Phi = Plo + 16;
                bf16x8 a[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) a[o] = tr_frag2(gl + o * PLG + Plo * 32 + pp * 8, gl + o * PLG + Phi * 32 + pp * 8);
                const int hlo = (Plo >> thw_log2) * HALO + ((Plo >> TWL) & (TH - 1)) * HWD + (Plo & (TW - 1));
                const int hhi = (Phi >> thw_log2) * HALO + ((Phi >> TWL) & (TH - 1)) * HWD + (Phi & (TW - 1));
                const char* const blo = xl + hlo * 32 + pp * 8;
                const char* const bhi = xl + hhi * 32 + pp * 8;
#pragma unroll
                for (int dy = 0; dy < NT; ++dy)
#pragma unroll
                    for (int dx = 0; dx < NT; ++dx) {
                        const bf16x8 b = tr_frag2(blo + (dy * HWD + dx) * 32, bhi + (dy * HWD + dx) * 32);
#pragma unroll
                        for (int o = 0; o < 4; ++o)
                            acc[dy * NT + dx][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[o], b, acc[dy * NT + dx][o], 0, 0, 0);
                    }
            }
        }
        // ---- partial tile -> workspace slice s, final layout [tap][Cout][Cin]
        float* __restrict__ slice = p.ws + (int64_t)s * p.Lp;
        const int ci = ci0 + wave * 16 + li;
#pragma unroll
        for (int tap = 0; tap < NT * NT; ++tap) {
            float* __restrict__ dst = slice + (int64_t)tap * p.Cout * p.Cin;
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + o * 16 + g * 4 + r;
                    if (co < p.Cout && ci < p.Cin) dst[(int64_t)co * p.Cin + ci] = acc[tap][o][r];
                }
        }
    }
}

// dw[co][ci][tap] (+)= gain * sum_s ws[s][tap][co][ci]: folds the split-K slices and writes the parameter's own
// layout (cout x cin x taps, channel padding dropped), so the caller needs neither a zeroed target nor a
// permute / slice copy afterwards.  A block owns 64 quads of consecutive ci; its SG waves split the slices.
template <int SG, int QL>
__global__ void __launch_bounds__(QL * SG) finalize_dw_k(const float* __restrict__ ws, float* __restrict__ out, int64_t L, int64_t Lp, int S,
                                                         int taps, int Cout, int Cin, int cout_o, int cin_o, float gain, int accum,
                                                         float* __restrict__ bias, float bias_gain, int bias_accum) {
    // QL lanes of quads x SG slice groups (thin layers: 512 slices of a few thousand floats -- 64 groups of 16 lanes
    // spread them over 4x the workgroups of 16 x 64 and sum 8 slices per thread instead of 32: 8.3 -> 6 us)
    __shared__ float4 part[SG * QL];
    const int lane = threadIdx.x % QL, sg = threadIdx.x / QL;
    const int64_t e4 = (int64_t)blockIdx.x * QL + lane;
    const bool live = e4 * 4 < Lp;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float* __restrict__ src = ws + e4 * 4;
#pragma unroll 8
        for (int s = sg; s < S; s += SG) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)s * Lp);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    if (SG > 1) {  // tree over the slice groups, fixed order
        part[sg * QL + lane] = a;
        __syncthreads();
#pragma unroll
        for (int st = SG / 2; st > 0; st >>= 1) {
            if (sg < st) {
                const float4 u = part[sg * QL + lane], v = part[(sg + st) * QL + lane];
                part[sg * QL + lane] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
            }
            __syncthreads();
        }
        a = part[lane];
    }
    if (sg != 0 || !live) return;
    const int64_t idx = e4 * 4;
    if (idx >= L) {  // the channel sums behind the slice: the bias gradient
        const int c = (int)(idx - L);
        const float r[4] = {a.x * bias_gain, a.y * bias_gain, a.z * bias_gain, a.w * bias_gain};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (bias && c + j < cout_o) bias[c + j] = bias_accum ? bias[c + j] + r[j] : r[j];
        return;
    }
    const int plane = Cout * Cin;
    const int tap = (int)(idx / plane);
    const int rem = (int)(idx - (int64_t)tap * plane);
    const int co = rem / Cin, ci = rem - co * Cin;  // Cin % 4 == 0: the quad stays in one row
    if (co >= cout_o) return;
    const float r[4] = {a.x * gain, a.y * gain, a.z * gain, a.w * gain};
    float* __restrict__ o = out + ((int64_t)co * cin_o + ci) * taps + tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < cin_o) o[(int64_t)j * taps] = accum ? o[(int64_t)j * taps] + r[j] : r[j];
}

struct TiledPlan {
    bool ok;
    int tw_log2, th_log2, nb_log2, tiles_x, tiles_y, ntiles, S, co_tiles, ci_tiles, lds;
};

TiledPlan plan_tiled(int N, int Ho, int Wo, int Cin, int Cout, int ks) {
    TiledPlan pl;
    pl.ok = false;
    if (Wo >= 32) {
        pl.tw_log2 = 5; pl.th_log2 = 2; pl.nb_log2 = 0;
    } else if (Wo == 16 && Ho == 16) {
        pl.tw_log2 = 4; pl.th_log2 = 3; pl.nb_log2 = 0;
    } else if (Wo == 8 && Ho == 8) {
        pl.tw_log2 = 3; pl.th_log2 = 3; pl.nb_log2 = 1;
    } else if (Wo == 4 && Ho == 4) {
        pl.tw_log2 = 2; pl.th_log2 = 2; pl.nb_log2 = 3;
    } else {
        return pl;
    }
    const int TW = 1 << pl.tw_log2, TH = 1 << pl.th_log2, NB = 1 << pl.nb_log2;
    pl.tiles_x = (Wo + TW - 1) / TW;
    pl.tiles_y = (Ho + TH - 1) / TH;
    const int64_t nt = NB == 1 ? (int64_t)N * pl.tiles_x * pl.tiles_y : (N + NB - 1) / NB;
    if (nt <= 0 || nt >= (1LL << 30)) return pl;
    pl.ntiles = (int)nt;
    pl.co_tiles = (Cout + 63) / 64;
    pl.ci_tiles = (Cin + 63) / 64;
    const int ct = pl.co_tiles * pl.ci_tiles;
    int S = (256 + ct - 1) / ct;  // one 7-wave workgroup per CU (two LDS stages, 256-register consumers)
    if (S > pl.ntiles) S = pl.ntiles;
    if (S > 65535) S = 65535;
    pl.S = S;
    const int NPX = NB * (TW + ks - 1) * (TH + ks - 1);
    pl.lds = 2 * (4 * NPX * 32 + 4 * 128 * 32) + 64 * TW_LOADERS * 16;  // two stages + the producers' dump slots
    pl.ok = true;
    return pl;
}

}  // namespace

// floats of workspace gan_conv2d_wgrad_tiled needs for this shape (0: shape not served by it)
int64_t gan_wgrad_tiled_workspace(int N, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int py, int px) {
    // 1x1 / 3x3 "same" convolutions, and 2x2 valid ones (stride-2 layers folded by space-to-depth: gan_space_to_depth)
    if (kh != kw || px != py || !((kh == 1 && py == 0) || (kh == 3 && py == 1) || (kh == 2 && py == 0))) return 0;
    if ((Cin & 7) || (Cout & 7)) return 0;
    if ((int64_t)N * Ho * Wo * (Cin > Cout ? Cin : Cout) >= (1LL << 30)) return 0;  // 32-bit byte offsets, INV past the end
    const TiledPlan pl = plan_tiled(N, Ho, Wo, Cin, Cout, kh);
    if (!pl.ok) return 0;
    return (int64_t)pl.S * ((int64_t)kh * kw * Cout * Cin + Cout);  // (room for the channel sums whether or not a call wants them)
}

// 1 = served, 2 = dw served but not dwo.bias, 0 = not eligible / workspace too small, < 0 = error.
int gan_conv2d_wgrad_tiled(const void* x, const void* gy, const DwOut& dwo, const float* in_scale, int N, int H, int W,
                           int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, float* ws,
                           int64_t ws_floats, const WgradFuse& fu, hipStream_t st) {
    const int64_t need = gan_wgrad_tiled_workspace(N, Ho, Wo, Cin, Cout, kh, kw, py, px);
    if (need == 0 || ws == nullptr || ws_floats < need || (((uintptr_t)ws) & 15)) return 0;
    const TiledPlan pl = plan_tiled(N, Ho, Wo, Cin, Cout, kh);
    TWArgs a;
    a.fu = fu;
    a.x = x; a.gy = gy; a.ws = ws; a.in_scale = in_scale;
    a.N = N; a.H = H; a.W = W; a.Hl = up2 ? 2 * H : H; a.Wl = up2 ? 2 * W : W; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cin = Cin; a.Cout = Cout; a.py = py; a.px = px;
    a.x_bytes = (unsigned)((int64_t)N * H * W * Cin * 2);
    a.gy_bytes = (unsigned)(((int64_t)N * (Ho >> fu.gy_up2) * (Wo >> fu.gy_up2)) * Cout * 2);
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.ntiles = pl.ntiles; a.S = pl.S; a.ci_tiles = pl.ci_tiles;
    // The bias wave costs the kernel ~12 % (compiled in, the consumers spill 12-36 bytes more: 157 against 132 us on
    // 128 ch @ 32^2, batch 256) and saves a channel-sum pass and its fold (12-17 us): taken where 12 % of the kernel is
    // less than that, i.e. below ~65 us of MFMA work.
    a.want_bias = dwo.bias != nullptr && in_scale == nullptr && (int64_t)N * Ho * Wo * Cin * Cout <= (1LL << 31);
    a.Lp = (int64_t)kh * kw * Cout * Cin + (a.want_bias ? Cout : 0);
    const dim3 block(a.want_bias ? TW_THREADS : TW_THREADS - 64);  // (without the bias wave the barriers count seven waves)
    const dim3 grid((unsigned)(pl.co_tiles * pl.ci_tiles), (unsigned)pl.S);
#define TW_LAUNCH2(KS_, TWL_, THL_, NBL_, MOD_, BIAS_) \
    hipLaunchKernelGGL((conv_wgrad_tiled_k<KS_, TWL_, THL_, NBL_, MOD_, BIAS_>), grid, block, (size_t)pl.lds, st, a)
#define TW_LAUNCH(KS_, TWL_, THL_, NBL_)                                  \
    do {                                                                  \
        if (in_scale) TW_LAUNCH2(KS_, TWL_, THL_, NBL_, true, false);     \
        else if (a.want_bias) TW_LAUNCH2(KS_, TWL_, THL_, NBL_, false, true); \
        else TW_LAUNCH2(KS_, TWL_, THL_, NBL_, false, false);             \
    } while (0)
    if (kh == 3) {
        if (pl.tw_log2 == 5) TW_LAUNCH(3, 5, 2, 0);
        else if (pl.tw_log2 == 4) TW_LAUNCH(3, 4, 3, 0);
        else if (pl.tw_log2 == 3) TW_LAUNCH(3, 3, 3, 1);
        else TW_LAUNCH(3, 2, 2, 3);
    } else if (kh == 2) {
        if (pl.tw_log2 == 5) TW_LAUNCH(2, 5, 2, 0);
        else if (pl.tw_log2 == 4) TW_LAUNCH(2, 4, 3, 0);
        else if (pl.tw_log2 == 3) TW_LAUNCH(2, 3, 3, 1);
        else TW_LAUNCH(2, 2, 2, 3);
    } else {
        if (pl.tw_log2 == 5) TW_LAUNCH(1, 5, 2, 0);
        else if (pl.tw_log2 == 4) TW_LAUNCH(1, 4, 3, 0);
        else if (pl.tw_log2 == 3) TW_LAUNCH(1, 3, 3, 1);
        else TW_LAUNCH(1, 2, 2, 3);
    }
#undef TW_LAUNCH
#undef TW_LAUNCH2
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gan_set_error("conv2d_wgrad(tiled): launch failed: %s", hipGetErrorString(e));
        return GAN_ERR_LAUNCH;
    }
    // (gy at half resolution is staged four times per source pixel: the sums come out 4x)
    const int rc = gan_finalize_dw(ws, pl.S, kh * kw, Cout, Cin, dwo, fu.gain, st, a.want_bias, fu.gy_up2 ? 0.25f : 1.f);
    if (rc != GAN_OK) return rc;
    return dwo.bias != nullptr && !a.want_bias ? 2 : 1;  // 2: dw served, the bias gradient is left to the caller
}

int gan_finalize_dw(const float* ws, int S, int taps, int Cout, int Cin, const DwOut& dwo, float gain, hipStream_t st, int bias_plane,
                    float bias_gain) {
    const int64_t L = (int64_t)taps * Cout * Cin;
    if ((Cin & 3) || S < 1 || dwo.cout > Cout || dwo.cin > Cin || (bias_plane && (Cout & 3))) {
        gan_set_error("finalize_dw: bad shape (Cin %d must be a multiple of 4, S %d >= 1)", Cin, S);
        return GAN_ERR_ARG;
    }
    const int64_t Lp = L + (bias_plane ? Cout : 0);
    const int64_t quads = ceil_div64(Lp, 4);
#define FIN_ARGS ws, dwo.out, L, Lp, S, taps, Cout, Cin, dwo.cout, dwo.cin, gain, dwo.accumulate, dwo.bias, bias_gain, dwo.bias_accumulate
    if (S <= 8)
        hipLaunchKernelGGL((finalize_dw_k<1, 64>), dim3((unsigned)ceil_div64(quads, 64)), dim3(64), 0, st, FIN_ARGS);
    else if (S <= 64)
        hipLaunchKernelGGL((finalize_dw_k<4, 64>), dim3((unsigned)ceil_div64(quads, 64)), dim3(256), 0, st, FIN_ARGS);
    else if (quads <= 8192)
        hipLaunchKernelGGL((finalize_dw_k<64, 16>), dim3((unsigned)ceil_div64(quads, 16)), dim3(1024), 0, st, FIN_ARGS);
    else
        hipLaunchKernelGGL((finalize_dw_k<16, 64>), dim3((unsigned)ceil_div64(quads, 64)), dim3(1024), 0, st, FIN_ARGS);
#undef FIN_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gan_set_error("finalize_dw: launch failed: %s", hipGetErrorString(e));
        return GAN_ERR_LAUNCH;
    }
    return GAN_OK;
}
