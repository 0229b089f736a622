// Implicit-GEMM convolution for WIDE layers on MFMA (gfx950), bf16: Cin % 64 == 0 (128..512 channels on
// 4x4..64x64 maps: the inner blocks of the StyleGAN2 generator and discriminator, modelling/stylegan2.py:29-36,
// modelling/progressive_gan.py:45-50,61, and their input gradients).
//
//   D[co][m] = sum_k Wt[co][k] * X[pix(m) + tap(k)][ci(k)],   k = tap*Cin + ci,   M = N*Ho*Wo pixels
//
// Same GEMM as conv_fwd.hip; what differs is how the operand tiles reach LDS and how deep a step is:
//   * BK = 64: a K step is 64 consecutive channels of ONE tap (Cin % 64 == 0), i.e. one 128-byte run per
//     pixel row and per weight row;
//   * both tiles are written to LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 bytes per lane, 1 KiB per
//     wave instruction): no staging registers, no ds_write pass, no VALU beyond the address; lanes whose tap
//     falls outside the image (zero padding) or past M / Cout get an out-of-range buffer offset and the
//     hardware writes zeros.  Nearest-2x upsampling is part of the address;
//   * the LDS image of a wave instruction is lane-linear (8 rows x 128 B), so the bank swizzle is applied on
//     the SOURCE side: slot `pos` of row r holds chunk pos ^ ((r >> 1) & 7); fragment reads use the same XOR;
//   * two LDS stages; the DMA of step s+1 is in flight while step s is on the matrix cores;
//   * operand roles are swapped (A = weights, B = pixels) so a lane owns 4 consecutive output channels of one
//     pixel: 8-byte stores, and with the pooled row order (2x2 window = 4 consecutive rows = 4 adjacent
//     lanes) AvgPool2d is two cross-lane adds.
// Input-side stages (style modulation, leaky-ReLU before the convolution) cannot ride on LDS-DMA; callers
// apply them in a pass of their own for these layers (the tensors are small: <= 134 MB at batch 256), or the
// register-staged kernel in conv_fwd.hip takes the call.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned WIDE_INV = 0x80000000u;

struct WideArgs {
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
    int N, H, W, Cin, Hl, Wl, up2, Ho, Wo, Cout, kw, py, px, Kpad, ksteps, pool;
    int M;
    unsigned x_bytes, w_bytes, y_bytes;
    float* ws;        // split-K: fp32 partial tiles [ksplit][M][Cout], folded (with every output stage) by wide_splitk_fold_k
    int ksplit, spk;  // K splits (blockIdx.z) and steps per split; ksplit == 1: the kernel finishes the tile itself
};

// buffer offsets of this lane's four pixel rows for tap (dy, dx): recomputed once per tap, not per K step
__device__ __forceinline__ void wide_set_tap(unsigned (&va)[4], const int (&a_n)[4], const int (&a_yx)[4], int dy, int dx, int csrc,
                                             const WideArgs& p) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int iy = (a_yx[j] >> 16) + dy - p.py;
        const int ix = (a_yx[j] & 0xffff) + dx - p.px;
        const bool ok = a_n[j] >= 0 && (unsigned)iy < (unsigned)p.Hl && (unsigned)ix < (unsigned)p.Wl;
        const unsigned off = (unsigned)(((a_n[j] * p.H + (iy >> p.up2)) * p.W + (ix >> p.up2)) * p.Cin + csrc * 8) * 2u;
        va[j] = ok ? off : WIDE_INV;
    }
}

// the K step the next wide_issue() stages, and this lane's row offsets for its tap
struct WideCursor {
    unsigned va[4];
    int tap, cb, dy, dx;
};

// LDS-DMA of one K step (64 channels of one tap) into stage `buf`, then advance the cursor
template <int BN, int NST>
__device__ __forceinline__ void wide_issue(WideCursor& cur, char* lds, int buf, int wave, __amdgpu_buffer_rsrc_t rx,
                                           __amdgpu_buffer_rsrc_t rw, const unsigned (&b_off)[BN / 32], const int (&a_n)[4],
                                           const int (&a_yx)[4], int csrc, const WideArgs& p) {
    constexpr int A_BYTES = 128 * 128, STAGE = A_BYTES + BN * 128;
    char* const A = lds + buf * STAGE;
    char* const B = A + A_BYTES;
    const int soa = cur.cb * 2, sob = (cur.tap * p.Cin + cur.cb) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(A + (4 * j + wave) * 1024), 16, cur.va[j], soa,
                                                 0, 0);
#pragma unroll
    for (int j = 0; j < BN / 32; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(B + (4 * j + wave) * 1024), 16, b_off[j], sob, 0,
                                                 0);
    cur.cb += 64;
    if (cur.cb == p.Cin) {  // wave-uniform: next tap
        cur.cb = 0;
        ++cur.tap;
        if (++cur.dx == p.kw) {
            cur.dx = 0;
            ++cur.dy;
        }
        wide_set_tap(cur.va, a_n, a_yx, cur.dy, cur.dx, csrc, p);
    }
}

// NST = 2: two stages, two workgroups per CU -- the shape for grids that fill the chip, where the L2 -> LDS
// stream (not its latency) bounds a step.  NST = 4: small grids (small maps / per-GPU batches: fewer tiles than
// CUs) are bound by the DMA round trip of every step instead (0.76 us per step at 512 channels on 8x8, batch
// 32); three steps in flight hide it.  One barrier per step either way.
template <int BN, int NST>
__global__ void __launch_bounds__(256, NST == 2 ? 2 : 1) conv_wide_k(const WideArgs p) {
    constexpr int BM = 128, FP = 4, FC = BN / 32;          // wave tile: 64 pixels x BN/2 channels
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;  // one stage: pixel tile, weight tile (64 k each)
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int PER = 4 + BN / 32;                       // LDS-DMA instructions per wave and step
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    char* const lds = lds_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
    const int HoWo = p.Ho * p.Wo;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, p.w_bytes, 0x00020000);

    // ---- staging bookkeeping: wave instruction j covers rows (4j + wave)*8 .. +7, 8 lanes per row ----------
    const int rsub = lane >> 3;
    // source chunk of this lane's LDS slot: pos ^ ((row >> 1) & 7).  A fragment read touches 16 consecutive rows
    // per 16 lanes; two rows span all 64 banks, so the key must be distinct for the 8 row PAIRS (keyed on
    // row & 7 the rows r and r+8 collided: 2-way conflicts on every ds_read_b128 of a kernel that moves
    // 128 B/clk through LDS at full MFMA rate).  Row = (4j + wave)*8 + rsub  =>  (row >> 1) & 7 = (wave&1)*4 + rsub/2.
    const int csrc = (lane & 7) ^ (((wave & 1) << 2) | (rsub >> 1));
    int a_n[4], a_yx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = bm0 + (4 * j + wave) * 8 + rsub;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        int oy, ox;
        if (p.pool) {  // quad-major rows: 4q..4q+3 are the 2x2 window of pooled pixel q
            const int q = rem >> 2, wo2 = p.Wo >> 1;
            const int qy = q / wo2;
            oy = 2 * qy + ((rem >> 1) & 1);
            ox = 2 * (q - qy * wo2) + (rem & 1);
        } else {
            oy = rem / p.Wo;
            ox = rem - oy * p.Wo;
        }
        a_n[j] = ok ? n : -1;
        a_yx[j] = (oy << 16) | ox;
    }
    unsigned b_off[BN / 32];
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) {
        const int co = bn0 + (4 * j + wave) * 8 + rsub;
        b_off[j] = co < p.Cout ? (unsigned)(co * p.Kpad + csrc * 8) * 2u : WIDE_INV;
    }

    // Per-lane source offsets change only when the K loop moves to the next tap; the channel block inside a tap
    // (and the k offset of the weight rows) is wave-uniform and rides in the instruction's scalar offset, so a
    // K step costs no address VALU at all.  (PMC before: 144 VALU instructions per step against 32 MFMAs.)
    // this workgroup's K range (split-K over blockIdx.z: small grids, where a tile's serial chain of steps is the time)
    const int s_begin = blockIdx.z * p.spk;
    const int nsteps = (p.ksteps - s_begin) < p.spk ? (p.ksteps - s_begin) : p.spk;
    WideCursor cur;
    {
        const int k0 = s_begin * 64;
        cur.tap = k0 / p.Cin;
        cur.cb = k0 - cur.tap * p.Cin;
        cur.dy = cur.tap / p.kw;
        cur.dx = cur.tap - cur.dy * p.kw;
    }
    wide_set_tap(cur.va, a_n, a_yx, cur.dy, cur.dx, csrc, p);

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int f = 0; f < FP; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // byte offset of this lane's fragment chunk (k half h) inside a 128-byte row r: ((4h + fq) ^ ((r >> 1) & 7)) * 16;
    // (r >> 1) & 7 == frow >> 1 for every fragment row (tile offsets are multiples of 16)
    const int sw0 = ((fq) ^ (frow >> 1)) << 4, sw1 = ((4 + fq) ^ (frow >> 1)) << 4;
    const int xrow = (wm * 64 + frow) * 128, wrow = (wn * (BN / 2) + frow) * 128;

#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < nsteps) wide_issue<BN, NST>(cur, lds, i, wave, rx, rw, b_off, a_n, a_yx, csrc, p);

    for (int s = 0; s < nsteps; ++s) {
        // this wave's share of step s has landed: all but the (NST - 2) younger steps' DMAs are done (in the
        // tail fewer are in flight, so wait for everything)
        if (NST > 2 && s + NST - 2 < nsteps) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // ... everyone's share, and everyone is done reading the stage of step s - 1.  A bare s_barrier:
        // __syncthreads() carries a fence that drains vmcnt to 0, i.e. the younger steps' DMAs with it.
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int buf = s % NST;
        if (s + NST - 1 < nsteps) wide_issue<BN, NST>(cur, lds, (s + NST - 1) % NST, wave, rx, rw, b_off, a_n, a_yx, csrc, p);
        const char* const A = lds + buf * STAGE;
        const char* const B = A + A_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sw = h ? sw1 : sw0;
            bf16x8 wf[FC], xf[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(B + wrow + i * 16 * 128 + sw);
#pragma unroll
            for (int f = 0; f < FP; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(A + xrow + f * 16 * 128 + sw);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int f = 0; f < FP; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[f], acc[i][f], 0, 0, 0);
        }
    }

    // ---- split-K: the raw fp32 partial tile goes to the workspace; wide_splitk_fold_k applies the output stages
    if (p.ksplit > 1) {
        float* __restrict__ part = p.ws + (int64_t)blockIdx.z * p.M * p.Cout;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            const int m = bm0 + wm * 64 + f * 16 + frow;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int co = bn0 + wn * (BN / 2) + i * 16 + fq * 4;
                if (m < p.M && co < p.Cout) *reinterpret_cast<f32x4*>(part + (int64_t)m * p.Cout + co) = acc[i][f];
            }
        }
        return;
    }
    // ---- epilogue: lane holds channels co0..co0+3 (co0 = bn0 + wn*BN/2 + 16*i + 4*fq) of pixel row f*16 + frow
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.aux), 0, p.aux ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0, p.resid ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rnz =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.noise), 0, p.noise ? p.y_bytes / (unsigned)p.Cout : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rosc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.out_scale), 0, p.out_scale ? (unsigned)(p.N * p.Cout) * 4u : 0u, 0x00020000);
    const int out_per_img = p.pool ? HoWo >> 2 : HoWo;
    const float gain = p.pool ? 0.25f * p.gain : p.gain;
    float4 bia[FC], nsc[FC];
    int co0[FC];
#pragma unroll
    for (int i = 0; i < FC; ++i) {
        co0[i] = bn0 + wn * (BN / 2) + i * 16 + fq * 4;
        const bool cok = co0[i] < p.Cout;  // Cout % 4 == 0
        bia[i] = (p.bias && cok) ? *reinterpret_cast<const float4*>(p.bias + co0[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        nsc[i] = (p.noise_scale && cok) ? *reinterpret_cast<const float4*>(p.noise_scale + co0[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the output stages in three compiled variants, picked once per workgroup (as conv_wide_halo_k's): the plain layer
    // issues no loads for the tensors it does not have and one fma per element
    auto epilogue = [&](auto modc, auto arc) {
    constexpr bool MOD = decltype(modc)::value, AR = decltype(arc)::value;
#pragma unroll
    for (int f = 0; f < FP; ++f) {
        const int m = bm0 + wm * 64 + f * 16 + frow;
        const bool live = m < p.M && (!p.pool || (frow & 3) == 0);
        const int opix = p.pool ? m >> 2 : m;
        const int n = opix / out_per_img;
        const unsigned obase = live ? (unsigned)opix * (unsigned)(p.Cout * 2) : WIDE_INV;
        // every per-pixel load of this fragment row first, then its stores
        float nz = 0.f;
        if (MOD && p.noise) nz = (float)__builtin_bit_cast(bf16, __builtin_amdgcn_raw_buffer_load_b16(rnz, live ? (unsigned)opix * 2u : WIDE_INV, 0, 0));
        v2u ax[FC], rs[FC];
        v4u os[FC];
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const bool cok = live && co0[i] < p.Cout;
            ax[i] = rs[i] = v2u{0u, 0u};
            os[i] = v4u{0u, 0u, 0u, 0u};
            if (AR) {  // (absent tensors read zeros past their zero-sized resource)
                ax[i] = __builtin_amdgcn_raw_buffer_load_b64(raux, cok ? obase + co0[i] * 2 : WIDE_INV, 0, 0);
                rs[i] = __builtin_amdgcn_raw_buffer_load_b64(rres, cok ? obase + co0[i] * 2 : WIDE_INV, 0, 0);
            }
            if (MOD) os[i] = __builtin_amdgcn_raw_buffer_load_b128(rosc, cok ? (unsigned)(n * p.Cout + co0[i]) * 4u : WIDE_INV, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = acc[i][f][r];
                if (p.pool) {  // executed by every lane: the window's four pixels are four adjacent lanes
                    a += quad_xor1(a);
                    a += quad_xor2(a);
                }
                v[r] = a;
            }
            if (MOD) {
                const float4 d4 = p.out_scale ? __builtin_bit_cast(float4, os[i]) : make_float4(1.f, 1.f, 1.f, 1.f);
                v[0] = fmaf(nz, nsc[i].x, fmaf(v[0] * gain, d4.x, bia[i].x));
                v[1] = fmaf(nz, nsc[i].y, fmaf(v[1] * gain, d4.y, bia[i].y));
                v[2] = fmaf(nz, nsc[i].z, fmaf(v[2] * gain, d4.z, bia[i].z));
                v[3] = fmaf(nz, nsc[i].w, fmaf(v[3] * gain, d4.w, bia[i].w));
            } else {
                v[0] = fmaf(v[0], gain, bia[i].x);
                v[1] = fmaf(v[1], gain, bia[i].y);
                v[2] = fmaf(v[2], gain, bia[i].z);
                v[3] = fmaf(v[3], gain, bia[i].w);
            }
            bf16x4 o;
            const bf16x4 a4 = __builtin_bit_cast(bf16x4, ax[i]), r4 = __builtin_bit_cast(bf16x4, rs[i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = lrelu_f(v[r], p.slope);
                if (AR) {
                    if (p.aux) u *= lrelu_mask((float)a4[r], p.aux_slope);
                    u += (float)r4[r];  // zeros without a residual (out-of-range load)
                }
                o[r] = (bf16)u;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), ry, (live && co0[i] < p.Cout) ? obase + co0[i] * 2 : WIDE_INV, 0, 0);
        }
    }
    };
    if (p.out_scale || p.noise) epilogue(std::true_type{}, std::true_type{});
    else if (p.aux || p.resid) epilogue(std::false_type{}, std::true_type{});
    else epilogue(std::false_type{}, std::false_type{});
}

// Split-K fold: y[opix][co..co+3] = stages( sum_z ws[z][m][co..co+3] ), the output stages of conv_wide_k's epilogue on the
// folded tile.  One thread per (output pixel, 4 channels); with pooling the pixel's four GEMM rows 4q .. 4q+3 are summed too.
__global__ void __launch_bounds__(256) wide_splitk_fold_k(const WideArgs p) {
    const int cq = p.Cout >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t npix = p.pool ? p.M >> 2 : p.M;
    if (idx >= npix * cq) return;
    const int opix = (int)(idx / cq), co = (int)(idx - (int64_t)opix * cq) * 4;
    const int rows = p.pool ? 4 : 1;
    const int64_t m0 = p.pool ? (int64_t)opix * 4 : opix;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.ksplit; ++z)
        for (int r = 0; r < rows; ++r) a += *reinterpret_cast<const f32x4*>(p.ws + ((int64_t)z * p.M + m0 + r) * p.Cout + co);
    const float gain = p.pool ? 0.25f * p.gain : p.gain;
    const int out_per_img = p.pool ? (p.Ho * p.Wo) >> 2 : p.Ho * p.Wo;
    const int n = opix / out_per_img;
    float4 d4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = b4;
    if (p.out_scale) d4 = *reinterpret_cast<const float4*>(p.out_scale + (int64_t)n * p.Cout + co);
    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + co);
    float nz = 0.f;
    if (p.noise) {
        nz = (float)reinterpret_cast<const bf16*>(p.noise)[opix];
        s4 = *reinterpret_cast<const float4*>(p.noise_scale + co);
    }
    float v[4] = {fmaf(nz, s4.x, fmaf(a[0] * gain, d4.x, b4.x)), fmaf(nz, s4.y, fmaf(a[1] * gain, d4.y, b4.y)),
                  fmaf(nz, s4.z, fmaf(a[2] * gain, d4.z, b4.z)), fmaf(nz, s4.w, fmaf(a[3] * gain, d4.w, b4.w))};
    const int64_t o = (int64_t)opix * p.Cout + co;
    bf16x4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float u = lrelu_f(v[r], p.slope);
        if (p.aux) u *= lrelu_mask((float)reinterpret_cast<const bf16*>(p.aux)[o + r], p.aux_slope);
        if (p.resid) u += (float)reinterpret_cast<const bf16*>(p.resid)[o + r];
        out[r] = (bf16)u;
    }
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.y) + o) = out;
}

// K splits for a grid of `wgs` 128 x 64 tiles with `ksteps` steps each: where the chip is mostly idle (fewer tiles than
// half its CUs) and the chain is long, never more splits than leave 8 steps each; two for grids up to the CU count on
// the 512-channel layers (64 x 512 x 8 x 8: 47 -> 44 us host-launched, batch 64: 20.5 -> 19.8 ms per iteration)
int wide_ksplit(int64_t wgs, int ksteps) {
    if (wgs > 256 || ksteps < 16) return 1;
    if (wgs > 128) return ksteps >= 48 ? 2 : 1;  // two-stage variant, two workgroups per CU (see the launch); 36 steps: the fold costs more
    int s = (int)(256 / wgs);
    if (s > 8) s = 8;
    if (s > ksteps / 8) s = ksteps / 8;
    return s < 2 ? 1 : s;
}

}  // namespace

// floats of workspace with which gan_conv2d_fwd_wide splits K for this shape (0: it does not)
int64_t gan_conv2d_fwd_wide_workspace(int N, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int Kpad) {
    if (Cin % 64 != 0 || Kpad != kh * kw * Cin || (Cout & 3)) return 0;
    const int64_t M = (int64_t)N * Ho * Wo;
    if (M < 128 || M >= (1LL << 31) - 256) return 0;
    const int64_t gm = (M + 127) / 128, gn = (Cout + 127) / 128, gn64 = (Cout + 63) / 64;
    static const int thr = getenv("GAN_AMD_WIDE_THR") ? atoi(getenv("GAN_AMD_WIDE_THR")) : 256;
    if ((Cout > 64 && gm * gn >= thr) || gm * gn64 >= 2 * thr) return 0;
    const int S = wide_ksplit(gm * gn64, Kpad / 64);
    return S > 1 ? (int64_t)S * M * Cout : 0;
}

// 1 = served, 0 = shape not eligible, < 0 = error.  Never takes in_scale / in_slope (see the header comment).
int gan_conv2d_fwd_wide(const void* x, const void* wt, void* y, const float* out_scale, const float* bias,
                        const void* noise, const float* noise_scale, float slope, int N, int H, int W, int Cin, int up2,
                        int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad, const ConvFuse& fu,
                        hipStream_t st) {
    if (Cin % 64 != 0 || Kpad != kh * kw * Cin || (Cout & 3) || fu.in_slope != 1.f) return 0;
    // 3x3 layers on grids that fill the chip: the resident-halo kernel (conv_wide_halo.hip)
    const int halo = gan_conv2d_fwd_wide_halo(x, wt, y, out_scale, bias, noise, noise_scale, slope, N, H, W, Cin, up2, Ho, Wo, Cout, kh, kw, py,
                                              px, Kpad, fu, st);
    if (halo != 0) return halo;
    const int64_t M = (int64_t)N * Ho * Wo;
    const int64_t xb = (int64_t)N * H * W * Cin * 2, wb = (int64_t)Cout * Kpad * 2;
    const int64_t yb = (fu.pool ? M / 4 : M) * Cout * 2;
    if (M < 128 || M >= (1LL << 31) - 256 || xb >= (1LL << 31) || wb >= (1LL << 31) || yb >= (1LL << 31)) return 0;
    if (((uintptr_t)y & 7) || ((uintptr_t)fu.aux & 7) || ((uintptr_t)fu.resid & 7) || ((uintptr_t)out_scale & 15) || ((uintptr_t)bias & 15) ||
        ((uintptr_t)noise_scale & 15))
        return 0;
    if (H >= 32768 || W >= 32768) return 0;
    WideArgs a;
    a.x = x; a.wt = wt; a.y = y; a.out_scale = out_scale; a.bias = bias; a.noise = noise; a.noise_scale = noise_scale;
    a.aux = fu.aux; a.resid = fu.resid; a.slope = slope; a.aux_slope = fu.aux_slope; a.gain = fu.gain;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Hl = up2 ? 2 * H : H; a.Wl = up2 ? 2 * W : W; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.kw = kw; a.py = py; a.px = px; a.Kpad = Kpad; a.ksteps = Kpad / 64;
    a.pool = fu.pool; a.M = (int)M;
    a.ws = nullptr; a.ksplit = 1; a.spk = a.ksteps;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
    const unsigned gm = (unsigned)((M + 127) / 128);
    const unsigned gn = (unsigned)((Cout + 127) / 128), gn64 = (unsigned)((Cout + 63) / 64);
    if (gn64 > 65535) return 0;
    static const int thr = getenv("GAN_AMD_WIDE_THR") ? atoi(getenv("GAN_AMD_WIDE_THR")) : 256;
    // 128 x 128 tiles when they fill the chip (two workgroups per CU); small maps / batches get 128 x 64 tiles
    // and a four-stage ring (conv_wide_k's header)
    if (Cout > 64 && (int64_t)gm * gn >= thr) {
        hipLaunchKernelGGL((conv_wide_k<128, 2>), dim3(gm, gn), dim3(256), 2 * (128 + 128) * 128, st, a);
    } else if ((int64_t)gm * gn64 >= 2 * thr) {
        hipLaunchKernelGGL((conv_wide_k<64, 2>), dim3(gm, gn64), dim3(256), 2 * (128 + 64) * 128, st, a);
    } else {
        // small grid: with a workspace, split K so the chip fills and a tile's chain of steps shortens (72 steps of a
        // 512-channel 3x3 layer are ~42 us whatever the batch), then fold the partial tiles with the output stages
        const int S = wide_ksplit((int64_t)gm * gn64, a.ksteps);
        if (S > 1 && fu.ws && fu.ws_floats >= (int64_t)S * M * Cout && (((uintptr_t)fu.ws) & 15) == 0) {
            a.ws = fu.ws; a.ksplit = S; a.spk = (a.ksteps + S - 1) / S;
            while ((a.ksplit - 1) * a.spk >= a.ksteps) --a.ksplit;  // no empty split
        }
        if (a.ksplit > 1 && (int64_t)gm * gn64 > 128)  // 2 x (129..256) workgroups: they only run at once with the 48 KB two-stage ring
            hipLaunchKernelGGL((conv_wide_k<64, 2>), dim3(gm, gn64, (unsigned)a.ksplit), dim3(256), 2 * (128 + 64) * 128, st, a);
        else
            hipLaunchKernelGGL((conv_wide_k<64, 4>), dim3(gm, gn64, (unsigned)a.ksplit), dim3(256), 4 * (128 + 64) * 128, st, a);
        if (a.ksplit > 1) {
            const int64_t quads = (fu.pool ? M / 4 : M) * (Cout / 4);
            hipLaunchKernelGGL(wide_splitk_fold_k, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gan_set_error("conv2d_fwd(wide): launch failed: %s", hipGetErrorString(e));
        return GAN_ERR_LAUNCH;
    }
    return 1;
}
