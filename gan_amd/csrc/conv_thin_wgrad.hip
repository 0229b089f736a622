// Weight gradient for THIN layers on MFMA (gfx950): Cin, Cout <= 64 at large spatial size, bf16.
//
//   dw[tap][co][ci] = sum_{n,oy,ox} gy[n,oy,ox,co] * X[n, oy+dy-py, ox+dx-px, ci] * in_scale[n][ci]
//
// Like the forward thin kernel (conv_thin.hip) this shape is HBM-bound (gy and x are each read once
// from HBM in the ideal), so: a persistent workgroup walks a contiguous range of TH x 32 pixel
// tiles; per tile the gy tile and the (TH+2) x 34 halo of x go to LDS ONCE with 16-byte loads
// (nearest-2x upsampling and style modulation applied on the way) and all kh*kw taps are formed from
// that one LDS image; accumulators for every tap live in registers across ALL tiles of the
// workgroup and are flushed with one fp32 atomic per element at the end.
//
// MFMA view (16x16x32 bf16): rows = co, cols = ci, k = 32 pixels = one tile row.  Both operands
// are pixel-major in memory, i.e. transposed for MFMA fragments; they are stored in LDS as
// [16-channel group][pixel][16 ch] (32-byte rows) and read with ds_read_b64_tr_b16, k <-> pixel
// map  k = 8g + j  <->  x = (j < 4 ? 4g + j : 16 + 4g + j - 4)  so that one instruction touches 16
// consecutive 32-byte rows (conflict-free).  Waves split the work as WR row groups x WT tap groups
// so that the per-wave accumulator set (taps x Cout/16 x Cin/16 x 4 registers) fits.
#include "common.h"

namespace {

struct ThinWArgs {
    const void* x;
    const void* gy;
    float* dw;
    const float* in_scale;
    int N, H, W, Hl, Wl, up2, Ho, Wo, Cin, Cout, py, px;
    int tiles_x, tiles_y;
    float* ws;  // [gridDim.x][taps][Cout][Cin] partial slices (plain stores), folded by finalize_dw_k
    unsigned x_bytes, g_bytes;  // < 2^31: 32-bit buffer offsets
    WgradFuse fu;
    DwOut dwo;  // host side only: where the slice reducer writes
    int want_bias;  // also sum gy over the pixels: Cout floats behind every slice (slice stride taps*Cout*Cin + Cout)
};

__device__ __forceinline__ uint4 wmod_chunk(uint4 raw, const float* s) {
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

// 8 pixels (k = 8g .. 8g+7 under the map above) of channel (lane & 15) of a [pixel][16 ch] LDS image;
// `p` already points at this lane's 8-byte piece of row (4g + q) of the 32-pixel run.
__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 16 * 32));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int CIN, int COUT, int KS, int TH, int WT>
struct ThinWCfg {
    static constexpr int CGI = CIN / 16, CGO = COUT / 16;
    static constexpr int HH = TH + KS - 1, HWD = 32 + KS - 1, NPX = HH * HWD, NPG = TH * 32;
    static constexpr int PLX = NPX * 32, PLG = NPG * 32;  // bytes per 16-channel plane
    static constexpr int WR = 4 / WT, ROWS = TH / WR, NT = (KS * KS + WT - 1) / WT;
    static constexpr int NCX = ((NPX + 7) / 8) * 8 * (CIN / 8), NCG = NPG * (COUT / 8);
    static constexpr int X_BYTES = CGI * PLX, G_BYTES = CGO * PLG, LDS_BYTES = X_BYTES + G_BYTES;
    // waves per SIMD the register allocation must leave room for: four workgroups per CU fit in LDS on the smallest
    // tiles, and the 16 -> 32 3x3 variant sits at 128 registers (the bias sums pushed it to 135: 236 -> 343 us)
    static constexpr int MIN_WAVES = (LDS_BYTES + 4096) * 4 <= 160 * 1024 && NT * CGO * CGI * 4 <= 72 ? 4 : 2;
};

template <int CIN, int COUT, int KS, int TH, int WT>
__global__ void __launch_bounds__(256, (ThinWCfg<CIN, COUT, KS, TH, WT>::MIN_WAVES)) conv_thin_wgrad_k(const ThinWArgs p) {
    typedef ThinWCfg<CIN, COUT, KS, TH, WT> C;
    constexpr int CGI = C::CGI, CGO = C::CGO, HWD = C::HWD, NPX = C::NPX, PLX = C::PLX, PLG = C::PLG;
    constexpr int ROWS = C::ROWS, NT = C::NT, CPX = CIN / 8, CPG = COUT / 8;

    __shared__ __attribute__((aligned(256))) char lds[C::LDS_BYTES + 256 * 16];
    char* const xl = lds;
    char* const gl = lds + C::X_BYTES;
    char* const dumpl = lds + C::LDS_BYTES;  // per-thread 16-byte slot for staging chunks past the tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wt = wave % WT, wr = wave / WT;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.N;
    const int t0 = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t1 = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gy), 0, p.g_bytes, 0x00020000);

    f32x4 acc[NT][CGO][CGI];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int o = 0; o < CGO; ++o)
#pragma unroll
            for (int i = 0; i < CGI; ++i) acc[j][o][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int lane_off = (4 * g + q) * 32 + pp * 8;  // this lane's piece inside a 32-pixel run
    // bias gradient: sum_pixels gy[co] = gy^T x ones, one more MFMA per tile row on the waves of tap group 0 (every
    // column of the result tile holds the sum); no VALU, no second pass over gy
    f32x4 accb[CGO];
#pragma unroll
    for (int o = 0; o < CGO; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    const bool sum_gy = p.want_bias && wt == 0;

    for (int t = t0; t < t1; ++t) {
        const int n = t / tiles_per_img;
        const int rem = t - n * tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * 32;
        __syncthreads();  // previous tile's fragment reads are done
        // Staging is branch-free: batches of UB raw buffer loads (lanes past the image get an out-of-range
        // offset and read zeros), then the LDS stores; chunk indices past the tile go to a per-thread dump slot.
        constexpr int UB = 4;
        constexpr int NLX = (C::NCX + 255) / 256, NLG = (C::NCG + 255) / 256;
        // ---- x halo -> LDS [ci/16][pixel][16]; 256 % (8*CPX) == 0: a thread's channel chunk is fixed ----
        {
            const int c8 = (tid % (8 * CPX)) >> 3;
            const bool cok = c8 * 8 < p.Cin;
            float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0;
            if (p.in_scale) {
                const float* sp = p.in_scale + (int64_t)n * p.Cin + (cok ? c8 * 8 : 0);
                s0 = *reinterpret_cast<const float4*>(sp);
                s1 = *reinterpret_cast<const float4*>(sp + 4);
            }
#pragma unroll
            for (int b0 = 0; b0 < NLX; b0 += UB) {
                v4u v[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    if (b0 + k >= NLX) continue;
                    const int idx = (b0 + k) * 256 + tid;
                    const int pix = (idx / (8 * CPX)) * 8 + (idx & 7);
                    const int hy = pix / HWD, hx = pix - hy * HWD;
                    const int ly = oy0 - p.py + hy, lx = ox0 - p.px + hx;
                    const bool ok = pix < NPX && (unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl && cok;
                    const unsigned off = (unsigned)(((n * p.H + (ly >> p.up2)) * p.W + (lx >> p.up2)) * (p.Cin * 2) + c8 * 16);
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : 0x80000000u, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    if (b0 + k >= NLX) continue;
                    const int idx = (b0 + k) * 256 + tid;
                    const int pix = (idx / (8 * CPX)) * 8 + (idx & 7);
                    uint4 u = __builtin_bit_cast(uint4, v[k]);
                    if (p.fu.x_slope != 1.f) u = lrelu_chunk(u, p.fu.x_slope, bf16());
                    if (p.in_scale) {
                        bf16x8 h = *reinterpret_cast<bf16x8*>(&u);
                        h[0] = (bf16)((float)h[0] * s0.x); h[1] = (bf16)((float)h[1] * s0.y);
                        h[2] = (bf16)((float)h[2] * s0.z); h[3] = (bf16)((float)h[3] * s0.w);
                        h[4] = (bf16)((float)h[4] * s1.x); h[5] = (bf16)((float)h[5] * s1.y);
                        h[6] = (bf16)((float)h[6] * s1.z); h[7] = (bf16)((float)h[7] * s1.w);
                        u = *reinterpret_cast<uint4*>(&h);
                    }
                    char* const dst = pix < NPX ? xl + (c8 >> 1) * PLX + pix * 32 + (c8 & 1) * 16 : dumpl + tid * 16;
                    *reinterpret_cast<uint4*>(dst) = u;
                }
            }
        }
        // ---- gy tile -> LDS [co/16][pixel][16] (gy_up2: gy is stored at half resolution) --------
        {
            const int c8 = (tid % (8 * CPG)) >> 3;
            const bool cok = c8 * 8 < p.Cout;
            const int gsh = p.fu.gy_up2;
            const int Hg = p.Ho >> gsh, Wg = p.Wo >> gsh;
#pragma unroll
            for (int b0 = 0; b0 < NLG; b0 += UB) {
                v4u v[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    if (b0 + k >= NLG) continue;
                    const int idx = (b0 + k) * 256 + tid;
                    const int pix = (idx / (8 * CPG)) * 8 + (idx & 7);
                    const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
                    const bool ok = idx < C::NCG && oy < p.Ho && ox < p.Wo && cok;
                    const unsigned off = (unsigned)(((n * Hg + (oy >> gsh)) * Wg + (ox >> gsh)) * (p.Cout * 2) + c8 * 16);
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? off : 0x80000000u, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    if (b0 + k >= NLG) continue;
                    const int idx = (b0 + k) * 256 + tid;
                    const int pix = (idx / (8 * CPG)) * 8 + (idx & 7);
                    char* const dst = idx < C::NCG ? gl + (c8 >> 1) * PLG + pix * 32 + (c8 & 1) * 16 : dumpl + tid * 16;
                    *reinterpret_cast<uint4*>(dst) = __builtin_bit_cast(uint4, v[k]);
                }
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int rr = 0; rr < ROWS; ++rr) {
            const int r = wr * ROWS + rr;
            bf16x8 a[CGO];
#pragma unroll
            for (int o = 0; o < CGO; ++o) a[o] = tr_frag(gl + o * PLG + r * 32 * 32 + lane_off);
            if (sum_gy) {
#pragma unroll
                for (int o = 0; o < CGO; ++o) accb[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[o], ones, accb[o], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tap = wt + j * WT;
                if (tap < KS * KS) {
                    const int dy = tap / KS, dx = tap - dy * KS;
                    const char* const xb = xl + ((r + dy) * HWD + dx) * 32 + lane_off;
#pragma unroll
                    for (int i = 0; i < CGI; ++i) {
                        const bf16x8 b = tr_frag(xb + i * PLX);
#pragma unroll
                        for (int o = 0; o < CGO; ++o)
                            acc[j][o][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[o], b, acc[j][o][i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- flush: D layout col = lane & 15 -> ci, row = 4*(lane >> 4) + r -> co ----------------------
    // The WR row groups of the workgroup hold partial sums of the same taps: they are folded through LDS
    // first (wave wr*WT + wt into wave wt), so a workgroup writes ONE slice of dw with plain stores and the
    // slice reducer reads WR times fewer of them (it was the larger share of this kernel's traffic on the
    // 32..64-channel layers).  Gain: in the slice reducer.
    if (t0 < t1) {
        if (C::WR > 1) {
            static_assert(NT * CGO * CGI * 1024 <= C::LDS_BYTES + 256 * 16, "partial sums of one wave do not fit the tile's LDS");
            f32x4* const red = reinterpret_cast<f32x4*>(lds);
            for (int src = WT; src < 4; ++src) {
                __syncthreads();  // the tile images / the previous round's partials are no longer read
                if (wave == src) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int o = 0; o < CGO; ++o)
#pragma unroll
                            for (int i = 0; i < CGI; ++i) red[((j * CGO + o) * CGI + i) * 64 + lane] = acc[j][o][i];
                }
                __syncthreads();
                if (wave == src % WT) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int o = 0; o < CGO; ++o)
#pragma unroll
                            for (int i = 0; i < CGI; ++i) {
                                const f32x4 v = red[((j * CGO + o) * CGI + i) * 64 + lane];
                                acc[j][o][i][0] += v[0]; acc[j][o][i][1] += v[1]; acc[j][o][i][2] += v[2]; acc[j][o][i][3] += v[3];
                            }
                }
            }
        }
        const int64_t L = (int64_t)KS * KS * p.Cout * p.Cin;
        const int64_t Lp = L + (p.want_bias ? p.Cout : 0);
        if (p.want_bias) {
            __syncthreads();  // the fold's LDS reads are done
            float* const bs = reinterpret_cast<float*>(lds);  // [WR][COUT] row-group partials
            if (wt == 0 && li == 0) {
#pragma unroll
                for (int o = 0; o < CGO; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bs[wr * COUT + o * 16 + g * 4 + r] = accb[o][r];
            }
            __syncthreads();
            if (tid < p.Cout) {
                float sum = 0.f;
                for (int w = 0; w < C::WR; ++w) sum += bs[w * COUT + tid];
                p.ws[(int64_t)blockIdx.x * Lp + L + tid] = sum;
            }
        }
        if (wr == 0) {
            float* __restrict__ base = p.ws + (int64_t)blockIdx.x * Lp;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int tap = wt + j * WT;
                if (tap >= KS * KS) continue;
                float* __restrict__ dwt = base + (int64_t)tap * p.Cout * p.Cin;
#pragma unroll
                for (int o = 0; o < CGO; ++o)
#pragma unroll
                    for (int i = 0; i < CGI; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = o * 16 + g * 4 + r;
                            const int ci = i * 16 + li;
                            if (co < p.Cout && ci < p.Cin) dwt[(int64_t)co * p.Cin + ci] = acc[j][o][i][r];
                        }
            }
        }
    }
}

// `need` != nullptr: only report the workspace (floats) this configuration would use.
template <int CIN, int COUT, int KS, int TH, int WT>
int launch_thin_w(ThinWArgs a, int64_t ws_floats, int64_t* need, hipStream_t st) {
    typedef ThinWCfg<CIN, COUT, KS, TH, WT> C;
    static_assert(C::LDS_BYTES + 4096 <= 160 * 1024, "tile does not fit LDS");
    static_assert(TH % C::WR == 0, "rows must split evenly over the row groups");
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + TH - 1) / TH;
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.N;
    int per_cu = (160 * 1024) / (C::LDS_BYTES + 4096);
    // every workgroup writes WR slices of dw that the reducer reads back: with few tiles per workgroup (small
    // per-GPU batches) that traffic outweighs the tail effect the extra workgroups smooth (batch 32: 98 -> 74 us)
    const int cap = ntiles <= 16384 ? 2 : 4;
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    int64_t grid = 256LL * per_cu;
    if (grid > ntiles) grid = ntiles;
    const int64_t L = (int64_t)KS * KS * a.Cout * a.Cin;
    const int64_t want = grid * (L + a.Cout);  // (room for the channel sums whether or not this call wants them)
    if (need) {
        *need = want;
        return GAN_OK;
    }
    if (a.ws == nullptr || ws_floats < want) return 1;  // not served without its slices (GAN_OK is 0; the caller maps 1 -> 0)
    hipLaunchKernelGGL((conv_thin_wgrad_k<CIN, COUT, KS, TH, WT>), dim3((unsigned)grid), dim3(256), 0, st, a);
    GAN_CHECK_LAUNCH("conv2d_wgrad(thin)");
    // (gy at half resolution is staged four times per source pixel: the sums come out 4x)
    return gan_finalize_dw(a.ws, (int)grid, KS * KS, a.Cout, a.Cin, a.dwo, a.fu.gain, st, a.want_bias, a.fu.gy_up2 ? 0.25f : 1.f);
}

template <int CIN, int COUT>
int pick_ks(const ThinWArgs& a, int ks, int64_t wsf, int64_t* need, hipStream_t st) {
    constexpr int F = (CIN / 16) * (COUT / 16);
    constexpr int TH = (CIN <= 16 && COUT <= 16) ? 16 : 8;
    if (ks == 1) return launch_thin_w<CIN, COUT, 1, TH, 1>(a, wsf, need, st);
    constexpr int WT = F <= 2 ? 1 : (F <= 8 ? 2 : 4);
    return launch_thin_w<CIN, COUT, 3, TH, WT>(a, wsf, need, st);
}

template <int CIN>
int pick_cout_w(const ThinWArgs& a, int ks, int64_t wsf, int64_t* need, hipStream_t st) {
    if (a.Cout <= 16) return pick_ks<CIN, 16>(a, ks, wsf, need, st);
    if (a.Cout <= 32) return pick_ks<CIN, 32>(a, ks, wsf, need, st);
    return pick_ks<CIN, 64>(a, ks, wsf, need, st);
}

}  // namespace

// 1 = served, 0 = shape not eligible or workspace too small, < 0 = error.
// need != nullptr: query only -- *need = workspace floats the launch would like (0 if not eligible).
int gan_conv2d_wgrad_thin(const void* x, const void* gy, const DwOut& dwo, const float* in_scale, int N, int H, int W,
                          int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, float* ws,
                          int64_t ws_floats, int64_t* need, const WgradFuse& fu, hipStream_t st) {
    if (need) *need = 0;
    if (kh != kw || (kh != 1 && kh != 3)) return 0;
    if (py != (kh - 1) / 2 || px != py) return 0;
    if (Cin > 64 || Cout > 64 || (Cin & 7) || (Cout & 7) || Wo < 32 || Ho < 8) return 0;
    if (Cin > 32 && Cout > 32 && kh == 3) return 0;  // a full 64 x 64 channel tile: the tiled kernel's shape (550 vs 390 TFLOP/s)
    const int64_t xb = (int64_t)N * H * W * Cin * 2, gb = (int64_t)N * (Ho >> fu.gy_up2) * (Wo >> fu.gy_up2) * Cout * 2;
    if (xb >= (1LL << 31) || gb >= (1LL << 31) || ((uintptr_t)in_scale & 15)) return 0;
    ThinWArgs a;
    a.x_bytes = (unsigned)xb; a.g_bytes = (unsigned)gb;
    a.fu = fu;
    a.x = x; a.gy = gy; a.dw = nullptr; a.dwo = dwo; a.in_scale = in_scale;
    a.want_bias = dwo.bias != nullptr;
    a.N = N; a.H = H; a.W = W; a.Hl = up2 ? 2 * H : H; a.Wl = up2 ? 2 * W : W; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cin = Cin; a.Cout = Cout; a.py = py; a.px = px;
    a.tiles_x = a.tiles_y = 0;
    a.ws = (ws && (((uintptr_t)ws) & 15) == 0) ? ws : nullptr;
    int rc;
    if (Cin <= 16) rc = pick_cout_w<16>(a, kh, ws_floats, need, st);
    else if (Cin <= 32) rc = pick_cout_w<32>(a, kh, ws_floats, need, st);
    else rc = pick_cout_w<64>(a, kh, ws_floats, need, st);
    return rc == GAN_OK ? 1 : (rc == 1 ? 0 : rc);
}
