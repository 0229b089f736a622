// upfirdn2d for bf16 channels-last tensors with the FIR evaluated on the matrix cores (gfx950).
//
// The op (modelling/nvidia_ops.py:18-34) is HBM-bound -- (numel_in + numel_out) * 2 bytes per call --
// but a straight VALU evaluation is not: 9 taps x 8 channels need ~16 VALU instructions per tap
// vector (bf16 -> fp32 unpack + fma), which caps a 3x3 FIR near 2.4 TB/s on MI355X (measured).  The
// matrix pipe has 16x the rate, and a 1-D FIR along x is a banded (Toeplitz) matrix product:
//
//   out[c][xo] = sum_i sum_xi  X_i[c][xi] * T_i[xi][xo],      T_i[xi][xo] = t[i][xi*up + px1 - xo*down]
//
// one v_mfma_f32_16x16x32_bf16 per (kernel row i, 16 channels, 16 output pixels), K = 32 input pixels.
// fp32 taps are split into bf16 hi + lo parts (two MFMAs), products are exact, accumulation is fp32,
// so the result matches the fp32-tap reference to ~1e-5.  Zero insertion (up), the stride (down) and
// the padding all live in the band structure of T; vertical polyphase skips the rows whose taps hit
// inserted zeros.  A = X^T is read from a [16-channel group][pixel][16 ch] LDS image with
// ds_read_b64_tr_b16 (the k <-> pixel map of conv_thin_wgrad.hip); D comes out as 4 consecutive
// channels of one pixel per lane -> 8-byte stores, 512 contiguous bytes per wave at C = 16.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct FirArgs {
    const bf16* x;
    const float* taps;
    bf16* y;
    int H, W, C, ky, kx, up_log2, down, px1, py1, Ho, Wo;
    int TH, ncg_log2, KC, RWr, RHmax, tiles_x, tiles_y, xstep, ncols, ncb;
    int64_t ntiles;
    unsigned rw_magic;  // ceil(2^32 / RWr)
    unsigned x_bytes;   // < 2^31: 32-bit buffer offsets
    int rowmajor;       // fast kernel: walk the tiles of an (image, channel block) row by row instead of column by column
};

__device__ __forceinline__ bf16x8 fir_tr_frag(const char* p) {
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 16 * 32));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}


// U: 16-byte chunks of the input region per lane (ceil(region chunks / 256)), all of them in flight in registers
template <int U>
__global__ void __launch_bounds__(256, U <= 6 ? 4 : 3) upfirdn2d_mfma_k(const FirArgs p) {
    extern __shared__ __attribute__((aligned(256))) char fir_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction: as an SGPR the row / region arithmetic below runs on the scalar unit
    // (PMC: 750 VALU instructions per wave and tile against 48 MFMAs, most of them uniform address math)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int up = 1 << p.up_log2, down = p.down;
    const int ncg = 1 << p.ncg_log2;
    char* const toep = fir_lds;                                  // [ky][KC][hi,lo][64 lanes][16 B]
    char* const reg = fir_lds + p.ky * p.KC * 2048;              // [ncg][RHmax * RWr pixels][32 B]
    const int PL = p.RHmax * p.RWr * 32;

    // ---- banded tap matrices, in MFMA B-fragment order (once per workgroup) ------------------------
    const int J0 = p.px1 & (up - 1);
    for (int e = tid; e < p.ky * p.KC * 64; e += 256) {
        const int l = e & 63, kc = (e >> 6) % p.KC, i = (e >> 6) / p.KC;
        const int xo = l & 15, gg = l >> 4;
        bf16x8 hi, lo;
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) {
            const int xrel = 32 * kc + (j8 < 4 ? 4 * gg + j8 : 16 + 4 * gg + j8 - 4);
            const int j = xrel * up - xo * down + J0;
            const float t = (j >= 0 && j < p.kx) ? p.taps[i * p.kx + j] : 0.f;
            hi[j8] = (bf16)t;
            lo[j8] = (bf16)(t - (float)hi[j8]);
        }
        char* dst = toep + ((i * p.KC + kc) * 2) * 1024 + l * 16;
        *reinterpret_cast<bf16x8*>(dst) = hi;
        *reinterpret_cast<bf16x8*>(dst + 1024) = lo;
    }

    // ---- persistent walk over a contiguous range of tiles: t -> (n, channel block, ty, tx) ---------
    const int64_t t0 = p.ntiles * blockIdx.x / gridDim.x;
    const int64_t t1 = p.ntiles * (blockIdx.x + 1) / gridDim.x;
    const int CP = ncg * 2;
    const int tiles_per_cb = p.tiles_x * p.tiles_y;

    auto decode = [&](int64_t t, int& n, int& cb, int& oy0, int& ox0) {
        const int64_t img = t / tiles_per_cb;
        const int rem = (int)(t - img * tiles_per_cb);
        n = (int)(img / p.ncb);
        cb = (int)(img - (int64_t)n * p.ncb);
        // column-major walk: a workgroup's consecutive tiles are vertical neighbours, so the halo rows a tile shares with
        // the one above it were read by this CU moments ago (L2) instead of by another workgroup on another XCD
        const int tx = rem / p.tiles_y;
        oy0 = (rem - tx * p.tiles_y) * p.TH;
        ox0 = tx * 32;
    };
    // Input region (zero outside the image and in the columns that only meet zero taps) -> registers -> LDS.
    // `issue` starts the loads of a tile's region (branch-free buffer loads: chunks outside the image read
    // zeros at an out-of-range offset), `commit` parks them in LDS one tile later: the next tile's HBM round
    // trip runs under this tile's MFMAs and stores.  (Ablations at 32x16x1024x1024: loads + stores alone took
    // 506 us of the 671 us the load -> barrier -> compute -> store sequence took; compute + stores alone 335.)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr unsigned INV = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), 0, p.x_bytes, 0x00020000);
    const int sh = p.ncg_log2 + 4;  // 8 * CP chunks per group of 8 pixels
    v4u pre[U];
    int pre_npix = 0;
    auto issue = [&](int n, int cb, int oy0, int ox0) {
        const int iy0 = (oy0 * down - p.py1 + up - 1) >> p.up_log2;
        const int bx0 = ox0 * down - p.px1;
        const int ix0 = (bx0 + up - 1) >> p.up_log2;
        const int iy1 = ((oy0 + p.TH - 1) * down - p.py1 + p.ky - 1) >> p.up_log2;
        const int npix = (iy1 - iy0 + 1) * p.RWr;
        pre_npix = npix;
        // wave-uniform byte offset of the region's first pixel (may be negative: folded into the lane offset)
        const int base = (((n * p.H + iy0) * p.W + ix0) * p.C + cb * ncg * 16) * 2;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = u * 256 + tid;
            const int p8 = idx >> sh, r8 = idx & ((1 << sh) - 1);
            const int c8 = r8 >> 3;
            const int pix = p8 * 8 + (r8 & 7);
            const int ry = (int)__umulhi((unsigned)pix, p.rw_magic), rx_ = pix - ry * p.RWr;  // pix / RWr, exact for pix < 2^16
            const bool ok = pix < npix && rx_ < p.ncols && (unsigned)(iy0 + ry) < (unsigned)p.H && (unsigned)(ix0 + rx_) < (unsigned)p.W;
            pre[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(base + ((ry * p.W + rx_) * p.C + c8 * 8) * 2) : INV, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = u * 256 + tid;
            const int p8 = idx >> sh, r8 = idx & ((1 << sh) - 1);
            const int c8 = r8 >> 3;
            const int pix = p8 * 8 + (r8 & 7);
            if (pix < pre_npix) *reinterpret_cast<v4u*>(reg + (c8 >> 1) * PL + pix * 32 + (c8 & 1) * 16) = pre[u];
        }
    };

    const int rows_per_wave = 4 >> p.ncg_log2;
    const int lane_off = (4 * g + q) * 32 + pp * 8;
    if (t0 < t1) {
        int n, cb, oy0, ox0;
        decode(t0, n, cb, oy0, ox0);
        issue(n, cb, oy0, ox0);
    }
    for (int64_t t = t0; t < t1; ++t) {
        int n, cb, oy0, ox0;
        decode(t, n, cb, oy0, ox0);
        const int iy0 = (oy0 * down - p.py1 + up - 1) >> p.up_log2;
        __syncthreads();  // previous tile's fragment reads (and the tap matrices' stores) are done
        commit();
        __syncthreads();
        if (t + 1 < t1) {
            int n2, cb2, oy2, ox2;
            decode(t + 1, n2, cb2, oy2, ox2);
            issue(n2, cb2, oy2, ox2);
        }
        // 8 output fragments (16 ch x 16 px) per wave: f -> (row, x half, channel group)
        f32x4 acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < p.ky; ++i) {
            for (int kc = 0; kc < p.KC; ++kc) {
                const char* tp = toep + ((i * p.KC + kc) * 2) * 1024 + lane * 16;
                const bf16x8 bhi = *reinterpret_cast<const bf16x8*>(tp);
                const bf16x8 blo = *reinterpret_cast<const bf16x8*>(tp + 1024);
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    const int cgi = f & (ncg - 1);
                    const int fx = (f >> p.ncg_log2) & 1;
                    const int rl = wave * rows_per_wave + (f >> (p.ncg_log2 + 1));
                    const int by = (oy0 + rl) * down - p.py1 + i;
                    if (by & (up - 1)) continue;  // this tap row falls on inserted zeros (wave-uniform)
                    const int ry = (by >> p.up_log2) - iy0;
                    const bf16x8 a = fir_tr_frag(reg + cgi * PL + (ry * p.RWr + fx * p.xstep + 32 * kc) * 32 + lane_off);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bhi, acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, blo, acc[f], 0, 0, 0);
                }
            }
        }
        // D: row = channel 4g + r, col = output pixel li -> 8-byte stores
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int cgi = f & (ncg - 1);
            const int fx = (f >> p.ncg_log2) & 1;
            const int oy = oy0 + wave * rows_per_wave + (f >> (p.ncg_log2 + 1));
            const int ox = ox0 + 16 * fx + li;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[f][r];
            *reinterpret_cast<bf16x4*>(p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.C + (cb * ncg + cgi) * 16 + 4 * g) = o;
        }
    }
}


// ---- the same kernel with its geometry at compile time (round 3) -----------------------------------------------------
// PMC on the kernel above: ~750 VALU instructions per wave and tile against 48 MFMAs -- chunk decoding with run-time
// shifts and a reciprocal multiply per load, fragment addresses per (fragment, tap row), 64-bit store addresses -- as
// much issue time as the tile's bytes take at HBM rate, so the op ran at ~0.45 of 8 TB/s with the memory system half idle.
// Here stride 1 (down = 1), taps <= 4 rows and the channel grouping are template parameters:
//   * a lane's chunks of the input region are decoded ONCE (byte offset from the region's origin, LDS slot, packed
//     (row, column)); per tile a chunk costs five VALU instructions: packed 16-bit subtract / max / compare against the
//     tile's wave-uniform valid (row, column) window, an add and a select;
//   * fragment addresses are lane base + an instruction immediate (without upsampling; with it, one scalar per row);
//   * stores are buffer stores at lane offset + scalar offset, the ragged-edge test one compare and one select.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// RX: rows per wave x RX (8 RX fragments per wave).  (Measured, not kept: RX = 2 under upsampling, where a tile's region is a
// quarter of its output -- 16 accumulators and the wider store phase cost occupancy: 3.2 -> 2.0 TB/s.)
template <int UPL, int KY, int NCGL, int RX = 1>
struct FirFastCfg {
    static constexpr int up = 1 << UPL, ncg = 1 << NCGL, TH = (16 >> NCGL) * RX, RPW = (4 >> NCGL) * RX, NF = 8 * RX, XSTEP = 16 >> UPL, RWR = 32 + XSTEP;
    static constexpr int RH = (TH - 1 + KY - 1) / up + 2, NPX = RH * RWR, PL = NPX * 32, SH = NCGL + 4;
    // Only the columns a tile's taps can reach (square FIR: kx <= KY) are ever loaded: the region's LDS image keeps the
    // RWR pitch the K blocks of the MFMA need, its other columns are zeroed once per workgroup.
    static constexpr int NCOLS = XSTEP + (15 + KY - 1) / up + 1, NPC = RH * NCOLS;
    static constexpr int NCHUNK = ((NPC + 7) >> 3) * 8 * 2 * ncg, U = (NCHUNK + 255) / 256;
    static constexpr int FULL = ((NPC >> 3) * 8 * 2 * ncg) / 256;  // slots whose 256 chunks all lie inside the region
    static constexpr int LDS = KY * 2048 + ncg * PL;
    // register sets of region chunks in flight per lane (tiles prefetched): with one, a tile's HBM round trip sat between
    // two barriers of every iteration (ablation: loads alone 174 us, stores alone 202 us, together 380 us on 32x64x256x256
    // upsampled: no overlap); two while the registers allow four waves per SIMD
    static constexpr int PD = U <= 8 ? 2 : 1;
};

template <int UPL, int KY, int NCGL>
__global__ void __launch_bounds__(256, (FirFastCfg<UPL, KY, NCGL>::U * FirFastCfg<UPL, KY, NCGL>::PD <= 8 ? 4 : 3)) upfirdn2d_fast_k(const FirArgs p) {
    typedef FirFastCfg<UPL, KY, NCGL> C;
    constexpr int up = C::up, ncg = C::ncg, TH = C::TH, RPW = C::RPW, XSTEP = C::XSTEP, RWR = C::RWR, RH = C::RH, NPX = C::NPX, PL = C::PL;
    constexpr int SH = C::SH, U = C::U, NF = C::NF, NCOLS = C::NCOLS, NPC = C::NPC, PD = C::PD;
    extern __shared__ __attribute__((aligned(256))) char fir_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    char* const toep = fir_lds;              // [KY][hi, lo][64 lanes][16 B]
    char* const reg = fir_lds + KY * 2048;   // [ncg][RH * RWR pixels][32 B]

    // ---- banded tap matrices, in MFMA B-fragment order (once per workgroup) ------------------------
    const int J0 = p.px1 & (up - 1);
    int my_lo = 0;
    for (int e = tid; e < KY * 64; e += 256) {
        const int l = e & 63, i = e >> 6;
        const int xo = l & 15, gg = l >> 4;
        bf16x8 hi, lo;
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) {
            const int xrel = j8 < 4 ? 4 * gg + j8 : 16 + 4 * gg + j8 - 4;
            const int j = xrel * up - xo + J0;
            const float t = (j >= 0 && j < p.kx) ? p.taps[i * p.kx + j] : 0.f;
            hi[j8] = (bf16)t;
            lo[j8] = (bf16)(t - (float)hi[j8]);
            my_lo |= (float)lo[j8] != 0.f;
        }
        char* dst = toep + (i * 2) * 1024 + l * 16;
        *reinterpret_cast<bf16x8*>(dst) = hi;
        *reinterpret_cast<bf16x8*>(dst + 1024) = lo;
    }

    const int64_t t0 = p.ntiles * blockIdx.x / gridDim.x;
    const int64_t t1 = p.ntiles * (blockIdx.x + 1) / gridDim.x;
    // taps that are exact in bf16 (the binomial kernels of Blur.make_kernel, nvidia_ops.py:73-78) have no low part: half the MFMAs
    const bool need_lo = __syncthreads_or(my_lo) != 0;
    // Walk: column-major inside (image, channel block) -- consecutive tiles are vertical neighbours and share halo rows
    // (measured, not kept: the channel blocks of a column next to each other in the walk, or the block fastest: 15-25 %
    // slower at 64 / 128 channels, a few % faster at 256 / 512).  The position is decoded once and then stepped: a
    // decode per tile was a 64-bit and three 32-bit integer divisions, ~300 instructions, twice per tile.
    struct Pos {
        int n, cb, tx, ty;
    };
    const int tiles_per_cb = p.tiles_x * p.tiles_y;
    const bool ROWMAJOR = p.rowmajor != 0;
    Pos cur;
    {
        const int64_t img = t0 / tiles_per_cb;
        const int rem = (int)(t0 - img * tiles_per_cb);
        cur.n = (int)(img / p.ncb);
        cur.cb = (int)(img - (int64_t)cur.n * p.ncb);
        if (ROWMAJOR) {
            cur.ty = rem / p.tiles_x;
            cur.tx = rem - cur.ty * p.tiles_x;
        } else {
            cur.tx = rem / p.tiles_y;
            cur.ty = rem - cur.tx * p.tiles_y;
        }
    }
    auto advance = [&](const Pos q) {
        Pos r = q;
        bool wrap;
        if (ROWMAJOR) {
            wrap = ++r.tx == p.tiles_x;
            if (wrap) {
                r.tx = 0;
                wrap = ++r.ty == p.tiles_y;
                if (wrap) r.ty = 0;
            }
        } else {
            wrap = ++r.ty == p.tiles_y;
            if (wrap) {
                r.ty = 0;
                wrap = ++r.tx == p.tiles_x;
                if (wrap) r.tx = 0;
            }
        }
        if (wrap && ++r.cb == p.ncb) {
            r.cb = 0;
            ++r.n;
        }
        return r;
    };

    // ---- this lane's chunks of the region, decoded once ------------------------------------------------
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr unsigned INV = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), 0, p.x_bytes, 0x00020000);
    unsigned ryx[U];   // (row << 16) | column of the chunk's pixel in the region; 0xffffffff where no tile has it
    int soff[U], ldso[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = u * 256 + tid;
        const int p8 = idx >> SH, r8 = idx & ((1 << SH) - 1);
        const int c8 = r8 >> 3;
        const int pc = p8 * 8 + (r8 & 7);              // pixel of the compact (RH x NCOLS) grid
        const int ry = pc / NCOLS, rxx = pc - ry * NCOLS;
        ryx[u] = (pc < NPC && rxx < p.ncols) ? ((unsigned)ry << 16 | (unsigned)rxx) : 0xffffffffu;
        soff[u] = ((ry * p.W + rxx) * p.C + c8 * 8) * 2;
        ldso[u] = pc < NPC ? (c8 >> 1) * PL + (ry * RWR + rxx) * 32 + (c8 & 1) * 16 : -1;
    }
    // the columns no chunk ever writes: zero (they meet zero taps, and 0 x stale bits must not be a NaN)
    for (int e = tid; e < ncg * RH * (RWR - NCOLS) * 2; e += 256) {
        const int h = e & 1, pz = e >> 1;
        const int gi = pz / (RH * (RWR - NCOLS)), r = pz - gi * (RH * (RWR - NCOLS));
        const int ry = r / (RWR - NCOLS), rxx = NCOLS + r - ry * (RWR - NCOLS);
        *reinterpret_cast<v4u*>(reg + gi * PL + (ry * RWR + rxx) * 32 + h * 16) = v4u{0u, 0u, 0u, 0u};
    }
    v4u pre[PD][U];
    auto issue = [&](v4u (&pre)[U], int n, int cb, int oy0, int ox0) {
        const int iy0 = (oy0 - p.py1 + up - 1) >> UPL;
        const int ix0 = (ox0 - p.px1 + up - 1) >> UPL;
        const int iy1 = (oy0 + TH - 1 - p.py1 + KY - 1) >> UPL;
        // the window of region (row, column) that lies inside the image and inside this tile's rows: wave-uniform
        int rlo = iy0 < 0 ? -iy0 : 0, rhi = min(iy1 - iy0, p.H - 1 - iy0);
        int xlo = ix0 < 0 ? -ix0 : 0, xhi = min(NCOLS - 1, p.W - 1 - ix0);
        if (rhi < rlo || xhi < xlo) { rlo = 0x7fff; rhi = 0x7fff; xlo = 0; xhi = 0; }
        const unsigned lo_u = (unsigned)rlo << 16 | (unsigned)xlo, rng_u = (unsigned)(rhi - rlo) << 16 | (unsigned)(xhi - xlo);
        const us2 lo_pk = __builtin_bit_cast(us2, lo_u), rng_pk = __builtin_bit_cast(us2, rng_u);
        const int base = (((n * p.H + iy0) * p.W + ix0) * p.C + cb * ncg * 16) * 2;  // may be negative: valid lanes add enough
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const us2 d = __builtin_bit_cast(us2, ryx[u]) - lo_pk;  // wraps to a large value below the window
            const us2 m = __builtin_elementwise_max(d, rng_pk);
            const bool ok = __builtin_bit_cast(unsigned, m) == rng_u;
#ifdef GAN_FIR_NOLOAD  // diagnostic build: no input traffic
            pre[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok && soff[u] == -12345 ? (unsigned)(base + soff[u]) : INV, 0, 0);
#else
            pre[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(base + soff[u]) : INV, 0, 0);
#endif
        }
    };
    auto commit = [&](const v4u (&pre)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < C::FULL || ldso[u] >= 0) *reinterpret_cast<v4u*>(reg + ldso[u]) = pre[u];  // (cells outside the window get zeros)
        }
    };

    // fragment base: this lane's (pixel, channel quad) of a 32-pixel K block, + the wave's first output row without upsampling
    const int lane_off = (4 * g + q) * 32 + pp * 8 + (UPL == 0 ? wave * RPW * RWR * 32 : 0);
    // after the fragment-pair swap of the epilogue: lane (g, li) holds channels 8 (g >> 1) .. + 7 of pixel li + 16 (g & 1)
    // (pair = two channel groups: channels 16 (g & 1) + 8 (g >> 1) .. + 7 of pixel li)
    const int st_pix = NCGL >= 1 ? li : li + 16 * (g & 1);
    const int st_lane = (st_pix * p.C + (NCGL >= 1 ? 16 * (g & 1) : 0) + 8 * (g >> 1)) * 2;
    const unsigned img_bytes = (unsigned)(p.Ho * p.Wo * p.C * 2);
    // tiles t0 .. t0 + PD - 1 in flight before the first barrier; iteration t parks tile t's chunks, then issues tile t + PD
    // into the register set it has just emptied
    Pos ahead = cur;
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        if (t0 + d < t1) issue(pre[d], ahead.n, ahead.cb, ahead.ty * TH, ahead.tx * 32);
        ahead = advance(ahead);
    }
    auto iteration = [&](auto setc, const int64_t t) {
        constexpr int SET = decltype(setc)::value;
        const int n = cur.n, cb = cur.cb, oy0 = cur.ty * TH, ox0 = cur.tx * 32;
        cur = advance(cur);
        __syncthreads();  // previous tile's fragment reads (and the tap matrices' stores) are done
        commit(pre[SET]);
        __syncthreads();
        if (t + PD < t1) issue(pre[SET], ahead.n, ahead.cb, ahead.ty * TH, ahead.tx * 32);
        ahead = advance(ahead);
        f32x4 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Fragment (f, tap row i) reads region row fr + i (no upsampling), or -- with 2x zero insertion -- only when
        // fr + i + wpar is even, row rbase + (fr + i + wpar) / 2: wpar (the parity of the wave's first output row minus the
        // pad) is the same for every tile of the launch, so which (f, i) pairs exist is decided by ONE wave-uniform branch
        // around two compile-time bodies, and every address is lane base + immediate.
        auto compute = [&](auto wparc, const int frag_base) {
            constexpr int WPAR = decltype(wparc)::value;
#pragma unroll
            for (int i = 0; i < KY; ++i) {
                const char* tp = toep + (i * 2) * 1024 + lane * 16;
                const bf16x8 bhi = *reinterpret_cast<const bf16x8*>(tp);
                const bf16x8 blo = *reinterpret_cast<const bf16x8*>(tp + 1024);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int cgi = f & (ncg - 1), fx = (f >> NCGL) & 1, fr = f >> (NCGL + 1);
                    if (UPL == 1 && ((fr + i + WPAR) & 1)) continue;  // (compile-time) this tap row falls on inserted zeros
                    const int row = UPL == 0 ? fr + i : (fr + i + WPAR) >> 1;
                    const bf16x8 a = fir_tr_frag(reg + frag_base + cgi * PL + (row * RWR + fx * XSTEP) * 32);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bhi, acc[f], 0, 0, 0);
                    if (need_lo) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, blo, acc[f], 0, 0, 0);
                }
            }
        };
        if constexpr (UPL == 0) {
            compute(std::integral_constant<int, 0>{}, lane_off);
        } else {
            const int E = oy0 - p.py1 + wave * RPW, wpar = E & 1;
            const int rbase = ((E - wpar) >> 1) - ((oy0 - p.py1 + 1) >> 1);  // >= 0: the region starts at the tile's first tap row
            if (wpar) compute(std::integral_constant<int, 1>{}, lane_off + rbase * (RWR * 32));
            else compute(std::integral_constant<int, 0>{}, lane_off + rbase * (RWR * 32));
        }
        // D: row = channel 4g + r, col = output pixel li -> 8-byte buffer stores into image n (a resource per image: outputs
        // may exceed 2^31 bytes in all)
        const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)n * p.Ho * p.Wo * p.C, 0, img_bytes, 0x00020000);
        // Fragment pairs: v_permlane16_swap trades the odd 16-lane rows' results of the first fragment for the even rows'
        // results of the second, after which a lane holds 8 consecutive channels of ONE pixel -- 16-byte stores, half as
        // many.  The pair is two adjacent channel groups of the same 16 pixels where the tile has them (a store instruction
        // then writes 64 contiguous bytes per pixel), else the two x halves of a row (C = 16: 32 pixels x 32 bytes, contiguous).
        constexpr bool PAIR_CG = NCGL >= 1;
        constexpr int F2 = PAIR_CG ? 1 : (1 << NCGL);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (f & F2) continue;
            const int cgi = f & (ncg - 1), fx = (f >> NCGL) & 1, fr = f >> (NCGL + 1);
            const int oy = oy0 + wave * RPW + fr;                                  // wave-uniform
            const int wlim = oy < p.Ho ? p.Wo - ox0 - 16 * fx : 0;                 // pixels of the pair inside the image
            const int sbase = ((oy * p.Wo + ox0 + 16 * fx) * p.C + (cb * ncg + cgi) * 16) * 2;
            bf16x4 oa, ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                oa[r] = (bf16)acc[f][r];
                ob[r] = (bf16)acc[f + F2][r];
            }
            typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
            v2u_ xa = __builtin_bit_cast(v2u_, oa), xb = __builtin_bit_cast(v2u_, ob);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const auto sw = __builtin_amdgcn_permlane16_swap(xa[d], xb[d], false, false);
                xa[d] = sw[0];
                xb[d] = sw[1];
            }
            const v4u o = {xa[0], xa[1], xb[0], xb[1]};
#ifdef GAN_FIR_NOSTORE  // diagnostic build: no output traffic
            __builtin_amdgcn_raw_buffer_store_b128(o, ry_, st_pix < wlim && o[0] == 0x12345678u ? (unsigned)st_lane : INV, sbase, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b128(o, ry_, st_pix < wlim ? (unsigned)st_lane : INV, sbase, 0);
#endif
        }
    };
    for (int64_t t = t0; t < t1; t += PD) {
        iteration(std::integral_constant<int, 0>{}, t);
        if (PD == 2 && t + 1 < t1) iteration(std::integral_constant<int, PD - 1>{}, t + 1);
    }
}

}  // namespace

// 1 = served, 0 = shape not eligible (caller falls back to the VALU kernels), < 0 = error.
int gan_upfirdn2d_mfma(const void* x, const float* taps, void* y, int N, int H, int W, int C, int ky, int kx, int up,
                       int down, int px1, int py1, int Ho, int Wo, hipStream_t st) {
    if ((C & 15) || (up != 1 && up != 2) || (down != 1 && down != 2)) return 0;
    if (Wo < 16 || Ho < 2 || ky > 16 || kx > 16 || N > 65535 || C / 16 > 65535) return 0;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 7)) return 0;
    if ((int64_t)N * H * W * C * 2 >= (1LL << 31)) return 0;  // 32-bit buffer offsets in the region loads
    const int kneed = (15 * down + kx - 1) / up;  // largest input offset inside a 16-pixel output group
    if (kneed > 63) return 0;
    FirArgs a;
    a.x = (const bf16*)x; a.taps = taps; a.y = (bf16*)y;
    a.H = H; a.W = W; a.C = C; a.ky = ky; a.kx = kx; a.up_log2 = up == 2 ? 1 : 0; a.down = down;
    a.px1 = px1; a.py1 = py1; a.Ho = Ho; a.Wo = Wo;
    a.KC = kneed > 31 ? 2 : 1;
    a.xstep = 16 * down / up;
    a.RWr = 32 * a.KC + a.xstep;
    a.ncols = a.xstep + kneed + 1;
    a.rw_magic = (unsigned)((0x100000000ULL + a.RWr - 1) / a.RWr);
    const bool fast = (((uintptr_t)y) & 15) == 0 && down == 1 && a.KC == 1 && ky >= 2 && ky <= 4 && kx <= ky && (int64_t)Ho * Wo * C * 2 < (1LL << 31) && !(getenv("GAN_AMD_FIR_OLD") && getenv("GAN_AMD_FIR_OLD")[0] == '1');
    int lds = 0;
    int ncg_log2 = 2;
    for (;; --ncg_log2) {
        const int ncg = 1 << ncg_log2;
        if ((C / 16) % ncg == 0) {
            a.TH = 16 >> ncg_log2;  // (x FirFastCfg::RX = 1)
            a.RHmax = ((a.TH - 1) * down + ky - 1) / up + 2;
            lds = ky * a.KC * 2048 + ncg * a.RHmax * a.RWr * 32;
            // >= 4 workgroups per CU (their load / MFMA phases overlap); four channel groups = whole 128-byte lines of a
            // >= 64-channel pixel are worth a third workgroup less (the fast kernel; half lines ran at 0.7x the rate)
            if (lds <= 40 * 1024 || (fast && ncg == 4 && lds <= 52 * 1024)) break;
        }
        if (ncg_log2 == 0) return 0;
    }
    a.ncg_log2 = ncg_log2;
    a.tiles_x = (Wo + 31) / 32;
    a.tiles_y = (Ho + a.TH - 1) / a.TH;
    a.ncb = C / 16 >> ncg_log2;
    a.ntiles = (int64_t)a.tiles_x * a.tiles_y * a.ncb * N;
    int per_cu = (160 * 1024) / lds;
    if (per_cu > 4) per_cu = 4;
    if (fast && lds > 40 * 1024) per_cu = 3;
    int64_t grid = 256LL * per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    a.x_bytes = (unsigned)((int64_t)N * H * W * C * 2);
    {
        // under 2x upsampling the output is 4x the input: rows of adjacent tiles written one after the other (3.3 -> 3.5 TB/s
        // at 64 channels) matter more than the halo rows shared with the tile above (stride 1: row-major is 3-6 % slower)
        const char* e = getenv("GAN_AMD_FIR_ROWMAJOR");
        a.rowmajor = e ? atoi(e) : (up == 2 ? 1 : 0);
    }
    if (fast) {
#define FIR_FAST(UPL_, KY_, NCGL_)                                                                                          \
    do {                                                                                                                    \
        if constexpr (FirFastCfg<UPL_, KY_, NCGL_>::U <= 11 && FirFastCfg<UPL_, KY_, NCGL_>::LDS <= 52 * 1024) {             \
            if (FirFastCfg<UPL_, KY_, NCGL_>::LDS != lds || FirFastCfg<UPL_, KY_, NCGL_>::TH != a.TH) {                     \
                gan_set_error("upfirdn2d(mfma): plan mismatch");                                                            \
                return GAN_ERR_ARG;                                                                                         \
            }                                                                                                               \
            hipLaunchKernelGGL((upfirdn2d_fast_k<UPL_, KY_, NCGL_>), dim3((unsigned)grid), dim3(256), (size_t)lds, st, a); \
        } else {                                                                                                            \
            gan_set_error("upfirdn2d(mfma): plan mismatch");                                                                \
            return GAN_ERR_ARG;                                                                                             \
        }                                                                                                                   \
    } while (0)
#define FIR_FAST_KY(UPL_, NCGL_)                      \
    do {                                              \
        if (ky == 2) FIR_FAST(UPL_, 2, NCGL_);        \
        else if (ky == 3) FIR_FAST(UPL_, 3, NCGL_);   \
        else FIR_FAST(UPL_, 4, NCGL_);                \
    } while (0)
        if (up == 1) {
            if (ncg_log2 == 2) FIR_FAST_KY(0, 2);
            else if (ncg_log2 == 1) FIR_FAST_KY(0, 1);
            else FIR_FAST_KY(0, 0);
        } else {
            if (ncg_log2 == 2) FIR_FAST_KY(1, 2);
            else if (ncg_log2 == 1) FIR_FAST_KY(1, 1);
            else FIR_FAST_KY(1, 0);
        }
#undef FIR_FAST_KY
#undef FIR_FAST
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            gan_set_error("upfirdn2d(mfma, fast): launch failed: %s", hipGetErrorString(e));
            return GAN_ERR_LAUNCH;
        }
        return 1;
    }
    const int nchunk_max = ((a.RHmax * a.RWr + 7) >> 3) * 8 * (2 << ncg_log2);
    const int slots = (nchunk_max + 255) / 256;
    if (slots <= 4)
        hipLaunchKernelGGL(upfirdn2d_mfma_k<4>, dim3((unsigned)grid), dim3(256), (size_t)lds, st, a);
    else if (slots <= 6)
        hipLaunchKernelGGL(upfirdn2d_mfma_k<6>, dim3((unsigned)grid), dim3(256), (size_t)lds, st, a);
    else if (slots <= 9)
        hipLaunchKernelGGL(upfirdn2d_mfma_k<9>, dim3((unsigned)grid), dim3(256), (size_t)lds, st, a);
    else
        return 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gan_set_error("upfirdn2d(mfma): launch failed: %s", hipGetErrorString(e));
        return GAN_ERR_LAUNCH;
    }
    return 1;
}
