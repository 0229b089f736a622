// The generator's mapping network (stylegan2.py:70-72,93: LayerNorm(z_dim) then 8 x [Linear(z_dim, z_dim), LeakyReLU(0.2)])
// and the style split behind it, for z_dim = 128 (the reference's default and what every recipe uses).
//
// The network is 131k MACs per sample: nothing to optimise in arithmetic, but as library calls it is 17 launches forward
// and ~50 backward (a GEMM, a bias add, an activation per layer; two GEMMs, a mask, a column sum per layer going back),
// each at the ~4.5 us floor of a dependent launch -- 0.4 ms of a 15 ms iteration at a per-GPU batch of 32.  Here:
//   gan_mapping_fwd : one launch.  A workgroup takes 8 rows of z through all layers; the layer's 128 x 128 weight is
//                     staged transposed in LDS (coalesced global reads, conflict-free LDS reads), activations stay in LDS
//                     and are also written out (the backward needs every layer's output, as autograd would keep them).
//   gan_mapping_bwd : two launches.  The same 8-row workgroups walk the layers backwards (mask, bias-gradient column sums,
//                     weight-gradient outer products, the product with W for the next layer down), writing per-workgroup
//                     partial gradients to a caller workspace; a fold kernel sums them in a fixed order (deterministic).
//   gan_split_cols  : one launch.  [N][T] -> the concatenation of contiguous [N][C_l] blocks (every ModulatedConv2d's
//                     style s_l = A_l(w) + 1 comes out of ONE GEMM over the stacked A_l; the kernels read s_l contiguous).
#include "common.h"

namespace {

constexpr int MD = 128;        // z_dim
constexpr int MROWS = 8;       // rows of z per workgroup
constexpr int MTPB = 256;
constexpr int MAXL = 8;        // layers
constexpr int WT_PITCH = MD + 1;

struct MapParams {
    const float* W[MAXL];  // (out, in) row-major, as nn.Linear keeps it
    const float* B[MAXL];
    int L;
};

// stage W (row-major [t][k]) transposed into LDS: wt[k * WT_PITCH + t].  16-byte loads, all 16 of a thread in flight before
// the first store (the network is a latency chain: 8 layers x [stage, 128 multiply-adds], a handful of workgroups)
__device__ __forceinline__ void stage_transposed(const float* __restrict__ W, float* __restrict__ wt) {
    float4 v[MD * MD / 4 / MTPB];
#pragma unroll
    for (int j = 0; j < MD * MD / 4 / MTPB; ++j) v[j] = reinterpret_cast<const float4*>(W)[j * MTPB + threadIdx.x];
#pragma unroll
    for (int j = 0; j < MD * MD / 4 / MTPB; ++j) {
        const int idx = (j * MTPB + threadIdx.x) * 4, t = idx >> 7, k = idx & (MD - 1);
        wt[k * WT_PITCH + t] = v[j].x;
        wt[(k + 1) * WT_PITCH + t] = v[j].y;
        wt[(k + 2) * WT_PITCH + t] = v[j].z;
        wt[(k + 3) * WT_PITCH + t] = v[j].w;
    }
}

// stage W as it is (row-major [t][k]) into LDS: the backward's product g W reads row t, consecutive threads consecutive k
__device__ __forceinline__ void stage_rows(const float* __restrict__ W, float* __restrict__ wl) {
    float4 v[MD * MD / 4 / MTPB];
#pragma unroll
    for (int j = 0; j < MD * MD / 4 / MTPB; ++j) v[j] = reinterpret_cast<const float4*>(W)[j * MTPB + threadIdx.x];
#pragma unroll
    for (int j = 0; j < MD * MD / 4 / MTPB; ++j) reinterpret_cast<float4*>(wl)[j * MTPB + threadIdx.x] = v[j];
}

__global__ void __launch_bounds__(MTPB) mapping_fwd_k(const float* __restrict__ z, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                      const MapParams p, float* __restrict__ acts, float* __restrict__ stats, int N, float slope,
                                                      float eps) {
    extern __shared__ float wt[];  // MD x WT_PITCH floats (66 KB: past the static limit, so sized at launch)
    __shared__ float cur[MROWS][MD];
    const int row0 = blockIdx.x * MROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // LayerNorm: wave w normalises rows 2w, 2w + 1 (biased variance, as nn.LayerNorm)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wave * 2 + j, n = row0 + r;
        float a = 0.f, b = 0.f;
        if (n < N) {
            a = z[(int64_t)n * MD + lane];
            b = z[(int64_t)n * MD + lane + 64];
        }
        float s = a + b;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s * (1.f / MD);
        float v = (a - mean) * (a - mean) + (b - mean) * (b - mean);
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        const float rstd = rsqrtf(v * (1.f / MD) + eps);
        const float ya = (a - mean) * rstd * (ln_w ? ln_w[lane] : 1.f) + (ln_b ? ln_b[lane] : 0.f);
        const float yb = (b - mean) * rstd * (ln_w ? ln_w[lane + 64] : 1.f) + (ln_b ? ln_b[lane + 64] : 0.f);
        cur[r][lane] = ya;
        cur[r][lane + 64] = yb;
        if (n < N) {
            acts[(int64_t)n * MD + lane] = ya;
            acts[(int64_t)n * MD + lane + 64] = yb;
            if (lane == 0) {
                stats[n * 2] = mean;
                stats[n * 2 + 1] = rstd;
            }
        }
    }
    // thread (t, half): output feature t of rows 4*half .. 4*half + 3
    const int t = threadIdx.x & (MD - 1), half = threadIdx.x >> 7;
    for (int l = 0; l < p.L; ++l) {
        __syncthreads();  // cur is complete, wt is free
        stage_transposed(p.W[l], wt);
        __syncthreads();
        // four partial sums per output (k mod 4), added pairwise at the end: the rounding error of a 128-term dot product
        // then grows like that of a blocked library GEMM, not like a 128-long chain
        float acc[4][4];
        const float bias = p.B[l] ? p.B[l][t] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
#pragma unroll 2
        for (int k = 0; k < MD; k += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float w = wt[(k + q) * WT_PITCH + t];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][q] = fmaf(cur[half * 4 + j][k + q], w, acc[j][q]);
            }
        }
        __syncthreads();  // everyone has read cur
        float* __restrict__ out = acts + (int64_t)(l + 1) * N * MD;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = lrelu_f(((acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3])) + bias, slope);
            cur[half * 4 + j][t] = y;
            const int n = row0 + half * 4 + j;
            if (n < N) out[(int64_t)n * MD + t] = y;
        }
    }
}

// per-workgroup partials, in this order: dW[L][MD][MD], dB[L][MD], dln_w[MD], dln_b[MD]
__host__ __device__ constexpr int64_t map_partial_floats(int L) { return (int64_t)L * (MD * MD + MD) + 2 * MD; }

__global__ void __launch_bounds__(MTPB) mapping_bwd_k(const float* __restrict__ gout, const float* __restrict__ acts, const float* __restrict__ stats,
                                                      const float* __restrict__ z, const float* __restrict__ ln_w, const MapParams p,
                                                      float* __restrict__ ws, int N, float slope) {
    extern __shared__ float wl[];      // the layer's weight, MD x MD floats (sized at launch)
    __shared__ float g[MROWS][MD];     // gradient at the layer's output, then at its pre-activation
    __shared__ float prev[MROWS][MD];  // the layer's input
    __shared__ float gn[MROWS][MD];    // gradient at the layer's input
    const int row0 = blockIdx.x * MROWS;
    float* __restrict__ part = ws + (int64_t)blockIdx.x * map_partial_floats(p.L);
    const int t = threadIdx.x & (MD - 1), half = threadIdx.x >> 7;
    for (int idx = threadIdx.x; idx < MROWS * MD; idx += MTPB) {
        const int r = idx >> 7, c = idx & (MD - 1), n = row0 + r;
        g[r][c] = n < N ? gout[(int64_t)n * MD + c] : 0.f;
    }
    for (int l = p.L - 1; l >= 0; --l) {
        __syncthreads();  // g is complete; everyone is done with the previous layer's wl and prev
        stage_rows(p.W[l], wl);
        const float* __restrict__ yl = acts + (int64_t)(l + 1) * N * MD;
        const float* __restrict__ xl = acts + (int64_t)l * N * MD;
        for (int idx = threadIdx.x; idx < MROWS * MD; idx += MTPB) {
            const int r = idx >> 7, c = idx & (MD - 1), n = row0 + r;
            const float y = n < N ? yl[(int64_t)n * MD + c] : 0.f;
            g[r][c] *= lrelu_mask(y, slope);  // the output's sign is the pre-activation's (slope > 0)
            prev[r][c] = n < N ? xl[(int64_t)n * MD + c] : 0.f;
        }
        __syncthreads();
        // bias gradient: column sums (threads of half 0)
        if (half == 0) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < MROWS; ++r) s += g[r][t];
            part[(int64_t)p.L * MD * MD + l * MD + t] = s;
        }
        // weight gradient: dW[t][k] = sum_r g[r][t] * prev[r][k]; thread (t, half) owns k in [64*half, 64*half + 64)
        {
            float gr[MROWS];
#pragma unroll
            for (int r = 0; r < MROWS; ++r) gr[r] = g[r][t];
            float* __restrict__ dw = part + (int64_t)l * MD * MD + t * MD + half * 64;
#pragma unroll 4
            for (int k = 0; k < 64; ++k) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < MROWS; ++r) s = fmaf(gr[r], prev[r][half * 64 + k], s);
                dw[k] = s;
            }
        }
        // gradient at the layer's input: gn[r][k] = sum_t g[r][t] * W[t][k]; thread (k = t, half) owns rows 4*half ..
        {
            float acc[4][4];  // partial sums over tt mod 4, as in the forward
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
#pragma unroll 2
            for (int tt = 0; tt < MD; tt += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float w = wl[(tt + q) * MD + t];  // consecutive threads, consecutive columns: conflict-free
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][q] = fmaf(g[half * 4 + j][tt + q], w, acc[j][q]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) gn[half * 4 + j][t] = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < MROWS * MD; idx += MTPB) g[idx >> 7][idx & (MD - 1)] = gn[idx >> 7][idx & (MD - 1)];
    }
    __syncthreads();
    // LayerNorm's affine parameters: g is the gradient at its output
    if (half == 0) {
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int r = 0; r < MROWS; ++r) {
            const int n = row0 + r;
            if (n < N) {
                const float xhat = (z[(int64_t)n * MD + t] - stats[n * 2]) * stats[n * 2 + 1];
                sw = fmaf(g[r][t], xhat, sw);
                sb += g[r][t];
            }
        }
        float* __restrict__ tail = part + (int64_t)p.L * (MD * MD + MD);
        tail[t] = sw;
        tail[MD + t] = sb;
    }
    (void)ln_w;
}

// out[e] = sum over the nb workgroups' partials (fixed order)
__global__ void __launch_bounds__(MTPB) mapping_fold_k(const float* __restrict__ ws, float* __restrict__ out, int nb, int64_t L) {
    const int64_t e = (int64_t)blockIdx.x * MTPB + threadIdx.x;
    if (e >= L) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += ws[(int64_t)b * L + e];
    out[e] = s;
}

// [N][T] -> blocks: layer l (columns off[l] .. off[l+1]) becomes the contiguous [N][C_l] block at N * off[l]
__global__ void __launch_bounds__(MTPB) split_cols_k(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ off, int nl,
                                                     int N, int T, int reverse) {
    const int64_t idx = (int64_t)blockIdx.x * MTPB + threadIdx.x;
    if (idx >= (int64_t)N * T) return;
    const int n = (int)(idx / T), col = (int)(idx - (int64_t)n * T);
    int lo = 0, hi = nl;  // the layer whose column range holds col
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= col) lo = mid;
        else hi = mid;
    }
    const int64_t blk = (int64_t)N * off[lo] + (int64_t)n * (off[lo + 1] - off[lo]) + (col - off[lo]);
    if (reverse) out[idx] = in[blk];
    else out[blk] = in[idx];
}

int check_params(const float* const* W, const float* const* B, int L, MapParams& p, const char* who) {
    if (L < 1 || L > MAXL || !W) {
        gan_set_error("%s: 1..%d layers (got %d)", who, MAXL, L);
        return GAN_ERR_ARG;
    }
    p.L = L;
    for (int l = 0; l < MAXL; ++l) {
        p.W[l] = l < L ? W[l] : nullptr;
        p.B[l] = (l < L && B) ? B[l] : nullptr;
        if (l < L && (!p.W[l] || ((uintptr_t)p.W[l] & 15))) {
            gan_set_error("%s: weight of layer %d is null or not 16-byte aligned", who, l);
            return GAN_ERR_ARG;
        }
    }
    return GAN_OK;
}

}  // namespace

extern "C" int gan_mapping_fwd(const float* z, const float* ln_w, const float* ln_b, const float* const* W, const float* const* B, int L,
                               float* acts, float* stats, int N, int dim, float slope, float eps, gan_stream_t stream) {
    GAN_CHECK_ARG(dim == MD, "mapping_fwd: z_dim %d (this kernel serves %d)", dim, MD);
    GAN_CHECK_ARG(N >= 0 && slope > 0.f, "mapping_fwd: bad batch / slope");
    MapParams p;
    const int rc = check_params(W, B, L, p, "mapping_fwd");
    if (rc != GAN_OK) return rc;
    if (N == 0) return GAN_OK;
    GAN_CHECK_ARG(z && acts && stats, "mapping_fwd: null pointer");
    hipLaunchKernelGGL(mapping_fwd_k, dim3((unsigned)ceil_div64(N, MROWS)), dim3(MTPB), sizeof(float) * MD * WT_PITCH, (hipStream_t)stream, z, ln_w, ln_b, p, acts, stats, N, slope,
                       eps);
    GAN_CHECK_LAUNCH("mapping_fwd");
    return GAN_OK;
}

extern "C" int64_t gan_mapping_bwd_workspace(int N, int L) { return N > 0 && L > 0 ? ceil_div64(N, MROWS) * map_partial_floats(L) : 0; }

extern "C" int gan_mapping_bwd(const float* gout, const float* acts, const float* stats, const float* z, const float* ln_w, const float* const* W, int L,
                               float* grads, float* workspace, int64_t workspace_floats, int N, int dim, float slope, gan_stream_t stream) {
    GAN_CHECK_ARG(dim == MD, "mapping_bwd: z_dim %d (this kernel serves %d)", dim, MD);
    GAN_CHECK_ARG(N > 0 && slope > 0.f, "mapping_bwd: bad batch / slope");
    MapParams p;
    const int rc = check_params(W, nullptr, L, p, "mapping_bwd");
    if (rc != GAN_OK) return rc;
    GAN_CHECK_ARG(gout && acts && stats && z && grads, "mapping_bwd: null pointer");
    GAN_CHECK_ARG(workspace && workspace_floats >= gan_mapping_bwd_workspace(N, L), "mapping_bwd: needs a workspace of %lld floats",
                  (long long)gan_mapping_bwd_workspace(N, L));
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)ceil_div64(N, MROWS);
    hipLaunchKernelGGL(mapping_bwd_k, dim3((unsigned)nb), dim3(MTPB), sizeof(float) * MD * MD, st, gout, acts, stats, z, ln_w, p, workspace, N, slope);
    GAN_CHECK_LAUNCH("mapping_bwd");
    const int64_t Lf = map_partial_floats(L);
    hipLaunchKernelGGL(mapping_fold_k, dim3((unsigned)ceil_div64(Lf, MTPB)), dim3(MTPB), 0, st, workspace, grads, nb, Lf);
    GAN_CHECK_LAUNCH("mapping_bwd(fold)");
    return GAN_OK;
}

extern "C" int gan_split_cols(const float* in, float* out, const int* offsets, int nl, int N, int T, int reverse, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && T >= 0 && nl >= 1, "split_cols: bad shape");
    if (N == 0 || T == 0) return GAN_OK;
    GAN_CHECK_ARG(in && out && offsets, "split_cols: null pointer");
    const int64_t blocks = ceil_div64((int64_t)N * T, MTPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "split_cols: tensor too large");
    hipLaunchKernelGGL(split_cols_k, dim3((unsigned)blocks), dim3(MTPB), 0, (hipStream_t)stream, in, out, offsets, nl, N, T, reverse);
    GAN_CHECK_LAUNCH("split_cols");
    return GAN_OK;
}
