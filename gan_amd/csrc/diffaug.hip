// DiffAugment (diff_augment.py:13-60: ColorJitter -> RandomTranslate -> RandomCutout in front of the discriminator,
// train_celeba.py:63) as ONE streaming pass over the <= 4-channel image tensor, plus a per-sample sum in front of it.
//
// The reference runs eight elementwise / reduction / pad / slice-assign kernels per call, each a round trip over the
// batch.  All of it is an affine map of the image,  y = Cut . T . Con . Sat (x + beta),  with
//   beta = u_b - 0.5                                  rand_brightness (:13-14)
//   Sat v = m + (v - m) * S,  m = channel mean, S = 2 u_s    rand_saturation (:17-19)     (per pixel, symmetric)
//   Con v = mu + (v - mu) * K, mu = sample mean, K = u_c + 0.5   rand_contrast (:22-24)   (per sample, symmetric)
//   T     = F.pad(x, (tx, -tx, ty, -ty))               translate (:35-36): out[y][x] = in[y - ty][x - tx] or 0
//   Cut   = zero rows cy .. cy + H/2, columns cx .. cx + H/2   RandomCutout (:45-51; the reference uses H // 2 for both)
// Sat and Con commute and the mean of Sat(x + beta) is mean(x) + beta, so the only reduction the forward needs is the
// per-sample sum of the INPUT.  The gradient is the adjoint  Sat . Con . T' . Cut  (one sum of the shifted, masked
// incoming gradient, one pass), and the gradient of that -- R1 differentiates through the augmentation -- is the linear
// part of the forward (mode 1: beta = 0).  Three modes of two kernels serve every order.
// u_b, u_s, u_c are per-sample device floats and (tx, ty, cy, cx) device ints: nothing here needs the host to know the
// draws, so the augmented discriminator pass can be captured in a HIP graph (the reference's `.item()` calls cannot).
#include "common.h"

namespace {

constexpr int DTPB = 256;
constexpr int DA_MAXC = 4;
constexpr int DA_BLOCKS = 64;  // partial-sum workgroups per sample

struct DaGeom {
    int64_t sn, sc, sh, sw;  // element strides of the input (any layout)
    int N, C, H, W;
};

__device__ __forceinline__ bool da_cut(int y, int x, int cy, int cx, int half) { return y >= cy && y < cy + half && x >= cx && x < cx + half; }

// stage 1: part[n][block] = sum over this block's pixels (and all channels) of in -- for the adjoint only of the pixels
// whose value survives Cut and T' (not cut out, and translated from inside the image)
template <typename T>
__global__ void __launch_bounds__(DTPB) da_sums_k(const T* __restrict__ in, const DaGeom g, const int* __restrict__ shift, int adjoint,
                                                  float* __restrict__ part) {
    __shared__ float red[DTPB / 64];
    const int n = blockIdx.y;
    const int tx = shift[0], ty = shift[1], cy = shift[2], cx = shift[3], half = g.H / 2;
    const int HW = g.H * g.W;
    float acc = 0.f;
    for (int pix = blockIdx.x * DTPB + threadIdx.x; pix < HW; pix += DA_BLOCKS * DTPB) {
        const int y = pix / g.W, x = pix - y * g.W;
        bool ok = true;
        if (adjoint) ok = !da_cut(y, x, cy, cx, half) && (unsigned)(y - ty) < (unsigned)g.H && (unsigned)(x - tx) < (unsigned)g.W;
        if (!ok) continue;
        const T* __restrict__ p = in + (int64_t)n * g.sn + (int64_t)y * g.sh + (int64_t)x * g.sw;
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < g.C) acc += (float)p[c * g.sc];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < DTPB / 64; ++w) t += red[w];
        part[n * DA_BLOCKS + blockIdx.x] = t;
    }
}

// stage 2: one wave per sample folds its partials in a fixed order
__global__ void __launch_bounds__(64) da_fold_k(const float* __restrict__ part, float* __restrict__ sums) {
    float t = part[blockIdx.x * DA_BLOCKS + threadIdx.x];  // DA_BLOCKS == 64
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}

// mode 0: y = A x + c (the augmentation);  mode 1: y = A x (its linear part);  mode 2: y = A' x (its adjoint).
// One thread per output pixel; out is [N][H][W][C] (channels-last) in the input's dtype.
template <typename T>
__global__ void __launch_bounds__(DTPB) da_apply_k(const T* __restrict__ in, const DaGeom g, T* __restrict__ out, const float* __restrict__ sums,
                                                   const float* __restrict__ usk, const int* __restrict__ shift, int mode) {
    const int n = blockIdx.y;
    const int pix = blockIdx.x * DTPB + threadIdx.x;
    if (pix >= g.H * g.W) return;
    const int y = pix / g.W, x = pix - y * g.W;
    const int tx = shift[0], ty = shift[1], cy = shift[2], cx = shift[3], half = g.H / 2;
    const float beta = mode == 0 ? usk[n * 3 + 0] - 0.5f : 0.f;
    const float S = 2.f * usk[n * 3 + 1], K = usk[n * 3 + 2] + 0.5f;
    const float mean = sums[n] / (float)((int64_t)g.C * g.H * g.W);
    const float invc = 1.f / (float)g.C;
    T* __restrict__ o = out + ((int64_t)n * g.H * g.W + pix) * g.C;
    float v[DA_MAXC] = {0.f, 0.f, 0.f, 0.f};
    if (mode != 2) {
        const int sy = y - ty, sx = x - tx;
        if (da_cut(y, x, cy, cx, half) || (unsigned)sy >= (unsigned)g.H || (unsigned)sx >= (unsigned)g.W) {
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c)
                if (c < g.C) o[c] = (T)0.f;
            return;
        }
        const T* __restrict__ p = in + (int64_t)n * g.sn + (int64_t)sy * g.sh + (int64_t)sx * g.sw;
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < g.C) {
                v[c] = (float)p[c * g.sc] + beta;
                m += v[c];
            }
        m *= invc;
        const float mu = mean + beta;
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < g.C) {
                const float s = m + (v[c] - m) * S;
                o[c] = (T)(mu + (s - mu) * K);
            }
    } else {
        const int dy = y + ty, dx = x + tx;
        const bool live = (unsigned)dy < (unsigned)g.H && (unsigned)dx < (unsigned)g.W && !da_cut(dy, dx, cy, cx, half);
        float m = 0.f;
        if (live) {
            const T* __restrict__ p = in + (int64_t)n * g.sn + (int64_t)dy * g.sh + (int64_t)dx * g.sw;
#pragma unroll
            for (int c = 0; c < DA_MAXC; ++c)
                if (c < g.C) v[c] = (float)p[c * g.sc];
        }
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < g.C) {
                v[c] = K * v[c] + (1.f - K) * mean;  // Con' (the mean of the shifted, masked gradient)
                m += v[c];
            }
        m *= invc;
#pragma unroll
        for (int c = 0; c < DA_MAXC; ++c)
            if (c < g.C) o[c] = (T)(S * v[c] + (1.f - S) * m);  // Sat'
    }
}

template <typename T>
int da_launch(const void* in, const DaGeom& g, void* out, const float* usk, const int* shift, int mode, float* ws, hipStream_t st) {
    float* sums = ws + (int64_t)g.N * DA_BLOCKS;
    hipLaunchKernelGGL(da_sums_k<T>, dim3(DA_BLOCKS, (unsigned)g.N), dim3(DTPB), 0, st, (const T*)in, g, shift, mode == 2 ? 1 : 0, ws);
    GAN_CHECK_LAUNCH("diffaug(sums)");
    hipLaunchKernelGGL(da_fold_k, dim3((unsigned)g.N), dim3(64), 0, st, ws, sums);
    GAN_CHECK_LAUNCH("diffaug(fold)");
    const dim3 grid((unsigned)ceil_div64((int64_t)g.H * g.W, DTPB), (unsigned)g.N);
    hipLaunchKernelGGL(da_apply_k<T>, grid, dim3(DTPB), 0, st, (const T*)in, g, (T*)out, sums, usk, shift, mode);
    GAN_CHECK_LAUNCH("diffaug(apply)");
    return GAN_OK;
}

}  // namespace

extern "C" int64_t gan_diffaug_workspace(int N) { return N > 0 ? (int64_t)N * (DA_BLOCKS + 1) : 0; }

extern "C" int gan_diffaug(const void* in, int64_t sn, int64_t sc, int64_t sh, int64_t sw, void* out, const float* usk, const int* shift, int mode,
                           int N, int C, int H, int W, float* workspace, int64_t workspace_floats, int dtype, gan_stream_t stream) {
    GAN_CHECK_ARG(N >= 0 && H > 0 && W > 0 && C >= 1 && C <= DA_MAXC, "diffaug: serves image tensors of 1..4 channels (got N=%d C=%d H=%d W=%d)", N, C, H, W);
    GAN_CHECK_ARG(mode >= 0 && mode <= 2, "diffaug: mode %d (0 = augment, 1 = its linear part, 2 = its adjoint)", mode);
    GAN_CHECK_ARG(dtype == GAN_F32 || dtype == GAN_BF16, "diffaug: bad dtype %d", dtype);
    if (N == 0) return GAN_OK;
    GAN_CHECK_ARG(in && out && usk && shift, "diffaug: null pointer");
    GAN_CHECK_ARG(N <= 65535 && (int64_t)H * W < (1LL << 31), "diffaug: tensor too large for the grid");
    GAN_CHECK_ARG(workspace != nullptr && workspace_floats >= gan_diffaug_workspace(N),
                  "diffaug: needs a workspace of gan_diffaug_workspace(N) = %lld floats", (long long)gan_diffaug_workspace(N));
    const DaGeom g{sn, sc, sh, sw, N, C, H, W};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GAN_F32) return da_launch<float>(in, g, out, usk, shift, mode, workspace, st);
    return da_launch<bf16>(in, g, out, usk, shift, mode, workspace, st);
}
