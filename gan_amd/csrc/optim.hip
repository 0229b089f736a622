// Optimizer step over a flat parameter arena (gfx950): Adam / AdamW for ALL parameters of a network in one launch.
//
// The reference builds `getattr(torch.optim, cfg.optimizer)(model.parameters(), lr, **kwargs)` (train_celeba.py:75-77);
// with Adam that is, per step and per parameter tensor, a handful of tiny launches (or, with capturable=True for
// graph capture, one launch per tensor for each of the bias-correction scalars: ~600 launches per iteration for the
// two StyleGAN2 networks).  Parameters, gradients and both moments live in flat fp32 arenas here
// (gan_amd/utils.py: FlatArena), so the whole update is one streaming pass: 16 B read + 12 B written per element.
//
// The step count lives on the device (a float the kernel pair below advances), so the launch can be captured in a
// HIP graph and replayed.
#include "common.h"

namespace {

constexpr int ATPB = 256;

__global__ void step_inc_k(float* step) { step[0] += 1.f; }

// p, g, m, v: [n] fp32, 16-byte aligned.  step[0] = t >= 1 (already advanced for this update).
// torch.optim.Adam (amsgrad = False, maximize = False):
//   g' = g + wd * p (coupled, Adam)           or  p *= 1 - lr * wd (decoupled, AdamW)
//   m += (g' - m) * (1 - b1);  v = v * b2 + g'^2 * (1 - b2)
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void __launch_bounds__(ATPB) adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t n4, int64_t n, const float* __restrict__ step, float lr,
                                               float b1, float b2, float eps, float wd, int decoupled) {
    __shared__ float sh[2];
    if (threadIdx.x == 0) {
        const double t = (double)step[0];
        const double bc1 = 1.0 - pow((double)b1, t);
        const double bc2 = 1.0 - pow((double)b2, t);
        sh[0] = (float)((double)lr / bc1);
        sh[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = sh[0], bc2_sqrt = sh[1];
    const float decay = decoupled ? 1.f - lr * wd : 1.f;
    const float l2 = decoupled ? 0.f : wd;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp *= decay;
        gg = fmaf(l2, pp, gg);
        mm = fmaf(gg - mm, 1.f - b1, mm);
        vv = fmaf(gg * gg, 1.f - b2, vv * b2);
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = fmaf(-step_size, mm / denom, pp);
    };
    const int64_t i = (int64_t)blockIdx.x * ATPB + threadIdx.x;
    if (i < n4) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        upd(pp.x, gg.x, mm.x, vv.x);
        upd(pp.y, gg.y, mm.y, vv.y);
        upd(pp.z, gg.z, mm.z, vv.z);
        upd(pp.w, gg.w, mm.w, vv.w);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    } else {
        const int64_t t = n4 * 4 + (i - n4);
        if (t < n) {
            float pp = p[t], mm = m[t], vv = v[t];
            upd(pp, g[t], mm, vv);
            p[t] = pp; m[t] = mm; v[t] = vv;
        }
    }
}

}  // namespace

extern "C" int gan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float* step, int advance, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int decoupled, gan_stream_t stream) {
    GAN_CHECK_ARG(p && g && m && v && step, "adam_step: null pointer");
    GAN_CHECK_ARG(n >= 0, "adam_step: negative size");
    GAN_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                  "adam_step: arenas must be 16-byte aligned");
    GAN_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f && weight_decay >= 0.f,
                  "adam_step: bad hyper-parameters (lr %g, betas %g %g, eps %g, weight_decay %g)", lr, beta1, beta2, eps, weight_decay);
    hipStream_t st = (hipStream_t)stream;
    if (advance) {
        hipLaunchKernelGGL(step_inc_k, dim3(1), dim3(1), 0, st, step);
        GAN_CHECK_LAUNCH("adam_step(step)");
    }
    if (n == 0) return GAN_OK;
    const int64_t n4 = n / 4;
    const int64_t threads = n4 + (n - n4 * 4);
    const int64_t blocks = ceil_div64(threads, ATPB);
    GAN_CHECK_ARG(blocks < (1LL << 31), "adam_step: arena too large");
    hipLaunchKernelGGL(adam_k, dim3((unsigned)blocks), dim3(ATPB), 0, st, p, g, m, v, n4, n, step, lr, beta1, beta2, eps, weight_decay,
                       decoupled);
    GAN_CHECK_LAUNCH("adam_step");
    return GAN_OK;
}
