// Dispatcher of the thin-layer direct convolution (conv_thin_impl.h; one translation unit per mode).
#include "conv_thin_impl.h"

// Returns 1 if the shape is served by the thin kernel (and enqueues it), 0 if not, < 0 on error.
int gan_conv2d_fwd_thin(const void* x, const void* wt, void* y, const float* in_scale, const float* out_scale,
                        const float* bias, const void* noise, const float* noise_scale, float slope, int N, int H,
                        int W, int Cin, int up2, int Ho, int Wo, int Cout, int kh, int kw, int py, int px, int Kpad,
                        const ConvFuse& fu_in, hipStream_t st) {
    // a residual behind a plain (unmasked, unpooled) convolution -- the blur-enabled D block's merge on its shortcut
    // convolution -- runs in the mask mode with the residual as its own mask operand at slope 1 (a mask of ones)
    ConvFuse fu = fu_in;
    if (fu.resid && !fu.aux && !fu.pool && !fu.sc_x && !fu.dot_out) {
        fu.aux = fu.resid;
        fu.aux_slope = 1.f;
    }
    if (kh != kw || (kh != 1 && kh != 3)) return 0;
    if (py != (kh - 1) / 2 || px != py) return 0;
    if (Cin != 8 && Cin != 16 && Cin != 32 && Cin != 64) return 0;
    if (Cout > 64 || Wo < 32 || Ho < 8) return 0;
    if (Kpad != (kh * kw * Cin + 31) / 32 * 32) return 0;
    if (((uintptr_t)y % 8) != 0 || ((uintptr_t)fu.aux % 8) != 0 || ((uintptr_t)in_scale % 16) != 0) return 0;
    if (fu.dot_out) {
        // the modulated convolution's input gradient with its scale + dot pass on the accumulators (THIN_DS / THIN_DPS)
        if (!fu.dot_x || !out_scale || in_scale || noise || bias || fu.aux || fu.resid || fu.in_slope != 1.f || slope != 1.f || (Cout & 3)) return 0;
        if (((uintptr_t)fu.dot_x % 8) != 0 || ((uintptr_t)out_scale % 4) != 0 || ((uintptr_t)fu.dot_out % 4) != 0) return 0;
    }
    if (fu.sc_x) {
        // the D block's conv1 input gradient with the shortcut's input gradient formed in the kernel (THIN_DAS)
        if (fu.dot_out || !fu.aux || fu.resid || fu.pool || in_scale || out_scale || noise || bias || up2 || fu.in_slope != 1.f || slope != 1.f) return 0;
        if (kh != 3 || (Cout != 16 && Cout != 32 && !(Cout == 64 && Cin == 64)) || fu.sc_c != 2 * Cout || !fu.sc_w || (Ho & 1) || (Wo & 1)) return 0;
        if (((uintptr_t)fu.sc_x % 16) != 0 || ((uintptr_t)fu.sc_w % 16) != 0) return 0;
        if ((int64_t)N * (Ho / 2) * (Wo / 2) * fu.sc_c * 2 >= (1LL << 31)) return 0;
    }
    const bool g_stage = !fu.dot_out && (in_scale || out_scale || noise);
    const bool d_stage = fu.in_slope != 1.f || fu.pool || fu.aux || fu.gain != 1.f || fu.resid;
    if (g_stage && d_stage) return 0;  // no caller mixes them; the implicit-GEMM kernel takes every combination
    if (fu.pool && fu.aux) return 0;
    // the residual rides in the mask mode (input gradients) and in the pooled mode (the D block's residual merge)
    if (fu.resid && ((!fu.aux && !fu.pool) || ((uintptr_t)fu.resid % 8) != 0 || (Cout & 3))) return 0;
    if ((Cout & 3) && (Cout > 16 || fu.pool)) return 0;  // 2-byte stores: to-/from-RGB widths only
    // 32-bit buffer offsets
    const int64_t xb = (int64_t)N * H * W * Cin * 2;
    const int64_t yb = (int64_t)N * (fu.pool ? Ho / 2 : Ho) * (fu.pool ? Wo / 2 : Wo) * Cout * 2;
    if (xb >= (1LL << 31) || yb >= (1LL << 31) || (int64_t)N * Ho * Wo >= (1LL << 31)) return 0;
    ThinArgs a;
    a.fu = fu;
    a.x = x; a.wt = wt; a.y = y;
    a.in_scale = in_scale; a.out_scale = out_scale; a.bias = bias;
    a.noise = noise; a.noise_scale = noise_scale; a.slope = slope;
    a.N = N; a.H = H; a.W = W; a.Hl = up2 ? 2 * H : H; a.Wl = up2 ? 2 * W : W; a.up2 = up2;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.py = py; a.px = px; a.Kpad = Kpad;
    a.tiles_x = a.tiles_y = 0;
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
    a.sc_bytes = fu.sc_x ? (unsigned)((int64_t)N * (Ho / 2) * (Wo / 2) * fu.sc_c * 2) : 0u;
    int rc;
    if (fu.dot_out) rc = fu.pool ? thin_launch_dps(a, Cin, kh, st) : thin_launch_ds(a, Cin, kh, st);
    else if (fu.sc_x) rc = thin_launch_das(a, Cin, kh, st);
    else if (g_stage) rc = thin_launch_g(a, Cin, kh, st);
    else if (fu.pool && fu.resid) rc = thin_launch_dpr(a, Cin, kh, st);
    else if (fu.pool) rc = thin_launch_dp(a, Cin, kh, st);
    else if (fu.aux) rc = thin_launch_da(a, Cin, kh, st);
    else rc = thin_launch_d(a, Cin, kh, st);
    return rc == GAN_OK ? 1 : rc;
}
