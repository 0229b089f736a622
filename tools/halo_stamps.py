"""Reads the s_memtime stamps of a diagnostic build of conv_wide_halo.hip (make -C gan_amd/csrc clean all
EXTRA=-DGAN_HALO_STAMPS; rebuild without it afterwards): python tools/halo_stamps.py N Cin H W Cout [a|ar]   (a: mask stage, ar: mask + residual)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan_amd.functional as GF
from gan_amd import _lib

n, cin, h, w, cout = map(int, sys.argv[1:6])
dev = torch.device("cuda:0")
x = torch.randn(n, cin, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
wgt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
mode = sys.argv[6] if len(sys.argv) > 6 else ""
aux = torch.randn(n, cout, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) if mode else None
res = torch.randn(n, cout, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) if mode == "ar" else None
for _ in range(20):
    GF.raw_conv2d(x, wgt, 1, 1, aux=aux, aux_slope=0.2 if mode else 1.0, resid=res)
torch.cuda.synchronize()
lib = _lib.load()
waves = 4 if (h % 16 == 0 and w % 16 == 0) else 8  # mirror of the launch selection
nb = min(1024, n * h * w // (32 * waves))
buf = np.zeros(nb * waves * 16, dtype=np.uint64)
lib.gan_debug_halo_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.gan_debug_halo_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(nb, waves, 16).astype(np.int64)
names = {(0, 1): "prologue (decode + first DMA issue)", (1, 9): "K loop", (9, 10): "epilogue", (0, 10): "kernel (wave lifetime)",
         (10, 13): "epilogue: stores drained",
         (3, 4): "step: vmcnt wait", (4, 5): "step: barrier", (5, 6): "step: first-half reads + DMA issue", (6, 7): "step: second-half reads + 16 MFMA (h=0)",
         (7, 8): "step: 16 MFMA (h=1)", (3, 8): "step total"}
for (a, b), name in names.items():
    d = st[:, :, b] - st[:, :, a]
    print(f"{name:45s} median {np.median(d):9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
steps = cin // 64 * 9
print("K loop / step", np.median(st[:, :, 9] - st[:, :, 1]) / steps)

