import sys, torch
sys.path.insert(0, '/root/repo')
import gan_amd.functional as GF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
bad = 0
for (n, cin, h, w, cout, k, pooled) in [(32, 512, 8, 8, 512, 3, False), (32, 256, 16, 16, 512, 3, True), (32, 128, 32, 32, 128, 3, False), (32, 64, 64, 64, 128, 1, False),
                                        (32, 16, 256, 256, 16, 3, False), (32, 32, 128, 128, 64, 3, True), (64, 64, 64, 64, 64, 3, False), (32, 512, 4, 4, 512, 3, False)]:
    pad = (k - 1) // 2
    x = torch.randn(n, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gh, gw = (h // 2, w // 2) if pooled else (h, w)
    gy = torch.randn(n, cout, gh, gw, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    ref_dw, ref_db = GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad, x_slope=0.2, pooled=pooled, bias=True)
    want = gy.double().sum(dim=(0, 2, 3))
    err = float((ref_db.double() - want).abs().max() / want.abs().max())
    for i in range(200):
        dw, db = GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad, x_slope=0.2, pooled=pooled, bias=True)
        if not (torch.equal(dw, ref_dw) and torch.equal(db, ref_db)):
            bad += 1
    torch.cuda.synchronize()
    print((n, cin, h, w, cout, k, pooled), 'bias rel err', err, 'mismatches so far', bad, flush=True)
print('BAD' if bad else 'OK')
