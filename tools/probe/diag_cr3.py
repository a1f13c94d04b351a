import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    sys.path.insert(0, p)
import torch, mia_hip
from mia_hip import CONV_G3S1, ops
dev = torch.device("cuda:0")
for (n, h, w) in ((2, 512, 512), (32, 128, 128)):
    c = 64
    g = torch.Generator().manual_seed(7 * n + h)
    dyb = torch.randn(n, h, w, c, generator=g).to(dev, torch.bfloat16)
    y = (torch.randn(n, h, w, c, generator=g) * 1.5).to(dev, torch.bfloat16)
    coefs = (torch.rand(5, n, c, generator=g) + 0.5).to(dev)
    wt = (torch.randn(c, c, 3, 3, generator=g) / 24).to(dev)
    pc = ops.PackCache()
    wb, npad, kpad = pc.get(wt, mia_hip.BF16, False)
    ref, _, _ = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w))
    got, _, part = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w), cr=(y, coefs, 0.01))
    torch.cuda.synchronize()
    ty, tx = h // 16, w // 16
    bad = (got.float() != ref.float()) | torch.isnan(got.float())
    bt = bad.view(n, ty, 16, tx, 16, c).any(5).any(4).any(2).view(-1)      # per tile
    idx = bt.nonzero().flatten().tolist()
    its = {}
    for t in idx:
        its[t // 512] = its.get(t // 512, 0) + 1
    print(n, h, w, "tiles", bt.numel(), "bad tiles", len(idx), "by iteration", its, "first bad", idx[:10])
    # inside a bad tile: which rows / channels
    if idx:
        t = idx[0]
        im, r = divmod(t, ty * tx); a, b = divmod(r, tx)
        blk = bad[im, a * 16:(a + 1) * 16, b * 16:(b + 1) * 16]
        print(" rows with errors", blk.any(2).any(1).nonzero().flatten().tolist(), "cols", blk.any(2).any(0).nonzero().flatten().tolist(),
              "channels", blk.any(0).any(0).nonzero().flatten().tolist()[:40])
        gv = got[im, a * 16:(a + 1) * 16, b * 16:(b + 1) * 16].float(); rv = ref[im, a * 16:(a + 1) * 16, b * 16:(b + 1) * 16].float()
        e = blk.nonzero()[:5].tolist()
        print(" samples", [(tuple(i), float(gv[tuple(i)]), float(rv[tuple(i)])) for i in e])
