import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    sys.path.insert(0, p)
import torch, mia_hip
from mia_hip import CONV_G3S1, ops
dev = torch.device("cuda:0")
for (n, h, w) in ((2, 128, 128), (2, 512, 512), (3, 512, 256), (32, 128, 128)):
    c = 64
    g = torch.Generator().manual_seed(7 * n + h)
    dyb = torch.randn(n, h, w, c, generator=g).to(dev, torch.bfloat16)
    y = (torch.randn(n, h, w, c, generator=g) * 1.5).to(dev, torch.bfloat16)
    coefs = torch.zeros(5, n, c)
    coefs[0] = torch.rand(n, c, generator=g) + 0.5
    coefs[1] = torch.randn(n, c, generator=g) * 0.3
    coefs[2] = torch.randn(n, c, generator=g)
    coefs[3] = torch.randn(n, c, generator=g) * 0.7
    coefs = coefs.to(dev)
    wt = (torch.randn(c, c, 3, 3, generator=g) / 24).to(dev)
    pc = ops.PackCache()
    wb, npad, kpad = pc.get(wt, mia_hip.BF16, False)
    ref, _, _ = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w))
    got, _, part = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w), cr=(y, coefs, 0.01))
    torch.cuda.synchronize()
    eq = torch.equal(got, ref)
    fin = bool(torch.isfinite(part).all())
    dz, yf, cf = got.double(), y.double(), coefs.double()
    u = cf[2][:, None, None, :] * yf + cf[3][:, None, None, :]
    gg = torch.where(u > 0, dz, dz * 0.01)
    xhat = cf[0][:, None, None, :] * yf + cf[1][:, None, None, :]
    want1, want2 = gg.sum((1, 2)), (gg * xhat).sum((1, 2))
    p = part.double().sum(1)
    e1 = float((p[..., 0] - want1).abs().max()); e2 = float((p[..., 1] - want2).abs().max())
    # per-tile check
    th = 16
    ty, tx = (h + 15) // 16, (w + 15) // 16
    pt = part.double().view(n, ty, tx, c, 2)
    gt = gg.view(n, ty, 16, tx, 16, c).sum((2, 4))
    bad = ((pt[..., 0] - gt).abs() > 1e-2 * gt.abs().max()).nonzero()
    print(n, h, w, "dz equal", eq, "partials finite", fin, "err", e1, e2, "bad tiles", bad.shape[0], bad[:6].tolist(), flush=True)
