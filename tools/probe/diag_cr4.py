import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    sys.path.insert(0, p)
import torch, mia_hip
from mia_hip import CONV_G3S1, ops
dev = torch.device("cuda:0")
n, h, w, c = 32, 512, 512, 64
g = torch.Generator().manual_seed(1)
dyb = torch.randn(n, h, w, c, generator=g).to(dev, torch.bfloat16)
y = (torch.randn(n, h, w, c, generator=g) * 1.5).to(dev, torch.bfloat16)
coefs = (torch.rand(5, n, c, generator=g) + 0.5).to(dev)
wt = (torch.randn(c, c, 3, 3, generator=g) / 24).to(dev)
pc = ops.PackCache()
wb, npad, kpad = pc.get(wt, mia_hip.BF16, False)
ref, _, _ = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w))
bad_tiles = 0
for it in range(10):
    got, _, part = ops.conv_mma(CONV_G3S1, dyb, None, wb, npad, kpad, True, None, c, (h, w), cr=(y, coefs, 0.01))
    bad = (got != ref) | torch.isnan(got.float())
    bad_tiles += int(bad.view(n, 32, 16, 32, 16, c).any(5).any(4).any(2).sum())
print(os.environ.get("MIA_HIP_LIB", "default"), "bad tiles in 10 x 32768:", bad_tiles)
