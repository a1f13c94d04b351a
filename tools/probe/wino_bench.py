"""conv64 direct vs Winograd (option conv64_wino): canonical launch 64 -> 64 @512^2 x 32 bf16, plain / flipped / normalise-on-load."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch  # noqa: E402

import mia_hip  # noqa: E402
from mia_hip import CONV_G3S1, ops  # noqa: E402

dev = torch.device("cuda:0")
n, s, c = 32, 512, 64
x = torch.nn.functional.leaky_relu(torch.randn(n, s, s, c, device=dev), 0.01).to(torch.bfloat16)
wt = torch.randn(c, c, 3, 3, device=dev) / 24
b = torch.zeros(c, device=dev)
coefs = torch.zeros(5, n, c, device=dev)
coefs[2] = 1.0 + 0.1 * torch.randn(n, c, device=dev)
coefs[3] = 0.1 * torch.randn(n, c, device=dev)
pc = ops.PackCache()
wp, npad, kpad = pc.get(wt, mia_hip.BF16, True)
wb, npb, kpb = pc.get(wt, mia_hip.BF16, False)
for v in (0, 1, 0, 1):
    mia_hip.set_option("conv64_wino", v)
    line = f"conv64_wino={v}:"
    for name, fn in (("plain+stats", lambda: ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, c, (s, s), want_stats=True)),
                     ("dgrad", lambda: ops.conv_mma(CONV_G3S1, x, None, wb, npb, kpb, True, None, c, (s, s))),
                     ("NL+stats", lambda: ops.conv_mma(CONV_G3S1, x, None, wp, npad, kpad, False, b, c, (s, s), want_stats=True, nl=(coefs, 0.01)))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) / 20 * 1e3:6.0f} us"
    print(line, flush=True)
mia_hip.set_option("conv64_wino", 0)
