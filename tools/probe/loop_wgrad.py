import os, sys, time
sys.path.insert(0, "medical-image-analysis_amd")
import torch, mia_hip
from mia_hip import WGRAD_3S1, ops
dev = torch.device("cuda:0")
x = torch.randn(32, 512, 512, 64, device=dev).to(torch.bfloat16)
dy = torch.randn(32, 512, 512, 64, device=dev).to(torch.bfloat16)
if len(sys.argv) > 1:
    mia_hip.lib().mia_set_option(b"wgrad_dma", int(sys.argv[1]))
t0 = time.time(); n = 0
while time.time() - t0 < float(os.environ.get("LOOP_S", "8")):
    for _ in range(50):
        ops.conv_wgrad(WGRAD_3S1, x, None, dy, (64, 64, 3, 3), 64, 64)
    torch.cuda.synchronize(); n += 50
print("iters", n, "ms/iter", (time.time() - t0) / n * 1e3)
