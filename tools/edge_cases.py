import sys, os, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/medical-image-analysis_amd")
import torch
from models.unet import UNet
from losses.compound_losses import DiceAndCELoss
from training.engine import TrainEngine
from oracle import unet_ref
dev = torch.device("cuda:0")
loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
def check(name, channels, n, h, w, dtype, norm="instance", steps=2):
    torch.manual_seed(0)
    m = UNet(2, 1, 3, channels, normalization=norm, dropout_prob=0.1).to(dev)
    m.set_compute_dtype(dtype)
    x = torch.rand(n, 1, h, w, device=dev); y = torch.randint(0, 3, (n, h, w), device=dev)
    eng = TrainEngine(m, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=10, lr_warmup_iter=1)
    ls = [eng.train_step({"image": x, "label": y}).item() for _ in range(steps)]
    m.eval()
    with torch.no_grad():
        out = m(x)
    ok = all(math.isfinite(v) for v in ls) and bool(torch.isfinite(out).all())
    # eval logits vs oracle on the first image (fp32 only)
    extra = ""
    if dtype == torch.float32:
        p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref = unet_ref.unet_forward(p, x[:1].cpu(), normalization=norm, training=False)
        extra = f" max|dlogit|={float((out[:1].cpu()-ref).abs().max()):.2e}"
    print(f"{name:40s} losses {['%.4f'%v for v in ls]} finite={ok}{extra}", flush=True)
check("bs1 64..1024 512x512 bf16", [64,128,256,512,1024], 1, 512, 512, torch.bfloat16)
check("bs2 64..512 1024x1024 bf16", [64,128,256,512], 2, 1024, 1024, torch.bfloat16)
check("bs3 32..256 336x544 (FUGC native) f32", [32,64,128,256], 3, 336, 544, torch.float32)
check("bs3 32..256 336x544 batch norm bf16", [32,64,128,256], 3, 336, 544, torch.bfloat16, "batch")
check("bs5 48,96,192 80x112 f32", [48,96,192], 5, 80, 112, torch.float32)
check("bs2 16,32,64 128x128 f32 batch", [16,32,64], 2, 128, 128, torch.float32, "batch")
check("bs1 64,128 2048x2048 bf16", [64,128], 1, 2048, 2048, torch.bfloat16)
