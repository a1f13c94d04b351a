"""Diagnostic (not a test): per-tensor gradient error of the full-width fp32 model vs the fp32 and the fp64 CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import losses_ref, unet_ref, train_ref
import test_gpu_configs as T

dev = torch.device("cuda:0")
torch.set_num_threads(16)
dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32
channels = [int(c) for c in os.environ.get("DIAG_CH", "64,128,256,512,1024").split(",")]
norm, k1 = os.environ.get("DIAG_NORM", "instance"), 3
m = T._model(dev, channels, norm, k1, dtype).train()
state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
SIZE = int(os.environ.get("DIAG_SIZE", "128")); NB = int(os.environ.get("DIAG_N", "2"))
x, y = T._batch(NB, SIZE, seed=int(os.environ.get("DIAG_SEED", "3")))

def oracle(dt):
    p = {k: (v.detach().to(dt).clone() if v.is_floating_point() else v.clone()) for k, v in state.items()}
    for v in train_ref.trainable(p).values():
        v.requires_grad_(True)
    out = unet_ref.unet_forward(p, x.to(dt), norm, True)
    loss = losses_ref.dice_and_ce(out, y, k1 - 1)
    loss.backward()
    return out.detach(), float(loss), {k: v.grad.detach() for k, v in train_ref.trainable(p).items()}

o32, l32, g32 = oracle(torch.float32)
o64, l64, g64 = oracle(torch.float64) if SIZE <= 128 else (o32.double(), l32, {k: v.double() for k, v in g32.items()})
out = m(x.to(dev))
loss = T._loss_fn(k1)(out, y.to(dev))
loss.backward()
print("logits: gpu-vs-f64 %.3e  cpu32-vs-f64 %.3e   loss gpu %.7f cpu32 %.7f f64 %.7f" % (
    float((out.detach().cpu().double() - o64).abs().max()), float((o32.double() - o64).abs().max()), loss.item(), l32, l64))
for name, p in m.named_parameters():
    ref = g64[name]
    mx = max(float(ref.abs().max()), 1e-3)
    eg = float((p.grad.cpu().double() - ref).abs().max()) / mx
    ec = float((g32[name].double() - ref).abs().max()) / mx
    rl2 = float((p.grad.cpu().double() - ref).norm() / ref.norm())
    print(f"{name:45s} gpu-vs-f64 {eg:.2e}  cpu32-vs-f64 {ec:.2e}  relL2 {rl2:.2e}")
