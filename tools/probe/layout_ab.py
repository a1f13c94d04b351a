"""Step time depends on WHERE the caching allocator put the activations (round 5): the same resident step ran 40.9 ms before and 37.5 ms
after a few host-fed steps had allocated their staging buffers.  Run under rocprofv3 --kernel-trace; tools/probe/layout_ab_diff.py
splits the trace at the optimizer launches and compares per-kernel averages of the two phases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
import bench
from losses.compound_losses import DiceAndCELoss
from losses.dice_loss import DiceLoss
from models.unet import UNet
from training.engine import TrainEngine
dev = torch.device("cuda:0")
channels, size, batch, dt = bench.CONFIGS["cfg3"]
torch.manual_seed(1337)
model = UNet(2, 1, 3, channels, normalization="instance", dropout_prob=0.1).to(dev)
model.set_compute_dtype(torch.bfloat16)
loss_fn = DiceAndCELoss(dice_loss=DiceLoss, dice_kwargs=dict(num_classes=2, smooth=1e-5, do_bg=True, softmax=True, batch=False, squared=False),
                        ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})
eng = TrainEngine(model, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=4000, lr_warmup_iter=250)
img, lab = bench.synth_batch(batch, size, 1337)
res = {"image": img.to(dev), "label": lab.to(dev)}
host = {"image": img.clone(), "label": lab.clone()}
def loop(b, k=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): eng.train_step(b)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k
for _ in range(4): eng.train_step(res)
print("phase 1 (steps 5-12):  %.2f ms" % loop(res))
def addrs():
    out = {}
    return out
for _ in range(5): eng.train_step(host)
torch.cuda.synchronize()
print("phase 2 (steps 18-25): %.2f ms" % loop(res))
