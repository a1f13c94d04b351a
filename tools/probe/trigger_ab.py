"""Resident cfg3 steps, the speed-up trigger (a host-fed stage() while the GPU is busy), resident steps again -- for rocprofv3 --pmc runs
(tools/r5_trigger_pmc.sh): per-kernel counters of the two phases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
import bench
from losses.compound_losses import DiceAndCELoss
from losses.dice_loss import DiceLoss
from models.unet import UNet
from training.engine import TrainEngine
from training.feed import HostFeed
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
for _ in range(6): eng.train_step(res)       # steps 0-5 (phase 1 = steps 2-5)
hf = HostFeed(dev)
eng.train_step(res); hf.stage(img, lab); eng.train_step(res)   # steps 6, 7 + the trigger
for _ in range(4): eng.train_step(res)       # steps 8-11 (phase 2)
torch.cuda.synchronize()
