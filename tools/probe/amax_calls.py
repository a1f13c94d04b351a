"""How many standalone mia_amax reductions does one fp32 train step launch (the rest ride in the norm / activation passes)?"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
from losses.compound_losses import DiceAndCELoss
from mia_hip import ops
from models.unet import UNet

dev = torch.device("cuda:0")
ops.F32_SPLIT_MIN_MACS = 0
torch.manual_seed(0)
m = UNet(2, 1, 3, [32, 64, 128, 256], normalization=sys.argv[1] if len(sys.argv) > 1 else "instance", dropout_prob=0.1).to(dev).train()
x = torch.rand(2, 1, 64, 96).to(dev)
lab = torch.randint(0, 3, (2, 64, 96)).to(dev)
loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
cnt = collections.Counter()
raw = ops.call
def counting(name, *a):
    cnt[name] += 1
    return raw(name, *a)
ops.call = counting
loss_fn(m(x), lab).backward()
torch.cuda.synchronize()
for k in ("mia_amax", "mia_conv_mma", "mia_conv_mma_acc", "mia_conv_wgrad", "mia_norm_act_fwd", "mia_norm_act_bwd", "mia_norm_act_bwd_head_w"):
    print(k, cnt[k])
