"""cfg4 host-fed loop, piece by piece (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
from training.feed import HostFeed
from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
dev = torch.device("cuda:0")
n, H0, W0 = 32, 496, 608
img = torch.rand(n, 1, H0, W0); lab = torch.randint(0, 3, (n, H0, W0))
aug = BatchedAugment(al_train_transforms("busi", elastic=True), image_size=256, do_normalize=False)
hf = HostFeed(dev)
di0, dl0 = img.to(dev), lab.to(dev)
def loop(f, k=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k
print("aug on resident      %.2f ms" % loop(lambda: aug(di0, dl0)))
print("stage only           %.2f ms" % loop(lambda: hf.stage(img, lab)))
print("stage + aug          %.2f ms" % loop(lambda: aug(*hf.stage(img, lab))))
def h2d_plain():
    a = img.to(dev); b = lab.to(dev); return aug(a, b)
print("pageable .to() + aug %.2f ms" % loop(h2d_plain))
def A():
    hf.stage(img, lab); return aug(di0, dl0)
print("stage ; aug(resident) %.2f ms" % loop(A))
ds, ls = hf.stage(img, lab)
print("aug(slot tensors)     %.2f ms" % loop(lambda: aug(ds, ls)))
w0 = aug._arena.wait_s
t = loop(lambda: aug(*hf.stage(img, lab)))
print("stage + aug again     %.2f ms; arena wait per call %.2f ms" % (t, 1e3 * (aug._arena.wait_s - w0) / 25))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): aug(*hf.stage(img, lab))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
