#!/usr/bin/env python
"""Host-side cost of one BatchedAugment call (cProfile): python tools/prof_augment_host.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

from transforms.gpu_pipeline import BatchedAugment, al_train_transforms  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
aug = BatchedAugment(al_train_transforms("busi", elastic=True), image_size=256)
img = torch.rand(32, 1, 496, 608, device=dev)
lab = torch.randint(0, 3, (32, 496, 608), device=dev)
for _ in range(5):
    aug(img, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    aug(img, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / 20:.3f} ms / batch; with device drain {1e3 * (t2 - t0) / 20:.3f} ms / batch")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    aug(img, lab)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
