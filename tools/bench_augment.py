#!/usr/bin/env python
"""Throughput of the GPU augmentation hand-off (transforms/gpu_pipeline.py): al_train's FUGC pipeline on a batch of
native-resolution images, then JointResize to the training size.  python tools/bench_augment.py [--batch 32 --size 512]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--h0", type=int, default=336)   # FUGC native resolution
    ap.add_argument("--w0", type=int, default=544)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
    dev = torch.device("cuda:0")
    torch.manual_seed(1337)
    aug = BatchedAugment(al_train_transforms("fugc"), image_size=a.size, do_normalize=False)
    img = torch.rand(a.batch, 1, a.h0, a.w0, device=dev)
    lab = torch.randint(0, 3, (a.batch, a.h0, a.w0), device=dev)
    out = aug(img, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = aug(img, lab)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    inb = a.batch * a.h0 * a.w0 * (4 + 8)
    outb = a.batch * a.size * a.size * (4 + 8)
    print(f"augment {a.batch} x {a.h0}x{a.w0} -> {a.size}x{a.size}: {ms:.3f} ms/batch = {a.batch / ms * 1e3:.0f} img/s "
          f"(in {inb / 1e6:.0f} MB + out {outb / 1e6:.0f} MB per batch); out image {tuple(out['image'].shape)} label {tuple(out['label'].shape)}")


if __name__ == "__main__":
    main()
