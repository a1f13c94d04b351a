"""Host-side cost of HostFeed.stage per part, cfg3- and cfg4-sized batches (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
from training.feed import HostFeed
dev = torch.device("cuda:0")
print("torch threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))
for name, (n, h, w) in {"cfg3": (32, 512, 512), "cfg4 native": (32, 496, 608)}.items():
    img = torch.rand(n, 1, h, w); lab = torch.randint(0, 3, (n, h, w))
    hf = HostFeed(dev)
    for _ in range(4):
        hf.stage(img, lab)
    torch.cuda.synchronize()
    def t(f, k=10):
        f(); t0 = time.perf_counter()
        for _ in range(k): f()
        return 1e3 * (time.perf_counter() - t0) / k
    s = hf.slots[0]
    print(name, "stage() total %.2f ms" % t(lambda: hf.stage(img, lab)), "| aminmax %.2f" % t(lambda: HostFeed.labels_fit_a_byte(lab)),
          "| img->pinned %.2f" % t(lambda: s.pin_img.copy_(img)), "| lab->pinned u8 %.2f" % t(lambda: s.pin_lab.copy_(lab)))
    torch.cuda.synchronize()
