"""Does a tile kernel's speed depend on the ADDRESSES of its operands?  One big-tile conv launch (128 -> 128 @ 256 x 256 x 32, bf16: level 1
of cfg3) with input / output carved out of one arena at different byte offsets (round 5: the same train step ran 41.1 ms or 37.5 ms
of kernel time depending on where the caching allocator had put things)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch, mia_hip
from mia_hip import CONV_G3S1, ops
dev = torch.device("cuda:0")
n, h, w, c = 32, 256, 256, 128
nb = n * h * w * c * 2
arena = torch.empty(6 * (1 << 30), dtype=torch.uint8, device=dev)
base = arena.data_ptr()
print("arena base 0x%x  tensor bytes %d MiB" % (base, nb >> 20))
wt = (torch.randn(c, c, 3, 3) / 34).to(dev)
wp, npad, kpad = ops.PackCache().get(wt, mia_hip.BF16, True)
bias = torch.zeros(c, device=dev)
src = torch.randn(n, h, w, c).to(torch.bfloat16).to(dev)
def carve(off):
    t = arena[off:off + nb].view(torch.bfloat16).view(n, h, w, c)
    return t
def run(xoff, yoff, iters=10):
    x = carve(xoff); x.copy_(src)
    # conv_mma allocates its own output: emulate placement by timing with a pre-carved output through the raw call
    from mia_hip import call
    from mia_hip.ops import _p, _stream
    y = carve(yoff)
    stats = torch.empty((n, ops.conv_tiles(CONV_G3S1, h, w), c, 2), device=dev)
    f = lambda: call("mia_conv_mma", CONV_G3S1, mia_hip.BF16, _p(x), c, None, 0, _p(wp), npad, kpad, 0, _p(bias), _p(y), c, None, 0, _p(stats),
                     n, h, w, h, w, None, None, None, None, None, None, _stream())
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
G = 1 << 30
for xoff, yoff in ((0, G), (0, G + (2 << 20)), (0, G + (64 << 10)), (0, G + 4096), (2 << 20, G), (0, 2 * G), (0, G + (512 << 20)), (0, G + (1 << 20)), (4096, G + 8192),
                   (0, G + (256 << 10)), (0, G + (16 << 20)), (0, G + (128 << 20))):
    print("x at +%8d KiB, y at +%8d KiB: %.3f ms" % (xoff >> 10, yoff >> 10, run(xoff, yoff)))
