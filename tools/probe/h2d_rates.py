"""H2D copy rates on this box: pageable vs pinned, default vs side stream, sizes of cfg3 / cfg4 batches (debug aid)."""
import time, torch
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
def rate(f, nbytes, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    return "%.2f ms %.1f GB/s" % (1e3 * dt, nbytes / dt / 1e9)
for shape in ((32, 1, 512, 512), (32, 1, 496, 608), (32, 496, 608)):
    for dt in (torch.float32, torch.uint8):
        src = torch.zeros(shape, dtype=dt); pin = torch.zeros(shape, dtype=dt, pin_memory=True); dst = torch.empty(shape, dtype=dt, device=dev)
        nb = src.numel() * src.element_size()
        def side_copy():
            with torch.cuda.stream(side):
                dst.copy_(pin, non_blocking=True)
        print(shape, str(dt).replace("torch.", ""), "| pageable", rate(lambda: dst.copy_(src), nb), "| pinned default-stream", rate(lambda: dst.copy_(pin, non_blocking=True), nb),
              "| pinned side-stream", rate(side_copy, nb), "| host->pinned memcpy", rate(lambda: pin.copy_(src), nb))
