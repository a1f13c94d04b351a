"""Round 5 finding: one pinned H2D copy on a side stream WHILE the GPU is busy makes every later matrix-bound kernel of the process ~10 %
faster (cfg3 step 42.4 -> 37.7 ms).  Is it the shader clock?  The synthetic MFMA loop of mfma_power.hip (sustained TFLOP/s and in-kernel
clock from s_memtime / s_memrealtime) before and after that trigger, in one process."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd")]
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ab", "mfma_power.so"))
lib.mfma_power_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
units = 1 << 26
buf = torch.randint(0, 2 ** 31 - 1, (units * 4,), dtype=torch.int32, device=dev)
buf &= 0xBFFFBFFF - (1 << 32)
out = torch.zeros(256, device=dev); clocks = torch.zeros(128, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def point(shape, ldsr, hbm, iters=400000):
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = lib.mfma_power_run(shape, ldsr, hbm, 0, buf.data_ptr(), out.data_ptr(), iters, units, clocks.data_ptr(), 512, st); e1.record()
        torch.cuda.synchronize(); ms = e0.elapsed_time(e1)
    tf = 8 * 16 * 16 * 32 * 2 * iters * 512 * 4 / (ms * 1e-3) / 1e12
    c = clocks.cpu(); ghz = float((c[0::2].double() / (c[1::2].double() / 100e6)).mean()) / 1e9
    return "%.0f TFLOP/s at %.3f GHz" % (tf, ghz)
def sweep(tag):
    print(tag, "| MFMA only:", point(0, 0, 0), "| + 4 LDS reads:", point(0, 4, 0), "| 512 flop/B + 4 LDS reads:", point(0, 4, 4), flush=True)
sweep("fresh process      ")
sweep("again              ")
# the trigger: a pinned H2D copy on a side stream while the matrix loop is running
side = torch.cuda.Stream()
pin = torch.empty(40 << 20, dtype=torch.uint8, pin_memory=True); pin.fill_(3)
dst = torch.empty(40 << 20, dtype=torch.uint8, device=dev)
lib.mfma_power_run(0, 4, 4, 0, buf.data_ptr(), out.data_ptr(), 400000, units, clocks.data_ptr(), 512, st)
with torch.cuda.stream(side):
    dst.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
sweep("after the side copy")
sweep("again              ")
