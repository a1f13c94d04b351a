"""Record of a wrong turn (round 5): the experiments that tried to find what made the host-fed loop of bench.py run 13 % FASTER
than the resident one ("stage", "hoststeps", "part<N>", "q3:", "f3:", "t3:", ...), and the one that settled it ("nan:<res|host|mixed>": 40 steps,
losses printed -- the first HostFeed overwrote live activations from its side stream, the parameters went NaN, and NaN operands clock higher).
Kept as it was run; not a tool."""
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
model.set_compute_dtype(torch.bfloat16 if dt == "bf16" else torch.float32)
loss_fn = DiceAndCELoss(dice_loss=DiceLoss, dice_kwargs=dict(num_classes=2, smooth=1e-5, do_bg=True, softmax=True, batch=False, squared=False),
                        ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})
eng = TrainEngine(model, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=4000, lr_warmup_iter=250)
img, lab = bench.synth_batch(batch, size, 1337)
res = {"image": img.to(dev), "label": lab.to(dev)}
host = {"image": img.clone(), "label": lab.clone()}
def loop(b, k=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): eng.train_step(b)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k
for _ in range(3): eng.train_step(res)
import subprocess
def clk():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "Temperature (Sensor junction)", "socclk"))]
        return " | ".join(keep)[:600]
    except Exception as e:
        return repr(e)
def meas(tag):
    print(tag, " ".join("%.2f" % loop(res) for _ in range(2)), flush=True)
meas("start:")
side = torch.cuda.Stream()
pin = torch.empty(40 << 20, dtype=torch.uint8, pin_memory=True); dbuf = torch.empty(40 << 20, dtype=torch.uint8, device=dev)
main = torch.cuda.current_stream()
which = sys.argv[1] if len(sys.argv) > 1 else "h2d"
if which == "start":
    sys.exit(0)
if which == "synthetic":
    import ctypes
    libp = ctypes.CDLL(os.path.join(ROOT, "tools", "ab", "mfma_power.so"))
    libp.mfma_power_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    units = 1 << 26
    buf = torch.randint(0, 2 ** 31 - 1, (units * 4,), dtype=torch.int32, device=dev); buf &= 0xBFFFBFFF - (1 << 32)
    outp = torch.zeros(256, device=dev); clocks = torch.zeros(128, dtype=torch.int64, device=dev)
    stp = torch.cuda.current_stream().cuda_stream
    def point(shape, ldsr, hbm, iters=400000):
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); libp.mfma_power_run(shape, ldsr, hbm, 0, buf.data_ptr(), outp.data_ptr(), iters, units, clocks.data_ptr(), 512, stp); e1.record()
            torch.cuda.synchronize(); ms = e0.elapsed_time(e1)
        tf = 8 * 16 * 16 * 32 * 2 * iters * 512 * 4 / (ms * 1e-3) / 1e12
        c = clocks.cpu(); ghz = float((c[0::2].double() / (c[1::2].double() / 100e6)).mean()) / 1e9
        return "%.0f TF %.3f GHz" % (tf, ghz)
    def sweep(tag):
        print(tag, "| MFMA only:", point(0, 0, 0), "| +4 LDS:", point(0, 4, 0), "| 512 flop/B +4 LDS:", point(0, 4, 4), "| 256 flop/B +4 LDS:", point(0, 4, 2), flush=True)
    sweep("before")
    meas("step before:")
    from training.feed import HostFeed
    hf = HostFeed(dev)
    eng.train_step(res); hf.stage(host["image"], host["label"]); eng.train_step(res)
    meas("step after the trigger:")
    sweep("after ")
    meas("step again:")
    sys.exit(0)
if which.startswith("p3:"):
    fp, fd, sd, ev_, two = [int(c) for c in which[3:]]
    image = host["image"]
    nbytes = image.numel() * 4
    old_pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True); old_pin.fill_(1)
    old_dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    old_pin2 = torch.empty(nbytes // 4, dtype=torch.uint8, pin_memory=True); old_dev2 = torch.empty(nbytes // 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    meas("start:")
    eng.train_step(res)
    pin_ = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) if fp else old_pin
    if fp: pin_.fill_(2)
    dev_ = torch.empty(nbytes, dtype=torch.uint8, device=dev) if fd else old_dev
    strm = side if sd else main
    with torch.cuda.stream(strm):
        dev_.copy_(pin_, non_blocking=True)
        if two: old_dev2.copy_(old_pin2, non_blocking=True)
        if ev_:
            e_ = torch.cuda.Event(); e_.record(strm)
    eng.train_step(res)
    meas("after fresh_pinned=%d fresh_dev=%d side=%d event=%d two_copies=%d:" % (fp, fd, sd, ev_, two))
    sys.exit(0)
if which.startswith("q3:"):
    import ctypes
    from mia_hip import lib
    from training.feed import HostFeed, _fit
    hostcp, img_copy, lab_copy, ev_, newfeed = [int(c) for c in which[3:]]
    image, label = host["image"], host["label"]
    meas("start:")
    hf = HostFeed(dev) if newfeed else None
    cs = hf.copy_stream if newfeed else side
    eng.train_step(res)
    skip = os.environ.get("SKIP", "")
    pin_img = torch.empty(image.shape, dtype=torch.float32, pin_memory=True)
    if "pin8" not in skip: pin_lab8 = torch.empty(label.shape, dtype=torch.uint8, pin_memory=True)
    if hostcp:
        if "hostcopy" not in skip: lib().mia_host_copy(ctypes.c_void_p(pin_img.data_ptr()), ctypes.c_void_p(image.data_ptr()), ctypes.c_int64(image.numel() * 4), 0)
        if "narrow" not in skip and "pin8" not in skip: lib().mia_host_narrow_labels(ctypes.c_void_p(label.data_ptr()), ctypes.c_void_p(pin_lab8.data_ptr()), ctypes.c_int64(label.numel()), 0)
    dev_img = torch.empty(image.shape, dtype=torch.float32, device=dev)
    if "devlab" not in skip:
        dev_lab_raw = torch.empty(label.shape, dtype=torch.uint8, device=dev)
        dev_lab = torch.empty(label.shape, dtype=torch.int64, device=dev)
    evt = torch.cuda.Event()
    with torch.cuda.stream(cs):
        if img_copy: dev_img.copy_(pin_img, non_blocking=True)
        if lab_copy: dev_lab_raw.copy_(pin_lab8, non_blocking=True)
        if ev_: evt.record(cs)
    eng.train_step(res)
    meas("after SKIP=%s hostcopy=%d img_copy=%d lab_copy=%d event=%d HostFeed_stream=%d:" % (skip, hostcp, img_copy, lab_copy, ev_, newfeed))
    sys.exit(0)
if which.startswith("t3:"):
    delay_ms, mb, sd = [int(c) for c in which[3:].split(",")]
    old_pin = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True); old_pin.fill_(1)
    old_dev = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    meas("start:")
    eng.train_step(res)
    time.sleep(delay_ms * 1e-3)
    with torch.cuda.stream(side if sd else main):
        old_dev.copy_(old_pin, non_blocking=True)
    eng.train_step(res)
    meas("after a %d MB pinned H2D issued %d ms into the step on the %s stream:" % (mb, delay_ms, "side" if sd else "main"))
    sys.exit(0)
if which.startswith("f3:"):
    import ctypes
    from mia_hip import lib
    fill, prealloc = [int(c) for c in which[3:]]
    image = host["image"]
    pre = os.environ.get("PRE", "zeros,pin,dev")
    zeros = torch.zeros_like(image) if "zeros" in pre else None; rnd = torch.randn_like(image) * 1e3 if "zeros" in pre else None
    pre_pin = torch.empty(image.shape, dtype=torch.float32, pin_memory=True) if "pin" in pre else None
    pre_dev = torch.empty(image.shape, dtype=torch.float32, device=dev) if "dev" in pre else None
    torch.cuda.synchronize()
    meas("start:")
    eng.train_step(res)
    pin_img = pre_pin if prealloc else torch.empty(image.shape, dtype=torch.float32, pin_memory=True)
    hc = lambda srct: lib().mia_host_copy(ctypes.c_void_p(pin_img.data_ptr()), ctypes.c_void_p(srct.data_ptr()), ctypes.c_int64(srct.numel() * 4), 0)
    if fill == 1: hc(image)
    elif fill == 2: pin_img.copy_(image)
    elif fill == 3: hc(zeros)
    elif fill == 4: hc(rnd)
    elif fill == 5: pin_img.copy_(zeros)
    extra = os.environ.get("EXTRA", "")
    label = host["label"]
    if "narrow" in extra:
        pin_lab8 = torch.empty(label.shape, dtype=torch.uint8, pin_memory=True)
        lib().mia_host_narrow_labels(ctypes.c_void_p(label.data_ptr()), ctypes.c_void_p(pin_lab8.data_ptr()), ctypes.c_int64(label.numel()), 0)
    if "pin8" in extra:
        pin_lab8 = torch.empty(label.shape, dtype=torch.uint8, pin_memory=True)
    if "dev64" in extra:
        dev_lab = torch.empty(label.shape, dtype=torch.int64, device=dev); dev_lab_raw = torch.empty(label.shape, dtype=torch.uint8, device=dev)
    if "busy" in extra:
        t_ = time.perf_counter()
        while time.perf_counter() - t_ < 0.008: pass
    dev_img = pre_dev if prealloc else torch.empty(image.shape, dtype=torch.float32, device=dev)
    with torch.cuda.stream(side):
        dev_img.copy_(pin_img, non_blocking=True)
    eng.train_step(res)
    meas("after PRE=%s EXTRA=%s fill=%s prealloc=%d:" % (pre, extra, ["none", "mia_host_copy(image)", "torch copy_(image)", "mia_host_copy(zeros)", "mia_host_copy(randn*1e3)", "torch copy_(zeros)"][fill], prealloc))
    sys.exit(0)
if which.startswith("nan:"):
    mode = which[4:]
    losses = []
    for i in range(40):
        l = eng.train_step(host if (mode == "host" or (mode == "mixed" and i < 5)) else res)
        losses.append(l)
    torch.cuda.synchronize()
    print(mode, " ".join("%.4f" % l.item() for l in losses))
    print("params finite:", bool(torch.isfinite(eng.optimizer.flat_param).all()), "grad finite:", bool(torch.isfinite(eng.optimizer.flat_grad).all()))
    sys.exit(0)
if which.startswith("loss:"):
    trig = int(which[5:])
    import ctypes
    from mia_hip import lib
    image = host["image"]
    losses = []
    for i in range(24):
        if trig and i == 8:
            pin_img = torch.empty(image.shape, dtype=torch.float32, pin_memory=True)
            lib().mia_host_copy(ctypes.c_void_p(pin_img.data_ptr()), ctypes.c_void_p(image.data_ptr()), ctypes.c_int64(image.numel() * 4), 0)
            dev_img = torch.empty(image.shape, dtype=torch.float32, device=dev)
            with torch.cuda.stream(side):
                dev_img.copy_(pin_img, non_blocking=True)
        losses.append(eng.train_step(res))
    torch.cuda.synchronize()
    print("trigger=%d losses:" % trig, " ".join("%.6f" % l.item() for l in losses))
    meas("speed:")
    import hashlib
    print("param hash", hashlib.md5(eng.optimizer.flat_param.detach().cpu().numpy().tobytes()).hexdigest())
    sys.exit(0)
keep = []
for rep in range(1, 5):
    if which == "sidewaits":       # the SIDE stream waits for an event of the busy main stream, then does a tiny kernel
        eng.train_step(res)
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            dbuf[:1024].zero_()
        eng.train_step(res)
    elif which == "evsync":        # host blocks on an event while the GPU is busy
        eng.train_step(res); ev = torch.cuda.Event(); ev.record(main); eng.train_step(res); ev.synchronize()
    elif which == "hostthreads":   # the library's host helpers (plain std::threads) while the GPU is busy
        import ctypes
        from mia_hip import lib
        eng.train_step(res)
        lib().mia_host_copy(ctypes.c_void_p(pin.data_ptr()), ctypes.c_void_p(host["image"].data_ptr()), ctypes.c_int64(32 << 20), 0)
        eng.train_step(res)
    elif which == "stage":         # HostFeed.stage alone (result unused) between resident steps, no sync
        from training.feed import HostFeed
        if rep == 1: hf = HostFeed(dev)
        eng.train_step(res); hf.stage(host["image"], host["label"]); eng.train_step(res)
    elif which == "pinalloc":      # a pinned host allocation while the GPU is busy
        eng.train_step(res); keep.append(torch.empty((40 << 20) + rep * 4096, dtype=torch.uint8, pin_memory=True)); eng.train_step(res)
    elif which == "devalloc":      # a NEW device segment (hipMalloc) while the GPU is busy
        eng.train_step(res); keep.append(torch.empty((300 << 20) + rep * (2 << 20), dtype=torch.uint8, device=dev)); eng.train_step(res)
    elif which == "pinalloc_idle":
        torch.cuda.synchronize(); keep.append(torch.empty((40 << 20) + rep * 4096, dtype=torch.uint8, pin_memory=True))
    elif which == "h2d_fresh_pinned":   # H2D from a pinned buffer allocated while idle, copy issued while busy
        torch.cuda.synchronize(); p2 = torch.empty((40 << 20), dtype=torch.uint8, pin_memory=True); keep.append(p2)
        eng.train_step(res); dbuf.copy_(p2, non_blocking=True); eng.train_step(res)
    elif which.startswith("part"):  # HostFeed.stage cut short after part N (1: host copies, 2: + device buffers, 3: + side-stream H2D, 4: + main waits, 5: + widen)
        import ctypes
        from mia_hip import lib, ops as _ops
        from mia_hip.ops import _c_i64, _p, call
        from training.feed import HostFeed, _fit
        N = int(which[4:])
        if rep == 1: hf = HostFeed(dev)
        eng.train_step(res)
        image, label = host["image"], host["label"]
        sl = hf.slots[hf.i % 3]; hf.i += 1
        sl.pin_img = _fit(sl.pin_img, image.shape, torch.float32, pin_memory=True)
        lib().mia_host_copy(ctypes.c_void_p(sl.pin_img.data_ptr()), ctypes.c_void_p(image.data_ptr()), ctypes.c_int64(image.numel() * 4), 0)
        sl.pin_lab8 = _fit(sl.pin_lab8, label.shape, torch.uint8, pin_memory=True)
        lib().mia_host_narrow_labels(ctypes.c_void_p(label.data_ptr()), ctypes.c_void_p(sl.pin_lab8.data_ptr()), ctypes.c_int64(label.numel()), 0)
        if N >= 2:
            sl.dev_img = _fit(sl.dev_img, image.shape, torch.float32, device=dev); sl.dev_lab_raw = _fit(sl.dev_lab_raw, label.shape, torch.uint8, device=dev)
            sl.dev_lab = _fit(sl.dev_lab, label.shape, torch.int64, device=dev)
        if N >= 3:
            with torch.cuda.stream(hf.copy_stream):
                sl.dev_img.copy_(sl.pin_img, non_blocking=True); sl.dev_lab_raw.copy_(sl.pin_lab8, non_blocking=True); sl.h2d_done.record(hf.copy_stream)
        if N >= 4:
            main.wait_event(sl.h2d_done)
        if N >= 5:
            call("mia_widen_u8_i64", _p(sl.dev_lab_raw), _p(sl.dev_lab), _c_i64(label.numel()), _ops._stream())
        eng.train_step(res)
    elif which == "hoststeps":     # consecutive host-fed steps
        eng.train_step(host); eng.train_step(host)
    meas("after %d x %s:" % (rep, which))
sys.exit(0)
