#!/usr/bin/env python
"""Headline benchmark: training images/sec of the UNet-2D hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one full train step (lr schedule, forward, Dice+CE, backward, gradient all-reduce when
N>1, clip-by-global-norm 10, Adam) on one synthetic minibatch that is resident in HBM before the
timed region.  Workload (BASELINE.json metric / configs[2]): UNet channels [64,128,256,512,1024],
512x512 1-channel, batch 32 per GPU, bf16 activations / MFMA operands with fp32 accumulation,
statistics, parameters, logits and loss.  Weak scaling: per-GPU batch is fixed.

One JSON line on rank 0 with `roofline` = the canonical full-resolution C0 -> C0 PlainBlock: conv launch +
statistics finalize + normalise/LeakyReLU apply, each timed live with HIP events on the launch stream, priced against its
algorithmic bytes (read x once, write z once); the conv launch alone is the sub-record `roofline.conv`.  `cpu_baseline` =
the oracle's train step (oracle/train_ref.py, a port of the reference loop validated against the reference) on the host
cores: 2 warm-ups, median of 5 (SURVEY 8d protocol).  `parity` = the benchmarked widths and dtype against the fp32 oracle.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "medical-image-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

CONFIGS = {
    # name: (channels_list, size, per-GPU batch, compute dtype)
    "cfg1": ([16, 32, 64], 128, 4, "f32"),
    "cfg2": ([64, 128, 256, 512, 1024], 256, 32, "f32"),
    "cfg3": ([64, 128, 256, 512, 1024], 512, 32, "bf16"),
    "cfg4": ([32, 64, 128, 256, 512], 256, 32, "f32"),
    "cfg5": ([96, 192, 384, 768, 1536, 3072], 768, 16, "bf16"),
}
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense, MI355X_MICROARCH.md
# fp32 tensors with split-f16 products (library option f32_split, the default): every fp32 multiply-add is FOUR f16 products on the
# matrix cores (csrc/common.h SplitF16; THREE in the convs under the default value 2: h H + h L + l H on planes), so the roof for
# fp32-equivalent FLOPs of the conv launches the line prices is the dense f16 peak (2500) / products -- pricing them against the
# fp32-MFMA peak (157.3) gave fractions above 1
PEAK_F32_SPLIT_TFLOPS = {1: 2500.0 / 4, 2: 2500.0 / 3}
PEAK_HBM_GBS = 8000.0


def synth_batch(b, s, seed, k1=3):
    """Images U[0,1); FUGC-shaped masks: two disjoint filled ellipses (classes 1, 2), background 0."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(b, 1, s, s, generator=g)
    yy, xx = torch.meshgrid(torch.arange(s, dtype=torch.float32), torch.arange(s, dtype=torch.float32), indexing="ij")
    lab = torch.zeros(b, s, s, dtype=torch.long)
    for i in range(b):
        r = torch.rand(8, generator=g)
        for cls, (cx, cy, ax, ay) in enumerate([(0.3 + 0.1 * r[0], 0.35 + 0.3 * r[1], 0.10 + 0.06 * r[2], 0.14 + 0.08 * r[3]),
                                                (0.7 - 0.1 * r[4], 0.35 + 0.3 * r[5], 0.10 + 0.06 * r[6], 0.14 + 0.08 * r[7])], start=1):
            m = ((xx / s - cx) / ax) ** 2 + ((yy / s - cy) / ay) ** 2 <= 1.0
            lab[i][m] = min(cls, k1 - 1)
    return img, lab


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box exposes all host
    cores but grants a share; oversubscribing PyTorch's intra-op pool makes the CPU baseline slower, not faster)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(channels, size, budget_s=30.0):
    """Oracle train step (port of al_trainer.py:1350-1381) on the host cores: SURVEY 8(d) protocol -- 2 warm-ups, median of
    >= 5 timed steps, img/s = B / median -- at batch 1 (flagged: the GPU runs batch 32), bounded to ~budget_s of CPU work."""
    import statistics
    from oracle import train_ref, unet_ref
    torch.set_num_threads(usable_cores())
    torch.manual_seed(1337)
    p = unet_ref.init_params(1, 3, channels, "instance")
    opt = train_ref.make_optimizer(p, "adam", weight_decay=5e-4)
    bs = 1
    img, lab = synth_batch(bs, size, 1337)
    cores = torch.get_num_threads()
    t0 = time.time()
    train_ref.train_step(p, opt, img, lab, 2, "instance", lr=1e-3)  # warm-up 1 (also sizes the sample)
    first = time.time() - t0
    train_ref.train_step(p, opt, img, lab, 2, "instance", lr=1e-3)  # warm-up 2
    steps = 5 if first * 7 <= budget_s else max(1, int(budget_s / max(first, 1e-3)) - 2)
    times = []
    for _ in range(steps):
        t0 = time.time()
        train_ref.train_step(p, opt, img, lab, 2, "instance", lr=1e-3)
        times.append(time.time() - t0)
    med = statistics.median(times)
    return {"value": bs / med, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"median of {steps} timed train steps after 2 warm-ups of the same UNet at {size}x{size}, batch {bs} "
                      f"(the GPU figure is batch 32), fp32, PyTorch-CPU"}


def parity_gate(dev):
    """BASELINE.json configs[0] ("UNet-tiny 16 base ch, 3 levels, 128x128, bs 4, reference CPU path"): the HIP fp32
    forward + Dice/CE against the oracle on identical weights and inputs.  Reported, not timed."""
    from losses.compound_losses import DiceAndCELoss
    from models.unet import UNet
    from oracle import losses_ref, unet_ref
    torch.manual_seed(1337)
    model = UNet(2, 1, 3, [16, 32, 64], normalization="instance", dropout_prob=None)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    img, lab = synth_batch(4, 128, 1337)
    with torch.no_grad():
        out = model(img.to(dev))
        loss = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True))(out, lab.to(dev)).item()
        ref = unet_ref.unet_forward(params, img, "instance", False)
        ref_loss = losses_ref.dice_and_ce(ref, lab, 2).item()
    out = out.float().cpu()
    pg, pc = out.softmax(1).argmax(1), ref.softmax(1).argmax(1)
    dices = [losses_ref.hard_dice(pg == k, pc == k) if (pc == k).any() else 1.0 for k in range(3)]
    return {"config": "UNet [16,32,64] 128x128 bs4 fp32 (BASELINE configs[0]) vs CPU oracle", "max_abs_logit_diff": float((out - ref).abs().max()),
            "loss_diff": abs(loss - ref_loss), "label_map_mismatch_px": int((pg != pc).sum()), "hard_dice_gpu_vs_cpu_labelmaps": min(dices)}


def parity_gate_benchmarked(dev, channels, dt):
    """The benchmarked widths and compute dtype against the fp32 CPU oracle (same check as
    tests/test_gpu_configs.py::test_full_width_bf16_train_step_vs_fp32_oracle, forward part): 2 images of 128x128, train-mode
    forward + Dice/CE.  Reported, not timed."""
    from losses.compound_losses import DiceAndCELoss
    from models.unet import UNet
    from oracle import losses_ref, unet_ref
    torch.manual_seed(1337)
    model = UNet(2, 1, 3, channels, normalization="instance", dropout_prob=None)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    model.set_compute_dtype(torch.bfloat16 if dt == "bf16" else torch.float32)
    size = 128 if len(channels) <= 5 else 96
    img, lab = synth_batch(2, size, 7)
    with torch.no_grad():
        out = model(img.to(dev))
        loss = DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True))(out, lab.to(dev)).item()
        ref = unet_ref.unet_forward(params, img, "instance", True)
        ref_loss = losses_ref.dice_and_ce(ref, lab, 2).item()
    out = out.float().cpu()
    rng = float(ref.max() - ref.min())
    err = float((out - ref).abs().max())
    tol = 1e-4 if dt == "f32" else 2.5e-2 * rng
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * tol
    mism = int((out.argmax(1)[safe] != ref.argmax(1)[safe]).sum())
    # no label-map Dice here: a random-init net's label maps sit on near-ties (see `parity_trained` for the Dice gate)
    return {"config": f"UNet {channels} {size}x{size} bs2 {dt} (benchmarked widths and dtype, RANDOM-INIT weights: logit bound only) "
                      f"vs fp32 CPU oracle",
            "max_abs_logit_diff": err, "logit_range": rng, "tolerance": tol, "loss_diff": abs(loss - ref_loss),
            "label_map_mismatch_px_outside_tolerance_margin": mism, "ok": bool(err < tol and mism == 0)}


def ellipse_set(n, s, seed=1337):
    """Learnable synthetic set: images whose intensity follows two nested ellipses (classes 1, 2) + noise."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
    img = torch.zeros(n, 1, s, s)
    lab = torch.zeros(n, s, s, dtype=torch.long)
    for i in range(n):
        cy, cx, ry, rx = (torch.rand(4, generator=g) * torch.tensor([s / 2, s / 2, s / 6, s / 6]) +
                          torch.tensor([s / 4, s / 4, s / 10, s / 10])).tolist()
        m1 = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        m2 = ((yy - cy - 0.6 * ry) / (0.5 * ry)) ** 2 + ((xx - cx) / (0.5 * rx)) ** 2 < 1
        lab[i][m1] = 1
        lab[i][m2] = 2
        img[i, 0] = 0.25 + 0.35 * m1.float() + 0.3 * m2.float() + 0.08 * torch.randn(s, s, generator=g)
    return img.clamp(0, 1), lab


def parity_gate_trained(dev, channels, dt, steps=60, size=128, nimg=16, batch=8, neval=4):
    """Dice gate on a TRAINED net (north_star: "Dice within ... of CPU reference"; a random-init net's label maps sit on
    near-ties and say little): train the benchmarked widths `steps` engine steps in the benchmarked compute dtype on a small
    learnable set, then compare the eval-mode label maps of the HIP path with the fp32 CPU oracle run on the SAME trained
    weights, per class (medpy-form hard Dice, al_trainer.py:1539-1556), and both against the ground truth.  Reported, not
    timed; tests/test_gpu_configs.py asserts on the same record."""
    from losses.compound_losses import DiceAndCELoss
    from models.unet import UNet
    from oracle import losses_ref, unet_ref
    from training.engine import TrainEngine
    torch.manual_seed(1337)
    model = UNet(2, 1, 3, channels, normalization="instance", dropout_prob=None).to(dev)
    model.set_compute_dtype(torch.bfloat16 if dt == "bf16" else torch.float32)
    loss_fn = DiceAndCELoss(dice_kwargs=dict(num_classes=2, smooth=1e-5, do_bg=True), ce_loss=torch.nn.CrossEntropyLoss)
    eng = TrainEngine(model, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=steps, lr_warmup_iter=max(1, steps // 10))
    img, lab = ellipse_set(nimg, size)
    gi, gl = img.to(dev), lab.to(dev)
    g = torch.Generator().manual_seed(7)
    first = last = None
    for it in range(steps):
        idx = torch.randperm(nimg, generator=g)[:batch].to(dev)
        loss = eng.train_step({"image": gi[idx], "label": gl[idx]})
        if it == 0:
            first = loss.item()
    last = loss.item()
    model.eval()
    with torch.no_grad():
        out = model(gi[:neval]).float().cpu()
        params = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        ref = unet_ref.unet_forward(params, img[:neval], "instance", False)
    pg, pc = out.argmax(1), ref.argmax(1)
    vs_cpu = [losses_ref.hard_dice(pg == k, pc == k) if (pc == k).any() else 1.0 for k in range(3)]
    gt_gpu = [losses_ref.hard_dice(pg == k, lab[:neval] == k) for k in (1, 2)]
    gt_cpu = [losses_ref.hard_dice(pc == k, lab[:neval] == k) for k in (1, 2)]
    rng = float(ref.max() - ref.min())
    return {"config": f"UNet {channels} {size}x{size} {dt}: {steps} engine steps (batch {batch}) on {nimg} synthetic ellipse images, then eval "
                      f"on {neval}: HIP label maps vs the fp32 CPU oracle on the same trained weights",
            "loss_first": first, "loss_last": last, "max_abs_logit_diff_over_range": float((out - ref).abs().max()) / rng,
            "label_map_mismatch_px": int((pg != pc).sum()), "pixels": int(pg.numel()),
            "hard_dice_gpu_vs_cpu_labelmaps": min(vs_cpu), "hard_dice_vs_ground_truth_gpu": sum(gt_gpu) / 2,
            "hard_dice_vs_ground_truth_cpu": sum(gt_cpu) / 2,
            "dice_gap_gpu_vs_cpu": abs(sum(gt_gpu) / 2 - sum(gt_cpu) / 2)}


def pmc_traffic(config, batch, dt):
    """HBM bytes per launch of the canonical block's kernels as measured with the PMC counters
    (profiles/r05_pmc_canonical_block_<cfg>.json, written by tools/r5_pmc_block.sh + tools/r5_pmc_block.py: FETCH_SIZE and WRITE_SIZE in
    separate --pmc passes; bytes = (2 x FETCH_SIZE -- gfx950 half-count correction -- + WRITE_SIZE) x 1024).  {kernel role: bytes} or None."""
    path = os.path.join(ROOT, "profiles", f"r05_pmc_canonical_block_{config}.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None
    if rec.get("config") != config or rec.get("batch") != batch or rec.get("dtype") != dt:
        return None
    out = {k: v["hbm_bytes"] for k, v in rec["kernels"].items() if "hbm_bytes" in v}
    if "conv" in out:
        out["conv_or_dgrad"] = out["conv"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--norm", default=None, choices=["instance", "batch"],
                    help="default: instance (DP-exact, SURVEY 8d) -- batch for cfg4, the al_train default (train.py:25)")
    ap.add_argument("--augment", default=None, choices=["on", "off"],
                    help="cfg4 default on: native-resolution BUSI-shaped inputs resident in HBM -> on-GPU elastic + affine + "
                         "intensity pipeline (al_trainer.py:670-697 + RandomElastic) -> JointResize -> train step, all timed")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="Dropout2d probability of every PlainBlock (al_train default 0.1, al_trainer.py:109); 0 = None")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the train step from a captured hipGraph (TrainEngine(graph=True); single GPU; for the launch-bound small "
                         "configs: the per-launch roofline probes cannot see into a replay, so `roofline` is omitted)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # MIA_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on ONE GPU (RCCL refuses two ranks on one device); every rank
    # then uses device 0.  The driver's runs use the default, RCCL with one GPU per rank.
    backend = os.environ.get("MIA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the ring kernels of the gradient all-reduce run in the CUs TrainEngine reserves (dp_reserve_cus = 8, one per XCD):
        # 8 channels move cfg3's 124 MB in a few ms against ~30 ms of backward, and the compute grids never wait for a CU
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from losses.compound_losses import DiceAndCELoss
    from losses.dice_loss import DiceLoss
    import mia_hip
    from mia_hip import CONV_G3S1, ops
    from models.unet import UNet
    from training.engine import TrainEngine

    channels, size, batch, dt = CONFIGS[args.config]
    batch = args.batch or batch
    dt = args.dtype or dt
    if dt == "f32" and mia_hip.get_option("f32_split"):
        PEAK_MFMA_TFLOPS["f32"] = PEAK_F32_SPLIT_TFLOPS[int(mia_hip.get_option("f32_split"))]
    args.norm = args.norm or ("batch" if args.config == "cfg4" else "instance")
    augment = (args.augment or ("on" if args.config == "cfg4" else "off")) == "on"
    torch.manual_seed(1337)  # identical weights on every rank
    drop = args.dropout if args.dropout > 0 else None
    model = UNet(2, 1, 3, channels, normalization=args.norm, dropout_prob=drop).to(dev)
    model.set_compute_dtype(torch.bfloat16 if dt == "bf16" else torch.float32)
    torch.manual_seed(1337 + rank)  # per-rank stream for the Dropout2d masks (seed + worker id, al_trainer.py:282-288)
    loss_fn = DiceAndCELoss(dice_loss=DiceLoss, dice_kwargs=dict(num_classes=2, smooth=1e-5, do_bg=True, softmax=True,
                                                                 batch=False, squared=False),
                            ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})
    eng = TrainEngine(model, loss_fn, "adam", {"weight_decay": 5e-4}, start_lr=1e-3, num_iters=4000, lr_warmup_iter=250,
                      graph=True if args.graph else None)  # None: the engine's auto mode (replay when its first steps are host-bound)
    img, lab = synth_batch(batch, size, 1337 + rank)
    batch_d = {"image": img.to(dev), "label": lab.to(dev)}  # resident in HBM before timing
    aug = aug_in = None
    aug_log = []
    if augment:
        # BASELINE.json configs[3]: "BUSI 3-class 256x256 with full on-GPU elastic+affine+intensity augmentation pipeline".
        # Native-resolution inputs (BUSI-like ~500x600) resident in HBM; every step draws fresh parameters in the
        # reference's order (host, global torch CPU generator) and runs the batched kernels + JointResize in the timed region.
        from transforms.gpu_pipeline import BatchedAugment, al_train_transforms
        H0, W0 = 496, 608
        g0 = torch.Generator().manual_seed(4242 + rank)
        nat_img = torch.rand(batch, 1, H0, W0, generator=g0)
        _, nat_lab = synth_batch(batch, max(H0, W0), 77 + rank)
        nat_lab = nat_lab[:, :H0, :W0].contiguous()
        aug = BatchedAugment(al_train_transforms("busi", elastic=True), image_size=size, do_normalize=False)
        aug_in = (nat_img.to(dev), nat_lab.to(dev))

    def one_step():
        if aug is None:
            return eng.train_step(batch_d)
        if stream_on[0]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            th = time.perf_counter()
            w0 = aug._arena.wait_s if aug._arena is not None else 0.0
            e0.record()
            b = aug(*aug_in)
            e1.record()
            # host issue = wall time of the call minus the time it sat blocked on a parameter staging buffer the device had not
            # consumed yet (the loop is GPU-bound: the host runs ParamArena.RING batches ahead and then waits there)
            aug_log.append((e0, e1, time.perf_counter() - th - (aug._arena.wait_s - w0), b.get("_bytes", 0)))
        else:
            b = aug(*aug_in)
        return eng.train_step(b)

    c0 = channels[0]
    def _match(mode, c1, c2, nout, h, w, flip):
        if mode != CONV_G3S1:
            return None
        if c1 == c0 and c2 == 0 and nout == c0 and h == size and not flip:
            return "canonical"  # the full-resolution C0 -> C0 block (encoder.levels.0.1 forward)
        # every other 3x3 / stride-1 launch (forward and input-gradient) of the same kernel family: bracketed only in the extra
        # steps AFTER the timed region (two HIP events around each of ~45 launches and ~35 stream calls per step cost the timed
        # loop several per cent: the host-fed loop without them ran 9 % faster than the instrumented resident one)
        return "other_3x3" if sweep_on[0] else None

    probe = ops.LaunchProbe(_match)

    # HBM-bound streams (norm + activation forward, norm backward = reduce + apply): HIP events around the C-ABI call
    # on the launch stream, algorithmic bytes from the call's own shape arguments
    stream_log = {"norm_act_fwd": [], "norm_act_bwd": []}
    block_log = {"finalize": [], "apply": []}  # the canonical block's statistics finalize and normalise + LeakyReLU apply
    stream_on = [False]
    sweep_on = [False]  # True in the extra, untimed steps that bracket EVERY conv / stream launch
    sweep_from = [0]    # index of the first probe record of those steps
    raw_call = ops.call
    val = lambda v: getattr(v, "value", v)

    def timed_call(name, *cargs):
        if stream_on[0] and not sweep_on[0] and name == "mia_norm_finalize" and val(cargs[1]) == batch and val(cargs[3]) == c0 and val(cargs[4]) == size * size:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_call(name, *cargs)
            e1.record()
            block_log["finalize"].append((e0, e1))
            return
        key = {"mia_norm_act_fwd": "norm_act_fwd", "mia_norm_act_bwd": "norm_act_bwd"}.get(name) if stream_on[0] else None
        if key is not None and not sweep_on[0] and not (key == "norm_act_fwd" and val(cargs[5]) == batch and val(cargs[6]) == size * size and val(cargs[7]) == c0):
            key = None  # timed region: only the canonical block's apply pass is bracketed
        if key is None:
            return raw_call(name, *cargs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        raw_call(name, *cargs)
        e1.record()
        if key == "norm_act_fwd" and not sweep_on[0] and val(cargs[5]) == batch and val(cargs[6]) == size * size and val(cargs[7]) == c0:
            block_log["apply"].append((e0, e1))
        if key == "norm_act_fwd":   # (y, z, dtype, scale, shift, n, hw, c, slope, stream): read y, write z
            es = 2 if cargs[2] == 1 else 4
            nbytes = 2.0 * val(cargs[5]) * val(cargs[6]) * val(cargs[7]) * es
        else:                       # (dz, dz2, y, dy, dtype, ..., n, hw, c, ...): reduce reads dz (+dz2), y; apply reads them again, writes dy
            es = 2 if cargs[4] == 1 else 4
            pieces = 2 if cargs[1] is not None else 1
            nbytes = (2.0 * (pieces + 1) + 1.0) * val(cargs[10]) * val(cargs[11]) * val(cargs[12]) * es
        stream_log[key].append((e0, e1, nbytes))

    ops.call = timed_call
    ops.PROBE = probe

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    n_warm = max(args.warmup, TrainEngine.GRAPH_WARMUP + 1) if args.graph else args.warmup  # graph mode: past the capture
    for i in range(n_warm):
        loss = one_step()
    # the engine's auto graph mode decides (and captures) within its first steps: let it settle before the timed region
    extra = 0
    while extra < 8 and ((eng.graph_mode and not eng._graphs) or (eng.graph_auto and not eng.graph_mode)):
        loss = one_step()
        extra += 1
    sync()
    replaying = bool(eng.graph_mode)  # --graph, or the engine's auto mode found the warm-up steps host-bound
    probes = not replaying and os.environ.get("MIA_BENCH_PROBES", "1") != "0"  # (0: A/B of what the instrumentation itself costs)
    probe.enabled = probes
    stream_on[0] = probes
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    if not replaying:  # two extra steps, outside the timed region, with every conv / stream launch bracketed (roofline sub-records)
        sweep_on[0] = True
        sweep_from[0] = len(probe.pairs)
        for v in stream_log.values():
            v.clear()
        for _ in range(2):
            one_step()
        sync()
        sweep_on[0] = False
    probe.enabled = False
    stream_on[0] = False
    elapsed_noaug = None
    if aug is not None:  # the same step on a fixed, already augmented batch: what the pipeline costs (reported, not `value`)
        fixed = aug(*aug_in)
        for _ in range(2):
            eng.train_step(fixed)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            loss2 = eng.train_step(fixed)
        sync()
        elapsed_noaug = time.perf_counter() - t1
    # The same steps fed from HOST memory (the reference's iteration starts with image.to(device) / label.to(device) on DataLoader
    # output, al_trainer.py:1366-1368): pageable CPU tensors in, staged by training.feed.HostFeed (pinned ring, side-stream H2D, labels
    # as bytes), everything inside the timed region.  Reported beside `value`, never as `value` (inputs resident in HBM).
    elapsed_host = host_bytes = None
    if world == 1:
        if aug is None:
            host_batch = {"image": img.clone(), "label": lab.clone()}
            host_step = lambda: eng.train_step(host_batch)
        else:
            from training.feed import HostFeed
            hf = HostFeed(dev)
            nat_cpu = (aug_in[0].cpu(), aug_in[1].cpu())

            def host_step():
                di, dl = hf.stage(*nat_cpu)
                return eng.train_step(aug(di, dl))
        for _ in range(5):  # past the first use of every staging slot (pinned allocations, first-touch page faults)
            host_step()
        sync()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            loss3 = host_step()
        sync()
        elapsed_host = time.perf_counter() - t2
        host_bytes = (eng._feed.bytes_h2d if aug is None else hf.bytes_h2d)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_v = float(loss.item())
    import math
    if not math.isfinite(loss_v) or (elapsed_host is not None and not math.isfinite(float(loss3.item()))):
        raise SystemExit("bench.py: the loss is not finite -- the timed steps did not train (a benchmark on NaN operands also runs at a "
                         "higher clock: NaN payloads toggle fewer bits; round 5)")

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * batch * args.steps / elapsed
        torch.cuda.synchronize()
        kt = [p_[0].elapsed_time(p_[1]) for p_ in probe.pairs[:sweep_from[0] or None] if p_[2] == "canonical"]  # timed region only
        roof = None
        if kt:
            avg_ms = sum(kt) / len(kt)
            flops = 2.0 * 9 * c0 * c0 * size * size * batch
            esz = 2 if dt == "bf16" else 4
            abytes = 2.0 * c0 * size * size * batch * esz + 9 * c0 * c0 * esz
            ach = flops / (avg_ms * 1e-3) / 1e12
            pmc = pmc_traffic(args.config, batch, dt) or {}
            persistent = dt == "bf16" and c0 == 64 and mia_hip.get_option("conv64") != 0
            fused = persistent and ops.FUSE_NL  # both canonical launches consume the previous block's raw output (normalise-on-load)
            bigtile = (not persistent and dt == "bf16" and mia_hip.get_option("conv_bt") != 0 and c0 % 32 == 0 and c0 >= 64 and
                       any(c0 % n == 0 for n in (128, 96, 64)))  # conv_bt_eligible (csrc/conv_bt.hip)
            kname = (f"conv64_persist_kernel<{'NL' if fused else 'plain'}> {c0}->{c0} 3x3 @{size}x{size} x{batch}" if persistent else
                     f"conv_bt_kernel<{128 if c0 % 128 == 0 else 96 if c0 % 96 == 0 else 64}-channel blocks> {c0}->{c0} 3x3 @{size}x{size} x{batch}" if bigtile else
                     f"conv_mma_fast_kernel<{dt},G3S1,MT4,NT4> {c0}->{c0} 3x3 @{size}x{size} x{batch}") + " (encoder.levels.0.1 / decoder.levels.3.1)"
            traffic = pmc.get("conv_nl" if fused else "conv_or_dgrad")
            gbs = abytes / (avg_ms * 1e-3) / 1e9
            t_hbm, t_mfma = abytes / (PEAK_HBM_GBS * 1e9), flops / (PEAK_MFMA_TFLOPS[dt] * 1e12)
            mfma = {"achieved": round(ach, 2), "peak": PEAK_MFMA_TFLOPS[dt], "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_TFLOPS[dt], 4)}
            hbmr = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
            lead = hbmr if t_hbm >= t_mfma else mfma
            conv_rec = {"bound": "hbm" if t_hbm >= t_mfma else "mfma", "achieved": lead["achieved"], "peak": lead["peak"],
                        "unit": lead["unit"], "frac": lead["frac"], "traffic": traffic, "kernel": kname,
                        "avg_launch_ms": round(avg_ms, 4), "launches": len(kt), "flops_per_launch": flops,
                        "algorithmic_bytes_per_launch": abytes, "mfma": mfma, "hbm": hbmr}
            # THE BLOCK (north_star: ">= 70 % of HBM roofline on the fused 3x3 conv block"; SURVEY 8d prices it at read x once +
            # write z once).  Fused (round 4): the conv takes the previous block's RAW output and normalises on load, writes its
            # own raw output + statistics, and its own norm + LeakyReLU is applied on load by ITS consumer -- for
            # decoder.levels.3.1 that is the head kernel, so that block is conv + finalize and nothing else: one activation read,
            # one written.  encoder.levels.0.1 (same conv) still needs an apply pass because the consumers of its output (the
            # stride-2 conv of level 1, the decoder's two-source LDS-DMA conv and their weight gradients) cannot transform on
            # load; that block is reported beside it as `encoder_block` (conv + finalize + apply).
            torch.cuda.synchronize()
            fin = [a_.elapsed_time(b_) for a_, b_ in block_log["finalize"]]
            app = [a_.elapsed_time(b_) for a_, b_ in block_log["apply"]]

            def _block(parts, desc, tr):
                blk_ms = sum(parts.values())
                bgbs = abytes / (blk_ms * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": round(bgbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(bgbs / PEAK_HBM_GBS, 4),
                        "traffic": tr, "kernel": desc, "avg_launch_ms": round(blk_ms, 4), "parts_ms": {k: round(v, 4) for k, v in parts.items()},
                        "launches": len(kt), "algorithmic_bytes_per_launch": abytes, "flops_per_launch": flops,
                        "mfma": {"achieved": round(flops / (blk_ms * 1e-3) / 1e12, 2), "peak": PEAK_MFMA_TFLOPS[dt], "unit": "TFLOP/s",
                                 "frac": round(flops / (blk_ms * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS[dt], 4)}}

            if fin and app:
                fin_ms, app_ms = sum(fin) / len(fin), sum(app) / len(app)
                tsum = lambda *ks: (sum(pmc[k] for k in ks) if all(k in pmc for k in ks) else None)
                unf = _block({"conv": avg_ms, "norm_finalize": fin_ms, "norm_act_fwd": app_ms},
                             f"PlainBlock {c0}->{c0} @{size}x{size} x{batch} (encoder.levels.0.1): conv{' (normalise-on-load)' if fused else ''} + "
                             "norm_finalize + norm_act_fwd (blocks.py:83-102)",
                             tsum("conv_nl" if fused else "conv_or_dgrad", "norm_finalize", "norm_act_fwd"))
                if fused:
                    roof = _block({"conv_nl": avg_ms, "norm_finalize": fin_ms},
                                  f"fused PlainBlock {c0}->{c0} @{size}x{size} x{batch} (decoder.levels.3.1): conv64_persist_kernel<NL> (normalises "
                                  "its input on load, writes raw output + statistics) + norm_finalize; its own norm + LeakyReLU is applied "
                                  "on load by its consumer, the head kernel (blocks.py:83-102, unet.py:157-176)", tsum("conv_nl", "norm_finalize"))
                    roof["encoder_block"] = unf
                    # the two canonical instances side by side: `frac` stays the fused one (the launch the 70 % target names),
                    # `frac_mean` is the mean over the model's two full-resolution C0 -> C0 blocks
                    roof["frac_mean"] = round(0.5 * (roof["frac"] + unf["frac"]), 4)
                else:
                    roof = unf
                roof["conv"] = conv_rec
                if dt == "bf16" and c0 == 64:
                    # context, not a peak: what a synthetic loop with this block's mix (256 flop per HBM byte, 3-4 LDS reads per 8 MFMAs
                    # = the kernel's PMC 0.42-0.45 LDS instructions per MFMA, random bf16 operands) sustains under the board's power
                    # limit -- profiles/r04_mfma_power_frontier.txt (hbm = 2 rows, between ldsr 2 and 4)
                    # (fused launch: + the transform's ~2 packed-fp32 VALU per MFMA -> 3.42 TB/s / 0.88 PF, the "+16pk" row)
                    fr_gbs, fr_tf = (3420.0, 876.0) if fused else (4190.0, 1071.0)
                    roof["power_frontier"] = {"hbm_GBps_at_this_mix": fr_gbs, "mfma_TFLOPs_at_this_mix": fr_tf,
                                              "frac_of_frontier": round(roof["achieved"] / fr_gbs, 3),
                                              "hbm_GBps_with_no_lds_traffic": 4790.0, "mfma_only_TFLOPs": 2062.2,
                                              "source": "profiles/r04_mfma_power_frontier.txt (tools/probe/mfma_power.py)"}
            else:
                roof = conv_rec
        if roof is not None:
            # the same kernel over ALL its launches in the timed region (every 3x3 / stride-1 forward and input-gradient
            # conv): flop-weighted, comparable with the per-symbol average of the rocprofv3 summary in profiles/
            recs = probe.records()[sweep_from[0]:]  # the two extra steps behind the timed region (every launch bracketed there)
            fl = sum(2.0 * 9 * m[1] * m[2] * m[3] * m[4] * m[5] for _, _, m in recs)
            tm = sum(r[0] for r in recs) * 1e-3
            roof["all_3x3_s1_launches"] = {"launches": len(recs), "avg_launch_ms": round(1e3 * tm / len(recs), 4),
                                           "achieved": round(fl / tm / 1e12, 2), "unit": "TFLOP/s",
                                           "frac": round(fl / tm / 1e12 / PEAK_MFMA_TFLOPS[dt], 4), "bound": "mfma",
                                           "measured_in": "two extra steps right after the timed region (every launch bracketed by HIP events there; inside it only the canonical block is)"}
        if roof is not None:
            torch.cuda.synchronize()
            hbm = {}
            for key, recs in stream_log.items():
                if recs:
                    tm = sum(a.elapsed_time(b) for a, b, _ in recs) * 1e-3
                    by = sum(r[2] for r in recs)
                    hbm[key] = {"launches": len(recs), "achieved": round(by / tm / 1e9, 1), "unit": "GB/s", "peak": PEAK_HBM_GBS,
                                "frac": round(by / tm / 1e9 / PEAK_HBM_GBS, 4)}
            hbm["measured_in"] = "two extra steps right after the timed region"
            roof["hbm_streams"] = hbm
        if aug is not None and aug_log:
            # every stage reads its image (+ label) once and writes it once; `_bytes` is counted by BatchedAugment
            # from the stages that actually ran on this batch (per-sample stages still stream the whole batch through)
            torch.cuda.synchronize()
            tg = sum(a.elapsed_time(b) for a, b, _, _ in aug_log) * 1e-3
            th = sum(h for _, _, h, _ in aug_log)
            by = sum(n for _, _, _, n in aug_log)
            aug_rec = {"bound": "hbm", "launches": len(aug_log), "avg_ms_device_span": round(1e3 * tg / len(aug_log), 4),
                       "avg_ms_host_issue": round(1e3 * th / len(aug_log), 4), "algorithmic_bytes_per_batch": by / len(aug_log),
                       "achieved": round(by / tg / 1e9, 1), "unit": "GB/s", "peak": PEAK_HBM_GBS, "frac": round(by / tg / 1e9 / PEAK_HBM_GBS, 4),
                       "share_of_step": round((elapsed - elapsed_noaug) / elapsed, 4) if elapsed_noaug else None,
                       "pipeline": "RandomElastic(p .2) + al_train busi stages (affine scale / rotate, noise, blur, 2 x contrast, low-res, gamma) at "
                                   f"{aug_in[0].shape[-2]}x{aug_in[0].shape[-1]} -> JointResize({size}) (al_trainer.py:670-697, fugc_dataset.py:140-164)"}
            if roof is None:
                roof = {}
            roof["augment"] = aug_rec
        out = {"metric": f"training images/sec (whole node), UNet {size}x{size} 1ch bs={batch}/GPU", "value": round(value, 2),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
               "config": {"workload": f"UNet-2D channels {channels} {size}x{size} 1ch, batch {batch}/GPU, {args.norm} norm, "
                                      f"dropout {drop}, Dice+CE, Adam(wd 5e-4), clip 10 ({args.config})",
                          "global_batch": world * batch, "parallelism": f"dp{world}"},
               "final_loss": round(loss_v, 6), "roofline": roof}
        if replaying:
            out["config"]["graph"] = ("train step replayed from one captured hipGraph" + ("" if args.graph else " (TrainEngine auto mode: the warm-up steps were host-bound)")
                                      + "; per-launch roofline probes cannot see into a replay")
        if dt == "f32":  # fp32 tensors either way; 1 = conv / weight-gradient products from split-f16 operands (DESIGN: fp32 on the f16 matrix cores)
            out["config"]["f32_split"] = int(mia_hip.get_option("f32_split"))
            if out["config"]["f32_split"] and roof is not None:
                nprod = 3 if out["config"]["f32_split"] == 2 else 4
                roof["mfma_peak_note"] = (f"fp32-equivalent TFLOP/s against {2500.0 / nprod:.0f} = dense f16 MFMA peak 2500 / {nprod} products per fp32 "
                                          "multiply-add in the convs (split-f16 operands, csrc/common.h SplitF16)")
        if elapsed_host is not None:
            out["value_host_fed"] = round(world * batch * args.steps / elapsed_host, 2)
            out["ms_per_step_host_fed"] = round(1e3 * elapsed_host / args.steps, 3)
            out["host_fed"] = {"h2d_bytes_per_step": host_bytes, "how": "pageable CPU batch -> pinned staging ring -> H2D on a side stream "
                               "(labels as uint8, widened on the device) -> the same step; training/feed.py HostFeed (al_trainer.py:1366-1368)"}
        if elapsed_noaug is not None:
            out["value_without_augmentation"] = round(world * batch * args.steps / elapsed_noaug, 2)
            out["ms_per_step_without_augmentation"] = round(1e3 * elapsed_noaug / args.steps, 3)
            out["config"]["workload"] += "; inputs: native-resolution batch resident in HBM -> on-GPU elastic+affine+intensity pipeline -> resize, inside the timed step"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(channels, size)
            out["parity"] = parity_gate_benchmarked(dev, channels, dt)
            out["parity_cfg1"] = parity_gate(dev)
            if len(channels) <= 5:
                out["parity_trained"] = parity_gate_trained(dev, channels, dt)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
