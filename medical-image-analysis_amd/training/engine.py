"""Training-step harness for the MI355X UNet path.

Mirrors ``ALTrainer.train_step`` (reference `src/training/al_trainer.py:1350-1399`) with
``_setup_optimizer`` (:737-780), ``_setup_loss`` (:782-800) and ``PolyLRScheduler.step``
(`src/scheduler/lr_scheduler.py:31-55`): set lr, forward, Dice+CE, zero_grad, backward,
clip_grad_norm_(10), optimizer step -- but with

* all parameters / gradients / Adam moments in ONE flat fp32 buffer each (``FlatOptimizer``), so the
  global norm is one reduction and the update one fused HIP launch, the clip coefficient staying on
  the device (the reference's per-iteration ``loss.item()`` host sync, :1381, is gone);
* data parallelism (build-side addition, the reference is single-process): one process per GPU,
  the minibatch sharded across ranks, ONE exchange per step -- a bucketed RCCL all-reduce (sum) of
  the flat gradient buffer, issued from autograd hooks while backward is still running (RCCL runs it
  on its own HIP stream), joined before the global-norm clip; gradients are averaged by folding
  1/world into the optimizer launch.  Buckets are laid out in REVERSE forward order so the first
  gradients produced fill the first bucket.  ``sync_batchnorm=True`` adds the second, small
  collective SURVEY 8e names (per-channel batch statistics, 3*C + 2*C floats per layer) so batch-norm models are
  exact under sharding too.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from mia_hip import OPT_ADAM, OPT_ADAMW, OPT_SGD, ops
from scheduler.lr_scheduler import PolyLRScheduler

_KIND = {"adam": OPT_ADAM, "adamw": OPT_ADAMW, "sgd": OPT_SGD}


class FlatOptimizer:
    """Adam / AdamW / SGD(momentum=0.9) over one flat buffer (reference al_trainer.py:744-761 semantics:
    betas=(0.9, 0.999), no lr at construction -> torch default 1e-3 until the scheduler overwrites it)."""

    def __init__(self, model: torch.nn.Module, name: str = "adamw", lr: float = 1e-3, weight_decay: float = 0.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, momentum: float = 0.9, bucket_bytes: int = 16 << 20):
        if name not in _KIND:
            raise ValueError(f'Optimizer "{name}" not supported')
        self.kind = _KIND[name]
        self.name = name
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise RuntimeError("FlatOptimizer: the model has no trainable parameters")
        dev = params[0].device  # buffers follow the model; step() itself needs a HIP device (no CPU fallback)
        order = params[::-1]  # reverse forward order: first-ready gradients first
        offs, total = [], 0
        for p in order:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32) if self.kind != OPT_SGD else None
        self.params, self.offsets = order, offs
        self._token = object()  # marks the parameters as owned by THIS optimizer: hooks of an earlier one go quiet
        with torch.no_grad():
            for p, o in zip(order, offs):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                p.grad = None
                ops.register_grad_dest(p, self.flat_grad, o)  # backward kernels write the gradient straight into the slice
                p._mia_flat_owner = self._token
                p.register_post_accumulate_grad_hook(self._make_fixup(o))
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps, momentum=momentum)]
        self.step_count = 0
        # ids of the parameters that have been updated at least once (checkpoint export: torch.optim keeps no state for a
        # parameter that never had a gradient).  Limitation, documented: ONE step counter serves the Adam bias correction
        # of every parameter, so a parameter whose first gradient arrives at step t > 1 is corrected with 1 - beta^t where
        # torch.optim would use 1 - beta^1 (does not occur on the al_train path: the set of used parameters is fixed).
        self.stepped = set()
        self.last_norm: Optional[torch.Tensor] = None
        self._pack_plan = None
        # bucket table for data parallelism
        nb = max(1, bucket_bytes // 4)
        self.buckets: List[tuple] = []
        start = 0
        while start < total:
            end = min(total, start + nb)
            # extend to a parameter boundary
            for o, p in zip(offs, order):
                pe = o + (p.numel() + 3) // 4 * 4
                if o < end <= pe:
                    end = pe
                    break
            self.buckets.append((start, end))
            start = end
        self.param_bucket = []
        for o in offs:
            for bi, (s, e) in enumerate(self.buckets):
                if s <= o < e:
                    self.param_bucket.append(bi)
                    break

    def _make_fixup(self, o: int):
        """A gradient that did not come from a registered destination (other autograd nodes, accumulated twice, cloned by
        autograd) is copied into its slice right after accumulation, so the flat buffer is always the truth."""
        def hook(p):
            if getattr(p, "_mia_flat_owner", None) is not self._token:
                return  # a newer FlatOptimizer took the parameter over (al_train builds a new optimizer every round)
            ops.release_grad_dest(p)  # this accumulation is complete: the slice may be claimed again
            g = p.grad
            if g is not None and g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                dst = self.flat_grad[o:o + p.numel()].view(p.shape)
                dst.copy_(g)
                p.grad = dst
        return hook

    def zero_grad(self, set_to_none: bool = False):
        """Unset .grad (autograd then ADOPTS the first gradient -- a view of the flat buffer -- instead of adding into
        it) and clear the buffer, so a parameter that gets no gradient this step contributes zeros."""
        if self.flat_grad.is_cuda:
            ops.call("mia_zero", ops._p(self.flat_grad), ops._c_i64(self.flat_grad.numel() * 4), ops._stream())
        else:
            self.flat_grad.zero_()  # host-side tests of the layout / reducer
        for p in self.params:
            p.grad = None
            ops.release_grad_dest(p)

    def step(self, max_grad_norm: float = 0.0, grad_scale: float = 1.0):
        """Clip-by-global-norm + Adam / AdamW / SGD over the flat buffers (al_trainer.py:1376-1379, :744-761).

        LIMITATION (documented): ONE step counter serves every parameter's Adam bias correction
        (`1 - beta^t`).  torch.optim keeps a counter per parameter, so a parameter that receives its FIRST gradient late -- one
        that was skipped (`.grad is None`) for its first k steps, e.g. a deep-supervision head switched on mid-run -- is
        bias-corrected here as if it were k steps old: its first updates are up to 1 / (1 - beta1^1) : 1 / (1 - beta1^(k+1))
        smaller than torch's.  `al_train` never does this (every parameter it owns gets a gradient from step 1; parameters that
        NEVER get one are never updated, as in torch), and `checkpoint.py` writes the shared counter into every per-parameter
        `step` entry of the torch-format optimizer state."""
        g = self.param_groups[0]
        self.step_count += 1
        clip = None
        if max_grad_norm and max_grad_norm > 0:
            clip = ops.grad_norm(self.flat_grad, float(max_grad_norm), grad_scale)
            self.last_norm = clip
        b1, b2 = (g["momentum"], 0.0) if self.kind == OPT_SGD else g["betas"]
        # torch.optim skips a parameter whose .grad is None (no update, no weight decay, no moment decay) -- e.g. deep-
        # supervision heads al_train never evaluates; update only the runs of the flat buffer that received a gradient
        self.stepped.update(id(p) for p in self.params if p.grad is not None)
        for s0, e0 in self._live_runs():
            ops.optim_step(self.kind, self.flat_param[s0:e0], self.flat_grad[s0:e0], self.m[s0:e0],
                           None if self.v is None else self.v[s0:e0], float(g["lr"]), b1, b2, g["eps"],
                           float(g["weight_decay"]), self.step_count, clip, grad_scale)
        ops.bump_param_epoch()  # the kernel bypasses tensor version counters: packed weights must be rebuilt
        self._repack()

    def _live_runs(self):
        total = self.flat_param.numel()
        if all(p.grad is not None for p in self.params):
            return [(0, total)]
        runs, start = [], None
        for p, o in zip(self.params, self.offsets):
            end = o + (p.numel() + 3) // 4 * 4
            if p.grad is not None:
                start = o if start is None else start
                last = end
            elif start is not None:
                runs.append((start, last))
                start = None
        if start is not None:
            runs.append((start, last))
        return runs

    def _repack(self):
        """Rebuild every packed weight copy in one launch (ops.PackPlan) once the first steps have shown which are used."""
        if self.step_count < 2 or not self.flat_param.is_cuda:
            return
        if self._pack_plan is None or not self._pack_plan.valid():
            dts = set()
            for p in self.params:
                c = ops._caches.get(id(p))
                if c is not None and c[0]() is p:
                    dts.update(dt for dt, _ in c[1]._store)
            if len(dts) != 1:  # nothing packed yet, or mixed compute dtypes: keep the per-tensor path
                return
            self._pack_plan = ops.PackPlan(self.params, dts.pop())
        self._pack_plan.repack()

    def state_dict(self) -> Dict[str, object]:
        return {"step": self.step_count, "m": self.m, "v": self.v, "param_groups": self.param_groups, "name": self.name}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        if self.v is not None and sd.get("v") is not None:
            self.v.copy_(sd["v"])
        self.param_groups[0].update({k: v for k, v in sd["param_groups"][0].items()})


DP_RESERVE_CUS = int(os.environ.get("MIA_DP_RESERVE_CUS", "8"))  # default CU reservation of a multi-rank TrainEngine


class GradBucketReducer:
    """Bucketed all-reduce of FlatOptimizer.flat_grad overlapped with backward (one process per GPU)."""

    def __init__(self, opt: FlatOptimizer, process_group=None, force: bool = False):
        """force (or MIA_DP_FORCE=1): register the hooks and issue the bucketed all-reduces even when the group has ONE
        rank -- a sum over one rank is the identity, so results are bit-identical to the reducer-less path, but the whole
        RCCL branch (ProcessGroupNCCL's stream handshake against the custom kernels' stream) runs on a one-GPU box."""
        self.opt = opt
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = bool(force or os.environ.get("MIA_DP_FORCE") == "1") and dist.is_initialized()
        self.pending = [0] * len(opt.buckets)
        self.counts = [0] * len(opt.buckets)
        for bi in opt.param_bucket:
            self.counts[bi] += 1
        self.works = []
        self.handles = []
        self.suspended = False  # hooks are no-ops (TrainEngine sets it while it RECORDS a backward into a hipGraph)
        self.trace = [] if os.environ.get("MIA_DP_TRACE") else None
        if self.world > 1 or self.force:
            for p, bi in zip(opt.params, opt.param_bucket):
                self.handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_hook(self, bi):
        def hook(_p):
            if self.suspended or getattr(_p, "_mia_flat_owner", None) is not self.opt._token:
                return  # recording into a graph; or the stale reducer of an optimizer that no longer owns the parameter
            self.pending[bi] += 1
            if self.pending[bi] == self.counts[bi]:
                s, e = self.opt.buckets[bi]
                if self.trace is not None:
                    self.trace.append((bi, time.perf_counter()))
                self.works.append(dist.all_reduce(self.opt.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        return hook

    def start_step(self):
        self.pending = [0] * len(self.pending)
        self.works = []
        if self.trace is not None:
            self.trace = [(-1, time.perf_counter())]

    def finish(self):
        """Join every in-flight bucket; launch buckets whose hooks never fired (unused parameters)."""
        if self.world == 1 and not self.force:
            return
        if self.trace is not None and dist.get_rank(self.pg) == 0:
            # MIA_DP_TRACE=1: host-side issue times of the bucket all-reduces relative to the start of backward -- buckets leave
            # while backward is still being issued (the collective itself runs on the backend's own stream / thread)
            t0, tend = self.trace[0][1], time.perf_counter()
            print("[dp-trace] backward issued in %.2f ms; buckets issued at %s ms (bucket id: offset)" % (
                1e3 * (tend - t0), ", ".join("%d: %.2f" % (bi, 1e3 * (t - t0)) for bi, t in self.trace[1:])), flush=True)
        for bi, (s, e) in enumerate(self.opt.buckets):
            if self.pending[bi] != self.counts[bi]:
                self.works.append(dist.all_reduce(self.opt.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for w in self.works:
            w.wait()
        self.works = []

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def reduce_all(self):
        """Every bucket, in index order, issued and joined now (graph mode: the captured backward has filled flat_grad and no hook
        ran; every rank replays the same graphs, so every rank issues the same collectives in the same order)."""
        self.start_step()
        self.finish()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


class _CaptureFailed(RuntimeError):
    """Recording the train step into a hipGraph failed (TrainEngine falls back to eager steps)."""


class _CapturedStep:
    """A captured train step: static inputs, the device-resident per-step scalars, the graph, the static loss."""
    __slots__ = ("img", "lab", "dyn", "graph", "graph_opt", "loss")


class _ReservedCUs:
    """`with _ReservedCUs(k):` -- library option reserve_cus = k inside, the previous value restored on the way out.  The option is
    process-wide and also moves the weight gradients' split-K count: left set, every later kernel of the process (single-rank
    validation, selectors, another engine, the tests) would inherit it."""

    def __init__(self, k):
        self.k, self.prev = k, None

    def __enter__(self):
        if self.k is not None:
            import mia_hip
            self.prev = mia_hip.get_option("reserve_cus")
            mia_hip.set_option("reserve_cus", int(self.k))
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            import mia_hip
            mia_hip.set_option("reserve_cus", self.prev)
        return False


class TrainEngine:
    """One object = model + loss + flat optimizer + poly LR (+ DP reducer).  ``train_step`` returns the loss
    TENSOR (no host sync); call ``.item()`` only when you log."""

    def __init__(self, model, loss_fn, optimizer_name: str = "adam", optimizer_kwargs: Optional[dict] = None,
                 start_lr: float = 1e-3, num_iters: int = 4000, lr_warmup_iter: int = 250, lr_interval: int = 1,
                 lr_scheduler_name: str = "poly", grad_norm: float = 10.0, process_group=None, bucket_bytes: int = 16 << 20,
                 sync_batchnorm: Optional[bool] = None, force_reducer: bool = False, dp_reserve_cus: Optional[int] = None,
                 graph: Optional[bool] = None):
        """force_reducer: see GradBucketReducer(force=...).
        graph: replay the whole step (forward, loss, backward, clip, optimizer, weight re-pack) from ONE captured hipGraph
        (`torch.cuda.CUDAGraph`) after `GRAPH_WARMUP` eager steps -- for small models whose step is launch-bound (cfg1: ~117
        launches for 1.1 ms of kernels).  Same kernels, same order, same arithmetic as the eager step: the values that change per
        iteration (poly LR, Adam bias corrections, Dropout2d Philox offsets) are read from device memory (`ops.StepDyn`).
        Under data parallelism (or `force_reducer`) the step is TWO graphs split at the reducer's join -- forward + loss + backward |
        the bucket all-reduces, issued eagerly in bucket order (RCCL is not captured; nothing overlaps backward, which is what a
        host-bound step can afford) | clip + optimizer + re-pack; every rank must then be constructed with graph=True (auto mode is
        single-rank: ranks deciding for themselves could issue their collectives in different orders), and a model whose forward
        holds a collective (SyncBatchNorm) stays eager.  One graph (pair) per input shape.  None = env MIA_ENGINE_GRAPH if set, else AUTO
        (round 5): a single-rank engine on a HIP device switches to replay by itself when its first eager steps are HOST-bound -- the
        host needed at least 0.6 of the time the device spent between the step's first and last launch to issue them, in at least two
        steps (HIP events around the eager step, read back without blocking) -- and stays eager otherwise (cfg2 / cfg3 / cfg5 are device-bound; cfg1 and al_train-sized models are not: 3.4-4.1 -> 1.06 ms per step).
        In auto mode a third distinct input shape, or a failed capture, returns the engine to eager steps for good.
        dp_reserve_cus: CUs the persistent conv / weight-gradient kernels leave free for RCCL's ring kernels (library option
        `reserve_cus`, include/mia_hip.h).  None (default) = DP_RESERVE_CUS (8: one CU per XCD) when the group has more than
        one rank, else untouched; 0 = never reserve.  Why: the persistent kernels take one 512-thread workgroup per CU on a
        STATIC work list, so a collective kernel that sits on a CU when the next compute kernel launches does not slow that
        kernel by 1/256 -- the displaced workgroup runs after the others and the launch takes twice as long.  Pair it with
        NCCL_MAX_NCHANNELS <= dp_reserve_cus (bench.py does) so the ring kernels fit into the reserved CUs.
        sync_batchnorm: None (default) = ON whenever the model holds batch-norm blocks and the process group has more
        than one rank, so N ranks x bs reproduce one process at N*bs (`normalization="batch"` is the al_train
        default, train.py:25); False keeps per-rank statistics (DDP-without-SyncBN behaviour) and says so once."""
        self.model = model
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if dp_reserve_cus is None:
            dp_reserve_cus = DP_RESERVE_CUS if world > 1 else None
        self.dp_reserve_cus = dp_reserve_cus if next(model.parameters()).is_cuda else None  # applied around each train_step, then restored
        if world > 1:
            from models.unet.blocks import PlainBlock, ResidualBlock, convert_sync_batchnorm
            # ResidualBlock counts too: convert_sync_batchnorm raises NotImplementedError for it, which is the honest
            # answer to "N ranks x bs reproduce one process at N*bs" for that block type (never a silent per-rank fallback)
            has_bn = any(isinstance(m, (PlainBlock, ResidualBlock)) and getattr(m, "normalization", None) == "batch"
                         for m in model.modules())
            if has_bn and (sync_batchnorm is None or sync_batchnorm):
                convert_sync_batchnorm(model, process_group)
            elif has_bn:
                import warnings
                warnings.warn("TrainEngine: batch-norm model on %d ranks with sync_batchnorm=False -- statistics are per rank, "
                              "results differ from a single process at the global batch size" % world)
        self.loss_fn = loss_fn
        kw = dict(optimizer_kwargs or {})
        self.optimizer = FlatOptimizer(model, optimizer_name, bucket_bytes=bucket_bytes, **kw)
        if lr_scheduler_name == "poly":
            self.lr_scheduler = PolyLRScheduler(self.optimizer, initial_lr=start_lr, max_steps=num_iters,
                                                warmup_steps=lr_warmup_iter, interval=lr_interval)
        elif lr_scheduler_name == "none":
            self.lr_scheduler = None
        else:
            raise ValueError(f'Learning rate scheduler "{lr_scheduler_name}" not supported')
        self.grad_norm = grad_norm
        self.reducer = GradBucketReducer(self.optimizer, process_group, force=force_reducer)
        self.current_iter = 0
        self._one = None
        sync_bn = any(getattr(m, "batch_sync", None) is not None for m in model.modules())  # (convert_sync_batchnorm: a collective inside forward)
        can_graph = next(model.parameters()).is_cuda and not sync_bn
        self.graph_auto = False
        if graph is None:
            env = os.environ.get("MIA_ENGINE_GRAPH")
            if env is None or env == "auto":
                graph, self.graph_auto = False, can_graph and not self.reducer.active
            else:
                graph = env != "0"
        if graph and not can_graph:
            raise ValueError("TrainEngine(graph=True) captures the step of a model on a HIP device whose forward holds no collective "
                             "(SyncBatchNorm under data parallelism is not captured)")
        self.graph_mode = bool(graph)
        self._auto_votes, self._auto_seen, self._auto_pending = 0, 0, []  # auto mode: host-bound steps, judged steps, (start, end, host ms) not yet read
        self._graphs: Dict[tuple, "_CapturedStep"] = {}
        self._eager_steps = 0
        self._graph_epoch = None  # ops.PARAM_EPOCH as the last capture / replay left it
        self._feed = None  # training.feed.HostFeed, created by the first batch that arrives in host memory

    GRAPH_WARMUP = 3  # eager steps before capture: lazy module loads, PackPlan creation (step 2), allocator warm-up

    def _capture(self, image: torch.Tensor, label: torch.Tensor) -> "_CapturedStep":
        """Record one train step on static input buffers.  Nothing executes while capturing: the optimizer's step counter and
        the device generator are left where they were, and the first replay performs the step."""
        opt = self.optimizer
        g = _CapturedStep()
        g.img, g.lab = torch.empty_like(image), torch.empty_like(label)
        g.dyn = ops.StepDyn(image.device)
        if self._one is None:
            self._one = torch.ones((), device=image.device, dtype=torch.float32)
        saved = (opt.step_count, set(opt.stepped))
        torch.cuda.synchronize()
        g.graph = torch.cuda.CUDAGraph()
        g.graph_opt = None
        split = self.reducer.active  # two graphs, cut where the gradient buckets are summed over the ranks
        ops._STEP_DYN = g.dyn
        ops.amax_arena_reset()  # the slots this capture uses come from chunks zeroed INSIDE it
        self.reducer.suspended = True
        try:
            with torch.cuda.graph(g.graph, capture_error_mode="thread_local"):
                output = self.model(g.img)
                loss = self.loss_fn(output, g.lab)
                opt.zero_grad()
                loss.backward(self._one if self._one.shape == loss.shape else torch.ones_like(loss))
                if not split:
                    opt.step(max_grad_norm=self.grad_norm, grad_scale=self.reducer.grad_scale)
                g.loss = loss.detach()
            if split:
                g.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g.graph_opt, pool=g.graph.pool(), capture_error_mode="thread_local"):
                    opt.step(max_grad_norm=self.grad_norm, grad_scale=self.reducer.grad_scale)
        finally:
            self.reducer.suspended = False
            ops._STEP_DYN = None
            ops.amax_arena_reset()
            opt.step_count, opt.stepped = saved
        return g

    def _train_step_graph(self, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        key = (tuple(image.shape), tuple(label.shape))
        g = self._graphs.get(key)
        if g is None:
            try:
                g = self._graphs[key] = self._capture(image, label)
                self._graph_epoch = ops.PARAM_EPOCH
            except Exception as e:  # capture errors surface as RuntimeError from torch / MiaError from a launch
                raise _CaptureFailed() from e
        opt = self.optimizer
        pg = opt.param_groups[0]
        opt.step_count += 1
        b1, b2 = (pg["momentum"], 0.0) if opt.kind == OPT_SGD else pg["betas"]
        dev = image.device
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        seed, off = int(gen.initial_seed()), int(gen.get_offset())
        gen.set_offset(off + g.dyn.rng_span)  # what the eager step's mask launches would have consumed
        g.dyn.set(float(pg["lr"]), 1.0 - b1 ** opt.step_count, 1.0 - b2 ** opt.step_count, opt.step_count == 1, seed, off)
        g.img.copy_(image, non_blocking=True)
        g.lab.copy_(label, non_blocking=True)
        if self._graph_epoch != ops.PARAM_EPOCH and opt._pack_plan is not None:
            # parameters changed outside the graphs (load_state_dict, a checkpoint restore, an in-place edit: they bump the epoch):
            # the captured forward holds no pack launch -- at capture time every packed copy was current -- so rebuild them here
            opt._pack_plan.repack()
        g.graph.replay()
        if g.graph_opt is not None:
            self.reducer.reduce_all()
            g.graph_opt.replay()
        # the replay's optimizer rewrote the parameters and its captured re-pack rewrote the plan's copies: new epoch (a packed
        # copy made outside the plan must miss next time), plan copies marked current
        ops.bump_param_epoch()
        if opt._pack_plan is not None:
            opt._pack_plan.plant()
        self._graph_epoch = ops.PARAM_EPOCH
        opt.stepped.update(id(p) for p in opt.params if p.grad is not None)
        return g.loss.clone()

    def train_step(self, sampled_batch) -> torch.Tensor:
        with _ReservedCUs(self.dp_reserve_cus):
            return self._train_step(sampled_batch)

    def _train_step(self, sampled_batch) -> torch.Tensor:
        self.model.train()
        if self.lr_scheduler:
            self.lr_scheduler.step(self.current_iter)
        dev = self.optimizer.flat_param.device
        image, label = sampled_batch["image"], sampled_batch["label"]
        if dev.type == "cuda" and not image.is_cuda and not label.is_cuda and label.dtype == torch.int64:
            # a batch in host memory (DataLoader output; the reference's image.to(device) / label.to(device), al_trainer.py:1366-1368):
            # pinned staging ring, copies on a side stream, labels as bytes when they fit -- the host does not wait for the transfer
            if self._feed is None:
                from training.feed import HostFeed
                self._feed = HostFeed(dev)
            image, label = self._feed.stage(image, label)
        else:
            image = image.to(dev, dtype=torch.float32, non_blocking=True)
            label = label.to(dev, dtype=torch.long, non_blocking=True)
        if self.graph_mode and self.graph_auto and len(self._graphs) >= 2 and (tuple(image.shape), tuple(label.shape)) not in self._graphs:
            self.graph_mode = self.graph_auto = False  # auto mode: input shapes keep changing -- a capture per shape would cost more than it saves
            self._graphs.clear()
        if self.graph_mode and self._eager_steps >= self.GRAPH_WARMUP:
            try:
                loss = self._train_step_graph(image.contiguous(), label.contiguous())
                self.current_iter += 1
                return loss
            except _CaptureFailed as e:
                # no kernel has executed (a capture records, it does not run), but the HOST side of the recorded part did: packed
                # copies were marked current for pack launches that never ran, producer -> consumer hints and gradient-slice claims
                # point at unwritten tensors.  Invalidate all of it, then drop to the eager step for good.
                import warnings
                warnings.warn(f"TrainEngine: hipGraph capture of the train step failed ({e.__cause__!r}); continuing with eager steps")
                self.graph_mode = self.graph_auto = False
                self._graphs.clear()
                ops.bump_param_epoch()
                ops.clear_hints()
                ops.amax_arena_reset()
                self.optimizer._pack_plan = None
                self.optimizer.zero_grad()
        self._eager_steps += 1
        watch = self.graph_auto and not self.graph_mode and self._eager_steps >= 2
        if watch:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            t_host = time.perf_counter()
        output = self.model(image)
        loss = self.loss_fn(output, label)
        self.optimizer.zero_grad()
        self.reducer.start_step()
        if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)  # reused d(loss)/d(loss): autograd would launch a fill kernel per step
        loss.backward(self._one)
        self.reducer.finish()
        self.optimizer.step(max_grad_norm=self.grad_norm, grad_scale=self.reducer.grad_scale)
        self.current_iter += 1
        if watch:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self._auto_pending.append((ev0, ev1, 1e3 * (time.perf_counter() - t_host)))
        if self.graph_auto and not self.graph_mode:
            self._auto_decide(loss)
        return loss.detach()

    AUTO_GIVE_UP = 8  # eager steps after which auto mode stops looking

    def _auto_decide(self, loss: torch.Tensor) -> None:
        """Auto graph mode: is this engine host-bound?  No synchronisation: every watched eager step left (start event, end event,
        host milliseconds spent issuing it); records whose end event has completed are read back.  Device-bound: the device reaches the
        start event late and works through the step's kernels -- its span is the kernel time, many times the host's issue time.
        Host-bound: the device executes each launch as it arrives -- its span IS the host's issue time."""
        if loss.dim() != 0 or not loss.is_cuda:
            self.graph_auto = False
            return
        while self._auto_pending and self._auto_pending[0][1].query():
            ev0, ev1, host_ms = self._auto_pending.pop(0)
            self._auto_seen += 1
            if host_ms >= 0.6 * ev0.elapsed_time(ev1):
                self._auto_votes += 1
        if self._eager_steps >= self.GRAPH_WARMUP and self._auto_votes >= 2:
            self.graph_mode = True
            self._auto_pending = []
        elif self._auto_seen >= 3 and self._auto_votes == 0 or self._eager_steps >= self.AUTO_GIVE_UP:
            self.graph_auto = False
            self._auto_pending = []

    def loss_value(self, loss: torch.Tensor) -> float:
        """`loss.item()` for logging (al_trainer.py:1381) -- the one host sync of a logged step -- plus the label-range check
        the reference performs by raising inside the loss (dice_loss.py:25-30)."""
        v = float(loss.item())
        ops.check_labels()
        return v

    def save_state_dict(self, save_path, save_training_state: bool = False, **extra) -> None:
        """model.pth (+ training_state.pth) in the reference's layout (al_trainer.py:1719-1733)."""
        from training import checkpoint
        checkpoint.save_state_dict(self, save_path, save_training_state, **extra)

    def load_state_dict(self, save_path) -> dict:
        """al_trainer.py:1704-1717."""
        from training import checkpoint
        return checkpoint.load_state_dict(self, save_path)

    @torch.no_grad()
    def predict(self, image: torch.Tensor) -> torch.Tensor:
        """valid_slices core (al_trainer.py:1428-1431): eval forward -> softmax -> argmax."""
        self.model.eval()
        out = self.model(image.to(self.optimizer.flat_param.device, dtype=torch.float32))
        return out.softmax(1).argmax(1)
