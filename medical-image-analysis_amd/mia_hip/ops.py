"""Tensor-level wrappers + autograd glue over libmia_hip.so.

PyTorch-ROCm is used only as a container (device memory, streams, autograd graph); every FLOP and
every byte moved on the hot path goes through the HIP kernels behind ``include/mia_hip.h``.
Activations are plain contiguous NHWC tensors ``[N, H, W, C]`` in fp32 or bf16.  There is no CPU
fallback: tensors must live on a HIP device and the shared object must be built.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch

from . import (BF16, CONV_G1, CONV_G2S2, CONV_G3S1, CONV_G3S2, CONV_T2S2, CONV_T3S2, F32, LOSS_BATCH, LOSS_DO_BG,
               LOSS_DENSE, LOSS_SOFTMAX, LOSS_SQUARED, NORM_BATCH, NORM_INSTANCE, WGRAD_2S2, WGRAD_3S1, WGRAD_3S2, MiaError, call, lib)

LRELU_SLOPE = 0.01
_c_int, _c_float, _c_i64 = ctypes.c_int, ctypes.c_float, ctypes.c_int64


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise MiaError(f"unsupported activation dtype {d} (fp32 or bf16)")


def _need_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MiaError("libmia_hip operates on HIP device tensors only (no CPU fallback); got a CPU tensor")


def _pad(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _kb(dtype: int) -> int:
    return 32 if dtype == BF16 else 16


# ------------------------------------------------------------------ operand maxima (fp32 convs on the f16 matrix cores)
# The fp32 tile kernels multiply on the f16 matrix cores from two-part split operands (csrc/common.h SplitF16); each operand tensor is
# scaled by a power of two taken from its max |x|, which the kernels read from a 4-byte device slot.  A slot travels with its
# tensor as a Python attribute (`_mia_amax` = (slot, tensor version)): the forward conv, the weight gradient that contracts the same
# activation in backward, and the two consumers of one gradient tensor share one reduction.  An attribute cannot outlive its tensor
# (unlike a table keyed by address), and a version change (in-place edit) makes the next consumer measure again.
F32_SPLIT_MIN_MACS = int(__import__('os').environ.get('MIA_F32_SPLIT_MIN_MACS', str(1 << 26)))  # below this many multiply-adds the exact kernel is as fast as the reduction + split


class _SlotArena:
    """Zeroed 4-byte slots, carved from chunks that ONE launch zeroes (a slot per norm / activation pass and per measured tensor
    would otherwise cost a zeroing launch each).  A slot is a view of its chunk, so the chunk lives as long as any of its slots.
    A chunk started outside a stream capture is never used inside one and vice versa: the zeroing launch must be part of the
    graph that uses the slots, or replays would fold into the previous replay's maxima (still safe -- a larger maximum only
    costs precision -- but no longer the eager step's arithmetic)."""
    CHUNK = 128

    def __init__(self):
        self.chunk, self.used, self.capturing, self.device = None, 0, False, None

    def reset(self) -> None:
        self.chunk = None

    def take(self, device) -> torch.Tensor:
        cap = torch.cuda.is_current_stream_capturing()
        if self.chunk is None or self.used >= self.CHUNK or cap != self.capturing or device != self.device:
            self.chunk = torch.empty(self.CHUNK, device=device, dtype=torch.int32)
            call("mia_zero", _p(self.chunk), _c_i64(self.CHUNK * 4), _stream())
            self.used, self.capturing, self.device = 0, cap, device
        self.used += 1
        return self.chunk[self.used - 1:self.used]


_ARENA = _SlotArena()


def amax_arena_reset() -> None:
    """Start a fresh chunk at the next request (training/engine.py: at both ends of a graph capture)."""
    _ARENA.reset()


def amax_slot(t: torch.Tensor) -> torch.Tensor:
    """Device slot (int32[1]) with max |t| as an fp32 bit pattern; measured once per tensor version (`mia_amax`)."""
    hit = getattr(t, "_mia_amax", None)
    if hit is not None and hit[1] == t._version:
        return hit[0]
    slot = _ARENA.take(t.device)
    tc = t if t.is_contiguous() else t.contiguous()
    call("mia_amax", _p(tc), _c_i64(tc.numel()), _p(slot), 0, _stream())
    try:
        t._mia_amax = (slot, t._version)
    except (AttributeError, RuntimeError):
        pass
    return slot


def _amax_new(t: torch.Tensor) -> Optional[torch.Tensor]:
    """A fresh slot, attached to `t`, that the kernel about to WRITE t fills as a by-product (`amax_out` of the norm / activation
    passes): the convs that consume t then need no reduction pass of their own.  fp32 tensors only."""
    if t.dtype != torch.float32:
        return None
    slot = _ARENA.take(t.device)
    t._mia_amax = (slot, t._version)
    return slot


def _dup_view(z: torch.Tensor) -> torch.Tensor:
    """Second handle on z (PlainBlockFn dup=True) that carries the same maximum slot."""
    v = z.view(z.shape)
    h = getattr(z, "_mia_amax", None)
    if h is not None:
        v._mia_amax = (h[0], v._version)
    return v


def _split_descs(items, device):
    """Device table for mia_split_f16_batch: items = [(packed fp32 buffer, int32 destination of the same shape, maximum slot)]."""
    import ctypes as C

    class SDesc(C.Structure):
        _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64), ("amax", C.c_void_p)]

    assert C.sizeof(SDesc) == lib().mia_split_desc_bytes()
    arr = (SDesc * len(items))()
    for i, (buf, dst, slot) in enumerate(items):
        arr[i] = SDesc(buf.data_ptr(), dst.data_ptr(), buf.numel(), slot.data_ptr())
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def _split_ok(x: torch.Tensor, macs: int) -> bool:
    return x.dtype == torch.float32 and macs >= F32_SPLIT_MIN_MACS


# ------------------------------------------------------------------ weight packing (cached per parameter version)
PARAM_EPOCH = 0


def bump_param_epoch() -> None:
    """Parameters were modified behind torch's back (the fused optimizer kernel writes the flat buffer directly and
    does not touch tensor version counters): invalidate every packed-weight cache."""
    global PARAM_EPOCH
    PARAM_EPOCH += 1


class PackCache:
    """Packed copies of one fp32 parameter, invalidated by the tensor's version counter or the global epoch."""

    def __init__(self):
        self._store = {}

    def get(self, w: torch.Tensor, dtype: int, n_from_d0: bool, kpad_mult: Optional[int] = None) -> Tuple[torch.Tensor, int, int]:
        key = (dtype, n_from_d0)
        ver = (w._version, w.data_ptr(), PARAM_EPOCH)
        hit = self._store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2], hit[3]
        d0, d1 = w.shape[0], w.shape[1]
        taps = w.shape[2] * w.shape[3]
        nn, kk = (d0, d1) if n_from_d0 else (d1, d0)
        npad, kpad = _pad(nn, 64), _pad(kk, _kb(dtype))
        buf = hit[1] if hit is not None else \
            torch.empty((taps, npad, kpad), device=w.device, dtype=torch.bfloat16 if dtype == BF16 else torch.float32)
        wc = w.detach()
        if not wc.is_contiguous():
            wc = wc.contiguous()
        call("mia_pack_weight", _p(wc), _p(buf), dtype, d0, d1, taps, npad, kpad, int(n_from_d0), _stream())
        if dtype == F32:  # the split-f16 kernels scale the weights by their maximum (amax_slot docstring); slot rides on the packed buffer
            slot = getattr(buf, "_mia_amax", None)
            slot = slot[0] if slot is not None else torch.empty(1, device=w.device, dtype=torch.int32)
            call("mia_amax", _p(wc), _c_i64(wc.numel()), _p(slot), 1, _stream())
            buf._mia_amax = (slot, buf._version)
            # ... and the packed weights once more as (h | l) fp16 words of w * 2^e, so the split kernels do not split them per tile
            sp = getattr(buf, "_mia_split", None)
            if sp is None:
                sp = torch.empty(buf.shape, device=w.device, dtype=torch.int32)
                buf._mia_split = sp
                buf._mia_split_desc = _split_descs([(buf, sp, slot)], w.device)
            call("mia_split_f16_batch", _p(buf._mia_split_desc), 1, _stream())
        self._store[key] = (ver, buf, npad, kpad)
        return buf, npad, kpad


class PackPlan:
    """All packed copies a model's step needs, refreshed by ONE `mia_pack_weight_batch` launch after the optimizer has
    rewritten the parameters (instead of one launch per tensor and orientation on first use).  Built from the packs the
    first steps left in the parameters' PackCaches; the results are planted into each parameter's PackCache with the current version key, so a
    parameter changed behind the plan's back (load_state_dict, in-place edits) simply misses and re-packs itself."""

    def __init__(self, params, dtype: int):
        import ctypes as C
        ents = []
        for w in params:
            c = _caches.get(id(w))
            if c is None or c[0]() is not w or not w.is_cuda or not w.is_contiguous() or w.ndim != 4:
                continue
            for (dt, n_from_d0) in c[1]._store:
                if dt == dtype:
                    ents.append(((id(w), dt, n_from_d0), (w, c[1])))
        if not ents:
            raise MiaError("PackPlan: no packed weights were requested yet (run a step first)")

        class Desc(C.Structure):
            _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("taps", C.c_int),
                        ("npad", C.c_int), ("kpad", C.c_int), ("n_from_d0", C.c_int), ("brick_begin", C.c_int), ("bricks_x", C.c_int)]

        assert C.sizeof(Desc) == lib().mia_pack_desc_bytes()
        arr = (Desc * len(ents))()
        self.entries, self.dtype, bricks, self.max_taps = [], dtype, 0, 1
        for i, ((_, _, n_from_d0), (w, pc)) in enumerate(ents):
            key = (dtype, n_from_d0)
            _, buf, npad, kpad = pc._store[key]
            d0, d1, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            if d0 * d1 * taps * 4 >= 1 << 31:
                raise MiaError("PackPlan: a weight tensor of 2 GiB or more is outside the batched pack kernel's 32-bit offsets")
            pa, pb = (npad, kpad) if n_from_d0 else (kpad, npad)  # padded extents along D0 / D1
            ta, tb = (16, 64) if n_from_d0 else (64, 16)
            bx, by = -(-pb // tb), -(-pa // ta)
            arr[i] = Desc(w.data_ptr(), buf.data_ptr(), d0, d1, taps, npad, kpad, int(n_from_d0), bricks, bx)
            bricks += bx * by
            self.max_taps = max(self.max_taps, taps)
            self.entries.append((w, pc, key, buf, npad, kpad, w.data_ptr()))
        self.total_bricks = bricks
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.descs = host.to(self.entries[0][0].device)
        self.amax_descs = None
        if dtype == F32:  # maxima of every packed parameter in one launch; slot i rides on entry i's packed buffer
            class ADesc(C.Structure):
                _fields_ = [("src", C.c_void_p), ("n", C.c_int64)]

            assert C.sizeof(ADesc) == lib().mia_amax_desc_bytes()
            aarr = (ADesc * len(self.entries))()
            for i, (w, *_rest) in enumerate(self.entries):
                aarr[i] = ADesc(w.data_ptr(), w.numel())
            self.amax_descs = torch.frombuffer(bytearray(bytes(aarr)), dtype=torch.uint8).to(self.entries[0][0].device)
            self.amax_slots = torch.zeros(len(self.entries), device=self.entries[0][0].device, dtype=torch.int32)
            self.split_bufs = [getattr(e[3], "_mia_split", None) for e in self.entries]
            for i, e in enumerate(self.entries):
                if self.split_bufs[i] is None:
                    self.split_bufs[i] = torch.empty(e[3].shape, device=e[3].device, dtype=torch.int32)
            self.split_descs = _split_descs([(e[3], self.split_bufs[i], self.amax_slots[i:i + 1]) for i, e in enumerate(self.entries)],
                                            self.entries[0][0].device)

    def valid(self) -> bool:
        return all(w.data_ptr() == ptr for w, _, _, _, _, _, ptr in self.entries)

    def repack(self) -> None:
        call("mia_pack_weight_batch", _p(self.descs), len(self.entries), self.total_bricks, self.max_taps, self.dtype, _stream())
        if self.amax_descs is not None:
            call("mia_amax_batch", _p(self.amax_descs), len(self.entries), _p(self.amax_slots), _stream())
            call("mia_split_f16_batch", _p(self.split_descs), len(self.entries), _stream())
        self.plant()

    def plant(self) -> None:
        """Mark the plan's packed copies as current (host bookkeeping only): after `repack`, and after a graph replay whose
        captured re-pack launch rewrote them while the host-side keys stayed where the capture left them."""
        for i, (w, pc, key, buf, npad, kpad, _) in enumerate(self.entries):
            pc._store[key] = ((w._version, w.data_ptr(), PARAM_EPOCH), buf, npad, kpad)
            if self.amax_descs is not None:
                buf._mia_amax = (self.amax_slots[i:i + 1], buf._version)
                buf._mia_split = self.split_bufs[i]


_caches = {}


def pack_cache(w: torch.Tensor) -> PackCache:
    c = _caches.get(id(w))
    if c is None or c[0]() is not w:
        import weakref
        pc = PackCache()
        _caches[id(w)] = (weakref.ref(w, lambda _r, k=id(w): _caches.pop(k, None)), pc)
        return pc
    return c[1]


# ------------------------------------------------------------------ layout helpers
def to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Logical NCHW tensor (any dense strides) -> contiguous NHWC tensor in `dtype` (HIP relayout kernel)."""
    _need_dev(x)
    n, c, h, w = x.shape
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    sn, sc, sh, sw = x.stride()
    if x.dtype == dtype:
        if c == 1 and sw == 1 and (h == 1 or sh == w) and (n == 1 or sn == h * w):
            return x.reshape(n, h, w, 1)  # NCHW with C=1 is already NHWC
        if sc == 1 and sw == c and (h == 1 or sh == w * c) and (n == 1 or sn == h * w * c):
            return x.permute(0, 2, 3, 1)  # channels_last storage: zero-copy
    if h > 1 and sh != w * sw:
        x = x.contiguous()
        sn, sc, sh, sw = x.stride()
    out = torch.empty((n, h, w, c), device=x.device, dtype=dtype)
    call("mia_relayout", _p(x), _dt(x), _p(out), _dt(out), n, c, _c_i64(h * w), _c_i64(sn), _c_i64(sc), _c_i64(sw),
         _c_i64(h * w * c), _c_i64(1), _c_i64(c), _stream())
    return out


def nhwc_as_nchw(x: torch.Tensor) -> torch.Tensor:
    """Zero-copy logical-NCHW view (channels_last strides) of a contiguous NHWC tensor."""
    return x.permute(0, 3, 1, 2)


def cast_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    n, h, w, c = x.shape
    out = torch.empty_like(x, dtype=dtype)
    call("mia_relayout", _p(x), _dt(x), _p(out), _dt(out), n, c, _c_i64(h * w), _c_i64(h * w * c), _c_i64(1), _c_i64(c),
         _c_i64(h * w * c), _c_i64(1), _c_i64(c), _stream())
    return out


# ------------------------------------------------------------------ raw op wrappers
class LaunchProbe:
    """Brackets matching conv_mma launches with HIP events on the launch stream (bench.py roofline leg)."""

    def __init__(self, match):
        self.match, self.pairs, self.enabled = match, [], False

    def times_ms(self, tag=None):
        """Durations of the bracketed launches (all, or those whose match() returned `tag`)."""
        torch.cuda.synchronize()
        return [p[0].elapsed_time(p[1]) for p in self.pairs if tag is None or p[2] == tag]

    def records(self):
        """(duration_ms, tag, (mode, c1 + c2, nout, n, hout, wout)) per bracketed launch."""
        torch.cuda.synchronize()
        return [(p[0].elapsed_time(p[1]), p[2], p[3]) for p in self.pairs]


PROBE: Optional[LaunchProbe] = None


def conv_tiles(mode: int, hout: int, wout: int) -> int:
    ty, tx, th = _c_int(), _c_int(), _c_int()
    call("mia_conv_mma_tiles", mode, hout, wout, ctypes.byref(ty), ctypes.byref(tx), ctypes.byref(th))
    return ty.value * tx.value


def conv_mma(mode: int, x1: torch.Tensor, x2: Optional[torch.Tensor], wpack: torch.Tensor, npad: int, kpad: int,
             flip: bool, bias: Optional[torch.Tensor], nout: int, out_hw: Tuple[int, int], want_stats: bool = False,
             out_split: Optional[int] = None, nl=None, cr=None):
    """nl = (coefs [5, N, C] of the producing block, slope): x1 is that block's RAW conv output and the kernel normalises on
    load (`mia_conv_mma_nl`; the fused PlainBlock).  cr = (y_prod, coefs, slope): this launch is the input gradient that
    produces dz for the block with raw output `y_prod`, and its epilogue does that block's norm-backward reduction
    (`mia_conv_mma_cr`); the third return value is then the partials tensor [N, tiles, nout, 2]."""
    n, hin, win, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[3]
    hout, wout = out_hw
    if out_split is None:
        out1 = torch.empty((n, hout, wout, nout), device=x1.device, dtype=x1.dtype)
        out2, o1, o2 = None, nout, 0
    else:
        o1, o2 = out_split, nout - out_split
        out1 = torch.empty((n, hout, wout, o1), device=x1.device, dtype=x1.dtype)
        out2 = torch.empty((n, hout, wout, o2), device=x1.device, dtype=x1.dtype)
    stats = None
    if want_stats or cr is not None:
        stats = torch.empty((n, conv_tiles(mode, hout, wout), nout, 2), device=x1.device, dtype=torch.float32)
    # fp32: operand maxima for the split-f16 products (None -> the library runs its exact fp32 kernels)
    am1 = am2 = amw = wsp = None
    if cr is None and nl is None and _split_ok(x1, n * hout * wout * nout * (c1 + c2)):
        wh = getattr(wpack, "_mia_amax", None)
        if wh is not None:
            amw, am1 = wh[0], amax_slot(x1)
            am2 = None if x2 is None else amax_slot(x2)
            wsp = getattr(wpack, "_mia_split", None)
    tag = PROBE.match(mode, c1, c2, nout, hin, win, bool(flip)) if (PROBE is not None and PROBE.enabled) else None
    probe = PROBE if tag else None
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if cr is not None:
        assert x2 is None and out_split is None and nl is None and bias is None
        yp, cf, sl = cr
        call("mia_conv_mma_cr", mode, _dt(x1), _p(x1), c1, _p(wpack), npad, kpad, int(flip), _p(out1), o1, _p(yp), _p(cf[2]), _p(cf[3]),
             _p(cf[0]), _p(cf[1]), _c_float(sl), _p(stats), n, hin, win, hout, wout, _stream())
    elif nl is not None:
        assert x2 is None and out_split is None and not flip
        call("mia_conv_mma_nl", mode, _dt(x1), _p(x1), c1, _p(nl[0][2]), _p(nl[0][3]), _c_float(nl[1]), _p(wpack), npad, kpad,
             _p(bias), _p(out1), o1, _p(stats), n, hin, win, hout, wout, _stream())
    else:
        # outputs that another conv consumes as they are (ConvTranspose output; the decoder conv's gradient w.r.t. it): maximum in the epilogue
        ao1 = _amax_new(out1) if (mode == CONV_T2S2 and x1.dtype == torch.float32) else None
        ao2 = _amax_new(out2) if (out2 is not None and x1.dtype == torch.float32) else None
        call("mia_conv_mma", mode, _dt(x1), _p(x1), c1, _p(x2), c2, _p(wpack), npad, kpad, int(flip), _p(bias), _p(out1), o1,
             _p(out2), o2, _p(stats), n, hin, win, hout, wout, _p(am1), _p(am2), _p(amw), _p(wsp), _p(ao1), _p(ao2), _stream())
    if probe is not None:
        e1.record()
        probe.pairs.append((e0, e1, tag, (mode, c1 + c2, nout, n, hout, wout)))
    return out1, out2, stats


WGRAD_TARGET_BLOCKS = int(__import__('os').environ.get('MIA_WGRAD_BLOCKS', '0'))  # 0 = ask the library (one or two workgroups per CU)


WGRAD_KSPLIT_MODEL = __import__('os').environ.get('MIA_WGRAD_KSPLIT_MODEL', '1') != '0'  # A/B knob


def _ksplit_by_cost(base: int, slots: int, ntiles: int, slab_bytes: int, flops: float) -> int:
    """Split-K count where base * ksplit cannot hit the machine's workgroup slots exactly (channel counts that are not powers of two:
    cfg5's 96-multiples).  The workgroups run in rounds of `slots`, so a count just above a multiple of `slots` wastes most of a round --
    384 channels: 36 column blocks x 8 slices = 288 of 512 slots, one round at 56 %; 768: 144 x 4 = 576 = two rounds at 56 % -- while every
    extra slice writes and re-reads one more fp32 slab.  Minimise rounds(k) * slots / (base * k) * compute + k * slab traffic over
    k = 1 .. 7 and the multiples of 8 (the XCD-aware launch order needs those)."""
    t_compute = flops / 1.2e15            # the kernels' sustained rate on these shapes
    cands = [k for k in range(1, 8) if k <= ntiles] + [k for k in range(8, min(ntiles, 1024) + 1, 8)]
    best, best_t = 1, None
    for k in cands:
        rounds = -(-base * k // slots)
        t = rounds * slots / (base * k) * t_compute + k * 2.0 * slab_bytes / 4.0e12
        if best_t is None or t < best_t * 0.98:   # (a larger k must win by 2 %)
            best, best_t = k, t
    return best


def conv_wgrad(mode: int, x1: torch.Tensor, x2: Optional[torch.Tensor], dy: torch.Tensor, grad_shape, nn: int, kk: int,
               out: Optional[torch.Tensor] = None, nl=None) -> torch.Tensor:
    """Weight gradient in the parameter's native layout (fp32), written into `out` when given.  nl = (coefs, slope): x1 is the
    producing block's raw conv output, normalised on load (`mia_conv_wgrad_nl`)."""
    n, hx, wx, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[3]
    _, hy, wy, cdy = dy.shape
    dtype = _dt(x1)
    taps = 4 if mode == WGRAD_2S2 else 9
    npad, kpad = _pad(nn, 64), _pad(kk, 64)
    # column blocks of one split-K slice, workgroups to aim for and the tile height of the kernel the library will pick for this shape
    # (mia_wgrad_plan: 64-wide blocks cut per source, or 96-wide ones where every channel count is a multiple of 96 and not of 64)
    pb, pt, ph = _c_int(), _c_int(), _c_int()
    call("mia_wgrad_plan", mode, dtype, c1, c2, cdy, npad, hy, ctypes.byref(pb), ctypes.byref(pt), ctypes.byref(ph))
    ntiles = n * -(-hy // ph.value) * -(-wy // 16)
    # column blocks of one split-K slice: the kernels cut the input channels per SOURCE (ceil(c1 / 64) + ceil(c2 / 64) blocks, which
    # is more than kpad / 64 when neither source is a multiple of 64: cfg5's 96 + 96), and the XCD-aware launch order needs a split
    # count that is a multiple of 8 (86, 57 or 29 slices dropped those launches to the plain 2-D order and overfilled the chip:
    # 688 workgroups for 512 slots on cfg5's 192 -> 96 gradient)
    base = pb.value
    target = WGRAD_TARGET_BLOCKS or pt.value
    ksplit = max(1, min(ntiles, -(-target // base), 1024))
    if 8 <= ksplit < ntiles:  # (one tile per slice already: a small problem, leave it)
        ksplit -= ksplit % 8
    if dtype == BF16 and WGRAD_KSPLIT_MODEL and base * ksplit != target:
        ksplit = _ksplit_by_cost(base, target, ntiles, taps * npad * kpad * 4, 2.0 * taps * nn * kk * n * hy * wy)
    slabs = torch.empty((ksplit, taps, npad, kpad), device=x1.device, dtype=torch.float32)
    if nl is not None:
        assert x2 is None
        call("mia_conv_wgrad_nl", mode, dtype, _p(x1), c1, _p(nl[0][2]), _p(nl[0][3]), _c_float(nl[1]), _p(dy), cdy, _p(slabs),
             ksplit, npad, kpad, n, hx, wx, hy, wy, _stream())
    else:
        am1 = am2 = amd = None
        if _split_ok(x1, n * hy * wy * cdy * (c1 + c2)) and dy.dtype == torch.float32:
            am1, amd = amax_slot(x1), amax_slot(dy)
            am2 = None if x2 is None else amax_slot(x2)
        call("mia_conv_wgrad", mode, dtype, _p(x1), c1, _p(x2), c2, _p(dy), cdy, _p(slabs), ksplit, npad, kpad, n, hx, wx, hy,
             wy, _p(am1), _p(am2), _p(amd), _stream())
    grad = out if out is not None else torch.empty(grad_shape, device=x1.device, dtype=torch.float32)
    call("mia_wgrad_reduce", _p(slabs), ksplit, taps, npad, kpad, _p(grad), nn, kk, 0, _stream())
    return grad


def colsum(x: torch.Tensor) -> torch.Tensor:
    c = x.shape[-1]
    p = x.numel() // c
    ws = torch.empty(lib().mia_colsum_workspace(_c_i64(p), c), device=x.device, dtype=torch.float32)
    out = torch.empty(c, device=x.device, dtype=torch.float32)
    call("mia_colsum", _p(x), _dt(x), _c_i64(p), c, _p(ws), _p(out), 0, _stream())
    return out


# Gradient destinations: a flat optimizer registers, per parameter, the slice of its flat gradient buffer; backward nodes
# then let their kernels write the parameter gradient THERE and return that view, which autograd adopts as .grad without a
# copy or an accumulation kernel (82 tiny `add` launches per step otherwise).  Unregistered parameters get fresh tensors.
_GRAD_DEST = {}
_GRAD_CLAIMED = set()  # id(param) whose slice a backward node of the CURRENT accumulation already writes


def register_grad_dest(param: torch.Tensor, flat: torch.Tensor, offset: int) -> None:
    _GRAD_DEST[id(param)] = (__import__("weakref").ref(param), flat, offset, param.numel(), tuple(param.shape))
    _GRAD_CLAIMED.discard(id(param))


def unregister_grad_dest(param: torch.Tensor) -> None:
    _GRAD_DEST.pop(id(param), None)
    _GRAD_CLAIMED.discard(id(param))


def release_grad_dest(param: torch.Tensor) -> None:
    """The parameter's gradient has been accumulated (post-accumulate hook) or dropped (zero_grad): its slice may be
    handed out again."""
    _GRAD_CLAIMED.discard(id(param))


def grad_dest(param: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A FRESH view of the registered slice (autograd only adopts a gradient nobody else references), handed out to at
    most ONE backward node per accumulation: a second node that uses the same parameter in the same backward pass (model
    called twice before one backward, shared weights) gets None -> a fresh tensor, and autograd sums the two; without the
    claim both kernels would write the same slice and autograd would add two aliased views (2x the last writer)."""
    if param is None:
        return None
    hit = _GRAD_DEST.get(id(param))
    if hit is None:
        return None
    if hit[0]() is not param:
        _GRAD_DEST.pop(id(param), None)
        _GRAD_CLAIMED.discard(id(param))
        return None
    if param.grad is not None:
        return None  # a gradient is already there (accumulation over several backward passes): autograd must ADD to it
    if id(param) in _GRAD_CLAIMED:
        return None
    _GRAD_CLAIMED.add(id(param))
    _, flat, off, n, shape = hit
    return flat[off:off + n].view(shape)


# Column sums that a producer kernel already had for free (conv epilogue statistics), keyed by the tensor they describe;
# consumed by the next backward node that would otherwise re-read that tensor just to sum it.
_COLSUM_HINT = {}


def _hint_colsum(t: torch.Tensor, sums: torch.Tensor) -> None:
    if len(_COLSUM_HINT) > 16:
        _COLSUM_HINT.clear()
    _COLSUM_HINT[t.data_ptr()] = (tuple(t.shape), t._version, sums)


def _take_colsum(t: torch.Tensor) -> Optional[torch.Tensor]:
    hit = _COLSUM_HINT.pop(t.data_ptr(), None)
    if hit is not None and hit[0] == tuple(t.shape) and hit[1] == t._version:
        return hit[2]
    return None


# Norm-backward partial sums computed by the epilogue of the input-gradient conv that produced a dz tensor (mia_conv_mma_cr):
# keyed by that tensor's storage address; the producing block's backward takes them and skips its own reduction pass.
_CR_HINT = {}
# (experiment of round 4, measured +-0 on the cfg3 step; the kernels exist only in probe builds of the library, -DMIA_EXPERIMENTS)
FUSE_CR = __import__('os').environ.get('MIA_FUSE_CR', '0') != '0'


def _hint_cr(dz: torch.Tensor, partials: torch.Tensor) -> None:
    if len(_CR_HINT) > 64:
        _CR_HINT.clear()
    _CR_HINT[dz.data_ptr()] = (partials, tuple(dz.shape))


def _take_cr(dz: torch.Tensor) -> Optional[torch.Tensor]:
    h = _CR_HINT.pop(dz.data_ptr(), None)
    return h[0] if (h is not None and h[1] == tuple(dz.shape)) else None


# Skip tensors have two consumers, so their gradient arrives in two pieces.  The decoder block (which runs first in backward) leaves
# its piece here, keyed by the skip tensor's storage address; the stride-2 block of the next encoder level then ADDS its piece into
# that tensor inside its input-gradient kernel (mia_conv_mma_acc) and returns no gradient of its own, so the skip block's norm
# backward reads one gradient tensor instead of two.
_ACC_HINT = {}
FUSE_ACC = __import__('os').environ.get('MIA_FUSE_ACC', '1') != '0'  # A/B knob


def _hint_acc(x: torch.Tensor, dx: torch.Tensor) -> None:
    if FUSE_ACC and dx is not None:
        _ACC_HINT[x.data_ptr()] = (dx, tuple(x.shape))


def _take_acc(x: torch.Tensor) -> Optional[torch.Tensor]:
    h = _ACC_HINT.pop(x.data_ptr(), None)
    return h[0] if (h is not None and h[1] == tuple(x.shape) and h[0].dtype == x.dtype) else None


def clear_hints() -> None:
    """Drop every producer -> consumer hint and gradient-destination claim (a step that was recorded but never ran -- a failed graph
    capture -- leaves them pointing at tensors nobody wrote)."""
    _COLSUM_HINT.clear()
    _CR_HINT.clear()
    _ACC_HINT.clear()
    _GRAD_CLAIMED.clear()


def cr_supported(dtype, cin: int, cout: int, h: int, w: int) -> bool:
    return (FUSE_CR and dtype == torch.bfloat16 and hasattr(lib(), "mia_conv_mma_cr")
            and bool(lib().mia_conv_cr_supported(CONV_G3S1, BF16, cout, cin, h, w)))


_COL_SLABS = int(__import__('os').environ.get('MIA_COL_SLABS', '64'))


def _slabs_for(hw: int) -> int:
    return max(1, min(_COL_SLABS, hw // 1024))


# ------------------------------------------------------------------ Dropout2d masks
class StepDyn:
    """Per-step scalars of a train step that is replayed from a captured hipGraph (training/engine.py graph mode).  Kernel
    arguments are frozen at capture, so what changes every iteration lives in 32 bytes of device memory that the host rewrites
    (one one-thread launch, `mia_step_dyn_set`) before each replay: fp32 {lr, 1 - beta1^t, 1 - beta2^t, first step} for `mia_optim_step_dyn`
    and u64 {Philox seed, base offset} for `mia_dropout_mask_dyn` (each mask launch adds its fixed distance from the base).
    While `ops._STEP_DYN` is set, `optim_step` and `dropout_mask` take these variants."""

    def __init__(self, device):
        self.dev = torch.zeros(32, dtype=torch.uint8, device=device)
        self.rng_span = 0   # Philox offsets one step consumes (counted while capturing)

    def set(self, lr: float, bc1: float, bc2: float, first: bool, seed: int, base_offset: int) -> None:
        """One one-thread launch whose arguments carry the values (copied at enqueue time: the host may run any number of
        replays ahead of the device, which a pinned staging buffer + async copy would not survive)."""
        call("mia_step_dyn_set", _p(self.dev), _c_float(lr), _c_float(bc1), _c_float(bc2), int(bool(first)),
             ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(base_offset), _stream())

    @property
    def f32_ptr(self):
        return ctypes.c_void_p(self.dev.data_ptr())

    @property
    def u64_ptr(self):
        return ctypes.c_void_p(self.dev.data_ptr() + 16)


_STEP_DYN: Optional[StepDyn] = None


def dropout_mask(numel: int, keep: float, device) -> torch.Tensor:
    """`numel` Dropout2d channel multipliers in {0, 1/keep} (reference blocks.py:92-96) from the library's Philox kernel.
    Keyed by the device generator's (seed, offset) -- the pair torch.manual_seed / set_rng_state control, so runs are
    reproducible and per-rank seeds give per-rank masks (al_trainer.py:282-288) -- and the offset is advanced on the host
    like a PyTorch CUDA RNG consumer would: no host random numbers are drawn and no PyTorch kernel runs."""
    dev = torch.device(device)
    if _STEP_DYN is not None:  # captured step: seed / base offset come from device memory, this launch sits `rng_span` past the base
        out = torch.empty(numel, device=dev, dtype=torch.float32)
        call("mia_dropout_mask_dyn", _p(out), _c_i64(numel), _c_float(keep), _STEP_DYN.u64_ptr, ctypes.c_uint64(_STEP_DYN.rng_span), _stream())
        _STEP_DYN.rng_span += 4 * ((numel + 3) // 4)
        return out
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    seed, off = int(gen.initial_seed()), int(gen.get_offset())
    gen.set_offset(off + 4 * ((numel + 3) // 4))
    out = torch.empty(numel, device=dev, dtype=torch.float32)
    call("mia_dropout_mask", _p(out), _c_i64(numel), _c_float(keep), ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(off),
         _stream())
    return out


# ------------------------------------------------------------------ PlainBlock: conv3x3 -> dropout2d -> norm -> lrelu
class NormCfg:
    __slots__ = ("mode", "training", "eps", "momentum", "running_mean", "running_var", "num_batches", "drop_scale", "sync")

    def __init__(self, mode, training, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, num_batches=None,
                 drop_scale=None, sync=None):
        self.mode, self.training, self.eps, self.momentum = mode, training, eps, momentum
        self.running_mean, self.running_var, self.num_batches, self.drop_scale = running_mean, running_var, num_batches, drop_scale
        self.sync = sync  # BatchSync or None


class BatchSync:
    """Process group for synchronised batch-norm statistics (one process per GPU)."""

    def __init__(self, process_group=None):
        import torch.distributed as dist
        self.dist, self.pg = dist, process_group
        self.world = dist.get_world_size(process_group)

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.world,) + tuple(local.shape), device=local.device, dtype=local.dtype)
        self.dist.all_gather([out[r] for r in range(self.world)], local, group=self.pg)  # works on gloo and RCCL
        return out

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.pg)
        return t


def _norm_finalize(cfg: NormCfg, stats, gamma, beta, n, cout, hw, coefs):
    """Per-(n,c) coefficients from the conv-epilogue partials; batch statistics cover every rank when cfg.sync is set."""
    sync = cfg.sync if (cfg.sync is not None and cfg.mode == NORM_BATCH and cfg.training and cfg.sync.world > 1) else None
    if sync is None:
        call("mia_norm_finalize", _p(stats), n, 0 if stats is None else stats.shape[1], cout, _c_i64(hw), cfg.mode, int(cfg.training),
             _p(cfg.drop_scale), _p(gamma.detach()), _p(beta.detach()), _c_float(cfg.eps), _c_float(cfg.momentum),
             _p(cfg.running_mean), _p(cfg.running_var), _p(cfg.num_batches), _p(coefs[0]), _p(coefs[1]), _p(coefs[2]),
             _p(coefs[3]), _p(coefs[4]), _stream())
        return None
    local = torch.empty((3, cout), device=stats.device, dtype=torch.float32)
    call("mia_bn_sync_local_stats", _p(stats), n, stats.shape[1], cout, _c_i64(hw), _p(cfg.drop_scale), _p(coefs[0]),
         _p(coefs[1]), _p(local), _stream())
    gathered = sync.all_gather(local)
    call("mia_norm_finalize_sync", _p(gathered), sync.world, n, cout, _c_i64(hw), _p(cfg.drop_scale), _p(gamma.detach()),
         _p(beta.detach()), _c_float(cfg.eps), _c_float(cfg.momentum), _p(cfg.running_mean), _p(cfg.running_var),
         _p(cfg.num_batches), _p(coefs[0]), _p(coefs[1]), _p(coefs[2]), _p(coefs[3]), _p(coefs[4]), _stream())
    return sync


class LazyAct:
    """Output of a PlainBlock whose normalisation + LeakyReLU (blocks.py:98-102) is NOT materialised: `y` is the raw conv
    output (it carries the autograd history and, by convention, the gradient delivered to it is d loss / d z), `coefs`
    the block's [5, N, C] coefficient table (rows 2 / 3 = scale / shift with Dropout2d folded in).  The next block's conv
    and weight gradient form z = lrelu(scale * y + shift) on load (`mia_conv_mma_nl`, `mia_conv_wgrad_nl`)."""
    __slots__ = ("y", "coefs", "slope")

    def __init__(self, y, coefs, slope):
        self.y, self.coefs, self.slope = y, coefs, slope

    @property
    def shape(self):
        return self.y.shape

    @property
    def dtype(self):
        return self.y.dtype

    @property
    def device(self):
        return self.y.device

    def materialize(self) -> torch.Tensor:
        return LazyMaterializeFn.apply(self.y, self.coefs, self.slope)


class LazyMaterializeFn(torch.autograd.Function):
    """z = lrelu(scale * y + shift) as a tensor (a consumer outside the fused kernels' contract).  The gradient passes
    through unchanged: the producer expects d loss / d z on its raw-output handle (see LazyAct)."""

    @staticmethod
    def forward(ctx, y, coefs, slope):
        n, h, w, c = y.shape
        z = torch.empty_like(y)
        call("mia_norm_act_fwd", _p(y), _p(z), _dt(y), _p(coefs[2]), _p(coefs[3]), n, _c_i64(h * w), c, _c_float(slope), _p(_amax_new(z)), _stream())
        return z

    @staticmethod
    def backward(ctx, dz):
        return dz, None, None


FUSE_HEAD_W = __import__('os').environ.get('MIA_FUSE_HEAD_W', '1') != '0'  # A/B knob: head dW / db in the norm-backward reduction pass
FUSE_STEM_BWD = __import__('os').environ.get('MIA_FUSE_STEM_BWD', '1') != '0'  # A/B knob: the stem's backward apply pass folded into its weight gradient
FUSE_NL = __import__('os').environ.get('MIA_FUSE_NL', '1') != '0'  # A/B knob: 0 = every block materialises its activation


def nl_supported(dtype, cin: int, cout: int, h: int, w: int, train: bool) -> bool:
    """Can a stride-1 3x3 block with these shapes consume its predecessor's raw output (normalise-on-load)?  bf16 only: the 64 -> 64
    register-staged kernels (the fp32 form of round 4 measured +-0 and is gone: the split-f16 products need the maximum of the
    NORMALISED tensor, which nobody has computed when the raw output is loaded)."""
    if not FUSE_NL or dtype != torch.bfloat16:
        return False
    if not lib().mia_conv_nl_supported(CONV_G3S1, BF16, cin, cout, h, w):
        return False
    return (not train) or bool(lib().mia_wgrad_nl_supported(WGRAD_3S1, BF16, cin, cout))


class PlainBlockFn(torch.autograd.Function):
    """Fused reference PlainBlock (src/models/unet/blocks.py:66-105) on NHWC tensors.

    inputs x1 [N,H,W,C1] (+ optional x2 [N,H,W,C2], concatenated along C: unet.py:213)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, gamma, beta, stride: int, cfg: NormCfg, out_dtype=None, slope: float = LRELU_SLOPE,
                dup: bool = False, lazy: bool = False, nl_coefs=None, nl_slope: float = LRELU_SLOPE):
        """lazy=True: the norm + LeakyReLU pass is skipped and (y, coefs) is returned for a `LazyAct` (the next block
        normalises on load).  nl_coefs: x1 is such a raw output of the previous block (its coefficient table).
        dup=True returns the output TWICE (two tensors on one storage): a skip tensor has two consumers, and handing
        each its own output makes autograd deliver their gradients separately to backward, which sums them on load inside
        the norm kernels instead of autograd launching an `add` over the full activation."""
        ctx.dup = dup
        # x1 is one of the two views a `dup=True` block handed out (tagged by the model): exactly one gradient will ever arrive
        # for it, so the two consumers may share one gradient tensor (see _ACC_HINT).  Untagged tensors (e.g. a ResidualBlock's
        # output, whose several gradient pieces autograd itself adds up) never take that path.
        ctx.dup_in = bool(getattr(x1, "_mia_dup", False))
        ctx.set_materialize_grads(False)
        _need_dev(x1, x2, weight)
        x1 = x1.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        out_dtype = out_dtype or x1.dtype
        cout_ = weight.shape[0]
        assert not (lazy and dup)
        ctx.nl = None
        stem = (weight.shape[1] == 1 and x2 is None and stride == 1 and x1.shape[3] == 1 and not x1.requires_grad
                and cout_ % (8 if out_dtype == torch.bfloat16 else 4) == 0 and cout_ <= 256 and nl_coefs is None)
        if stem:
            z = PlainBlockFn._stem_forward(ctx, x1, weight, bias, gamma, beta, cfg, out_dtype, slope, lazy)
            if lazy:
                return z
            return (z, _dup_view(z)) if dup else z
        if x1.dtype != out_dtype:
            x1 = cast_nhwc(x1, out_dtype)
        if x2 is not None and (x2.shape[:3] != x1.shape[:3] or x2.dtype != x1.dtype):
            # same failure point as torch.cat([skip, up], 1) in the reference (unet.py:213)
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1: {tuple(x1.shape)} vs {tuple(x2.shape)} (NHWC)")
        dtype = _dt(x1)
        n, h, w, c1 = x1.shape
        cout = weight.shape[0]
        if weight.shape[1] != c1 + (0 if x2 is None else x2.shape[3]):
            raise RuntimeError(f"conv weight expects {weight.shape[1]} input channels, got {c1 + (0 if x2 is None else x2.shape[3])}")
        ho, wo = ((h + 1) // 2, (w + 1) // 2) if stride == 2 else (h, w)
        wp, npad, kpad = pack_cache(weight).get(weight, dtype, n_from_d0=True)
        mode = CONV_G3S2 if stride == 2 else CONV_G3S1
        fixed = cfg.mode == NORM_BATCH and not cfg.training  # eval batch norm: running statistics, no batch sums needed
        nl = None if nl_coefs is None else (nl_coefs, float(nl_slope))
        if nl is not None and (x2 is not None or stride != 1):
            raise MiaError("normalise-on-load serves one-source stride-1 blocks only")
        y, _, stats = conv_mma(mode, x1, x2, wp, npad, kpad, False, bias.detach().float(), cout, (ho, wo), want_stats=not fixed,
                               nl=nl)
        dev = x1.device
        coefs = torch.empty((5, n, cout), device=dev, dtype=torch.float32)  # xa, xb, scale, shift, sum_y
        ctx.sync = _norm_finalize(cfg, stats, gamma, beta, n, cout, ho * wo, coefs)
        ctx.stride, ctx.mode, ctx.fixed, ctx.stem, ctx.slope = stride, cfg.mode, fixed, False, slope
        ctx.small = (bias, beta)  # identities only (gradient destinations); values are not needed in backward
        ctx.nl_slope = float(nl_slope)
        if lazy:  # the consumer normalises on load: no apply pass, no activation tensor
            ctx.save_for_backward(x1, x2, y, coefs, weight, gamma, nl_coefs)
            ctx.mark_non_differentiable(coefs)
            return y, coefs
        z = torch.empty_like(y)
        call("mia_norm_act_fwd", _p(y), _p(z), dtype, _p(coefs[2]), _p(coefs[3]), n, _c_i64(ho * wo), cout,
             _c_float(slope), _p(_amax_new(z)), _stream())
        ctx.save_for_backward(x1, x2, y, coefs, weight, gamma, nl_coefs)
        return (z, _dup_view(z)) if dup else z

    @staticmethod
    def _norm_act(ctx, y, stats, gamma, beta, cfg, n, cout, hw, slope=LRELU_SLOPE):
        dev = y.device
        coefs = torch.empty((5, n, cout), device=dev, dtype=torch.float32)  # xa, xb, scale, shift, sum_y
        ctx.sync = _norm_finalize(cfg, stats, gamma, beta, n, cout, hw, coefs)
        z = torch.empty_like(y)
        call("mia_norm_act_fwd", _p(y), _p(z), _dt(y), _p(coefs[2]), _p(coefs[3]), n, _c_i64(hw), cout, _c_float(slope),
             _p(_amax_new(z)), _stream())
        return z, coefs

    @staticmethod
    def _stem_forward(ctx, x1, weight, bias, gamma, beta, cfg, out_dtype, slope, lazy=False):
        n, h, w, _ = x1.shape
        cout = weight.shape[0]
        dev = x1.device
        y = torch.empty((n, h, w, cout), device=dev, dtype=out_dtype)
        stats = torch.empty((n, lib().mia_stem_slabs(), cout, 2), device=dev, dtype=torch.float32)
        w2 = weight.detach().reshape(cout, 9)
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        call("mia_stem_fwd", _p(x1), _dt(x1), _p(w2), _p(bias.detach()), _p(y), _dt(y), _p(stats), n, h, w, cout, _stream())
        ctx.stride, ctx.mode, ctx.fixed, ctx.stem, ctx.slope = 1, cfg.mode, cfg.mode == NORM_BATCH and not cfg.training, True, slope
        ctx.small = (bias, beta)
        ctx.nl_slope = LRELU_SLOPE
        if lazy:
            coefs = torch.empty((5, n, cout), device=dev, dtype=torch.float32)
            ctx.sync = _norm_finalize(cfg, stats, gamma, beta, n, cout, h * w, coefs)
            ctx.save_for_backward(x1, None, y, coefs, weight, gamma, None)
            ctx.mark_non_differentiable(coefs)
            return y, coefs
        z, coefs = PlainBlockFn._norm_act(ctx, y, stats, gamma, beta, cfg, n, cout, h * w, slope)
        ctx.save_for_backward(x1, None, y, coefs, weight, gamma, None)
        return z

    @staticmethod
    def backward(ctx, *grads):
        x1, x2, y, coefs, weight, gamma, nl_coefs = ctx.saved_tensors
        dtype = _dt(y)
        n, ho, wo, cout = y.shape
        pieces = [g_.contiguous() for g_ in grads if g_ is not None]
        if not pieces:
            return (None,) * 14
        dz, dz2 = pieces[0], (pieces[1] if len(pieces) > 1 else None)
        if dz2 is not None and not (lib().mia_norm_two_piece_ok(dtype, cout) and dz.data_ptr() % 16 == 0 and dz2.data_ptr() % 16 == 0):
            dz, dz2 = dz + dz2, None  # shapes outside the vectorised kernels: one explicit sum
        dev = y.device
        hw = ho * wo
        slabs = _slabs_for(hw)
        part = torch.empty((n, slabs, cout, 2), device=dev, dtype=torch.float32)
        cc = torch.empty((2, n, cout), device=dev, dtype=torch.float32)
        dgb = torch.empty((3, cout), device=dev, dtype=torch.float32)
        bias_p, beta_p = ctx.small
        dgamma, dbeta, dbias = grad_dest(gamma), grad_dest(beta_p), grad_dest(bias_p)
        dgamma = dgb[0] if dgamma is None else dgamma
        dbeta = dgb[1] if dbeta is None else dbeta
        dbias = dgb[2] if dbias is None else dbias
        pre = _take_cr(dz) if (dz2 is None and ctx.sync is None) else None  # reduction already done by the conv that produced dz
        if ctx.stem and ctx.sync is None and FUSE_STEM_BWD:
            # the stem has no input gradient: its weight gradient is the only consumer of dy and forms it on load -- no apply pass
            if pre is not None:
                call("mia_norm_act_bwd_pre", None, None, None, dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                     _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout, ctx.mode, int(ctx.fixed), _c_float(ctx.slope),
                     pre.shape[1], _p(pre), _p(cc[0]), _p(cc[1]), _p(dgamma), _p(dbeta), _p(dbias), 0, None, _stream())
            else:
                call("mia_norm_bwd_sums", _p(dz), _p(dz2), _p(y), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                     _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout, ctx.mode, int(ctx.fixed), _c_float(ctx.slope), slabs,
                     _p(part), _p(cc[0]), _p(cc[1]), _p(dgamma), _p(dbeta), _p(dbias), 0, _stream())
            if dz2 is not None:
                dz = dz + dz2
            ws = torch.empty(lib().mia_stem_wgrad_workspace(cout), device=dev, dtype=torch.float32)
            dw = grad_dest(weight)
            if dw is None:
                dw = torch.empty(weight.shape, device=dev, dtype=torch.float32)
            call("mia_stem_wgrad_fused", _p(x1), _dt(x1), _p(dz), _p(y), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                 _p(cc[0]), _p(cc[1]), _c_float(ctx.slope), _p(ws), _p(dw), n, ho, wo, cout, 0, _stream())
            return (None, None, dw, dbias, dgamma, dbeta) + (None,) * 8
        dy = torch.empty_like(y)
        if pre is not None:
            call("mia_norm_act_bwd_pre", _p(dz), _p(y), _p(dy), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                 _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout, ctx.mode, int(ctx.fixed), _c_float(ctx.slope),
                 pre.shape[1], _p(pre), _p(cc[0]), _p(cc[1]), _p(dgamma), _p(dbeta), _p(dbias), 0, _p(_amax_new(dy)), _stream())
        elif ctx.sync is None:
            call("mia_norm_act_bwd", _p(dz), _p(dz2), _p(y), _p(dy), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
                 _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout, ctx.mode, int(ctx.fixed), _c_float(ctx.slope), slabs,
                 _p(part), _p(cc[0]), _p(cc[1]), _p(dgamma), _p(dbeta), _p(dbias), 0, _p(_amax_new(dy)), _stream())
        else:  # synchronised batch norm: the group means of g and g*xhat cover every rank's shard
            tot = torch.empty((3, cout), device=dev, dtype=torch.float32)
            call("mia_norm_act_bwd_reduce", _p(dz), _p(dz2), _p(y), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]), n,
                 _c_i64(hw), cout, _c_float(ctx.slope), slabs, _p(part), _p(cc[0]), _p(cc[1]), _p(tot), _stream())
            ctx.sync.all_reduce_sum(tot)
            call("mia_norm_act_bwd_apply_sync", _p(dz), _p(dz2), _p(y), _p(dy), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]),
                 _p(coefs[1]), _p(coefs[4]), n, _c_i64(hw), cout, _c_float(ctx.slope), _p(cc[0]), _p(cc[1]), _p(tot),
                 _p(dgamma), _p(dbeta), _p(dbias), 0, _p(_amax_new(dy)), _stream())
        cin = weight.shape[1]
        if ctx.stem:
            ws = torch.empty(lib().mia_stem_wgrad_workspace(cout), device=dev, dtype=torch.float32)
            dw = grad_dest(weight)
            if dw is None:
                dw = torch.empty(weight.shape, device=dev, dtype=torch.float32)
            call("mia_stem_wgrad", _p(x1), _dt(x1), _p(dy), dtype, _p(ws), _p(dw), n, ho, wo, cout, 0, _stream())
            return (None, None, dw, dbias, dgamma, dbeta) + (None,) * 8
        wmode = WGRAD_3S2 if ctx.stride == 2 else WGRAD_3S1
        dw = conv_wgrad(wmode, x1, x2, dy, weight.shape, cout, cin, out=grad_dest(weight),
                        nl=None if nl_coefs is None else (nl_coefs, ctx.nl_slope))
        dx1 = dx2 = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            wb, npad, kpad = pack_cache(weight).get(weight, dtype, n_from_d0=False)
            c1 = x1.shape[3]
            split = c1 if x2 is not None else None
            if ctx.stride == 2:
                other = _take_acc(x1) if (x2 is None and ctx.dup_in) else None
                if other is not None and lib().mia_conv_acc_supported(CONV_T3S2, dtype, cout, cin):
                    # the other consumer of x1 (a skip tensor) has already written its gradient piece: add ours into it
                    amw = getattr(wb, "_mia_amax", None) if _split_ok(dy, n * ho * wo * cout * cin) else None
                    call("mia_conv_mma_acc", CONV_T3S2, dtype, _p(dy), cout, _p(wb), npad, kpad, 0, _p(other), cin, n, ho, wo,
                         x1.shape[1], x1.shape[2], _p(None if amw is None else amax_slot(dy)), _p(None if amw is None else amw[0]),
                         _p(None if amw is None else getattr(wb, "_mia_split", None)), _stream())
                    return (None, None, dw, dbias, dgamma, dbeta) + (None,) * 8
                dx1, dx2, _ = conv_mma(CONV_T3S2, dy, None, wb, npad, kpad, False, None, cin, (x1.shape[1], x1.shape[2]),
                                       out_split=split)
            else:
                # decoder block behind a ConvTranspose2d: dx2 is that layer's output gradient and its per-channel sum is
                # the transposed conv's bias gradient -- the epilogue statistics deliver it without another pass over dx2
                want = x2 is not None and ctx.needs_input_grad[1]
                if nl_coefs is not None and cr_supported(y.dtype, cin, cout, ho, wo):
                    # x1 is the previous block's raw output: its norm-backward reduction rides in this conv's epilogue
                    dx1, _, partials = conv_mma(CONV_G3S1, dy, None, wb, npad, kpad, True, None, cin, (ho, wo),
                                                cr=(x1, nl_coefs, ctx.nl_slope))
                    _hint_cr(dx1, partials)
                    return (dx1, None, dw, dbias, dgamma, dbeta) + (None,) * 8
                dx1, dx2, st = conv_mma(CONV_G3S1, dy, None, wb, npad, kpad, True, None, cin, (ho, wo), want_stats=want,
                                        out_split=split)
                if want:
                    sums = colsum(st.view(-1, 2 * cin)).view(cin, 2)[c1:, 0]
                    _hint_colsum(dx2, sums)
                if x2 is not None and ctx.needs_input_grad[0] and ctx.dup_in:
                    _hint_acc(x1, dx1)  # x1 is the skip tensor: its other consumer may add its gradient piece into dx1
        return (dx1, dx2, dw, dbias, dgamma, dbeta) + (None,) * 8


class PlainBlockHeadFn(torch.autograd.Function):
    """Last decoder PlainBlock (stride 1, one input) fused with the 1x1 segmentation head behind it
    (src/models/unet/blocks.py:66-105 + unet.py:176): logits = W_head * lrelu(norm(conv(x))) + b_head.

    The block's activated output has exactly one consumer, so it is never materialised: the head recomputes it from the
    raw conv output on load (`mia_head_norm_fwd` / `mia_head_norm_wgrad`) and the norm backward recomputes the head's
    input gradient W^T dlogits on the fly (`mia_norm_act_bwd_head`).  Saves the forward apply pass, the head's
    input-gradient kernel and two reads of that gradient (about 4.3 GB of HBM traffic per step on cfg3)."""

    @staticmethod
    def eligible(x1, weight, head_w, cfg: NormCfg, dtype) -> bool:
        if cfg.sync is not None or x1.ndim != 4 or weight.shape[1] == 1:
            return False
        n, h, w, _ = x1.shape
        cout, k1 = weight.shape[0], head_w.shape[0]
        return cout % 32 == 0 and lib().mia_head_norm_eligible(_dt(dtype), n, _c_i64(h * w), cout, k1) > 0

    @staticmethod
    def forward(ctx, x1, weight, bias, gamma, beta, cfg: NormCfg, head_w, head_b, slope: float = LRELU_SLOPE, nl_coefs=None,
                nl_slope: float = LRELU_SLOPE):
        """nl_coefs: x1 is the previous block's RAW conv output with that coefficient table (normalise-on-load)."""
        _need_dev(x1, weight, head_w)
        x1 = x1.contiguous()
        dtype = _dt(x1)
        n, h, w, c1 = x1.shape
        cout, k1 = weight.shape[0], head_w.shape[0]
        if weight.shape[1] != c1:
            raise RuntimeError(f"conv weight expects {weight.shape[1]} input channels, got {c1}")
        wp, npad, kpad = pack_cache(weight).get(weight, dtype, n_from_d0=True)
        fixed = cfg.mode == NORM_BATCH and not cfg.training
        y, _, stats = conv_mma(CONV_G3S1, x1, None, wp, npad, kpad, False, bias.detach().float(), cout, (h, w), want_stats=not fixed,
                               nl=None if nl_coefs is None else (nl_coefs, float(nl_slope)))
        coefs = torch.empty((5, n, cout), device=x1.device, dtype=torch.float32)  # xa, xb, scale, shift, sum_y
        ctx.sync = _norm_finalize(cfg, stats, gamma, beta, n, cout, h * w, coefs)
        ctx.nl_slope = float(nl_slope)
        w2 = head_w.detach().reshape(k1, cout).contiguous()
        logits = torch.empty((n, h, w, k1), device=x1.device, dtype=torch.float32)
        call("mia_head_norm_fwd", _p(y), dtype, _p(coefs[2]), _p(coefs[3]), _c_float(slope), _p(w2), _p(head_b.detach()),
             _p(logits), n, _c_i64(h * w), cout, k1, _c_i64(h * w * k1), _c_i64(1), _c_i64(k1), _stream())
        ctx.save_for_backward(x1, y, coefs, weight, gamma, head_w, nl_coefs)
        ctx.mode, ctx.fixed, ctx.slope = cfg.mode, cfg.mode == NORM_BATCH and not cfg.training, slope
        ctx.small = (bias, beta, head_b)
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dl):
        x1, y, coefs, weight, gamma, head_w, nl_coefs = ctx.saved_tensors
        bias_p, beta_p, head_b = ctx.small
        dtype = _dt(y)
        n, h, w, cout = y.shape
        k1, cin = head_w.shape[0], weight.shape[1]
        dev, hw = y.device, h * w
        if dl.dtype != torch.float32:
            dl = dl.float()
        st = _pix_strides(dl)
        if st is None or st[0] != hw * st[2]:
            dl = dl.contiguous()
            st = _pix_strides(dl)
        w2 = head_w.detach().reshape(k1, cout).contiguous()
        # head parameters
        dwh, dbh = grad_dest(head_w), grad_dest(head_b)
        dwh = torch.empty((k1, cout), device=dev, dtype=torch.float32) if dwh is None else dwh
        dbh = torch.empty(k1, device=dev, dtype=torch.float32) if dbh is None else dbh
        slabs = _slabs_for(hw)
        part = torch.empty((n, slabs, cout, 2), device=dev, dtype=torch.float32)
        cc = torch.empty((2, n, cout), device=dev, dtype=torch.float32)
        dgb = torch.empty((3, cout), device=dev, dtype=torch.float32)
        dgamma, dbeta, dbias = grad_dest(gamma), grad_dest(beta_p), grad_dest(bias_p)
        dgamma = dgb[0] if dgamma is None else dgamma
        dbeta = dgb[1] if dbeta is None else dbeta
        dbias = dgb[2] if dbias is None else dbias
        dy = torch.empty_like(y)
        if FUSE_HEAD_W and lib().mia_head_w_supported(dtype, cout, k1):
            # head dW / db and the block's norm-backward sums in ONE pass over (dl, y); then the apply pass with dz = W^T dl recomputed
            ws = torch.empty(n * slabs * k1 * (cout + 1), device=dev, dtype=torch.float32)
            call("mia_norm_act_bwd_head_w", _p(dl), _p(w2), k1, _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), _p(y), _p(dy), dtype,
                 _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]), _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout,
                 ctx.mode, int(ctx.fixed), _c_float(ctx.slope), slabs, _p(part), _p(cc[0]), _p(cc[1]), _p(dgamma), _p(dbeta),
                 _p(dbias), 0, _p(ws), _p(dwh), _p(dbh), 0, _p(_amax_new(dy)), _stream())
        else:
            ws = torch.empty(lib().mia_head_bwd_workspace(cout, k1), device=dev, dtype=torch.float32)
            call("mia_head_norm_wgrad", _p(dl), _p(y), dtype, _p(coefs[2]), _p(coefs[3]), _c_float(ctx.slope), _p(dwh), _p(dbh),
                 _p(ws), n, _c_i64(hw), cout, k1, _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), 0, _stream())
            # norm backward with dz = W^T dl recomputed
            call("mia_norm_act_bwd_head", _p(dl), _p(w2), k1, _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), _p(y), _p(dy), dtype,
                 _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]), _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout,
                 ctx.mode, int(ctx.fixed), _c_float(ctx.slope), slabs, _p(part), _p(cc[0]), _p(cc[1]), _p(dgamma), _p(dbeta),
                 _p(dbias), 0, _p(_amax_new(dy)), _stream())
        dw = conv_wgrad(WGRAD_3S1, x1, None, dy, weight.shape, cout, cin, out=grad_dest(weight),
                        nl=None if nl_coefs is None else (nl_coefs, ctx.nl_slope))
        dx1 = None
        if ctx.needs_input_grad[0]:
            wb, npad, kpad = pack_cache(weight).get(weight, dtype, n_from_d0=False)
            if nl_coefs is not None and cr_supported(y.dtype, cin, cout, h, w):
                dx1, _, partials = conv_mma(CONV_G3S1, dy, None, wb, npad, kpad, True, None, cin, (h, w), cr=(x1, nl_coefs, ctx.nl_slope))
                _hint_cr(dx1, partials)
            else:
                dx1, _, _ = conv_mma(CONV_G3S1, dy, None, wb, npad, kpad, True, None, cin, (h, w))
        return dx1, dw, dbias, dgamma, dbeta, None, dwh.reshape(head_w.shape), dbh, None, None, None


# ------------------------------------------------------------------ ConvTranspose2d(k=2, s=2)
class ConvTranspose2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(cin, cout, 2, 2) (src/models/unet/unet.py:142) as a pointwise MFMA GEMM + pixel shuffle."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_dev(x, weight)
        x = x.contiguous()
        dtype = _dt(x)
        n, h, w, cin = x.shape
        cout = weight.shape[1]
        wp, npad, kpad = pack_cache(weight).get(weight, dtype, n_from_d0=False)  # [tap][co][ci]
        out, _, _ = conv_mma(CONV_T2S2, x, None, wp, npad, kpad, False, bias.detach().float(), cout, (2 * h, 2 * w))
        ctx.save_for_backward(x, weight)
        ctx.small = (bias,)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        dout = dout.contiguous()
        dtype = _dt(x)
        n, h, w, cin = x.shape
        cout = weight.shape[1]
        dbias = _take_colsum(dout)
        if dbias is None:
            dbias = colsum(dout)
        dst = grad_dest(ctx.small[0])
        if dst is not None:  # straight into the flat gradient slice (the hinted sums are a strided view of interleaved statistics)
            call("mia_gather_f32", _p(dbias), _c_i64(dbias.stride(0)), _p(dst), dbias.numel(), _stream())
            dbias = dst
        dw = conv_wgrad(WGRAD_2S2, dout, None, x, weight.shape, cin, cout, out=grad_dest(weight))
        dx = None
        if ctx.needs_input_grad[0]:
            wb, npad, kpad = pack_cache(weight).get(weight, dtype, n_from_d0=True)  # [tap][ci][co]
            dx, _, _ = conv_mma(CONV_G2S2, dout, None, wb, npad, kpad, False, None, cin, (h, w))
        return dx, dw, dbias


# ------------------------------------------------------------------ 1x1 head
def _pix_strides(t: torch.Tensor):
    """(sn, sk, sp) element strides of a logical [B,K,H,W] tensor whose H,W dims collapse, else None."""
    b, k, h, w = t.shape
    sn, sk, sh, sw = t.stride()
    if h == 1 or sh == w * sw:
        return sn, sk, sw
    return None


class HeadFn(torch.autograd.Function):
    """seg_output = Conv2d(c0, K1, 1) (src/models/unet/unet.py:176): NHWC activations -> fp32 logits
    returned as a logical-NCHW view with channels_last strides."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_dev(x, weight)
        x = x.contiguous()
        n, h, w, c0 = x.shape
        k1 = weight.shape[0]
        w2 = weight.detach().reshape(k1, c0).contiguous()
        logits = torch.empty((n, h, w, k1), device=x.device, dtype=torch.float32)
        call("mia_head_fwd", _p(x), _dt(x), _p(w2), _p(bias.detach()), _p(logits), n, _c_i64(h * w), c0, k1,
             _c_i64(h * w * k1), _c_i64(1), _c_i64(k1), _stream())
        ctx.save_for_backward(x, weight)
        ctx.small = (bias,)
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dl):
        x, weight = ctx.saved_tensors
        n, h, w, c0 = x.shape
        k1 = weight.shape[0]
        if dl.dtype != torch.float32:
            dl = dl.float()
        st = _pix_strides(dl)
        if st is None:
            dl = dl.contiguous()
            st = _pix_strides(dl)
        w2 = weight.detach().reshape(k1, c0).contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = grad_dest(weight), grad_dest(ctx.small[0])
        dw = torch.empty((k1, c0), device=x.device, dtype=torch.float32) if dw is None else dw
        db = torch.empty(k1, device=x.device, dtype=torch.float32) if db is None else db
        ws = torch.empty(lib().mia_head_bwd_workspace(c0, k1), device=x.device, dtype=torch.float32)
        call("mia_head_bwd", _p(dl), _p(x), _dt(x), _p(w2), _p(dx), _p(dw), _p(db), _p(ws), n, _c_i64(h * w), c0, k1,
             _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), 0, _stream())
        return dx, dw.reshape(weight.shape), db


# ------------------------------------------------------------------ Dice + CE
def loss_flags(softmax: bool, do_bg: bool, batch: bool, squared: bool) -> int:
    return (LOSS_SOFTMAX if softmax else 0) | (LOSS_DO_BG if do_bg else 0) | (LOSS_BATCH if batch else 0) | \
        (LOSS_SQUARED if squared else 0)


_BAD_FLAGS = {}


def _bad_flags(dev) -> torch.Tensor:
    """Per-device int32[2], zeroed once: allocating and clearing a flag per loss call would put two fill kernels in every step."""
    key = (dev.type, dev.index)
    t = _BAD_FLAGS.get(key)
    if t is None:
        t = torch.zeros(2, device=dev, dtype=torch.int32)
        _BAD_FLAGS[key] = t
    return t


class DiceCEFn(torch.autograd.Function):
    """dice_w * DiceLoss + ce_w * CrossEntropy in one pass over the logits
    (src/losses/dice_loss.py:32-76, src/losses/compound_losses.py:33-49)."""

    @staticmethod
    def forward(ctx, logits, labels, flags: int, smooth: float, dice_w: float, ce_w: float, which: int):
        _need_dev(logits, labels)
        if logits.dtype != torch.float32:
            logits = logits.float()
        st = _pix_strides(logits)
        if st is None:
            logits = logits.contiguous()
            st = _pix_strides(logits)
        b, k1, h, w = logits.shape
        if labels.shape == logits.shape and k1 > 1:
            # dense (already one-hot / soft) target: the reference skips its encoder (dice_loss.py:40-41) and
            # torch's CrossEntropyLoss treats it as class probabilities
            flags |= LOSS_DENSE
            labels = labels.to(torch.float32).contiguous()
        else:
            labels = labels.reshape(b, h, w)
            if labels.dtype != torch.long:
                labels = labels.long()
            labels = labels.contiguous()
        hw = h * w
        slabs = max(1, min(256, hw // 8192))
        dev = logits.device
        ws = torch.empty(lib().mia_dice_ce_workspace(b, k1, slabs), device=dev, dtype=torch.float32)
        sums = torch.empty((b, k1, 3), device=dev, dtype=torch.float32)
        coef = torch.empty((b, k1, 2), device=dev, dtype=torch.float32)  # every entry is written by the finalize kernel
        out = torch.empty(3, device=dev, dtype=torch.float32)
        bad = _bad_flags(dev)  # [0] working flag (set by the pixel kernels, re-armed by finalize), [1] verdict of the latest forward
        call("mia_dice_ce_fwd", _p(logits), _p(labels), b, _c_i64(hw), k1, _c_i64(st[0]), _c_i64(st[1]), _c_i64(st[2]), flags,
             _c_float(smooth), _c_float(dice_w), _c_float(ce_w), slabs, _p(ws), _p(sums), _p(coef), _p(out), _p(bad), _stream())
        ctx.save_for_backward(logits, labels, coef)
        ctx.flags, ctx.dice_w, ctx.ce_w, ctx.st = flags, dice_w, ce_w, st
        ctx.bad = bad
        DiceCEFn.last_bad_label = bad
        DiceCEFn.last_sums = sums
        return out[which]

    @staticmethod
    def backward(ctx, gout):
        logits, labels, coef = ctx.saved_tensors
        b, k1, h, w = logits.shape
        dl = torch.empty_like(logits)  # preserves (dense) strides
        gst = _pix_strides(dl)
        g = gout.reshape(1).float().contiguous()
        st = ctx.st
        call("mia_dice_ce_bwd", _p(logits), _p(labels), _p(coef), _p(g), _p(dl), b, _c_i64(h * w), k1, _c_i64(st[0]),
             _c_i64(st[1]), _c_i64(st[2]), _c_i64(gst[0]), _c_i64(gst[1]), _c_i64(gst[2]), ctx.flags, _c_float(ctx.dice_w),
             _c_float(ctx.ce_w), _stream())
        return dl, None, None, None, None, None, None


DiceCEFn.last_bad_label = None
DiceCEFn.last_sums = None


def check_labels() -> None:
    """Raise if ANY Dice/CE forward on this device since the previous check met a label outside [0, K1) (host sync: call it
    where you already sync, e.g. next to the `loss.item()` you log).  The reference raises at the offending call (scatter
    index error in `DiceLoss._one_hot_encoder`, dice_loss.py:25-30, and the target bound check of CrossEntropyLoss); the
    kernels cannot, so they return NaN losses / NaN gradients for such a batch and keep a STICKY per-device verdict: a clean
    forward in between (validation, a second loss term) does not erase it; this call reads it and clears it.
    Ordering contract: every loss forward whose labels you want checked must have been ISSUED on the current stream before
    this call; losses running on other streams of the same device share the flag (their verdicts OR together)."""
    bad = DiceCEFn.last_bad_label
    if bad is not None and int(bad[1].item()) != 0:
        bad[1].zero_()
        raise MiaError("Dice/CE loss: a label lies outside [0, num_classes] (e.g. 255-valued masks or ignore_index -100); "
                       "the reference raises an index error for such targets")


# ------------------------------------------------------------------ optimizer helpers
def grad_norm(flat_grad: torch.Tensor, max_norm: float, grad_scale: float = 1.0) -> torch.Tensor:
    """out[0] = ||grad_scale*g||_2, out[1] = clip coefficient; stays on device (no sync)."""
    ws = torch.empty(lib().mia_grad_norm_workspace(), device=flat_grad.device, dtype=torch.float32)
    out = torch.empty(2, device=flat_grad.device, dtype=torch.float32)
    call("mia_grad_norm", _p(flat_grad), _c_i64(flat_grad.numel()), _c_float(max_norm), _c_float(grad_scale), _p(ws), _p(out),
         _stream())
    return out


def optim_step(kind: int, param, grad, m, v, lr, beta1, beta2, eps, wd, step: int, clip: Optional[torch.Tensor],
               grad_scale: float = 1.0):
    if _STEP_DYN is not None:  # captured step: lr / bias corrections / first-step flag are read from device memory
        call("mia_optim_step_dyn", _p(param), _p(grad), _p(m), _p(v), _c_i64(param.numel()), kind, _c_float(beta1), _c_float(beta2),
             _c_float(eps), _c_float(wd), _STEP_DYN.f32_ptr, _p(clip), _c_float(grad_scale), _stream())
        return
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    call("mia_optim_step", _p(param), _p(grad), _p(m), _p(v), _c_i64(param.numel()), kind, _c_float(lr), _c_float(beta1),
         _c_float(beta2), _c_float(eps), _c_float(wd), _c_float(bc1), _c_float(bc2), int(step == 1), _p(clip),
         _c_float(grad_scale), _stream())


def global_avg_pool(x_nhwc: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] -> [N,C] fp32 mean over H,W (adaptive_avg_pool2d(.,1).view(B,-1); reference unet.py:87-91)."""
    _need_dev(x_nhwc)
    x = x_nhwc.contiguous()
    n, h, w, c = x.shape
    part = torch.empty((n, 1, c, 2), device=x.device, dtype=torch.float32)
    call("mia_norm_stats", _p(x), _dt(x), n, _c_i64(h * w), c, 1, _p(part), _stream())
    return part[:, 0, :, 0] / float(h * w)


# ------------------------------------------------------------------ ResidualBlock pieces (reference blocks.py:108-164)
class PointwiseNormFn(torch.autograd.Function):
    """Conv2d(cin, cout, 1, stride) + Instance/BatchNorm without activation: the ``downsample_skip`` branch of the
    reference ResidualBlock (blocks.py:147-153).  Stride 1 runs the 1x1 MFMA mode; stride 2 runs the 2x2/s2 gather with
    only tap (0,0) populated (the other three taps are zero weights), so no extra kernels are needed."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, stride: int, cfg: NormCfg):
        _need_dev(x, weight)
        x = x.contiguous()
        dtype = _dt(x)
        n, h, w, cin = x.shape
        cout = weight.shape[0]
        if stride == 2 and (h % 2 or w % 2):
            raise MiaError("ResidualBlock stride-2 skip needs even H, W on the MI355X path")
        ho, wo = (h // 2, w // 2) if stride == 2 else (h, w)
        w4 = PointwiseNormFn._as_taps(weight, stride)
        wp, npad, kpad = PackCache().get(w4, dtype, n_from_d0=True)
        y, _, stats = conv_mma(CONV_G2S2 if stride == 2 else CONV_G1, x, None, wp, npad, kpad, False, bias.detach().float(), cout,
                               (ho, wo), want_stats=True)
        z, coefs = PlainBlockFn._norm_act(ctx, y, stats, gamma, beta, cfg, n, cout, ho * wo, 1.0)
        ctx.save_for_backward(x, y, coefs, weight, gamma)
        ctx.stride, ctx.mode, ctx.fixed = stride, cfg.mode, cfg.mode == NORM_BATCH and not cfg.training
        return z

    @staticmethod
    def _as_taps(weight, stride):
        w = weight.detach()
        if stride == 1:
            return w.contiguous()
        w4 = torch.zeros((w.shape[0], w.shape[1], 2, 2), device=w.device, dtype=w.dtype)
        w4[:, :, 0, 0] = w[:, :, 0, 0]
        return w4

    @staticmethod
    def backward(ctx, dz):
        x, y, coefs, weight, gamma = ctx.saved_tensors
        dz = dz.contiguous()
        dtype = _dt(y)
        n, ho, wo, cout = y.shape
        cin = weight.shape[1]
        dev = y.device
        hw = ho * wo
        slabs = _slabs_for(hw)
        part = torch.empty((n, slabs, cout, 2), device=dev, dtype=torch.float32)
        cc = torch.empty((2, n, cout), device=dev, dtype=torch.float32)
        dgb = torch.empty((3, cout), device=dev, dtype=torch.float32)
        dy = torch.empty_like(y)
        call("mia_norm_act_bwd", _p(dz), None, _p(y), _p(dy), dtype, _p(coefs[2]), _p(coefs[3]), _p(coefs[0]), _p(coefs[1]),
             _p(None if ctx.fixed else coefs[4]), n, _c_i64(hw), cout, ctx.mode, int(ctx.fixed), _c_float(1.0), slabs, _p(part),
             _p(cc[0]), _p(cc[1]), _p(dgb[0]), _p(dgb[1]), _p(dgb[2]), 0, _p(_amax_new(dy)), _stream())
        w4 = PointwiseNormFn._as_taps(weight, ctx.stride)
        if ctx.stride == 2:
            g4 = conv_wgrad(WGRAD_2S2, x, None, dy, w4.shape, cout, cin)
            dw = g4[:, :, 0:1, 0:1].contiguous()
        else:  # 1x1 stride 1: centre tap of the 3x3 weight-gradient kernel
            g9 = conv_wgrad(WGRAD_3S1, x, None, dy, (cout, cin, 3, 3), cout, cin)
            dw = g9[:, :, 1:2, 1:2].contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wb, npad, kpad = PackCache().get(w4, dtype, n_from_d0=False)
            if ctx.stride == 2:
                dx, _, _ = conv_mma(CONV_T2S2, dy, None, wb, npad, kpad, False, None, cin, (x.shape[1], x.shape[2]))
            else:
                dx, _, _ = conv_mma(CONV_G1, dy, None, wb, npad, kpad, False, None, cin, (ho, wo))
        return dx, dw, dgb[2], dgb[0], dgb[1], None, None


class ScaleLReLUFn(torch.autograd.Function):
    """z = LeakyReLU(m[n,c] * v): Dropout2d AFTER the norm followed by the activation (ResidualBlock order
    conv -> norm -> dropout -> lrelu, blocks.py:141); m = None means no dropout."""

    @staticmethod
    def forward(ctx, v, m, slope: float):
        _need_dev(v)
        v = v.contiguous()
        n, h, w, c = v.shape
        coef = torch.zeros((2, n, c), device=v.device, dtype=torch.float32)
        if m is None:
            coef[0].fill_(1.0)
        else:
            coef[0].copy_(m)
        z = torch.empty_like(v)
        call("mia_norm_act_fwd", _p(v), _p(z), _dt(v), _p(coef[0]), _p(coef[1]), n, _c_i64(h * w), c, _c_float(slope), _p(_amax_new(z)), _stream())
        ctx.save_for_backward(v, coef)
        ctx.slope = slope
        return z

    @staticmethod
    def backward(ctx, dz):
        v, coef = ctx.saved_tensors
        dz = dz.contiguous()
        n, h, w, c = v.shape
        dev = v.device
        slabs = _slabs_for(h * w)
        part = torch.empty((n, slabs, c, 2), device=dev, dtype=torch.float32)
        cc = torch.empty((2, n, c), device=dev, dtype=torch.float32)
        junk = torch.empty((2, c), device=dev, dtype=torch.float32)
        dv = torch.empty_like(v)
        # frozen statistics: dv = scale * dz * lrelu'(scale*v)
        call("mia_norm_act_bwd", _p(dz), None, _p(v), _p(dv), _dt(v), _p(coef[0]), _p(coef[1]), _p(coef[0]), _p(coef[1]), None, n,
             _c_i64(h * w), c, NORM_INSTANCE, 1, _c_float(ctx.slope), slabs, _p(part), _p(cc[0]), _p(cc[1]), _p(junk[0]), _p(junk[1]),
             None, 0, _p(_amax_new(dv)), _stream())
        return dv, None, None


class AddFn(torch.autograd.Function):
    """residual + out (blocks.py:164)."""

    @staticmethod
    def forward(ctx, a, b):
        _need_dev(a, b)
        a, b = a.contiguous(), b.contiguous()
        if a.shape != b.shape or a.dtype != b.dtype:
            raise RuntimeError(f"The size of tensor a {tuple(a.shape)} must match the size of tensor b {tuple(b.shape)}")
        out = torch.empty_like(a)
        call("mia_add", _p(a), _p(b), _p(out), _dt(a), _c_i64(a.numel()), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g
