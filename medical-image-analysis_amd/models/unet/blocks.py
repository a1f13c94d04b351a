"""Conv blocks of the UNet, MI355X-native.

Same classes / constructor arguments / ``state_dict`` keys as the reference
(`src/models/unet/blocks.py:66-164`): ``self.all = Sequential(conv, dropout, norm, nonlin)`` holds
ordinary torch parameter containers (so default init consumes the RNG exactly like the reference and
checkpoints interchange), but ``forward`` never calls them: the whole block runs as the fused HIP
sequence conv3x3(MFMA, +stats epilogue) -> finalize -> normalise+LeakyReLU.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from mia_hip import NORM_BATCH, NORM_INSTANCE, ops

norm_dict = {"instance": {2: nn.InstanceNorm2d}, "batch": {2: nn.BatchNorm2d}}
conv_dict = {2: nn.Conv2d}
dropout_dict = {2: nn.Dropout2d}
_NORM_MODE = {"instance": NORM_INSTANCE, "batch": NORM_BATCH}


def _only_2d(dimension):
    if dimension != 2:
        raise NotImplementedError("the MI355X path implements dimension=2 only (the al_train hot path); got %r" % (dimension,))


class Identity(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, input):
        return input


class Normalize(nn.Module):
    def __init__(self, p=2, dim=1):
        super().__init__()
        self.p, self.dim = p, dim

    def forward(self, x):
        return F.normalize(x, p=self.p, dim=self.dim)


class Upsample(nn.Module):
    """reference blocks.py:45-63; bilinear, align_corners=False -> HIP resize kernel."""

    def __init__(self, size=None, scale_factor=None, mode="nearest", align_corners=False):
        super().__init__()
        self.align_corners, self.mode, self.scale_factor, self.size = align_corners, mode, scale_factor, size

    def forward(self, x):
        from transforms.hip import functional_hip as FH
        n, c, h, w = x.shape
        if self.size is not None:
            oh, ow = (self.size, self.size) if isinstance(self.size, int) else self.size
        else:
            oh, ow = int(h * self.scale_factor), int(w * self.scale_factor)
        if self.mode == "bilinear" and not self.align_corners:
            return FH.ResizeBilinearFn.apply(x, oh, ow)
        if self.mode == "nearest":
            return FH.resize_nearest(x, oh, ow)
        raise NotImplementedError(f"Upsample mode {self.mode!r} align_corners={self.align_corners}")


class PlainBlock(nn.Module):
    def __init__(self, dimension, input_channels, output_channels, stride=1, kernel_size=3, normalization="instance",
                 dropout_prob=None):
        super().__init__()
        _only_2d(dimension)
        if kernel_size != 3:
            raise NotImplementedError("MI355X PlainBlock implements kernel_size=3 (the only value the reference uses)")
        if stride not in (1, 2):
            raise NotImplementedError("stride must be 1 or 2")
        conv = conv_dict[dimension](input_channels, output_channels, kernel_size, stride=stride,
                                    padding=(kernel_size - 1) // 2, bias=True)
        do = Identity() if dropout_prob is None else dropout_dict[dimension](p=dropout_prob, inplace=True)
        norm = norm_dict[normalization][dimension](output_channels, eps=1e-5, affine=True)
        nonlin = nn.LeakyReLU(inplace=True)
        self.all = nn.Sequential(conv, do, norm, nonlin)
        self.stride = stride
        self.normalization = normalization
        self.dropout_prob = dropout_prob
        self.drop_mask_override = None  # [N, Cout] tensor of {0, 1/(1-p)} for parity runs
        self.batch_sync = None  # ops.BatchSync: batch-norm statistics over every data-parallel rank (convert_sync_batchnorm)

    def _cfg(self, n: int, device) -> ops.NormCfg:
        norm = self.all[2]
        drop = None
        if self.drop_mask_override is not None:
            drop = self.drop_mask_override.to(device=device, dtype=torch.float32).contiguous()
        elif self.dropout_prob is not None and self.training and self.dropout_prob > 0:
            pooled = self.__dict__.pop("_drop_from_pool", None)  # one RNG launch per model forward (UNet._draw_dropout)
            if pooled is not None and pooled.shape == (n, norm.num_features) and pooled.device == device:
                drop = pooled
            else:
                keep = 1.0 - float(self.dropout_prob)
                drop = ops.dropout_mask(n * norm.num_features, keep, device).view(n, norm.num_features)
        if self.normalization == "batch":
            return ops.NormCfg(NORM_BATCH, self.training, norm.eps, norm.momentum, norm.running_mean, norm.running_var,
                               norm.num_batches_tracked, drop, sync=self.batch_sync)
        return ops.NormCfg(NORM_INSTANCE, self.training, norm.eps, 0.1, None, None, None, drop)

    def consumes_lazy(self, producer: "PlainBlock", h: int, w: int, dtype) -> bool:
        """True when this block can take `producer`'s output as an `ops.LazyAct` (raw conv output + coefficient table): the
        fused PlainBlock -- the producer's norm + LeakyReLU pass (blocks.py:98-102) is folded into this block's conv and
        weight gradient, and its activation tensor is never written.  `h, w` = this block's input size."""
        conv = self.all[0]
        return (self.stride == 1 and producer.all[0].out_channels == conv.in_channels and producer.batch_sync is None
                and ops.nl_supported(dtype, conv.in_channels, conv.out_channels, h, w, torch.is_grad_enabled()))

    def forward_nhwc(self, x1, x2=None, out_dtype=None, dup=False, lazy=False):
        """dup=True: the output twice (one storage), for a tensor with two consumers -- see ops.PlainBlockFn.forward.
        lazy=True: returns an `ops.LazyAct` (the caller checked `next_block.consumes_lazy`); x1 may itself be one."""
        conv, norm = self.all[0], self.all[2]
        nl_coefs, nl_slope = None, ops.LRELU_SLOPE
        if isinstance(x1, ops.LazyAct):
            if x2 is None and self.stride == 1:
                nl_coefs, nl_slope, x1 = x1.coefs, x1.slope, x1.y
            else:
                x1 = x1.materialize()
        out = ops.PlainBlockFn.apply(x1, x2, conv.weight, conv.bias, norm.weight, norm.bias, self.stride,
                                     self._cfg(x1.shape[0], x1.device), out_dtype, ops.LRELU_SLOPE, dup, lazy, nl_coefs, nl_slope)
        return ops.LazyAct(out[0], out[1], ops.LRELU_SLOPE) if lazy else out

    def forward_head_nhwc(self, x1, head):
        """This block followed by the 1x1 head `head` (nn.Conv2d) in one fused node, or None when the fused kernels do
        not cover the case (the caller then runs the block and the head separately)."""
        conv, norm = self.all[0], self.all[2]
        if self.stride != 1 or x1.dtype not in (torch.float32, torch.bfloat16):
            return None
        lazy_in = x1 if isinstance(x1, ops.LazyAct) else None
        if lazy_in is not None:
            x1 = lazy_in.y
        if not ops.PlainBlockHeadFn.eligible(x1, conv.weight, head.weight, self._cfg_probe(), x1.dtype):
            return None
        cfg = self._cfg(x1.shape[0], x1.device)
        if lazy_in is not None:
            return ops.PlainBlockHeadFn.apply(x1, conv.weight, conv.bias, norm.weight, norm.bias, cfg, head.weight, head.bias,
                                              ops.LRELU_SLOPE, lazy_in.coefs, lazy_in.slope)
        return ops.PlainBlockHeadFn.apply(x1, conv.weight, conv.bias, norm.weight, norm.bias, cfg, head.weight, head.bias)

    def _cfg_probe(self) -> ops.NormCfg:
        """NormCfg for eligibility questions only: no dropout mask is drawn (a pooled mask must survive until the real call)."""
        return ops.NormCfg(_NORM_MODE[self.normalization], self.training, sync=self.batch_sync)

    def forward(self, x):
        dt = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
        return ops.nhwc_as_nchw(self.forward_nhwc(ops.to_nhwc(x, dt)))


class ResidualBlock(nn.Module):
    """Drop-in for the reference ResidualBlock (blocks.py:108-164): ``all = Sequential(conv, norm, dropout, nonlin)``
    (note: norm BEFORE dropout, and the ctor takes ``norm_key=``, not ``normalization=``), optional
    ``downsample_skip = Sequential(conv1x1(stride), norm)``, output ``residual + out`` with no activation after the add.
    Unreachable from ``al_train`` (it passes ``normalization=`` -> TypeError, in the reference too)."""

    def __init__(self, dimension, input_channels, output_channels, stride=1, kernel_size=3, norm_key="instance",
                 dropout_prob=None):
        super().__init__()
        _only_2d(dimension)
        if kernel_size != 3:
            raise NotImplementedError("MI355X ResidualBlock implements kernel_size=3")
        if stride not in (1, 2):
            raise NotImplementedError("stride must be 1 or 2")
        conv = conv_dict[dimension](input_channels, output_channels, kernel_size, stride=stride,
                                    padding=(kernel_size - 1) // 2, bias=True)
        norm = norm_dict[norm_key][dimension](output_channels, eps=1e-5, affine=True)
        do = Identity() if dropout_prob is None else dropout_dict[dimension](p=dropout_prob, inplace=True)
        nonlin = nn.LeakyReLU(inplace=True)
        self.all = nn.Sequential(conv, norm, do, nonlin)
        if (input_channels != output_channels) or (stride != 1):
            self.downsample_skip = nn.Sequential(conv_dict[dimension](input_channels, output_channels, 1, stride, bias=True),
                                                 norm_dict[norm_key][dimension](output_channels, eps=1e-5, affine=True))
        else:
            self.downsample_skip = None
        self.stride, self.normalization, self.dropout_prob = stride, norm_key, dropout_prob
        self.drop_mask_override = None

    def _cfg(self, norm):
        if self.normalization == "batch":
            return ops.NormCfg(NORM_BATCH, self.training, norm.eps, norm.momentum, norm.running_mean, norm.running_var,
                               norm.num_batches_tracked, None)
        return ops.NormCfg(NORM_INSTANCE, self.training, norm.eps, 0.1, None, None, None, None)

    def forward_nhwc(self, x1, x2=None, out_dtype=None):
        if out_dtype is not None and x1.dtype != out_dtype:
            x1 = ops.cast_nhwc(x1, out_dtype)
        if x2 is not None:  # decoder: the reference concatenates; keep the two-source read for the 3x3 conv only
            x_cat = torch.cat([x1, x2], dim=3)
        else:
            x_cat = x1
        conv, norm = self.all[0], self.all[1]
        n1 = ops.PlainBlockFn.apply(x1, x2, conv.weight, conv.bias, norm.weight, norm.bias, self.stride, self._cfg(norm), None, 1.0)
        m = None
        if self.drop_mask_override is not None:
            m = self.drop_mask_override.to(device=x1.device, dtype=torch.float32)
        elif self.dropout_prob is not None and self.training and self.dropout_prob > 0:
            keep = 1.0 - float(self.dropout_prob)
            m = ops.dropout_mask(x1.shape[0] * norm.num_features, keep, x1.device).view(x1.shape[0], norm.num_features)
        out = ops.ScaleLReLUFn.apply(n1, m, ops.LRELU_SLOPE)
        if self.downsample_skip is not None:
            sc, sn = self.downsample_skip[0], self.downsample_skip[1]
            residual = ops.PointwiseNormFn.apply(x_cat, sc.weight, sc.bias, sn.weight, sn.bias, self.stride, self._cfg(sn))
        else:
            residual = x_cat
        return ops.AddFn.apply(residual, out)

    def forward(self, x):
        dt = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
        return ops.nhwc_as_nchw(self.forward_nhwc(ops.to_nhwc(x, dt)))


def convert_sync_batchnorm(model: nn.Module, process_group=None) -> nn.Module:
    """Make every batch-norm PlainBlock of `model` take its batch statistics over all ranks of `process_group`
    (one process per GPU), so N ranks x bs reproduce one process at N*bs (the reference is
    single-process, blocks.py:98).  Needs an initialised torch.distributed group; a no-op for instance norm."""
    sync = ops.BatchSync(process_group)
    for m in model.modules():
        if isinstance(m, PlainBlock) and m.normalization == "batch":
            m.batch_sync = sync
        elif isinstance(m, ResidualBlock) and m.normalization == "batch":
            raise NotImplementedError("synchronised batch norm is built for PlainBlock (the block al_train uses)")
    return model
