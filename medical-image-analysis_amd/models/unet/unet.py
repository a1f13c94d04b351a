"""UNet-2D, MI355X-native, drop-in for the reference ``models.unet.UNet``
(`src/models/unet/unet.py:28-298`): same constructor, methods, sub-module names and
``state_dict`` keys/shapes; activations flow between blocks as NHWC tensors on HIP kernels, the
decoder's ``torch.cat([skip, up], 1)`` is never materialised (the conv reads two sources).
"""
from __future__ import annotations

from typing import Union

import torch
import torch.nn as nn

from mia_hip import ops

from .blocks import PlainBlock, ResidualBlock, Upsample, _only_2d

conv_dict = {2: nn.Conv2d}
transpose_conv_dict = {2: nn.ConvTranspose2d}
upsample_dict = {2: "bilinear"}
block_dict = {"plain": PlainBlock, "res": ResidualBlock}


class UNetEncoder(nn.Module):
    def __init__(self, dimension, input_channels, channels_list, block=PlainBlock, **block_kwargs):
        super().__init__()
        _only_2d(dimension)
        self.dimension = dimension
        self.input_channels = input_channels
        self.channels_list = channels_list
        self.block_type = block
        self.levels = nn.ModuleList()
        for l, num_channels in enumerate(self.channels_list):
            in_channels = self.input_channels if l == 0 else self.channels_list[l - 1]
            first_stride = 1 if l == 0 else 2
            self.levels.append(nn.Sequential(
                block(dimension, in_channels, num_channels, stride=first_stride, **block_kwargs),
                block(dimension, num_channels, num_channels, stride=1, **block_kwargs)))

    def forward_nhwc(self, x, compute_dtype=None):
        """x: NHWC.  A 1-channel fp32 image may be passed as is: the stem kernel reads it directly and emits
        `compute_dtype` activations (no separate cast pass)."""
        skips = []
        last = len(self.levels) - 1
        for l, s in enumerate(self.levels):
            od = compute_dtype if l == 0 else None
            x = s[0].forward_nhwc(x, out_dtype=od, lazy=True) if _pair_is_fused(s, x, od) else s[0].forward_nhwc(x, out_dtype=od)
            if l < last and isinstance(s[1], PlainBlock) and torch.is_grad_enabled():
                # a level's output feeds the next level AND the decoder: two outputs on one storage, so each consumer's
                # gradient reaches the block separately and is summed on load in its norm backward (no `add` pass)
                x, skip = s[1].forward_nhwc(x, dup=True)
                x._mia_dup = skip._mia_dup = True  # each view has exactly one consumer (ops.PlainBlockFn: ctx.dup_in)
                skips.append(skip)
            else:
                x = s[1].forward_nhwc(x)
                skips.append(x)
        return skips

    def forward(self, x, return_skips=False):
        skips = self.forward_nhwc(ops.to_nhwc(x, _compute_dtype(self, x)))
        if return_skips:
            return [ops.nhwc_as_nchw(s) for s in skips]
        return ops.nhwc_as_nchw(skips[-1])

    def get_feature(self, x):
        """adaptive_avg_pool2d(bottleneck, 1).view(B, -1)  (reference unet.py:87-91)."""
        b = self.forward_nhwc(ops.to_nhwc(x, _compute_dtype(self, x)))[-1]
        return ops.global_avg_pool(b)


def _pair_is_fused(level, x, out_dtype=None, x2=None) -> bool:
    """The two PlainBlocks of a level (unet.py:54-76, 157-173) run as a fused pair when the second block's kernels can
    normalise on load: the first block then hands over its raw conv output (`ops.LazyAct`) and never writes its activation."""
    b0, b1 = level[0], level[1]
    if not (isinstance(b0, PlainBlock) and isinstance(b1, PlainBlock)):
        return False
    h, w = x.shape[1], x.shape[2]
    if b0.stride == 2:
        h, w = (h + 1) // 2, (w + 1) // 2
    return b1.consumes_lazy(b0, h, w, out_dtype or x.dtype)


def _compute_dtype(module, x):
    dt = getattr(module, "compute_dtype", None)
    if dt is not None:
        return dt
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


class UNetDecoder(nn.Module):
    def __init__(self, dimension, output_classes, channels_list, upconv=True, deep_supervision=False, ds_layer=0,
                 block: Union[PlainBlock, ResidualBlock] = PlainBlock, **block_kwargs):
        super().__init__()
        _only_2d(dimension)
        self.dimension = dimension
        self.output_classes = output_classes
        self.channels_list = channels_list
        self.deep_supervision = deep_supervision
        self.block_type = block
        num_upsample = len(self.channels_list) - 1
        assert ds_layer <= num_upsample
        if not upconv:
            # the reference's upconv=False branch builds nn.Sequential([list]) and raises TypeError (unet.py:144-153)
            raise TypeError("upconv=False is broken in the reference (nn.Sequential of a list); not supported")
        conv_type = conv_dict[dimension]
        self.levels = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        for l in range(num_upsample):
            in_channels, out_channels = self.channels_list[l], self.channels_list[l + 1]
            self.upsamples.append(transpose_conv_dict[dimension](in_channels, out_channels, kernel_size=2, stride=2))
            self.levels.append(nn.Sequential(
                block(dimension, out_channels * 2, out_channels, stride=1, **block_kwargs),
                block(dimension, out_channels, out_channels, stride=1, **block_kwargs)))
        self.seg_output = conv_type(self.channels_list[-1], self.output_classes, kernel_size=1, stride=1)
        if self.deep_supervision and ds_layer > 1:
            self.ds_layer_list = list(range(num_upsample - ds_layer, num_upsample - 1))
            self.ds = nn.ModuleList()
            for l in range(num_upsample - 1):
                if l in self.ds_layer_list:
                    in_channels = self.channels_list[l + 1]
                    up_factor = in_channels // self.channels_list[-1]
                    assert up_factor > 1
                    ds = nn.Sequential(conv_type(in_channels, self.output_classes, kernel_size=1, stride=1),
                                       Upsample(scale_factor=up_factor, mode=upsample_dict[dimension], align_corners=False))
                else:
                    ds = None
                self.ds.append(ds)

    def _run(self, skips, return_ds, fuse_head=False):
        """skips: NHWC tensors, encoder order (bottleneck last)."""
        skips = list(skips)[::-1]
        x = skips.pop(0)
        ds_feats, ds_outputs = [], []
        for l, feat in enumerate(skips):
            up = self.upsamples[l]
            x = ops.ConvTranspose2x2Fn.apply(x, up.weight, up.bias)
            # cat([skip, up], 1) folded into the conv's two-source read
            b0 = self.levels[l][0]
            x = b0.forward_nhwc(feat, x, lazy=True) if _pair_is_fused(self.levels[l], feat) else b0.forward_nhwc(feat, x)
            last = self.levels[l][1]
            if fuse_head and l == len(skips) - 1 and isinstance(last, PlainBlock):
                # nobody but the segmentation head reads the last block's output: one fused node, no activation tensor
                seg = last.forward_head_nhwc(x, self.seg_output)
                if seg is not None:
                    return seg, None, ds_outputs, ds_feats
            x = last.forward_nhwc(x)  # (accepts an ops.LazyAct)
            if return_ds and self.deep_supervision and (l in self.ds_layer_list):
                head = self.ds[l][0]
                ds_feats.append(x)
                ds_outputs.append(self.ds[l][1](ops.HeadFn.apply(x, head.weight, head.bias)))
        seg = ops.HeadFn.apply(x, self.seg_output.weight, self.seg_output.bias)
        return seg, x, ds_outputs, ds_feats

    def forward_nhwc(self, skips, return_ds=False):
        seg, _, ds_outputs, _ = self._run(skips, return_ds, fuse_head=True)
        if return_ds:
            return [seg] + ds_outputs[::-1]
        return seg

    def forward(self, skips, return_ds=False):
        dt = _compute_dtype(self, skips[0])
        return self.forward_nhwc([ops.to_nhwc(s, dt) for s in skips], return_ds)

    def get_feature_nhwc(self, skips, return_ds=False):
        seg, x, ds_outputs, ds_feats = self._run(skips, return_ds)
        if return_ds:
            return [seg] + ds_outputs[::-1], [ops.nhwc_as_nchw(f) for f in [x] + ds_feats[::-1]]
        return seg, ops.nhwc_as_nchw(x)

    def get_feature(self, skips, return_ds=False):
        dt = _compute_dtype(self, skips[0])
        return self.get_feature_nhwc([ops.to_nhwc(s, dt) for s in skips], return_ds)


class UNet(nn.Module):
    """Drop-in for reference ``UNet`` (unet.py:247-298).

    Build-side addition: ``compute_dtype`` (default ``torch.float32`` = the reference's arithmetic;
    ``torch.bfloat16`` = bf16 activations / MFMA operands with fp32 accumulation, statistics,
    parameters, logits and loss).  Set it as an attribute or via ``set_compute_dtype``.
    """

    def __init__(self, dimension, input_channels, output_classes, channels_list, deep_supervision=False, ds_layer=0,
                 block_type="plain", **block_kwargs):
        super().__init__()
        block = block_dict[block_type]
        self.encoder = UNetEncoder(dimension, input_channels, channels_list, block=block, **block_kwargs)
        self.decoder = UNetDecoder(dimension, output_classes, channels_list[::-1], block=block,
                                   deep_supervision=deep_supervision, ds_layer=ds_layer, **block_kwargs)
        self.compute_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    def _skips(self, x):
        if x.shape[1] == 1 and x.dtype == torch.float32:  # stem reads the fp32 image directly
            return self.encoder.forward_nhwc(ops.to_nhwc(x, torch.float32), self.compute_dtype)
        return self.encoder.forward_nhwc(ops.to_nhwc(x, self.compute_dtype), self.compute_dtype)

    def _draw_dropout(self, n: int, device) -> None:
        """All Dropout2d channel masks of this forward ([N, C] of {0, 1/(1-p)} per PlainBlock, blocks.py:92-96) from ONE
        launch of the library's Philox mask kernel instead of one per block; a block without a pooled mask draws its own."""
        if not self.training:
            return
        blocks = [m for m in self.modules() if isinstance(m, PlainBlock) and m.dropout_prob and m.drop_mask_override is None]
        if len(blocks) < 2 or any(b.dropout_prob != blocks[0].dropout_prob for b in blocks):
            return
        keep = 1.0 - float(blocks[0].dropout_prob)
        sizes = [n * b.all[2].num_features for b in blocks]
        pool = ops.dropout_mask(sum(sizes), keep, device)
        off = 0
        for b, sz in zip(blocks, sizes):
            b._drop_from_pool = pool[off:off + sz].view(n, -1)
            off += sz

    def forward(self, x, return_ds=False):
        ops._COLSUM_HINT.clear()  # hints are only valid inside the backward pass of the forward that produced them
        ops._ACC_HINT.clear()
        ops._CR_HINT.clear()
        self._draw_dropout(x.shape[0], x.device)
        return self.decoder.forward_nhwc(self._skips(x), return_ds=return_ds)

    def get_enc_feature(self, x):
        return ops.global_avg_pool(self._skips(x)[-1])

    def get_pixel_feature(self, x, return_ds=False):
        return self.decoder.get_feature_nhwc(self._skips(x), return_ds=return_ds)
