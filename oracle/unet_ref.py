"""Functional CPU restatement of the reference UNet (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/models/unet/blocks.py:66-105 (PlainBlock),
blocks.py:108-164 (ResidualBlock), unet.py:28-91 (UNetEncoder),
unet.py:94-244 (UNetDecoder), unet.py:247-298 (UNet).

The network is expressed as pure functions over a flat ``{state_dict key ->
tensor}`` mapping (the reference's checkpoint format), so it can
run on any weights the reference or the HIP path produce, with or without
autograd.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

LRELU_SLOPE = 0.01  # nn.LeakyReLU() default, blocks.py:100
NORM_EPS = 1e-5  # blocks.py:98
BN_MOMENTUM = 0.1  # nn.BatchNorm2d default


def _norm(p: Params, prefix: str, y: torch.Tensor, normalization: str, training: bool,
          update_running: bool = True) -> torch.Tensor:
    """blocks.py:98 -- InstanceNorm2d / BatchNorm2d(C, eps=1e-5, affine=True)."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    if normalization == "instance":
        # track_running_stats=False -> eval == train
        return F.instance_norm(y, None, None, w, b, True, BN_MOMENTUM, NORM_EPS)
    if normalization == "batch":
        rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
        if training and update_running:
            out = F.batch_norm(y, rm, rv, w, b, True, BN_MOMENTUM, NORM_EPS)
            nb = prefix + ".num_batches_tracked"
            if nb in p:
                p[nb] += 1
            return out
        if training:
            return F.batch_norm(y, None, None, w, b, True, BN_MOMENTUM, NORM_EPS)
        return F.batch_norm(y, rm, rv, w, b, False, BN_MOMENTUM, NORM_EPS)
    raise KeyError(normalization)


def plain_block(p: Params, prefix: str, x: torch.Tensor, stride: int, normalization: str,
                training: bool, drop_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3(bias) -> Dropout2d -> norm -> LeakyReLU(0.01)   (blocks.py:83-102).

    ``drop_mask`` is the explicit per-(n,c) keep/scale mask (values 0 or
    1/(1-p)); RNG streams cannot match across devices, so parity runs pass it.
    """
    w = p[prefix + ".all.0.weight"]
    y = F.conv2d(x, w, p[prefix + ".all.0.bias"], stride=stride, padding=(w.shape[-1] - 1) // 2)
    if drop_mask is not None:
        y = y * drop_mask[:, :, None, None]
    y = _norm(p, prefix + ".all.2", y, normalization, training)
    return F.leaky_relu(y, LRELU_SLOPE)


def residual_block(p: Params, prefix: str, x: torch.Tensor, stride: int, normalization: str,
                   training: bool, drop_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv -> norm -> dropout -> lrelu, plus (1x1 conv + norm) skip; residual + out
    (blocks.py:127-164).  Note the different module order: norm is ``all.1``."""
    w = p[prefix + ".all.0.weight"]
    y = F.conv2d(x, w, p[prefix + ".all.0.bias"], stride=stride, padding=(w.shape[-1] - 1) // 2)
    y = _norm(p, prefix + ".all.1", y, normalization, training)
    if drop_mask is not None:
        y = y * drop_mask[:, :, None, None]
    y = F.leaky_relu(y, LRELU_SLOPE)
    if (prefix + ".downsample_skip.0.weight") in p:
        r = F.conv2d(x, p[prefix + ".downsample_skip.0.weight"], p[prefix + ".downsample_skip.0.bias"],
                     stride=stride)
        r = _norm(p, prefix + ".downsample_skip.1", r, normalization, training)
    else:
        r = x
    return r + y


_BLOCKS = {"plain": plain_block, "res": residual_block}


def encoder_forward(p: Params, x: torch.Tensor, num_levels: int, normalization: str, training: bool,
                    block_type: str = "plain", drop_masks: Optional[dict] = None) -> List[torch.Tensor]:
    """unet.py:78-85 -- returns the per-level outputs (skips), bottleneck last."""
    blk = _BLOCKS[block_type]
    dm = drop_masks or {}
    skips = []
    for l in range(num_levels):
        pre = f"encoder.levels.{l}"
        x = blk(p, pre + ".0", x, 1 if l == 0 else 2, normalization, training, dm.get(pre + ".0"))
        x = blk(p, pre + ".1", x, 1, normalization, training, dm.get(pre + ".1"))
        skips.append(x)
    return skips


def _ds_levels(num_levels: int, deep_supervision: bool, ds_layer: int) -> List[int]:
    nu = num_levels - 1
    if deep_supervision and ds_layer > 1:
        return list(range(nu - ds_layer, nu - 1))  # unet.py:180
    return []


def decoder_forward(p: Params, skips: Sequence[torch.Tensor], normalization: str, training: bool,
                    block_type: str = "plain", deep_supervision: bool = False, ds_layer: int = 0,
                    return_ds: bool = False, return_feat: bool = False,
                    drop_masks: Optional[dict] = None, channels_list: Optional[Sequence[int]] = None):
    """unet.py:206-244."""
    blk = _BLOCKS[block_type]
    dm = drop_masks or {}
    sk = list(skips)[::-1]
    x = sk.pop(0)
    ds_levels = _ds_levels(len(skips), deep_supervision, ds_layer)
    ds_out, ds_feat = [], []
    for l, feat in enumerate(sk):
        x = F.conv_transpose2d(x, p[f"decoder.upsamples.{l}.weight"], p[f"decoder.upsamples.{l}.bias"], stride=2)
        x = torch.cat([feat, x], dim=1)  # skip first, unet.py:213
        pre = f"decoder.levels.{l}"
        x = blk(p, pre + ".0", x, 1, normalization, training, dm.get(pre + ".0"))
        x = blk(p, pre + ".1", x, 1, normalization, training, dm.get(pre + ".1"))
        if return_ds and l in ds_levels:
            h = F.conv2d(x, p[f"decoder.ds.{l}.0.weight"], p[f"decoder.ds.{l}.0.bias"])
            # Upsample(scale_factor=c_l/c_0, bilinear, align_corners=False), unet.py:193-197
            scale = x.shape[1] // p["decoder.seg_output.weight"].shape[1]
            h = F.interpolate(h, scale_factor=scale, mode="bilinear", align_corners=False)
            ds_out.append(h)
            ds_feat.append(x)
    seg = F.conv2d(x, p["decoder.seg_output.weight"], p["decoder.seg_output.bias"])
    if return_ds:
        outs = [seg] + ds_out[::-1]
        if return_feat:
            return outs, [x] + ds_feat[::-1]
        return outs
    if return_feat:
        return seg, x
    return seg


def unet_forward(p: Params, x: torch.Tensor, normalization: str = "instance", training: bool = True,
                 block_type: str = "plain", deep_supervision: bool = False, ds_layer: int = 0,
                 return_ds: bool = False, drop_masks: Optional[dict] = None):
    """unet.py:291-292."""
    L = num_levels(p)
    skips = encoder_forward(p, x, L, normalization, training, block_type, drop_masks)
    return decoder_forward(p, skips, normalization, training, block_type, deep_supervision, ds_layer,
                           return_ds, False, drop_masks)


def enc_feature(p: Params, x: torch.Tensor, normalization: str = "instance", training: bool = False,
                block_type: str = "plain") -> torch.Tensor:
    """unet.py:87-91 -- global average pool of the bottleneck."""
    skips = encoder_forward(p, x, num_levels(p), normalization, training, block_type)
    return F.adaptive_avg_pool2d(skips[-1], (1, 1)).view(x.shape[0], -1)


def pixel_feature(p: Params, x: torch.Tensor, normalization: str = "instance", training: bool = False,
                  block_type: str = "plain"):
    """unet.py:297-298 with return_ds=False -> (logits, last feature map)."""
    skips = encoder_forward(p, x, num_levels(p), normalization, training, block_type)
    return decoder_forward(p, skips, normalization, training, block_type, return_feat=True)


def num_levels(p: Params) -> int:
    n = 0
    while f"encoder.levels.{n}.0.all.0.weight" in p:
        n += 1
    return n


def init_params(input_channels: int, output_classes: int, channels_list: Sequence[int],
                normalization: str = "instance", seed: Optional[int] = None,
                deep_supervision: bool = False, ds_layer: int = 0,
                generator: Optional[torch.Generator] = None) -> Params:
    """Default-torch-init parameters with the reference's key set and creation order
    (unet.py:54-76, :131-204).

    conv: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weight and bias with
    fan_in = Cin*k*k; ConvTranspose2d fan_in is taken from ``weight.size(1)*k*k`` =
    Cout*4.  Draw order = module construction order, weight before bias, so with
    ``torch.manual_seed(s)`` this reproduces ``UNet(...)`` bit-for-bit.
    """
    if seed is not None:
        torch.manual_seed(seed)
    g = generator
    p: Params = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound if g is not None else \
            torch.empty(shape).uniform_(-bound, bound)

    def conv(prefix, cin, cout, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        p[prefix + ".weight"] = uni((cout, cin, k, k), bound)
        p[prefix + ".bias"] = uni((cout,), bound)

    def norm(prefix, c):
        p[prefix + ".weight"] = torch.ones(c)
        p[prefix + ".bias"] = torch.zeros(c)
        if normalization == "batch":
            p[prefix + ".running_mean"] = torch.zeros(c)
            p[prefix + ".running_var"] = torch.ones(c)
            p[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(prefix, cin, cout):
        conv(prefix + ".all.0", cin, cout, 3)
        norm(prefix + ".all.2", cout)

    cl = list(channels_list)
    for l, c in enumerate(cl):
        block(f"encoder.levels.{l}.0", input_channels if l == 0 else cl[l - 1], c)
        block(f"encoder.levels.{l}.1", c, c)
    dl = cl[::-1]
    nu = len(dl) - 1
    for l in range(nu):
        cin, cout = dl[l], dl[l + 1]
        bound = 1.0 / math.sqrt(cout * 4)
        p[f"decoder.upsamples.{l}.weight"] = uni((cin, cout, 2, 2), bound)
        p[f"decoder.upsamples.{l}.bias"] = uni((cout,), bound)
        block(f"decoder.levels.{l}.0", 2 * cout, cout)
        block(f"decoder.levels.{l}.1", cout, cout)
    conv("decoder.seg_output", dl[-1], output_classes, 1)
    for l in _ds_levels(len(cl), deep_supervision, ds_layer):
        conv(f"decoder.ds.{l}.0", dl[l + 1], output_classes, 1)
    return p
