"""Loader for the upstream reference (DEV CONTAINER ONLY; TEST INFRASTRUCTURE ONLY).

Imports the reference's own Python from /root/reference/src so that
``gen_golden.py`` can produce the committed vectors under tests/golden/.
/root/reference does not exist on the GPU box; nothing imported by the
``-m gpu`` tests, ``smoke()`` or ``bench.py`` may use this module.

Recipe: ``models/unet/unet.py:6`` has an *unused*
``import torchvision.transforms.functional as F`` and ``transforms/*.py`` import
torchvision at module level; torchvision is not installed.  Empty placeholder
modules satisfy those imports; no reference arithmetic is replaced -- every
function that would actually *call* into torchvision is simply not used here.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REF_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(REF_SRC)


def _placeholders():
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    class _Missing:  # constructor placeholder so RandomContrast/Brightness modules import
        def __init__(self, *a, **k):
            raise RuntimeError("torchvision is not installed; this transform is parity-unpinned")

    tvt.ColorJitter = _Missing
    tvt.RandomAffine = _Missing
    tvt.RandomRotation = _Missing
    tvt.RandomCrop = _Missing
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Ref:
    """Namespace with the reference modules that are importable here."""

    def __init__(self):
        if not available():
            raise RuntimeError("/root/reference is not present")
        _placeholders()
        # models.unet: bypass models/unet/__init__.py (pulls cv2 + torchvision via unet_processor)
        pkg = types.ModuleType("_ref_unet")
        pkg.__path__ = [os.path.join(REF_SRC, "models", "unet")]
        sys.modules["_ref_unet"] = pkg
        self.blocks = _load("_ref_unet.blocks", os.path.join(REF_SRC, "models", "unet", "blocks.py"))
        self.unet = _load("_ref_unet.unet", os.path.join(REF_SRC, "models", "unet", "unet.py"))
        lp = types.ModuleType("_ref_losses")
        lp.__path__ = [os.path.join(REF_SRC, "losses")]
        sys.modules["_ref_losses"] = lp
        self.dice_loss = _load("_ref_losses.dice_loss", os.path.join(REF_SRC, "losses", "dice_loss.py"))
        self.ce_loss = _load("_ref_losses.ce_loss", os.path.join(REF_SRC, "losses", "ce_loss.py"))
        self.compound = _load("_ref_losses.compound_losses", os.path.join(REF_SRC, "losses", "compound_losses.py"))
        self.lr_scheduler = _load("_ref_lr_scheduler", os.path.join(REF_SRC, "scheduler", "lr_scheduler.py"))
        tp = types.ModuleType("_ref_transforms")
        tp.__path__ = [os.path.join(REF_SRC, "transforms")]
        sys.modules["_ref_transforms"] = tp
        self.t_common = _load("_ref_transforms.common", os.path.join(REF_SRC, "transforms", "common.py"))
        self.t_image = _load("_ref_transforms.image_transform", os.path.join(REF_SRC, "transforms", "image_transform.py"))
        self.t_joint = _load("_ref_transforms.joint_transform", os.path.join(REF_SRC, "transforms", "joint_transform.py"))
        self.t_norm = _load("_ref_transforms.normalization", os.path.join(REF_SRC, "transforms", "normalization.py"))
