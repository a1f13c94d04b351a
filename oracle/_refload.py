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

Selectors (``Ref.load_selectors``, round 4): ``activelearning/*.py`` import ``h5py`` (absent) and
``datasets.active_dataset`` (the reference's dataset package needs PIL / h5py / skimage) at module level, the first only
for the ``feature_path`` option and the second only for a type annotation.  Both get EMPTY placeholders for the duration
of the import; ``utils`` is the reference's own ``utils/common.py`` loaded by path.  The selector arithmetic
(softmax scores, ``kcenter_greedy``, feature standardisation, gradient embeddings) is the reference's, unmodified.
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

    def load_selectors(self):
        """Load the reference's `activelearning` modules by path (see the module docstring).  Returns a namespace with
        `entropy`, `confidence`, `margin`, `coreset`, `kmean`, `badge`, `random` modules."""
        saved = {k: sys.modules.get(k) for k in ("h5py", "datasets", "datasets.active_dataset", "utils")}
        h5 = types.ModuleType("h5py")

        class _NoH5:
            def __init__(self, *a, **k):
                raise RuntimeError("h5py is not installed; the feature_path option is parity-unpinned")

        h5.File = _NoH5
        h5.Dataset = _NoH5
        ds = types.ModuleType("datasets")
        ds.__path__ = []
        dsa = types.ModuleType("datasets.active_dataset")

        class ActiveDataset:  # annotation only; the generator passes a duck-typed stand-in
            pass

        dsa.ActiveDataset = ActiveDataset
        ds.active_dataset = dsa
        try:
            sys.modules["h5py"] = h5
            sys.modules["datasets"] = ds
            sys.modules["datasets.active_dataset"] = dsa
            sys.modules["utils"] = _load("_ref_utils_common", os.path.join(REF_SRC, "utils", "common.py"))
            ap = types.ModuleType("_ref_al")
            ap.__path__ = [os.path.join(REF_SRC, "activelearning")]
            sys.modules["_ref_al"] = ap
            ns = types.SimpleNamespace()
            _load("_ref_al.active_selector", os.path.join(REF_SRC, "activelearning", "active_selector.py"))
            for short in ("random", "entropy", "confidence", "margin", "coreset", "kmean", "badge"):
                setattr(ns, short, _load(f"_ref_al.{short}_selector",
                                         os.path.join(REF_SRC, "activelearning", f"{short}_selector.py")))
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
        return ns
