"""The `al_train` drop-in layout resolves: with PYTHONPATH = <repo>/medical-image-analysis_amd : <reference>/src the
modules only the reference has still come from the reference, and the hot-path modules come from this repo
(reference import sites: src/entry/activelearning/train.py:3, src/training/al_trainer.py:31-84).

Dev-container test: skipped where /root/reference does not exist (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-image-analysis_amd")
REF_SRC = "/root/reference/src"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference tree not present")

PROBE = r"""
import importlib.util, json, sys
out = {}
for name in sys.argv[1:]:
    try:
        spec = importlib.util.find_spec(name)
        out[name] = None if spec is None else (spec.origin or list(spec.submodule_search_locations or []))
    except Exception as e:  # parent package failed to import
        out[name] = "ERR " + repr(e)
print(json.dumps(out))
"""


def _resolve(names, pythonpath):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath))
    r = subprocess.run([sys.executable, "-c", PROBE] + list(names), capture_output=True, text=True, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_reference_only_modules_resolve_into_the_reference():
    names = ["training.al_trainer", "training.base_trainer", "models._unet", "metric.metric", "scheduler.ramps",
             "losses.adv_loss", "activelearning.badge_selector", "datasets", "utils"]  # last two: top-level, not executed
    got = _resolve(names, [PKG, REF_SRC])
    for n in names:
        assert isinstance(got[n], str) and got[n].startswith(REF_SRC + os.sep), (n, got[n])


def test_hot_path_modules_resolve_into_this_repo():
    names = ["models.unet", "models.unet.unet", "models.unet.blocks", "losses.dice_loss", "losses.compound_losses",
             "losses.ce_loss", "scheduler.lr_scheduler", "training.engine", "metric.segmentation", "activelearning.selectors",
             "transforms.hip.joint_transform", "transforms.hip.image_transform", "transforms.gpu_pipeline", "mia_hip"]
    got = _resolve(names, [PKG, REF_SRC])
    for n in names:
        assert isinstance(got[n], str) and got[n].startswith(PKG + os.sep), (n, got[n])


def test_dataloader_workers_get_the_reference_cpu_transforms():
    """Per-sample transforms run on CPU tensors in forked DataLoader workers (fugc_dataset.py:140-164): with the
    reference on the path they must be the reference's classes, never the device-only HIP ones."""
    names = ["transforms.common", "transforms.image_transform", "transforms.joint_transform", "transforms.normalization"]
    got = _resolve(names, [PKG, REF_SRC])
    for n in names:
        assert isinstance(got[n], str) and got[n].startswith(REF_SRC + os.sep), (n, got[n])
    alone = _resolve(names, [PKG])  # stand-alone: the same names are the HIP classes
    for n in names:
        assert isinstance(alone[n], str) and alone[n].startswith(PKG + os.sep), (n, alone[n])


def test_al_trainer_import_lines_execute():
    """Execute the al_trainer.py:31-84 imports that need nothing absent from this image (torchvision / wandb / medpy /
    SimpleITK are not installed here, so `training.al_trainer` itself cannot be executed -- SURVEY 8c)."""
    code = r"""
import torch
from training.base_trainer import BaseTrainer
from losses.compound_losses import DiceAndCELoss
from losses.dice_loss import DiceLoss
from scheduler.lr_scheduler import PolyLRScheduler
from scheduler.ramps import BaseRampUp
from models._unet import _UNet
from models.unet import UNet, UnetProcessor
from metric import cal_hd
import activelearning as al
names = ["ActiveSelector", "RandomSelector", "EntropySelector", "ConfidenceSelector", "MarginSelector", "CoresetSelector",
         "KMeanSelector", "BADGESelector"]
assert all(hasattr(al, n) for n in names)
import models.unet.unet as u, losses.dice_loss as d, training.base_trainer as b, models._unet as r
print(u.__file__); print(d.__file__); print(b.__file__); print(r.__file__)
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, REF_SRC]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()[-4:]
    assert lines[0].startswith(PKG) and lines[1].startswith(PKG)
    assert lines[2].startswith(REF_SRC) and lines[3].startswith(REF_SRC)


def test_decoy_packages_on_the_path_are_not_merged(tmp_path):
    """A foreign `transforms` / `models` package on sys.path (site-packages, a working directory) must not be merged into the
    drop-in: only the reference's src/ (marker training/al_trainer.py, or MIA_REFERENCE_SRC) is."""
    decoy = tmp_path / "decoy"
    for pkg, mod in (("transforms", "common"), ("models", "_unet"), ("training", "base_trainer")):
        d = decoy / pkg
        d.mkdir(parents=True)
        (d / "__init__.py").write_text("")
        (d / (mod + ".py")).write_text("DECOY = True\n")
    names = ["transforms.common", "models._unet", "training.base_trainer", "training.al_trainer"]
    # decoy in front of the reference on the path (a decoy in the WORKING directory of `python -c` shadows every PYTHONPATH
    # entry by Python's own rules, before any of this repo's code runs -- not something a package can defend against)
    got = _resolve(names, [PKG, str(decoy), REF_SRC])
    assert got["transforms.common"].startswith(REF_SRC + os.sep), got
    assert got["models._unet"].startswith(REF_SRC + os.sep), got
    assert got["training.base_trainer"].startswith(REF_SRC + os.sep), got
    # without the reference, the decoy is still ignored: this repo's alias module serves transforms.common, the others are absent
    got = _resolve(["transforms.common", "models._unet"], [PKG, str(decoy)])
    assert got["transforms.common"].startswith(PKG + os.sep), got
    assert got["models._unet"] is None, got
