"""Package-path merging for the drop-in layout.

The reference installs ``src/*`` as TOP-LEVEL packages (`pyproject.toml:38-42`): ``models``, ``losses``, ``transforms``,
``scheduler``, ``training``, ``metric``, ``activelearning``.  This repo ships regular packages with the same names (that
is what makes ``from models.unet import UNet`` in `src/training/al_trainer.py:47` pick the MI355X model), and regular
packages do not merge across ``sys.path`` entries on their own: without help, ``training.al_trainer``,
``models._unet``, ``metric.metric`` and ``scheduler.ramps`` (modules only the reference has) would stop resolving as soon
as this repo comes first on the path.  Each colliding package therefore calls :func:`extend_over_reference` from its
``__init__``: the same-named directories found LATER on ``sys.path`` are appended to the package's ``__path__``, so a
module this repo does not define still resolves to the reference's file, while every module this repo does define wins.
"""
from __future__ import annotations

import os
import sys
from typing import List


def same_named_dirs(own_dir: str, name: str) -> List[str]:
    """Directories ``<sys.path entry>/<name>`` that hold a regular package of that name, other than `own_dir`."""
    own = os.path.realpath(own_dir)
    found = []
    for entry in sys.path:
        if not isinstance(entry, str):
            continue
        cand = os.path.join(entry or os.curdir, *name.split("."))
        if not os.path.isfile(os.path.join(cand, "__init__.py")):
            continue
        real = os.path.realpath(cand)
        if real != own and real not in found:
            found.append(real)
    return found


def extend_over_reference(path: List[str], name: str, reference_first: bool = False) -> List[str]:
    """New ``__path__`` for package `name`: this repo's directory plus the same-named package directories further down
    ``sys.path``.  ``reference_first=True`` puts the others in FRONT (used by ``transforms`` only: the per-sample
    transforms run on CPU tensors inside forked DataLoader workers, where no HIP kernel can serve them, so with the
    reference present its own CPU classes keep that job and the HIP classes live under ``transforms.hip``)."""
    own = [p for p in path]
    others = same_named_dirs(own[0], name) if own else []
    return (others + own) if reference_first else (own + others)
