"""Package-path merging for the drop-in layout.

The reference installs ``src/*`` as TOP-LEVEL packages (`pyproject.toml:38-42`): ``models``, ``losses``, ``transforms``,
``scheduler``, ``training``, ``metric``, ``activelearning``.  This repo ships regular packages with the same names (that
is what makes ``from models.unet import UNet`` in `src/training/al_trainer.py:47` pick the MI355X model), and regular
packages do not merge across ``sys.path`` entries on their own: without help, ``training.al_trainer``,
``models._unet``, ``metric.metric`` and ``scheduler.ramps`` (modules only the reference has) would stop resolving as soon
as this repo comes first on the path.  Each colliding package therefore calls :func:`extend_over_reference` from its
``__init__``: the same-named directories of the REFERENCE's ``src/`` (recognised by ``training/al_trainer.py`` or named by
``MIA_REFERENCE_SRC``; unrelated ``models`` / ``transforms`` packages elsewhere on ``sys.path`` and the working directory are
ignored) are appended to the package's ``__path__``, so a module this repo does not define still resolves to the
reference's file, while every module this repo does define wins.  The scan runs once, when the package is first imported:
put the reference on ``sys.path`` (or set ``MIA_REFERENCE_SRC`` and add it) BEFORE importing any of these packages.
"""
from __future__ import annotations

import os
import sys
from typing import List


REFERENCE_MARKER = os.path.join("training", "al_trainer.py")  # a file only the reference's src/ holds


def is_reference_src(entry: str) -> bool:
    """True when `entry` (a sys.path entry) is the reference's ``src/`` directory: it holds ``training/al_trainer.py``, or it
    is the directory named by ``MIA_REFERENCE_SRC``.  Generic package names such as ``models`` / ``transforms`` / ``training``
    are common in site-packages and in working directories; only a directory recognisably the reference may be merged."""
    if not entry:  # '' = the current working directory: never merged
        return False
    explicit = os.environ.get("MIA_REFERENCE_SRC")
    if explicit and os.path.realpath(entry) == os.path.realpath(explicit):
        return True
    return os.path.isfile(os.path.join(entry, REFERENCE_MARKER))


def same_named_dirs(own_dir: str, name: str) -> List[str]:
    """Directories ``<reference src>/<name>`` that hold a regular package of that name, other than `own_dir`; only sys.path
    entries that :func:`is_reference_src` accepts are considered."""
    own = os.path.realpath(own_dir)
    found = []
    for entry in sys.path:
        if not isinstance(entry, str) or not is_reference_src(entry):
            continue
        cand = os.path.join(entry, *name.split("."))
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
