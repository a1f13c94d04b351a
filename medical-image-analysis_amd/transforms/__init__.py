"""`transforms` of the MI355X drop-in.

The HIP implementations live in `transforms.hip` (same class names, constructors, `__call__(data)` /
`get_params_dict()` contract as `src/transforms/*.py`, plus `draw()` / `apply_batch()` for the batched pipeline).

Who serves `transforms.common` / `.image_transform` / `.joint_transform` / `.normalization`:
* this repo alone on sys.path: the HIP classes (thin alias modules next to this file);
* the reference behind this repo on sys.path (the `al_train` drop-in): the REFERENCE's own modules.  `al_train` calls the
  per-sample transforms on CPU tensors inside forked DataLoader workers (`src/datasets/fugc/fugc_dataset.py:140-164`,
  `--num-workers` default 1), where a HIP kernel cannot run; those workers keep the reference's CPU classes, and the GPU
  pipeline is applied after collation with `transforms.gpu_pipeline.BatchedAugment`.
`transforms.hip.*`, `transforms.gpu_pipeline` and `transforms.functional_hip` always resolve to this repo."""
from mia_hip.dropin import extend_over_reference

__path__ = extend_over_reference(__path__, __name__, reference_first=True)
