"""CPU oracle for the UNet-2D training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a PyTorch-CPU / numpy restatement of
the reference algorithm (trnKhanh/medical-image-analysis, `src/models/unet`,
`src/losses`, `src/transforms`, `src/scheduler`, `ALTrainer.train_step`).  It is
the checker for the HIP path and the CPU baseline that `bench.py` times.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product package (`medical-image-analysis_amd/`) never does.

Pinning status
--------------
* model / losses / poly-LR / train step / pure-torch transforms (gamma, noise,
  low-res, rot90, mirror, z-score, combinators): PINNED.  `oracle/gen_golden.py`
  imports the reference's own `blocks.py`, `unet.py`, `losses/*.py`,
  `scheduler/lr_scheduler.py` and `transforms/*.py` in the dev container and
  writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this
  restatement against those vectors.
* active-learning selectors (entropy / confidence / margin scores, k-centre
  greedy, k-means feature standardisation, BADGE gradient embeddings): PINNED
  since round 4 by `tests/golden/selectors.npz`, produced by the reference's own
  selector classes (`oracle/_refload.Ref.load_selectors`).
* transforms whose arithmetic lives in torchvision (RandomAffine,
  RandomRotation, RandomCrop2D, JointResize, RandomGaussianBlur,
  RandomContrast, RandomBrightness): PARITY UNPINNED.  torchvision is a
  third-party dependency, un-versioned in the reference's `pyproject.toml:19`,
  absent from /root/reference and not installed here.  They are restated from
  torchvision's published algorithm and pinned by analytic known-answer tests
  only (see `tests/test_oracle_transforms.py`).
* `UnetProcessor.denoise_one_mask` (`oracle/processor_ref.py`, scipy.ndimage): PARITY
  UNPINNED.  The reference calls cv2 (not importable here, no fixture held); the
  restatement follows OpenCV's documented uint8 arithmetic (rectangular dilate /
  erode ignoring outside pixels, the fixed small-kernel Gaussian table in 8.8
  fixed point with REFLECT_101, threshold > 127) and has known-answer tests.
"""
