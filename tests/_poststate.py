"""Post-optimizer state check for the golden `post/*` vectors (round 5; replaces a blanket `atol = 2.5 * lr`, which a no-op or a
sign-flipped optimizer would have passed: Adam's FIRST step moves every weight by lr * g / (|g| + eps) ~ +-lr).

Where the reference gradient is clearly above noise -- |g| > strict_frac * max|g| of its tensor (the gradient tests bound our own
gradient's distance from it well below that, so the SIGN is certain) and |g| > 1e-5 (>> Adam's eps = 1e-8, so the step is +-lr to
1e-3) -- the post-state must agree within 0.05 * lr: a skipped step is off by lr there, a wrong sign by 2 * lr.  Elsewhere (conv bias
in front of a norm layer: analytically zero, pure rounding noise, in the reference too) only |delta| <= 2 * lr can hold.
Returns the number of strictly checked elements; callers assert that the model as a whole had some."""
import numpy as np


def check_post_adam(got, ref_post, ref_grad, lr, name, strict_frac=2e-2):
    got = np.asarray(got, dtype=np.float64)
    ref_post = np.asarray(ref_post, dtype=np.float64)
    d = np.abs(got - ref_post)
    if ref_grad is None:
        assert d.max() <= 2.02 * lr + 1e-6, (name, float(d.max()), lr)
        return 0
    g = np.abs(np.asarray(ref_grad, dtype=np.float64))
    strict = (g > strict_frac * g.max()) & (g > 1e-5)
    if strict.any():
        assert d[strict].max() <= 0.05 * lr, (name, "post-optimizer state off where the gradient sign is certain", float(d[strict].max()), lr)
    if (~strict).any():
        assert d[~strict].max() <= 2.02 * lr + 1e-6, (name, float(d[~strict].max()), lr)
    return int(strict.sum())
