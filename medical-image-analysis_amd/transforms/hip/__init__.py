"""HIP implementations of the reference's `src/transforms` classes (batched, per-sample parameters, device tensors only)."""
