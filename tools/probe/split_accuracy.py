"""Accuracy record of option f32_split (fp32 tensors, split-bf16 products) against the fp32 CPU oracle and the exact fp32 kernels:
max |logits - oracle|, loss distance, label-map disagreements, worst gradient rel-L2 vs the fp64 oracle.  Run on the GPU box:
    python tools/probe/split_accuracy.py > gpurun_out/split_accuracy.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-image-analysis_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import mia_hip  # noqa: E402
import test_gpu_configs as T  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
for channels, size, n in (([16, 32, 64], 128, 4), ([32, 64, 128, 256, 512], 256, 2), ([64, 128, 256, 512, 1024], 256, 2)):
    k1, norm, lr = 3, "instance", 1e-3
    m = T._model(dev, channels, norm, k1).train()
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, y = T._batch(n, size, seed=7, k1=k1)
    ref_logits, ref_loss, ref_grads, _, _ = T._oracle_step(state, x, y, k1, norm, lr)
    _, _, g64, _, _ = T._oracle_step(state, x, y, k1, norm, lr, dtype=torch.float64)
    loss_fn = T._loss_fn(k1)
    for flag in (0, 1):
        mia_hip.set_option("f32_split", flag)
        m.zero_grad(set_to_none=True)
        out = m(x.to(dev))
        loss = loss_fn(out, y.to(dev))
        loss.backward()
        o = out.detach().cpu()
        worst = max(T._grad_err(p.grad.cpu(), g64[nm])[0] for nm, p in m.named_parameters())
        cpu_worst = max(T._grad_err(ref_grads[nm], g64[nm])[0] for nm in g64)
        print(f"{channels[0]}..{channels[-1]} {size}x{size}x{n} f32_split={flag}: max|logits - oracle| {float((o - ref_logits).abs().max()):.2e} "
              f"(range {float(ref_logits.max() - ref_logits.min()):.2f}), |loss - oracle| {abs(loss.item() - ref_loss):.2e}, label maps differ at "
              f"{int((o.argmax(1) != ref_logits.argmax(1)).sum())} of {o.argmax(1).numel()} px, worst gradient rel-L2 from fp64 {worst:.2e} "
              f"(fp32 CPU oracle's worst {cpu_worst:.2e})", flush=True)
mia_hip.set_option("f32_split", 0)
