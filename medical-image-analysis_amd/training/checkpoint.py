"""Checkpoint I/O in the reference's on-disk format (`src/training/al_trainer.py:521-547,:1694-1733`).

``model.pth`` is ``model.state_dict()`` (fp32, reference keys and OIHW shapes) -- interchangeable with the reference in
both directions; a ``{"model": state_dict}`` wrapper is accepted on load like ``load_model_checkpoint`` (:527-530).
``training_state.pth`` holds ``optimizer`` / ``current_iter`` / ``current_epoch`` / ``current_round`` / ``data_list``
(:1694-1702).  The reference pickles the optimizer OBJECT there (":1697 ``"optimizer": self.optimizer``) and then feeds
it to ``Optimizer.load_state_dict`` (:1714), which cannot work; this module writes the optimizer as a
``torch.optim.{Adam,AdamW,SGD}``-compatible ``state_dict`` instead (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``
or ``momentum_buffer``, parameters indexed in ``model.parameters()`` order), so either side can resume from it, and
everything is loadable with ``torch.load(..., weights_only=True)`` (tensors, numbers, strings, lists, dicts only).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, Optional

import torch

from mia_hip import OPT_SGD, ops


def model_state_for_save(model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    """Detached copies: FlatOptimizer parameters are views of one flat buffer and must not drag it into the file."""
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def save_model_checkpoint(model: torch.nn.Module, ckpt) -> None:
    torch.save(model_state_for_save(model), ckpt)


def load_model_checkpoint(model: torch.nn.Module, ckpt, map_location=None) -> None:
    sd = torch.load(ckpt, map_location=map_location, weights_only=True)
    if "model" in sd:
        sd = sd["model"]
    model.load_state_dict(sd)  # copy_ into the existing (possibly flat-buffer) storage
    ops.bump_param_epoch()  # packed weights are stale now


def optimizer_state_to_torch(opt, model: torch.nn.Module) -> dict:
    """FlatOptimizer -> the state_dict a torch.optim.Adam/AdamW/SGD over ``model.parameters()`` would produce."""
    index = {id(p): i for i, p in enumerate(p for p in model.parameters() if p.requires_grad)}
    state = {}
    if opt.step_count > 0:
        for p, o in zip(opt.params, opt.offsets):
            if id(p) not in opt.stepped:
                continue  # never received a gradient (e.g. unused deep-supervision heads): torch.optim holds no state for it
            n, i = p.numel(), index[id(p)]
            if opt.kind == OPT_SGD:
                state[i] = {"momentum_buffer": opt.m[o:o + n].view(p.shape).clone()}
            else:
                state[i] = {"step": torch.tensor(float(opt.step_count)), "exp_avg": opt.m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": opt.v[o:o + n].view(p.shape).clone()}
    g = opt.param_groups[0]
    group = {"lr": float(g["lr"]), "weight_decay": float(g["weight_decay"]), "params": sorted(index.values())}
    if opt.kind == OPT_SGD:
        group.update(momentum=float(g["momentum"]), dampening=0.0, nesterov=False)
    else:
        group.update(betas=[float(b) for b in g["betas"]], eps=float(g["eps"]), amsgrad=False)
    return {"state": state, "param_groups": [group], "name": opt.name}


def optimizer_state_from_torch(opt, model: torch.nn.Module, sd: dict) -> None:
    params = [p for p in model.parameters() if p.requires_grad]
    index = {id(p): i for i, p in enumerate(params)}
    state = sd.get("state", {})
    step = 0
    opt.stepped = set()
    opt.m.zero_()
    if opt.v is not None:
        opt.v.zero_()
    with torch.no_grad():
        for p, o in zip(opt.params, opt.offsets):
            st = state.get(index[id(p)], state.get(str(index[id(p)])))
            if not st:
                continue
            n = p.numel()
            opt.stepped.add(id(p))
            if opt.kind == OPT_SGD:
                if st.get("momentum_buffer") is not None:
                    opt.m[o:o + n].copy_(st["momentum_buffer"].reshape(-1))
                    step = max(step, 1)
            else:
                opt.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                opt.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(float(st["step"])))
    if "step_count" in sd:
        step = int(sd["step_count"])
    opt.step_count = step
    g, src = opt.param_groups[0], sd["param_groups"][0]
    for k in ("lr", "weight_decay", "eps", "momentum"):
        if k in src:
            g[k] = float(src[k])
    if "betas" in src:
        g["betas"] = tuple(float(b) for b in src["betas"])


def save_state_dict(engine, save_path, save_training_state: bool = False, current_epoch: int = 0, current_round: int = 0,
                    data_list: Optional[list] = None) -> None:
    """al_trainer.py:1719-1733: ``model.pth`` always, ``training_state.pth`` on request."""
    save_path = Path(save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    save_model_checkpoint(engine.model, save_path / "model.pth")
    if save_training_state:
        st = {"optimizer": optimizer_state_to_torch(engine.optimizer, engine.model),
              # the reference writes self.current_iter as it stands when the state is saved (:1698), i.e. AFTER train_step's
              # `self.current_iter += 1` (:1399), and adds 1 again on load (:1716): same value on disk here, same quirk on load
              "current_iter": int(engine.current_iter),
              "current_epoch": int(current_epoch), "current_round": int(current_round), "data_list": list(data_list or [])}
        st["optimizer"]["step_count"] = int(engine.optimizer.step_count)
        torch.save(st, save_path / "training_state.pth")


def load_state_dict(engine, save_path, map_location=None) -> dict:
    """al_trainer.py:1704-1717.  Returns {"current_epoch", "current_round", "data_list"} already offset by +1 like the
    reference; ``engine.current_iter`` = stored value + 1 exactly as the reference does (:1716) -- since the stored value
    is already the count of finished iterations, a resumed run skips one poly-LR step, in the reference and here alike
    (files written by either side resume identically on either side)."""
    save_path = Path(save_path)
    out = {}
    model_path, ts_path = save_path / "model.pth", save_path / "training_state.pth"
    if model_path.is_file():
        load_model_checkpoint(engine.model, model_path, map_location or engine.optimizer.flat_param.device)
    if ts_path.is_file():
        ts = torch.load(ts_path, map_location=map_location or engine.optimizer.flat_param.device, weights_only=True)
        optimizer_state_from_torch(engine.optimizer, engine.model, ts["optimizer"])
        engine.current_iter = int(ts["current_iter"]) + 1
        out = {"current_epoch": int(ts["current_epoch"]) + 1, "current_round": int(ts["current_round"]) + 1,
               "data_list": ts.get("data_list", [])}
    return out
