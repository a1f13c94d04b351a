"""Checkpoint I/O host logic (CPU): model.pth / training_state.pth in the reference's layout
(src/training/al_trainer.py:521-547,:1694-1733), round trip + interchange with torch.optim state dicts."""
import numpy as np
import torch

from losses.compound_losses import DiceAndCELoss
from models.unet import UNet
from training import checkpoint
from training.engine import TrainEngine


def _engine(opt="adam", seed=0):
    torch.manual_seed(seed)
    m = UNet(2, 1, 2, [4, 8, 16], normalization="batch", dropout_prob=None)
    return TrainEngine(m, DiceAndCELoss(dice_kwargs=dict(num_classes=2, do_bg=True)), opt, {"weight_decay": 5e-4},
                       start_lr=1e-3, num_iters=50, lr_warmup_iter=5)


def _fake_progress(eng, steps=3):
    g = torch.Generator().manual_seed(5)
    o = eng.optimizer
    for p, off in zip(o.params, o.offsets):  # the 16-byte alignment gaps between tensors stay zero
        o.m[off:off + p.numel()].copy_(torch.randn(p.numel(), generator=g))
        if o.v is not None:
            o.v[off:off + p.numel()].copy_(torch.rand(p.numel(), generator=g))
    o.step_count = steps
    o.stepped = {id(p) for p in o.params}
    o.param_groups[0]["lr"] = 7e-4
    eng.current_iter = steps


def test_round_trip_and_reference_layout(tmp_path, golden_dir):
    a = _engine(seed=1)
    _fake_progress(a)
    a.save_state_dict(tmp_path, save_training_state=True, current_epoch=4, current_round=2, data_list=["case_1", "case_7"])
    sd = torch.load(tmp_path / "model.pth", weights_only=True)  # plain tensors only
    assert list(sd.keys()) == list(a.model.state_dict().keys())
    assert all(v.dtype in (torch.float32, torch.int64) for v in sd.values())
    # no flat-buffer storage dragged into the file
    assert max(v.untyped_storage().nbytes() for v in sd.values()) <= max(v.numel() * v.element_size() for v in sd.values())
    assert "encoder.levels.0.0.all.0.weight" in sd and "decoder.seg_output.weight" in sd
    assert "encoder.levels.0.0.all.2.num_batches_tracked" in sd
    b = _engine(seed=2)
    extra = b.load_state_dict(tmp_path)
    np.testing.assert_array_equal(a.optimizer.flat_param.numpy(), b.optimizer.flat_param.numpy())
    np.testing.assert_array_equal(a.optimizer.m.numpy(), b.optimizer.m.numpy())
    np.testing.assert_array_equal(a.optimizer.v.numpy(), b.optimizer.v.numpy())
    # reference semantics (al_trainer.py:1698 / :1716): the file holds current_iter as it stands at save time (= finished
    # iterations, train_step has already incremented it, :1399) and the loader adds 1
    assert torch.load(tmp_path / "training_state.pth", weights_only=True)["current_iter"] == 3
    assert b.optimizer.step_count == 3 and b.current_iter == 4 and abs(b.optimizer.param_groups[0]["lr"] - 7e-4) < 1e-12
    assert extra == {"current_epoch": 5, "current_round": 3, "data_list": ["case_1", "case_7"]}
    # parameters are still views of the flat buffer after loading
    p0 = b.optimizer.params[0]
    assert p0.data_ptr() == b.optimizer.flat_param.data_ptr() + 4 * b.optimizer.offsets[0]
    # {"model": ...} wrapper accepted (al_trainer.py:527-530)
    torch.save({"model": sd}, tmp_path / "wrapped.pth")
    c = _engine(seed=3)
    checkpoint.load_model_checkpoint(c.model, tmp_path / "wrapped.pth")
    np.testing.assert_array_equal(a.optimizer.flat_param.numpy(), c.optimizer.flat_param.numpy())


def test_optimizer_state_interchanges_with_torch_optim():
    for name, cls in (("adam", torch.optim.Adam), ("adamw", torch.optim.AdamW), ("sgd", torch.optim.SGD)):
        a = _engine(name, seed=1)
        _fake_progress(a, steps=4)
        exported = checkpoint.optimizer_state_to_torch(a.optimizer, a.model)
        kw = dict(momentum=0.9) if name == "sgd" else {}
        t = cls(a.model.parameters(), lr=1e-3, **kw)
        t.load_state_dict({k: v for k, v in exported.items() if k in ("state", "param_groups")})
        params = list(a.model.parameters())
        for p, o in zip(a.optimizer.params, a.optimizer.offsets):
            st = t.state[p]
            key = "momentum_buffer" if name == "sgd" else "exp_avg"
            np.testing.assert_array_equal(st[key].numpy().ravel(), a.optimizer.m[o:o + p.numel()].numpy())
            if name != "sgd":
                assert float(st["step"]) == 4.0
        assert abs(t.param_groups[0]["lr"] - 7e-4) < 1e-12 and len(t.param_groups[0]["params"]) == len(params)
        # and back: a torch optimizer's own state_dict loads into the flat buffers
        b = _engine(name, seed=1)
        checkpoint.optimizer_state_from_torch(b.optimizer, b.model, t.state_dict())
        np.testing.assert_array_equal(a.optimizer.m.numpy(), b.optimizer.m.numpy())
        if name != "sgd":
            np.testing.assert_array_equal(a.optimizer.v.numpy(), b.optimizer.v.numpy())
            assert b.optimizer.step_count == 4


def test_unused_parameters_export_no_optimizer_state():
    """torch.optim keeps no state for a parameter that never had a gradient (unused deep-supervision heads): neither
    does the exported state_dict."""
    a = _engine("adam", seed=1)
    _fake_progress(a, steps=2)
    unused = a.optimizer.params[:2]
    a.optimizer.stepped -= {id(p) for p in unused}
    exported = checkpoint.optimizer_state_to_torch(a.optimizer, a.model)
    index = {id(p): i for i, p in enumerate(a.model.parameters())}
    assert all(index[id(p)] not in exported["state"] for p in unused)
    assert len(exported["state"]) == len(a.optimizer.params) - 2
    t = torch.optim.Adam(a.model.parameters(), lr=1e-3)
    t.load_state_dict({k: v for k, v in exported.items() if k in ("state", "param_groups")})
    assert all(p not in t.state for p in unused)
    b = _engine("adam", seed=1)
    checkpoint.optimizer_state_from_torch(b.optimizer, b.model, exported)
    assert b.optimizer.stepped == {id(p) for p in b.optimizer.params[2:]}
