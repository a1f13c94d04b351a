"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/mia_hip.h declares; argument errors surface as exceptions; no CPU fallback exists."""
import ctypes
import os

import pytest
import torch

import mia_hip


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    protos = mia_hip.parse_header()
    assert len(protos) >= 25
    l = ctypes.CDLL(mia_hip.LIB_PATH)
    for name in protos:
        assert hasattr(l, name), f"{name} declared in include/mia_hip.h but not exported"
    assert mia_hip.lib().mia_version() >= 100


def test_python_constants_match_header_defines():
    """Every enum / flag the ctypes side passes is declared in include/mia_hip.h with the same value."""
    d = mia_hip.parse_defines()
    pairs = {"MIA_F32": mia_hip.F32, "MIA_BF16": mia_hip.BF16, "MIA_NORM_INSTANCE": mia_hip.NORM_INSTANCE,
             "MIA_NORM_BATCH": mia_hip.NORM_BATCH, "MIA_LOSS_SOFTMAX": mia_hip.LOSS_SOFTMAX, "MIA_LOSS_DO_BG": mia_hip.LOSS_DO_BG,
             "MIA_LOSS_BATCH": mia_hip.LOSS_BATCH, "MIA_LOSS_SQUARED": mia_hip.LOSS_SQUARED, "MIA_LOSS_DENSE": mia_hip.LOSS_DENSE,
             "MIA_OPT_ADAM": mia_hip.OPT_ADAM, "MIA_OPT_ADAMW": mia_hip.OPT_ADAMW, "MIA_OPT_SGD": mia_hip.OPT_SGD,
             "MIA_CONV_G3S1": mia_hip.CONV_G3S1, "MIA_CONV_G3S2": mia_hip.CONV_G3S2, "MIA_CONV_G2S2": mia_hip.CONV_G2S2,
             "MIA_CONV_T3S2": mia_hip.CONV_T3S2, "MIA_CONV_T2S2": mia_hip.CONV_T2S2, "MIA_CONV_G1": mia_hip.CONV_G1,
             "MIA_WGRAD_3S1": mia_hip.WGRAD_3S1, "MIA_WGRAD_3S2": mia_hip.WGRAD_3S2, "MIA_WGRAD_2S2": mia_hip.WGRAD_2S2}
    for name, val in pairs.items():
        assert d.get(name) == val, (name, d.get(name), val)


def test_argument_errors_are_reported_without_a_gpu():
    l = mia_hip.lib()
    rc = l.mia_conv_mma(99, 0, None, 0, None, 0, None, 0, 0, 0, None, None, 0, None, 0, None, 1, 1, 1, 1, 1, None, None, None, None, None, None, None)
    assert rc < 0 and b"bad mode" in l.mia_last_error()
    with pytest.raises(mia_hip.MiaError):
        mia_hip.call("mia_grad_norm", None, ctypes.c_int64(0), ctypes.c_float(1.0), ctypes.c_float(1.0), None, None, None)


def test_no_cpu_fallback():
    from mia_hip import ops
    from losses.dice_loss import DiceLoss
    with pytest.raises(mia_hip.MiaError):
        DiceLoss(2, do_bg=True)(torch.zeros(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))
    with pytest.raises(mia_hip.MiaError):
        ops.to_nhwc(torch.zeros(1, 1, 4, 4), torch.float32)


def test_state_dict_keys_match_reference_checkpoint_format(golden_dir):
    import numpy as np
    from models.unet import UNet
    for tag, norm in (("instance", "instance"), ("batch", "batch")):
        d = np.load(os.path.join(golden_dir, f"unet_{tag}.npz"))
        want = {k[5:]: d[k].shape for k in d.files if k.startswith("init/")}
        torch.manual_seed(1337)
        m = UNet(2, 1, 3, [4, 8, 16], normalization=norm, dropout_prob=None)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == want
        # default torch init consumes the RNG like the reference: conv weights are bit-identical
        np.testing.assert_array_equal(m.state_dict()["encoder.levels.0.0.all.0.weight"].numpy(),
                                      d["init/encoder.levels.0.0.all.0.weight"])
        np.testing.assert_array_equal(m.state_dict()["decoder.seg_output.weight"].numpy(), d["init/decoder.seg_output.weight"])


def test_reference_api_surface():
    from losses.compound_losses import DiceAndCELoss
    from losses.dice_loss import DiceLoss
    from models.unet import UNet
    m = UNet(2, 1, 3, [4, 8], normalization="batch", dropout_prob=0.1)
    assert "decoder.seg_output.weight" in dict(m.named_parameters())
    assert hasattr(m, "encoder") and hasattr(m, "decoder") and hasattr(m, "get_enc_feature") and hasattr(m, "get_pixel_feature")
    l = DiceAndCELoss(dice_loss=DiceLoss, dice_kwargs={"num_classes": 2, "do_bg": True}, ce_loss=torch.nn.CrossEntropyLoss, ce_kwargs={})
    assert l.dice_loss.num_classes == 3 and hasattr(l, "ce_loss") and hasattr(l, "get_dice_loss") and hasattr(l, "get_ce_loss")
    with pytest.raises(KeyError):
        UNet(2, 1, 3, [4, 8], normalization="group")


def test_no_new_store_data_hazard_site_in_the_built_library():
    """ADVICE r4 / VERDICT r4 #14: the store-data hazard found on hardware in round 4 (a VALU write 2 wait states behind a 16-byte
    store reached the store's last lane phase; profiles/r05_store_hazard.txt) is guarded by a scan of the BUILT library's ISA:
    tools/check_store_hazard.py fails on a site closer than hipcc's own rule, and on a compiler-minimum site in a kernel family that
    tools/store_hazard_allow.json does not list with the bit-exact test that covers it."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_store_hazard.py"), "--quiet"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
