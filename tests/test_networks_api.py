"""The networks module API is a drop-in for the reference's: names, constructor arguments, state_dict keys,
checkpoint format, hparams, quirks (no GPU needed: nothing here runs a forward)."""
import os

import numpy as np
import pytest
import torch
import yaml

from music_mixing_style_transfer_amd import networks
from music_mixing_style_transfer_amd.utils import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_star_exports():
    for name in ("FXencoder", "TCNModel", "TCNBlock", "FiLM", "Res_ConvBlock", "ConvBlock", "Conv1d_layer"):
        assert hasattr(networks, name)


def test_state_dict_keys_equal_the_reference_modules():
    g = np.load(os.path.join(GOLD, "nets_tiny.npz"))
    enc = networks.FXencoder({"channels": [4, 8, 8], "kernels": [5, 4, 3], "strides": [2, 2, 1], "dilation": [1, 1, 1],
                              "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"})
    assert list(enc.state_dict().keys()) == list(g["tiny_enc_keys"])
    tcn = networks.TCNModel(nparams=16, ninputs=2, noutputs=2, nblocks=4, dilation_growth=2, kernel_size=5,
                            channel_width=8, stack_size=15, cond_dim=16, causal=False)
    assert list(tcn.state_dict().keys()) == list(g["tiny_tcn_keys"])
    assert tcn.compute_receptive_field() == int(g["tiny_tcn_rf"])
    assert tcn.hparams.kernel_size == 5 and tcn.hparams.nblocks == 4 and tcn.hparams.causal is False


def test_default_config_counts_and_reference_checkpoint_format(tmp_path):
    g = np.load(os.path.join(GOLD, "nets_full.npz"))
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfgs = yaml.full_load(f)
    cfg = cfgs["Effects_Encoder"]["default"]
    enc = networks.FXencoder(cfg)
    assert cfg["channels"][0] == 2 and len(cfg["channels"]) == 13          # in-place insert, like the reference
    t = cfgs["TCN"]["default"]
    tcn = networks.TCNModel(nparams=t["condition_dimension"], ninputs=2, noutputs=2, nblocks=t["nblocks"],
                            dilation_growth=t["dilation_growth"], kernel_size=t["kernel_size"],
                            channel_width=t["channel_width"], stack_size=t["stack_size"],
                            cond_dim=t["condition_dimension"], causal=t["causal"])
    assert len(enc.state_dict()) == int(g["enc_nkeys"]) == 168
    assert len(tcn.state_dict()) == int(g["tcn_nkeys"]) == 128
    assert sum(p.numel() for p in enc.parameters()) == int(g["enc_nparams"])
    assert sum(p.numel() for p in tcn.parameters()) == int(g["tcn_nparams"])
    assert tcn.compute_receptive_field() == 229363
    assert [b.dilation for b in tcn.blocks] == [2 ** n for n in range(14)]
    # checkpoint written the way the reference's trainer does, loaded the way style_transfer.py:94-108 does
    sd = synth.tcn_state_dict(seed=0)
    path = str(tmp_path / "MixFXcloner_ps.pt")
    synth.save_reference_format_checkpoint(path, sd)
    ckpt = torch.load(path, map_location="cpu")
    stripped = {k[7:]: v for k, v in ckpt["model"].items()}
    assert tcn.load_state_dict(stripped).missing_keys == []
    with pytest.raises(RuntimeError):
        tcn.load_state_dict({k: v for k, v in list(stripped.items())[:-1]})   # strict, like the reference


def test_no_torch_fallback():
    tcn = networks.TCNModel(nparams=16, ninputs=2, noutputs=2, nblocks=2, kernel_size=15, channel_width=128, cond_dim=16,
                            dilation_growth=2, stack_size=15)
    with pytest.raises((RuntimeError, ImportError)):
        tcn(torch.zeros(1, 2, 64), torch.zeros(1, 16))            # CPU tensor: refused, never computed on the host
    # the building blocks run on their own too - on the MI355X only: a CPU tensor is refused, never computed on the host
    with pytest.raises((RuntimeError, ImportError)):
        tcn.blocks[0](torch.zeros(1, 2, 64), torch.zeros(1, 16))
    with pytest.raises((RuntimeError, ImportError)):
        networks.FiLM(16, 8)(torch.zeros(1, 8, 4), torch.zeros(1, 16))
    with pytest.raises((RuntimeError, ImportError)):
        networks.Conv1d_layer(2, 4, 5)(torch.zeros(1, 2, 64))
