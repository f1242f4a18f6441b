"""Checkpoint bridge, host logic: Lightning-style checkpoint <-> module state_dict."""
import collections

import torch

from mono_depth_estimation_amd import checkpoint
from mono_depth_estimation_amd.network import FCRN
from oracle import weights as W


def _net(seed):
    net = FCRN.ResNet(layers=50, output_size=(32, 48), out_channels=1, pretrained=False)
    W.fill_state_dict(net, seed)
    return net


def test_lightning_checkpoint_roundtrip(tmp_path):
    a, b = _net(1), _net(2)
    path = str(tmp_path / "epoch=3-val_loss=0.5.ckpt")
    ck = checkpoint.save_checkpoint(a, path, epoch=3, global_step=77, hyper_parameters={"method": "laina"})
    assert all(k.startswith("model.") for k in ck["state_dict"])
    assert set(k[len("model."):] for k in ck["state_dict"]) == set(a.state_dict())
    got = checkpoint.load_checkpoint(b, path)
    assert got["epoch"] == 3 and got["global_step"] == 77 and got["hyper_parameters"]["method"] == "laina"
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(va, vb), k


def test_checkpoint_key_handling():
    a, b = _net(3), _net(4)
    sd = a.state_dict()
    # a plain state_dict (no prefix) and a Lightning dict with foreign entries next to the model's
    assert list(checkpoint.model_state_from_checkpoint(sd)) == list(sd)
    ck = {"state_dict": collections.OrderedDict([("criterion.weight", torch.zeros(1))] + [("model." + k, v) for k, v in sd.items()])}
    assert list(checkpoint.model_state_from_checkpoint(ck)) == list(sd)
    checkpoint.load_checkpoint(b, ck)
    assert torch.equal(b.conv3.weight, a.conv3.weight) and torch.equal(b.bn1.running_var, a.bn1.running_var)
    # the parameter order the optimiser bridge relies on is the reference's: encoder group, then decoder group
    g1, g10 = checkpoint._param_groups(a)
    assert g1[0] is a.conv1.weight and g10[0] is a.conv2.weight and g10[-1] is a.conv3.weight
    assert len(g1) + len(g10) == len(list(a.parameters()))


def test_vnl_resnext_key_bridge_matches_the_reference():
    """N3: checkpoint.vnl_resnext_keys against the pairs the reference's own convert_state_dict_resnext (VNL.py:44-67)
    produced (tests/golden/vnl_keymap.json), and load_vnl_imagenet_weights into the HIP module's body."""
    import json
    import os
    from mono_depth_estimation_amd import checkpoint
    from mono_depth_estimation_amd.network import VNL
    from oracle import nets
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vnl_keymap.json")))
    src = {k: i for i, k in enumerate(list(g["pairs"]) + g["dropped"])}
    got = checkpoint.vnl_resnext_keys(src)
    assert {k: got_k for got_k, i in got.items() for k in [list(src)[i]]} == g["pairs"]
    torch.manual_seed(0)
    net = VNL.MetricDepthModel(nets.vnl_params())
    body = net.depth_model.encoder_modules.bottomup
    own = body.state_dict()
    assert set(g["pairs"].values()) == {k for k in own if not k.endswith("num_batches_tracked")}      # every body tensor is reachable
    fake = {s: torch.full_like(own[d], float(i % 7)) for i, (s, d) in enumerate(g["pairs"].items())}
    fake["10.1.weight"] = torch.zeros(3)
    assert checkpoint.load_vnl_imagenet_weights(net, fake) == []
    for i, (s, d) in enumerate(g["pairs"].items()):
        assert float(body.state_dict()[d].flatten()[0]) == float(i % 7), d
