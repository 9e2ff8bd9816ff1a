"""Seeded-init and state-dict parity of the product parameter tree with the reference
(tests/golden/manifest.json was written by the reference's own DeepLab, seed 1337)."""
import json
import os

import torch

from uda_clr_amd.networks.deeplabv3 import DeepLab


def test_state_dict_matches_reference_manifest(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "manifest.json")))
    torch.manual_seed(man["seed"])
    m = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16, sync_bn=True, freeze_bn=False,
                method="prototype_full")
    sd = m.state_dict()
    assert len(sd) == man["n_state_keys"] == 675
    assert list(sd.keys()) == [e["key"] for e in man["entries"]]
    for e in man["entries"]:
        v = sd[e["key"]]
        assert list(v.shape) == e["shape"], e["key"]
        assert abs(float(v.double().sum()) - e["sum"]) <= 1e-9 * max(1.0, abs(e["sum"])), e["key"]
    params = list(m.parameters())
    assert len(params) == man["n_param_tensors"] == 186
    assert sum(p.numel() for p in params) == man["n_params"] == 5812135
    assert abs(float(sum(p.double().sum() for p in params)) - man["param_sum"]) < 1e-6


def test_resnet_state_dict_matches_reference_manifest(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "manifest_resnet.json")))
    torch.manual_seed(man["seed"])
    m = DeepLab(num_classes=2, backbone="resnet", output_stride=16, sync_bn=True, freeze_bn=False,
                method="prototype_full")
    sd = m.state_dict()
    assert len(sd) == man["n_state_keys"] == 687
    assert list(sd.keys()) == [e["key"] for e in man["entries"]]
    for e in man["entries"]:
        v = sd[e["key"]]
        assert list(v.shape) == e["shape"], e["key"]
        assert abs(float(v.double().sum()) - e["sum"]) <= 1e-9 * max(1.0, abs(e["sum"])), e["key"]
    assert sum(p.numel() for p in m.parameters()) == man["n_params"] == 59340391
    n1, n10 = sum(1 for _ in m.get_1x_lr_params()), sum(1 for _ in m.get_10x_lr_params())
    assert n1 + n10 == man["n_param_tensors"]


def test_cpu_forward_fails_loudly():
    import pytest
    m = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 3, 64, 64))
