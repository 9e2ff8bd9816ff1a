"""-m gpu: the device-side input pipeline (uda_normalize_tf, uda_field_smooth + uda_elastic_warp, uda_photometric_u8, and the
UDA_CLR_DEVICE_INPUT=2 path of TrainerBase._decode that chains them) reproduces, BYTE FOR BYTE, what the REFERENCE's
dataloaders/custom_transforms.py produced on the same seeded samples and random streams (tests/golden/input_pipeline.json)."""
import numpy as np
import pytest
import torch
from PIL import Image

import input_cases as ic
from make_golden_inputs import fundus_u8
from uda_clr_amd import ops
from uda_clr_amd.dataloaders import custom_transforms as tr
from uda_clr_amd.train_process._common import HipOps, TrainerBase

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


class _T(TrainerBase):
    def __init__(self):
        self.ops = HipOps()

    def _to(self, t):
        return t.to(DEV)


def _level2_records(transforms, img, lab, info, monkeypatch):
    """Run the worker side of UDA_CLR_DEVICE_INPUT=2 on one sample: same draws, outcomes only recorded."""
    monkeypatch.setattr(tr, "DEVICE_TAIL", 2)
    ic.seed_streams(info)
    s = {"image": Image.fromarray(img), "label": Image.fromarray(lab), "img_name": "s"}
    for t in transforms:
        s = t(s)
    s = tr.ToTensor()(tr.Normalize_tf()(s))
    monkeypatch.setattr(tr, "DEVICE_TAIL", 0)
    return {k: v[None] for k, v in s.items() if k != "img_name"}


@pytest.mark.parametrize("tag", ["ntf_small", "ntf_512"])
def test_normalize_tf_kernels_equal_reference_bytes(tag):
    names = ic.cases(tag)
    info = ic.META[names[0]]
    img, lab = fundus_u8(info["B"], info["H"], info["W"], info["seed"])
    image, mp, bd = ops.normalize_tf(torch.from_numpy(img).to(DEV), torch.from_numpy(lab).to(DEV))
    for name in names:
        b = ic.META[name]["b"]
        ic.expect(name, {"image": image[b].cpu().numpy(), "map": mp[b].cpu().numpy(), "boundary": bd[b].cpu().numpy()})


@pytest.mark.parametrize("name", ic.cases("elastic_small") + ic.cases("elastic_512"))
def test_elastic_kernels_equal_reference_bytes(name):
    """float64 Gaussian field in scipy's summation order + bilinear warp, given the uniform noise numpy drew for the reference."""
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    noise = torch.from_numpy(ic.elastic_noise(info["noise_seed"], info["H"], info["W"]))[:, None].to(DEV)
    io, lo = ops.elastic_deform(torch.from_numpy(img)[None].to(DEV), torch.from_numpy(lab)[None].to(DEV),
                                apply=torch.ones(1, dtype=torch.uint8, device=DEV), noise=noise)
    ic.expect(name, {"image": io[0].cpu().numpy(), "label": lo[0].cpu().numpy()})


@pytest.mark.parametrize("name", ic.cases("salt_pepper") + ic.cases("adjust_light") + ic.cases("eraser"))
def test_photometric_kernel_equals_reference_bytes(name, monkeypatch):
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    t = {"salt_pepper": tr.add_salt_pepper_noise, "adjust_light": tr.adjust_light, "eraser": tr.eraser}[name.split(".")[0]]()
    to_np = lambda s: {"image": np.array(s["image"]), "label": np.array(s["label"]), "img_name": "s"}
    rec = _level2_records([to_np, t], img, lab, info, monkeypatch)
    out = ops.photometric_u8(rec["image_u8"].to(DEV).contiguous(), rec["aug_sp_pos"].to(DEV), rec["aug_sp_n"].to(DEV),
                             rec["aug_sp_val"].to(DEV), rec["aug_lut"].to(DEV), rec["aug_erase"].to(DEV))
    ic.expect(name, {"image": out[0].cpu().numpy()})


@pytest.mark.parametrize("name", ic.cases("chain"))
def test_level2_device_chain_equals_reference_bytes(name, monkeypatch):
    """elastic -> salt-and-pepper -> gamma -> eraser -> Normalize_tf -> ToTensor entirely on the device from the workers'
    recorded draws (+ numpy's elastic noise): image / map / boundary equal the reference chain's tensors."""
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    batch = _level2_records([tr.elastic_transform(), tr.add_salt_pepper_noise(), tr.adjust_light(), tr.eraser()], img, lab, info, monkeypatch)
    assert int(batch["aug_elastic"][0, 0]) == info["fired"][0]
    batch["aug_noise"] = torch.from_numpy(ic.elastic_noise(info["noise_seed"], info["H"], info["W"]))[None]
    dec = _T()._decode(batch)
    ic.expect(name, {k: dec[k][0].cpu().numpy() for k in ("image", "map", "boundary")})
