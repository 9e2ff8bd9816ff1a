"""not-gpu: this repo's CPU transform chain (uda_clr_amd.dataloaders.custom_transforms) reproduces, byte for byte, what the
REFERENCE's dataloaders/custom_transforms.py produced on the same seeded samples and random streams
(tests/golden/input_pipeline.json, written by tests/golden/make_golden.py `input`): Normalize_tf + GetBoundary + ToTensor,
elastic_transform, add_salt_pepper_noise, adjust_light, eraser, and the four array transforms chained (which pins the
number and order of draws from `random` / `np.random`)."""
import numpy as np
import pytest
from PIL import Image

import input_cases as ic
from uda_clr_amd.dataloaders import custom_transforms as tr


def _tensors(s):
    return {k: s[k].numpy() for k in ("image", "map", "boundary")}


@pytest.mark.parametrize("name", ic.cases("ntf_small") + ic.cases("ntf_512"))
def test_normalize_tf_totensor(name):
    img, lab = ic.sample_of(ic.META[name])
    s = tr.ToTensor()(tr.Normalize_tf()({"image": Image.fromarray(img), "label": Image.fromarray(lab), "img_name": "s"}))
    ic.expect(name, _tensors(s))


@pytest.mark.parametrize("name", ic.cases("elastic_small") + ic.cases("elastic_512"))
def test_elastic_transform(name):
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    ic.seed_streams(info)
    with ic.seeded_noise(info["noise_seed"]):
        o = tr.elastic_transform()({"image": Image.fromarray(img), "label": Image.fromarray(lab), "img_name": "s"})
    ic.expect(name, {"image": o["image"], "label": o["label"]})


@pytest.mark.parametrize("name", ic.cases("salt_pepper") + ic.cases("adjust_light") + ic.cases("eraser"))
def test_photometric_transform(name):
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    ic.seed_streams(info)
    t = {"salt_pepper": tr.add_salt_pepper_noise, "adjust_light": tr.adjust_light, "eraser": tr.eraser}[name.split(".")[0]]()
    o = t({"image": img.copy(), "label": lab, "img_name": "s"})
    ic.expect(name, {"image": np.asarray(o["image"])})


@pytest.mark.parametrize("name", ic.cases("chain"))
def test_array_part_of_the_training_chain(name):
    """elastic -> salt-and-pepper -> gamma -> eraser -> Normalize_tf -> ToTensor (train_use_fix_initial.py:153-159) from ONE
    seed per sample: equal outputs mean every transform consumed `random` / `np.random` exactly like the reference's."""
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    ic.seed_streams(info)
    s = {"image": Image.fromarray(img), "label": Image.fromarray(lab), "img_name": "s"}
    with ic.seeded_noise(info["noise_seed"]):
        for t in (tr.elastic_transform(), tr.add_salt_pepper_noise(), tr.adjust_light(), tr.eraser()):
            s = t(s)
    ic.expect(name, _tensors(tr.ToTensor()(tr.Normalize_tf()(s))))


@pytest.mark.parametrize("name", ic.cases("chain"))
def test_level2_records_applied_in_chain_order_equal_reference(name, monkeypatch):
    """UDA_CLR_DEVICE_INPUT=2 worker side (draws recorded, not applied) + a numpy statement of what the Trainer's kernels do with
    the records (scipy elastic on the reference's noise -> scatter -> table -> box -> Normalize_tf): the reference's tensors.
    The HIP kernels run the same comparison in tests/test_input_golden_gpu.py."""
    from scipy import ndimage
    from test_dataloaders_cpu import numpy_apply_recorded
    info = ic.META[name]
    img, lab = ic.sample_of(info)
    monkeypatch.setattr(tr, "DEVICE_TAIL", 2)
    ic.seed_streams(info)
    s = {"image": Image.fromarray(img), "label": Image.fromarray(lab), "img_name": "s"}
    for t in (tr.elastic_transform(), tr.add_salt_pepper_noise(), tr.adjust_light(), tr.eraser()):
        s = t(s)
    rec = tr.ToTensor()(tr.Normalize_tf()(s))
    monkeypatch.setattr(tr, "DEVICE_TAIL", 0)
    image, label = rec["image_u8"].numpy(), rec["label_u8"].numpy()
    if int(rec["aug_elastic"][0]):
        H, W = label.shape
        nz = ic.elastic_noise(info["noise_seed"], H, W)
        dx, dy = (ndimage.gaussian_filter(n, 0.08 * W, mode="constant", cval=0) * (2 * W) for n in nz)
        gx, gy = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        idx = np.reshape(gx + dx, (-1, 1)), np.reshape(gy + dy, (-1, 1))
        image = np.stack([ndimage.map_coordinates(image[:, :, c], idx, order=1).reshape(H, W) for c in range(3)], -1)
        label = ndimage.map_coordinates(label, idx, order=1, mode="nearest").reshape(H, W)
    rec["image_u8"] = __import__("torch").from_numpy(np.ascontiguousarray(image))
    out = tr.ToTensor()(tr.Normalize_tf()({"image": numpy_apply_recorded(rec), "label": label, "img_name": "s"}))
    ic.expect(name, _tensors(out))
