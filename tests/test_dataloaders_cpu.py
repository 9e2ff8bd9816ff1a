"""not-gpu: the data harness behind the reference's import names (dataset layout, transform chain of
train_use_fix_initial.py:150-166, sample format of custom_transforms.py:496-507)."""
import random

import numpy as np
import torch
from torch.utils.data import DataLoader

from uda_clr_amd.dataloaders import custom_transforms as tr
from uda_clr_amd.dataloaders import fundus_dataloader as DL
from uda_clr_amd.dataloaders.synthetic import write_dataset
from uda_clr_amd.dropin import install


class Compose:
    def __init__(self, ts): self.ts = ts
    def __call__(self, s):
        for t in self.ts:
            s = t(s)
        return s


def test_dataset_and_transform_chain(tmp_path):
    random.seed(0); np.random.seed(0)
    write_dataset(str(tmp_path), "refuge", "train", 5, size=160, seed=1)
    train_tf = Compose([tr.RandomScaleCrop(128), tr.RandomRotate(), tr.RandomFlip(), tr.elastic_transform(),
                        tr.add_salt_pepper_noise(), tr.adjust_light(), tr.eraser(), tr.Normalize_tf(), tr.ToTensor()])
    ds = DL.FundusSegmentation(base_dir=str(tmp_path), dataset="refuge", split="train", transform=train_tf)
    assert len(ds) == 5
    for _ in range(3):                                    # several random draws of every branch
        for batch in DataLoader(ds, batch_size=2, shuffle=True, num_workers=0):
            img, mp, bd = batch["image"], batch["map"], batch["boundary"]
            assert img.shape[1:] == (3, 128, 128) and mp.shape[1:] == (2, 128, 128) and bd.shape[1:] == (1, 128, 128)
            assert img.dtype == mp.dtype == bd.dtype == torch.float32
            assert -1.0 <= float(img.min()) and float(img.max()) <= 1.0
            assert set(torch.unique(mp).tolist()) <= {0.0, 1.0}
            assert bool((mp[:, 0] <= mp[:, 1]).all()), "cup must be a subset of disc"
            assert 0.0 <= float(bd.min()) and float(bd.max()) <= 1.0 and float(bd.max()) > 0.2
    test_tf = Compose([tr.RandomCrop(128), tr.Normalize_tf(), tr.ToTensor()])
    s = DL.FundusSegmentation(base_dir=str(tmp_path), dataset="refuge", split="train", transform=test_tf)[0]
    assert s["map"].sum() > 0 and isinstance(s["img_name"], str)


def test_deferred_tail_hands_over_uint8_and_the_trainer_decodes_it(tmp_path, monkeypatch):
    """UDA_CLR_DEVICE_INPUT=1: Normalize_tf + ToTensor emit the uint8 image and grey mask; TrainerBase._decode turns the batch
    into image / map / boundary through ops.normalize_tf (here: a stand-in that runs the CPU Normalize_tf, so the plumbing is
    checked without a GPU; the kernel itself is compared with scipy bit for bit in tests/kernel_cases.py)."""
    from uda_clr_amd.train_process._common import TrainerBase
    write_dataset(str(tmp_path), "refuge", "train", 3, size=96, seed=2)
    chain = lambda: Compose([tr.RandomCrop(96), tr.Normalize_tf(), tr.ToTensor()])
    plain = DL.FundusSegmentation(base_dir=str(tmp_path), dataset="refuge", split="train", transform=chain())
    want = next(iter(DataLoader(plain, batch_size=3, shuffle=False, num_workers=0)))
    monkeypatch.setattr(tr, "DEVICE_TAIL", True)
    got = next(iter(DataLoader(plain, batch_size=3, shuffle=False, num_workers=0)))
    assert set(got) == {"image_u8", "label_u8", "img_name"}
    assert got["image_u8"].dtype == got["label_u8"].dtype == torch.uint8
    assert got["image_u8"].shape == (3, 96, 96, 3) and got["label_u8"].shape == (3, 96, 96)
    monkeypatch.setattr(tr, "DEVICE_TAIL", False)

    def cpu_normalize_tf(img_u8, lab_u8):
        outs = [tr.ToTensor()(tr.Normalize_tf()({"image": i.numpy(), "label": l.numpy(), "img_name": ""}))
                for i, l in zip(img_u8, lab_u8)]
        return tuple(torch.stack([o[k] for o in outs]) for k in ("image", "map", "boundary"))

    class T(TrainerBase):
        def __init__(self):
            self.ops = type("Ops", (), {"normalize_tf": staticmethod(cpu_normalize_tf)})()
        def _to(self, t): return t
    dec = T()._decode(got)
    for k in ("image", "map", "boundary"):
        assert torch.equal(dec[k], want[k]), k
    assert T()._decode(want) is want


class _ToNumpy:
    """what elastic_transform leaves behind when it does not fire (numpy arrays), without its random draw"""
    def __call__(self, s):
        return {"image": np.array(s["image"]), "label": np.array(s["label"]), "img_name": s["img_name"]}


def _photometric_chain():
    return Compose([tr.RandomCrop(96), _ToNumpy(), tr.add_salt_pepper_noise(), tr.adjust_light(), tr.eraser(), tr.Normalize_tf(), tr.ToTensor()])


def numpy_apply_recorded(s):
    """CPU statement of what the Trainer does with the recorded outcomes (uda_photometric_u8's arithmetic)."""
    img = s["image_u8"].numpy().copy()
    n = int(s["aug_sp_n"][0])
    pos = s["aug_sp_pos"].numpy()[:n]
    img[pos[:, 0], pos[:, 1], :] = int(s["aug_sp_val"][0])
    top, left, h, w, val = [int(v) for v in s["aug_erase"]]
    out = s["aug_lut"].numpy()[img]
    if h > 0:
        out[top:top + h, left:left + w, :] = val
    return out


def test_level2_records_the_same_draws_and_reproduces_the_cpu_chain(tmp_path, monkeypatch):
    """UDA_CLR_DEVICE_INPUT=2: the workers make the same random draws in the same order but only record them; applying the
    records (salt/pepper scatter -> gamma table -> erased box) to the uint8 image gives the CPU chain's image byte for byte."""
    write_dataset(str(tmp_path), "refuge", "train", 4, size=96, seed=3)
    ds = DL.FundusSegmentation(base_dir=str(tmp_path), dataset="refuge", split="train", transform=_photometric_chain())
    fired = {"sp": 0, "lut": 0, "erase": 0}
    for seed in range(6):
        for idx in range(len(ds)):
            random.seed(100 * seed + idx); np.random.seed(100 * seed + idx)
            monkeypatch.setattr(tr, "DEVICE_TAIL", 0)
            want = ds[idx]
            random.seed(100 * seed + idx); np.random.seed(100 * seed + idx)
            monkeypatch.setattr(tr, "DEVICE_TAIL", 2)
            got = ds[idx]
            monkeypatch.setattr(tr, "DEVICE_TAIL", 0)
            assert {"image_u8", "label_u8", "aug_elastic", "aug_sp_pos", "aug_sp_n", "aug_sp_val", "aug_lut", "aug_erase"} <= set(got)
            ref = tr.ToTensor()(tr.Normalize_tf()({"image": numpy_apply_recorded(got), "label": got["label_u8"].numpy(), "img_name": ""}))
            for k in ("image", "map", "boundary"):
                assert torch.equal(ref[k], want[k]), (seed, idx, k)
            fired["sp"] += int(got["aug_sp_n"][0]) > 0
            fired["lut"] += not torch.equal(got["aug_lut"], torch.arange(256, dtype=torch.uint8))
            fired["erase"] += int(got["aug_erase"][2]) > 0
            # the random state after the sample is the same in both modes: identical consumption of both generators
    assert all(v > 0 for v in fired.values()), fired
    # a batch collates (fixed-size records)
    monkeypatch.setattr(tr, "DEVICE_TAIL", 2)
    full = Compose([tr.RandomScaleCrop(96), tr.RandomRotate(), tr.RandomFlip(), tr.elastic_transform(), tr.add_salt_pepper_noise(),
                    tr.adjust_light(), tr.eraser(), tr.Normalize_tf(), tr.ToTensor()])
    ds2 = DL.FundusSegmentation(base_dir=str(tmp_path), dataset="refuge", split="train", transform=full)
    b = next(iter(DataLoader(ds2, batch_size=4, shuffle=False, num_workers=0)))
    assert b["image_u8"].shape == (4, 96, 96, 3) and b["aug_lut"].shape == (4, 256) and b["aug_sp_pos"].shape[0] == 4
    assert b["aug_elastic"].shape == (4, 1) and b["aug_erase"].shape == (4, 5)


def test_dropin_publishes_reference_import_names():
    import sys
    install()
    from networks.deeplabv3 import DeepLab                                   # noqa: F401
    from networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator   # noqa: F401
    from train_process import Trainer, Trainer_baseline, Trainer_prototype_full  # noqa: F401
    from dataloaders import fundus_dataloader, custom_transforms               # noqa: F401
    from utils.Utils import gen_prototype, gen_prototype_retrify              # noqa: F401
    from utils.metrics import dice_coeff_2label                               # noqa: F401
    import mypath                                                              # noqa: F401
    assert sys.modules["networks.deeplabv3"].__name__ == "uda_clr_amd.networks.deeplabv3"
    for k in [k for k in sys.modules if k.split(".")[0] in ("networks", "train_process", "dataloaders", "mypath") or k in ("utils", "utils.Utils", "utils.metrics")]:
        del sys.modules[k]
