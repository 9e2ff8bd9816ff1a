"""-m gpu: the device-side tail of the input pipeline inside the Trainer.

A Trainer_baseline epoch over uint8 batches (the deferred tail of dataloaders.custom_transforms with
UDA_CLR_DEVICE_INPUT=1: uda_normalize_tf decodes on the GPU) must write the same loss rows as the same epoch over the
float batches the CPU chain produces (numpy + scipy.ndimage, the reference's arithmetic): the decoded tensors are
bit-identical, so the rows are.  The kernels themselves are compared with scipy in tests/kernel_cases.py."""
import numpy as np
import pytest
import torch

import model_cases
from kernel_cases import _fundus_like, gen
from uda_clr_amd import ops
from uda_clr_amd.dataloaders import custom_transforms as tr
from uda_clr_amd.train_process import Trainer_baseline

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _batches(n, B, S):
    out_u8, out_f = [], []
    for i in range(n):
        img, lab = _fundus_like(B, S, S, gen(100 + i))
        out_u8.append({"image_u8": torch.from_numpy(img), "label_u8": torch.from_numpy(lab), "img_name": ["s"] * B})
        dec = [tr.ToTensor()(tr.Normalize_tf()({"image": img[b], "label": lab[b], "img_name": "s"})) for b in range(B)]
        out_f.append({k: torch.stack([d[k] for d in dec]) for k in ("image", "map", "boundary")})
    return out_u8, out_f


def _rows(path):
    with open(path) as f:
        return [l.split(",") for l in f.read().strip().split("\n")[1:]]


def test_trainer_epoch_on_uint8_batches_equals_epoch_on_cpu_decoded_batches(tmp_path):
    u8, fl = _batches(2, 4, 64)
    rows = []
    for tag, loader in (("u8", u8), ("f32", fl)):
        m = model_cases.seeded_model().to(DEV)
        m._engine_for(torch.empty(1, device=DEV)).seed = 1337            # same dropout streams in both runs
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.99))
        t = Trainer_baseline.Trainer(cuda=True, model_gen=m, optimizer_gen=opt, lr_gen=1e-3, lr_decrease_rate=0.1,
                                     val_loader=loader, domain_loaderS=loader, domain_loaderT=loader, out=str(tmp_path / tag),
                                     max_epoch=1, stop_epoch=1, interval_validate=1, batch_size=4, warmup_epoch=-1)
        t.epoch = 0
        t.iteration = 0
        t.train()
        t.validate()
        rows.append(_rows(tmp_path / tag / "log.csv"))
    a, b = rows
    assert len(a) == len(b) >= 3
    for ra, rb in zip(a, b):
        assert ra[2:8] == rb[2:8], (ra, rb)          # loss columns and the validation tuple, digit for digit


def test_elastic_deform_is_identity_where_not_applied_and_moves_pixels_where_applied():
    img, lab = _fundus_like(4, 128, 128, gen(5))
    iu, lu = torch.from_numpy(img).to(DEV), torch.from_numpy(lab).to(DEV)
    apply = torch.tensor([1, 0, 1, 0], dtype=torch.uint8, device=DEV)
    io, lo = ops.elastic_deform(iu, lu, apply=apply)
    assert torch.equal(io[1], iu[1]) and torch.equal(lo[3], lu[3])
    assert not torch.equal(io[0], iu[0]) and not torch.equal(lo[2], lu[2])
    assert set(torch.unique(lo).tolist()) <= set(range(256))
    # labels stay a valid grey coding up to bilinear blends along the class borders: most pixels keep a pure class value
    pure = torch.isin(lo[0], torch.tensor([0, 128, 255], dtype=torch.uint8, device=DEV)).float().mean().item()
    assert pure > 0.9


def test_level2_decode_on_the_device_equals_the_cpu_chain(tmp_path, monkeypatch):
    """UDA_CLR_DEVICE_INPUT=2 through TrainerBase._decode on the HIP kernels: with the recorded photometric outcomes (elastic not
    fired) the decoded batch equals the CPU chain's tensors exactly; with the elastic transform fired on some samples the others
    still do and everything stays in range."""
    import random
    from torch.utils.data import DataLoader
    from test_dataloaders_cpu import _photometric_chain, Compose
    from uda_clr_amd.dataloaders import fundus_dataloader as DL
    from uda_clr_amd.dataloaders.synthetic import write_dataset
    from uda_clr_amd.train_process._common import HipOps, TrainerBase

    class T(TrainerBase):
        def __init__(self):
            self.ops = HipOps()
        def _to(self, t): return t.to(DEV)

    write_dataset(str(tmp_path), "refuge", "train", 4, size=128, seed=4)
    ds = DL.FundusSegmentation(base_dir=str(tmp_path), dataset="refuge", split="train", transform=_photometric_chain())
    want, got = [], []
    for idx in range(4):
        for mode, dst in ((0, want), (2, got)):
            random.seed(7 + idx); np.random.seed(7 + idx)
            monkeypatch.setattr(tr, "DEVICE_TAIL", mode)
            dst.append(ds[idx])
    monkeypatch.setattr(tr, "DEVICE_TAIL", 0)
    batch = {k: torch.stack([g[k] for g in got]) for k in got[0] if k != "img_name"}
    dec = T()._decode(batch)
    for k in ("image", "map", "boundary"):
        assert torch.equal(dec[k].cpu(), torch.stack([w[k] for w in want])), k
    # elastic fired on samples 0 and 2
    batch["aug_elastic"] = torch.tensor([[1], [0], [1], [0]], dtype=torch.uint8)
    dec2 = T()._decode(batch)
    for b in (1, 3):
        assert torch.equal(dec2["image"][b], dec["image"][b]) and torch.equal(dec2["boundary"][b], dec["boundary"][b])
    assert not torch.equal(dec2["image"][0], dec["image"][0])
    assert float(dec2["image"].min()) >= -1.0 and float(dec2["image"].max()) <= 1.0
    assert bool((dec2["map"][:, 0] <= dec2["map"][:, 1]).all())


def test_utils_postprocessing_dropin_matches_scipy_and_the_reference_return_types():
    """utils.Utils.postprocessing (Utils.py:438-463): one [2,H,W] probability map -> numpy masks; uint8 for the default branch,
    float (a copy of the probabilities overwritten by the masks) for dataset names starting with 'D'."""
    from kernel_cases import _scipy_postprocess
    from uda_clr_amd.utils import Utils
    rs = np.random.RandomState(3)
    yy, xx = np.meshgrid(np.arange(160), np.arange(160), indexing="ij")
    prob = np.stack([np.clip(1.3 - np.sqrt(((yy - 80) / r) ** 2 + ((xx - 70) / (r * 1.2)) ** 2), 0, 1) for r in (30.0, 55.0)]).astype(np.float32)
    prob = np.clip(prob + 0.3 * (rs.rand(2, 160, 160) < 0.02), 0, 1).astype(np.float32)
    out = Utils.postprocessing(torch.from_numpy(prob), threshold=0.75, dataset='G')
    assert out.dtype == np.uint8 and out.shape == (2, 160, 160)
    assert np.array_equal(out, _scipy_postprocess(prob, 0.75, 0.75))
    outd = Utils.postprocessing(torch.from_numpy(prob), dataset='Drishti-GS')
    assert outd.dtype == np.float32 and np.array_equal(outd.astype(np.uint8), _scipy_postprocess(prob, 0.1, 0.5))
    batch = Utils.postprocessing_batch(torch.from_numpy(np.stack([prob, prob[:, ::-1].copy()])).to(DEV))
    assert batch.shape == (2, 2, 160, 160) and np.array_equal(batch[0].cpu().numpy(), out)
    assert np.array_equal(batch[1].cpu().numpy(), _scipy_postprocess(prob[:, ::-1].copy(), 0.75, 0.75))
