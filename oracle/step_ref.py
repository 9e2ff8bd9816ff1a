"""CPU restatement of one training iteration (TEST INFRASTRUCTURE).

Reference anchors
  train_process/Trainer_baseline.py:198-243          source-only step
  train_process/Trainer_prototype_full.py:261-517    prototype_full step
  train_use_fix_initial.py:210-226                   optimiser settings

``model`` is any callable returning the reference 7-tuple (oracle.deeplab_ref.OracleDeepLab
on CPU).  The functions return plain floats in the column order of the reference's
``log.csv`` rows (Trainer_prototype_full.py:578-592).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import proto_ref

SMOOTH = 1e-7


def seg_loss(oS, boundaryS, target_map, target_boundary):
    """BCELoss(sigmoid(oS), map) + MSELoss(sigmoid(boundaryS), boundary), both 'mean'
    (Trainer_prototype_full.py:18-19, 292-294; quirk Q3: log clamped at -100)."""
    return (F.binary_cross_entropy(torch.sigmoid(oS), target_map)
            + F.mse_loss(torch.sigmoid(boundaryS), target_boundary))


def make_optimizers(model_gen, model_dis=None, model_dis2=None, lr_gen=1e-3, lr_dis=2.5e-5,
                    momentum=0.99, weight_decay=5e-4):
    """train_use_fix_initial.py:210-226."""
    og = torch.optim.Adam(model_gen.parameters(), lr=lr_gen, betas=(0.9, 0.99))
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=lr_dis, momentum=momentum,
                                   weight_decay=weight_decay)
    return og, (mk(model_dis) if model_dis is not None else None), \
        (mk(model_dis2) if model_dis2 is not None else None)


def baseline_step(model, optim, imageS, target_map, target_boundary):
    """Trainer_baseline.py:198-243."""
    optim.zero_grad()
    oS, bS = model(imageS)[:2]
    loss = seg_loss(oS, bS, target_map, target_boundary)
    val = loss.item()
    if math.isnan(val):
        raise ValueError("loss is nan while training")
    loss.backward()
    optim.step()
    return val


def _uncertainty(o):
    s = torch.sigmoid(o)
    return -1.0 * s * torch.log(s + SMOOTH)         # Trainer_prototype_full.py:452


def _adv(d, label):
    return F.binary_cross_entropy_with_logits(d, torch.full_like(d, float(label)))


class PrototypeFullStep:
    """State + one iteration of Trainer_prototype_full.train_epoch (:261-517) with
    ``use_global=True``.  ``mc_model`` lets a test run the 4 no-grad stochastic passes."""

    def __init__(self, model_gen, model_dis, model_dis2, optim_gen, optim_dis, optim_dis2,
                 use_pid=True, retrify_pesudo=True, global_pro_weight=0.9, pro_weight=0.1):
        self.g, self.d, self.d2 = model_gen, model_dis, model_dis2
        self.og, self.od, self.od2 = optim_gen, optim_dis, optim_dis2
        self.use_pid, self.retrify = use_pid, retrify_pesudo
        self.pro_weight = pro_weight
        self.bank = proto_ref.PrototypeBank(global_pro_weight)

    def _set_grad(self, gen, dis):
        for p in self.g.parameters():
            p.requires_grad = gen
        for m in (self.d, self.d2):
            for p in m.parameters():
                p.requires_grad = dis

    def __call__(self, imageS, target_map, target_boundary, imageT, prototypes_on=True):
        self.og.zero_grad(); self.od.zero_grad(); self.od2.zero_grad()
        self._set_grad(True, False)                                           # :266-271
        oT, bT, _, _, xt_feature, oT_before, _ = self.g(imageT)               # :287
        oS, bS, _, _, xs_feature, oS_before, _ = self.g(imageS)               # :288
        loss_seg = seg_loss(oS, bS, target_map, target_boundary)              # :292-294
        row = {"seg": loss_seg.item()}
        intra = None
        if self.use_pid and prototypes_on:                                    # :328-449
            pred_oS = F.interpolate(target_map.clone(), size=oS_before.shape[2:], mode="nearest")
            src = self.bank.update("src", proto_ref.gen_prototype(pred_oS, xs_feature))
            T = 8
            rep = imageT.repeat(2, 1, 1, 1)
            stride = rep.shape[0] // 2
            preds = torch.zeros([stride * T, 2, imageT.shape[2], imageT.shape[3]])
            for i in range(T // 2):                                           # :364-368
                with torch.no_grad():
                    preds[2 * stride * i:2 * stride * (i + 1)] = self.g(rep)[0]
            if self.retrify:
                cur = proto_ref.gen_prototype_retrify(oT_before, xt_feature, preds, T, stride)[:4]
            else:
                cur = proto_ref.gen_prototype(torch.sigmoid(oT_before), xt_feature)
            tgt = self.bank.update("tgt", cur)
            intra, inter = proto_ref.alignment_losses(src, tgt)
            row["intra"], row["inter"] = intra.item(), inter.item()
        d2_out = self.d(torch.sigmoid(bT))                                    # :452-458
        d1_out = self.d2(_uncertainty(oT))
        loss_adv = 0.01 * (_adv(d1_out, 1) + _adv(d2_out, 1))
        row["adv"] = loss_adv.item()
        loss_all = loss_seg + loss_adv
        if intra is not None:
            loss_all = loss_all + self.pro_weight * intra                     # :465
        loss_all.backward()
        self.og.step()
        self._set_grad(False, True)                                           # :472-477
        oS, bS, oT, bT = oS.detach(), bS.detach(), oT.detach(), bT.detach()
        loss_same = _adv(self.d2(_uncertainty(oS)), 1) + _adv(self.d(torch.sigmoid(bS)), 1)
        row["D_same"] = loss_same.item()
        loss_same.backward()                                                  # :495
        loss_diff = _adv(self.d2(_uncertainty(oT)), 0) + _adv(self.d(torch.sigmoid(bT)), 0)
        row["D_diff"] = loss_diff.item()
        loss_diff.backward()                                                  # :513
        self.od.step(); self.od2.step()                                       # :516-517
        for v in row.values():
            if math.isnan(v):
                raise ValueError("loss is nan while training")
        return row
