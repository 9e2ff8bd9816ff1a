"""Prototype-alignment trainer - drop-in for the reference's
``train_process/Trainer_prototype_full.py`` AND for the newer call site in
``train_use_fix_initial.py:276-304`` (the shipped class and its caller disagree; this constructor
accepts the union of both keyword sets, ``use_global`` defaulting to True - SURVEY.md 0.2).

One iteration (Trainer_prototype_full.py:261-517):
  1. generator on target then source batch (training-mode BN, quirk Q2)
  2. seg loss on source                                                      [fused HIP kernels]
  3. if use_pid and epoch > warmup_epoch:
       source prototypes from the (nearest-resized) labels, EMA             [one fused reduction]
       4 no-grad stochastic passes on the doubled target batch (T = 8)
       retrified target prototypes (or soft ones), EMA                      [fused HIP kernels]
       EMA + intra = sum_k MSE(src_k, tgt_k) + inter (logged only)            [one fused kernel, one for its gradients]
  4. adversarial term through the two patch discriminators (native kernels; the sigmoid / uncertainty maps are formed
     inside their first layer, the two BCE-with-logits terms are one kernel)
  5. generator backward + Adam; discriminator steps on detached outputs (SGD)

Deviations from the shipped file, all documented in DESIGN.md:
  - with use_pid and epoch <= warmup_epoch the shipped code reaches ``loss_all.backward()`` with
    ``loss_all`` unbound (:463-468); here that phase trains ``loss_seg + loss_adv_diff``.
  - loss values are fetched with ONE host sync per iteration instead of >= 6 ``.item()`` calls.
  - data parallel (one process per GPU): generator / discriminator gradients are averaged over ranks,
    prototype sums are all-reduced before the division so every rank holds the global centroids, and
    the replicated alignment loss is weighted by the world size before the averaged backward so the
    update equals the single-process update on the global batch.
"""
import os.path as osp
import timeit

import numpy as np
import torch
import torch.nn.functional as F

from ..optim import take_over
from ..parallel import FlatGradAllReduce
from ._common import (HipOps, TrainerBase, decorrelate_dropout, get_lr, nan_guard, prefer_fused, progress, shard_loader,
                      sync_replicas, trange)

mseloss = torch.nn.MSELoss()


class Trainer(TrainerBase):
    log_headers = ['epoch', 'iteration', 'train/loss_seg', 'train/cup_dice', 'train/disc_dice', 'train/loss_adv',
                   'train/loss_D_same', 'train/loss_D_diff', 'valid/loss_CE', 'valid/cup_dice', 'valid/disc_dice',
                   'elapsed_time']

    def __init__(self, cuda, model_gen, model_dis, model_uncertainty_dis, optimizer_gen, optimizer_dis,
                 optimizer_uncertainty_dis, val_loader, domain_loaderS, domain_loaderT, out, max_epoch,
                 use_global=True, use_pid=False, retrify_pesudo=False, global_pro_weight=0.9, pro_weight=0.1,
                 stop_epoch=None, lr_gen=1e-3, lr_dis=1e-3, lr_decrease_rate=0.1, interval_validate=None,
                 batch_size=8, warmup_epoch=25, target_name='Drishti-GS',
                 model_geninitial_pesudolabel=None, use_fix_initial=False, use_TN=False,
                 use_trg_cons=False, src_reg=False, aug_weight=1.0, src_reg_weight=1.0):
        self.First_src = True
        self.First = True
        self.target_name = target_name
        self.use_global = use_global
        self.use_pid = use_pid
        self.retrify_pesudo = retrify_pesudo
        self.global_pro_weight = global_pro_weight
        self.pro_weight = pro_weight
        self.cuda = cuda
        self.warmup_epoch = warmup_epoch
        self.model_gen = model_gen
        self.model_geninitial_pesudolabel = model_geninitial_pesudolabel     # accepted, unused (as shipped)
        self.use_fix_initial = use_fix_initial
        self.use_TN = use_TN
        # the two losses of SURVEY.md Appendix B (no shipped source, parity unpinned); off = shipped behaviour
        self.use_trg_cons, self.src_reg = use_trg_cons, src_reg
        self.aug_weight, self.src_reg_weight = aug_weight, src_reg_weight
        self.model_dis2 = model_uncertainty_dis
        self.model_dis = model_dis
        self.optim_gen = prefer_fused(take_over(optimizer_gen))      # Adam on the flat multi-tensor kernel (uda_clr_amd.optim)
        self.optim_dis = prefer_fused(optimizer_dis)
        self.optim_dis2 = prefer_fused(optimizer_uncertainty_dis)
        self.lr_gen = lr_gen
        self.lr_dis = lr_dis
        self.lr_decrease_rate = lr_decrease_rate
        self.batch_size = batch_size
        self.interval_validate = 10 if interval_validate is None else interval_validate
        if use_pid and not use_global:
            raise NotImplementedError("use_pid without use_global leaves the centroids undefined in the "
                                      "reference (Trainer_prototype_full.py:335-355, :428)")
        self._setup_io(out)
        self.val_loader = val_loader
        self.domain_loaderS = shard_loader(domain_loaderS, self.rank, self.world)
        self.domain_loaderT = shard_loader(domain_loaderT, self.rank, self.world)
        self.ops = HipOps()
        self._reducers = None
        if self.world > 1:
            self._reducers = [FlatGradAllReduce(list(m.parameters())) for m in (model_gen, model_dis, model_uncertainty_dis)]
            sync_replicas((model_gen, model_dis, model_uncertainty_dis), self.rank, self.world)
            decorrelate_dropout(model_gen, self.rank, self.world)
        self.epoch = 0
        self.iteration = 0
        self.max_epoch = max_epoch
        self.stop_epoch = stop_epoch if stop_epoch is not None else max_epoch
        self.best_disc_dice = 0.0
        self.running_loss_tr = 0.0
        self.running_adv_diff_loss = 0.0
        self.running_adv_same_loss = 0.0
        self.best_mean_dice = 0.0
        self.best_epoch = -1
        self.src_centroids = None      # detached EMA state, (cup_obj, disc_obj, cup_bck, disc_bck)
        self.tgt_centroids = None

    # ------------------------------------------------------------------ helpers
    def update_pro(self, centroid_0_obj, global_centroid_0_obj, name='moving_average'):
        if name == 'moving_average':
            w = self.global_pro_weight
            global_centroid_0_obj = global_centroid_0_obj * (1 - w) + w * centroid_0_obj
        return global_centroid_0_obj

    def _ema(self, stored, current):
        """First use stores ``current``; later (1-decay)*stored.detach() + decay*current - the gradient
        flows only through the current term (quirk Q4, :335-355, :378-398)."""
        if stored is None:
            new = tuple(current)
        else:
            d = self.global_pro_weight
            new = tuple((1 - d) * s + d * c for s, c in zip(stored, current))
        return new, tuple(t.detach() for t in new)

    @staticmethod
    def _grad_mode(models, mode):
        for m in models:                      # the native discriminators' switch (networks/GAN.py); stock modules have none
            if hasattr(m, "grad_mode"):
                m.grad_mode = mode

    @staticmethod
    def _set_requires_grad(models, flag):
        for m in models:
            for p in m.parameters():
                p.requires_grad = flag

    @staticmethod
    def _uncertainty(o, smooth=1e-7):
        s = torch.sigmoid(o)
        return -1.0 * s * torch.log(s + smooth)

    @staticmethod
    def _adv(d_out, label):
        return F.binary_cross_entropy_with_logits(d_out, torch.full_like(d_out, float(label)))

    def _disc(self, d, logits, pre):
        """The discriminator on sigmoid(logits) (boundary branch) or on the uncertainty map of the logits (:452-454).  The native
        discriminators form the map inside their first layer (networks/GAN.py ``forward(x, pre=...)``); any other module gets
        the reference's elementwise expressions."""
        if getattr(d, "fused_pre", False):
            return d(logits, pre=pre)
        return d(torch.sigmoid(logits) if pre == "sigmoid" else self._uncertainty(logits))

    def _adv_pair(self, d1, d2, label, scale):
        """scale * (BCEWithLogits(d1, label) + BCEWithLogits(d2, label)) (:456-458, :479-513)"""
        if hasattr(self.ops, "adv_loss"):
            return self.ops.adv_loss(d1, d2, label, scale)
        s = self._adv(d1, label) + self._adv(d2, label)
        return s if scale == 1.0 else scale * s

    def _prototypes_on(self):
        return self.use_pid and self.epoch > self.warmup_epoch

    # ------------------------------------------------------------------ validation / checkpoints
    def _checkpoint(self, epoch_tag):
        torch.save({
            'epoch': self.epoch,
            'iteration': self.iteration,
            'arch': self.model_gen.__class__.__name__,
            'optim_state_dict': self.optim_gen.state_dict(),
            'optim_dis_state_dict': self.optim_dis.state_dict(),
            'optim_dis2_state_dict': self.optim_dis2.state_dict(),
            'model_state_dict': self.model_gen.state_dict(),
            'model_dis_state_dict': self.model_dis.state_dict(),
            'model_dis2_state_dict': self.model_dis2.state_dict(),
            'learning_rate_gen': get_lr(self.optim_gen),
            'learning_rate_dis': get_lr(self.optim_dis),
            'learning_rate_dis2': get_lr(self.optim_dis2),
            'best_mean_dice': self.best_mean_dice,
        }, osp.join(self.out, 'checkpoint_%d.pth.tar' % epoch_tag))

    def validate(self):
        if self.rank != 0:
            return
        training = self.model_gen.training
        self.model_gen.eval()
        val_loss, cup, disc, pa_c, pa_d, iou_c, iou_d = self._validate_core()
        n = self.epoch * len(self.domain_loaderS)
        for tag, v in (('val_data/val_CUP_PA', pa_c), ('val_data/val_DISC_PA', pa_d), ('val_data/val_CUP_IOU', iou_c),
                       ('val_data/val_DISC_IOU', iou_d), ('val_data/loss_CE', val_loss), ('val_data/val_CUP_dice', cup),
                       ('val_data/val_DISC_dice', disc)):
            self.writer.add_scalar(tag, v, n)
        mean_dice = cup + disc
        if mean_dice > self.best_mean_dice:
            self.best_epoch = self.epoch + 1
            self.best_mean_dice = mean_dice
            self._checkpoint(self.best_epoch)
        elif (self.epoch + 1) % 50 == 0:
            self._checkpoint(self.epoch + 1)
        self._log_row([self.epoch, self.iteration] + [''] * 5 + [(val_loss, cup, disc)] + [self.elapsed()] +
                      ['best model epoch: %d' % self.best_epoch])
        self.writer.add_scalar('best_model_epoch', self.best_epoch, n)
        self.last_val = (val_loss, cup, disc)
        if training:
            self.model_gen.train()
            self.model_dis.train()
            self.model_dis2.train()

    # ------------------------------------------------------------------ one iteration
    def train_step(self, sampleS, sampleT):
        """Returns the log row values (seg, adv, D_same, D_diff[, intra, inter]) as floats."""
        ops = self.ops
        gen, dis, dis2 = self.model_gen, self.model_dis, self.model_dis2
        self.optim_gen.zero_grad()
        self.optim_dis.zero_grad()
        self.optim_dis2.zero_grad()
        # :266-271 freezes the discriminators for the generator step and re-runs them on the detached target
        # outputs afterwards (:471-517).  Their weights do not change in between, so the target-side forward is
        # run ONCE with a graph: the generator step back-propagates through it into the generator only
        # (backward(inputs=...) prunes the discriminator weight gradients), the discriminator step re-uses its
        # outputs with label 0 and back-propagates into the discriminator weights only.  Same values, two
        # discriminator forwards per iteration fewer.
        self._set_requires_grad((dis, dis2), True)
        self._set_requires_grad((gen,), True)
        gen_params = [q for q in gen.parameters() if q.requires_grad]
        dis_params = [q for m in (dis, dis2) for q in m.parameters()]
        sampleS, sampleT = self._decode(sampleS), self._decode(sampleT)
        imageS, target_map = self._to(sampleS['image']), self._to(sampleS['map'])
        target_boundary = self._to(sampleS['boundary'])
        imageT = self._to(sampleT['image'])
        # the generator's parameters do not change until optim_gen.step(), the discriminators' until optim_dis.step() at the end of
        # the step: one set of kernel-side weight layouts per module and step instead of one per pass
        import contextlib
        with contextlib.ExitStack() as stack:
            for m in (gen, dis, dis2):
                if hasattr(m, "shared_weight_layouts"):
                    stack.enter_context(m.shared_weight_layouts())
            return self._train_step_body(ops, gen, dis, dis2, gen_params, dis_params, imageS, imageT, target_map, target_boundary)

    def _train_step_body(self, ops, gen, dis, dis2, gen_params, dis_params, imageS, imageT, target_map, target_boundary):
        oT, boundaryT, _, _, xt_feature, oT_before, _ = gen(imageT)                          # :287
        oS, boundaryS, _, _, xs_feature, oS_before, _ = gen(imageS)                          # :288
        loss_seg = ops.seg_loss(oS, boundaryS, target_map, target_boundary)                  # :292-294
        scalars = [loss_seg.detach()]
        intra_loss = None
        if self._prototypes_on():                                                            # :328-449
            cur_src = ops.gen_prototype_from_labels(target_map, xs_feature)                  # :330-334
            prev_src = self.src_centroids
            T = 8
            volume_batch_r = imageT.repeat(2, 1, 1, 1)
            stride = volume_batch_r.shape[0] // 2
            if hasattr(gen, "mc_dropout_logits"):
                # fused form of :358-368: the deterministic pre-dropout part of the T forward above is
                # reused, only the dropout-dependent decoder tail runs T/2 times on the doubled batch
                preds_trg = gen.mc_dropout_logits(imageT, passes=T // 2, reps=2)
            else:
                preds_trg = torch.empty([stride * T, 2, imageT.shape[2], imageT.shape[3]], device=imageT.device)
                with torch.no_grad():                                                        # :364-368 (features_trg is dead, Q5)
                    for i in range(T // 2):
                        preds_trg[2 * stride * i:2 * stride * (i + 1)] = gen(volume_batch_r)[0]
            if self.retrify_pesudo:
                res = ops.gen_prototype_retrify(oT_before, xt_feature, preds_trg, None, T, stride)
                cur_tgt = getattr(res, "centroids", None) or res[:4]
                self.target_std_map, self.mask_0, self.mask_1 = res[4:]
            else:
                cur_tgt = ops.gen_prototype(torch.sigmoid(oT_before), xt_feature)            # :375-377
            if hasattr(ops, "proto_align"):
                # :335-355, :378-398, :428-444 as one launch (+ one for both gradients): EMA of the eight centroids with the
                # stored (detached) state, intra / inter; the returned centroids are the detached EMA state of the next step
                intra_loss, inter_loss, self.src_centroids, self.tgt_centroids = ops.proto_align(
                    cur_src, cur_tgt, prev_src, self.tgt_centroids, self.global_pro_weight)
                src = self.src_centroids
            else:
                src, self.src_centroids = self._ema(prev_src, cur_src)
                tgt, self.tgt_centroids = self._ema(self.tgt_centroids, cur_tgt)
                intra_loss = sum(mseloss(s, t) for s, t in zip(src, tgt))                    # :428-441
                inter_loss = mseloss(src[1], src[3]) + mseloss(src[0], src[2])               # :443-444 (logged only)
            self.First_src = self.First = False
            if self.src_reg:                                                                 # Appendix B (unpinned)
                pred_oS = F.interpolate(target_map, size=xs_feature.shape[2:], mode='nearest')
                self.loss_src_reg = ops.discriminative_loss(xs_feature, src, pred_oS)
        D_out2 = self._disc(dis, boundaryT, "sigmoid")                                       # :452-454
        D_out1 = self._disc(dis2, oT, "entropy")
        loss_adv_diff = self._adv_pair(D_out1, D_out2, 1, 0.01)                              # :456-458
        scalars.append(loss_adv_diff.detach())
        loss_all = loss_seg + loss_adv_diff
        if intra_loss is not None:
            loss_all = loss_all + (self.pro_weight * self.world) * intra_loss                # :465 (x world: see module doc)
            if self.src_reg:
                loss_all = loss_all + self.src_reg_weight * self.loss_src_reg
        self._grad_mode((dis, dis2), "input")
        import contextlib
        with (gen.fused_grad_accumulation() if hasattr(gen, "fused_grad_accumulation") else contextlib.nullcontext()):
            loss_all.backward(inputs=gen_params, retain_graph=True)       # (T and S passes: the second adds into .grad in one launch)
        self._grad_mode((dis, dis2), "auto")
        if self.use_trg_cons and intra_loss is not None and self.retrify_pesudo:             # Appendix B (unpinned)
            # augmented consistency: pseudo labels of the clean target prediction supervise the prediction on
            # a photometrically augmented copy, on the pixels the MC-dropout std marked reliable
            oT_aug = gen(ops.photometric_augment(imageT))[0]
            loss_aug = ops.consistency_loss(oT_aug, oT, self.mask_0, self.mask_1, self.epoch, self.aug_weight)
            loss_aug.backward()
            self.loss_aug = loss_aug.detach()
        # data parallel: the generator's gradient all-reduce is STARTED here and waited for after the discriminator step below - that
        # step reads only the detached generator outputs and the discriminators' own weights, so ~12 ms of kernels overlap the
        # collective; optim_gen.step() moves behind it (the discriminator step does not read the generator's parameters either)
        pending = self._reducers[0].start() if self._reducers is not None else None
        if self._reducers is None:
            self.optim_gen.step()
            if hasattr(self.model_gen, 'note_params_changed'):
                self.model_gen.note_params_changed()       # activations kept for the MC passes are stale now
        # ---- discriminators on detached generator outputs (:471-517)
        self._set_requires_grad((gen,), False)
        oS, boundaryS = oS.detach(), boundaryS.detach()
        loss_D_same = self._adv_pair(self._disc(dis2, oS, "entropy"), self._disc(dis, boundaryS, "sigmoid"), 1, 1.0)
        loss_D_same.backward()
        loss_D_diff = self._adv_pair(D_out1, D_out2, 0, 1.0)
        self._grad_mode((dis, dis2), "weights")
        loss_D_diff.backward(inputs=dis_params)
        self._grad_mode((dis, dis2), "auto")
        del D_out1, D_out2, loss_all
        if self._reducers is not None:
            self._reducers[0].finish(pending)
            self.optim_gen.step()
            if hasattr(self.model_gen, 'note_params_changed'):
                self.model_gen.note_params_changed()
        if self._reducers is not None:
            self._reducers[1].all_reduce_mean()
            self._reducers[2].all_reduce_mean()
        self.optim_dis.step()
        self.optim_dis2.step()
        scalars += [loss_D_same.detach(), loss_D_diff.detach()]
        if intra_loss is not None:
            scalars += [intra_loss.detach(), inter_loss.detach()]
        return self._fetch(scalars)                                                          # the single host sync

    # ------------------------------------------------------------------ one epoch
    def train_epoch(self):
        self.model_gen.train()
        self.model_dis.train()
        self.model_dis2.train()
        run = np.zeros(6)
        domain_t_loader = enumerate(self.domain_loaderT)
        start_time = timeit.default_timer()
        nS = len(self.domain_loaderS)
        for batch_idx, sampleS in progress(enumerate(self.domain_loaderS), total=nS,
                                           desc='Train epoch=%d' % self.epoch, ncols=80, leave=False):
            self.iteration = batch_idx + self.epoch * nS
            assert self.model_gen.training and self.model_dis.training and self.model_dis2.training
            try:
                _, sampleT = next(domain_t_loader)
            except StopIteration:
                domain_t_loader = enumerate(self.domain_loaderT)
                _, sampleT = next(domain_t_loader)
            vals = self.train_step(sampleS, sampleT)
            run[:len(vals)] += vals
            it = self.iteration
            self.writer.add_scalar('train_gen/loss_seg', vals[0], it)
            self.writer.add_scalar('train_adv/loss_adv_diff', vals[1], it)
            self.writer.add_scalar('train_dis/loss_D_same', vals[2], it)
            self.writer.add_scalar('train_dis/loss_D_diff', vals[3], it)
            if len(vals) == 6:
                self.writer.add_scalar('train_pro/loss_intra', vals[4], it)
                self.writer.add_scalar('train_pro/loss_inter', vals[5], it)
            self._log_row([self.epoch, self.iteration] + list(vals) + [''] * 5 + [self.elapsed()])   # :585-592
        run /= max(nS, 1)
        (self.running_seg_loss, self.running_adv_diff_loss, self.running_dis_same_loss, self.running_dis_diff_loss,
         self.running_intra, self.running_inter) = run.tolist()
        if self.rank == 0:
            print('\n[Epoch: %d] lr:%f,  Average segLoss: %f,  Average advLoss: %f, Average dis_same_Loss: %f, '
                  'Average dis_diff_Lyoss: %f, Average intra_Loss: %f, Average inter_Loss: %f,Execution time: %.5f' %
                  ((self.epoch, get_lr(self.optim_gen)) + tuple(run.tolist()) + (timeit.default_timer() - start_time,)))

    def train(self):
        for epoch in trange(self.epoch, self.max_epoch, desc='Train', ncols=80):
            self.epoch = epoch
            self.train_epoch()
            if self.stop_epoch == self.epoch:
                print('Stop epoch at %d' % self.stop_epoch)
                break
            self._lr_schedule(epoch)
            self.writer.add_scalar('lr_gen', get_lr(self.optim_gen), self.epoch * len(self.domain_loaderS))
            if (self.epoch + 1) % self.interval_validate == 0:
                self.validate()
        self.writer.close()
