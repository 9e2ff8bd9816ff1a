"""Source-only trainer - drop-in for the reference's ``train_process/Trainer_baseline.py``
(same constructor signature, attributes ``epoch`` / ``iteration``, ``train()``, ``log.csv`` columns
and checkpoint keys).  The step is

    zero_grad -> generator forward on the source batch -> BCE+MSE seg loss -> backward -> Adam

(Trainer_baseline.py:198-243) with the generator as one fused autograd node on the HIP kernels and
the loss as one fused kernel pair.  Data parallel when launched one process per GPU: the loaders are
re-sharded by rank, generator gradients are averaged with one flat RCCL all-reduce, rank 0 logs,
validates and checkpoints.
"""
import os.path as osp
import timeit

import torch

from ..optim import take_over
from ..parallel import FlatGradAllReduce
from ._common import (HipOps, TrainerBase, decorrelate_dropout, get_lr, nan_guard, prefer_fused, progress, shard_loader,
                      sync_replicas, trange)


class Trainer(TrainerBase):
    log_headers = ['epoch', 'iteration', 'train/loss_seg', 'train/cup_dice', 'train/disc_dice',
                   'valid/loss_CE', 'valid/cup_dice', 'valid/disc_dice', 'elapsed_time']

    def __init__(self, cuda, model_gen, optimizer_gen, val_loader, domain_loaderS,
                 domain_loaderT, out, max_epoch, stop_epoch=None,
                 lr_gen=1e-3, lr_decrease_rate=0.1, interval_validate=None, batch_size=8, warmup_epoch=10):
        self.cuda = cuda
        self.warmup_epoch = warmup_epoch
        self.model_gen = model_gen
        self.optim_gen = prefer_fused(take_over(optimizer_gen))      # Adam on the flat multi-tensor kernel (uda_clr_amd.optim)
        self.lr_gen = lr_gen
        self.lr_decrease_rate = lr_decrease_rate
        self.batch_size = batch_size
        self.interval_validate = 10 if interval_validate is None else interval_validate
        self._setup_io(out)
        self.val_loader = val_loader
        self.domain_loaderS = shard_loader(domain_loaderS, self.rank, self.world)
        self.domain_loaderT = shard_loader(domain_loaderT, self.rank, self.world)
        self.ops = HipOps()
        self._reducer = FlatGradAllReduce(list(model_gen.parameters())) if self.world > 1 else None
        sync_replicas((model_gen,), self.rank, self.world)
        decorrelate_dropout(model_gen, self.rank, self.world)
        self.epoch = 0
        self.iteration = 0
        self.max_epoch = max_epoch
        self.stop_epoch = stop_epoch if stop_epoch is not None else max_epoch
        self.best_disc_dice = 0.0
        self.running_loss_tr = 0.0
        self.best_mean_dice = 0.0
        self.best_epoch = -1

    # ------------------------------------------------------------------ validation / checkpoints
    def _checkpoint(self, epoch_tag):
        torch.save({
            'epoch': self.epoch,
            'iteration': self.iteration,
            'arch': self.model_gen.__class__.__name__,
            'optim_state_dict': self.optim_gen.state_dict(),
            'model_state_dict': self.model_gen.state_dict(),
            'learning_rate_gen': get_lr(self.optim_gen),
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

    # ------------------------------------------------------------------ one epoch
    def train_epoch(self):
        self.model_gen.train()
        self.running_seg_loss = 0.0
        start_time = timeit.default_timer()
        nS = len(self.domain_loaderS)
        for batch_idx, sampleS in progress(enumerate(self.domain_loaderS), total=nS,
                                           desc='Train epoch=%d' % self.epoch, ncols=80, leave=False):
            self.iteration = batch_idx + self.epoch * nS
            assert self.model_gen.training
            self.optim_gen.zero_grad()
            sampleS = self._decode(sampleS)
            imageS = self._to(sampleS['image'])
            target_map = self._to(sampleS['map'])
            target_boundary = self._to(sampleS['boundary'])
            oS, boundaryS = self.model_gen(imageS)[:2]
            loss_seg = self.ops.seg_loss(oS, boundaryS, target_map, target_boundary)
            loss_seg.backward()
            if self._reducer is not None:
                self._reducer.all_reduce_mean()
            self.optim_gen.step()
            if hasattr(self.model_gen, 'note_params_changed'):
                self.model_gen.note_params_changed()       # activations kept for the MC passes are stale now
            loss_seg_data = self._fetch([loss_seg])[0]      # the step's single host sync (loss + non-finite flag)
            self.running_seg_loss += loss_seg_data
            self.writer.add_scalar('train_gen/loss_seg', loss_seg_data, self.iteration)
            self._log_row([self.epoch, self.iteration, loss_seg_data] + [''] * 5 + [self.elapsed()])
        self.running_seg_loss /= max(nS, 1)
        if self.rank == 0:
            print('\n[Epoch: %d] lr:%f,  Average segLoss: %f, Execution time: %.5f' %
                  (self.epoch, get_lr(self.optim_gen), self.running_seg_loss, timeit.default_timer() - start_time))

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
