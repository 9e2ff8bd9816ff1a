"""Shared host logic of the Trainer classes: logging, checkpoints, validation, data-parallel
bootstrap.  Nothing here touches the reference; behaviour follows the cited lines of
``train_process/Trainer_baseline.py`` / ``Trainer_prototype_full.py``."""
from __future__ import annotations

import math
import os
import os.path as osp
import socket
from datetime import datetime, timedelta, timezone

import torch
import torch.distributed as dist

try:  # optional, absent in this image
    from tensorboardX import SummaryWriter  # type: ignore
except Exception:  # noqa: BLE001
    class SummaryWriter:  # minimal stand-in: scalars/images are dropped
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def close(self): pass

try:
    import pytz  # type: ignore
    _TZ = pytz.timezone('Asia/Hong_Kong')
except Exception:  # noqa: BLE001
    _TZ = timezone(timedelta(hours=8))

try:
    import tqdm  # type: ignore
    def progress(it, **k): return tqdm.tqdm(it, **k)
    def trange(*a, **k): return tqdm.trange(*a, **k)
except Exception:  # noqa: BLE001
    def progress(it, **k): return it
    def trange(*a, **k): return range(*a)


def now():
    return datetime.now(_TZ)


def prefer_fused(optimizer):
    """Ask a stock torch optimizer (the entry script builds Adam / SGD objects and hands them in) for its fused
    multi-tensor step: one launch instead of ~40 `foreach` launches per step for the 186 generator tensors.
    Only fresh optimizers over device parameters are switched (a resumed one keeps the layout of its state)."""
    import torch
    if not isinstance(optimizer, (torch.optim.Adam, torch.optim.SGD)) or len(optimizer.state) > 0:
        return optimizer
    for g in optimizer.param_groups:
        if "fused" not in g or g.get("differentiable") or g.get("capturable"):
            return optimizer
        if not all(p.is_cuda and p.dtype == torch.float32 for p in g["params"]):
            return optimizer
    for g in optimizer.param_groups:
        g["fused"], g["foreach"] = True, False
    return optimizer


def get_lr(optimizer):
    for g in optimizer.param_groups:
        return g['lr']


class HipOps:
    """The device ops a Trainer uses (tests may hand a Trainer another object with these names)."""
    def __init__(self):
        from .. import ops
        from ..utils import metrics
        self.seg_loss = ops.seg_loss
        self.gen_prototype_from_labels = ops.gen_prototype_from_labels
        self.gen_prototype = ops.gen_prototype
        self.gen_prototype_retrify = ops.gen_prototype_retrify
        self.discriminative_loss = ops.discriminative_loss
        self.photometric_augment = ops.photometric_augment
        self.normalize_tf = ops.normalize_tf
        self.elastic_deform = ops.elastic_deform
        self.photometric_u8 = ops.photometric_u8
        self.consistency_loss = ops.consistency_loss
        self.proto_align = ops.proto_align
        self.adv_loss = ops.adv_loss
        self.dice_coeff_2label = metrics.dice_coeff_2label
        self.pixel_acc = metrics.pixel_acc


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def bootstrap_from_env():
    """One process per GPU (torch.distributed.run / torchrun env): pick the local device and join
    the RCCL group before the unchanged entry script calls ``.cuda()``."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or not dist.is_available() or dist.is_initialized():
        return
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")


def agree_int(value):
    """The same integer on every rank (rank 0's), e.g. a shuffling seed; identity for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, 0)
    return int(t.item())


def shard_count(n, world, batch_size=None):
    """Samples (or batches) every rank takes from n: ceil(n / world) - the set is padded by wrapping around, as
    torch's DistributedSampler does, so ALL ranks run the same number of iterations (a rank with one batch less would
    skip an all-reduce and hang the others).  With a batch size, a per-rank count that would leave a trailing batch of ONE
    sample is shortened by that sample: the image-pooling BatchNorm cannot train on a single image (quirk Q8,
    aspp.py:55-58) and would raise on every rank."""
    if n <= 0:
        return 0
    L = (n + world - 1) // world
    if batch_size and L > 1 and L % batch_size == 1:
        L -= 1
    return L


class RankSampler(torch.utils.data.Sampler):
    """Index stream of ONE rank: each epoch the same global order on every rank (a permutation drawn from ``seed + epoch``
    when shuffling, else 0..n-1), wrapped around to world * L entries, of which this rank takes every world-th starting at
    its rank (dataset index striding).  ``len`` is identical on all ranks."""

    def __init__(self, n, rank, world, shuffle, seed, batch_size=None):
        self.n, self.rank, self.world, self.shuffle, self.seed = n, rank, world, shuffle, int(seed)
        self.L = shard_count(n, world, batch_size)
        self.epoch = 0

    def __len__(self):
        return self.L

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        self.epoch += 1
        need = self.world * self.L
        order = (order * (need // max(len(order), 1) + 1))[:need] if order else []
        return iter(order[self.rank:need:self.world])


class _Strided:
    """Rank-strided view of a sequence of ready-made batches: rank r takes batches r, r + world, ... wrapped around so that
    every rank iterates ceil(n / world) times."""
    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world

    def __len__(self):
        return shard_count(len(self.loader), self.world)

    def __iter__(self):
        items = self.loader if hasattr(self.loader, "__getitem__") else list(self.loader)
        n = len(items)
        for i in range(shard_count(n, self.world)):
            yield items[(i * self.world + self.rank) % n]


def shard_loader(loader, rank, world, seed=None):
    """Every rank sees 1/world of an epoch and ALL ranks run the same number of iterations (see ``shard_count``).
    DataLoaders are rebuilt around a ``RankSampler`` (keeps batch size / workers / collate / pinning; the shuffling seed is
    rank 0's, so the ranks agree on each epoch's order); plain sequences of batches are strided directly."""
    if world == 1 or loader is None:
        return loader
    from torch.utils.data import DataLoader
    if isinstance(loader, DataLoader):
        shuffle = isinstance(loader.sampler, torch.utils.data.RandomSampler)
        if seed is None:
            seed = agree_int(torch.initial_seed() % (2 ** 31))
        sampler = RankSampler(len(loader.dataset), rank, world, shuffle, seed, loader.batch_size)
        return DataLoader(loader.dataset, batch_size=loader.batch_size, sampler=sampler, num_workers=loader.num_workers,
                          pin_memory=loader.pin_memory, collate_fn=loader.collate_fn, drop_last=loader.drop_last)
    return _Strided(loader, rank, world)


def sync_replicas(modules, rank, world):
    """Data-parallel start state: parameters AND buffers of every model are rank 0's (the entry script seeds every process
    alike, but nothing guarantees that a resumed checkpoint or an unseeded initialisation is identical everywhere; replicas
    that start apart never meet again and nothing would report it).  One flat broadcast per model."""
    if world <= 1:
        return
    for m in modules:
        if m is None:
            continue
        ts = [t for t in list(m.parameters()) + list(m.buffers()) if t.is_floating_point()]
        if not ts:
            continue
        flat = torch.cat([t.detach().reshape(-1).float() for t in ts])
        dist.broadcast(flat, 0)
        o = 0
        with torch.no_grad():
            for t in ts:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()
        ints = [t for t in m.buffers() if not t.is_floating_point()]
        for t in ints:                       # num_batches_tracked
            dist.broadcast(t, 0)


def decorrelate_dropout(model_gen, rank, world):
    """Each rank draws its own dropout masks: the engine's Philox key is offset by the rank (identical keys would apply the
    SAME masks to every rank's images, i.e. correlated noise across the global batch)."""
    if world > 1 and hasattr(model_gen, "set_dropout_seed"):
        model_gen.set_dropout_seed(1337 + 7919 * rank)


def nan_guard(values, what):
    for v in values:
        if math.isnan(v):
            raise ValueError('%s is nan while training' % what)


class TrainerBase(object):
    log_headers = []

    def _setup_io(self, out):
        self.rank, self.world = rank_world()
        self.out = out
        self.timestamp_start = now()
        if self.rank == 0:
            if not osp.exists(out):
                os.makedirs(out)
            if not osp.exists(osp.join(out, 'log.csv')):
                with open(osp.join(out, 'log.csv'), 'w') as f:
                    f.write(','.join(self.log_headers) + '\n')
            log_dir = osp.join(out, 'tensorboard', datetime.now().strftime('%b%d_%H-%M-%S') + '_' + socket.gethostname())
            self.writer = SummaryWriter(log_dir=log_dir)
        else:
            self.writer = SummaryWriter.__new__(SummaryWriter) if False else _NullWriter()

    def _device(self):
        return next(self.model_gen.parameters()).device

    def _to(self, t):
        return t.to(self._device(), non_blocking=True)

    def _decode(self, sample):
        """A batch whose Normalize_tf + ToTensor tail was deferred (dataloaders.custom_transforms.DEVICE_TAIL: uint8 image and
        grey mask) is decoded here, on the device, for the whole batch; any other sample passes through."""
        if 'image_u8' not in sample:
            return sample
        iu, lu = self._to(sample['image_u8']), self._to(sample['label_u8'])
        if 'aug_lut' in sample:          # UDA_CLR_DEVICE_INPUT=2: the recorded elastic / photometric outcomes, in the chain's order
            # 'aug_noise' ([B,2,H,W] float64): the uniform fields of the elastic transform when a caller supplies them (the parity
            # tests hand over numpy's draw); normally absent - the noise then comes from the device generator
            noise = self._to(sample['aug_noise']).transpose(0, 1).contiguous() if 'aug_noise' in sample else None
            iu, lu = self.ops.elastic_deform(iu, lu, apply=self._to(sample['aug_elastic']).view(-1), noise=noise)
            iu = self.ops.photometric_u8(iu.contiguous(), self._to(sample['aug_sp_pos']), self._to(sample['aug_sp_n']),
                                         self._to(sample['aug_sp_val']), self._to(sample['aug_lut']), self._to(sample['aug_erase']))
        image, mp, bd = self.ops.normalize_tf(iu, lu)
        out = dict(sample)
        out.update(image=image, map=mp, boundary=bd)
        return out

    def _fetch(self, scalars):
        """The step's single host sync: the loss scalars plus the generator's device-side non-finite flag (a NaN / Inf that
        went through any BatchNorm statistic of the step's passes; the fused activation clamps do not propagate NaN, see
        GeneratorEngine.nonfinite).  Raises ValueError like the reference's NaN checks (Trainer_prototype_full.py:296-299)."""
        flag = self.model_gen.pop_nonfinite() if hasattr(self.model_gen, "pop_nonfinite") else None
        ts = [s.detach().float().reshape(()) for s in scalars]
        if flag is not None:
            ts.append(flag.float().reshape(()))
        vals = torch.stack(ts).tolist()
        if flag is not None and vals.pop() > 0:
            raise ValueError('activations or gradients are nan/inf while training')
        nan_guard(vals, 'loss')
        return vals

    def _log_row(self, fields):
        if self.rank != 0:
            return
        with open(osp.join(self.out, 'log.csv'), 'a') as f:
            f.write(','.join(map(str, fields)) + '\n')

    def elapsed(self):
        return (now() - self.timestamp_start).total_seconds()

    # ---------------------------------------------------------------- validation (Trainer_*.validate)
    def _validate_core(self):
        """eval-mode pass over val_loader: mean BCE-with-logits, batch-level Dice (cup, disc) and
        PA / IoU, averaged over BATCHES (Trainer_prototype_full.py:110-159)."""
        import torch.nn.functional as F
        n = len(self.val_loader)
        acc = [0.0] * 7
        with torch.no_grad():
            for sample in progress(self.val_loader, total=n, desc='Valid iteration=%d' % self.iteration, ncols=80, leave=False):
                sample = self._decode(sample)
                data, target_map = self._to(sample['image']), self._to(sample['map'])
                predictions = self.model_gen(data)[0]
                loss = F.binary_cross_entropy_with_logits(predictions, target_map).item()
                if math.isnan(loss):
                    raise ValueError('loss is nan while validating')
                dc, dd = self.ops.dice_coeff_2label(predictions, target_map)
                pc, pd, ic, idc = self.ops.pixel_acc(predictions, target_map)
                for i, v in enumerate((loss, dc, dd, pc, pd, ic, idc)):
                    acc[i] += v
        return [v / max(n, 1) for v in acc]

    def _lr_schedule(self, epoch):
        """every 100 epochs lr = lr_gen * 0.2, not cumulative (quirk Q9, Trainer_prototype_full.py:637-640)"""
        if (epoch + 1) % 100 == 0:
            for g in self.optim_gen.param_groups:
                g['lr'] = self.lr_gen * 0.2


class _NullWriter:
    def add_scalar(self, *a, **k): pass
    def add_image(self, *a, **k): pass
    def close(self): pass
