"""Census of EVERY kernel-binding call of one prototype_full step (GPU box): method, operand shapes, launches, total time and the
bytes of its tensor operands per second (inputs + outputs, each counted once: a rough achieved-bandwidth figure that ranks the
memory-bound kernels).      python tests/tools/kernel_census.py [--batch 16] [--top 60]"""
import os, sys, collections, argparse, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from uda_clr_amd.acts import Act
from uda_clr_amd.kernels import HipKernels, load_library
from uda_clr_amd.networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
from uda_clr_amd.networks.deeplabv3 import DeepLab
from uda_clr_amd.train_process import Trainer_prototype_full

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--top", type=int, default=70)
a = ap.parse_args()
dev = torch.device("cuda:0")
load_library()
torch.manual_seed(1337)
model = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16, sync_bn=True, freeze_bn=False, method="prototype_full").to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.99))
img, tmap, tbd = bench.synth_batch(a.batch, 512, 1337, dev)
imgT = bench.synth_batch(a.batch, 512, 4242, dev)[0]
d1, d2 = BoundaryDiscriminator().to(dev).train(), UncertaintyDiscriminator().to(dev).train()
od = torch.optim.SGD(d1.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
od2 = torch.optim.SGD(d2.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
tr = Trainer_prototype_full.Trainer(cuda=True, model_gen=model, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=opt, optimizer_dis=od,
                                    optimizer_uncertainty_dis=od2, val_loader=[], domain_loaderS=[], domain_loaderT=[], out="/tmp/census", max_epoch=1,
                                    use_global=True, use_pid=True, retrify_pesudo=True, global_pro_weight=0.9, pro_weight=0.1, batch_size=a.batch, warmup_epoch=-1)
tr.epoch = 0
sS, sT = {"image": img, "map": tmap, "boundary": tbd}, {"image": imgT}
for _ in range(3):
    tr.train_step(sS, sT)
torch.cuda.synchronize()
rec = []
depth = [0]


def tensors(v, out):
    if isinstance(v, torch.Tensor):
        out.append(v)
    elif isinstance(v, Act):
        out.append(v.x)
        if v.mask is not None:
            out.append(v.mask)
    elif isinstance(v, (list, tuple)):
        for t in v:
            tensors(t, out)


def wrap(name, fn):
    def f(inst, *args, **kw):
        if depth[0]:                   # nested binding calls (image groups, packing inside conv) belong to the outer call
            return fn(inst, *args, **kw)
        depth[0] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            r = fn(inst, *args, **kw)
        finally:
            depth[0] -= 1
        e1.record()
        ts = []
        tensors(list(args) + list(kw.values()), ts)
        if isinstance(r, torch.Tensor):
            ts.append(r)
        seen, by, shp = set(), 0, []
        for t in ts:
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            by += t.numel() * t.element_size()
            if t.numel() > 4096:
                shp.append("x".join(map(str, t.shape)))
        rec.append((name, " ".join(shp[:3]), by, e0, e1))
        return r
    return f


orig = {}
for name, fn in inspect.getmembers(HipKernels, predicate=inspect.isfunction):
    if name.startswith("_") or name in ("x3_pack", "x3_pack_rows", "x3_pack_act"):
        continue
    orig[name] = fn
    setattr(HipKernels, name, wrap(name, fn))
tr.train_step(sS, sT)
torch.cuda.synchronize()
for name, fn in orig.items():
    setattr(HipKernels, name, fn)
tot, cnt, byt = collections.Counter(), collections.Counter(), collections.Counter()
for name, shp, by, e0, e1 in rec:
    k = (name, shp)
    tot[k] += e0.elapsed_time(e1)
    cnt[k] += 1
    byt[k] += by
fam = collections.Counter()
for (name, shp), ms in tot.items():
    fam[name] += ms
print("per binding method (ms per step):", ", ".join("%s %.2f" % kv for kv in fam.most_common()))
print("%-22s %-58s %5s %8s %8s %8s" % ("method", "large operands", "calls", "ms/step", "avg_us", "GB/s"))
for k, ms in tot.most_common(a.top):
    print("%-22s %-58s %5d %8.3f %8.1f %8.0f" % (k[0], k[1][:58], cnt[k], ms, 1e3 * ms / cnt[k], byt[k] / ms / 1e6))
print("total %.2f ms in %d calls" % (sum(tot.values()), len(rec)))
