"""Census of the conv / weight-gradient launches of one prototype_full step (GPU box): shape, kernel family, total time.
    python tests/tools/conv_census.py [--batch 16]"""
import os, sys, collections, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from uda_clr_amd.kernels import HipKernels, load_library
from uda_clr_amd.networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
from uda_clr_amd.networks.deeplabv3 import DeepLab
from uda_clr_amd.train_process import Trainer_prototype_full

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
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
oc, ow = HipKernels.conv, HipKernels.conv_wgrad


def conv(inst, src, w, ksize, dil, out, *aa, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = oc(inst, src, w, ksize, dil, out, *aa, **kw)
    e1.record()
    rec.append(("conv", src.P, src.C, out.shape[1], ksize, dil, bool(src.lazy), src.mask is not None, e0, e1))
    return r


def wgrad(inst, src, dy, ksize, dil, dw, *aa, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = ow(inst, src, dy, ksize, dil, dw, *aa, **kw)
    e1.record()
    rec.append(("wgrad", src.P, src.C, dy.shape[1], ksize, dil, bool(src.lazy), src.mask is not None, e0, e1))
    return r


HipKernels.conv, HipKernels.conv_wgrad = conv, wgrad
tr.train_step(sS, sT)
torch.cuda.synchronize()
HipKernels.conv, HipKernels.conv_wgrad = oc, ow
tot, cnt = collections.Counter(), collections.Counter()
for r in rec:
    k = r[:8]
    tot[k] += r[8].elapsed_time(r[9])
    cnt[k] += 1
print("%-6s %8s %5s %5s %2s %3s %5s %5s %5s %9s %9s %8s %8s" % ("kind", "P", "Cin", "Cout", "k", "dil", "lazy", "mask", "calls", "ms/step", "avg_us", "TF", "GB/s"))
allms = 0.0
for k, ms in tot.most_common():
    kind, P, Cin, Cout, ks, dil, lazy, mask = k
    fl = 2.0 * P * Cin * Cout * ks * ks * cnt[k]
    by = 4.0 * (P * (Cin + Cout) + Cin * Cout * ks * ks) * cnt[k]
    allms += ms
    print("%-6s %8d %5d %5d %2d %3d %5s %5s %5d %9.3f %9.1f %8.1f %8.0f" % (kind, P, Cin, Cout, ks, dil, lazy, mask, cnt[k], ms, 1e3 * ms / cnt[k],
                                                                          fl / ms / 1e9, by / ms / 1e6))
print("total %.2f ms in %d launches (packing passes of the bf16x3 mode included in their conv)" % (allms, len(rec)))
