import os, sys, time, json, subprocess
sys.path.insert(0, os.getcwd())
import torch
import bench
# reuse bench's setup by monkeypatching: run main() pieces manually
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
from uda_clr_amd.kernels import load_library
from uda_clr_amd.networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
from uda_clr_amd.networks.deeplabv3 import DeepLab
from uda_clr_amd.train_process import Trainer_prototype_full
dev = torch.device("cuda", 0)
load_library(); torch.manual_seed(1337)
B = 16
model = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16, sync_bn=True, freeze_bn=False, method="prototype_full").to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.99))
img, tmap, tbd = bench.synth_batch(B, 512, 1337, dev); imgT = bench.synth_batch(B, 512, 4242, dev)[0]
d1, d2 = BoundaryDiscriminator().to(dev).train(), UncertaintyDiscriminator().to(dev).train()
od = torch.optim.SGD(d1.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
od2 = torch.optim.SGD(d2.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
tr = Trainer_prototype_full.Trainer(cuda=True, model_gen=model, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=opt, optimizer_dis=od,
    optimizer_uncertainty_dis=od2, val_loader=[], domain_loaderS=[], domain_loaderT=[], out="/tmp/ht", max_epoch=1, use_global=True, use_pid=True,
    retrify_pesudo=True, global_pro_weight=0.9, pro_weight=0.1, batch_size=B, warmup_epoch=-1)
tr.epoch = 0
sS, sT = {"image": img, "map": tmap, "boundary": tbd}, {"image": imgT}
for _ in range(3): tr.train_step(sS, sT)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter(); tr.train_step(sS, sT); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("one step: host returns after %.1f ms, GPU done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# host enqueue time of the generator's training forward + backward alone (no host sync inside)
from uda_clr_amd import ops
for prm in model.parameters():
    prm.requires_grad_(True)
for _ in range(2):
    opt.zero_grad(set_to_none=True)
    o = model(img); l = ops.seg_loss(o[0], o[1], tmap, tbd); l.backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
opt.zero_grad(set_to_none=True)
o = model(img); l = ops.seg_loss(o[0], o[1], tmap, tbd); l.backward()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("generator fwd+bwd (B=16): host enqueue %.1f ms, GPU done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# memory stability over many iterations of the full step (no growth expected: contexts are released by backward / _forget)
for prm in model.parameters():
    prm.requires_grad_(True)
torch.cuda.reset_peak_memory_stats()
marks = []
for it in range(40):
    tr.train_step(sS, sT)
    if it in (4, 39):
        torch.cuda.synchronize()
        marks.append((it, torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
for m in marks:
    print("after step %d: allocated %.2f GiB, peak %.2f GiB, reserved %.2f GiB" % m)
