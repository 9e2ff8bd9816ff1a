#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own modules (networks.deeplabv3, utils.Utils, utils.metrics,
train_process.Trainer_baseline / Trainer_prototype_full) with the local monkey-patches of
SURVEY.md section 8c (no edits to the reference, nothing copied from it), feeds them seeded
inputs and stores inputs' seeds + expected outputs as small data files:

  manifest.json          state-dict keys/shapes of DeepLab(mobilenet) + seeded-init checksums
  manifest_resnet.json   the same for DeepLab(resnet) (ResNet-101, pretrained fetch patched out)
  manifest_tn.json, forward_tn_64.npz   the --use_TN model (DeepLab(sync_bn=False): TransNorm layers), B = 4
  forward_*.npz          7-tuple outputs (checksums + strided samples), BN running stats,
                         seg loss and per-parameter gradient norms (eval / train, 64^2 / 512^2)
  proto.npz              gen_prototype / gen_prototype_retrify inputs (by seed) and outputs
  metrics.json           dice_coeff_2label / pixel_acc on seeded logits
  trainer_*.json         loss rows written by the reference's own Trainer loops
  input_pipeline.json, input_pipeline_small.npz   outputs of the reference's dataloaders/custom_transforms.py on seeded samples
  forward_frozen_64.npz  DeepLab(freeze_bn=True) in training mode (eval-mode BatchNorm, live dropout): outputs, loss, gradient norms
  forward_frozen_tn_64.npz  the same for DeepLab(sync_bn=False, freeze_bn=True): frozen TransNorm layers, B = 4

While generating, every fixture is also compared with the oracle restatement (oracle/), so a
successful run pins the oracle against the reference on full tensors, not only on the samples
that are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------- shims (8c)
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _lazy_attr(factory):
    def getter(name):
        if name.startswith("__"):
            raise AttributeError(name)
        return factory()
    return getter


def install_reference():
    import torch._dynamo  # noqa: F401  (import before the stubs so its introspection never sees them)
    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return None
        def __getattr__(self, n): return _Any()
    for n in ("cv2", "albumentations", "skimage", "skimage.morphology", "skimage.measure",
              "skimage.transform"):
        _stub(n, __getattr__=_lazy_attr(_Any))
    sw = type("SummaryWriter", (), {"__init__": lambda s, *a, **k: None,
                                    "__getattr__": lambda s, n: (lambda *a, **k: None)})
    _stub("tensorboardX", SummaryWriter=sw)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", make_grid=lambda t, *a, **k: t)
        tv.transforms = _stub("torchvision.transforms")
    if not hasattr(np, "bool"):
        np.bool = bool
    if not hasattr(np, "float"):
        np.float = float
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from networks.backbone import mobilenet, resnet
    mobilenet.MobileNetV2._load_pretrained_model = lambda self: None
    resnet.ResNet._load_pretrained_model = lambda self: None


def ref_model(seed=1337, backbone="mobilenet", sync_bn=True):
    """sync_bn=False is the --use_TN model of train_use_fix_initial.py:180-181 (TransNorm layers)."""
    from networks.deeplabv3 import DeepLab
    torch.manual_seed(seed)
    return DeepLab(num_classes=2, backbone=backbone, output_stride=16, sync_bn=sync_bn,
                   freeze_bn=False, method="prototype_full")


# ----------------------------------------------------------------------------- helpers
NAMES = ("x1", "x2", "feature", "x_bu_feature", "x_feature", "x1_before", "x2_before")


def sample(t, n=97):
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].numpy()


def summarize(prefix, t, out):
    d = t.detach().double()
    out[prefix + ".sum"] = np.float64(d.sum().item())
    out[prefix + ".abs"] = np.float64(d.abs().sum().item())
    out[prefix + ".smp"] = sample(t)


def synth_targets(B, H, W, seed):
    """Seeded concentric-ellipse cup/disc maps + soft ring boundary (no scipy needed)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    maps, bds = [], []
    for _ in range(B):
        cy, cx = (0.4 + 0.2 * torch.rand(2, generator=g)) * torch.tensor([H, W])
        a, b = (0.18 + 0.09 * torch.rand(2, generator=g)) * min(H, W)
        k = 0.4 + 0.3 * torch.rand(1, generator=g)
        r = torch.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2)
        disc, cup = (r <= 1).float(), (r <= k).float()
        ring = torch.exp(-((r - 1) * min(a, b) / 3.0) ** 2) + torch.exp(-((r - k) * min(a, b) / 3.0) ** 2)
        maps.append(torch.stack([cup, disc]))
        bds.append(ring.clamp(0, 1)[None])
    return torch.stack(maps), torch.stack(bds)


def check(name, a, b, tol=1e-5):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    status = "ok" if err <= tol else "MISMATCH"
    print("  oracle-vs-reference %-42s rel-max-err %.3e  %s" % (name, err, status))
    if err > tol:
        raise SystemExit("oracle restatement disagrees with the reference: " + name)


# ----------------------------------------------------------------------------- fixtures
def make_manifest(backbone="mobilenet", fname="manifest.json", sync_bn=True):
    m = ref_model(backbone=backbone, sync_bn=sync_bn)
    sd = m.state_dict()
    entries = [{"key": k, "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                "sum": float(v.double().sum())} for k, v in sd.items()]
    params = list(m.parameters())
    man = {"n_state_keys": len(sd), "n_param_tensors": len(params),
           "n_params": int(sum(p.numel() for p in params)),
           "param_sum": float(sum(p.double().sum() for p in params)),
           "param_abs_sum": float(sum(p.double().abs().sum() for p in params)),
           "seed": 1337, "entries": entries}
    with open(os.path.join(HERE, fname), "w") as f:
        json.dump(man, f, indent=0)
    print("manifest: %d keys, %d param tensors, %d params, sum=%.6f" %
          (man["n_state_keys"], man["n_param_tensors"], man["n_params"], man["param_sum"]))
    return m


def make_forward(m, B, S, tag, do_eval=True):
    from oracle import deeplab_ref, step_ref
    out = {"B": B, "S": S, "input_seed": 0, "dropout_seed": 7, "target_seed": 11}
    torch.manual_seed(0)
    x = torch.randn(B, 3, S, S)
    tmap, tbd = synth_targets(B, S, S, 11)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    # ---- eval (skipped for the ResNet-101 TransNorm model: with its initial buffers eval-mode TransNorm doubles the output of
    # each of 104 layers and overflows fp32)
    if do_eval:
        m.eval()
        with torch.no_grad():
            ref = m(x)
        osd = deeplab_ref.canonical_state(sd0)
        with torch.no_grad():
            mine = deeplab_ref.deeplab_forward(osd, x, training=False)
        for n, r, o in zip(NAMES, ref, mine):
            summarize("eval." + n, r, out)
            check(tag + " eval " + n, o, r)
    # ---- train: forward (dropout from the global generator), seg loss, backward
    m.load_state_dict(sd0)
    m.train()
    m.zero_grad()
    torch.manual_seed(7)
    ref = m(x)
    loss = torch.nn.BCELoss()(torch.sigmoid(ref[0]), tmap) + torch.nn.MSELoss()(torch.sigmoid(ref[1]), tbd)
    loss.backward()
    osd = deeplab_ref.canonical_state(sd0, requires_grad=True)
    rec = {}
    torch.manual_seed(7)
    mine = deeplab_ref.deeplab_forward(osd, x, training=True, record=rec)
    oloss = step_ref.seg_loss(mine[0], mine[1], tmap, tbd)
    oloss.backward()
    for n, r, o in zip(NAMES, ref, mine):
        summarize("train." + n, r, out)
        check(tag + " train " + n, o, r)
    check(tag + " train loss", oloss, loss, 1e-6)
    out["train.loss"] = np.float64(loss.item())
    gn, keys = [], []
    for k, p in m.named_parameters():
        if ".low_level_features." in k or ".high_level_features." in k:
            continue
        keys.append(k)
        gn.append(p.grad.double().norm().item())
        check(tag + " grad " + k, osd[k].grad, p.grad, 2e-4)
    out["train.grad_norm"] = np.array(gn)
    out["train.grad_keys"] = np.array(keys)
    rs = m.state_dict()
    bn_keys = [k for k in rs if k.rsplit(".", 1)[-1].startswith(("running_mean", "running_var"))]
    bn_keys = [k for k in bn_keys if "_level_features" not in k]
    out["train.bn_keys"] = np.array(bn_keys)
    out["train.bn_sum"] = np.array([rs[k].double().sum().item() for k in bn_keys])
    for k in bn_keys:
        check(tag + " running " + k, osd[k], rs[k], 1e-5)
    # the keep-masks the reference drew (recovered through the oracle's identical draw)
    for k, v in rec.items():
        out["mask." + k + ".sum"] = np.int64(v.sum().item())
    m.load_state_dict(sd0)
    np.savez_compressed(os.path.join(HERE, "forward_%s.npz" % tag), **out)


def make_frozen(B=2, S=64, transnorm=False):
    """transnorm=True: DeepLab(sync_bn=False, freeze_bn=True) - freeze_bn() evals the TransNorm layers too (deeplabv3.py:47-50), so
    the TARGET running statistics normalise and the gain 1 + alpha comes from both domains' running statistics
    (batchnorm.py:497-520) -> forward_frozen_tn_64.npz.
    DeepLab(freeze_bn=True) (deeplabv3.py:43-50; train_use_fix_initial.py:92-96 turns any --freeze-bn value into True): the
    BatchNorm modules sit in eval mode (running statistics normalise, nothing is updated) while the model itself trains -
    dropout active, gamma / beta and all weights receive gradients.  Running statistics and affine parameters are perturbed
    (seeded) so the frozen statistics differ from the batch's."""
    from networks.deeplabv3 import DeepLab
    from oracle import deeplab_ref, step_ref
    torch.manual_seed(1337)
    from networks.sync_batchnorm.batchnorm import BatchNorm2d as RefTransNorm
    m = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16, sync_bn=not transnorm, freeze_bn=True, method="prototype_full")
    g = torch.Generator().manual_seed(5)
    for k, v in m.state_dict().items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf.startswith("running_mean"):           # incl. TransNorm's *_source / *_target buffers
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif leaf.startswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, RefTransNorm)):
            assert not mod.training
            mod.weight.data.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
            mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    assert m.training
    if transnorm:
        # TransNorm scales every layer by 1 + alpha (2 on average): on statistics that do not describe the activations the network
        # is un-normalised and any two fp32 evaluation orders drift apart ~2.3x per block.  So the frozen statistics are CALIBRATED:
        # one training-mode forward with momentum 1 on a 6-image batch whose halves differ (running := batch statistics per domain)
        bns = [mod for mod in m.modules() if isinstance(mod, RefTransNorm)]
        for mod in bns:
            mod.train()
            mod.momentum = 1.0
        xc = torch.randn(6, 3, S, S, generator=torch.Generator().manual_seed(1))
        xc[3:] = 0.6 * xc[3:] - 0.3
        with torch.no_grad():
            m(xc)
        for mod in bns:
            mod.momentum = 0.1
        m.freeze_bn()
        assert m.training and not any(mod.training for mod in bns)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    out = {"B": B, "S": S, "input_seed": 0, "dropout_seed": 7, "target_seed": 11, "perturb_seed": 5, "calibration_seed": 1}
    torch.manual_seed(0)
    x = torch.randn(B, 3, S, S)
    tmap, tbd = synth_targets(B, S, S, 11)
    torch.manual_seed(7)
    ref = m(x)
    loss = torch.nn.BCELoss()(torch.sigmoid(ref[0]), tmap) + torch.nn.MSELoss()(torch.sigmoid(ref[1]), tbd)
    loss.backward()
    osd = deeplab_ref.canonical_state(sd0, requires_grad=True)
    rec = {}
    torch.manual_seed(7)
    mine = deeplab_ref.deeplab_forward(osd, x, training=True, record=rec, bn_training=False)
    oloss = step_ref.seg_loss(mine[0], mine[1], tmap, tbd)
    oloss.backward()
    for n, r, o in zip(NAMES, ref, mine):
        summarize("train." + n, r, out)
        check("frozen train " + n, o, r)
    check("frozen train loss", oloss, loss, 1e-6)
    out["train.loss"] = np.float64(loss.item())
    gn, keys = [], []
    for k, p in m.named_parameters():
        if ".low_level_features." in k or ".high_level_features." in k:
            continue
        keys.append(k)
        gn.append(p.grad.double().norm().item())
        check("frozen grad " + k, osd[k].grad, p.grad, 2e-4)
    out["train.grad_norm"] = np.array(gn)
    out["train.grad_keys"] = np.array(keys)
    rs = m.state_dict()
    for k in rs:
        if k.rsplit(".", 1)[-1].startswith(("running_", "num_batches")):
            assert torch.equal(rs[k], sd0[k]), "frozen statistics moved: " + k
    for k, v in rec.items():
        out["mask." + k + ".sum"] = np.int64(v.sum().item())
    np.savez_compressed(os.path.join(HERE, "forward_frozen_%s%d.npz" % ("tn_" if transnorm else "", S)), **out)


def make_proto():
    from utils.Utils import gen_prototype, gen_prototype_retrify
    from oracle import proto_ref
    out = {}
    g = torch.Generator().manual_seed(21)
    # gen_prototype: hard labels and soft predictions
    B, C, h = 2, 305, 32
    feat = torch.randn(B, C, h, h, generator=g)
    hard = (torch.rand(B, 2, h, h, generator=g) > 0.6).float()
    soft = torch.rand(B, 2, h, h, generator=g)
    for tag, pred in (("hard", hard), ("soft", soft)):
        ref = gen_prototype(pred, feat)
        mine = proto_ref.gen_prototype(pred, feat)
        for i, (r, o) in enumerate(zip(ref, mine)):
            out["gp.%s.%d" % (tag, i)] = r.reshape(-1).numpy()
            check("gen_prototype %s %d" % (tag, i), o, r, 1e-6)
    # soft branch with gradient into the prediction (Trainer_prototype_full.py:375-377, quirk Q6): seeded cotangents
    v = torch.randn(4, C, generator=g)
    for tag, fn in (("ref", gen_prototype), ("mine", proto_ref.gen_prototype)):
        sp, ft = soft.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        sum((c.reshape(-1) * v[i]).sum() for i, c in enumerate(fn(sp, ft))).backward()
        if tag == "ref":
            d_pred, d_feat = sp.grad, ft.grad
            summarize("gp.soft.d_pred", d_pred, out)
            summarize("gp.soft.d_feat", d_feat, out)
        else:
            check("gen_prototype soft d_pred", sp.grad, d_pred, 1e-5)
            check("gen_prototype soft d_feat", ft.grad, d_feat, 1e-5)
    out["gp.seed"] = 21
    # gen_prototype_retrify: hard-coded 305 x 128 x 128 and 512^2 predictions (quirk Q7)
    g = torch.Generator().manual_seed(22)
    B, T = 1, 8
    base = 2.0 * torch.randn(B, 2, 512, 512, generator=g)
    base = torch.nn.functional.avg_pool2d(base, 9, 1, 4) * 6.0
    preds = base.repeat(T, 1, 1, 1) + 0.35 * torch.randn(T * B, 2, 512, 512, generator=g) * \
        (torch.rand(1, 2, 512, 512, generator=g) > 0.5).float()
    oT_before = torch.nn.functional.interpolate(base, size=(128, 128), mode="bilinear",
                                                align_corners=True).clone().requires_grad_(True)
    xt = torch.randn(B, 305, 128, 128, generator=g).requires_grad_(True)
    feats = torch.zeros(T * B, 305, 128, 128)
    ref = gen_prototype_retrify(oT_before, xt, preds, feats, T, B)
    mine = proto_ref.gen_prototype_retrify(oT_before, xt, preds, T, B)
    names = ("c0_obj", "c1_obj", "c0_bck", "c1_bck", "std_map", "mask_0", "mask_1")
    for n, r, o in zip(names, ref, mine):
        check("gen_prototype_retrify " + n, o, r, 1e-6)
        if n.startswith("c"):
            out["rt." + n] = r.detach().reshape(-1).numpy()
        else:
            summarize("rt." + n, r, out)
    (sum(r.sum() for r in ref[:4])).backward()
    out["rt.grad_xt.abs"] = np.float64(xt.grad.double().abs().sum().item())
    out["rt.grad_oT.abs"] = np.float64(0.0 if oT_before.grad is None else oT_before.grad.abs().sum().item())
    out["rt.seed"] = 22
    np.savez_compressed(os.path.join(HERE, "proto.npz"), **out)
    print("  retrify: mask_0 on %.0f px, mask_1 on %.0f px, grad_oT abs %.3g" %
          (ref[5].sum().item() / 2, ref[6].sum().item() / 2, out["rt.grad_oT.abs"]))


def make_metrics():
    from utils.metrics import dice_coeff_2label, pixel_acc
    from oracle import metrics_ref
    g = torch.Generator().manual_seed(31)
    tmap, _ = synth_targets(3, 96, 96, 32)
    logits = (tmap * 2 - 1) * 2.0 + 1.5 * torch.randn(3, 2, 96, 96, generator=g)
    d = dice_coeff_2label(logits.clone(), tmap)
    p = pixel_acc(logits.clone(), tmap)
    md, mp = metrics_ref.dice_coeff_2label(logits, tmap), metrics_ref.pixel_acc(logits, tmap)
    assert np.allclose(d, md, rtol=0, atol=1e-12) and np.allclose(p, mp, rtol=0, atol=1e-12), (d, md, p, mp)
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump({"logit_seed": 31, "target_seed": 32, "B": 3, "S": 96,
                   "dice": [float(v) for v in d], "pixel_acc": [float(v) for v in p]}, f)
    print("metrics: dice", d, "pa/iou", p)


def synth_loader(n_batches, B, S, seed):
    out = []
    for i in range(n_batches):
        g = torch.Generator().manual_seed(seed + i)
        tmap, tbd = synth_targets(B, S, S, seed + 100 + i)
        img = (torch.rand(B, 3, S, S, generator=g) * 2 - 1) * 0.5 + (tmap[:, 1:2] * 0.3 + tmap[:, 0:1] * 0.3)
        out.append({"image": img, "map": tmap, "boundary": tbd, "img_name": ["s%d" % i] * B})
    return out


def read_log(path):
    rows = []
    with open(path) as f:
        for line in f.read().strip().split("\n")[1:]:
            rows.append(line.split(","))
    return rows


def make_trainer_baseline(tmp, S=64, B=4, nS=3, nV=2, epochs=2, seedS=500, seedV=700, fname="trainer_baseline.json"):
    """The reference's own Trainer_baseline loop on in-memory synthetic loaders: 4 x 64^2 for 2 epochs x 3 iterations
    (trainer_baseline.json), and the BASELINE.json configs[0] shape, 8 x 256^2, for one epoch of 4 iterations
    (trainer_baseline_256.json: BN statistics over >= 2048 samples in every layer, so two fp32 evaluation orders stay on the same
    trajectory)."""
    from train_process import Trainer_baseline
    from oracle import deeplab_ref, step_ref, metrics_ref
    m = ref_model()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    loaderS, loaderV = synth_loader(nS, B, S, seedS), synth_loader(nV, B, S, seedV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.99))
    out = os.path.join(tmp, "baseline_%d" % S)
    torch.manual_seed(99)
    tr = Trainer_baseline.Trainer(cuda=False, model_gen=m, optimizer_gen=opt, lr_gen=1e-3,
                                  lr_decrease_rate=0.1, val_loader=loaderV, domain_loaderS=loaderS,
                                  domain_loaderT=loaderS, out=out, max_epoch=epochs, stop_epoch=epochs,
                                  interval_validate=1, batch_size=B, warmup_epoch=-1)
    tr.epoch = 0; tr.iteration = 0
    tr.train()
    rows = read_log(os.path.join(out, "log.csv"))
    train_loss = [float(r[2]) for r in rows if r[2] != ""]
    val_rows = [r for r in rows if r[2] == ""]
    # oracle re-run of the same trajectory
    om = deeplab_ref.OracleDeepLab(sd0)
    oo = torch.optim.Adam(om.parameters(), lr=1e-3, betas=(0.9, 0.99))
    torch.manual_seed(99)
    mine, mval = [], []
    for ep in range(epochs):
        om.train()
        for s in loaderS:
            mine.append(step_ref.baseline_step(om, oo, s["image"], s["map"], s["boundary"]))
        om.eval()
        vl = vc = vd = 0.0
        with torch.no_grad():
            for s in loaderV:
                p = om(s["image"])[0]
                vl += torch.nn.functional.binary_cross_entropy_with_logits(p, s["map"]).item()
                c, d = metrics_ref.dice_coeff_2label(p, s["map"])
                vc += c; vd += d
        mval.append((vl / len(loaderV), vc / len(loaderV), vd / len(loaderV)))
    print("  baseline trainer loss rows (reference):", train_loss)
    print("  baseline trainer loss rows (oracle)   :", mine)
    assert np.allclose(train_loss, mine, rtol=2e-4), "oracle baseline trajectory differs"
    ref_val = []
    for r in val_rows:
        txt = ",".join(r)
        tup = txt[txt.index("(") + 1: txt.index(")")].split(",")
        ref_val.append([float(v.replace("np.float64(", "").replace(")", "")) for v in tup])
    print("  val (reference):", ref_val, "\n  val (oracle)   :", mval)
    with open(os.path.join(HERE, fname), "w") as f:
        json.dump({"S": S, "B": B, "loaderS_seed": seedS, "loaderV_seed": seedV, "n_batches_S": nS,
                   "n_batches_V": nV, "epochs": epochs, "torch_seed": 99, "train_loss": train_loss,
                   "val": ref_val}, f)


def make_trainer_proto(tmp):
    """The reference's own Trainer_prototype_full loop (shipped signature, use_global=True,
    use_pid, retrify_pesudo, warmup_epoch=-1) for 2 iterations at 512^2, B=1+1 is impossible
    (GAP-branch BN needs batch >= 2, quirk Q8) so B=2."""
    from train_process import Trainer_prototype_full
    from networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
    from oracle import deeplab_ref, step_ref
    m = ref_model()
    torch.manual_seed(1338)
    d1, d2 = BoundaryDiscriminator(), UncertaintyDiscriminator()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    dsd1 = {k: v.clone() for k, v in d1.state_dict().items()}
    dsd2 = {k: v.clone() for k, v in d2.state_dict().items()}
    S, B = 512, 2
    loaderS, loaderT = synth_loader(2, B, S, 900), synth_loader(2, B, S, 950)
    og, od, od2 = step_ref.make_optimizers(m, d1, d2)
    out = os.path.join(tmp, "proto")
    torch.manual_seed(199)
    tr = Trainer_prototype_full.Trainer(
        cuda=False, model_gen=m, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=og,
        optimizer_dis=od, optimizer_uncertainty_dis=od2, val_loader=loaderT, domain_loaderS=loaderS,
        domain_loaderT=loaderT, out=out, max_epoch=1, use_global=True, use_pid=True,
        retrify_pesudo=True, global_pro_weight=0.9, pro_weight=0.1, stop_epoch=1, lr_gen=1e-3,
        lr_dis=2.5e-5, interval_validate=100, batch_size=B, warmup_epoch=-1)
    tr.epoch = 0; tr.iteration = 0
    tr.train()
    rows = read_log(os.path.join(out, "log.csv"))
    ref_rows = [[float(v) for v in r[2:8]] for r in rows if r[2] != ""]
    # oracle: same weights, same seeds
    om = deeplab_ref.OracleDeepLab(sd0)
    o1, o2 = BoundaryDiscriminator(), UncertaintyDiscriminator()
    o1.load_state_dict(dsd1); o2.load_state_dict(dsd2)
    oog, ood, ood2 = step_ref.make_optimizers(om, o1, o2)
    step = step_ref.PrototypeFullStep(om, o1, o2, oog, ood, ood2)
    om.train(); o1.train(); o2.train()
    torch.manual_seed(199)
    mine = []
    for sS, sT in zip(loaderS, loaderT):
        r = step(sS["image"], sS["map"], sS["boundary"], sT["image"])
        mine.append([r["seg"], r["adv"], r["D_same"], r["D_diff"], r["intra"], r["inter"]])
    print("  proto trainer rows (reference):", ref_rows)
    print("  proto trainer rows (oracle)   :", mine)
    assert np.allclose(ref_rows, mine, rtol=5e-4), "oracle prototype_full trajectory differs"
    with open(os.path.join(HERE, "trainer_proto.json"), "w") as f:
        json.dump({"S": S, "B": B, "loaderS_seed": 900, "loaderT_seed": 950, "n_batches": 2,
                   "dis_seed": 1338, "torch_seed": 199, "columns": ["seg", "adv", "D_same", "D_diff",
                                                                    "intra", "inter"],
                   "rows": ref_rows}, f)


# ----------------------------------------------------------------------------- input pipeline
def digest(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return "%s%s:%s" % (a.dtype.str, list(a.shape), hashlib.sha256(a.tobytes()).hexdigest())


def make_input_pipeline():
    """The reference's own dataloaders/custom_transforms.py (:22-147 salt-and-pepper / adjust_light / eraser /
    elastic_transform, :414-466 GetBoundary / Normalize_tf, :496-507 ToTensor) on seeded uint8 samples
    (tests/make_golden_inputs.fundus_u8).  It needs cv2 only for cv2.LUT (a table lookup, stubbed as table[image]) and the
    removed alias np.float.  elastic_transform seeds its noise from OS entropy (RandomState(None)): numpy's RandomState is
    swapped for a seeded one while it runs, so a test can redraw the same uniform fields.  Stores, per case, the seeds and the
    SHA-256 of every output array (+ the full arrays of the small cases, to locate a mismatching byte)."""
    import random
    from PIL import Image
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from make_golden_inputs import fundus_u8
    sys.modules["cv2"].LUT = lambda img, table: table[img]
    from dataloaders import custom_transforms as rt
    meta, small = {}, {}

    def tensors(s):
        return {k: s[k].numpy() for k in ("image", "map", "boundary")}

    def record(name, arrays, info, keep):
        meta[name] = dict(info, sha256={k: digest(v) for k, v in arrays.items()})
        if keep:
            for k, v in arrays.items():
                small["%s.%s" % (name, k)] = np.ascontiguousarray(v)

    # ---- Normalize_tf + GetBoundary + ToTensor
    for tag, (B, H, W, seed) in {"ntf_small": (3, 96, 80, 41), "ntf_512": (2, 512, 512, 42)}.items():
        img, lab = fundus_u8(B, H, W, seed)
        for b in range(B):
            s = rt.ToTensor()(rt.Normalize_tf()({"image": Image.fromarray(img[b]), "label": Image.fromarray(lab[b]), "img_name": "s"}))
            record("%s.%d" % (tag, b), tensors(s), dict(B=B, H=H, W=W, seed=seed, b=b), H < 128 and b == 0)

    # ---- elastic_transform with a seeded RandomState (fires when random.random() > 0.5)
    RS = np.random.RandomState

    def fire_seed(want, start):        # a `random` seed whose first draw does / does not fire the transform
        s = start
        while (random.Random(s).random() > 0.5) != want:
            s += 1
        return s
    for tag, (B, S, seed) in {"elastic_small": (2, 96, 43), "elastic_512": (1, 512, 44)}.items():
        img, lab = fundus_u8(B, S, S, seed)
        for b in range(B):
            np_seed, py_seed = 1000 + seed + b, fire_seed(True, 10 * seed + b)
            np.random.RandomState = lambda s=None, _k=np_seed: RS(_k)
            try:
                random.seed(py_seed)
                o = rt.elastic_transform()({"image": Image.fromarray(img[b]), "label": Image.fromarray(lab[b]), "img_name": "s"})
            finally:
                np.random.RandomState = RS
            assert not np.array_equal(o["image"], img[b])
            record("%s.%d" % (tag, b), {"image": o["image"], "label": o["label"]},
                   dict(B=B, H=S, W=S, seed=seed, b=b, noise_seed=np_seed, py_seed=py_seed), S < 128 and b == 0)

    # ---- photometric transforms: every branch (salt / pepper / none, gamma / none, erase / none)
    img, lab = fundus_u8(1, 96, 96, 45)
    def branch_seeds(pred, n=1, start=0):
        out, s = [], start
        while len(out) < n:
            if pred(random.Random(s).random()):
                out.append(s)
            s += 1
        return out
    sp_seeds = branch_seeds(lambda r: r > 0.75) + branch_seeds(lambda r: 0.5 < r <= 0.75) + branch_seeds(lambda r: r <= 0.5)
    for s_ in sp_seeds:
        random.seed(s_); np.random.seed(s_)
        o = rt.add_salt_pepper_noise()({"image": img[0].copy(), "label": lab[0], "img_name": "s"})
        record("salt_pepper.%d" % s_, {"image": o["image"]}, dict(seed=45, py_seed=s_, np_seed=s_), True)
    for s_ in branch_seeds(lambda r: r > 0.5, 3) + branch_seeds(lambda r: r <= 0.5):
        random.seed(s_)
        o = rt.adjust_light()({"image": img[0].copy(), "label": lab[0], "img_name": "s"})
        record("adjust_light.%d" % s_, {"image": np.asarray(o["image"])}, dict(seed=45, py_seed=s_), True)
    for s_ in branch_seeds(lambda r: r <= 0.5, 2) + branch_seeds(lambda r: r > 0.5):
        random.seed(s_); np.random.seed(s_)
        im = img[0].copy()
        o = rt.eraser()({"image": im, "label": lab[0], "img_name": "s"})
        record("eraser.%d" % s_, {"image": np.asarray(o["image"])}, dict(seed=45, py_seed=s_, np_seed=s_), True)

    # ---- the whole array part of the training chain (train_use_fix_initial.py:153-159) with one seed per sample: pins the ORDER
    # and NUMBER of draws each transform takes from `random` and `np.random`
    img, lab = fundus_u8(8, 96, 96, 46)
    fired = set()
    for b in range(8):
        np.random.RandomState = lambda s=None, _k=2000 + b: RS(_k)
        try:
            random.seed(300 + b); np.random.seed(300 + b)
            s = {"image": Image.fromarray(img[b]), "label": Image.fromarray(lab[b]), "img_name": "s"}
            trail = []
            for t in (rt.elastic_transform(), rt.add_salt_pepper_noise(), rt.adjust_light(), rt.eraser()):
                before = np.array(s["image"]).copy()
                s = t(s)
                trail.append(int(not np.array_equal(before, np.array(s["image"]))))
            s = rt.ToTensor()(rt.Normalize_tf()(s))
        finally:
            np.random.RandomState = RS
        fired |= {(i, f) for i, f in enumerate(trail)}
        record("chain.%d" % b, tensors(s), dict(B=8, H=96, W=96, seed=46, b=b, py_seed=300 + b, np_seed=300 + b,
                                                noise_seed=2000 + b, fired=trail), b == 0)
    assert len(fired) == 8, "the chain seeds must exercise both branches of all four transforms: %s" % sorted(fired)
    with open(os.path.join(HERE, "input_pipeline.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "input_pipeline_small.npz"), **small)
    print("input pipeline: %d cases, %d arrays kept in full" % (len(meta), len(small)))


if __name__ == "__main__":
    import tempfile
    install_reference()
    which = sys.argv[1:] or ["manifest", "fwd64", "fwd512", "proto", "metrics", "tb", "tb256", "tp", "rn", "tn", "input", "frozen", "rntn", "frozentn"]
    if "frozen" in which:
        make_frozen()
        if which == ["frozen"]:
            raise SystemExit(0)
    if "frozentn" in which:
        make_frozen(B=4, transnorm=True)
        if which == ["frozentn"]:
            raise SystemExit(0)
    if "input" in which:
        make_input_pipeline()
        if which == ["input"]:
            raise SystemExit(0)
    if "tn" in which:        # TransNorm variant (--use_TN): B = 4 so each domain half has 2 images
        mt = make_manifest("mobilenet", "manifest_tn.json", sync_bn=False)
        make_forward(mt, 4, 64, "tn_64")
        del mt
        if which == ["tn"]:
            raise SystemExit(0)
    if "rntn" in which:      # ResNet-101 with TransNorm layers (deeplabv3.py:17-23 allows the combination), B = 4
        mrt = make_manifest("resnet", "manifest_resnet_tn.json", sync_bn=False)
        make_forward(mrt, 4, 128, "resnet_tn_128", do_eval=False)
        del mrt
        if which == ["rntn"]:
            raise SystemExit(0)
    if "rn" in which:        # ResNet-101 variant (BASELINE.json configs[4]); pretrained fetch patched out (8c)
        mr = make_manifest("resnet", "manifest_resnet.json")
        make_forward(mr, 2, 128, "resnet_128")
        make_forward(mr, 2, 256, "resnet_256")     # 16x16 maps in layer3/4: BN statistics over 512 samples
        del mr
        if which == ["rn"]:
            raise SystemExit(0)
    m = make_manifest() if "manifest" in which else ref_model()
    if "fwd64" in which:
        make_forward(m, 2, 64, "64")
    if "fwd512" in which:
        make_forward(m, 2, 512, "512")
    if "proto" in which:
        make_proto()
    if "metrics" in which:
        make_metrics()
    with tempfile.TemporaryDirectory() as tmp:
        if "tb" in which:
            make_trainer_baseline(tmp)
        if "tb256" in which:
            make_trainer_baseline(tmp, S=256, B=8, nS=4, nV=2, epochs=1, seedS=520, seedV=720, fname="trainer_baseline_256.json")
        if "tp" in which:
            make_trainer_proto(tmp)
    print("golden fixtures written to", HERE)
