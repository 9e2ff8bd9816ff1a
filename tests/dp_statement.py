"""Shared by the data-parallel tests of Trainer_prototype_full (gloo on CPU: tests/test_parallel_cpu.py; real devices:
tests/test_parallel_gpu.py): the miniature of BASELINE.json configs[3] - two ranks x (B source + B target) - and the hand-made
SINGLE-PROCESS statement of the same update on the concatenated batch, evaluated with the oracle on the host:
  * both ranks end with IDENTICAL parameters (generator and both discriminators), although rank 1 started perturbed;
  * the centroids on both ranks are the centroids of the CONCATENATED batch;
  * the update equals one SGD step on  mean_r(seg_r + adv_r) + pro_weight * intra(global centroids).
Test infrastructure (imports oracle/)."""
import torch
import torch.nn.functional as F

PF = dict(S=128, B=2, lr=0.05, lr_d=0.01, pro_weight=0.1, loaderS_seed=80, loaderT_seed=90, drop_seed=500)


def proto_setup(seed_dis=1338):
    """The oracle generator (seed 1337 parameters of the product's DeepLab) and the oracle discriminators."""
    from oracle import deeplab_ref
    from oracle.gan_ref import BoundaryDiscriminator, UncertaintyDiscriminator
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    torch.manual_seed(1337)
    m = deeplab_ref.OracleDeepLab(DeepLab(num_classes=2, backbone="mobilenet", output_stride=16).state_dict())
    torch.manual_seed(seed_dis)
    d1, d2 = BoundaryDiscriminator(), UncertaintyDiscriminator()
    return m, d1, d2


def rank_record(tr, m, d1, d2, vals):
    """What a rank saves after its train_step (CPU tensors)."""
    c = lambda t: t.detach().cpu().clone()
    # (the oracle module names its parameters with "__" for ".": one spelling for both kinds of generator)
    return {"gen": {k.replace("__", "."): c(v) for k, v in m.named_parameters()}, "dis": {k: c(v) for k, v in d1.named_parameters()},
            "dis2": {k: c(v) for k, v in d2.named_parameters()}, "src": [c(t) for t in tr.src_centroids],
            "tgt": [c(t) for t in tr.tgt_centroids], "vals": vals}


def check_ranks_agree(r0, r1):
    for grp in ("gen", "dis", "dis2"):
        for k in r0[grp]:
            assert torch.equal(r0[grp][k], r1[grp][k]), "ranks diverged on %s %s" % (grp, k)
    for a, b in zip(r0["src"] + r0["tgt"], r1["src"] + r1["tgt"]):
        assert torch.equal(a, b), "the ranks hold different centroids"
    assert r0["vals"][4:] == r1["vals"][4:], "intra / inter are global quantities"


def _draw_masks(B, S):
    """The keep-masks nn.Dropout draws for one training forward on B x 3 x S x S, from the global CPU generator in the reference's
    order (float32 draws whatever the statement's dtype: the stream the product ranks replay)."""
    from oracle import deeplab_ref
    ps = dict(deeplab_ref.DROPOUT_SITES)
    return {n: (F.dropout(torch.ones(shp), ps[n], True) != 0).to(torch.uint8)
            for n, shp in deeplab_ref.dropout_mask_shapes(B, S, S).items()}


def global_statement(dtype=torch.float32):
    """The step as ONE process over both ranks' batches, on the host in `dtype`.  Returns the global centroids, intra / inter and
    the parameters after one SGD step on  mean_r(seg_r + adv_r) + pro_weight * intra(global centroids)  (generator) and on
    mean_r(D_same_r + D_diff_r) (discriminators), plus the initial parameters."""
    from make_golden_inputs import synth_loader
    from oracle import proto_ref, step_ref
    c = PF
    m, d1, d2 = proto_setup()
    m, d1, d2 = m.to(dtype), d1.to(dtype), d2.to(dtype)
    m.train(); d1.train(); d2.train()
    name = lambda k: k.replace("__", ".")
    init = {g: {name(k): v.detach().clone() for k, v in mod.named_parameters()} for g, mod in (("gen", m), ("dis", d1), ("dis2", d2))}
    loaderS, loaderT = synth_loader(2, c["B"], c["S"], c["loaderS_seed"]), synth_loader(2, c["B"], c["S"], c["loaderT_seed"])
    cast = lambda t: t.to(dtype)

    def fwd(x):
        m.masks = _draw_masks(x.shape[0], x.shape[2])
        return m(cast(x))
    per = []
    for r in range(2):
        torch.manual_seed(c["drop_seed"] + r)
        sS, sT = loaderS[r], loaderT[r]
        oT, bT, _, _, xt, oT_before, _ = fwd(sT["image"])
        oS, bS, _, _, xs, _, _ = fwd(sS["image"])
        rep = sT["image"].repeat(2, 1, 1, 1)
        with torch.no_grad():
            preds = torch.cat([fwd(rep)[0] for _ in range(4)], 0)
        per.append(dict(oT=oT, bT=bT, xt=xt, oTb=oT_before, oS=oS, bS=bS, xs=xs, preds=preds, sS=sS))
    m.masks = None
    # global source centroids: labels nearest-resized, features of both ranks concatenated
    lab = torch.cat([F.interpolate(cast(p["sS"]["map"]).clone(), size=p["xs"].shape[2:], mode="nearest") for p in per])
    src = proto_ref.gen_prototype(lab, torch.cat([p["xs"] for p in per]))
    # global target centroids: the retrify weights are per pixel, so the concatenated batch gives the concatenated weights
    T = 8
    preds_cat = torch.cat([torch.cat([p["preds"][i * c["B"]:(i + 1) * c["B"]] for p in per]) for i in range(T)])
    tgt = proto_ref.gen_prototype_retrify(torch.cat([p["oTb"] for p in per]), torch.cat([p["xt"] for p in per]), preds_cat, T,
                                          2 * c["B"])[:4]
    intra, inter = proto_ref.alignment_losses(src, tgt)
    total = c["pro_weight"] * intra
    for p in per:
        adv = 0.01 * (step_ref._adv(d2(step_ref._uncertainty(p["oT"])), 1) + step_ref._adv(d1(torch.sigmoid(p["bT"])), 1))
        total = total + 0.5 * (step_ref.seg_loss(p["oS"], p["bS"], cast(p["sS"]["map"]), cast(p["sS"]["boundary"])) + adv)
    grads = torch.autograd.grad(total, list(m.parameters()), allow_unused=True)
    out = {"init": init, "src": [t.detach() for t in src], "tgt": [t.detach() for t in tgt], "intra": intra.item(), "inter": inter.item(),
           "gen": {k: (v0 if g is None else v0 - c["lr"] * g) for (k, v0), g in zip(init["gen"].items(), grads)}}
    # discriminators: mean over ranks of (D_same_r + D_diff_r) on detached generator outputs
    dl = 0.0
    for p in per:
        oS, bS, oT, bT = (p[k].detach() for k in ("oS", "bS", "oT", "bT"))
        dl = dl + 0.5 * (step_ref._adv(d2(step_ref._uncertainty(oS)), 1) + step_ref._adv(d1(torch.sigmoid(bS)), 1) +
                         step_ref._adv(d2(step_ref._uncertainty(oT)), 0) + step_ref._adv(d1(torch.sigmoid(bT)), 0))
    for grp, mod in (("dis", d1), ("dis2", d2)):
        gs = torch.autograd.grad(dl, list(mod.parameters()), retain_graph=True)
        out[grp] = {k: v0 - c["lr_d"] * g for (k, v0), g in zip(init[grp].items(), gs)}
    return out


def check_global_statement(r0):
    """CPU ranks on the oracle arithmetic: the update equals the fp32 statement to reassociation noise."""
    st = global_statement(torch.float32)
    for got, want in zip(r0["src"] + r0["tgt"], st["src"] + st["tgt"]):
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-6), "centroids are not those of the global batch"
    assert abs(r0["vals"][4] - st["intra"]) < 1e-4 * abs(st["intra"]) and abs(r0["vals"][5] - st["inter"]) < 1e-4 * abs(st["inter"])
    for k, want in st["gen"].items():
        assert torch.allclose(r0["gen"][k], want, rtol=2e-4, atol=2e-6), (k, (r0["gen"][k] - want).abs().max().item())
    for grp in ("dis", "dis2"):
        for k, want in st[grp].items():
            assert torch.allclose(r0[grp][k], want, rtol=2e-4, atol=1e-7), (grp, k)


def check_against_fp64_statement(r0):
    """Device ranks on the HIP kernels: every tensor's UPDATE is held to the fp64 statement with the criterion of the generator
    parity tests (tests/model_cases.py::grads_ok): within 10x the fp32 statement's own distance to fp64 (+ 2e-3), since several
    BatchNorm-affine gradients are near-cancelling sums that the fp32 oracle itself gets 1e-2 wrong.  Returns (geometric-mean
    ratio, worst tensors)."""
    import model_cases
    s32, s64 = global_statement(torch.float32), global_statement(torch.float64)
    for got, want in zip(r0["src"] + r0["tgt"], s64["src"] + s64["tgt"]):
        assert model_cases.rel(got.reshape(-1), want.reshape(-1)) < 2e-3, "centroids are not those of the global batch"
    assert abs(r0["vals"][4] - s64["intra"]) < 5e-3 * abs(s64["intra"]) and abs(r0["vals"][5] - s64["inter"]) < 5e-3 * abs(s64["inter"])
    errs = {}
    for k, w64 in s64["gen"].items():
        u64 = w64 - s64["init"]["gen"][k]
        if float(u64.abs().max()) == 0.0:
            assert torch.equal(r0["gen"][k], s32["init"]["gen"][k]), k       # no gradient reaches it: unchanged
            continue
        u_hip = r0["gen"][k].double() - s64["init"]["gen"][k]
        u32 = (s32["gen"][k] - s32["init"]["gen"][k]).double()
        errs[k] = (model_cases.l2rel(u_hip, u64), model_cases.l2rel(u32, u64))
    bad, gmean = model_cases.grads_ok(errs)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:5]
    assert gmean < 4.0, gmean
    for grp in ("dis", "dis2"):
        for k, w64 in s64[grp].items():
            u64 = w64 - s64["init"][grp][k]
            e_hip = model_cases.l2rel(r0[grp][k].double() - s64["init"][grp][k], u64)
            e_32 = model_cases.l2rel((s32[grp][k] - s32["init"][grp][k]).double(), u64)
            # (the discriminators see the generator's outputs: the fp32 statement itself is 3-7e-3 from fp64 on these updates)
            assert model_cases.grad_ok(e_hip, e_32), (grp, k, e_hip, e_32)
    return gmean, sorted(errs.items(), key=lambda kv: -kv[1][0])[:3]
