"""Where does the bf16x3 matrix mode leave the pack on the multi-step fixture?  (TEST INFRASTRUCTURE, run on the GPU box.)

tests/golden/trainer_baseline_256.json (BASELINE.json configs[0]: 8 x 256^2, 4 Adam steps + validation, rows written by the
reference's own Trainer_baseline) is the one fixture that carries a trajectory.  This tool replays it - same batches, same
dropout masks (the reference's CPU stream under the fixture's seed), Adam(1e-3, (0.9, 0.99)) written out by hand so that every
arithmetic runs the identical update rule - in five arithmetics:

    fp64      the oracle in float64 on the host                                   (the anchor)
    ref32/1   the oracle in float32, 1 host thread      (= the reference arithmetic, bit-identical to the reference here)
    ref32/N   the oracle in float32, N host threads     (another summation order of the SAME fp32 arithmetic)
    hip/f32   the HIP path, v_mfma_f32_32x32x2_f32 wide tiles
    hip/x3    the HIP path, bf16x3 wide tiles (default mode)

and reports
  A. free-running: train loss per step, validation loss, and per step the distance of the parameters / running statistics to
     the fp64 trajectory, relative to the fp64 update size so far;
  B. one step from a COMMON state: before step k every arithmetic is reset to the fp64 state after step k-1 (parameters,
     running statistics, Adam moments), takes step k, and its update is compared with fp64's - per-step error without the
     trajectory's own amplification (a stale cache across optimizer steps would show as A >> B on the first steps);
  C. validation on a COMMON state: the eval-mode loss of every arithmetic on the fp64 end state.

    python tests/tools/trajectory_anchor.py [host threads N=16]      -> gpurun_out/trajectory_anchor.txt"""
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
import torch.nn.functional as F

from make_golden_inputs import synth_loader
from oracle import deeplab_ref, step_ref

LR, B1, B2, EPS = 1e-3, 0.9, 0.99, 1e-8


def adam(params, grads, st, step):
    """torch.optim.Adam's rule, written out (same expression order in every arithmetic)."""
    bc1, bc2 = 1.0 - B1 ** step, 1.0 - B2 ** step
    for k, p in params.items():
        g = grads[k]
        if g is None:
            continue
        m, v = st["m"][k], st["v"][k]
        m.mul_(B1).add_(g, alpha=1.0 - B1)
        v.mul_(B2).addcmul_(g, g, value=1.0 - B2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(EPS)
        p.addcdiv_(m, denom, value=-LR / bc1)


class OracleRun:
    def __init__(self, init, dtype, threads):
        self.dtype, self.threads = dtype, threads
        self.sd = {k: (v.detach().to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in init.items()}
        self.keys = deeplab_ref.parameter_keys(self.sd)
        self.st = {"m": {k: torch.zeros_like(self.sd[k]) for k in self.keys}, "v": {k: torch.zeros_like(self.sd[k]) for k in self.keys}}

    def load(self, state, st):
        for k, v in state.items():
            self.sd[k] = v.to(self.dtype).clone() if v.is_floating_point() else v.clone()
        self.st = {n: {k: t.to(self.dtype).clone() for k, t in d.items()} for n, d in st.items()}

    def train_step(self, batch, masks, step):
        torch.set_num_threads(self.threads)
        for k in self.keys:
            self.sd[k] = self.sd[k].detach().requires_grad_(True)
        o = deeplab_ref.deeplab_forward(self.sd, batch["image"].to(self.dtype), training=True, masks=masks)
        loss = step_ref.seg_loss(o[0], o[1], batch["map"].to(self.dtype), batch["boundary"].to(self.dtype))
        grads = torch.autograd.grad(loss, [self.sd[k] for k in self.keys], allow_unused=True)
        with torch.no_grad():
            params = {k: self.sd[k].detach() for k in self.keys}
            for k in self.keys:
                self.sd[k] = params[k]
            adam(params, dict(zip(self.keys, grads)), self.st, step)
        return float(loss.detach())

    def validate(self, loader):
        torch.set_num_threads(self.threads)
        tot = 0.0
        with torch.no_grad():
            for s in loader:
                o = deeplab_ref.deeplab_forward(self.sd, s["image"].to(self.dtype), training=False)
                tot += float(F.binary_cross_entropy_with_logits(o[0], s["map"].to(self.dtype)))
        return tot / len(loader)

    def state(self):
        return {k: v.detach().double().clone() if v.is_floating_point() else v.clone() for k, v in self.sd.items()}


class HipRun:
    def __init__(self, mode):
        import model_cases
        from uda_clr_amd import ops
        os.environ["UDA_CLR_MFMA"] = mode
        self.dev = torch.device("cuda:0")
        self.m = model_cases.seeded_model().to(self.dev).train()
        self.ops = ops
        self.live = self.m._flat_state()
        self.keys = [k for k in deeplab_ref.parameter_keys(self.live)]
        self.st = {"m": {k: torch.zeros_like(self.live[k]) for k in self.keys}, "v": {k: torch.zeros_like(self.live[k]) for k in self.keys}}
        x = torch.zeros(2, 3, 64, 64, device=self.dev)
        with torch.no_grad():
            self.m.eval()
            self.m(x)                               # builds the engine under this mode
            self.m.train()
        from uda_clr_amd.kernels import HipKernels
        want = HipKernels.MFMA_F32 if mode == "f32" else HipKernels.MFMA_BF16X3
        assert self.m._engine.K.mfma == want, "engine did not pick up the matrix mode"

    def load(self, state, st):
        with torch.no_grad():
            for k, v in state.items():
                self.live[k].copy_(v.to(self.live[k].dtype))
            for n, d in st.items():
                for k, t in d.items():
                    self.st[n][k].copy_(t.float())
        self.m.note_params_changed()

    def train_step(self, batch, masks, step):
        m, dev = self.m, self.dev
        m.train()
        for p in m.parameters():
            p.grad = None
        m.set_dropout_masks(masks)
        o = m(batch["image"].to(dev))
        loss = self.ops.seg_loss(o[0], o[1], batch["map"].to(dev), batch["boundary"].to(dev))
        loss.backward()
        with torch.no_grad():
            adam({k: self.live[k] for k in self.keys}, {k: self.live[k].grad for k in self.keys}, self.st, step)
        m.note_params_changed()
        return float(loss.detach())

    def validate(self, loader):
        self.m.eval()
        tot = 0.0
        with torch.no_grad():
            for s in loader:
                o = self.m(s["image"].to(self.dev))
                tot += float(F.binary_cross_entropy_with_logits(o[0], s["map"].to(self.dev)))
        self.m.train()
        return tot / len(loader)

    def state(self):
        return {k: (v.detach().double().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()) for k, v in self.live.items()}


def section(k):
    return "backbone" if k.startswith("backbone") else ("aspp" if k.startswith("aspp") else "decoder")


def distance(state, anchor, base, keys):
    """||state - anchor|| / ||anchor - base|| over `keys`, total and per section."""
    num, den = {}, {}
    for k in keys:
        s = section(k)
        num[s] = num.get(s, 0.0) + float((state[k] - anchor[k]).pow(2).sum())
        den[s] = den.get(s, 0.0) + float((anchor[k] - base[k]).pow(2).sum())
    tot = math.sqrt(sum(num.values()) / max(sum(den.values()), 1e-300))
    return tot, {s: math.sqrt(num[s] / max(den[s], 1e-300)) for s in num}


def main():
    nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    os.makedirs("gpurun_out", exist_ok=True)
    out = open("gpurun_out/trajectory_anchor.txt", "w")

    def emit(s=""):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    z = json.load(open(os.path.join(TESTS, "golden", "trainer_baseline_256.json")))
    loaderS = synth_loader(z["n_batches_S"], z["B"], z["S"], z["loaderS_seed"])
    loaderV = synth_loader(z["n_batches_V"], z["B"], z["S"], z["loaderV_seed"])
    ps = dict(deeplab_ref.DROPOUT_SITES)
    torch.manual_seed(z["torch_seed"])
    masks = []
    for _ in loaderS:                     # the reference's nn.Dropout draws, in its order (tests/test_trainers_gpu.py::MaskFeeder)
        masks.append({n: (F.dropout(torch.ones(shp), ps[n], True) != 0).to(torch.uint8)
                      for n, shp in deeplab_ref.dropout_mask_shapes(z["B"], z["S"], z["S"]).items()})
    import model_cases
    init = deeplab_ref.canonical_state(model_cases.seeded_model().state_dict())
    runs = {"fp64": lambda: OracleRun(init, torch.float64, nthreads), "ref32/1": lambda: OracleRun(init, torch.float32, 1),
            "ref32/%d" % nthreads: lambda: OracleRun(init, torch.float32, nthreads),
            "hip/f32": lambda: HipRun("f32"), "hip/x3": lambda: HipRun("bf16x3")}
    if os.environ.get("UDA_ANCHOR_NO_HIP"):          # host-only rehearsal of the tool itself
        runs = {k: v for k, v in runs.items() if not k.startswith("hip")}
    pkeys = deeplab_ref.parameter_keys(init)
    rkeys = [k for k in init if k not in pkeys and init[k].is_floating_point()]
    base = {k: v.double() for k, v in init.items() if v.is_floating_point()}
    nstep = len(loaderS)

    # ---------------------------------------------------------------- A. free-running
    emit("A. free-running trajectories (fixture: train %s, val %.6f)" % (["%.6f" % v for v in z["train_loss"]], z["val"][0][0]))
    traj, anchors, adam_states = {}, [], []
    for name, mk in runs.items():
        r = mk()
        losses, states = [], []
        for i, s in enumerate(loaderS):
            losses.append(r.train_step(s, masks[i], i + 1))
            states.append(r.state())
            if name == "fp64":
                adam_states.append({n: {k: t.detach().clone() for k, t in d.items()} for n, d in r.st.items()})
        val = r.validate(loaderV)
        traj[name] = (losses, states, val)
        if name == "fp64":
            anchors = states
        emit("  %-9s train %s  val %.6f" % (name, " ".join("%.6f" % v for v in losses), val))
        del r
    v64 = traj["fp64"][2]
    emit("")
    emit("  validation loss relative to fp64 (%.6f):  " % v64 + "  ".join("%s %+.3f%%" % (n, 100 * (t[2] / v64 - 1)) for n, t in traj.items() if n != "fp64") +
         "  fixture %+.3f%%" % (100 * (z["val"][0][0] / v64 - 1)))
    emit("  train loss relative to fp64, per step:")
    for n, t in traj.items():
        if n != "fp64":
            emit("    %-9s %s" % (n, " ".join("%+.2e" % (a / b - 1) for a, b in zip(t[0], traj["fp64"][0]))))
    emit("  parameter distance to the fp64 trajectory / fp64 update size so far (total | backbone aspp decoder), running statistics beside it:")
    for n, t in traj.items():
        if n == "fp64":
            continue
        for i in range(nstep):
            tot, per = distance(t[1][i], anchors[i], base, pkeys)
            rs, _ = distance(t[1][i], anchors[i], base, rkeys)
            emit("    %-9s step %d  %.3e | %.3e %.3e %.3e   running %.3e" % (n, i + 1, tot, per["backbone"], per["aspp"], per["decoder"], rs))

    # ---------------------------------------------------------------- B. one step from the common (fp64) state
    emit("")
    emit("B. one step from the fp64 state after step k-1: ||update - fp64 update|| / ||fp64 update|| (total | backbone aspp decoder), loss rel. to fp64")
    zero_st = {"m": {k: torch.zeros_like(base[k]) for k in pkeys}, "v": {k: torch.zeros_like(base[k]) for k in pkeys}}
    init64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in init.items()}
    for name, mk in runs.items():
        if name == "fp64":
            continue
        r = mk()
        for i, s in enumerate(loaderS):
            start = init64 if i == 0 else anchors[i - 1]
            r.load(start, zero_st if i == 0 else adam_states[i - 1])
            loss = r.train_step(s, masks[i], i + 1)
            st = r.state()
            startd = {k: v for k, v in start.items() if v.is_floating_point()}
            tot, per = distance(st, anchors[i], startd, pkeys)
            rs, _ = distance(st, anchors[i], startd, rkeys)
            emit("    %-9s step %d  %.3e | %.3e %.3e %.3e   running %.3e   loss %+.2e" %
                 (name, i + 1, tot, per["backbone"], per["aspp"], per["decoder"], rs, loss / traj["fp64"][0][i] - 1))
        # ------------------------------------------------------------ C. validation on the common end state
        r.load(anchors[-1], adam_states[-1])
        v = r.validate(loaderV)
        emit("  C. %-9s validation on the fp64 end state: %.6f (%+.4f%% of fp64's %.6f)" % (name, v, 100 * (v / v64 - 1), v64))
        del r


if __name__ == "__main__":
    main()
