"""Run every GPU parity case without stopping at failures and print one table
(gpurun_out/report.txt).  Usage on the GPU box:  python tests/gpu_report.py [kernels] [model] [golden]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kernel_cases import CASES
import model_cases


def main():
    what = sys.argv[1:] or ["kernels", "model", "golden"]
    dev = torch.device("cuda:0")
    os.makedirs("gpurun_out", exist_ok=True)
    out = open("gpurun_out/report.txt", "w")

    def emit(s):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    emit("device: %s" % torch.cuda.get_device_name(0))
    nfail = 0
    if "kernels" in what:
        only = os.environ.get("KCASE")
        for name, fn in CASES:
            if only and only not in name:
                continue
            t0 = time.time()
            try:
                err, tol = fn(dev)
                torch.cuda.synchronize()
                ok = err <= tol
                emit("%-48s err %.3e tol %.1e %s (%.2fs)" % (name, err, tol, "ok" if ok else "FAIL", time.time() - t0))
                if not ok:
                    import kernel_cases
                    emit("      detail: %s" % kernel_cases.DETAIL)
                nfail += (not ok)
            except Exception as e:  # noqa: BLE001
                nfail += 1
                emit("%-48s EXC %s: %s" % (name, type(e).__name__, str(e).split("\n")[0][:200]))
                traceback.print_exc()
    if "model" in what:
        try:
            for S in (64, 96):
                e = model_cases.eval_parity(dev, 2, S)
                emit("eval parity %d: %s" % (S, {k: "%.2e" % v for k, v in e.items()}))
            fwd, grads, stats, _ = model_cases.train_parity(dev)
            emit("train fwd parity: %s" % {k: "%.2e" % v for k, v in fwd.items()})
            emit("running stats err: %.2e" % stats)
            bad, gmean = model_cases.grads_ok(grads)
            print('gradient geometric-mean ratio hip/fp32 distance: %.2f' % gmean)
            worst = max(grads.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1e-4))
            emit("grad parity: %d params, %d outside 3x the fp32 noise floor; worst %s err %.2e floor %.2e" %
                 (len(grads), len(bad), worst[0], worst[1][0], worst[1][1]))
            for k, v in list(bad.items())[:40]:
                emit("   BAD %-50s err %.3e floor %.3e" % (k, v[0], v[1]))
            nfail += len(bad)
        except Exception as e:  # noqa: BLE001
            nfail += 1
            emit("model parity EXC %s: %s" % (type(e).__name__, str(e)[:300]))
            traceback.print_exc()
    if "dual" in what:
        try:
            from dual_kernels import DualKernels
            from kernel_spec import SpecKernels
            from uda_clr_amd.engine import GeneratorEngine
            from uda_clr_amd.kernels import HipKernels
            from oracle import deeplab_ref, step_ref
            m = model_cases.seeded_model(perturb=True).train().to(dev)
            dual = DualKernels(HipKernels(), SpecKernels(), log=emit)
            m._engine_override = GeneratorEngine(dual)
            gen = torch.Generator().manual_seed(3)
            B, S = 2, 64
            x = torch.randn(B, 3, S, S, generator=gen)
            tmap = (torch.rand(B, 2, S, S, generator=gen) > 0.5).float()
            tbd = torch.rand(B, 1, S, S, generator=gen)
            masks = deeplab_ref.draw_masks(B, S, S, gen)
            wf = [torch.randn(t, generator=gen) for t in (256, 304, 305, 2, 1)]
            m.set_dropout_masks(masks)
            outs = m(x.to(dev))
            loss = step_ref.seg_loss(outs[0], outs[1], tmap.to(dev), tbd.to(dev))
            for t, w in zip(outs[2:], wf):
                loss = loss + 1e-2 * (t * w.to(dev).view(1, -1, 1, 1)).pow(2).mean()
            loss.backward()
            emit("dual run: %d kernel calls, %d mismatching outputs" % (dual.calls, len(dual.bad)))
        except Exception as e:  # noqa: BLE001
            nfail += 1
            emit("dual EXC %s: %s" % (type(e).__name__, str(e)[:300]))
            traceback.print_exc()
    if "golden" in what:
        for tag in ("64", "512"):
            try:
                e = model_cases.golden_parity(dev, tag)
                emit("golden %s: max err %.3e  %s" % (tag, max(e.values()), {k: "%.1e" % v for k, v in e.items() if v > 1e-4}))
            except Exception as ex:  # noqa: BLE001
                nfail += 1
                emit("golden %s EXC %s: %s" % (tag, type(ex).__name__, str(ex)[:300]))
                traceback.print_exc()
    emit("FAILURES: %d" % nfail)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
