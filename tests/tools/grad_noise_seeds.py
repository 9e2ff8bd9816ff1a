"""Per-tensor gradient distance to the fp64 oracle over several input seeds, both matrix modes (GPU box): which tensors are
noise-dominated (the fp32 ORACLE itself > 1e-3 from fp64) and how their HIP / fp32-oracle ratio scatters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import model_cases
from uda_clr_amd.kernels import HipKernels
dev = torch.device("cuda:0")
for mode in ("bf16x3", "f32"):
    os.environ["UDA_CLR_MFMA"] = mode
    for seed in (3, 4, 5, 6):
        fwd, grads, stats, _ = model_cases.train_parity(dev, S=64, seed=seed)
        bad, gmean = model_cases.grads_ok(grads)
        noisy = {k: v for k, v in grads.items() if v[1] > 1e-3}
        print("mode %s seed %d: gmean %.3f, %d tensors with fp32-oracle noise > 1e-3; ratio > 10: %s" % (
            mode, seed, gmean, len(noisy), {k: (round(v[0], 4), round(v[1], 4)) for k, v in bad.items()}), flush=True)
        print("   noisy tensors (hip, oracle32): " + ", ".join("%s %.3g/%.3g" % (k.replace("backbone.features.", "f"), v[0], v[1]) for k, v in sorted(noisy.items())), flush=True)
