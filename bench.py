#!/usr/bin/env python
"""bench.py - training images/sec of the UDA_CLR per-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic 512x512 inputs resident in HBM:
workload ``source_only`` (BASELINE.json configs[1]): zero_grad -> DeepLabV3+/MobileNetV2 forward
(training-mode BN, device-generated dropout) -> BCE+MSE seg loss -> backward -> Adam step, B=16/GPU
(Trainer_baseline.py:198-243).  Data parallel: one process per GPU, per-rank batch fixed (weak
scaling), one flat RCCL all-reduce of the generator gradients per step; BN statistics stay per rank
as in the reference.  Rank 0 prints ONE JSON line with the metric, the roofline of the dominant
kernel (FP32-MFMA implicit-GEMM 3x3 convolutions, timed live with HIP events on the launch stream)
and the CPU baseline (the oracle restatement on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32


def synth_batch(B, S, seed, device):
    """Seeded synthetic fundus-like batch: image U(-1,1), concentric-ellipse cup/disc map, soft ring
    boundary."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(S).float(), torch.arange(S).float(), indexing="ij")
    maps, bds = [], []
    for _ in range(B):
        cy, cx = (0.4 + 0.2 * torch.rand(2, generator=g)) * S
        a, b = (0.18 + 0.09 * torch.rand(2, generator=g)) * S
        k = 0.4 + 0.3 * torch.rand(1, generator=g)
        r = torch.sqrt(((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2)
        disc, cup = (r <= 1).float(), (r <= k).float()
        ring = torch.exp(-((r - 1) * min(a, b) / 3.0) ** 2) + torch.exp(-((r - k) * min(a, b) / 3.0) ** 2)
        maps.append(torch.stack([cup, disc]))
        bds.append(ring.clamp(0, 1)[None])
    img = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    return img.to(device), torch.stack(maps).to(device), torch.stack(bds).to(device)


class ConvTimer:
    """HIP-event timing of the dominant kernel's launches inside the timed region: wraps
    HipKernels.conv and records an event pair around every 3x3 implicit-GEMM launch (forward and
    input-gradient of the decoder / ASPP 3x3 convolutions - one kernel symbol)."""

    def __init__(self, kernels):
        self.k, self.orig = kernels, kernels.conv
        self.events, self.flops, self.enabled = [], [], False
        kernels.conv = self._conv

    def _conv(self, src, w, ksize, dil, out, bias=None, addend=None, stats=None):
        hot = self.enabled and ksize == 3 and out.shape[1] > 96
        if hot:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.orig(src, w, ksize, dil, out, bias, addend, stats)
        if hot:
            e1.record()
            self.events.append((e0, e1))
            self.flops.append(2.0 * src.P * out.shape[1] * 9 * src.C)      # algorithmic: 2*P*Cout*9*Cin

    def summary(self):
        if not self.events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.events]
        tf = sum(self.flops) / (sum(ms) * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "kernel": "igemm_conv_kernel<2,2,2,2> (3x3 fwd+dgrad, decoder+ASPP)",
                "launches": len(ms), "avg_launch_ms": round(sum(ms) / len(ms), 4),
                "algorithmic_gflop_per_launch": round(sum(self.flops) / len(ms) / 1e9, 2)}


def cpu_baseline(B, S, steps):
    """The oracle restatement of the same step on the host cores (bounded sample)."""
    from oracle import deeplab_ref, step_ref
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    torch.manual_seed(1337)
    sd = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16).state_dict()
    om = deeplab_ref.OracleDeepLab(sd).train()
    opt = torch.optim.Adam(om.parameters(), lr=1e-3, betas=(0.9, 0.99))
    img, tmap, tbd = synth_batch(B, S, 1337, "cpu")
    step_ref.baseline_step(om, opt, img, tmap, tbd)            # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step_ref.baseline_step(om, opt, img, tmap, tbd)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps (median) of the source-only step at B=%d, %dx%d on the host cores" % (steps, B, S, S)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from uda_clr_amd.kernels import load_library
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    from uda_clr_amd.parallel import FlatGradAllReduce
    load_library()
    torch.manual_seed(1337)
    model = DeepLab(num_classes=2, backbone="mobilenet", output_stride=16, sync_bn=True, freeze_bn=False,
                    method="baseline").to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.99))
    bce, mse = torch.nn.BCELoss(), torch.nn.MSELoss()
    img, tmap, tbd = synth_batch(args.batch, args.size, 1337 + rank, dev)
    reducer = FlatGradAllReduce(list(model.parameters())) if world > 1 else None

    def step():
        opt.zero_grad(set_to_none=True)
        oS, bS = model(img)[:2]
        loss = bce(torch.sigmoid(oS), tmap) + mse(torch.sigmoid(bS), tbd)
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_mean()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    timer = ConvTimer(model._engine_for(img).K)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    if rank == 0:
        images = args.batch * world * args.steps
        line = {
            "metric": "training images/sec (512x512, src+tgt) at 1/2/4/8 MI355X; val Dice vs ref",
            "value": round(images / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "source_only: Trainer_baseline step, DeepLabV3+/MobileNetV2 %dx%d bs=%d/GPU "
                                   "(BASELINE.json configs[1]); source images only" % (args.size, args.size, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "final_loss": round(final_loss, 5)},
            "roofline": timer.summary(),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(2, args.size, 5)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
