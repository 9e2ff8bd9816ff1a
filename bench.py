#!/usr/bin/env python
"""bench.py - training images/sec of the UDA_CLR per-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 without a launcher environment it starts its N ranks
                                                              itself - the parent makes no device call, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic 512x512 inputs resident in HBM.
Default workload ``prototype_full`` (BASELINE.json configs[2], the one the metric "src+tgt" is quoted
on): one Trainer_prototype_full iteration with B=16 source + 16 target images per GPU
(Trainer_prototype_full.py:261-517).  ``--workload source_only`` (configs[1]): zero_grad -> generator
forward -> BCE+MSE seg loss -> backward -> Adam (Trainer_baseline.py:198-243).  Data parallel: one process per GPU, per-rank batch fixed (weak
scaling), one flat RCCL all-reduce of the generator gradients per step; BN statistics stay per rank
as in the reference.  Rank 0 prints ONE JSON line with the metric, the roofline of the dominant
kernel (the wide-tile multi-tap implicit GEMM of the 3x3 / 2x2 convolutions, timed live with HIP events on the
launch stream) and the CPU baseline (the oracle restatement on the host cores, bounded sample).

``--mfma`` selects the matrix instructions of that kernel: ``bf16x3`` (default) = fp32 emulated on the bf16 pipe by exact
3-way operand splitting (fp32-level results), ``f32`` = v_mfma_f32_32x32x2_f32.  The line of the default run
also carries a short measurement of the other mode (``other_mfma``), outside the timed region.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md, dense bf16 (v_mfma_f32_32x32x16_bf16)
# bf16x3: six bf16 MFMA products per fp32 product, so the fp32-EQUIVALENT ceiling of that kernel is the bf16 peak / 6
PEAK_BF16X3_EQUIV_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


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


def host_input_leg(tr, args, per_step, world, sync, dist, dev, img, tmap, tbd, imgT):
    """PCIe-inclusive rate, outside the headline's timed region (never `value`): the same step, but every step's batches start in
    pinned HOST memory, as a DataLoader with pin_memory hands them over.  Two hand-over formats:
      u8  - the deferred input tail (UDA_CLR_DEVICE_INPUT: uint8 image + grey mask, 4 B/px; Normalize_tf + ToTensor run on the device),
      f32 - the reference's own collated float batches (image, map, boundary: 24 B/px source, 12 B/px target)."""
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(99)

    def u8_pair():
        im = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).pin_memory()
        lab = torch.full((B, S, S), 255, dtype=torch.uint8)
        lab[:, S // 4:3 * S // 4, S // 4:3 * S // 4] = 128
        lab[:, 3 * S // 8:5 * S // 8, 3 * S // 8:5 * S // 8] = 0
        return {"image_u8": im, "label_u8": lab.pin_memory()}
    feeds = {"u8": (u8_pair(), u8_pair()),
             "f32": ({"image": img.cpu().pin_memory(), "map": tmap.cpu().pin_memory(), "boundary": tbd.cpu().pin_memory()},
                     {"image": imgT.cpu().pin_memory()})}
    out = {}
    for tag, (sS, sT) in feeds.items():
        for _ in range(2):
            tr.train_step(sS, sT)
        sync()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.train_step(sS, sT)
        sync()
        dth = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dth], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dth = t.item()
        out[tag] = {"value": round(per_step * world * 5 / dth, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dth / 5, 3), "steps": 5}
    out["note"] = ("batches start in pinned host memory every step (H2D copy + device decode inside the step; no prefetch overlap "
                   "beyond what the stream order gives) - the PCIe-inclusive rate, reported beside the HBM-resident headline value")
    return out


class ConvTimer:
    """HIP-event timing of the dominant kernel's launches inside the timed region (events on torch's current stream, the stream
    the kernels are launched on).  The dominant kernel is the wide-tile multi-tap implicit GEMM: forward and input-gradient of
    the decoder / ASPP 3x3 convolutions and of the discriminators' 4x4 stride-2 convolutions in their 2x2 space-to-depth form -
    igemm_conv_x3_kernel<3,...> in bf16x3 mode (HipKernels._conv_x3 is that launch alone; the operand-packing passes are separate
    kernels and are NOT inside the bracket), igemm_conv_ws_kernel<3,...> in f32 mode."""

    def __init__(self, kernel_class):
        """Patches the binding CLASS so that every HipKernels instance (generator engine, discriminator engines) is timed."""
        self.kernel_class = kernel_class
        self.orig_conv, self.orig_x3 = kernel_class.conv, kernel_class._conv_x3
        self.events, self.flops, self.bytes, self.enabled = [], [], [], False
        self.outer = []
        self.pending = None
        timer = self

        def conv(inst, src, w, ksize, dil, out, *a, **kw):
            hot = timer.enabled and ksize >= 2 and out.shape[1] > 96 and ksize * ksize * src.C > 192   # the wide-tile multi-tap kernel
            if not hot:
                return timer.orig_conv(inst, src, w, ksize, dil, out, *a, **kw)
            taps = ksize * ksize
            flops = 2.0 * src.P * out.shape[1] * taps * src.C                                        # algorithmic: 2*P*Cout*taps*Cin
            if inst.mfma == inst.MFMA_F32:     # in + out + weights once, fp32
                work = (flops, 4.0 * (src.P * (src.C + out.shape[1]) + out.shape[1] * taps * src.C))
            else:                              # this kernel's operands are the PACKED ones: 3 bf16 per value, channels in 16-blocks
                c16 = (src.C + 15) // 16 * 16
                work = (flops, 6.0 * (src.P * c16 + out.shape[1] * taps * c16) + 4.0 * src.P * out.shape[1])
            if inst.mfma == inst.MFMA_F32:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = timer.orig_conv(inst, src, w, ksize, dil, out, *a, **kw)
                e1.record()
                timer._add(e0, e1, work)
                return r
            timer.pending = work                    # bf16x3: the bracket goes around the GEMM launch inside (after the packing passes)
            o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            o0.record()                             # a second, outer bracket: the same call INCLUDING the operand-packing passes it triggers
            try:
                return timer.orig_conv(inst, src, w, ksize, dil, out, *a, **kw)
            finally:
                o1.record()
                timer.outer.append((o0, o1))
                timer.pending = None

        def conv_x3(inst, a):
            if timer.pending is None:
                return timer.orig_x3(inst, a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer.orig_x3(inst, a)
            e1.record()
            timer._add(e0, e1, timer.pending)
            return r
        kernel_class.conv, kernel_class._conv_x3 = conv, conv_x3

    def _add(self, e0, e1, work):
        self.events.append((e0, e1))
        self.flops.append(work[0])
        self.bytes.append(work[1])

    def restore(self):
        self.kernel_class.conv, self.kernel_class._conv_x3 = self.orig_conv, self.orig_x3

    def summary(self, mode, traffic=None):
        """traffic: (GB per launch, source file) measured by PMC passes of this configuration, or None."""
        if not self.events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.events]
        tf = sum(self.flops) / (sum(ms) * 1e-3) / 1e12
        x3 = mode == "bf16x3"
        peak = PEAK_BF16X3_EQUIV_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
        out = {"bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1),
               "unit": "TFLOP/s" + (" (fp32-equivalent: algorithmic fp32 FLOPs; the kernel issues 6 bf16 MFMA FLOPs per algorithmic "
                                    "FLOP, peak = dense bf16 2500 / 6)" if x3 else ""),
               "frac": round(tf / peak, 4), "traffic": None if traffic is None else traffic[0],
               "traffic_source": None if traffic is None else traffic[1],
               "algorithmic_gb_per_launch": round(sum(self.bytes) / len(ms) / 1e9, 4),
               "kernel": ("igemm_conv_x3_kernel<3,*> (bf16x3" if x3 else "igemm_conv_ws_kernel<3,*> (fp32 MFMA") +
                         " multi-tap implicit GEMM: forward + input-gradient of the decoder / ASPP 3x3 convs and of the "
                         "discriminators' 4x4 s2 convs as 2x2 space-to-depth convs)",
               "launches": len(ms), "avg_launch_ms": round(sum(ms) / len(ms), 4),
               "algorithmic_gflop_per_launch": round(sum(self.flops) / len(ms) / 1e9, 2)}
        if x3:
            out["executed_bf16_tflops"] = round(6.0 * tf, 1)
            out["frac_of_f32_mfma_peak"] = round(tf / PEAK_F32_MFMA_TFLOPS, 3)
            if self.outer:
                # the same calls with their operand-packing passes inside the bracket (x3_pack_kernel of the activation / gradient
                # operand when no earlier consumer packed it, and of the weights once per step): what the convolution costs end to end
                mo = sum(a.elapsed_time(b) for a, b in self.outer)
                tfo = sum(self.flops) / (mo * 1e-3) / 1e12
                out["incl_packing"] = {"achieved": round(tfo, 2), "frac": round(tfo / peak, 4),
                                       "avg_call_ms": round(mo / len(self.outer), 4),
                                       "note": "`achieved` / `frac` above bracket the GEMM launches alone (main launch + K-split tail + its reduce: the "
                                               "kernel's own roofline); this figure brackets the whole binding call incl. the packing passes it triggers"}
        return out


def measured_traffic(args, mode):
    """(GB of HBM traffic per launch of the dominant kernel, file it comes from) out of the committed PMC passes of THIS
    configuration and matrix mode (profiles/r<round>_*_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE per launch, profiles/pmc_traffic.py;
    the newest round that holds the file); None for any other configuration.  It is NOT measured by this run - counters cannot be
    collected inside the timed region."""
    import glob
    kern = "x3" if mode == "bf16x3" else "ws"
    if (args.workload, args.backbone, args.size, args.use_tn) == ("prototype_full", "mobilenet", 512, False):
        pat = "r[0-9][0-9]_igemm_conv_%s_b%d_traffic.json" % (kern, args.batch)
    elif (args.workload, args.backbone, args.size, args.use_tn) == ("source_only", "resnet", 512, False):
        pat = "r[0-9][0-9]_resnet101_igemm_conv_%s_b%d_traffic.json" % (kern, args.batch)
    else:
        return None
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pat)))
    if not found:
        return None
    with open(found[-1]) as f:
        return round(json.load(f)["hbm_bytes_per_launch"] / 1e9, 4), "profiles/" + os.path.basename(found[-1])


def cpu_baseline(workload, B, S, backbone="mobilenet"):
    """The oracle restatement of the same step on the host cores, a bounded sample: one warm-up step, one timed step at each of
    8 / 16 / 32 threads, then two more at the fastest count; value = median of the three steps at that count (all cores is NOT
    the fastest for this network: round 1 reported 0.149 img/s on 128 threads against 0.295 on 8).  About 40 s of CPU work."""
    from oracle import deeplab_ref, step_ref
    from oracle.gan_ref import BoundaryDiscriminator, UncertaintyDiscriminator
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    torch.manual_seed(1337)
    sd = DeepLab(num_classes=2, backbone=backbone, output_stride=16).state_dict()
    om = deeplab_ref.OracleDeepLab(sd).train()
    img, tmap, tbd = synth_batch(B, S, 1337, "cpu")
    imgT = synth_batch(B, S, 4242, "cpu")[0]
    if workload == "source_only":
        opt = torch.optim.Adam(om.parameters(), lr=1e-3, betas=(0.9, 0.99))
        run = lambda: step_ref.baseline_step(om, opt, img, tmap, tbd)
        n_img = B
    else:
        d1, d2 = BoundaryDiscriminator().train(), UncertaintyDiscriminator().train()
        og, od, od2 = step_ref.make_optimizers(om, d1, d2)
        stepper = step_ref.PrototypeFullStep(om, d1, d2, og, od, od2)
        run = lambda: stepper(img, tmap, tbd, imgT)
        n_img = 2 * B
    ncpu = os.cpu_count() or 8
    keep = torch.get_num_threads()
    probes = {}
    for nt in sorted({min(n, ncpu) for n in (8, 16, 32)}):       # all cores of a 128+-core host oversubscribe this network's small layers
        torch.set_num_threads(nt)
        if not probes:
            run()                                   # warm-up (allocator, first touch)
        t0 = time.perf_counter()
        run()
        probes[nt] = time.perf_counter() - t0
    best = min(probes, key=probes.get)
    torch.set_num_threads(best)
    ts = [probes[best]]
    while len(ts) < 3:
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    torch.set_num_threads(keep)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(n_img / med, 3), "unit": "images/sec", "cores": best, "kind": "port",
            "sample": "%d timed step(s) (median) of the %s step at B=%d%s, %dx%d, oracle restatement on %d of the %d host cores; "
                      "probe seconds per step by thread count: %s"
                      % (len(ts), workload, B, "" if workload == "source_only" else "+%d" % B, S, S, best, ncpu,
                         {k: round(v, 2) for k, v in probes.items()})}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher environment: become the launcher.  Starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>` as a CHILD
    process (one rank per GPU over RCCL), relays rank 0's JSON line on stdout (everything else the ranks print goes to stderr) and
    returns the child's exit code; a run without a JSON line is a failure.  The parent never touches the GPU: no torch.cuda call,
    no HIP call - it only waits (a process that has initialised the GPU must not exec / fork GPU work on this pool)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank run exited with code %d\n" % (n, rc))
        return rc
    if line is None:
        sys.stderr.write("bench.py: the %d-rank run printed no result line\n" % n)
        return 1
    print(line, flush=True)
    return 0


def stub_worker(args):
    """Rehearsal of the launch + timing protocol WITHOUT a device (tests/test_bench_launcher.py): gloo ranks, a stub step (one small
    all-reduce), the same barrier / max-over-ranks / one-JSON-line contract.  `data` says "stub"; it measures nothing."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus disagree" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
    if os.environ.get("UDA_CLR_STUB_FAIL_RANK") == str(rank):       # a rank that dies: the launcher must hand the failure on
        raise SystemExit(3)
    x = torch.ones(1024)

    def step():
        y = x * 2.0
        if world > 1:
            dist.all_reduce(y)
        return y
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert float(y[0]) == 2.0 * world
    if rank == 0:
        print(json.dumps({"metric": "stub step (launcher rehearsal, no device)", "value": round(world * args.steps / dt, 2), "unit": "steps/sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub",
                          "config": {"workload": "stub", "parallelism": "dp%d" % world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="source (and target) images per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--workload", choices=("prototype_full", "source_only"), default="prototype_full")
    ap.add_argument("--backbone", choices=("mobilenet", "resnet"), default="mobilenet",
                    help="resnet = the ResNet-101 variant of BASELINE.json configs[4] (quoted at --batch 8)")
    ap.add_argument("--use-tn", action="store_true",
                    help="the --use_TN model of train_use_fix_initial.py:98-100,180-181 (TransNorm layers; not a BASELINE.json config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mfma", choices=("bf16x3", "f32"), default=None,
                    help="matrix instructions of the wide conv tiles (default: UDA_CLR_MFMA or bf16x3)")
    ap.add_argument("--no-other-mfma", action="store_true", help="skip the short measurement of the other matrix mode")
    ap.add_argument("--no-host-input", action="store_true", help="skip the PCIe-inclusive measurement (batches handed over from pinned host memory)")
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)   # launcher / timing-protocol rehearsal on CPU + gloo (tests)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This parent makes NO device call.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.stub_step:
        return stub_worker(args)
    if args.mfma:
        os.environ["UDA_CLR_MFMA"] = args.mfma
    mode = os.environ.get("UDA_CLR_MFMA", "bf16x3").lower()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus disagree" % (args.gpus, world))
    if os.environ.get("UDA_CLR_SHARE_GPU"):      # rehearsal of the multi-process path on a one-GPU box (with the gloo backend)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("UDA_CLR_DIST_BACKEND", "nccl")       # nccl = RCCL over xGMI; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "uda_clr_amd", "lib", "libuda_clr_hip.so")):
        if rank == 0:                                   # a checkout without the built library: compile it in-tree first
            import __graft_entry__
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()
    from uda_clr_amd.kernels import load_library
    from uda_clr_amd.networks.GAN import BoundaryDiscriminator, UncertaintyDiscriminator
    from uda_clr_amd.networks.deeplabv3 import DeepLab
    from uda_clr_amd.train_process import Trainer_baseline, Trainer_prototype_full
    load_library()
    torch.manual_seed(1337)
    model = DeepLab(num_classes=2, backbone=args.backbone, output_stride=16, sync_bn=not args.use_tn, freeze_bn=False,
                    method=args.workload).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.99))
    img, tmap, tbd = synth_batch(args.batch, args.size, 1337 + rank, dev)
    imgT = synth_batch(args.batch, args.size, 4242 + rank, dev)[0]
    sampleS = {"image": img, "map": tmap, "boundary": tbd}
    sampleT = {"image": imgT}
    out = os.path.join("/tmp", "uda_bench_%d" % os.getpid())
    if args.workload == "source_only":
        tr = Trainer_baseline.Trainer(cuda=True, model_gen=model, optimizer_gen=opt, val_loader=[], domain_loaderS=[],
                                      domain_loaderT=[], out=out, max_epoch=1, batch_size=args.batch, warmup_epoch=-1)

        def step():
            # Trainer_baseline.train_epoch body for one device-resident batch
            opt.zero_grad(set_to_none=True)
            oS, bS = model(img)[:2]
            loss = tr.ops.seg_loss(oS, bS, tmap, tbd)
            loss.backward()
            if tr._reducer is not None:
                tr._reducer.all_reduce_mean()
            opt.step()
            return loss
        per_step = args.batch
    else:
        d1, d2 = BoundaryDiscriminator().to(dev).train(), UncertaintyDiscriminator().to(dev).train()
        od = torch.optim.SGD(d1.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
        od2 = torch.optim.SGD(d2.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
        tr = Trainer_prototype_full.Trainer(
            cuda=True, model_gen=model, model_dis=d1, model_uncertainty_dis=d2, optimizer_gen=opt, optimizer_dis=od,
            optimizer_uncertainty_dis=od2, val_loader=[], domain_loaderS=[], domain_loaderT=[], out=out, max_epoch=1,
            use_global=True, use_pid=True, retrify_pesudo=True, global_pro_weight=0.9, pro_weight=0.1, batch_size=args.batch,
            warmup_epoch=-1)
        tr.epoch = 0

        def step():
            return tr.train_step(sampleS, sampleT)
        per_step = 2 * args.batch          # source + target images (the metric counts both)

    for _ in range(args.warmup):
        step()
    from uda_clr_amd.kernels import HipKernels
    timer = ConvTimer(HipKernels)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    peak_gib = torch.cuda.max_memory_allocated(dev) / 2.0 ** 30
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    losses = [round(float(v), 5) for v in (last if isinstance(last, list) else [last.item()])]
    # the other matrix mode, outside the timed region: same step, 2 warm-up + 5 timed steps (all ranks take part)
    other = None
    timer.restore()
    host = None
    if args.workload == "prototype_full" and not args.no_host_input:
        host = host_input_leg(tr, args, per_step, world, sync, dist, dev, img, tmap, tbd, imgT)
    if not args.no_other_mfma:
        om = "f32" if mode == "bf16x3" else "bf16x3"
        insts = [o for o in __import__("gc").get_objects() if isinstance(o, HipKernels)]
        for o in insts:
            o.mfma = o.MFMA_F32 if om == "f32" else o.MFMA_BF16X3
        for _ in range(2):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        sync()
        dto = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([dto], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dto = t.item()
        other = {"mfma": om, "value": round(per_step * world * 5 / dto, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dto / 5, 3),
                 "steps": 5}
    if rank == 0:
        images = per_step * world * args.steps
        desc = {"source_only": "source_only: Trainer_baseline step (fwd, BCE+MSE, bwd, Adam), DeepLabV3+/MobileNetV2 %dx%d "
                               "bs=%d/GPU (BASELINE.json configs[1]); source images only",
                "prototype_full": "prototype_full: Trainer_prototype_full step (T+S generator fwd/bwd, seg loss, source/target "
                                  "prototypes with 4 MC-dropout passes (T=8), retrified pseudo labels, EMA, alignment loss, "
                                  "adversarial G/D steps), DeepLabV3+/MobileNetV2 %dx%d bs=%d src + %d tgt per GPU "
                                  "(BASELINE.json configs[2]; images/sec counts src+tgt)"}[args.workload]
        fmt = (args.size, args.size, args.batch) + ((args.batch,) if args.workload == "prototype_full" else ())
        if args.backbone == "resnet":
            desc = desc.replace("DeepLabV3+/MobileNetV2", "DeepLabV3+/ResNet-101").replace("configs[1]", "configs[4] shape, 1 GPU").replace(
                "configs[2]", "configs[4] shape, 1 GPU")
        if args.use_tn:
            desc = desc.replace("DeepLabV3+/MobileNetV2", "DeepLabV3+/MobileNetV2 with TransNorm (--use_TN: per-domain-half launches)")
        line = {
            "metric": "training images/sec (512x512, src+tgt) at 1/2/4/8 MI355X; val Dice vs ref",
            "value": round(images / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc % fmt, "global_batch": per_step * world, "parallelism": "dp%d" % world,
                       "last_step_losses": losses,
                       "mfma": ("bf16x3: inputs, outputs, accumulators and statistics fp32/fp64; the wide conv tiles multiply on the "
                                "bf16 matrix pipe with every fp32 operand split exactly into three bf16 pieces (six piece products, "
                                "fp32 accumulate) - fp32-level results, error against float64 measured equal to the fp32-MFMA "
                                "kernel's (DESIGN.md 3d, tests/bench_x3.py); the exact fp32-MFMA path is `--mfma f32`, measured "
                                "in `other_mfma`") if mode == "bf16x3" else "f32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain)"},
            "roofline": timer.summary(mode, measured_traffic(args, mode)),
        }
        # step-level figures of SURVEY.md 8(d): algorithmic GFLOP / GB per counted image (reference algorithm,
        # i.e. the 4 MC passes counted as full forwards although the fast path recomputes only their stochastic tail)
        gflop = {("mobilenet", "source_only"): 158.2, ("mobilenet", "prototype_full"): 478.0,
                 ("resnet", "source_only"): 531.0}.get((args.backbone, args.workload))
        # FLOPs the product actually executes per counted image: generator pass 2 * (5.225 backbone/ASPP/1x1 +
        # 2.416 conv0 as low-level 3x3 + tap GEMM + 9.664 conv4) GMAC forward, x3 with the backward = 103.8 GFLOP; an MC image-pass
        # 2 * (0.604 tap GEMM + 9.664 conv4) = 20.6 GFLOP; discriminators 217 GFLOP per image pair:
        # prototype_full (2 * 103.8 + 8 * 20.6 + 217) / 2 = 294.7 GFLOP per counted image
        executed = {("mobilenet", "source_only"): 103.8, ("mobilenet", "prototype_full"): 294.7}.get((args.backbone, args.workload))
        if gflop is not None:
            tf = line["value"] / world * gflop / 1e3
            line["step_roofline"] = {"algorithmic_gflop_per_image": gflop, "achieved_tflops_per_gpu": round(tf, 2),
                                     "mfma_frac": round(tf / 157.3, 4),
                                     "note": "algorithmic = the REFERENCE's algorithm per counted image (SURVEY.md 8d: full-resolution "
                                             "decoder conv over the upsampled channels, 4 MC passes as full forwards); the product computes "
                                             "the same results with fewer executed FLOPs (low-resolution tap GEMMs + interpolation, "
                                             "MC passes reuse the deterministic part), so this is throughput in reference-FLOP units, "
                                             "not matrix-pipe utilisation - the kernel-level 'roofline' object is the utilisation figure"}
            if executed is not None and not args.use_tn:
                tfe = line["value"] / world * executed / 1e3
                line["step_roofline"].update(executed_gflop_per_image=executed, executed_tflops_per_gpu=round(tfe, 2),
                                             executed_mfma_frac=round(tfe / 157.3, 4))
            # SURVEY.md 8d reporting rule: the HBM fraction beside the FLOP-based one.  Algorithmic bytes per counted image:
            # 1.39 GB per training pass of the generator (source_only); prototype_full = (2 generator training passes * 1.39 +
            # 8 MC image-passes * 0.46 GB forward + two discriminators ~0.6 GB per image pair: ten passes over their ~32 MB of layer
            # outputs, written once and read once) / 2 = 3.5 GB
            gb_img = {("mobilenet", "source_only"): 1.39, ("mobilenet", "prototype_full"): 3.5}.get((args.backbone, args.workload))
            if gb_img is not None:
                gbs = line["value"] / world * gb_img
                line["step_roofline"].update(algorithmic_gb_per_image=gb_img, achieved_gb_s_per_gpu=round(gbs, 1),
                                             hbm_frac=round(gbs / 8000.0, 4))
        if world == 1 and not args.no_cpu_baseline and not args.use_tn:
            line["cpu_baseline"] = cpu_baseline(args.workload, 2, args.size, args.backbone)
        line["other_mfma"] = other
        line["peak_hbm_gib"] = round(peak_gib, 2)          # torch allocator peak of this rank up to the end of the timed region (of 288)
        line["host_input"] = host
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
