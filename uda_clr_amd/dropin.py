"""Run the reference's own entry script on this implementation, unchanged.

    python -m uda_clr_amd.dropin /path/to/UDA_CLR/train_use_fix_initial.py --method prototype_full --use_pid ...

The reference imports its packages by top-level name (``networks``, ``train_process``, ``utils``,
``dataloaders``; train_use_fix_initial.py:11-18).  ``install()`` publishes this package's modules
under those names in ``sys.modules`` BEFORE the script runs, so every ``from networks.deeplabv3 import *``
resolves here and nothing of the reference's own tree is imported.  Optional third-party modules that
the script imports at top level but this image lacks (torchvision.transforms.Compose) get minimal
stand-ins.  Data-parallel: launch the same command under ``python -m torch.distributed.run
--nproc-per-node N``; importing ``train_process`` joins the process group and picks the GPU.
"""
import importlib
import runpy
import sys
import types

_MAP = {
    "networks": "uda_clr_amd.networks",
    "networks.deeplabv3": "uda_clr_amd.networks.deeplabv3",
    "networks.GAN": "uda_clr_amd.networks.GAN",
    "networks.aspp": "uda_clr_amd.networks.aspp",
    "networks.decoder": "uda_clr_amd.networks.decoder",
    "networks.backbone": "uda_clr_amd.networks.backbone",
    "networks.sync_batchnorm": "uda_clr_amd.networks.sync_batchnorm",
    "networks.sync_batchnorm.batchnorm": "uda_clr_amd.networks.sync_batchnorm.batchnorm",
    "train_process": "uda_clr_amd.train_process",
    "train_process.Trainer": "uda_clr_amd.train_process.Trainer",
    "train_process.Trainer_baseline": "uda_clr_amd.train_process.Trainer_baseline",
    "train_process.Trainer_prototype_full": "uda_clr_amd.train_process.Trainer_prototype_full",
    "utils": "uda_clr_amd.utils",
    "utils.Utils": "uda_clr_amd.utils.Utils",
    "utils.metrics": "uda_clr_amd.utils.metrics",
    "dataloaders": "uda_clr_amd.dataloaders",
    "dataloaders.fundus_dataloader": "uda_clr_amd.dataloaders.fundus_dataloader",
    "dataloaders.custom_transforms": "uda_clr_amd.dataloaders.custom_transforms",
    "mypath": "uda_clr_amd.dataloaders.mypath",
}


def install():
    for alias, real in _MAP.items():
        try:
            sys.modules[alias] = importlib.import_module(real)
        except ImportError as e:           # e.g. the data pipeline is not built yet
            sys.stderr.write("uda_clr_amd.dropin: %s unavailable (%s)\n" % (alias, e))
    try:
        import torchvision  # noqa: F401
    except Exception:  # noqa: BLE001
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")

        class Compose:
            def __init__(self, transforms): self.transforms = transforms
            def __call__(self, x):
                for t in self.transforms:
                    x = t(x)
                return x
        tr.Compose = Compose
        ut = types.ModuleType("torchvision.utils")
        ut.make_grid = lambda t, *a, **k: t
        tv.transforms, tv.utils = tr, ut
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.utils": ut})


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
