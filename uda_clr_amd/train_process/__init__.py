"""Trainer classes with the reference's module names (``train_use_fix_initial.py:11`` imports
``Trainer, Trainer_baseline, Trainer_prototype_full`` from here).  Importing this package joins the
data-parallel process group when launched one-process-per-GPU, so the unchanged entry script's
``.cuda()`` calls land on the right device."""
from ._common import bootstrap_from_env

bootstrap_from_env()

from . import Trainer, Trainer_baseline, Trainer_prototype_full  # noqa: E402,F401
